// Shared device helpers for the gfx950 kernels of libmmmot_hip.so.
// CDNA4 only: 64-lane wavefronts, v_mfma_f32_32x32x2_f32 (exact fp32 MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mmmot_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MM_BM 128          // rows (positions) per workgroup tile
#define MM_BK 32           // reduction depth per LDS stage
#define MM_LDT (MM_BK + 4) // LDS row stride in floats: 144 B keeps ds_read_b128 conflict-free
#define MM_THREADS 256

static inline int mm_check(hipError_t e) { return e == hipSuccess ? MMMOT_OK : (int)e; }
static inline bool mm_al16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// CU count of the current device (one process drives one GPU); 0 when there is none.  A function-local static:
// initialised once, thread-safe (C++11).
static inline int mm_num_cu() {
  static const int n = [] {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    return (int)prop.multiProcessorCount;
  }();
  return n;
}

// XCD-aware bijective remap of a linear workgroup id: the dispatcher places
// workgroup b on XCD b % 8; give every XCD a contiguous chunk of logical ids so
// that tiles sharing an A panel hit the same 4 MiB L2.
__device__ __forceinline__ int mm_xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, slot = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + slot;
}

// One BK=32 stage of the wave-level MMA: the wave owns TM x TN 32x32 output
// tiles.  As / Bs are [rows][MM_LDT] fp32 with the reduction axis contiguous.
// Fragment mapping of v_mfma_f32_32x32x2_f32: lane l supplies A[i=l&31][k=l>>5]
// and B[k=l>>5][j=l&31].  Each lane reads 4 consecutive k with one
// ds_read_b128 (k = kc*8 + 4*(l>>5) + j) and issues 4 MFMAs; the pairing of k
// values inside one MFMA is (j, j+4), identical for A and B, so the sum is the
// plain dot product.
template <int TM, int TN>
__device__ __forceinline__ void mm_stage(const float* __restrict__ As, const float* __restrict__ Bs,
                                         f32x16 (&acc)[TM][TN], int a_row0, int b_row0, int lane) {
  const int lr = lane & 31;
  const int kh = (lane >> 5) * 4;
#pragma unroll
  for (int kc = 0; kc < MM_BK / 8; ++kc) {
    f32x4 a[TM], b[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
      a[tm] = *reinterpret_cast<const f32x4*>(&As[(a_row0 + tm * 32 + lr) * MM_LDT + kc * 8 + kh]);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
      b[tn] = *reinterpret_cast<const f32x4*>(&Bs[(b_row0 + tn * 32 + lr) * MM_LDT + kc * 8 + kh]);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][j], b[tn][j], acc[tm][tn], 0, 0, 0);
  }
}

// C/D mapping of the 32x32 MFMA: lane l, register e holds
// row = (e&3) + 8*(e>>2) + 4*(l>>5), col = l&31.
__device__ __forceinline__ int mm_acc_row(int e, int lane) { return (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5); }

// ---- register epilogues of a 64-row x 32-channel wave tile (two 32x32 accumulators; a lane holds 32 of its column's
// 64 rows, lane ^ 32 the other 32).  The half-wave exchange is v_permlane32_swap (gfx950), a VALU instruction, instead of
// the ds_bpermute behind __shfl_xor(.., 32) (an LDS round trip twice per channel block).  Measured and not adopted for
// the 32-value sums: packed fp32 (v_pk_add_f32 / v_pk_fma_f32) tree reductions - fewer instructions, no faster
// (tools/bench_ares.py A/B on one box: 0.44 ms sequential vs 0.47 ms packed, statistics pass of conv1).
__device__ __forceinline__ float mm_xor32_sum(float x) {  // x + (x of lane ^ 32)
  // v_permlane32_swap_b32 a, b: the upper 32 lanes of a <-> the lower 32 lanes of b.  Inline assembly: with this
  // toolchain (ROCm 7.2 clang) the second result of __builtin_amdgcn_permlane32_swap comes back equal to the first
  // (probed on an MI355X), and the hazard recogniser does not see into inline assembly - the instruction needs wait
  // states after the VALU write of its operands and before a VALU read of its results (without the s_nops the upper
  // lanes read stale values).
  float a = x, b = x;
  asm volatile("s_nop 3\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 3" : "+v"(a), "+v"(b));
  return a + b;
}

__device__ __forceinline__ float mm_sum32(const f32x16& a, const f32x16& b) {  // sum of the lane's 32 values
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) s += a[e];
#pragma unroll
  for (int e = 0; e < 16; ++e) s += b[e];
  return s;
}

__device__ __forceinline__ float mm_m2_32(const f32x16& a, const f32x16& b, float mu) {  // sum of (v - mu)^2
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const float d = a[e] - mu;
    s = fmaf(d, d, s);
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const float d = b[e] - mu;
    s = fmaf(d, d, s);
  }
  return s;
}

__device__ __forceinline__ float mm_relu_sum32(const f32x16& a, const f32x16& b, float m1, float m0) {
  float s = 0.f;  // sum of relu(v * m1 + m0)
#pragma unroll
  for (int e = 0; e < 16; ++e) s += fmaxf(fmaf(a[e], m1, m0), 0.f);
#pragma unroll
  for (int e = 0; e < 16; ++e) s += fmaxf(fmaf(b[e], m1, m0), 0.f);
  return s;
}

// The lo halves of the two-term fp16 split of (y0, y1), given their rounded hi halves packed in one register:
// (f16)(y - (float)hi) per element.  v_fma_mix{lo,hi}_f16 computes an fp32 fma of mixed fp16 / fp32 sources and rounds the
// result to f16 into one half of the destination: ONE instruction per element where convert-back + subtract + convert take
// 2.5 (y - hi is exact in fp32, the rounding to f16 is the mode's RNE with subnormals kept, like v_cvt_f16_f32: bit for bit
// the three-instruction form - tests/test_gemm_f16_gpu.py compares a kernel that uses this against one that does not).
__device__ __forceinline__ unsigned mm_split_lo2(unsigned hi2, float y0, float y1) {
  unsigned lo2;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(lo2)
      : "v"(hi2), "v"(y0), "v"(y1));
  return lo2;
}

// two values -> their packed fp16 hi halves and lo halves (RNE both; three instructions: v_cvt_pk_f16_f32 + the two above)
__device__ __forceinline__ void mm_split2(float y0, float y1, unsigned& hi2, unsigned& lo2) {
  typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
  const h2_t h = {(_Float16)y0, (_Float16)y1};
  hi2 = __builtin_bit_cast(unsigned, h);
  lo2 = mm_split_lo2(hi2, y0, y1);
}

__device__ __forceinline__ float mm_act(float v, int act) {
  if (act == MMMOT_ACT_RELU) return fmaxf(v, 0.f);
  if (act == MMMOT_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
  return v;
}

__device__ __forceinline__ float mm_sigmoid(float v) { return 1.f / (1.f + expf(-v)); }

// Power-of-two scale exponent that puts amax = f * 2^ex (f in [0.5, 1)) at 2^target: target - ex, for the fp16-split
// operands of the training step (device-side scales: gradients of 1e-4 .. 1e-7, freshly updated weights).  Guarded
// (ADVICE r4): a zero, denormal-tiny, infinite or NaN amax must not turn into 2^(+-huge) = inf and 0 * inf = NaN in the
// products - a non-finite or non-positive amax means "unscaled", and the exponent is clamped to [-100, 100] (2^100 times
// a denormal is still finite; the inverse scale 2^-shift stays a normal float).
__device__ __forceinline__ int mm_pow2_shift(float amax, int target) {
  if (!(amax > 0.f) || !(amax <= 3.4028234e38f)) return 0;
  int ex = 0;
  (void)frexpf(amax, &ex);
  const int s = target - ex;
  return s < -100 ? -100 : (s > 100 ? 100 : s);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
