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

__device__ __forceinline__ float mm_act(float v, int act) {
  if (act == MMMOT_ACT_RELU) return fmaxf(v, 0.f);
  if (act == MMMOT_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
  return v;
}

__device__ __forceinline__ float mm_sigmoid(float v) { return 1.f / (1.f + expf(-v)); }

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
