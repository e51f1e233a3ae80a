// The "hl16" split-half storage format of the fp16-matrix-core arithmetic, and its converters.
//
// gfx950 has no TF32/xf32 and its exact fp32 MFMA runs at 1/16 of the f16 rate (157 vs 2500
// TFLOP/s).  Plain bf16/fp16 misses the 1e-3 output budget by 20-200x.  Here every
// fp32 value x is carried as two halves  hi = fp16(x), lo = fp16(x - hi)  (22 significand bits)
// and every product a*w is evaluated as  a_hi*w_hi + a_hi*w_lo + a_lo*w_hi  on
// v_mfma_f32_32x32x16_f16 with fp32 accumulation: 3 MFMAs per algorithmic tile-product, i.e. an
// effective ceiling of 2.5 PF / 3 = 833 TFLOP/s, 5.3x the fp32 MFMA, with relative error
// ~2^-21 per product (the dropped a_lo*w_lo term is ~2^-22).
//
// Storage format hl16 (same bytes as fp32): a row of C channels is C/8 units of 32 bytes,
// unit u = [hi of channels 8u..8u+7 (8 halves) | lo of channels 8u..8u+7 (8 halves)].
// Activations are written in this format by the producing layer's epilogue (split once, not 9x
// Cout/128 times in the consumer); weights are split on the host after scaling every output channel by a
// power of two so that their lo parts stay in the fp16 normal range (the epilogue multiplies by the
// inverse, exact).  The trunk kernel is conv3x3_hl16_patch.hip, the row GEMMs gemm_rows.hip / gemm_ares.hip.
#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void hl_split8(f32x8 v, u32x4& hi, u32x4& lo) {
  f16x8 h, l;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float x = fminf(fmaxf(v[e], -65000.f), 65000.f);  // stay finite in fp16 (activations are O(1..100))
    h[e] = (_Float16)x;
    l[e] = (_Float16)(x - (float)h[e]);
  }
  hi = __builtin_bit_cast(u32x4, h);
  lo = __builtin_bit_cast(u32x4, l);
}

// ---------------------------------------------------------------------------
// fp32 rows <-> hl16 rows (tests, tools; C % 8 == 0).
__global__ void hl16_pack_kernel(const float* __restrict__ x, u32x4* __restrict__ y, long nunits) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nunits) return;
  const f32x8 v = *reinterpret_cast<const f32x8*>(x + i * 8);
  u32x4 hi, lo;
  hl_split8(v, hi, lo);
  y[i * 2] = hi;
  y[i * 2 + 1] = lo;
}

__global__ void hl16_unpack_kernel(const u32x4* __restrict__ x, float* __restrict__ y, long nunits) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nunits) return;
  const f16x8 h = __builtin_bit_cast(f16x8, x[i * 2]), l = __builtin_bit_cast(f16x8, x[i * 2 + 1]);
#pragma unroll
  for (int e = 0; e < 8; ++e) y[i * 8 + e] = (float)h[e] + (float)l[e];
}

extern "C" int mmmot_hl16_pack(const float* x, void* y, long n, void* stream) {
  if (!x || !y || n <= 0 || n % 8 != 0 || !mm_al16(x) || !mm_al16(y)) return MMMOT_EINVAL;
  const long nu = n / 8;
  hipLaunchKernelGGL(hl16_pack_kernel, dim3((unsigned)((nu + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                     (u32x4*)y, nu);
  return mm_check(hipGetLastError());
}

extern "C" int mmmot_hl16_unpack(const void* x, float* y, long n, void* stream) {
  if (!x || !y || n <= 0 || n % 8 != 0 || !mm_al16(x) || !mm_al16(y)) return MMMOT_EINVAL;
  const long nu = n / 8;
  hipLaunchKernelGGL(hl16_unpack_kernel, dim3((unsigned)((nu + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const u32x4*)x, y, nu);
  return mm_check(hipGetLastError());
}

// ---------------------------------------------------------------------------
// Device-side power-of-two scaling for the training step (no host round trip): a tensor whose magnitude is only known
// on the device (a gradient: 1e-4 .. 1e-7; freshly updated weights) is multiplied by 2^(target - ex), amax = f * 2^ex
// with f in [0.5, 1), before the split, so that its lo halves stay normal fp16 numbers; mmmot_pow2_oscale builds the
// per-output-channel vector the consuming kernel multiplies its accumulators with to undo the scale(s) exactly.
__device__ __forceinline__ int hl_pow2_shift(const float* amax, int target) {
  return mm_pow2_shift(amax ? *amax : 0.f, target);
}

__global__ void hl16_pack_pow2_kernel(const float* __restrict__ x, u32x4* __restrict__ y, long nunits,
                                      const float* __restrict__ amax, int target) {
  const float sd = ldexpf(1.f, hl_pow2_shift(amax, target));
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nunits; i += (long)gridDim.x * blockDim.x) {
    f32x8 v = *reinterpret_cast<const f32x8*>(x + i * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= sd;
    u32x4 hi, lo;
    hl_split8(v, hi, lo);
    y[i * 2] = hi;
    y[i * 2 + 1] = lo;
  }
}

// y = hl16(x * 2^(target - ex(amax[0]))); amax: device scalar (mmmot_absmax), NULL = no scaling.  n % 8 == 0.
extern "C" int mmmot_hl16_pack_pow2(const float* x, void* y, long n, const float* amax, int target, void* stream) {
  if (!x || !y || n <= 0 || n % 8 != 0 || !mm_al16(x) || !mm_al16(y) || target < -60 || target > 15) return MMMOT_EINVAL;
  const long nu = n / 8;
  const long nb = (nu + 255) / 256;
  hipLaunchKernelGGL(hl16_pack_pow2_kernel, dim3((unsigned)(nb < 8192 ? nb : 8192)), dim3(256), 0, (hipStream_t)stream, x,
                     (u32x4*)y, nu, amax, target);
  return mm_check(hipGetLastError());
}

__global__ void pow2_oscale_kernel(float* __restrict__ out, int C, const float* __restrict__ amax_a, int target_a,
                                   const float* __restrict__ amax_b, int target_b) {
  // (each exponent is clamped to +-100 for degenerate maxima, mm_pow2_shift; their sum to +-126 so that the inverse scale
  // of two clamped operands stays a finite normal float - with such maxima exactness is moot, finiteness is not)
  int ex = -(hl_pow2_shift(amax_a, target_a) + hl_pow2_shift(amax_b, target_b));
  ex = ex < -126 ? -126 : (ex > 126 ? 126 : ex);
  const float v = ldexpf(1.f, ex);
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < C; c += gridDim.x * blockDim.x) out[c] = v;
}

// out[0..C) = 2^-(shift_a + shift_b): the inverse of the scales mmmot_hl16_pack_pow2 applied with the same (amax, target)
// pairs (either may be NULL = unscaled operand)
extern "C" int mmmot_pow2_oscale(float* out, int C, const float* amax_a, int target_a, const float* amax_b, int target_b,
                                 void* stream) {
  if (!out || C <= 0) return MMMOT_EINVAL;
  hipLaunchKernelGGL(pow2_oscale_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, out, C, amax_a, target_a,
                     amax_b, target_b);
  return mm_check(hipGetLastError());
}
