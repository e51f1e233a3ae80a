// Weight-gradient GEMM on the fp16 matrix cores (3-term hi/lo split): dW[n][k] = sum_r dY[r][n] * A(r, k),
// db[n] = sum_r dY[r][n] - the f16x3 counterpart of gemm_tn_kernel (backward.hip, exact fp32 MFMA with the row axis as
// the K of v_mfma_f32_32x32x2_f32: 2 rows per 64-cycle instruction and no operand reuse - 60 % of the pair block's
// backward time, rocprof round 3).  Here the row axis is the K = 16 of v_mfma_f32_32x32x16_f16: a chunk of 128 rows of dY
// (TN channels) and of A (TK channels) is staged TRANSPOSED into LDS as hi / lo fp16 planes ([channel][row], like the Gram
// kernel, gram.hip), so a lane's fragment - 8 consecutive rows of one channel - is one 16-byte read; 4 waves own 2 x 2
// quadrants of the TN x TK tile of dW (2 x 2 MFMA blocks each for 128 x 128).
// Range: gradients can be tiny (1e-6 and below: fp16 subnormals), so dY is multiplied by a power of two taken from the
// tensor's absolute maximum (mmmot_absmax: a device scalar, no host round trip) that puts the maximum at 2^10, and the
// result is scaled back exactly; A (activations, O(1)) is clamped to the fp16 range like everywhere else.
// Deterministic: rows are split into gridDim.z contiguous shares with their own partial dW / db (the caller adds them).
#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#define TF_ROWS 128
#define TF_LD 136  // halves per transposed LDS row: [channel][row], 272 B (conflict-free 16-byte fragment reads)

// absolute maximum of a [R][C] tensor (row stride ld) into *out (float bits; the caller zeroes it): atomicMax on the bit
// pattern of a non-negative float is order independent, so the result is deterministic
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ X, int ld, long R, int C,
                                                     unsigned int* __restrict__ out) {
  __shared__ float red[4];
  const int C4 = C >> 2;
  float m = 0.f;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < R * C4; idx += (long)gridDim.x * 256) {
    const long r = idx / C4;
    const int c = (int)(idx - r * C4) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(&X[r * ld + c]);
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
  m = wave_max(m);
  // one atomic per workgroup: four per workgroup from 1 024 workgroups on one address were most of the 50 us a call took
  // whatever the tensor's size (20 calls per training step)
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (m > 0.f) atomicMax(out, __float_as_uint(m));
  }
}

extern "C" int mmmot_absmax(const float* X, int ld, long R, int C, float* out, void* stream) {
  if (!X || !out || R <= 0 || C <= 0 || C % 4 != 0 || ld % 4 != 0 || !mm_al16(X)) return MMMOT_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(out, 0, sizeof(float), s);
  if (e != hipSuccess) return (int)e;
  const long n4 = R * (C / 4);
  const long want = (n4 + 2047) / 2048;  // eight 16-byte loads per thread before another workgroup is worth its atomic
  const int grid = (int)(want < 1 ? 1 : (want < 1024 ? want : 1024));
  hipLaunchKernelGGL(absmax_kernel, dim3(grid), dim3(256), 0, s, X, ld, R, C, reinterpret_cast<unsigned int*>(out));
  return mm_check(hipGetLastError());
}

template <int TN, int TK>
__global__ __launch_bounds__(256) void gemm_tn_f16_kernel(mmmot_gemm_tn_args a, const float* __restrict__ dyamax) {
  constexpr int WN = TN / 64, WK = TK / 64;  // 32x32 blocks per wave along n / k
  __shared__ __attribute__((aligned(16))) _Float16 Dh[TN * TF_LD];
  __shared__ __attribute__((aligned(16))) _Float16 Dl[TN * TF_LD];
  __shared__ __attribute__((aligned(16))) _Float16 Ah[TK * TF_LD];
  __shared__ __attribute__((aligned(16))) _Float16 Al[TK * TF_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1;
  const int lr = lane & 31, kh = (lane >> 5) * 8;
  const int n0 = blockIdx.x * TN, k0 = blockIdx.y * TK;
  // power-of-two scale of dY: max |dY| -> [2^10, 2^11)
  const float amax = dyamax ? *dyamax : 0.f;
  const int shift = mm_pow2_shift(amax, 11);  // amax = f * 2^ex, f in [0.5, 1): 11 - ex (guarded, common.h)
  const float sd = ldexpf(1.f, shift), inv_sd = ldexpf(1.f, -shift);

  const int t_lo = (int)((long)a.T * blockIdx.z / gridDim.z), t_hi = (int)((long)a.T * (blockIdx.z + 1) / gridDim.z);
  float* dW = a.dW + (long)blockIdx.z * a.N * a.K;
  float* db = a.db ? a.db + (long)blockIdx.z * a.N : nullptr;

  f32x16 tot[WN][WK];
#pragma unroll
  for (int x = 0; x < WN; ++x)
#pragma unroll
    for (int y = 0; y < WK; ++y)
#pragma unroll
      for (int e = 0; e < 16; ++e) tot[x][y][e] = 0.f;
  float bsum = 0.f;  // threads < TN: column sum of dY (scaled)

  // staging: thread -> (4 consecutive rows sr4 .. +3 of the chunk, a run of channels): 8 threads per row quad
  const int sr4 = (tid >> 3) * 4, sq = tid & 7;
  constexpr int CPN = TN / 8, CPK = TK / 8;  // channels per staging thread: 8 / 16

  for (int t = t_lo; t < t_hi; ++t) {
    const int row0 = a.tile_row0[t], nrows = a.tile_nrows[t];
    const int g = a.tile_group ? a.tile_group[t] : 0;
    // ---- dY rows (scaled) -> transposed hi / lo planes ----
#pragma unroll
    for (int c4 = 0; c4 < CPN; c4 += 4) {
      f32x4 x[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const bool rv = sr4 + rr < nrows;
        x[rr] = *reinterpret_cast<const f32x4*>(a.dY + (long)(row0 + (rv ? sr4 + rr : 0)) * a.lddy + n0 + sq * CPN + c4);
        if (!rv) x[rr] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        f16x4 hi, lo;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const float y = x[rr][e] * sd;
          hi[rr] = (_Float16)y;
          lo[rr] = (_Float16)(y - (float)hi[rr]);
        }
        const int c = sq * CPN + c4 + e;
        *reinterpret_cast<f16x4*>(&Dh[c * TF_LD + sr4]) = hi;
        *reinterpret_cast<f16x4*>(&Dl[c * TF_LD + sr4]) = lo;
      }
    }
    // ---- A rows (plain / relu(norm) / pairwise op) -> transposed hi / lo planes ----
    int pi[4], pj[4];
    const float* fa = nullptr;
    const float* fb = nullptr;
    if (a.amode == MMMOT_A_PAIR) {
      const int grow0 = a.grp_row0[g], gM = a.grp_M[g];
      fa = a.FA + (long)a.grp_aoff[g] * a.ldf;
      fb = a.FB + (long)a.grp_boff[g] * a.ldf;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int local = row0 + (sr4 + rr < nrows ? sr4 + rr : 0) - grow0;
        pi[rr] = local / gM;
        pj[rr] = local - pi[rr] * gM;
      }
    }
#pragma unroll
    for (int c4 = 0; c4 < CPK; c4 += 4) {
      const int kc = k0 + sq * CPK + c4;
      f32x4 s4 = {1.f, 1.f, 1.f, 1.f}, h4 = {0.f, 0.f, 0.f, 0.f};
      if (a.amode == MMMOT_A_NORM_RELU) {
        s4 = *reinterpret_cast<const f32x4*>(a.sc + (long)g * a.ldsc + kc);
        h4 = *reinterpret_cast<const f32x4*>(a.sh + (long)g * a.ldsc + kc);
      }
      f32x4 x[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const bool rv = sr4 + rr < nrows;
        if (a.amode == MMMOT_A_PAIR) {
          const f32x4 u = *reinterpret_cast<const f32x4*>(fa + (long)pi[rr] * a.ldf + kc);
          const f32x4 v = *reinterpret_cast<const f32x4*>(fb + (long)pj[rr] * a.ldf + kc);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            x[rr][e] = (a.pairop == MMMOT_PAIR_MULTIPLY) ? u[e] * v[e]
                       : (a.pairop == MMMOT_PAIR_MINUS_ABS ? fabsf(u[e] - v[e]) * 0.5f : (u[e] - v[e]) * 0.5f);
        } else {
          x[rr] = *reinterpret_cast<const f32x4*>(a.X + (long)(row0 + (rv ? sr4 + rr : 0)) * a.ldx + kc);
          if (a.amode == MMMOT_A_NORM_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) x[rr][e] = fmaxf(fmaf(x[rr][e], s4[e], h4[e]), 0.f);
          }
        }
        if (!rv) x[rr] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        f16x4 hi, lo;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const float y = fminf(fmaxf(x[rr][e], -65000.f), 65000.f);
          hi[rr] = (_Float16)y;
          lo[rr] = (_Float16)(y - (float)hi[rr]);
        }
        const int c = sq * CPK + c4 + e;
        *reinterpret_cast<f16x4*>(&Ah[c * TF_LD + sr4]) = hi;
        *reinterpret_cast<f16x4*>(&Al[c * TF_LD + sr4]) = lo;
      }
    }
    __syncthreads();
    // ---- bias gradient: thread c adds its dY channel's 128 rows (deterministic order), first k block only ----
    if (db && blockIdx.y == 0 && tid < TN) {
      float acc = 0.f;
#pragma unroll 4
      for (int r8 = 0; r8 < TF_ROWS; r8 += 8) {
        const f16x8 h = *reinterpret_cast<const f16x8*>(&Dh[tid * TF_LD + r8]);
        const f16x8 l = *reinterpret_cast<const f16x8*>(&Dl[tid * TF_LD + r8]);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += (float)h[e] + (float)l[e];
      }
      bsum += acc;
    }
    // ---- dW tile of the chunk on the matrix cores ----
    f32x16 acc[WN][WK];
#pragma unroll
    for (int x = 0; x < WN; ++x)
#pragma unroll
      for (int y = 0; y < WK; ++y)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[x][y][e] = 0.f;
#pragma unroll
    for (int k16 = 0; k16 < TF_ROWS / 16; ++k16) {
      f16x8 dh[WN], dl[WN], ah[WK], al[WK];
#pragma unroll
      for (int x = 0; x < WN; ++x) {
        const int off = ((wi * WN + x) * 32 + lr) * TF_LD + k16 * 16 + kh;
        dh[x] = *reinterpret_cast<const f16x8*>(&Dh[off]);
        dl[x] = *reinterpret_cast<const f16x8*>(&Dl[off]);
      }
#pragma unroll
      for (int y = 0; y < WK; ++y) {
        const int off = ((wj * WK + y) * 32 + lr) * TF_LD + k16 * 16 + kh;
        ah[y] = *reinterpret_cast<const f16x8*>(&Ah[off]);
        al[y] = *reinterpret_cast<const f16x8*>(&Al[off]);
      }
#pragma unroll
      for (int x = 0; x < WN; ++x)
#pragma unroll
        for (int y = 0; y < WK; ++y) {
          acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dl[x], ah[y], acc[x][y], 0, 0, 0);
          acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dh[x], al[y], acc[x][y], 0, 0, 0);
          acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dh[x], ah[y], acc[x][y], 0, 0, 0);
        }
    }
#pragma unroll
    for (int x = 0; x < WN; ++x)
#pragma unroll
      for (int y = 0; y < WK; ++y)
#pragma unroll
        for (int e = 0; e < 16; ++e) tot[x][y][e] += acc[x][y][e];
    __syncthreads();  // the planes are rewritten by the next chunk
  }
#pragma unroll
  for (int x = 0; x < WN; ++x)
#pragma unroll
    for (int y = 0; y < WK; ++y)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int n = n0 + (wi * WN + x) * 32 + mm_acc_row(e, lane);
        const int k = k0 + (wj * WK + y) * 32 + lr;
        dW[(long)n * a.K + k] = tot[x][y][e] * inv_sd;
      }
  if (db && blockIdx.y == 0 && tid < TN) db[n0 + tid] = bsum * inv_sd;
}

// same arguments as mmmot_gemm_tn plus dyamax: device pointer to max |dY| (mmmot_absmax), NULL = no scaling.
// N % 64 == 0, K % 64 == 0; tiles of at most 128 rows.
extern "C" int mmmot_gemm_tn_f16(const mmmot_gemm_tn_args* a, const float* dyamax, void* stream) {
  if (!a || !a->dY || !a->dW || !a->tile_row0 || !a->tile_nrows || a->T <= 0) return MMMOT_EINVAL;
  if (a->N <= 0 || a->K <= 0 || a->N % 64 != 0 || a->K % 64 != 0 || a->lddy % 4 != 0) return MMMOT_EINVAL;
  if (a->amode == MMMOT_A_PAIR) {
    if (!a->FA || !a->FB || !a->grp_row0 || !a->grp_M || !a->grp_aoff || !a->grp_boff || !a->tile_group) return MMMOT_EINVAL;
    if (a->pairop < MMMOT_PAIR_MULTIPLY || a->pairop > MMMOT_PAIR_MINUS || a->ldf % 4 != 0) return MMMOT_EINVAL;
  } else {
    if (!a->X || a->ldx % 4 != 0) return MMMOT_EINVAL;
    if (a->amode == MMMOT_A_NORM_RELU && (!a->sc || !a->sh || a->ldsc % 4 != 0)) return MMMOT_EINVAL;
    if (a->amode != MMMOT_A_PLAIN && a->amode != MMMOT_A_NORM_RELU) return MMMOT_EINVAL;
  }
  if (a->nsplit < 1 || a->nsplit > 1024) return MMMOT_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const bool n128 = a->N % 128 == 0, k128 = a->K % 128 == 0;
#define TF_LAUNCH(TNV, TKV)                                                                                             \
  hipLaunchKernelGGL((gemm_tn_f16_kernel<TNV, TKV>), dim3(a->N / TNV, a->K / TKV, a->nsplit), dim3(256), 0, s, *a, dyamax)
  if (n128 && k128) TF_LAUNCH(128, 128);
  else if (n128) TF_LAUNCH(128, 64);
  else if (k128) TF_LAUNCH(64, 128);
  else TF_LAUNCH(64, 64);
#undef TF_LAUNCH
  return mm_check(hipGetLastError());
}
