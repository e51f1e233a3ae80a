// Row GEMM on the exact fp32 MFMA with fused operand generation and a
// GroupNorm-statistics epilogue (see include/mmmot_hip.h: mmmot_gemm_rows).
//
//   v[r][n] = sum_k A(r,k) W[n][k] + bias[n] + dbias[rowidx[r]][n]
//
// A modes: PLAIN (X), NORM_RELU (relu(X*sc+sh): the previous layer's GroupNorm
// + ReLU applied while staging the tile), PAIR (op(FA[i], FB[j]) generated on
// the fly - the 3x512xNxM tensor of reference modules/gcn.py:6-41 never reaches
// HBM).  Epilogue: per-tile per-channel sum / sum-of-squares of v (input of
// mmmot_gn_finalize), activation, store.
#include "common.h"

template <int BN, int AMODE>
__global__ __launch_bounds__(MM_THREADS, 2) void gemm_rows_kernel(mmmot_gemm_args a, int ntn) {
  constexpr int WM = (BN == 128) ? 2 : 4;
  constexpr int WN = 4 / WM;
  constexpr int TM = MM_BM / (WM * 32);
  constexpr int TN = BN / (WN * 32);
  constexpr int BLD = BN / 32;

  __shared__ __attribute__((aligned(16))) float smem[(MM_BM + BN) * MM_LDT];
  float* As = smem;
  float* Bs = smem + MM_BM * MM_LDT;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  const int lid = mm_xcd_remap(blockIdx.x, gridDim.x);
  const int t = lid / ntn, nt = lid % ntn;
  const int n0 = nt * BN;
  const int row0 = a.tile_row0[t];
  const int nrows = a.tile_nrows[t];
  const int grp = a.tile_group ? a.tile_group[t] : 0;

  const int lrow = tid >> 3;
  const int kq = tid & 7;

  // per-thread source rows of its 4 staging rows
  const float* pa[4];
  const float* pb[4];
  bool rval[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = lrow + 32 * i;
    rval[i] = r < nrows;
    pa[i] = nullptr;
    pb[i] = nullptr;
    if (rval[i]) {
      if constexpr (AMODE == MMMOT_A_PAIR) {
        const int q = row0 + r - a.grp_row0[grp];
        const int M = a.grp_M[grp];
        const int ii = q / M, jj = q - ii * M;
        pa[i] = a.FA + (long)(a.grp_aoff[grp] + ii) * a.ldf + kq * 4;
        pb[i] = a.FB + (long)(a.grp_boff[grp] + jj) * a.ldf + kq * 4;
      } else {
        pa[i] = a.X + (long)(row0 + r) * a.ldx + kq * 4;
      }
    }
  }
  const float* psc = nullptr;
  const float* psh = nullptr;
  if constexpr (AMODE == MMMOT_A_NORM_RELU) {
    psc = a.sc + (long)grp * a.ldsc + kq * 4;
    psh = a.sh + (long)grp * a.ldsc + kq * 4;
  }
  const float* pw = a.W + (long)(n0 + lrow) * a.K + kq * 4;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[tm][tn][e] = 0.f;

  f32x4 ra[4], rb[BLD];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  auto load_stage = [&](int it) {
    const int k0 = it * MM_BK;
    f32x4 s4 = zero4, h4 = zero4;
    if constexpr (AMODE == MMMOT_A_NORM_RELU) {
      s4 = *reinterpret_cast<const f32x4*>(psc + k0);
      h4 = *reinterpret_cast<const f32x4*>(psh + k0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 v = zero4;
      if (rval[i]) {
        v = *reinterpret_cast<const f32x4*>(pa[i] + k0);
        if constexpr (AMODE == MMMOT_A_NORM_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(v[e], s4[e], h4[e]), 0.f);
        } else if constexpr (AMODE == MMMOT_A_PAIR) {
          const f32x4 u = *reinterpret_cast<const f32x4*>(pb[i] + k0);
          if (a.pairop == MMMOT_PAIR_MULTIPLY) {
            v = v * u;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float d = (v[e] - u[e]) * 0.5f;
              v[e] = (a.pairop == MMMOT_PAIR_MINUS_ABS) ? fabsf(d) : d;
            }
          }
        }
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BLD; ++i) rb[i] = *reinterpret_cast<const f32x4*>(pw + (long)(32 * i) * a.K + k0);
  };

  const int nk = a.K / MM_BK;
  load_stage(0);
  for (int it = 0; it < nk; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<f32x4*>(&As[(lrow + 32 * i) * MM_LDT + kq * 4]) = ra[i];
#pragma unroll
    for (int i = 0; i < BLD; ++i)
      *reinterpret_cast<f32x4*>(&Bs[(lrow + 32 * i) * MM_LDT + kq * 4]) = rb[i];
    __syncthreads();
    if (it + 1 < nk) load_stage(it + 1);
    mm_stage<TM, TN>(As, Bs, acc, wm * TM * 32, wn * TN * 32, lane);
    __syncthreads();
  }

  // ---- epilogue -----------------------------------------------------------
  // Statistics are tile-centred (robust when |mean| >> std, SURVEY section 7 "hard parts"):
  //   part[t][0][n] = S  = sum_r v[r][n]            over the tile's valid rows
  //   part[t][1][n] = M2 = sum_r (v[r][n] - S/nrows)^2
  // mmmot_gn_finalize merges the tiles with the parallel-variance formula of Chan et al. in fp64.
  float* red = smem;                 // [WM][BN] per-wave column partials (LDS reuse: all waves are past the last stage)
  float* colmean = smem + WM * BN;   // [BN]
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int cl = wn * TN * 32 + tn * 32 + (lane & 31);  // column inside the tile
    const int n = n0 + cl;
    const float bv = a.bias ? a.bias[n] : 0.f;
    float s1 = 0.f;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int r = wm * TM * 32 + tm * 32 + mm_acc_row(e, lane);
        if (r < nrows) {
          float v = acc[tm][tn][e] + bv;
          if (a.dbias) v += a.dbias[(long)a.rowidx[row0 + r] * a.lddb + n];
          acc[tm][tn][e] = v;
          s1 += v;
          if (a.Y) a.Y[(long)(row0 + r) * a.ldy + n] = mm_act(v, a.act);
        }
      }
    }
    if (a.part) {
      s1 += __shfl_xor(s1, 32);
      if (lane < 32) red[wm * BN + cl] = s1;
    }
  }
  if (a.part) {
    __syncthreads();
    for (int cl = tid; cl < BN; cl += MM_THREADS) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) s += red[w * BN + cl];
      a.part[((long)t * 2 + 0) * a.N + n0 + cl] = s;
      colmean[cl] = s / (float)nrows;
    }
    __syncthreads();
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int cl = wn * TN * 32 + tn * 32 + (lane & 31);
      const float mu = colmean[cl];
      float s2 = 0.f;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int r = wm * TM * 32 + tm * 32 + mm_acc_row(e, lane);
          if (r < nrows) {
            const float d = acc[tm][tn][e] - mu;
            s2 += d * d;
          }
        }
      }
      s2 += __shfl_xor(s2, 32);
      if (lane < 32) red[wm * BN + cl] = s2;  // red[] was consumed before the barrier above
    }
    __syncthreads();
    for (int cl = tid; cl < BN; cl += MM_THREADS) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) s += red[w * BN + cl];
      a.part[((long)t * 2 + 1) * a.N + n0 + cl] = s;
    }
  }
}

template <int BN, int AMODE>
static int launch_gemm(const mmmot_gemm_args* a, hipStream_t s) {
  const int ntn = a->N / BN;
  hipLaunchKernelGGL((gemm_rows_kernel<BN, AMODE>), dim3(a->T * ntn), dim3(MM_THREADS), 0, s, *a, ntn);
  return mm_check(hipGetLastError());
}

extern "C" int mmmot_gemm_rows(const mmmot_gemm_args* a, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!a || !a->W || a->T <= 0 || !a->tile_row0 || !a->tile_nrows) return MMMOT_EINVAL;
  if (a->K <= 0 || a->K % MM_BK != 0 || a->N <= 0 || a->N % 64 != 0) return MMMOT_EINVAL;
  if (!mm_al16(a->W)) return MMMOT_EINVAL;
  if (a->amode == MMMOT_A_PAIR) {
    if (!a->FA || !a->FB || !a->grp_row0 || !a->grp_M || !a->grp_aoff || !a->grp_boff || !a->tile_group)
      return MMMOT_EINVAL;
    if (a->ldf % 4 != 0 || !mm_al16(a->FA) || !mm_al16(a->FB)) return MMMOT_EINVAL;
  } else {
    if (!a->X || a->ldx % 4 != 0 || !mm_al16(a->X)) return MMMOT_EINVAL;
    if (a->amode == MMMOT_A_NORM_RELU &&
        (!a->sc || !a->sh || a->ldsc % 4 != 0 || !mm_al16(a->sc) || !mm_al16(a->sh)))
      return MMMOT_EINVAL;
  }
  if (a->dbias && !a->rowidx) return MMMOT_EINVAL;
  const bool wide = (a->N % 128 == 0);
  switch (a->amode) {
    case MMMOT_A_PLAIN:
      return wide ? launch_gemm<128, MMMOT_A_PLAIN>(a, s) : launch_gemm<64, MMMOT_A_PLAIN>(a, s);
    case MMMOT_A_NORM_RELU:
      return wide ? launch_gemm<128, MMMOT_A_NORM_RELU>(a, s) : launch_gemm<64, MMMOT_A_NORM_RELU>(a, s);
    case MMMOT_A_PAIR:
      return wide ? launch_gemm<128, MMMOT_A_PAIR>(a, s) : launch_gemm<64, MMMOT_A_PAIR>(a, s);
  }
  return MMMOT_EINVAL;
}

// ---------------------------------------------------------------------------
// MFMA fragment-layout self test: one wave, C[32][32] = A[32][K] B[32][K]^T via
// the same mm_stage / accumulator mapping as the production kernels.
__global__ void selftest_mfma_kernel(const float* A, const float* B, float* C, int K) {
  __shared__ __attribute__((aligned(16))) float smem[64 * MM_LDT];
  float* As = smem;
  float* Bs = smem + 32 * MM_LDT;
  const int lane = threadIdx.x;
  f32x16 acc[1][1];
  for (int e = 0; e < 16; ++e) acc[0][0][e] = 0.f;
  for (int k0 = 0; k0 < K; k0 += MM_BK) {
    for (int idx = lane; idx < 32 * MM_BK; idx += 64) {
      const int r = idx / MM_BK, k = idx % MM_BK;
      As[r * MM_LDT + k] = (k0 + k < K) ? A[r * K + k0 + k] : 0.f;
      Bs[r * MM_LDT + k] = (k0 + k < K) ? B[r * K + k0 + k] : 0.f;
    }
    __syncthreads();
    mm_stage<1, 1>(As, Bs, acc, 0, 0, lane);
    __syncthreads();
  }
  for (int e = 0; e < 16; ++e) C[mm_acc_row(e, lane) * 32 + (lane & 31)] = acc[0][0][e];
}

extern "C" int mmmot_selftest_mfma(const float* A, const float* B, float* C, int K, void* stream) {
  if (!A || !B || !C || K <= 0) return MMMOT_EINVAL;
  hipLaunchKernelGGL(selftest_mfma_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, A, B, C, K);
  return mm_check(hipGetLastError());
}
