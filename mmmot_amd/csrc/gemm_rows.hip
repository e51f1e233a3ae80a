// Row GEMM with fused operand generation and a GroupNorm-statistics epilogue
// (see include/mmmot_hip.h: mmmot_gemm_rows).
//
//   v[r][n] = sum_k A(r,k) W[n][k] + bias[n] + dbias[rowidx[r]][n]
//
// A modes: PLAIN (X), NORM_RELU (relu(X*sc+sh): the previous layer's GroupNorm
// + ReLU applied while staging the tile), PAIR (op(FA[i], FB[j]) generated on
// the fly - the 3x512xNxM tensor of reference modules/gcn.py:6-41 never reaches
// HBM).  Epilogue: per-tile per-channel sum / tile-centred M2 of v (input of
// mmmot_gn_finalize), activation, store.
//
// Two arithmetic paths behind one template:
//   F16 = false : exact fp32 MFMA (v_mfma_f32_32x32x2_f32), BK = 32;
//   F16 = true  : fp16 matrix cores with the 3-term hi/lo split of conv3x3_hl16.hip
//                 (a_hi*w_hi + a_hi*w_lo + a_lo*w_hi on v_mfma_f32_32x32x16_f16, fp32 accumulate),
//                 BK = 64.  W arrives pre-split (hl16 format, host-scaled by 1/oscale); the fp32
//                 A operand is produced by the prologue and split while it is staged to LDS.
#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define G16_BK 64
#define G16_LDT 72  // halves per LDS row (144 B): conflict-free ds_read_b128

template <int BN, int AMODE, bool F16>
__global__ __launch_bounds__(MM_THREADS, 2) void gemm_rows_kernel(mmmot_gemm_args a, int ntn) {
  constexpr int WM = (BN == 128) ? 2 : 4;
  constexpr int WN = 4 / WM;
  constexpr int TM = MM_BM / (WM * 32);
  constexpr int TN = BN / (WN * 32);
  constexpr int BLD = BN / 32;
  constexpr int BK = F16 ? G16_BK : MM_BK;
  constexpr int KV = BK / 32;  // f32x4 loads per thread-row per stage (8 lanes cover a row's BK floats)
  constexpr int SMEM_BYTES = F16 ? (MM_BM + BN) * G16_LDT * 2 * 2 : (MM_BM + BN) * MM_LDT * 4;
  // wide tiles of the f16 path leave through LDS as 16-byte stores (the narrow 64-column instantiation
  // keeps the direct per-lane stores: hipcc spilled the staged version, and those layers are tiny)
  constexpr bool STAGED_OUT = F16 && BN == 128;

  __shared__ __attribute__((aligned(16))) unsigned char smem_raw[SMEM_BYTES];
  float* smem = reinterpret_cast<float*>(smem_raw);
  // fp32 path
  float* As = smem;
  float* Bs = smem + MM_BM * MM_LDT;
  // f16 path: [A hi][A lo][B hi][B lo] planes of halves
  _Float16* hs = reinterpret_cast<_Float16*>(smem_raw);
  _Float16* As_hi = hs;
  _Float16* As_lo = hs + MM_BM * G16_LDT;
  _Float16* Bs_hi = hs + 2 * MM_BM * G16_LDT;
  _Float16* Bs_lo = hs + 2 * MM_BM * G16_LDT + BN * G16_LDT;

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
  const int koff = kq * (BK / 8);  // first k (floats) of this thread inside the slab

  // per-thread source rows of its 4 staging rows (invalid rows read row 0 of the tile: always mapped,
  // and are zeroed when staged - keeps the loads branch-free)
  const float* pa[4];
  const float* pb[4];
  bool rval[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = lrow + 32 * i;
    rval[i] = r < nrows;
    const int rr = rval[i] ? r : 0;
    pb[i] = nullptr;
    if constexpr (AMODE == MMMOT_A_PAIR) {
      const int q = row0 + rr - a.grp_row0[grp];
      const int M = a.grp_M[grp];
      const int ii = q / M, jj = q - ii * M;
      pa[i] = a.FA + (long)(a.grp_aoff[grp] + ii) * a.ldf + koff;
      pb[i] = a.FB + (long)(a.grp_boff[grp] + jj) * a.ldf + koff;
    } else {
      pa[i] = a.X + (long)(row0 + rr) * a.ldx + koff;
    }
  }
  const float* psc = nullptr;
  const float* psh = nullptr;
  if constexpr (AMODE == MMMOT_A_NORM_RELU) {
    psc = a.sc + (long)grp * a.ldsc + koff;
    psh = a.sh + (long)grp * a.ldsc + koff;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[tm][tn][e] = 0.f;

  f32x4 ra[4][KV];      // raw A (and FB for PAIR in rb2) of the next stage
  f32x4 rb2[4][KV];
  f32x4 rs[KV], rh[KV];  // scale / shift of the next stage (NORM_RELU)
  f32x4 rw32[BLD];       // fp32 weights (F16 = false)
  u32x4 rw16[BLD][2];    // hl16 weights (F16 = true): hi8, lo8

  auto load_stage = [&](int it) {
    const int k0 = it * BK;
    if constexpr (AMODE == MMMOT_A_NORM_RELU) {
#pragma unroll
      for (int v = 0; v < KV; ++v) {
        rs[v] = *reinterpret_cast<const f32x4*>(psc + k0 + 4 * v);
        rh[v] = *reinterpret_cast<const f32x4*>(psh + k0 + 4 * v);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int v = 0; v < KV; ++v) {
        ra[i][v] = *reinterpret_cast<const f32x4*>(pa[i] + k0 + 4 * v);
        if constexpr (AMODE == MMMOT_A_PAIR) rb2[i][v] = *reinterpret_cast<const f32x4*>(pb[i] + k0 + 4 * v);
      }
    if constexpr (F16) {
      // hl16 weight row n: K/8 units of [hi8 | lo8]; this thread's unit = (k0 + koff) / 8
      const u32x4* wp = reinterpret_cast<const u32x4*>(a.W);
      const long ku = (long)(a.K >> 3);
#pragma unroll
      for (int i = 0; i < BLD; ++i) {
        const u32x4* p = wp + ((long)(n0 + lrow + 32 * i) * ku + ((k0 + koff) >> 3)) * 2;
        rw16[i][0] = p[0];
        rw16[i][1] = p[1];
      }
    } else {
#pragma unroll
      for (int i = 0; i < BLD; ++i)
        rw32[i] = *reinterpret_cast<const f32x4*>(a.W + (long)(n0 + lrow + 32 * i) * a.K + k0 + koff);
    }
  };

  // A(r, k) of the staged registers -> fp32 values (prologue), zero for rows beyond the tile
  auto a_value = [&](int i, int v) -> f32x4 {
    f32x4 x = ra[i][v];
    if constexpr (AMODE == MMMOT_A_NORM_RELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = fmaxf(fmaf(x[e], rs[v][e], rh[v][e]), 0.f);
    } else if constexpr (AMODE == MMMOT_A_PAIR) {
      const f32x4 u = rb2[i][v];
      if (a.pairop == MMMOT_PAIR_MULTIPLY) {
        x = x * u;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = (x[e] - u[e]) * 0.5f;
          x[e] = (a.pairop == MMMOT_PAIR_MINUS_ABS) ? fabsf(d) : d;
        }
      }
    }
    if constexpr (!F16) {  // (the f16 path zeroes rows beyond the tile with its range clamp: see the staging loop)
      if (!rval[i]) x = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    return x;
  };

  const int nk = a.K / BK;
  const int lr = lane & 31;
  load_stage(0);
  for (int it = 0; it < nk; ++it) {
    if constexpr (F16) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 x0 = a_value(i, 0), x1 = a_value(i, 1);
        // fp16 range clamp and the zeroing of rows beyond the tile in ONE v_med3_f32 per value (bound 0 for such rows;
        // NORM_RELU values are >= 0 already, so the lower bound does no harm there)
        const float top = rval[i] ? 65000.f : 0.f;
        u32x4 h, l;
        float y[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          y[e] = __builtin_amdgcn_fmed3f(x0[e], -top, top);
          y[4 + e] = __builtin_amdgcn_fmed3f(x1[e], -top, top);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {  // lo = (f16)(y - (float)hi) as one v_fma_mix per value (common.h)
          unsigned h2, l2;
          mm_split2(y[2 * e], y[2 * e + 1], h2, l2);
          h[e] = h2;
          l[e] = l2;
        }
        *reinterpret_cast<u32x4*>(&As_hi[(lrow + 32 * i) * G16_LDT + kq * 8]) = h;
        *reinterpret_cast<u32x4*>(&As_lo[(lrow + 32 * i) * G16_LDT + kq * 8]) = l;
      }
#pragma unroll
      for (int i = 0; i < BLD; ++i) {
        *reinterpret_cast<u32x4*>(&Bs_hi[(lrow + 32 * i) * G16_LDT + kq * 8]) = rw16[i][0];
        *reinterpret_cast<u32x4*>(&Bs_lo[(lrow + 32 * i) * G16_LDT + kq * 8]) = rw16[i][1];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<f32x4*>(&As[(lrow + 32 * i) * MM_LDT + kq * 4]) = a_value(i, 0);
#pragma unroll
      for (int i = 0; i < BLD; ++i)
        *reinterpret_cast<f32x4*>(&Bs[(lrow + 32 * i) * MM_LDT + kq * 4]) = rw32[i];
    }
    __syncthreads();
    if (it + 1 < nk) load_stage(it + 1);
    if constexpr (F16) {
      const int kh = (lane >> 5) * 8;
#pragma unroll
      for (int k16 = 0; k16 < G16_BK / 16; ++k16) {
        f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
          const int off = (wm * TM * 32 + tm * 32 + lr) * G16_LDT + k16 * 16 + kh;
          ah[tm] = *reinterpret_cast<const f16x8*>(&As_hi[off]);
          al[tm] = *reinterpret_cast<const f16x8*>(&As_lo[off]);
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          const int off = (wn * TN * 32 + tn * 32 + lr) * G16_LDT + k16 * 16 + kh;
          bh[tn] = *reinterpret_cast<const f16x8*>(&Bs_hi[off]);
          bl[tn] = *reinterpret_cast<const f16x8*>(&Bs_lo[off]);
        }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) {
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[tm], bh[tn], acc[tm][tn], 0, 0, 0);
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tm], bl[tn], acc[tm][tn], 0, 0, 0);
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tm], bh[tn], acc[tm][tn], 0, 0, 0);
          }
      }
    } else {
      mm_stage<TM, TN>(As, Bs, acc, wm * TM * 32, wn * TN * 32, lane);
    }
    __syncthreads();
  }

  // ---- epilogue -----------------------------------------------------------
  // Statistics are tile-centred (robust when |mean| >> std, SURVEY section 7 "hard parts"):
  //   part[t][0][n] = S  = sum_r v[r][n]            over the tile's valid rows
  //   part[t][1][n] = M2 = sum_r (v[r][n] - S/nrows)^2
  // mmmot_gn_finalize merges the tiles with the parallel-variance formula of Chan et al. in fp64.
  const float oscale = F16 ? a.oscale : 1.f;
  // Order of the sums (shared with gemm_wide.hip, which must reproduce part[] bit for bit): per 32-row block of the tile the
  // lane's 16 values in register order, then its two lane halves; per tile (s0 + s1) + (s2 + s3) over the four blocks.
  float* red = smem;                 // [4 row blocks][BN] column partials (LDS reuse: all waves are past the last stage)
  float* colmean = smem + 4 * BN;    // [BN]
  static_assert(WM * TM == 4, "four 32-row blocks per tile");
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int cl = wn * TN * 32 + tn * 32 + (lane & 31);  // column inside the tile
    const int n = n0 + cl;
    const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      float s1 = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int r = wm * TM * 32 + tm * 32 + mm_acc_row(e, lane);
        if (r < nrows) {
          float v = fmaf(acc[tm][tn][e], oscale, bv);
          if (a.dbias) v += a.dbias[(long)a.rowidx[row0 + r] * a.lddb + n];
          acc[tm][tn][e] = v;
          s1 += v;
          if constexpr (!STAGED_OUT) {
            if (a.Y) a.Y[(long)(row0 + r) * a.ldy + n] = mm_act(v, a.act);
          }
        }
      }
      if (a.part) {
        s1 = mm_xor32_sum(s1);
        if (lane < 32) red[(wm * TM + tm) * BN + cl] = s1;
      }
    }
  }
  if (a.part) {
    __syncthreads();
    for (int cl = tid; cl < BN; cl += MM_THREADS) {
      const float s = (red[cl] + red[BN + cl]) + (red[2 * BN + cl] + red[3 * BN + cl]);
      a.part[((long)t * 2 + 0) * a.N + n0 + cl] = s;
      colmean[cl] = s / (float)nrows;
    }
    __syncthreads();
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int cl = wn * TN * 32 + tn * 32 + (lane & 31);
      const float mu = colmean[cl];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        float s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int r = wm * TM * 32 + tm * 32 + mm_acc_row(e, lane);
          if (r < nrows) {
            const float d = acc[tm][tn][e] - mu;
            s2 = fmaf(d, d, s2);
          }
        }
        s2 = mm_xor32_sum(s2);
        if (lane < 32) red[(wm * TM + tm) * BN + cl] = s2;  // red[] was consumed before the barrier above
      }
    }
    __syncthreads();
    for (int cl = tid; cl < BN; cl += MM_THREADS) {
      const float s = (red[cl] + red[BN + cl]) + (red[2 * BN + cl] + red[3 * BN + cl]);
      a.part[((long)t * 2 + 1) * a.N + n0 + cl] = s;
    }
  }
  // Fused consumer of the next GroupNorm: column sums of relu(v * osc + osh) over the tile's rows.
  if (a.colsum) {
    __syncthreads();  // red[] may still be read by the statistics path / the last K stage
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int cl = wn * TN * 32 + tn * 32 + (lane & 31);
      const float os = a.osc[(long)grp * a.ldosc + n0 + cl], oh = a.osh[(long)grp * a.ldosc + n0 + cl];
      float s3 = 0.f;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int r = wm * TM * 32 + tm * 32 + mm_acc_row(e, lane);
          if (r < nrows) s3 += fmaxf(fmaf(acc[tm][tn][e], os, oh), 0.f);
        }
      }
      s3 = mm_xor32_sum(s3);
      if (lane < 32) red[wm * BN + cl] = s3;
    }
    __syncthreads();
    for (int cl = tid; cl < BN; cl += MM_THREADS) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) s += red[w * BN + cl];
      a.colsum[(long)t * a.N + n0 + cl] = s;
    }
  }
  // F16 path: the output tile goes through LDS (fp32 [128][BN+4]) and leaves as 16-byte stores, 512
  // contiguous bytes per row.  The per-lane 4-byte stores of the fp32 path above made the short-K
  // PointNet layers (K = 64 / 128, N = 512 / 1024) store-issue bound: 1.5 TB/s of output.
  if constexpr (STAGED_OUT) {
    if (a.Y) {
      constexpr int CLD = BN + 4;
      static_assert(MM_BM * CLD * 4 <= SMEM_BYTES, "output staging must fit the K-loop LDS");
      __syncthreads();  // statistics scratch (red / colmean) and the last stage are dead
      float* Cs = smem;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int e = 0; e < 16; ++e)
            Cs[(wm * TM * 32 + tm * 32 + mm_acc_row(e, lane)) * CLD + wn * TN * 32 + tn * 32 + (lane & 31)] =
                acc[tm][tn][e];  // acc[] holds v = scaled accumulator + biases for valid rows
      __syncthreads();
      constexpr int C4 = BN / 4;
      for (int w = tid; w < MM_BM * C4; w += MM_THREADS) {
        const int r = w / C4, c = (w - r * C4) * 4;
        if (r < nrows) {
          f32x4 v = *reinterpret_cast<const f32x4*>(&Cs[r * CLD + c]);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = mm_act(v[e], a.act);
          *reinterpret_cast<f32x4*>(&a.Y[(long)(row0 + r) * a.ldy + n0 + c]) = v;
        }
      }
    }
  }
}

template <int BN, int AMODE, bool F16>
static int launch_gemm(const mmmot_gemm_args* a, hipStream_t s) {
  const int ntn = a->N / BN;
  hipLaunchKernelGGL((gemm_rows_kernel<BN, AMODE, F16>), dim3(a->T * ntn), dim3(MM_THREADS), 0, s, *a, ntn);
  return mm_check(hipGetLastError());
}

template <bool F16>
static int dispatch_gemm(const mmmot_gemm_args* a, hipStream_t s) {
  const bool wide = (a->N % 128 == 0);
  switch (a->amode) {
    case MMMOT_A_PLAIN:
      return wide ? launch_gemm<128, MMMOT_A_PLAIN, F16>(a, s) : launch_gemm<64, MMMOT_A_PLAIN, F16>(a, s);
    case MMMOT_A_NORM_RELU:
      return wide ? launch_gemm<128, MMMOT_A_NORM_RELU, F16>(a, s) : launch_gemm<64, MMMOT_A_NORM_RELU, F16>(a, s);
    case MMMOT_A_PAIR:
      return wide ? launch_gemm<128, MMMOT_A_PAIR, F16>(a, s) : launch_gemm<64, MMMOT_A_PAIR, F16>(a, s);
  }
  return MMMOT_EINVAL;
}

// gemm_wide.hip: the K >= 256 layers of the pairwise block on a batch that fills the chip
int mmmot_gemm_wide_try(const mmmot_gemm_args* a, hipStream_t s, int* status);

extern "C" int mmmot_gemm_rows(const mmmot_gemm_args* a, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!a || !a->W || a->T <= 0 || !a->tile_row0 || !a->tile_nrows) return MMMOT_EINVAL;
  const int bk = a->w_hl16 ? G16_BK : MM_BK;
  if (a->K <= 0 || a->K % bk != 0 || a->N <= 0 || a->N % 64 != 0) return MMMOT_EINVAL;
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
  if (a->colsum && (!a->osc || !a->osh)) return MMMOT_EINVAL;
  if (a->w_hl16 && a->Y && (a->ldy % 4 != 0 || !mm_al16(a->Y))) return MMMOT_EINVAL;
  int status = MMMOT_OK;
  if (mmmot_gemm_wide_try(a, s, &status)) return status;
  return a->w_hl16 ? dispatch_gemm<true>(a, s) : dispatch_gemm<false>(a, s);
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
