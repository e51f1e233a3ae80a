// Training backward of the pairwise block (first slice): the gradients of
// `affinity_module.forward` + `NewEndIndicator_v2.forward` + the softmax modes of `TrackingNet.associate`
// (reference modules/gcn.py:68-82, new_end.py:62-82, tracking_net.py:106-126; the training step that needs them is
// tracking_model.py:50-66).  In training mode this block behaves exactly like in eval mode (GroupNorm only, no
// BatchNorm, no dropout), so the forward is the inference forward with the pre-norm tensors kept.
//
// Layout as in the forward: position-major rows, channels contiguous.  Every layer of the block is
//     y = A W^T + b,   yhat = (y - mean) * rstd,   z = yhat * gamma + beta,   a = relu(z)
// with statistics per (group, norm group) over the group's rows.  The backward of one layer, given dA = dL/da:
//     dz = dA * [z > 0];  dgamma_c = sum_r dz*yhat;  dbeta_c = sum_r dz;
//     dy = rstd * (gamma*dz - m1 - yhat*m2),  m1 = mean(gamma*dz), m2 = mean(gamma*dz*yhat) over the norm set;
//     dA_in = dy W  (mmmot_gemm_rows with the transposed weight);   dW = dy^T A_in;   db = sum_r dy.
// yhat is recomputed from the stored y with sc1 = rstd, sh1 = -mean*rstd (mmmot_gn_finalize with gamma = 1,
// beta = 0).  Kernels here are plain fp32 (exact fp32 MFMA for dW): this slice is about correct gradients, checked
// against torch.autograd on the oracle (tests/test_backward_gpu.py), not yet about speed.
#include "common.h"

// ---------------------------------------------------------------------------
// GroupNorm(+ReLU) backward pass 1: per-tile partial sums.  One workgroup per tile, thread -> 4 channels.
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(
    const float* __restrict__ dA, int ldda, const float* __restrict__ Y, int ldy, int C,
    const float* __restrict__ sc1, const float* __restrict__ sh1, int ldsc, const float* __restrict__ gamma,
    const float* __restrict__ beta, int relu, const int* __restrict__ tile_row0, const int* __restrict__ tile_nrows,
    const int* __restrict__ tile_group, float* __restrict__ P) {
  const int t = blockIdx.x;
  const int row0 = tile_row0[t], nrows = tile_nrows[t], g = tile_group ? tile_group[t] : 0;
  const int C4 = C >> 2;
  if (C4 <= 128) {
    // narrow rows (64 .. 512 channels): 256 / (C / 4) rows at a time, the row phases added in a fixed order through LDS
    // (with one thread per 4 channels walking the whole tile, a 64-channel layer kept 16 threads of a workgroup busy)
    __shared__ __attribute__((aligned(16))) float red[2][256 * 4];
    const int RP = 256 / C4, tid = threadIdx.x, cq = tid % C4, ph = tid / C4, c = cq * 4;
    const f32x4 s = *reinterpret_cast<const f32x4*>(&sc1[(long)g * ldsc + c]);
    const f32x4 h = *reinterpret_cast<const f32x4*>(&sh1[(long)g * ldsc + c]);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(&gamma[c]);
    const f32x4 be = *reinterpret_cast<const f32x4*>(&beta[c]);
    f32x4 p0 = {0.f, 0.f, 0.f, 0.f}, p1 = p0;
    if (ph < RP)
      for (int r = ph; r < nrows; r += RP) {
        const f32x4 y = *reinterpret_cast<const f32x4*>(&Y[(long)(row0 + r) * ldy + c]);
        const f32x4 d = *reinterpret_cast<const f32x4*>(&dA[(long)(row0 + r) * ldda + c]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float yh = fmaf(y[e], s[e], h[e]);
          const float z = fmaf(yh, ga[e], be[e]);
          const float dz = (relu && !(z > 0.f)) ? 0.f : d[e];
          p0[e] += dz;
          p1[e] = fmaf(dz, yh, p1[e]);
        }
      }
    *reinterpret_cast<f32x4*>(&red[0][tid * 4]) = p0;
    *reinterpret_cast<f32x4*>(&red[1][tid * 4]) = p1;
    __syncthreads();
    if (tid < 2 * C4) {
      const int w = tid / C4, q = tid - w * C4;
      f32x4 tot = *reinterpret_cast<const f32x4*>(&red[w][q * 4]);
      for (int k = 1; k < RP; ++k) tot += *reinterpret_cast<const f32x4*>(&red[w][(k * C4 + q) * 4]);
      *reinterpret_cast<f32x4*>(&P[((long)t * 2 + w) * C + q * 4]) = tot;
    }
    return;
  }
  for (int c = (blockIdx.y * 256 + threadIdx.x) * 4; c < C; c += gridDim.y * 1024) {
    const f32x4 s = *reinterpret_cast<const f32x4*>(&sc1[(long)g * ldsc + c]);
    const f32x4 h = *reinterpret_cast<const f32x4*>(&sh1[(long)g * ldsc + c]);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(&gamma[c]);
    const f32x4 be = *reinterpret_cast<const f32x4*>(&beta[c]);
    f32x4 p0 = {0.f, 0.f, 0.f, 0.f}, p1 = p0;
    for (int r = 0; r < nrows; ++r) {
      const f32x4 y = *reinterpret_cast<const f32x4*>(&Y[(long)(row0 + r) * ldy + c]);
      const f32x4 d = *reinterpret_cast<const f32x4*>(&dA[(long)(row0 + r) * ldda + c]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float yh = fmaf(y[e], s[e], h[e]);
        const float z = fmaf(yh, ga[e], be[e]);
        const float dz = (relu && !(z > 0.f)) ? 0.f : d[e];
        p0[e] += dz;
        p1[e] = fmaf(dz, yh, p1[e]);
      }
    }
    *reinterpret_cast<f32x4*>(&P[((long)t * 2 + 0) * C + c]) = p0;
    *reinterpret_cast<f32x4*>(&P[((long)t * 2 + 1) * C + c]) = p1;
  }
}

extern "C" int mmmot_gn_bwd_partial(const float* dA, int ldda, const float* Y, int ldy, int C, const float* sc1,
                                    const float* sh1, int ldsc, const float* gamma, const float* beta, int relu,
                                    const int* tile_row0, const int* tile_nrows, const int* tile_group, int T,
                                    float* P, void* stream) {
  if (!dA || !Y || !sc1 || !sh1 || !gamma || !beta || !tile_row0 || !tile_nrows || !P || T <= 0) return MMMOT_EINVAL;
  if (C <= 0 || C % 4 != 0 || ldda % 4 != 0 || ldy % 4 != 0 || ldsc % 4 != 0) return MMMOT_EINVAL;
  if (!mm_al16(dA) || !mm_al16(Y) || !mm_al16(sc1) || !mm_al16(sh1) || !mm_al16(gamma) || !mm_al16(beta) || !mm_al16(P))
    return MMMOT_EINVAL;
  hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3(T, (C + 1023) / 1024), dim3(256), 0, (hipStream_t)stream, dA, ldda, Y,
                     ldy, C, sc1, sh1, ldsc, gamma, beta, relu, tile_row0, tile_nrows, tile_group, P);
  return mm_check(hipGetLastError());
}

// pass 2: S [G][2][C] = per-group sums of P over the group's tiles (mmmot_segment_mean with divisor 1) ->
// M [G][2][C]: m1 / m2 of the norm group a channel belongs to, broadcast to its channels.  One workgroup per
// (group, norm group); fp64 accumulation.
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const float* __restrict__ S,
                                                              const int* __restrict__ grp_count, int C, int NG,
                                                              const float* __restrict__ gamma, float* __restrict__ M) {
  __shared__ double red[2][4];
  const int g = blockIdx.x / NG, ng = blockIdx.x % NG;
  const int CG = C / NG, c0 = ng * CG;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double a0 = 0.0, a1 = 0.0;
  for (int c = c0 + tid; c < c0 + CG; c += 256) {
    a0 += (double)gamma[c] * (double)S[((long)g * 2 + 0) * C + c];
    a1 += (double)gamma[c] * (double)S[((long)g * 2 + 1) * C + c];
  }
  a0 = wave_sum_d(a0);
  a1 = wave_sum_d(a1);
  if (lane == 0) { red[0][wave] = a0; red[1][wave] = a1; }
  __syncthreads();
  const double cnt = (double)grp_count[g] * (double)CG;
  const float m1 = (float)((red[0][0] + red[0][1] + red[0][2] + red[0][3]) / cnt);
  const float m2 = (float)((red[1][0] + red[1][1] + red[1][2] + red[1][3]) / cnt);
  for (int c = c0 + tid; c < c0 + CG; c += 256) {
    M[((long)g * 2 + 0) * C + c] = m1;
    M[((long)g * 2 + 1) * C + c] = m2;
  }
}

extern "C" int mmmot_gn_bwd_finalize(const float* S, const int* grp_count, int G, int C, int NG, const float* gamma,
                                     float* M, void* stream) {
  if (!S || !grp_count || !gamma || !M || G <= 0 || C <= 0 || NG <= 0 || C % NG != 0) return MMMOT_EINVAL;
  hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(G * NG), dim3(256), 0, (hipStream_t)stream, S, grp_count, C, NG,
                     gamma, M);
  return mm_check(hipGetLastError());
}

// pass 3: dY = rstd * (gamma*dz - m1 - yhat*m2)
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(
    const float* __restrict__ dA, int ldda, const float* __restrict__ Y, int ldy, int C,
    const float* __restrict__ sc1, const float* __restrict__ sh1, int ldsc, const float* __restrict__ gamma,
    const float* __restrict__ beta, int relu, const float* __restrict__ M, const int* __restrict__ tile_row0,
    const int* __restrict__ tile_nrows, const int* __restrict__ tile_group, float* __restrict__ dY, int lddy) {
  const int t = blockIdx.x;
  const int row0 = tile_row0[t], nrows = tile_nrows[t], g = tile_group ? tile_group[t] : 0;
  // narrow rows (C <= 512): C / 4 threads per row, 256 / (C / 4) rows at a time (elementwise: the mapping does not touch
  // the values); wider rows: a thread per 4 channels walks the tile
  const int C4 = C >> 2;
  const bool narrow = C4 <= 128;
  const int RP = narrow ? 256 / C4 : 1;
  const int ph = narrow ? (int)threadIdx.x / C4 : 0;
  const int cbeg = narrow ? ((int)threadIdx.x % C4) * 4 : (blockIdx.y * 256 + threadIdx.x) * 4;
  const int cstep = narrow ? C : gridDim.y * 1024;
  if (narrow && ph >= RP) return;
  for (int c = cbeg; c < C; c += cstep) {
    const f32x4 s = *reinterpret_cast<const f32x4*>(&sc1[(long)g * ldsc + c]);
    const f32x4 h = *reinterpret_cast<const f32x4*>(&sh1[(long)g * ldsc + c]);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(&gamma[c]);
    const f32x4 be = *reinterpret_cast<const f32x4*>(&beta[c]);
    const f32x4 m1 = *reinterpret_cast<const f32x4*>(&M[((long)g * 2 + 0) * C + c]);
    const f32x4 m2 = *reinterpret_cast<const f32x4*>(&M[((long)g * 2 + 1) * C + c]);
    for (int r = ph; r < nrows; r += RP) {
      const f32x4 y = *reinterpret_cast<const f32x4*>(&Y[(long)(row0 + r) * ldy + c]);
      const f32x4 d = *reinterpret_cast<const f32x4*>(&dA[(long)(row0 + r) * ldda + c]);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float yh = fmaf(y[e], s[e], h[e]);
        const float z = fmaf(yh, ga[e], be[e]);
        const float dz = (relu && !(z > 0.f)) ? 0.f : d[e];
        o[e] = s[e] * (ga[e] * dz - m1[e] - yh * m2[e]);
      }
      *reinterpret_cast<f32x4*>(&dY[(long)(row0 + r) * lddy + c]) = o;
    }
  }
}

extern "C" int mmmot_gn_bwd_apply(const float* dA, int ldda, const float* Y, int ldy, int C, const float* sc1,
                                  const float* sh1, int ldsc, const float* gamma, const float* beta, int relu,
                                  const float* M, const int* tile_row0, const int* tile_nrows, const int* tile_group,
                                  int T, float* dY, int lddy, void* stream) {
  if (!dA || !Y || !sc1 || !sh1 || !gamma || !beta || !M || !tile_row0 || !tile_nrows || !dY || T <= 0) return MMMOT_EINVAL;
  if (C <= 0 || C % 4 != 0 || ldda % 4 != 0 || ldy % 4 != 0 || ldsc % 4 != 0 || lddy % 4 != 0) return MMMOT_EINVAL;
  if (!mm_al16(dA) || !mm_al16(Y) || !mm_al16(sc1) || !mm_al16(sh1) || !mm_al16(M) || !mm_al16(dY)) return MMMOT_EINVAL;
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(T, (C + 1023) / 1024), dim3(256), 0, (hipStream_t)stream, dA, ldda, Y, ldy,
                     C, sc1, sh1, ldsc, gamma, beta, relu, M, tile_row0, tile_nrows, tile_group, dY, lddy);
  return mm_check(hipGetLastError());
}

// ---------------------------------------------------------------------------
// Weight gradient: dW[n][k] = sum_r dY[r][n] * A(r, k), db[n] = sum_r dY[r][n].
// A(r, k) is the layer's INPUT, regenerated like the forward's A operand: plain X, relu(X*sc + sh), or the
// pairwise op(FA_i, FB_j).  One workgroup = 4 waves = one 64 x 64 tile of dW (each wave 32 x 32), walking ALL rows:
// v_mfma_f32_32x32x2_f32 reduces two rows per instruction - lane l supplies dY[row + (l>>5)][n0 + (l&31)] and
// A(row + (l>>5), k0 + (l&31)), i.e. both operands are coalesced 128-byte row reads, no LDS.  Deterministic (no
// atomics).
__global__ __launch_bounds__(256) void gemm_tn_kernel(mmmot_gemm_tn_args a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = blockIdx.x * 64 + (wave >> 1) * 32, k0 = blockIdx.y * 64 + (wave & 1) * 32;
  const int lr = lane & 31, hf = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  float bsum = 0.f;
  const int n = n0 + lr, k = k0 + lr;
  // row split: workgroup z of gridDim.z walks its contiguous share of the tiles and writes its own partial
  // dW / db (the caller sums the gridDim.z partials: deterministic, no atomics)
  const int t_lo = (int)((long)a.T * blockIdx.z / gridDim.z), t_hi = (int)((long)a.T * (blockIdx.z + 1) / gridDim.z);
  float* dW = a.dW + (long)blockIdx.z * a.N * a.K;
  float* db = a.db ? a.db + (long)blockIdx.z * a.N : nullptr;
  for (int t = t_lo; t < t_hi; ++t) {
    const int row0 = a.tile_row0[t], nrows = a.tile_nrows[t];
    const int g = a.tile_group ? a.tile_group[t] : 0;
    float s = 1.f, h = 0.f;
    if (a.amode == MMMOT_A_NORM_RELU) {
      s = a.sc[(long)g * a.ldsc + k];
      h = a.sh[(long)g * a.ldsc + k];
    }
    int grow0 = 0, gM = 1;
    const float* fa = nullptr;
    const float* fb = nullptr;
    if (a.amode == MMMOT_A_PAIR) {
      grow0 = a.grp_row0[g];
      gM = a.grp_M[g];
      fa = a.FA + (long)a.grp_aoff[g] * a.ldf + k;
      fb = a.FB + (long)a.grp_boff[g] * a.ldf + k;
    }
    // branch-free (clamped index + select) so that the loads of several row pairs are in flight together
    for (int r = 0; r < nrows; r += 2) {
      const int rr = r + hf;
      const bool ok = rr < nrows;
      const long row = (long)row0 + (ok ? rr : 0);
      float dy = a.dY[row * a.lddy + n], av;
      if (a.amode == MMMOT_A_PAIR) {
        const int local = (int)(row - grow0);
        const int i = local / gM, j = local - i * gM;
        const float x = fa[(long)i * a.ldf], y = fb[(long)j * a.ldf];
        av = (a.pairop == MMMOT_PAIR_MULTIPLY) ? x * y : (a.pairop == MMMOT_PAIR_MINUS_ABS ? fabsf(x - y) * 0.5f : (x - y) * 0.5f);
      } else {
        av = a.X[row * a.ldx + k];
        if (a.amode == MMMOT_A_NORM_RELU) av = fmaxf(fmaf(av, s, h), 0.f);
      }
      dy = ok ? dy : 0.f;
      av = ok ? av : 0.f;
      bsum += dy;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(dy, av, acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) dW[(long)(n0 + mm_acc_row(e, lane)) * a.K + k0 + lr] = acc[e];
  if (db && blockIdx.y == 0 && (wave & 1) == 0) {
    bsum += __shfl_xor(bsum, 32);
    if (hf == 0) db[n] = bsum;
  }
}

extern "C" int mmmot_gemm_tn(const mmmot_gemm_tn_args* a, void* stream) {
  if (!a || !a->dY || !a->dW || !a->tile_row0 || !a->tile_nrows || a->T <= 0) return MMMOT_EINVAL;
  if (a->N <= 0 || a->K <= 0 || a->N % 64 != 0 || a->K % 64 != 0) return MMMOT_EINVAL;
  if (a->amode == MMMOT_A_PAIR) {
    if (!a->FA || !a->FB || !a->grp_row0 || !a->grp_M || !a->grp_aoff || !a->grp_boff || !a->tile_group) return MMMOT_EINVAL;
    if (a->pairop < MMMOT_PAIR_MULTIPLY || a->pairop > MMMOT_PAIR_MINUS) return MMMOT_EINVAL;
  } else {
    if (!a->X) return MMMOT_EINVAL;
    if (a->amode == MMMOT_A_NORM_RELU && (!a->sc || !a->sh)) return MMMOT_EINVAL;
    if (a->amode != MMMOT_A_PLAIN && a->amode != MMMOT_A_NORM_RELU) return MMMOT_EINVAL;
  }
  if (a->nsplit < 1 || a->nsplit > 1024) return MMMOT_EINVAL;
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(a->N / 64, a->K / 64, a->nsplit), dim3(256), 0, (hipStream_t)stream, *a);
  return mm_check(hipGetLastError());
}

// ---------------------------------------------------------------------------
// Backward of the pairwise operand generation (modules/gcn.py:6-41): x[(i,j)][c] = op(a_i[c], b_j[c]).
//   side 0: dF[aoff + i][c] += sum_j dX[(i,j)][c] * d op / d a;   side 1: dF[boff + j][c] += sum_i dX[(i,j)][c] * d op / d b
// One workgroup per (group, i) or (group, j); the rows written by one launch (one side) are distinct, so the
// accumulation into dF needs no atomics; the two sides are two launches (a middle frame of a >2-frame sample is
// the b side of one pair and the a side of the next).
__global__ __launch_bounds__(128) void pair_bwd_kernel(const float* __restrict__ dX, int lddx,
                                                       const float* __restrict__ F, int ldf, float* __restrict__ dF,
                                                       int lddf, int C, const int* __restrict__ grp_row0,
                                                       const int* __restrict__ grp_N, const int* __restrict__ grp_M,
                                                       const int* __restrict__ grp_aoff, const int* __restrict__ grp_boff,
                                                       const int* __restrict__ blk_group, const int* __restrict__ blk_idx,
                                                       int pairop, int side) {
  const int g = blk_group[blockIdx.x], idx = blk_idx[blockIdx.x];
  const int N = grp_N[g], M = grp_M[g];
  const long row0 = grp_row0[g];
  const float* Fa = F + (long)grp_aoff[g] * ldf;
  const float* Fb = F + (long)grp_boff[g] * ldf;
  for (int c = threadIdx.x * 4; c < C; c += 512) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (side == 0) {
      const f32x4 av = *reinterpret_cast<const f32x4*>(&Fa[(long)idx * ldf + c]);
      for (int j = 0; j < M; ++j) {
        const f32x4 d = *reinterpret_cast<const f32x4*>(&dX[(row0 + (long)idx * M + j) * lddx + c]);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(&Fb[(long)j * ldf + c]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float w = (pairop == MMMOT_PAIR_MULTIPLY) ? bv[e]
                          : (pairop == MMMOT_PAIR_MINUS_ABS ? (av[e] > bv[e] ? 0.5f : (av[e] < bv[e] ? -0.5f : 0.f)) : 0.5f);
          acc[e] = fmaf(d[e], w, acc[e]);
        }
      }
      f32x4* o = reinterpret_cast<f32x4*>(&dF[((long)grp_aoff[g] + idx) * lddf + c]);
      *o = *o + acc;
    } else {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(&Fb[(long)idx * ldf + c]);
      for (int i = 0; i < N; ++i) {
        const f32x4 d = *reinterpret_cast<const f32x4*>(&dX[(row0 + (long)i * M + idx) * lddx + c]);
        const f32x4 av = *reinterpret_cast<const f32x4*>(&Fa[(long)i * ldf + c]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float w = (pairop == MMMOT_PAIR_MULTIPLY) ? av[e]
                          : (pairop == MMMOT_PAIR_MINUS_ABS ? (av[e] > bv[e] ? -0.5f : (av[e] < bv[e] ? 0.5f : 0.f)) : -0.5f);
          acc[e] = fmaf(d[e], w, acc[e]);
        }
      }
      f32x4* o = reinterpret_cast<f32x4*>(&dF[((long)grp_boff[g] + idx) * lddf + c]);
      *o = *o + acc;
    }
  }
}

extern "C" int mmmot_pair_bwd(const float* dX, int lddx, const float* F, int ldf, float* dF, int lddf, int C,
                              const int* grp_row0, const int* grp_N, const int* grp_M, const int* grp_aoff,
                              const int* grp_boff, const int* blk_group, const int* blk_idx, int nblk, int pairop,
                              int side, void* stream) {
  if (!dX || !F || !dF || !grp_row0 || !grp_N || !grp_M || !grp_aoff || !grp_boff || !blk_group || !blk_idx) return MMMOT_EINVAL;
  if (nblk <= 0 || C <= 0 || C % 4 != 0 || lddx % 4 != 0 || ldf % 4 != 0 || lddf % 4 != 0 || side < 0 || side > 1) return MMMOT_EINVAL;
  if (pairop < MMMOT_PAIR_MULTIPLY || pairop > MMMOT_PAIR_MINUS || !mm_al16(dX) || !mm_al16(F) || !mm_al16(dF)) return MMMOT_EINVAL;
  hipLaunchKernelGGL(pair_bwd_kernel, dim3(nblk), dim3(128), 0, (hipStream_t)stream, dX, lddx, F, ldf, dF, lddf, C,
                     grp_row0, grp_N, grp_M, grp_aoff, grp_boff, blk_group, blk_idx, pairop, side);
  return mm_check(hipGetLastError());
}

// ---------------------------------------------------------------------------
// Backward of the strided means that make the new / end vectors (new_end.py:70-71): pair row (g, i, j) is one of the
// N rows averaged into "new" vector j and one of the M rows averaged into "end" vector i of its group:
//   dA[(g,i,j)][c] = dV[vrow0[g] + j][c] / N + dV[vrow0[g] + M + i][c] / M
__global__ __launch_bounds__(256) void pair_expand_bwd_kernel(const float* __restrict__ dV, int lddv,
                                                              float* __restrict__ dA, int ldda, int C,
                                                              const int* __restrict__ tile_row0,
                                                              const int* __restrict__ tile_nrows,
                                                              const int* __restrict__ tile_group,
                                                              const int* __restrict__ grp_row0,
                                                              const int* __restrict__ grp_N, const int* __restrict__ grp_M,
                                                              const int* __restrict__ grp_vrow0) {
  const int t = blockIdx.x;
  const int row0 = tile_row0[t], nrows = tile_nrows[t], g = tile_group[t];
  const int N = grp_N[g], M = grp_M[g], v0 = grp_vrow0[g];
  const float in = 1.f / (float)N, im = 1.f / (float)M;
  for (int idx = threadIdx.x; idx < nrows * (C / 4); idx += 256) {
    const int r = idx / (C / 4), c = (idx - r * (C / 4)) * 4;
    const int local = row0 + r - grp_row0[g];
    const int i = local / M, j = local - i * M;
    const f32x4 a = *reinterpret_cast<const f32x4*>(&dV[((long)v0 + j) * lddv + c]);
    const f32x4 b = *reinterpret_cast<const f32x4*>(&dV[((long)v0 + M + i) * lddv + c]);
    *reinterpret_cast<f32x4*>(&dA[(long)(row0 + r) * ldda + c]) = a * in + b * im;
  }
}

extern "C" int mmmot_pair_expand_bwd(const float* dV, int lddv, float* dA, int ldda, int C, const int* tile_row0,
                                     const int* tile_nrows, const int* tile_group, int T, const int* grp_row0,
                                     const int* grp_N, const int* grp_M, const int* grp_vrow0, void* stream) {
  if (!dV || !dA || !tile_row0 || !tile_nrows || !tile_group || !grp_row0 || !grp_N || !grp_M || !grp_vrow0) return MMMOT_EINVAL;
  if (T <= 0 || C <= 0 || C % 4 != 0 || lddv % 4 != 0 || ldda % 4 != 0 || !mm_al16(dV) || !mm_al16(dA)) return MMMOT_EINVAL;
  hipLaunchKernelGGL(pair_expand_bwd_kernel, dim3(T), dim3(256), 0, (hipStream_t)stream, dV, lddv, dA, ldda, C,
                     tile_row0, tile_nrows, tile_group, grp_row0, grp_N, grp_M, grp_vrow0);
  return mm_check(hipGetLastError());
}

// ---------------------------------------------------------------------------
// Backward of the 1-channel output layers (mmmot_rowdot): out[r] = act(sum_k a(r,k) w[k] + b), a = relu(X*sc + sh).
//   gpre[r] = gout[gidx ? gidx[r] : r] * act'(out[r]);  dA[r][k] = gpre[r] * w[k];
//   PW[t][k] = sum_{r in tile t} gpre[r] * a(r,k)  (k < K),  PW[t][K] = sum_r gpre[r]   (reduce over tiles: dw, db)
// One workgroup per tile, one wave per row (like the forward).
__global__ __launch_bounds__(256) void rowdot_bwd_kernel(
    const float* __restrict__ X, int ldx, int K, const float* __restrict__ w, float b, const float* __restrict__ sc,
    const float* __restrict__ sh, int ldsc, const int* __restrict__ tile_row0, const int* __restrict__ tile_nrows,
    const int* __restrict__ tile_group, int act, const float* __restrict__ gout, const int* __restrict__ gidx,
    float* __restrict__ dA, int ldda, float* __restrict__ PW, int ldpw) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [4][K + 4] per-wave partial dw (+ db at [K])
  const int t = blockIdx.x;
  const int row0 = tile_row0[t], nrows = tile_nrows[t];
  const int g = tile_group ? tile_group[t] : 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* mine = lds + wave * (K + 4);
  for (int k = lane; k < K + 4; k += 64) mine[k] = 0.f;
  for (int r = wave; r < nrows; r += 4) {
    const float* xr = X + (long)(row0 + r) * ldx;
    float acc = 0.f;
    for (int k = lane; k < K; k += 64) {
      const float a = fmaxf(fmaf(xr[k], sc[(long)g * ldsc + k], sh[(long)g * ldsc + k]), 0.f);
      acc = fmaf(a, w[k], acc);
    }
    const float pre = wave_sum(acc) + b;
    float gp = gout[gidx ? gidx[row0 + r] : row0 + r];
    if (act == MMMOT_ACT_SIGMOID) {
      const float s = mm_sigmoid(pre);
      gp *= s * (1.f - s);
    }
    for (int k = lane; k < K; k += 64) {
      const float a = fmaxf(fmaf(xr[k], sc[(long)g * ldsc + k], sh[(long)g * ldsc + k]), 0.f);
      mine[k] = fmaf(gp, a, mine[k]);
      dA[(long)(row0 + r) * ldda + k] = gp * w[k];
    }
    if (lane == 0) mine[K] += gp;
  }
  __syncthreads();
  for (int k = threadIdx.x; k <= K; k += 256)
    PW[(long)t * ldpw + k] = lds[k] + lds[(K + 4) + k] + lds[2 * (K + 4) + k] + lds[3 * (K + 4) + k];
}

extern "C" int mmmot_rowdot_bwd(const float* X, int ldx, int K, const float* w, float b, const float* sc,
                                const float* sh, int ldsc, const int* tile_row0, const int* tile_nrows,
                                const int* tile_group, int T, int act, const float* gout, const int* gidx, float* dA,
                                int ldda, float* PW, int ldpw, void* stream) {
  if (!X || !w || !sc || !sh || !tile_row0 || !tile_nrows || !gout || !dA || !PW || T <= 0) return MMMOT_EINVAL;
  if (K <= 0 || K > 2048 || ldpw < K + 1 || (act != MMMOT_ACT_NONE && act != MMMOT_ACT_SIGMOID)) return MMMOT_EINVAL;
  hipLaunchKernelGGL(rowdot_bwd_kernel, dim3(T), dim3(256), (size_t)4 * (K + 4) * sizeof(float), (hipStream_t)stream, X,
                     ldx, K, w, b, sc, sh, ldsc, tile_row0, tile_nrows, tile_group, act, gout, gidx, dA, ldda, PW, ldpw);
  return mm_check(hipGetLastError());
}

// ---------------------------------------------------------------------------
// Backward of mmmot_softmax_pairs (tracking_net.py:106-126).  p = softmax over j (per row i), q = softmax over i (per
// column j) of the logits of one N x M block; out = p | p*q | (p+q)/2 | max(p,q).
//   dl = p * (dp - sum_j dp*p) + q * (dq - sum_i dq*q),  (dp, dq) = (dO, 0) | (dO*q, dO*p) | (dO/2, dO/2) | (dO*[p>=q], dO*[q>p])
// One workgroup per block; row / column statistics and the two dot products in LDS.
__global__ __launch_bounds__(256) void softmax_pairs_bwd_kernel(const float* __restrict__ logits,
                                                                const float* __restrict__ dout,
                                                                float* __restrict__ dlogits,
                                                                const int* __restrict__ grp_row0,
                                                                const int* __restrict__ grp_N,
                                                                const int* __restrict__ grp_M, int mode) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int g = blockIdx.x;
  const int N = grp_N[g], M = grp_M[g];
  const float* x = logits + grp_row0[g];
  const float* go = dout + grp_row0[g];
  float* dl = dlogits + grp_row0[g];
  float* rmax = sm;
  float* rsum = rmax + N;
  float* rdot = rsum + N;
  float* cmax = rdot + N;
  float* csum = cmax + M;
  float* cdot = csum + M;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool dual = (mode != MMMOT_SM_SINGLE);
  for (int i = wave; i < N; i += 4) {
    float mx = -INFINITY;
    for (int j = lane; j < M; j += 64) mx = fmaxf(mx, x[(long)i * M + j]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int j = lane; j < M; j += 64) s += expf(x[(long)i * M + j] - mx);
    s = wave_sum(s);
    if (lane == 0) { rmax[i] = mx; rsum[i] = s; }
  }
  if (dual) {
    for (int j = tid; j < M; j += 256) {
      float mx = -INFINITY;
      for (int i = 0; i < N; ++i) mx = fmaxf(mx, x[(long)i * M + j]);
      float s = 0.f;
      for (int i = 0; i < N; ++i) s += expf(x[(long)i * M + j] - mx);
      cmax[j] = mx;
      csum[j] = s;
    }
  }
  __syncthreads();
  auto grads = [&](int i, int j, float& p, float& q, float& dp, float& dq) {
    const float v = x[(long)i * M + j], d = go[(long)i * M + j];
    p = expf(v - rmax[i]) / rsum[i];
    q = dual ? expf(v - cmax[j]) / csum[j] : 0.f;
    if (mode == MMMOT_SM_SINGLE) { dp = d; dq = 0.f; }
    else if (mode == MMMOT_SM_DUAL) { dp = d * q; dq = d * p; }
    else if (mode == MMMOT_SM_DUAL_ADD) { dp = 0.5f * d; dq = 0.5f * d; }
    else { dp = (p >= q) ? d : 0.f; dq = (p >= q) ? 0.f : d; }
  };
  for (int i = wave; i < N; i += 4) {
    float s = 0.f;
    for (int j = lane; j < M; j += 64) {
      float p, q, dp, dq;
      grads(i, j, p, q, dp, dq);
      s = fmaf(dp, p, s);
    }
    s = wave_sum(s);
    if (lane == 0) rdot[i] = s;
  }
  if (dual) {
    for (int j = tid; j < M; j += 256) {
      float s = 0.f;
      for (int i = 0; i < N; ++i) {
        float p, q, dp, dq;
        grads(i, j, p, q, dp, dq);
        s = fmaf(dq, q, s);
      }
      cdot[j] = s;
    }
  }
  __syncthreads();
  for (int idx = tid; idx < N * M; idx += 256) {
    const int i = idx / M, j = idx - i * M;
    float p, q, dp, dq;
    grads(i, j, p, q, dp, dq);
    float r = p * (dp - rdot[i]);
    if (dual) r += q * (dq - cdot[j]);
    dl[idx] = r;
  }
}

extern "C" int mmmot_softmax_pairs_bwd(const float* logits, const float* dout, float* dlogits, const int* grp_row0,
                                       const int* grp_N, const int* grp_M, int G, int max_nm, int mode, void* stream) {
  if (!logits || !dout || !dlogits || !grp_row0 || !grp_N || !grp_M || G <= 0 || max_nm <= 0) return MMMOT_EINVAL;
  if (mode < MMMOT_SM_SINGLE || mode > MMMOT_SM_DUAL_MAX) return MMMOT_EINVAL;
  const size_t lds = (size_t)3 * max_nm * sizeof(float);
  if (lds > 64 * 1024) return MMMOT_EINVAL;
  hipLaunchKernelGGL(softmax_pairs_bwd_kernel, dim3(G), dim3(256), lds, (hipStream_t)stream, logits, dout, dlogits,
                     grp_row0, grp_N, grp_M, mode);
  return mm_check(hipGetLastError());
}

// ---------------------------------------------------------------------------
// Backward of the fusion module C combine (modules/fusion_net.py:31-42; forward: mmmot_fusion_combine mode C):
//   fused = (a0*n0 + a1*n1) / (a0 + a1),  a_j = sigmoid(g_j),  n_j = i_j*sc_j + sh_j   (Y_j = [g_j | i_j] per row)
// given dfu = dL/dfused:  dn_j = dfu*a_j/den,  da_j = dfu*(n_j - fused)/den,  dg_j = da_j*a_j*(1 - a_j).
// Writes dg_j into DY_j[:, 0:C] (final: the gates have no norm) and dn_j into DN_j (to be passed through the
// GroupNorm backward into DY_j[:, C:2C]).  Modes A / B need no kernel: dn = dfu.
__global__ __launch_bounds__(256) void fusion_c_bwd_kernel(
    const float* __restrict__ dFu, const float* __restrict__ Y0, int ld0, const float* __restrict__ Y1, int ld1,
    const float* __restrict__ sc0, const float* __restrict__ sh0, const float* __restrict__ sc1,
    const float* __restrict__ sh1, int ldsc, const int* __restrict__ tile_row0, const int* __restrict__ tile_nrows,
    const int* __restrict__ tile_group, float* __restrict__ DY0, float* __restrict__ DY1, int lddy,
    float* __restrict__ DN0, float* __restrict__ DN1, int C) {
  const int t = blockIdx.x;
  const int row0 = tile_row0[t], nrows = tile_nrows[t];
  const int g = tile_group ? tile_group[t] : 0;
  const int C4 = C >> 2;
  for (int idx = threadIdx.x + 256 * blockIdx.y; idx < nrows * C4; idx += 256 * gridDim.y) {
    const int r = idx / C4, c = (idx - r * C4) * 4;
    const long d = row0 + r;
    const f32x4 df = *reinterpret_cast<const f32x4*>(&dFu[d * C + c]);
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(&Y0[d * ld0 + c]);
    const f32x4 i0 = *reinterpret_cast<const f32x4*>(&Y0[d * ld0 + C + c]);
    const f32x4 g1 = *reinterpret_cast<const f32x4*>(&Y1[d * ld1 + c]);
    const f32x4 i1 = *reinterpret_cast<const f32x4*>(&Y1[d * ld1 + C + c]);
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(&sc0[(long)g * ldsc + c]);
    const f32x4 h0 = *reinterpret_cast<const f32x4*>(&sh0[(long)g * ldsc + c]);
    const f32x4 s1 = *reinterpret_cast<const f32x4*>(&sc1[(long)g * ldsc + c]);
    const f32x4 h1 = *reinterpret_cast<const f32x4*>(&sh1[(long)g * ldsc + c]);
    f32x4 dg0, dg1, dn0, dn1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a0 = mm_sigmoid(g0[e]), a1 = mm_sigmoid(g1[e]);
      const float n0 = fmaf(i0[e], s0[e], h0[e]), n1 = fmaf(i1[e], s1[e], h1[e]);
      const float inv = 1.f / (a0 + a1);
      const float fused = (a0 * n0 + a1 * n1) * inv;
      const float w = df[e] * inv;
      dn0[e] = w * a0;
      dn1[e] = w * a1;
      dg0[e] = w * (n0 - fused) * a0 * (1.f - a0);
      dg1[e] = w * (n1 - fused) * a1 * (1.f - a1);
    }
    *reinterpret_cast<f32x4*>(&DY0[d * lddy + c]) = dg0;
    *reinterpret_cast<f32x4*>(&DY1[d * lddy + c]) = dg1;
    *reinterpret_cast<f32x4*>(&DN0[d * C + c]) = dn0;
    *reinterpret_cast<f32x4*>(&DN1[d * C + c]) = dn1;
  }
}

extern "C" int mmmot_fusion_c_bwd(const float* dFu, const float* Y0, int ld0, const float* Y1, int ld1,
                                  const float* sc0, const float* sh0, const float* sc1, const float* sh1, int ldsc,
                                  const int* tile_row0, const int* tile_nrows, const int* tile_group, int T, float* DY0,
                                  float* DY1, int lddy, float* DN0, float* DN1, int C, void* stream) {
  if (!dFu || !Y0 || !Y1 || !sc0 || !sh0 || !sc1 || !sh1 || !tile_row0 || !tile_nrows || !DY0 || !DY1 || !DN0 || !DN1)
    return MMMOT_EINVAL;
  if (T <= 0 || C <= 0 || C % 4 != 0 || ld0 % 4 != 0 || ld1 % 4 != 0 || ldsc % 4 != 0 || lddy % 4 != 0) return MMMOT_EINVAL;
  hipLaunchKernelGGL(fusion_c_bwd_kernel, dim3(T, 16), dim3(256), 0, (hipStream_t)stream, dFu, Y0, ld0, Y1, ld1, sc0,
                     sh0, sc1, sh1, ldsc, tile_row0, tile_nrows, tile_group, DY0, DY1, lddy, DN0, DN1, C);
  return mm_check(hipGetLastError());
}

// Y[r][c] = A[r][c] + B[r][c]  (accumulation of the two gradient paths into a feature: skip connection + conv)
__global__ __launch_bounds__(256) void add_rows_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                       int ldb, float* __restrict__ Y, int ldy, long R, int C) {
  const int C4 = C >> 2;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < R * C4; idx += (long)gridDim.x * 256) {
    const long r = idx / C4;
    const int c = (int)(idx - r * C4) * 4;
    const f32x4 a = *reinterpret_cast<const f32x4*>(&A[r * lda + c]);
    const f32x4 b = *reinterpret_cast<const f32x4*>(&B[r * ldb + c]);
    *reinterpret_cast<f32x4*>(&Y[r * ldy + c]) = a + b;
  }
}

extern "C" int mmmot_add_rows(const float* A, int lda, const float* B, int ldb, float* Y, int ldy, long R, int C,
                              void* stream) {
  if (!A || !B || !Y || R <= 0 || C <= 0 || C % 4 != 0 || lda % 4 != 0 || ldb % 4 != 0 || ldy % 4 != 0) return MMMOT_EINVAL;
  if (!mm_al16(A) || !mm_al16(B) || !mm_al16(Y)) return MMMOT_EINVAL;
  const long n4 = R * (C / 4);
  const int grid = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  hipLaunchKernelGGL(add_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, A, lda, B, ldb, Y, ldy, R, C);
  return mm_check(hipGetLastError());
}
