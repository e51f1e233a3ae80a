// Training step, second slice: what the backward of the LiDAR encoder and the loss need beyond the
// kernels of backward.hip (GroupNorm backward, weight-gradient GEMM, row GEMM with the transposed weight):
//   * mmmot_rows_gather_scale   - backward of the per-detection average pools (reference modules/point_net.py:32-39,
//                                 139-148): every point row receives its detection's gradient row / point count;
//   * mmmot_pointnet_layer1_bwd - weight / bias gradient of the first shared-MLP layer (K = 3 | 4 input channels: not
//                                 matrix-core shaped, like its forward mmmot_pointnet_layer1);
//   * mmmot_score_loss          - the elementwise terms of TrackingLoss (reference cost.py:97-185: binary cross entropy
//                                 with logits, masked L2, masked smooth L1) with their gradients in the same pass.
// Plain fp32, deterministic (per-block partial sums, no atomics).
#include "common.h"

// X[r][c] = S[rowidx[r]][c] * (scale ? scale[rowidx[r]] : 1)
__global__ __launch_bounds__(256) void rows_gather_scale_kernel(const float* __restrict__ S, int lds,
                                                                const int* __restrict__ rowidx,
                                                                const float* __restrict__ scale, float* __restrict__ X,
                                                                int ldx, long R, int C) {
  const int C4 = C >> 2;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < R * C4; idx += (long)gridDim.x * 256) {
    const long r = idx / C4;
    const int c = (int)(idx - r * C4) * 4;
    const int s = rowidx[r];
    f32x4 v = *reinterpret_cast<const f32x4*>(&S[(long)s * lds + c]);
    if (scale) {
      const float f = scale[s];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= f;
    }
    *reinterpret_cast<f32x4*>(&X[r * ldx + c]) = v;
  }
}

extern "C" int mmmot_rows_gather_scale(const float* S, int lds, const int* rowidx, const float* scale, float* X, int ldx,
                                       long R, int C, void* stream) {
  if (!S || !rowidx || !X || R <= 0 || C <= 0 || C % 4 != 0 || lds % 4 != 0 || ldx % 4 != 0) return MMMOT_EINVAL;
  if (!mm_al16(S) || !mm_al16(X)) return MMMOT_EINVAL;
  const long n4 = R * (C / 4);
  const int grid = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
  hipLaunchKernelGGL(rows_gather_scale_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, S, lds, rowidx, scale, X,
                     ldx, R, C);
  return mm_check(hipGetLastError());
}

// PW[t][c * (K + 1) + k] = sum over the tile's rows of dY[r][c] * X[r][k]   (k < K),   PW[t][c * (K + 1) + K] = sum dY[r][c]
// One workgroup per tile (<= 128 rows), thread -> (channel c = tid & 63, row quarter): same decomposition as the forward.
template <int K>
__global__ __launch_bounds__(256) void pointnet_layer1_bwd_kernel(const float* __restrict__ dY, const float* __restrict__ X,
                                                                  const int* __restrict__ tile_row0,
                                                                  const int* __restrict__ tile_nrows,
                                                                  float* __restrict__ PW) {
  __shared__ float xs[MM_BM * K];
  __shared__ double red[4][64][K + 1];
  const int t = blockIdx.x;
  const int row0 = tile_row0[t], nrows = tile_nrows[t];
  const int tid = threadIdx.x;
  for (int idx = tid; idx < nrows * K; idx += 256) xs[idx] = X[(long)row0 * K + idx];
  __syncthreads();
  const int c = tid & 63, rq = tid >> 6;
  // float64 accumulators: the GroupNorm backward makes sum_r dY[r][c] = 0, and the coordinates are metres from the
  // sensor (tens) while what distinguishes the points of a box is decimetres - the sum cancels by 2-3 digits
  double acc[K + 1];
#pragma unroll
  for (int k = 0; k <= K; ++k) acc[k] = 0.0;
  for (int i = 0; i < 32; ++i) {
    const int r = rq * 32 + i;
    if (r < nrows) {
      const double d = (double)dY[(long)(row0 + r) * 64 + c];
#pragma unroll
      for (int k = 0; k < K; ++k) acc[k] = fma(d, (double)xs[r * K + k], acc[k]);
      acc[K] += d;
    }
  }
#pragma unroll
  for (int k = 0; k <= K; ++k) red[rq][c][k] = acc[k];
  __syncthreads();
  for (int idx = tid; idx < 64 * (K + 1); idx += 256) {
    const int cc = idx / (K + 1), k = idx - cc * (K + 1);
    PW[(long)t * 64 * (K + 1) + idx] = (float)(red[0][cc][k] + red[1][cc][k] + red[2][cc][k] + red[3][cc][k]);
  }
}

extern "C" int mmmot_pointnet_layer1_bwd(const float* dY, const float* X, int K, const int* tile_row0,
                                         const int* tile_nrows, int T, float* PW, void* stream) {
  if (!dY || !X || !tile_row0 || !tile_nrows || !PW || T <= 0 || (K != 3 && K != 4)) return MMMOT_EINVAL;
  if (K == 3)
    hipLaunchKernelGGL(pointnet_layer1_bwd_kernel<3>, dim3(T), dim3(256), 0, (hipStream_t)stream, dY, X, tile_row0,
                       tile_nrows, PW);
  else
    hipLaunchKernelGGL(pointnet_layer1_bwd_kernel<4>, dim3(T), dim3(256), 0, (hipStream_t)stream, dY, X, tile_row0,
                       tile_nrows, PW);
  return mm_check(hipGetLastError());
}

// Elementwise loss term over x [R][C] (R modality rows, row stride ldx) against a target y [C] shared by the rows:
//   kind MMMOT_LOSS_BCE      : l = max(x, 0) - x y + log(1 + exp(-|x|))         (F.binary_cross_entropy_with_logits)
//   kind MMMOT_LOSS_L2       : l = (x m - y)^2                                  (F.mse_loss(x.mul(mask), y))
//   kind MMMOT_LOSS_SMOOTH_L1: l = 0.5 d^2 if |d| < 1 else |d| - 0.5, d = x m - y  (F.smooth_l1_loss(x.mul(mask), y))
// mask m = mr * mc with mr from mrow[idx / M], mc from mcol[idx % M] (absent: 1); mask_mode 1: value == 1, 2: value !=
// ignore.  Writes g[r][c] = scale * dl/dx and per-block partial sums PL[block] of scale * l (accumulate != 0: added to
// what PL[block] holds - with one block the terms of a loss add up in PL[0] launch after launch).
__global__ __launch_bounds__(256) void score_loss_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ y,
                                                         const float* __restrict__ mrow, const float* __restrict__ mcol,
                                                         int M, int mask_mode, float ignore, int kind, float scale, int R,
                                                         int C, float* __restrict__ g, int ldg, float* __restrict__ PL,
                                                         int accumulate) {
  __shared__ float red[4];
  float s = 0.f;
  const long n = (long)R * C;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
    const int r = (int)(idx / C), c = (int)(idx - (long)r * C);
    const float xv = x[(long)r * ldx + c], yv = y[c];
    float m = 1.f;
    if (mask_mode) {
      if (mrow) {
        const float v = mrow[c / M];
        m *= (mask_mode == 1) ? (v == 1.f ? 1.f : 0.f) : (v != ignore ? 1.f : 0.f);
      }
      if (mcol) {
        const float v = mcol[c % M];
        m *= (mask_mode == 1) ? (v == 1.f ? 1.f : 0.f) : (v != ignore ? 1.f : 0.f);
      }
    }
    float l, d;
    if (kind == MMMOT_LOSS_BCE) {
      l = fmaxf(xv, 0.f) - xv * yv + log1pf(expf(-fabsf(xv)));
      d = 1.f / (1.f + expf(-xv)) - yv;
    } else {
      const float e = xv * m - yv;
      if (kind == MMMOT_LOSS_L2) {
        l = e * e;
        d = 2.f * e * m;
      } else {
        const float a = fabsf(e);
        l = a < 1.f ? 0.5f * e * e : a - 0.5f;
        d = (a < 1.f ? e : (e > 0.f ? 1.f : -1.f)) * m;
      }
    }
    s += l;
    g[(long)r * ldg + c] = scale * d;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float v = scale * (red[0] + red[1] + red[2] + red[3]);
    PL[blockIdx.x] = accumulate ? PL[blockIdx.x] + v : v;  // launches on one stream are ordered: no race
  }
}

extern "C" int mmmot_score_loss(const float* x, int ldx, const float* y, const float* mrow, const float* mcol, int M,
                                int mask_mode, float ignore, int kind, float scale, int R, int C, float* g, int ldg,
                                float* PL, int nblocks, int accumulate, void* stream) {
  if (!x || !y || !g || !PL || R <= 0 || C <= 0 || nblocks <= 0 || nblocks > 4096) return MMMOT_EINVAL;
  if (kind < MMMOT_LOSS_BCE || kind > MMMOT_LOSS_SMOOTH_L1 || mask_mode < 0 || mask_mode > 2) return MMMOT_EINVAL;
  if (mask_mode && (mrow || mcol) && M <= 0) return MMMOT_EINVAL;
  hipLaunchKernelGGL(score_loss_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, x, ldx, y, mrow, mcol, M,
                     mask_mode, ignore, kind, scale, R, C, g, ldg, PL, accumulate);
  return mm_check(hipGetLastError());
}

// ---------------------------------------------------------------------------
// detloss_type / endloss_type 'ghm' of TrackingLoss (reference cost.py:105-110,126-128 -> modules/ghm_loss.py:15-61,
// GHMC_Loss(bins = 30, momentum = 0.75)): gradient-harmonised binary cross entropy of x [R][C] against the target y [C]
// shared by the R modality rows, entries with y == ignore masked out.  ONE workgroup (the det / new / end score tensors
// of a sample hold 3 x L values), three passes over the elements:
//   1. gradient length gl = |sigmoid(x) - y|, histogram of the valid elements over `bins` equal bins of [0, 1]
//      (float32 comparisons against float32 edges like the reference's tensor-vs-scalar comparisons; last edge + 1e-6),
//      tot = max(number of valid elements, 1);
//   2. per non-empty bin (float64, like the reference's Python floats): acc_sum[b] <- momentum * acc_sum[b] +
//      (1 - momentum) * count (momentum == 0: acc = count, acc_sum untouched), weight = float(tot / acc) / n with n the
//      number of non-empty bins; acc_sum [bins] is the loss module's STATE (device memory, updated in place);
//   3. l = w * bce_with_logits(x, y), g = scale * w * (sigmoid(x) - y) / tot (the weights are constants of the graph),
//      PL[0] (+)= scale * sum(l) / tot.
#define GHM_MAX_BINS 64
__global__ __launch_bounds__(256) void ghm_loss_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ y,
                                                       float ignore, float scale, int R, int C, int bins, float momentum,
                                                       double* __restrict__ acc_sum, float* __restrict__ g, int ldg,
                                                       float* __restrict__ PL, int accumulate) {
  __shared__ int cnt[GHM_MAX_BINS];
  __shared__ float wbin[GHM_MAX_BINS];
  __shared__ float edge[GHM_MAX_BINS + 1];
  __shared__ int nvalid, nbins_used;
  __shared__ float red[4];
  const int tid = threadIdx.x;
  const int n = R * C;
  if (tid < bins) cnt[tid] = 0;
  if (tid <= bins) edge[tid] = (float)((double)tid / (double)bins + (tid == bins ? 1e-6 : 0.0));
  if (tid == 0) nvalid = 0;
  __syncthreads();
  auto bin_of = [&](float xv, float yv) {
    const float gl = fabsf(1.f / (1.f + expf(-xv)) - yv);
    int b = -1;
    for (int i = 0; i < bins; ++i)
      if (gl >= edge[i] && gl < edge[i + 1]) b = i;
    return b;
  };
  for (int idx = tid; idx < n; idx += 256) {
    const int r = idx / C, c = idx - r * C;
    const float yv = y[c];
    if (yv != ignore) {
      atomicAdd(&nvalid, 1);
      const int b = bin_of(x[(long)r * ldx + c], yv);
      if (b >= 0) atomicAdd(&cnt[b], 1);
    }
  }
  __syncthreads();
  const double tot = nvalid > 1 ? (double)nvalid : 1.0;
  if (tid == 0) {
    int used = 0;
    for (int b = 0; b < bins; ++b)
      if (cnt[b] > 0) {
        double acc = (double)cnt[b];
        if (momentum > 0.f) {
          acc = (double)momentum * acc_sum[b] + (1.0 - (double)momentum) * (double)cnt[b];
          acc_sum[b] = acc;
        }
        wbin[b] = (float)(tot / acc);
        ++used;
      } else {
        wbin[b] = 0.f;
      }
    nbins_used = used;
  }
  __syncthreads();
  const float nused = (float)(nbins_used > 0 ? nbins_used : 1);
  const float ftot = (float)tot;
  float s = 0.f;
  for (int idx = tid; idx < n; idx += 256) {
    const int r = idx / C, c = idx - r * C;
    const float xv = x[(long)r * ldx + c], yv = y[c];
    float w = 0.f;
    if (yv != ignore) {
      const int b = bin_of(xv, yv);
      if (b >= 0) w = wbin[b] / nused;
    }
    const float l = fmaxf(xv, 0.f) - xv * yv + log1pf(expf(-fabsf(xv)));
    const float d = 1.f / (1.f + expf(-xv)) - yv;
    s += w * l;
    g[(long)r * ldg + c] = scale * (w * d) / ftot;
  }
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) {
    const float v = scale * ((red[0] + red[1] + red[2] + red[3]) / ftot);
    PL[0] = accumulate ? PL[0] + v : v;
  }
}

extern "C" int mmmot_ghm_loss(const float* x, int ldx, const float* y, float ignore, float scale, int R, int C, int bins,
                              float momentum, double* acc_sum, float* g, int ldg, float* PL, int accumulate, void* stream) {
  if (!x || !y || !acc_sum || !g || !PL || R <= 0 || C <= 0 || (long)R * C > (1L << 24)) return MMMOT_EINVAL;
  if (bins <= 0 || bins > GHM_MAX_BINS || !(momentum >= 0.f) || momentum >= 1.f || ldx < C || ldg < C) return MMMOT_EINVAL;
  hipLaunchKernelGGL(ghm_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ignore, scale, R, C, bins,
                     momentum, acc_sum, g, ldg, PL, accumulate);
  return mm_check(hipGetLastError());
}

