// PointNet shared-MLP layers with K = 64 input channels (feat.conv2 / conv3 64->64, conv4 64->128; reference
// modules/point_net.py:134-137): v = relu(gn(x)) W^T + b over every LiDAR point, raw output stored + per-tile GroupNorm
// statistics (the same contract as mmmot_gemm_rows with A_NORM_RELU and hl16 weights; see mmmot_pn_mlp64).
//
// Why a dedicated kernel: these layers are plain streaming passes (256 B in, 256 / 512 B out per point: HBM-bound), and the
// generic 128-row-tile kernel ran them at 2.8-3.3 TB/s (profiles/README.md r02): one workgroup per tile re-stages the
// whole weight matrix (16-32 KB from L2 for 32 KB of activations) and waits for its own loads before it computes.  Here
//   * workgroups are persistent (two per CU) and keep the layer's weights - hi and lo planes - in LDS for their lifetime;
//   * the raw rows of the NEXT tile are in flight (registers) while the current tile runs its MFMAs and epilogue;
//   * statistics and stores come straight from the accumulator registers (one wave = 32 rows x all channels).
// Arithmetic: fp16 matrix cores, 3-term hi/lo split (hl16 weights pre-scaled by 1/oscale), fp32 accumulation.
#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define PM_K 64
#define PM_LDT 72  // halves per LDS row (144 B): conflict-free ds_read_b128

template <int NOUT>
__global__ __launch_bounds__(256) void pn_mlp64_kernel(const float* __restrict__ X, int ldx,
                                                       const float* __restrict__ sc, const float* __restrict__ sh,
                                                       int ldsc, const u32x4* __restrict__ W16, float oscale,
                                                       const float* __restrict__ bias, float* __restrict__ Y, int ldy,
                                                       float* __restrict__ part, const int* __restrict__ tile_row0,
                                                       const int* __restrict__ tile_nrows,
                                                       const int* __restrict__ tile_group, int T) {
  constexpr int NT = NOUT / 32;
  __shared__ __attribute__((aligned(16))) _Float16 Wh[NOUT * PM_LDT];
  __shared__ __attribute__((aligned(16))) _Float16 Wl[NOUT * PM_LDT];
  __shared__ __attribute__((aligned(16))) _Float16 Ah[MM_BM * PM_LDT];
  __shared__ __attribute__((aligned(16))) _Float16 Al[MM_BM * PM_LDT];
  __shared__ float red[2][4][NOUT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, kh = (lane >> 5) * 8;

  // ---- the layer's weights: hl16 rows of 8 units [hi8 | lo8] -> hi / lo planes, once per workgroup ----
  for (int idx = tid; idx < NOUT * 8; idx += 256) {
    const int n = idx >> 3, u = idx & 7;
    const u32x4 hi = W16[(long)idx * 2], lo = W16[(long)idx * 2 + 1];
    *reinterpret_cast<u32x4*>(&Wh[n * PM_LDT + u * 8]) = hi;
    *reinterpret_cast<u32x4*>(&Wl[n * PM_LDT + u * 8]) = lo;
  }
  // staging: thread -> (rows (tid >> 4) + 16 i, channels 4 (tid & 15) .. + 3): one load instruction of a wave covers
  // four whole rows = 1 KB of contiguous memory
  const int srow = tid >> 4, sc0 = (tid & 15) * 4;
  f32x4 x[8];
  auto load_rows = [&](int tt) {
    const int row0 = tile_row0[tt], nrows = tile_nrows[tt];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = srow + 16 * i;
      x[i] = *reinterpret_cast<const f32x4*>(X + (long)(row0 + (r < nrows ? r : 0)) * ldx + sc0);
    }
  };
  int t = blockIdx.x;
  if (t < T) load_rows(t);
  float bv[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bv[nt] = bias ? bias[nt * 32 + lr] : 0.f;

  for (; t < T; t += gridDim.x) {
    const int row0 = tile_row0[t], nrows = tile_nrows[t];
    const int grp = tile_group ? tile_group[t] : 0;
    // ---- normalise + ReLU + hi/lo split of this tile's rows (in registers since the previous iteration) -> LDS ----
    {
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(sc + (long)grp * ldsc + sc0);
      const f32x4 h0 = *reinterpret_cast<const f32x4*>(sh + (long)grp * ldsc + sc0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = srow + 16 * i;
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        u32x2 hi, lo;
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          y[e] = fminf(fmaxf(fmaf(x[i][e], s0[e], h0[e]), 0.f), 65000.f);
          if (r >= nrows) y[e] = 0.f;
        }
        unsigned h0_, l0_, h1_, l1_;
        mm_split2(y[0], y[1], h0_, l0_);  // lo = (f16)(y - (float)hi) as one v_fma_mix per value (common.h)
        mm_split2(y[2], y[3], h1_, l1_);
        hi = u32x2{h0_, h1_};
        lo = u32x2{l0_, l1_};
        *reinterpret_cast<u32x2*>(&Ah[r * PM_LDT + sc0]) = hi;
        *reinterpret_cast<u32x2*>(&Al[r * PM_LDT + sc0]) = lo;
      }
    }
    const int tn = t + gridDim.x;
    if (tn < T) load_rows(tn);  // the next tile's rows travel while this one computes
    __syncthreads();
    // ---- 32 rows x NOUT channels per wave: 4 k-steps x 3 MFMAs per 32-channel block ----
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;
    const int offa = (wave * 32 + lr) * PM_LDT + kh;
#pragma unroll
    for (int j = 0; j < PM_K / 16; ++j) {
      const f16x8 ah = *reinterpret_cast<const f16x8*>(&Ah[offa + j * 16]);
      const f16x8 al = *reinterpret_cast<const f16x8*>(&Al[offa + j * 16]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int offb = (nt * 32 + lr) * PM_LDT + kh + j * 16;
        const f16x8 bh = *reinterpret_cast<const f16x8*>(&Wh[offb]);
        const f16x8 bl = *reinterpret_cast<const f16x8*>(&Wl[offb]);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[nt], 0, 0, 0);
      }
    }
    // ---- epilogue: v = acc * oscale + bias; raw store; per-tile sum and tile-centred M2 per channel ----
    bool valid[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) valid[e] = wave * 32 + mm_acc_row(e, lane) < nrows;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float s1 = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        acc[nt][e] = fmaf(acc[nt][e], oscale, bv[nt]);
        if (valid[e]) {
          s1 += acc[nt][e];
          Y[(long)(row0 + wave * 32 + mm_acc_row(e, lane)) * ldy + nt * 32 + lr] = acc[nt][e];
        }
      }
      s1 = mm_xor32_sum(s1);
      if (lane < 32) red[0][wave][nt * 32 + lr] = s1;
    }
    __syncthreads();
    const float inv = 1.f / (float)nrows;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = nt * 32 + lr;
      const float tot = red[0][0][n] + red[0][1][n] + red[0][2][n] + red[0][3][n];
      const float mu = tot * inv;
      float s2 = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float d = acc[nt][e] - mu;
        if (valid[e]) s2 = fmaf(d, d, s2);
      }
      s2 = mm_xor32_sum(s2);
      if (lane < 32) red[1][wave][n] = s2;
      if (wave == 0 && lane < 32) part[((long)t * 2 + 0) * NOUT + n] = tot;
    }
    __syncthreads();
    if (wave == 0 && lane < 32) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = nt * 32 + lr;
        part[((long)t * 2 + 1) * NOUT + n] = red[1][0][n] + red[1][1][n] + red[1][2][n] + red[1][3][n];
      }
    }
    // the next iteration's LDS writes (Ah / Al, then red[0]) come after this barrier for every wave; red[1] is next
    // written two barriers from now
  }
}

extern "C" int mmmot_pn_mlp64(const float* X, int ldx, const float* sc, const float* sh, int ldsc, const void* W16,
                              float oscale, const float* bias, float* Y, int ldy, float* part, const int* tile_row0,
                              const int* tile_nrows, const int* tile_group, int T, int N, void* stream) {
  if (!X || !sc || !sh || !W16 || !Y || !part || !tile_row0 || !tile_nrows || T <= 0) return MMMOT_EINVAL;
  if ((N != 64 && N != 128) || ldx % 4 != 0 || ldsc % 4 != 0 || ldy < N) return MMMOT_EINVAL;
  if (!mm_al16(X) || !mm_al16(sc) || !mm_al16(sh) || !mm_al16(W16)) return MMMOT_EINVAL;
  const int n_cu = mm_num_cu();
  if (n_cu <= 0) return MMMOT_EINVAL;
  const int grid = T < 2 * n_cu ? T : 2 * n_cu;  // persistent: two workgroups per CU (LDS 55 / 74 KB each)
  hipStream_t s = (hipStream_t)stream;
  if (N == 64)
    hipLaunchKernelGGL(pn_mlp64_kernel<64>, dim3(grid), dim3(256), 0, s, X, ldx, sc, sh, ldsc, (const u32x4*)W16, oscale,
                       bias, Y, ldy, part, tile_row0, tile_nrows, tile_group, T);
  else
    hipLaunchKernelGGL(pn_mlp64_kernel<128>, dim3(grid), dim3(256), 0, s, X, ldx, sc, sh, ldsc, (const u32x4*)W16, oscale,
                       bias, Y, ldy, part, tile_row0, tile_nrows, tile_group, T);
  return mm_check(hipGetLastError());
}
