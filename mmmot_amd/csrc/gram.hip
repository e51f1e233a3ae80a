// GroupNorm statistics of a 1x1-conv output WITHOUT running the conv: second moments of its input.
// (see include/mmmot_hip.h: mmmot_gram_rows / mmmot_gn_finalize_gram)
//
// For v = W a + b over the rows of a normalisation group (per-channel GroupNorm(C, C), e.g. PointNet conv5
// 128 -> 1024 over all points of a sample, reference modules/point_net.py:138):
//     mean_c = w_c . m + b_c,   var_c = w_c^T Cov(a) w_c,   m = E[a],  Cov(a) = E[a a^T] - m m^T
// so one pass over the K-channel INPUT (K = 64 / 128) replaces a full pass of the K -> N GEMM (N = 1024:
// 1.0 ms of a 17 ms step at cfg3).  Checked on the cfg3 weights against float64 statistics of v: relative
// variance error <= 8e-7 (the direct per-tile fp32 scheme: 1e-7), i.e. 4e-7 on the scale - noise next to the
// 1e-3 budget.  What makes it safe:
//   * a = relu(x*sc + sh) is non-negative with mean ~ std, so E[a a^T] - m m^T cancels at most a factor ~2;
//   * Gram tiles are accumulated over 128 rows in fp32 on the matrix cores (3-term hi/lo split of the fp16
//     operands: products exact to 2^-22), then merged into a COMPENSATED (two-float) running sum per element
//     and written as float64; tiles/groups are merged and the quadratic forms evaluated in float64.
#include <atomic>

#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define GR_ROWS 128      // rows per sub-tile (fp32 accumulation length)
#define GR_LD 136        // halves per transposed LDS row: [channel][row], 272 B
#define GR_THREADS 256

// Transposed planes [channel][row] are written 4 rows (8 bytes) at a time by threads whose channels lie CPT = K / 8 apart:
// 68 dwords per channel row x 16 (8) channels = a multiple of 32 banks - all eight threads of a row quad on ONE bank pair
// (PMC, round 6: 71 % of the LDS cycles of the K = 128 kernel were bank conflicts; 14 % with the permutation below).  The
// row quads of a channel are therefore stored permuted inside aligned groups of 16 quads (64 rows): quad q of channel c
// lives at q ^ (2 * ((c / CPT) & 7)) - the 16 lanes of a store (8 channels x 2 quads) then cover 16 different bank pairs.
// q and q + 1 (q even) stay neighbours in order, so a 16-byte fragment read (8 consecutive rows) is still one aligned read.
template <int CPT>
__device__ __forceinline__ int gr_quad(int q, int c) { return q ^ (((c / CPT) & 7) << 1); }

// Fast two-sum accumulate: (s, c) += x with the rounding error of s + x collected in c.
__device__ __forceinline__ void gr_acc(float& s, float& c, float x) {
  const float t = s + x;
  const float bp = t - s;
  c += (s - (t - bp)) + (x - bp);
  s = t;
}

// grid = super-tiles (<= 8 sub-tiles of 128 rows inside one group).  K = 64 or 128 channels.
template <int K>
__global__ __launch_bounds__(GR_THREADS) void gram_rows_kernel(const float* __restrict__ X, int ldx,
                                                                const float* __restrict__ sc,
                                                                const float* __restrict__ sh, int ldsc,
                                                                const int* __restrict__ tile_row0,
                                                                const int* __restrict__ tile_nrows,
                                                                const int* __restrict__ tile_group,
                                                                double* __restrict__ Gout, double* __restrict__ Sout) {
  constexpr int CT = K / 32;            // 32-channel MFMA tiles per side: 4 / 2
  constexpr int WT = (K == 128) ? 2 : 1;  // tiles per wave per side (4 waves: 2x2 waves over the K x K output)
  __shared__ __attribute__((aligned(16))) _Float16 Th[K * GR_LD];  // hi plane, transposed [channel][row]
  __shared__ __attribute__((aligned(16))) _Float16 Tl[K * GR_LD];  // lo plane
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1;  // this wave's block of the Gram matrix
  const int lr = lane & 31, kh = (lane >> 5) * 8;
  const int t = blockIdx.x;
  const int row0 = tile_row0[t], nrows = tile_nrows[t];
  const int grp = tile_group ? tile_group[t] : 0;

  // staging: thread -> (4 consecutive rows sr4 .. sr4+3, CPT consecutive channels): the transposed LDS image
  // [channel][row] then takes one 8-byte store per channel and plane instead of four 2-byte ones
  constexpr int TPR = GR_THREADS / (GR_ROWS / 4);  // threads per row quad: 8
  constexpr int CPT = K / TPR;                     // channels per staging thread: 16 / 8
  const int sr4 = (tid / TPR) * 4, sq = tid % TPR;
  float gs[WT][WT][16], gc[WT][WT][16];
#pragma unroll
  for (int a = 0; a < WT; ++a)
#pragma unroll
    for (int b = 0; b < WT; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) gs[a][b][e] = gc[a][b][e] = 0.f;
  float ss = 0.f, scc = 0.f;  // compensated column sum of channel `tid` (threads < K)

  // The raw rows of sub-tile r0 + 128 are requested while sub-tile r0 is split, summed and multiplied: one workgroup per
  // CU (the compensated Gram accumulators take 128 registers per lane), so nothing else would hide the load latency -
  // round 3 measured 2.1 TB/s, a quarter of the HBM rate, with the latency exposed once per 128 rows.
  f32x4 xr[CPT / 4][4];
  auto load_rows = [&](int r0) {
    const int nr = min(GR_ROWS, nrows - r0);
#pragma unroll
    for (int c4 = 0; c4 < CPT; c4 += 4)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const bool rv = sr4 + rr < nr;  // rows past the end read row 0 of the sub-tile (mapped) and are zeroed below
        xr[c4 / 4][rr] = *reinterpret_cast<const f32x4*>(X + (long)(row0 + r0 + (rv ? sr4 + rr : 0)) * ldx + sq * CPT + c4);
      }
  };
  load_rows(0);
  for (int r0 = 0; r0 < nrows; r0 += GR_ROWS) {
    const int nr = min(GR_ROWS, nrows - r0);
    // ---- stage: normalise + ReLU + hi/lo split, transposed into LDS ([channel][row]) ----
    {
      typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
      const float* ps = sc + (long)grp * ldsc + sq * CPT;
      const float* ph = sh + (long)grp * ldsc + sq * CPT;
#pragma unroll
      for (int c4 = 0; c4 < CPT; c4 += 4) {
        const f32x4 s4 = *reinterpret_cast<const f32x4*>(ps + c4);
        const f32x4 h4 = *reinterpret_cast<const f32x4*>(ph + c4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          f16x4 hi, lo;
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            float y = fminf(fmaxf(fmaf(xr[c4 / 4][rr][e], s4[e], h4[e]), 0.f), 65000.f);
            if (sr4 + rr >= nr) y = 0.f;  // rows past the end of the super-tile contribute nothing
            hi[rr] = (_Float16)y;
            lo[rr] = (_Float16)(y - (float)hi[rr]);
          }
          const int c = sq * CPT + c4 + e;
          const int o = c * GR_LD + (gr_quad<CPT>(sr4 >> 2, c) << 2);
          *reinterpret_cast<f16x4*>(&Th[o]) = hi;
          *reinterpret_cast<f16x4*>(&Tl[o]) = lo;
        }
      }
    }
    if (r0 + GR_ROWS < nrows) load_rows(r0 + GR_ROWS);
    __syncthreads();
    // ---- column sums (deterministic order): thread c adds its channel's 128 rows, hi and lo ----
    if (tid < K) {
      float acc = 0.f;
#pragma unroll 4
      for (int r8 = 0; r8 < GR_ROWS; r8 += 8) {
        const f16x8 h = *reinterpret_cast<const f16x8*>(&Th[tid * GR_LD + r8]);
        const f16x8 l = *reinterpret_cast<const f16x8*>(&Tl[tid * GR_LD + r8]);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += (float)h[e] + (float)l[e];
      }
      gr_acc(ss, scc, acc);
    }
    // ---- Gram of the sub-tile on the matrix cores: G[i][j] = sum_r a[r][i] a[r][j] ----
    f32x16 acc[WT][WT];
#pragma unroll
    for (int a = 0; a < WT; ++a)
#pragma unroll
      for (int b = 0; b < WT; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
#pragma unroll
    for (int k16 = 0; k16 < GR_ROWS / 16; ++k16) {
      f16x8 ah[WT], al[WT], bh[WT], bl[WT];
#pragma unroll
      for (int a = 0; a < WT; ++a) {
        const int ch = (wi * WT + a) * 32 + lr;
        const int off = ch * GR_LD + (gr_quad<CPT>((k16 * 16 + kh) >> 2, ch) << 2);
        ah[a] = *reinterpret_cast<const f16x8*>(&Th[off]);
        al[a] = *reinterpret_cast<const f16x8*>(&Tl[off]);
      }
#pragma unroll
      for (int b = 0; b < WT; ++b) {
        const int ch = (wj * WT + b) * 32 + lr;
        const int off = ch * GR_LD + (gr_quad<CPT>((k16 * 16 + kh) >> 2, ch) << 2);
        bh[b] = *reinterpret_cast<const f16x8*>(&Th[off]);
        bl[b] = *reinterpret_cast<const f16x8*>(&Tl[off]);
      }
#pragma unroll
      for (int a = 0; a < WT; ++a)
#pragma unroll
        for (int b = 0; b < WT; ++b) {
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[a], bh[b], acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bl[b], acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bh[b], acc[a][b], 0, 0, 0);
        }
    }
#pragma unroll
    for (int a = 0; a < WT; ++a)
#pragma unroll
      for (int b = 0; b < WT; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) gr_acc(gs[a][b][e], gc[a][b][e], acc[a][b][e]);
    __syncthreads();  // the planes are rewritten by the next sub-tile
  }
  // ---- write the super-tile's partial as float64 ----
  double* G = Gout + (long)t * K * K;
#pragma unroll
  for (int a = 0; a < WT; ++a)
#pragma unroll
    for (int b = 0; b < WT; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int i = (wi * WT + a) * 32 + mm_acc_row(e, lane);
        const int j = (wj * WT + b) * 32 + lr;
        G[(long)i * K + j] = (double)gs[a][b][e] + (double)gc[a][b][e];
      }
  if (tid < K) Sout[(long)t * K + tid] = (double)ss + (double)scc;
  (void)CT;
}

// K = 128, round 6: the UPPER TRIANGLE of the 4 x 4 grid of 32 x 32 blocks only (10 of 16; the matrix is symmetric and
// gram_reduce_kernel mirrors), float64 running sums (one v_add_f64 per element and sub-tile instead of the six-operation
// two-float merge: at least as accurate, a third of the instructions) and a register budget that lets TWO workgroups
// share a CU.  The generic kernel above needs 413 registers at K = 128 (64 accumulators + 128 compensated sums + the
// prefetched rows of the next sub-tile): one wave per SIMD, so its phases - convert + transpose to LDS (VALU), column sums
// (VALU), Gram (MFMA), merge (VALU) - ran one after the other with the matrix cores idle through three of them (2.3 TB/s,
// 0.29 of HBM).  Here a wave owns at most 3 blocks (48 accumulators + 96 registers of float64 sums, 250 registers in all)
// and one workgroup's VALU phases run under the other's matrix phase: 3.3 - 3.6 TB/s.  Waves: 0 -> (0,0) (0,1) (0,2);
// 1 -> (0,3) (1,3) (2,3); 2 -> (1,1) (1,2) (3,3); 3 -> (2,2); the column sums are spread over all 256 threads (channel,
// 64-row half).  Block (i, j) needs the fragments of channel blocks i and j: the A fragment of block row i and the B
// fragment of block column i are the same LDS data.  What is left (PMC): the waves sit in s_waitcnt half of their time -
// the rows of a sub-tile are requested where they are converted, and two workgroups do not cover an HBM round trip per
// 128 rows; holding the next sub-tile's rows in registers across the matrix phase (32 - 64 registers) spills (measured:
// 1.7 - 2.1 ms against 1.29), a quarter of them (16 registers) is worth 4 % and was left out.
template <int MASK, int NBLK, int A0, int B0, int A1, int B1, int A2, int B2>
__device__ __forceinline__ void gr_blocks(const _Float16* Th, const _Float16* Tl, int lr, int kh, f32x16 (&acc)[3]) {
#pragma unroll
  for (int k16 = 0; k16 < GR_ROWS / 16; ++k16) {
    f16x8 fh[4], fl[4];
#pragma unroll
    for (int f = 0; f < 4; ++f)
      if ((MASK >> f) & 1) {
        const int ch = f * 32 + lr;
        const int off = ch * GR_LD + (gr_quad<16>((k16 * 16 + kh) >> 2, ch) << 2);
        fh[f] = *reinterpret_cast<const f16x8*>(&Th[off]);
        fl[f] = *reinterpret_cast<const f16x8*>(&Tl[off]);
      }
    constexpr int AI[3] = {A0, A1, A2}, BI[3] = {B0, B1, B2};
#pragma unroll
    for (int b = 0; b < NBLK; ++b) {
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[AI[b]], fh[BI[b]], acc[b], 0, 0, 0);
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[AI[b]], fl[BI[b]], acc[b], 0, 0, 0);
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[AI[b]], fh[BI[b]], acc[b], 0, 0, 0);
    }
  }
}

__global__ __launch_bounds__(GR_THREADS, 2) void gram_rows128_kernel(const float* __restrict__ X, int ldx,
                                                                      const float* __restrict__ sc,
                                                                      const float* __restrict__ sh, int ldsc,
                                                                      const int* __restrict__ tile_row0,
                                                                      const int* __restrict__ tile_nrows,
                                                                      const int* __restrict__ tile_group,
                                                                      double* __restrict__ Gout, double* __restrict__ Sout) {
  constexpr int K = 128;
  __shared__ __attribute__((aligned(16))) _Float16 Th[K * GR_LD];  // hi plane, transposed [channel][row]: 34 KB
  __shared__ __attribute__((aligned(16))) _Float16 Tl[K * GR_LD];  // lo plane
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, kh = (lane >> 5) * 8;
  const int t = blockIdx.x;
  const int row0 = tile_row0[t], nrows = tile_nrows[t];
  const int grp = tile_group ? tile_group[t] : 0;
  // staging: thread -> (4 consecutive rows sr4 .. sr4 + 3, 16 consecutive channels)
  constexpr int TPR = GR_THREADS / (GR_ROWS / 4);  // 8
  constexpr int CPT = K / TPR;                     // 16
  const int sr4 = (tid / TPR) * 4, sq = tid % TPR;
  double gd[3][16];
#pragma unroll
  for (int b = 0; b < 3; ++b)
#pragma unroll
    for (int e = 0; e < 16; ++e) gd[b][e] = 0.0;
  double sd = 0.0;  // column sum of channel tid & 127 over rows 64 (tid >> 7) .. + 63 of every sub-tile

  for (int r0 = 0; r0 < nrows; r0 += GR_ROWS) {
    const int nr = min(GR_ROWS, nrows - r0);
    const bool full = (nr == GR_ROWS);  // workgroup-uniform: the row test leaves the arithmetic of every full sub-tile
    // ---- load, normalise + ReLU + hi/lo split, transposed into LDS ([channel][row]) ----
    {
      const float* ps = sc + (long)grp * ldsc + sq * CPT;
      const float* ph = sh + (long)grp * ldsc + sq * CPT;
#pragma unroll
      for (int c4 = 0; c4 < CPT; c4 += 4) {
        f32x4 xr[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const bool rv = sr4 + rr < nr;  // rows past the end read row 0 of the sub-tile (mapped) and are zeroed below
          xr[rr] = *reinterpret_cast<const f32x4*>(X + (long)(row0 + r0 + (rv ? sr4 + rr : 0)) * ldx + sq * CPT + c4);
        }
        const f32x4 s4 = *reinterpret_cast<const f32x4*>(ps + c4);
        const f32x4 h4 = *reinterpret_cast<const f32x4*>(ph + c4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float y[4];
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            y[rr] = __builtin_amdgcn_fmed3f(fmaf(xr[rr][e], s4[e], h4[e]), 0.f, 65000.f);  // ReLU + fp16 range clamp
            if (!full && sr4 + rr >= nr) y[rr] = 0.f;  // rows past the end of the super-tile contribute nothing
          }
          typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
          typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
          const f16x2 h01 = {(_Float16)y[0], (_Float16)y[1]}, h23 = {(_Float16)y[2], (_Float16)y[3]};
          const u32x2 hi = {__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)};
          const u32x2 lo = {mm_split_lo2(hi[0], y[0], y[1]), mm_split_lo2(hi[1], y[2], y[3])};
          const int c = sq * CPT + c4 + e;
          const int o = c * GR_LD + (gr_quad<CPT>(sr4 >> 2, c) << 2);
          *reinterpret_cast<u32x2*>(&Th[o]) = hi;
          *reinterpret_cast<u32x2*>(&Tl[o]) = lo;
        }
      }
    }
    __syncthreads();
    {  // column sums (deterministic order): thread -> one channel, 64 rows (stored permuted inside the 64: gr_quad), hi and
       // lo; fp32 over the 64 rows, then float64
      const int c = tid & 127, rb = (tid >> 7) * 64;
      float a = 0.f;
#pragma unroll 4
      for (int r8 = 0; r8 < 64; r8 += 8) {
        const f16x8 h = *reinterpret_cast<const f16x8*>(&Th[c * GR_LD + rb + r8]);
        const f16x8 l = *reinterpret_cast<const f16x8*>(&Tl[c * GR_LD + rb + r8]);
#pragma unroll
        for (int e = 0; e < 8; ++e) a += (float)h[e] + (float)l[e];
      }
      sd += (double)a;
    }
    f32x16 acc[3];
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
    if (wave == 0) gr_blocks<0x7, 3, 0, 0, 0, 1, 0, 2>(Th, Tl, lr, kh, acc);
    else if (wave == 1) gr_blocks<0xF, 3, 0, 3, 1, 3, 2, 3>(Th, Tl, lr, kh, acc);
    else if (wave == 2) gr_blocks<0xE, 3, 1, 1, 1, 2, 3, 3>(Th, Tl, lr, kh, acc);
    else gr_blocks<0x4, 1, 2, 2, 2, 2, 2, 2>(Th, Tl, lr, kh, acc);
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) gd[b][e] += (double)acc[b][e];
    __syncthreads();  // the planes are rewritten by the next sub-tile
  }
  double* G = Gout + (long)t * K * K;
  auto put = [&](int b, int bi, int bj) {
#pragma unroll
    for (int e = 0; e < 16; ++e) G[(long)(bi * 32 + mm_acc_row(e, lane)) * K + bj * 32 + lr] = gd[b][e];
  };
  if (wave == 0) { put(0, 0, 0); put(1, 0, 1); put(2, 0, 2); }
  else if (wave == 1) { put(0, 0, 3); put(1, 1, 3); put(2, 2, 3); }
  else if (wave == 2) { put(0, 1, 1); put(1, 1, 2); put(2, 3, 3); }
  else put(0, 2, 2);
  // the two row halves of every channel's sum meet in LDS (the planes are free: the loop ended with a barrier)
  double* sx = reinterpret_cast<double*>(Th);
  if (tid >= 128) sx[tid - 128] = sd;
  __syncthreads();
  if (tid < 128) Sout[(long)t * K + tid] = sd + sx[tid];
}

// K = 128, pipelined form: EIGHT waves, both LDS plane pairs (2 x 69.6 KB: sub-tile t is multiplied while sub-tile t + 1
// is converted into the other pair), the raw rows of sub-tiles t + 2 and t + 3 in flight in registers (2 x 32), ONE barrier
// per sub-tile.  A wave owns at most 2 of the 10 upper-triangle blocks (32 accumulators + 64 registers of float64 sums), so
// the prefetch registers fit where the four-wave form above spills.  Blocks per wave (SIMD = wave & 3 in dispatch order
// 0, 2, 1, 3: at most 3 blocks per SIMD): 0 -> (0,0) (0,1); 1 -> (0,2) (0,3); 2 -> (1,1) (1,2); 3 -> (1,3) (2,2); 4 -> (3,3);
// 5 -> (2,3); 6, 7 -> none.  Same arithmetic and summation order per block as the four-wave form: bit-identical partials.
template <int MASK, int NBLK, int A0, int B0, int A1, int B1>
__device__ __forceinline__ void gr_blocks2(const _Float16* Th, const _Float16* Tl, int lr, int kh, f32x16 (&acc)[2]) {
#pragma unroll
  for (int k16 = 0; k16 < GR_ROWS / 16; ++k16) {
    f16x8 fh[4], fl[4];
#pragma unroll
    for (int f = 0; f < 4; ++f)
      if ((MASK >> f) & 1) {
        const int ch = f * 32 + lr;
        const int off = ch * GR_LD + (gr_quad<16>((k16 * 16 + kh) >> 2, ch) << 2);
        fh[f] = *reinterpret_cast<const f16x8*>(&Th[off]);
        fl[f] = *reinterpret_cast<const f16x8*>(&Tl[off]);
      }
    constexpr int AI[2] = {A0, A1}, BI[2] = {B0, B1};
#pragma unroll
    for (int b = 0; b < NBLK; ++b) {
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[AI[b]], fh[BI[b]], acc[b], 0, 0, 0);
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[AI[b]], fl[BI[b]], acc[b], 0, 0, 0);
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[AI[b]], fh[BI[b]], acc[b], 0, 0, 0);
    }
  }
}

__global__ __launch_bounds__(512, 1) void gram_rows128p_kernel(const float* __restrict__ X, int ldx,
                                                                const float* __restrict__ sc, const float* __restrict__ sh,
                                                                int ldsc, const int* __restrict__ tile_row0,
                                                                const int* __restrict__ tile_nrows,
                                                                const int* __restrict__ tile_group,
                                                                double* __restrict__ Gout, double* __restrict__ Sout) {
  constexpr int K = 128;
  constexpr int PLANE = K * GR_LD;  // halves per plane
  __shared__ __attribute__((aligned(16))) _Float16 T[4 * PLANE];  // [pair 0: hi, lo | pair 1: hi, lo]: 139 KB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, kh = (lane >> 5) * 8;
  const int t = blockIdx.x;
  const int row0 = tile_row0[t], nrows = tile_nrows[t];
  const int grp = tile_group ? tile_group[t] : 0;
  // staging: thread -> (4 consecutive rows, 8 consecutive channels).  The 16 lanes of a store group are 8 channel octets
  // x 2 row quads (tid bits: [0:2] octet low, [3] quad bit 0, [4] octet high, [5:8] quad / 2): with gr_quad's permutation
  // they hit 16 different bank pairs
  constexpr int CPT = 8;
  const int sq = (tid & 7) | ((tid >> 1) & 8);
  const int sr4 = 4 * (((tid >> 3) & 1) | ((tid >> 5) << 1));
  const float* ps = sc + (long)grp * ldsc + sq * CPT;
  const float* ph = sh + (long)grp * ldsc + sq * CPT;
  double gd[2][16];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int e = 0; e < 16; ++e) gd[b][e] = 0.0;
  double sd = 0.0;  // threads < 256: column sum of channel tid & 127 over rows 64 (tid >> 7) .. + 63 of every sub-tile
  const int nsub = (nrows + GR_ROWS - 1) / GR_ROWS;

  f32x4 xr[2][2][4];  // [buffer][channel quad][row]: two sub-tiles in flight
  auto load_sub = [&](int u, f32x4 (&dst)[2][4]) {
    const int r0 = u * GR_ROWS;
    const int nr = min(GR_ROWS, nrows - r0);
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const bool rv = sr4 + rr < nr;  // rows past the end read row 0 of the sub-tile (mapped) and are zeroed when converted
        dst[q][rr] = *reinterpret_cast<const f32x4*>(X + (long)(row0 + r0 + (rv ? sr4 + rr : 0)) * ldx + sq * CPT + 4 * q);
      }
  };
  auto convert_sub = [&](int u, const f32x4 (&src)[2][4], int pair) {
    const int nr = min(GR_ROWS, nrows - u * GR_ROWS);
    const bool full = (nr == GR_ROWS);
    _Float16* Th = T + pair * 2 * PLANE;
    _Float16* Tl = Th + PLANE;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      // (scale / shift of the 8 channels are re-read per sub-tile - L2 hits - instead of held in 16 registers)
      const f32x4 s4 = *reinterpret_cast<const f32x4*>(ps + 4 * q), h4 = *reinterpret_cast<const f32x4*>(ph + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float y[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          y[rr] = __builtin_amdgcn_fmed3f(fmaf(src[q][rr][e], s4[e], h4[e]), 0.f, 65000.f);  // ReLU + fp16 range clamp
          if (!full && sr4 + rr >= nr) y[rr] = 0.f;
        }
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        unsigned h01, l01, h23, l23;
        mm_split2(y[0], y[1], h01, l01);
        mm_split2(y[2], y[3], h23, l23);
        const int c = sq * CPT + 4 * q + e;
        const int o = c * GR_LD + (gr_quad<16>(sr4 >> 2, c) << 2);
        *reinterpret_cast<u32x2*>(&Th[o]) = u32x2{h01, h23};
        *reinterpret_cast<u32x2*>(&Tl[o]) = u32x2{l01, l23};
      }
    }
  };
  // prologue: sub-tiles 0 and 1 requested, 0 converted, 2 requested
  load_sub(0, xr[0]);
  if (nsub > 1) load_sub(1, xr[1]);
  convert_sub(0, xr[0], 0);
  if (nsub > 2) load_sub(2, xr[0]);
  __syncthreads();
  for (int u = 0; u < nsub; ++u) {
    const int pair = u & 1;
    const _Float16* Th = T + pair * 2 * PLANE;
    const _Float16* Tl = Th + PLANE;
    // the next sub-tile into the other plane pair (every wave finished reading it before the previous barrier), then its
    // registers take the sub-tile after next
    if (u + 1 < nsub) {
      if (pair == 0) convert_sub(u + 1, xr[1], 1);
      else convert_sub(u + 1, xr[0], 0);
    }
    if (u + 3 < nsub) {
      if (pair == 0) load_sub(u + 3, xr[1]);
      else load_sub(u + 3, xr[0]);
    }
    if (tid < 256) {  // column sums: thread -> one channel, 64 rows - the partition and order of the four-wave form (same bits)
      const int c = tid & 127, rb = (tid >> 7) * 64;
      float a = 0.f;
#pragma unroll 4
      for (int r8 = 0; r8 < 64; r8 += 8) {
        const f16x8 hh = *reinterpret_cast<const f16x8*>(&Th[c * GR_LD + rb + r8]);
        const f16x8 ll = *reinterpret_cast<const f16x8*>(&Tl[c * GR_LD + rb + r8]);
#pragma unroll
        for (int e = 0; e < 8; ++e) a += (float)hh[e] + (float)ll[e];
      }
      sd += (double)a;
    }
    f32x16 acc[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
    if (wave == 0) gr_blocks2<0x3, 2, 0, 0, 0, 1>(Th, Tl, lr, kh, acc);
    else if (wave == 1) gr_blocks2<0xD, 2, 0, 2, 0, 3>(Th, Tl, lr, kh, acc);
    else if (wave == 2) gr_blocks2<0x6, 2, 1, 1, 1, 2>(Th, Tl, lr, kh, acc);
    else if (wave == 3) gr_blocks2<0xE, 2, 1, 3, 2, 2>(Th, Tl, lr, kh, acc);
    else if (wave == 4) gr_blocks2<0x8, 1, 3, 3, 3, 3>(Th, Tl, lr, kh, acc);
    else if (wave == 5) gr_blocks2<0xC, 1, 2, 3, 2, 3>(Th, Tl, lr, kh, acc);
    if (wave < 6) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) gd[b][e] += (double)acc[b][e];
    }
    __syncthreads();
  }
  double* G = Gout + (long)t * K * K;
  auto put = [&](int b, int bi, int bj) {
#pragma unroll
    for (int e = 0; e < 16; ++e) G[(long)(bi * 32 + mm_acc_row(e, lane)) * K + bj * 32 + lr] = gd[b][e];
  };
  if (wave == 0) { put(0, 0, 0); put(1, 0, 1); }
  else if (wave == 1) { put(0, 0, 2); put(1, 0, 3); }
  else if (wave == 2) { put(0, 1, 1); put(1, 1, 2); }
  else if (wave == 3) { put(0, 1, 3); put(1, 2, 2); }
  else if (wave == 4) put(0, 3, 3);
  else if (wave == 5) put(0, 2, 3);
  // the two row halves of every channel's sum meet in LDS (the planes are free: the loop ended with a barrier)
  double* sx = reinterpret_cast<double*>(T);
  if (tid >= 128 && tid < 256) sx[tid - 128] = sd;
  __syncthreads();
  if (tid < 128) Sout[(long)t * K + tid] = sd + sx[tid];
}

static std::atomic<int> g_gram128_variant{0};
// Test / A-B knob: 0 = automatic (by the number of super-tiles), 1 = the four-wave form with two workgroups per CU,
// 2 = the pipelined eight-wave form.  Gout / Sout do not depend on it (bit for bit).
extern "C" int mmmot_set_gram128_variant(int v) {
  if (v < 0 || v > 2) return MMMOT_EINVAL;
  g_gram128_variant.store(v);
  return MMMOT_OK;
}

// sum of the super-tile partials of every group -> covariance and mean of the group's rows:
// red[g][K*K + K] doubles = Cov(a) (K x K), then E[a] (K), formed ONCE per group here instead of once per workgroup of the
// finalize kernel.  A workgroup owns 64 consecutive elements (one row of the K x K matrix, or 64 of the means); its four
// waves take the super-tiles t = w, w + 4, ... of the group (a detection-aligned tiling has one per detection: 64 - 128
// per sample, and one thread walking them - plus the two column sums its element needs - was a chain of round trips:
// 0.23 ms at 32 LiDAR pairs), four tiles in flight each, and the four partial sums are added in wave order.  The order
// depends on the group's own tiles only (batched == single, bit for bit).
__global__ __launch_bounds__(256) void gram_reduce_kernel(const double* __restrict__ Gp, const double* __restrict__ Sp,
                                                          const int* __restrict__ grp_tile0,
                                                          const int* __restrict__ grp_ntiles,
                                                          const int* __restrict__ grp_count, int K,
                                                          double* __restrict__ red) {
  __shared__ double part[3][4][64];
  const int g = blockIdx.y;
  const int e = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int KK = K * K;  // a multiple of 64: a workgroup's elements are all of the matrix or all of the means
  const int idx0 = blockIdx.x * 64, idx = idx0 + e;
  const bool mat = idx0 < KK;
  const int ei = mat ? idx0 / K : 0;               // the row of this workgroup's matrix elements
  const int cj = mat ? idx - ei * K : idx - KK;    // this thread's column
  const int t0 = grp_tile0[g], nt = grp_ntiles[g];
  // K = 128: gram_rows128_kernel writes the blocks on and above the block diagonal only.  An element of a block above the
  // diagonal is also written to its mirror position below it (same value: the product of the two means commutes); the
  // threads of the blocks below the diagonal have nothing to do (reading (j, i) for (i, j) there was one cache line per
  // lane - the 0.2 ms of this kernel at 32 LiDAR pairs)
  const bool upper = K == 128 && mat && (ei >> 5) < (cj >> 5);
  const bool live = cj < K && !(K == 128 && mat && (ei >> 5) > (cj >> 5));
  double s = 0.0, cs = 0.0, rs = 0.0;
  if (live) {
    const double* gp = Gp + (long)t0 * KK + idx;
    const double* cp = Sp + (long)t0 * K + cj;
    const double* rp = Sp + (long)t0 * K + ei;
    int t = ph;
    for (; t + 12 < nt; t += 16) {
      double v[4], c[4], r[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        v[u] = mat ? gp[(long)(t + 4 * u) * KK] : 0.0;
        c[u] = cp[(long)(t + 4 * u) * K];
        r[u] = mat ? rp[(long)(t + 4 * u) * K] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        s += v[u];
        cs += c[u];
        rs += r[u];
      }
    }
    for (; t < nt; t += 4) {
      if (mat) {
        s += gp[(long)t * KK];
        rs += rp[(long)t * K];
      }
      cs += cp[(long)t * K];
    }
  }
  part[0][ph][e] = s;
  part[1][ph][e] = cs;
  part[2][ph][e] = rs;
  __syncthreads();
  if (ph == 0 && live) {
    const double cnt = (double)grp_count[g];
    const double S = ((part[0][0][e] + part[0][1][e]) + part[0][2][e]) + part[0][3][e];
    const double C = ((part[1][0][e] + part[1][1][e]) + part[1][2][e]) + part[1][3][e];
    const double R = ((part[2][0][e] + part[2][1][e]) + part[2][2][e]) + part[2][3][e];
    const double val = mat ? S / cnt - (R / cnt) * (C / cnt) : C / cnt;
    red[(long)g * (KK + K) + idx] = val;
    if (upper) red[(long)g * (KK + K) + cj * K + ei] = val;
  }
}

// scale / shift of GroupNorm(N, N) applied to v = W a + b: one workgroup per (group, GF_CH output channels);
// Cov(a) lives in LDS as float64 (K = 128: 128 KB).  A wave takes GF_CH / 4 channels, four at a time: every
// covariance element it reads from LDS feeds four float64 FMAs (the weights of the four channels are wave-uniform:
// scalar loads), so the kernel is bound by its 2 G N K^2 float64 FLOPs (~15 us at G = 32) instead of by LDS reads
// and by the 64 workgroups per group that each rebuilt the covariance (round 4: 222 us at G = 32).  Per channel
// the sums run in the order of the one-channel loop they replace (j ascending per row, rows lane, lane + 64, then
// the xor tree): scale / shift are bit for bit the same.
#define GF_CH 64
template <int K>
__global__ __launch_bounds__(256) void gn_finalize_gram_kernel(const double* __restrict__ red,
                                                               const int* __restrict__ grp_count,
                                                               const float* __restrict__ Wm,
                                                               const float* __restrict__ bias, int N,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps,
                                                               float* __restrict__ sc, float* __restrict__ sh) {
  __shared__ double C[K * K];
  __shared__ double m[K];
  constexpr int NR = K / 64;  // rows of C per lane
  const int g = blockIdx.x, n0 = blockIdx.y * GF_CH;
  const double* R = red + (long)g * (K * K + K);  // covariance and mean of the group (gram_reduce_kernel)
  for (int k = threadIdx.x; k < K; k += 256) m[k] = R[K * K + k];
  for (int idx = threadIdx.x; idx < K * K; idx += 256) C[idx] = R[idx];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int c4 = 0; c4 < GF_CH / 4; c4 += 4) {
    const int nb = n0 + wave * (GF_CH / 4) + c4;  // wave-uniform: channels nb .. nb + 3
    if (nb >= N) break;
    const float* w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) w[q] = Wm + (long)(nb + q < N ? nb + q : nb) * K;
    double y[4][NR];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < NR; ++r) y[q][r] = 0.0;
    for (int j0 = 0; j0 < K; j0 += 8) {  // C is symmetric: lanes read consecutive doubles of row j
      float wv[4][8];  // wave-uniform: eight weights of each of the four channels (one scalar load per channel)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) wv[q][e] = w[q][j0 + e];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        double c[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) c[r] = C[(j0 + e) * K + lane + 64 * r];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const double wj = (double)wv[q][e];
#pragma unroll
          for (int r = 0; r < NR; ++r) y[q][r] += c[r] * wj;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = nb + q;
      // var = w . (C w), mean = w . m + b
      double var = 0.0, mean = 0.0;
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int i = lane + 64 * r;
        var += (double)w[q][i] * y[q][r];
        mean += (double)w[q][i] * m[i];
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        var += __shfl_xor(var, o);
        mean += __shfl_xor(mean, o);
      }
      if (lane == 0 && n < N) {
        mean += bias ? (double)bias[n] : 0.0;
        var = var > 0.0 ? var : 0.0;
        const double scv = (double)gamma[n] / sqrt(var + (double)eps);
        sc[(long)g * N + n] = (float)scv;
        sh[(long)g * N + n] = (float)((double)beta[n] - mean * scv);
      }
    }
  }
}

extern "C" int mmmot_gram_rows(const float* X, int ldx, int K, const float* sc, const float* sh, int ldsc,
                               const int* tile_row0, const int* tile_nrows, const int* tile_group, int T,
                               double* Gout, double* Sout, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!X || !sc || !sh || !tile_row0 || !tile_nrows || !Gout || !Sout || T <= 0) return MMMOT_EINVAL;
  if ((K != 64 && K != 128) || ldx % 4 != 0 || ldsc % 4 != 0 || !mm_al16(X) || !mm_al16(sc) || !mm_al16(sh))
    return MMMOT_EINVAL;
  // (measured, tools/bench_gram.py: 245 super-tiles 0.163 ms pipelined / 0.222 four-wave; 1 024: 0.70 / 0.65; 2 048: 1.31 / 1.28 -
  // one workgroup per CU with everything prefetched wins while the launch is a single round of workgroups, two workgroups
  // per CU once there are several)
  const int gv = g_gram128_variant.load();
  const bool pipelined = gv == 2 || (gv == 0 && T <= 2 * mm_num_cu());
  if (K == 128 && pipelined)
    hipLaunchKernelGGL(gram_rows128p_kernel, dim3(T), dim3(512), 0, s, X, ldx, sc, sh, ldsc, tile_row0, tile_nrows,
                       tile_group, Gout, Sout);
  else if (K == 128)
    hipLaunchKernelGGL(gram_rows128_kernel, dim3(T), dim3(GR_THREADS), 0, s, X, ldx, sc, sh, ldsc, tile_row0, tile_nrows,
                       tile_group, Gout, Sout);
  else
    hipLaunchKernelGGL(gram_rows_kernel<64>, dim3(T), dim3(GR_THREADS), 0, s, X, ldx, sc, sh, ldsc, tile_row0, tile_nrows,
                       tile_group, Gout, Sout);
  return mm_check(hipGetLastError());
}

extern "C" int mmmot_gn_finalize_gram(const double* Gp, const double* Sp, const int* grp_tile0, const int* grp_ntiles,
                                      const int* grp_count, int G, int K, const float* W, const float* bias, int N,
                                      const float* gamma, const float* beta, float eps, double* work, float* sc,
                                      float* sh, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!Gp || !Sp || !grp_tile0 || !grp_ntiles || !grp_count || !W || !gamma || !beta || !work || !sc || !sh)
    return MMMOT_EINVAL;
  if ((K != 64 && K != 128) || G <= 0 || N <= 0) return MMMOT_EINVAL;
  hipLaunchKernelGGL(gram_reduce_kernel, dim3((K * K + K + 63) / 64, G), dim3(256), 0, s, Gp, Sp, grp_tile0,
                     grp_ntiles, grp_count, K, work);
  if (K == 128)
    hipLaunchKernelGGL(gn_finalize_gram_kernel<128>, dim3(G, (N + GF_CH - 1) / GF_CH), dim3(256), 0, s, work, grp_count, W, bias,
                       N, gamma, beta, eps, sc, sh);
  else
    hipLaunchKernelGGL(gn_finalize_gram_kernel<64>, dim3(G, (N + GF_CH - 1) / GF_CH), dim3(256), 0, s, work, grp_count, W, bias,
                       N, gamma, beta, eps, sc, sh);
  return mm_check(hipGetLastError());
}

// ---------------------------------------------------------------------------
// The same route for a layer with a GATHERED PER-DETECTION BIAS: v[p] = W a[p] + d[det(p)]  (PointNet_v1.conv1 after the
// 1088 -> 512 split, reference modules/point_net.py:26-28: the 1024 broadcast channels became the bias row dbias[det]).
// With the super-tiles cut along detections (every super-tile t lies inside one detection, tile_det[t]; c_t rows) the
// per-super-tile column sums S_t of mmmot_gram_rows are all that is needed beside the group's moments:
//   mean_n = w_n.m + dbar_n,                      dbar_n = sum_t c_t d_tn / P,   m = S / P,   S = sum_t S_t
//   var_n  = w_n^T Cov w_n + 2 (sum_t d_tn (w_n.S_t) / P - (w_n.m) dbar_n) + (sum_t c_t d_tn^2 / P - dbar_n^2)
// (covariance of the matrix product, cross covariance with the bias, variance of the bias), all in float64.  It replaces
// the statistics pass of the 64 -> 512 GEMM over every point (0.87 ms per 4.2 M points) by a pass over the 64-channel
// input (mmmot_gram_rows, K = 64) and this kernel.  One workgroup per (group, 64 output channels); thread = (channel,
// quarter): a quarter takes 16 rows of the quadratic form and every fourth super-tile; fixed-order combine.
template <int K>
__global__ __launch_bounds__(256) void gn_finalize_gram_dbias_kernel(
    const double* __restrict__ red, const double* __restrict__ Sp, const int* __restrict__ grp_tile0,
    const int* __restrict__ grp_ntiles, const int* __restrict__ grp_count, const int* __restrict__ tile_nrows,
    const int* __restrict__ tile_det, const float* __restrict__ Wm, const float* __restrict__ dbias, int lddb, int N,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float* __restrict__ sc,
    float* __restrict__ sh) {
  __shared__ double C[K * K];
  __shared__ double m[K];
  __shared__ double part[4][5][64];  // [quarter][quad, w.m, cross, sum c d, sum c d^2][channel]
  __shared__ double Sl[32 * K];      // column sums of 32 super-tiles
  const int g = blockIdx.x, n0 = blockIdx.y * 64;
  const double* R = red + (long)g * (K * K + K);  // covariance and mean of the group (gram_reduce_kernel)
  for (int k = threadIdx.x; k < K; k += 256) m[k] = R[K * K + k];
  for (int idx = threadIdx.x; idx < K * K; idx += 256) C[idx] = R[idx];
  __syncthreads();
  const int cl = threadIdx.x & 63, qt = threadIdx.x >> 6;
  const int n = n0 + cl;
  const bool live = n < N;
  float w[K];
#pragma unroll
  for (int k = 0; k < K; ++k) w[k] = live ? Wm[(long)n * K + k] : 0.f;
  // this quarter's rows of w^T C w and of w . m (four partial sums per row: the float64 FMA chain is what this costs)
  double quad = 0.0, wm = 0.0;
  for (int i = qt * (K / 4); i < (qt + 1) * (K / 4); ++i) {
    double y0 = 0.0, y1 = 0.0, y2 = 0.0, y3 = 0.0;
#pragma unroll
    for (int j = 0; j < K; j += 4) {  // C is symmetric; all lanes read the same elements (LDS broadcast)
      y0 += C[i * K + j] * (double)w[j];
      y1 += C[i * K + j + 1] * (double)w[j + 1];
      y2 += C[i * K + j + 2] * (double)w[j + 2];
      y3 += C[i * K + j + 3] * (double)w[j + 3];
    }
    double wi = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) wi = (k == i) ? (double)w[k] : wi;  // w[i] without a dynamically indexed register array
    quad += wi * ((y0 + y1) + (y2 + y3));
    wm += wi * m[i];
  }
  // the group's super-tiles, 32 at a time: their column sums are staged in LDS by the whole workgroup (coalesced; a
  // thread reading 64 doubles per super-tile straight from global memory made this kernel 0.2 ms), then a quarter takes
  // every fourth one: cross term and the bias moments
  double cross = 0.0, sd = 0.0, sd2 = 0.0;
  const int t0 = grp_tile0[g], nt = grp_ntiles[g];
  for (int tc = 0; tc < nt; tc += 32) {
    const int nc = min(32, nt - tc);
    __syncthreads();  // the previous chunk has been read by everyone
    for (int idx = threadIdx.x; idx < nc * K; idx += 256) Sl[idx] = Sp[(long)(t0 + tc) * K + idx];
    __syncthreads();
    for (int tl = qt; tl < nc; tl += 4) {
      const int t = t0 + tc + tl;
      const double c = (double)tile_nrows[t];
      const double d = live ? (double)dbias[(long)tile_det[t] * lddb + n] : 0.0;
      const double* S = Sl + tl * K;
      double u0 = 0.0, u1 = 0.0, u2 = 0.0, u3 = 0.0;
#pragma unroll
      for (int k = 0; k < K; k += 4) {
        u0 += (double)w[k] * S[k];
        u1 += (double)w[k + 1] * S[k + 1];
        u2 += (double)w[k + 2] * S[k + 2];
        u3 += (double)w[k + 3] * S[k + 3];
      }
      cross += d * ((u0 + u1) + (u2 + u3));
      sd += c * d;
      sd2 += c * d * d;
    }
  }
  part[qt][0][cl] = quad;
  part[qt][1][cl] = wm;
  part[qt][2][cl] = cross;
  part[qt][3][cl] = sd;
  part[qt][4][cl] = sd2;
  __syncthreads();
  if (qt == 0 && live) {
    double v[5];
#pragma unroll
    for (int e = 0; e < 5; ++e) v[e] = (part[0][e][cl] + part[1][e][cl]) + (part[2][e][cl] + part[3][e][cl]);
    const double P = (double)grp_count[g];
    const double dbar = v[3] / P;
    const double mean = v[1] + dbar;
    double var = v[0] + 2.0 * (v[2] / P - v[1] * dbar) + (v[4] / P - dbar * dbar);
    var = var > 0.0 ? var : 0.0;
    const double scv = (double)gamma[n] / sqrt(var + (double)eps);
    sc[(long)g * N + n] = (float)scv;
    sh[(long)g * N + n] = (float)((double)beta[n] - mean * scv);
  }
}

extern "C" int mmmot_gn_finalize_gram_dbias(const double* Gp, const double* Sp, const int* grp_tile0, const int* grp_ntiles,
                                            const int* grp_count, int G, const int* tile_nrows, const int* tile_det, int K,
                                            const float* W, const float* dbias, int lddb, int N, const float* gamma,
                                            const float* beta, float eps, double* work, float* sc, float* sh,
                                            void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!Gp || !Sp || !grp_tile0 || !grp_ntiles || !grp_count || !tile_nrows || !tile_det || !W || !dbias || !gamma ||
      !beta || !work || !sc || !sh)
    return MMMOT_EINVAL;
  if (K != 64 || G <= 0 || N <= 0 || lddb < N) return MMMOT_EINVAL;
  hipLaunchKernelGGL(gram_reduce_kernel, dim3((K * K + K + 63) / 64, G), dim3(256), 0, s, Gp, Sp, grp_tile0,
                     grp_ntiles, grp_count, K, work);
  hipLaunchKernelGGL(gn_finalize_gram_dbias_kernel<64>, dim3(G, (N + 63) / 64), dim3(256), 0, s, work, Sp, grp_tile0,
                     grp_ntiles, grp_count, tile_nrows, tile_det, W, dbias, lddb, N, gamma, beta, eps, sc, sh);
  return mm_check(hipGetLastError());
}

