// Training step, third slice: the VGG16-BN trunk in TRAINING mode and its backward
// (reference modules/vgg.py:67-80 under model.train(): conv3x3 -> BatchNorm2d on the statistics of the batch -> ReLU
// (-> MaxPool 2x2), tracking_model.py:50-66).  The inference trunk folds the running statistics into the weights and runs
// on the fp16 matrix cores; training cannot fold (the statistics are those of the batch and receive gradients), so this
// path is separate and plain: fp32 NHWC activations, exact fp32 matrix cores, every pre-BatchNorm tensor kept.
//   forward  : mmmot_conv3x3_raw (conv3x3.hip, bias only) -> mmmot_rows_stats -> mmmot_gn_finalize (per channel over
//              all pixels of the batch = GroupNorm(C, C) with one group) -> mmmot_bn_relu_pool
//   backward : mmmot_maxpool_bwd -> mmmot_gn_bwd_* (backward.hip; BatchNorm backward IS that GroupNorm backward) ->
//              mmmot_conv3x3_wgrad / mmmot_conv3x3_first_wgrad (dW, db) -> mmmot_conv3x3_raw with the flipped, transposed
//              weights (dX)
// Deterministic: per-tile / per-share partial sums, no atomics.
#include "common.h"

// part[t][0][c] = sum over the tile's rows of Y[r][c], part[t][1][c] = sum of (Y[r][c] - tile mean)^2: the statistics
// contract of mmmot_gemm_rows (input of mmmot_gn_finalize) for a tensor that already exists.  One workgroup per tile.
__global__ __launch_bounds__(256) void rows_stats_kernel(const float* __restrict__ Y, int ldy, int C,
                                                         const int* __restrict__ tile_row0,
                                                         const int* __restrict__ tile_nrows, float* __restrict__ part) {
  const int t = blockIdx.x;
  const int row0 = tile_row0[t], nrows = tile_nrows[t];
  const float inv = nrows > 0 ? 1.f / (float)nrows : 0.f;
  const int C4 = C >> 2;
  if (C4 <= 128) {
    // narrow rows (the trunk's 64 .. 512 channels): C / 4 lanes cover a row, so the 256 threads take 256 / (C / 4) rows
    // at a time instead of 16 .. 128 of them walking the tile alone (the 64-channel layers ran at 1 TB/s); the row
    // phases are added in a fixed order through LDS
    __shared__ __attribute__((aligned(16))) float red[256 * 4];
    __shared__ __attribute__((aligned(16))) float smu[128 * 4];
    const int RP = 256 / C4, tid = threadIdx.x, cq = tid % C4, ph = tid / C4;
    const bool live = ph < RP;
    const float* base = Y + (long)row0 * ldy + cq * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (live)
      for (int r = ph; r < nrows; r += RP) s += *reinterpret_cast<const f32x4*>(base + (long)r * ldy);
    *reinterpret_cast<f32x4*>(&red[tid * 4]) = s;
    __syncthreads();
    if (tid < C4) {
      f32x4 tot = *reinterpret_cast<const f32x4*>(&red[tid * 4]);
      for (int k = 1; k < RP; ++k) tot += *reinterpret_cast<const f32x4*>(&red[(k * C4 + tid) * 4]);
      *reinterpret_cast<f32x4*>(&part[((long)t * 2 + 0) * C + tid * 4]) = tot;
      *reinterpret_cast<f32x4*>(&smu[tid * 4]) = tot * inv;
    }
    __syncthreads();
    const f32x4 mu = *reinterpret_cast<const f32x4*>(&smu[cq * 4]);
    f32x4 q = {0.f, 0.f, 0.f, 0.f};
    if (live)
      for (int r = ph; r < nrows; r += RP) {
        const f32x4 d = *reinterpret_cast<const f32x4*>(base + (long)r * ldy) - mu;
        q += d * d;
      }
    *reinterpret_cast<f32x4*>(&red[tid * 4]) = q;
    __syncthreads();
    if (tid < C4) {
      f32x4 tot = *reinterpret_cast<const f32x4*>(&red[tid * 4]);
      for (int k = 1; k < RP; ++k) tot += *reinterpret_cast<const f32x4*>(&red[(k * C4 + tid) * 4]);
      *reinterpret_cast<f32x4*>(&part[((long)t * 2 + 1) * C + tid * 4]) = tot;
    }
    return;
  }
  for (int c = (blockIdx.y * 256 + threadIdx.x) * 4; c < C; c += gridDim.y * 1024) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < nrows; ++r) s += *reinterpret_cast<const f32x4*>(&Y[(long)(row0 + r) * ldy + c]);
    const f32x4 mu = s * inv;
    f32x4 q = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < nrows; ++r) {
      const f32x4 d = *reinterpret_cast<const f32x4*>(&Y[(long)(row0 + r) * ldy + c]) - mu;
      q += d * d;
    }
    *reinterpret_cast<f32x4*>(&part[((long)t * 2 + 0) * C + c]) = s;
    *reinterpret_cast<f32x4*>(&part[((long)t * 2 + 1) * C + c]) = q;
  }
}

extern "C" int mmmot_rows_stats(const float* Y, int ldy, int C, const int* tile_row0, const int* tile_nrows, int T,
                                float* part, void* stream) {
  if (!Y || !tile_row0 || !tile_nrows || !part || T <= 0 || C <= 0 || C % 4 != 0 || ldy % 4 != 0) return MMMOT_EINVAL;
  if (!mm_al16(Y) || !mm_al16(part)) return MMMOT_EINVAL;
  const int gy = (C + 1023) / 1024;
  hipLaunchKernelGGL(rows_stats_kernel, dim3(T, gy), dim3(256), 0, (hipStream_t)stream, Y, ldy, C, tile_row0, tile_nrows,
                     part);
  return mm_check(hipGetLastError());
}

// A[.][c] = relu(Z[.][c] * sc[c] + sh[c]), then (pool != 0) the 2x2 / stride-2 maximum with floor semantics
// (nn.MaxPool2d(2, 2): odd maps lose their last row / column).  Z NHWC [L][H][W][C] -> A [L][Ho][Wo][C].
__global__ __launch_bounds__(256) void bn_relu_pool_kernel(const float* __restrict__ Z, int C,
                                                           const float* __restrict__ sc, const float* __restrict__ sh,
                                                           int L, int H, int W, int pool, float* __restrict__ A) {
  const int Ho = pool ? H >> 1 : H, Wo = pool ? W >> 1 : W;
  const int C4 = C >> 2;
  const long n = (long)L * Ho * Wo * C4;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
    const int c = (int)(idx % C4) * 4;
    const long p = idx / C4;
    const int xo = (int)(p % Wo);
    const long q = p / Wo;
    const int yo = (int)(q % Ho);
    const long crop = q / Ho;
    const f32x4 s4 = *reinterpret_cast<const f32x4*>(&sc[c]);
    const f32x4 h4 = *reinterpret_cast<const f32x4*>(&sh[c]);
    f32x4 o;
    if (pool) {
      o = f32x4{0.f, 0.f, 0.f, 0.f};  // relu outputs are >= 0: the maximum of a window starts from 0
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const long pix = (crop * H + 2 * yo + (k >> 1)) * W + 2 * xo + (k & 1);
        const f32x4 z = *reinterpret_cast<const f32x4*>(&Z[pix * C + c]);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], fmaf(z[e], s4[e], h4[e]));
      }
    } else {
      const f32x4 z = *reinterpret_cast<const f32x4*>(&Z[p * C + c]);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = fmaxf(fmaf(z[e], s4[e], h4[e]), 0.f);
    }
    *reinterpret_cast<f32x4*>(&A[p * C + c]) = o;
  }
}

extern "C" int mmmot_bn_relu_pool(const float* Z, int C, const float* sc, const float* sh, int L, int H, int W, int pool,
                                  float* A, void* stream) {
  if (!Z || !sc || !sh || !A || L <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 != 0) return MMMOT_EINVAL;
  if (pool && (H < 2 || W < 2)) return MMMOT_EINVAL;
  if (!mm_al16(Z) || !mm_al16(A) || !mm_al16(sc) || !mm_al16(sh)) return MMMOT_EINVAL;
  const long n = (long)L * (pool ? H >> 1 : H) * (pool ? W >> 1 : W) * (C / 4);
  const int grid = (int)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
  hipLaunchKernelGGL(bn_relu_pool_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, Z, C, sc, sh, L, H, W, pool, A);
  return mm_check(hipGetLastError());
}

// Backward of the 2x2 max-pool of relu(Z * sc + sh): dA [L][H][W][C] = dP of the window routed to the window's FIRST
// maximum in (0,0), (0,1), (1,0), (1,1) order (PyTorch's max_pool2d backward), zero elsewhere - including the last row /
// column of an odd map, which belongs to no window.  One thread per (window, 4 channels) plus the leftover pixels.
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ Z, int C, const float* __restrict__ sc,
                                                          const float* __restrict__ sh, const float* __restrict__ dP, int L,
                                                          int H, int W, float* __restrict__ dA) {
  const int Hq = (H + 1) >> 1, Wq = (W + 1) >> 1, Ho = H >> 1, Wo = W >> 1;
  const int C4 = C >> 2;
  const long n = (long)L * Hq * Wq * C4;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
    const int c = (int)(idx % C4) * 4;
    const long p = idx / C4;
    const int xq = (int)(p % Wq);
    const long q = p / Wq;
    const int yq = (int)(q % Hq);
    const long crop = q / Hq;
    const bool win = yq < Ho && xq < Wo;
    const f32x4 s4 = *reinterpret_cast<const f32x4*>(&sc[c]);
    const f32x4 h4 = *reinterpret_cast<const f32x4*>(&sh[c]);
    f32x4 a[4], g = {0.f, 0.f, 0.f, 0.f};
    int arg[4] = {0, 0, 0, 0};
    if (win) {
      g = *reinterpret_cast<const f32x4*>(&dP[((crop * Ho + yq) * Wo + xq) * C + c]);
      f32x4 best;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const long pix = (crop * H + 2 * yq + (k >> 1)) * W + 2 * xq + (k & 1);
        const f32x4 z = *reinterpret_cast<const f32x4*>(&Z[pix * C + c]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a[k][e] = fmaxf(fmaf(z[e], s4[e], h4[e]), 0.f);
          if (k == 0 || a[k][e] > best[e]) {  // strictly greater: ties keep the first
            best[e] = a[k][e];
            arg[e] = k;
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int y = 2 * yq + (k >> 1), x = 2 * xq + (k & 1);
      if (y < H && x < W) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (win && arg[e] == k) ? g[e] : 0.f;
        *reinterpret_cast<f32x4*>(&dA[((crop * H + y) * W + x) * C + c]) = o;
      }
    }
  }
}

extern "C" int mmmot_maxpool_bwd(const float* Z, int C, const float* sc, const float* sh, const float* dP, int L, int H,
                                 int W, float* dA, void* stream) {
  if (!Z || !sc || !sh || !dP || !dA || L <= 0 || H < 2 || W < 2 || C <= 0 || C % 4 != 0) return MMMOT_EINVAL;
  if (!mm_al16(Z) || !mm_al16(dP) || !mm_al16(dA) || !mm_al16(sc) || !mm_al16(sh)) return MMMOT_EINVAL;
  const long n = (long)L * ((H + 1) >> 1) * ((W + 1) >> 1) * (C / 4);
  const int grid = (int)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, Z, C, sc, sh, dP, L, H, W, dA);
  return mm_check(hipGetLastError());
}

// Weight gradient of a 3x3 convolution (pad 1): dW[tap][co][ci] = sum over pixels p of dZ[p][co] * A[p + off(tap)][ci]
// (zero outside the image); dZ NHWC [L][H][W][Cout], A NHWC [L][H][W][Cin].  Like gemm_tn_kernel (backward.hip): one
// workgroup = 4 waves = one 64 x 64 tile of one tap's dW, the pixel axis is the K of v_mfma_f32_32x32x2_f32 (two pixels per
// instruction, both operands coalesced 128-byte row reads), pixels split into gridDim.z contiguous shares whose partial
// dW the caller adds.  grid = (Cout / 64, Cin / 64, 9 * nsplit).
__global__ __launch_bounds__(256) void conv3x3_wgrad_kernel(const float* __restrict__ dZ, const float* __restrict__ A, int L,
                                                            int H, int W, int Cin, int Cout, int nsplit,
                                                            float* __restrict__ dW) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = blockIdx.x * 64 + (wave >> 1) * 32, k0 = blockIdx.y * 64 + (wave & 1) * 32;
  const int tap = blockIdx.z / nsplit, share = blockIdx.z - tap * nsplit;
  const int dy = tap / 3 - 1, dx = tap % 3 - 1;
  const int lr = lane & 31, hf = lane >> 5;
  const long P = (long)L * H * W;
  const long p_lo = P * share / nsplit, p_hi = P * (share + 1) / nsplit;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int n = n0 + lr, k = k0 + lr;
  const int HW = H * W;
  for (long p = p_lo; p < p_hi; p += 2) {
    const long pp = p + hf;
    const bool ok = pp < p_hi;
    const long pc = ok ? pp : p_lo;
    const int rem = (int)(pc % HW);
    const int y = rem / W, x = rem - y * W;
    const int yy = y + dy, xx = x + dx;
    const bool in = ok && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
    float dz = dZ[pc * Cout + n];
    float av = A[(in ? pc + (long)dy * W + dx : pc) * Cin + k];
    dz = ok ? dz : 0.f;
    av = in ? av : 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(dz, av, acc, 0, 0, 0);
  }
  float* out = dW + ((long)share * 9 + tap) * Cout * Cin;
#pragma unroll
  for (int e = 0; e < 16; ++e) out[(long)(n0 + mm_acc_row(e, lane)) * Cin + k0 + lr] = acc[e];
}

extern "C" int mmmot_conv3x3_wgrad(const float* dZ, const float* A, int L, int H, int W, int Cin, int Cout, int nsplit,
                                   float* dW, void* stream) {
  if (!dZ || !A || !dW || L <= 0 || H <= 0 || W <= 0 || Cin % 64 != 0 || Cout % 64 != 0 || Cin <= 0 || Cout <= 0)
    return MMMOT_EINVAL;
  if (nsplit < 1 || nsplit > 256) return MMMOT_EINVAL;
  hipLaunchKernelGGL(conv3x3_wgrad_kernel, dim3(Cout / 64, Cin / 64, 9 * nsplit), dim3(256), 0, (hipStream_t)stream, dZ, A, L,
                     H, W, Cin, Cout, nsplit, dW);
  return mm_check(hipGetLastError());
}

// ---- the same weight gradient on the fp16 matrix cores (3-term hi/lo split) ----
// The fp32 kernel above feeds v_mfma_f32_32x32x2_f32 two pixels per instruction from two scalar global loads per lane and
// re-reads dZ / A once per (channel tile, tap): it is load-latency-bound and was most of the 63 ms of the whole-network
// backward (round 3).  Round 4 moved it to the fp16 matrix cores with one workgroup per (channel tile, tap, pixel share):
// load -> convert -> transposed LDS store -> barrier -> MFMA, strictly in sequence, dZ and A re-read and re-converted by
// each of the nine taps (0.1 of the matrix-core ceiling; the 64-channel layers were bound by 5 GB of re-reads).
// Round 6 (this kernel): a workgroup owns a ROW of taps (dy fixed, dx = -1, 0, 1):
//   * a chunk of 64 pixels of dZ (TN output channels, scaled by the power of two that puts max |dZ| at 2^10 - gradients
//     sit in fp16's subnormals otherwise) is staged ONCE for the three taps; the 66 pixels of A the three taps touch are
//     loaded once, converted once, and written as three dx-shifted, image-masked copies (fragment reads stay 16-byte
//     aligned): a third of the loads and conversions per tap, dZ read 3 times per layer instead of 9;
//   * the NEXT chunk's rows are requested (80 registers per thread) before the chunk's 144 MFMAs per wave and land under
//     them; the only serial part is convert + transposed store;
//   * LDS rows are [channel][pixel] hi / lo fp16 planes with the pixel axis as the K = 16 of v_mfma_f32_32x32x16_f16;
//     channel 4 q + e of the tile lives in row e * (T / 4) + q, so that the staging lanes (consecutive q: one coalesced
//     row segment per pixel in global memory) write consecutive LDS rows - 144-byte stride, every bank once - instead
//     of rows four apart (eight lanes per bank); the epilogue undoes the relabelling.
// 4 waves own 2 x 2 quadrants of the TN x TK tile, three accumulator sets each.  grid = (Cout / TN, Cin / TK, 3 * nsplit);
// partial dW per pixel share in the fp32 kernel's layout [share][tap][Cout][Cin].
typedef _Float16 wg_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int wg_u32x2 __attribute__((ext_vector_type(2)));
#define WG_PX 64  // pixels per chunk
#define WG_LD 72  // halves per LDS row: 144 B (36 dwords: 32 consecutive rows x 16 bytes touch every bank once)

template <int TN, int TK>
__global__ __launch_bounds__(256) void conv3x3_wgrad_f16_kernel(const float* __restrict__ dZ, const float* __restrict__ A,
                                                                int L, int H, int W, int Cin, int Cout, int nsplit,
                                                                float* __restrict__ dW, const float* __restrict__ dzamax) {
  constexpr int WN = TN / 64, WK = TK / 64;  // 32 x 32 blocks per wave and axis
  constexpr int QN = TN / 4, QK = TK / 4;    // channel quads per tile
  constexpr int UN = 16 * QN / 256, UK = 16 * QK / 256;  // staging units (4 pixels x 4 channels) per thread: 1 or 2
  __shared__ __attribute__((aligned(16))) _Float16 Dh[TN * WG_LD];
  __shared__ __attribute__((aligned(16))) _Float16 Dl[TN * WG_LD];
  __shared__ __attribute__((aligned(16))) _Float16 Ah[3][TK * WG_LD];
  __shared__ __attribute__((aligned(16))) _Float16 Al[3][TK * WG_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1;
  const int lr = lane & 31, kh = (lane >> 5) * 8;
  const int n0 = blockIdx.x * TN, k0 = blockIdx.y * TK;
  const int trow = blockIdx.z / nsplit, share = blockIdx.z - trow * nsplit;
  const int dy = trow - 1;
  const float amax = dzamax ? *dzamax : 0.f;
  const int shift = mm_pow2_shift(amax, 11);
  const float sd = ldexpf(1.f, shift), inv_sd = ldexpf(1.f, -shift);
  const int P = L * H * W;  // (the launcher checks that the pixel count fits an int: 32-bit index arithmetic below)
  const int nchunk = (P + WG_PX - 1) / WG_PX;
  const int c_lo = (int)((long)nchunk * share / nsplit), c_hi = (int)((long)nchunk * (share + 1) / nsplit);
  const int HW = H * W;

  f32x16 acc[3][WN][WK];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int x = 0; x < WN; ++x)
#pragma unroll
      for (int y = 0; y < WK; ++y)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][x][y][e] = 0.f;

  // staging unit u of a tile with Q channel quads: channel quad u % Q, pixel quad u / Q (lanes: consecutive channel quads)
  f32x4 zr[UN][4], ar[UK][6];
  auto request = [&](int c) {
    const int p0 = c * WG_PX;
#pragma unroll
    for (int i = 0; i < UN; ++i) {
      const int u = tid + 256 * i, cq = u % QN, pq = u / QN;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int p = min(p0 + 4 * pq + r, P - 1);
        zr[i][r] = *reinterpret_cast<const f32x4*>(dZ + (long)p * Cout + n0 + 4 * cq);
      }
    }
#pragma unroll
    for (int i = 0; i < UK; ++i) {
      const int u = tid + 256 * i, cq = u % QK, pq = u / QK;
      const int pf = p0 + 4 * pq + dy * W - 1;  // source pixel of slot 0
#pragma unroll
      for (int sl = 0; sl < 6; ++sl) {
        const int q = min(max(pf + sl, 0), P - 1);
        ar[i][sl] = *reinterpret_cast<const f32x4*>(A + (long)q * Cin + k0 + 4 * cq);
      }
    }
  };
  auto stage = [&](int c) {
    const int p0 = c * WG_PX;
#pragma unroll
    for (int i = 0; i < UN; ++i) {
      const int u = tid + 256 * i, cq = u % QN, pq = u / QN;
      const int nv = P - (p0 + 4 * pq);  // pixels of this quad inside the tensor (<= 0: none)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float y[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) y[r] = r < nv ? zr[i][r][e] * sd : 0.f;
        wg_u32x2 hi, lo;
        unsigned h0, l0, h1, l1;
        mm_split2(y[0], y[1], h0, l0);
        mm_split2(y[2], y[3], h1, l1);
        hi = wg_u32x2{h0, h1};
        lo = wg_u32x2{l0, l1};
        const int off = (e * QN + cq) * WG_LD + 4 * pq;
        *reinterpret_cast<wg_u32x2*>(&Dh[off]) = hi;
        *reinterpret_cast<wg_u32x2*>(&Dl[off]) = lo;
      }
    }
#pragma unroll
    for (int i = 0; i < UK; ++i) {
      const int u = tid + 256 * i, cq = u % QK, pq = u / QK;
      const int pb = p0 + 4 * pq;
      // image coordinates of the unit's four pixels -> per copy (dx = t - 1) a 32-bit keep mask per pixel pair
      // (unsigned 32-bit divisions, selects instead of branches: a 64-bit modulo and four branches stood here)
      unsigned mk[3][2];
      {
        const unsigned pc = (unsigned)min(pb, P - 1);
        const unsigned rem = pc % (unsigned)HW;
        int yy = (int)(rem / (unsigned)W), xx = (int)(rem - (unsigned)yy * (unsigned)W);
        bool ok[3][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool in = pb + r < P && (unsigned)(yy + dy) < (unsigned)H;
#pragma unroll
          for (int t = 0; t < 3; ++t) ok[t][r] = in && (unsigned)(xx + t - 1) < (unsigned)W;
          const bool wrap = xx + 1 == W;
          xx = wrap ? 0 : xx + 1;
          yy = wrap ? (yy + 1 == H ? 0 : yy + 1) : yy;
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          mk[t][0] = (ok[t][0] ? 0xffffu : 0u) | (ok[t][1] ? 0xffff0000u : 0u);
          mk[t][1] = (ok[t][2] ? 0xffffu : 0u) | (ok[t][3] ? 0xffff0000u : 0u);
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v[6];
#pragma unroll
        for (int sl = 0; sl < 6; ++sl) v[sl] = fminf(fmaxf(ar[i][sl][e], -65000.f), 65000.f);
        unsigned hp[5], lp[5];  // packed (slot k, slot k + 1) hi / lo halves
#pragma unroll
        for (int k = 0; k < 5; ++k) mm_split2(v[k], v[k + 1], hp[k], lp[k]);
        const int off = (e * QK + cq) * WG_LD + 4 * pq;
#pragma unroll
        for (int t = 0; t < 3; ++t) {  // copy dx = t - 1: pixel r reads slot r + t
          *reinterpret_cast<wg_u32x2*>(&Ah[t][off]) = wg_u32x2{hp[t] & mk[t][0], hp[t + 2] & mk[t][1]};
          *reinterpret_cast<wg_u32x2*>(&Al[t][off]) = wg_u32x2{lp[t] & mk[t][0], lp[t + 2] & mk[t][1]};
        }
      }
    }
  };

  if (c_lo < c_hi) request(c_lo);
  for (int c = c_lo; c < c_hi; ++c) {
    __syncthreads();  // the previous chunk's fragments are read
    stage(c);
    __syncthreads();
    if (c + 1 < c_hi) request(c + 1);  // lands under the MFMAs below
#pragma unroll
    for (int k16 = 0; k16 < WG_PX / 16; ++k16) {
      wg_f16x8 dh[WN], dl[WN];
#pragma unroll
      for (int x = 0; x < WN; ++x) {
        const int off = ((wi * WN + x) * 32 + lr) * WG_LD + k16 * 16 + kh;
        dh[x] = *reinterpret_cast<const wg_f16x8*>(&Dh[off]);
        dl[x] = *reinterpret_cast<const wg_f16x8*>(&Dl[off]);
      }
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        wg_f16x8 ah[WK], al[WK];
#pragma unroll
        for (int y = 0; y < WK; ++y) {
          const int off = ((wj * WK + y) * 32 + lr) * WG_LD + k16 * 16 + kh;
          ah[y] = *reinterpret_cast<const wg_f16x8*>(&Ah[t][off]);
          al[y] = *reinterpret_cast<const wg_f16x8*>(&Al[t][off]);
        }
#pragma unroll
        for (int x = 0; x < WN; ++x)
#pragma unroll
          for (int y = 0; y < WK; ++y) {
            acc[t][x][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dl[x], ah[y], acc[t][x][y], 0, 0, 0);
            acc[t][x][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dh[x], al[y], acc[t][x][y], 0, 0, 0);
            acc[t][x][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dh[x], ah[y], acc[t][x][y], 0, 0, 0);
          }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    float* out = dW + ((long)share * 9 + trow * 3 + t) * Cout * Cin;
#pragma unroll
    for (int x = 0; x < WN; ++x)
#pragma unroll
      for (int y = 0; y < WK; ++y)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int rn = (wi * WN + x) * 32 + mm_acc_row(e, lane), rk = (wj * WK + y) * 32 + lr;  // LDS rows
          const int n = n0 + 4 * (rn % QN) + rn / QN, k = k0 + 4 * (rk % QK) + rk / QK;         // -> channels
          out[(long)n * Cin + k] = acc[t][x][y][e] * inv_sd;
        }
  }
}

// dzamax: device pointer to max |dZ| (mmmot_absmax), NULL = no scaling.  Same layout and contract as mmmot_conv3x3_wgrad.
extern "C" int mmmot_conv3x3_wgrad_f16(const float* dZ, const float* A, int L, int H, int W, int Cin, int Cout, int nsplit,
                                       float* dW, const float* dzamax, void* stream) {
  if (!dZ || !A || !dW || L <= 0 || H <= 0 || W <= 0 || Cin % 64 != 0 || Cout % 64 != 0 || Cin <= 0 || Cout <= 0)
    return MMMOT_EINVAL;
  if (nsplit < 1 || nsplit > 256 || !mm_al16(dZ) || !mm_al16(A)) return MMMOT_EINVAL;
  if ((long)L * H * W > 0x7fffffffL - 4096) return MMMOT_EINVAL;  // the kernel indexes pixels with 32-bit integers
  hipStream_t s = (hipStream_t)stream;
  const bool n128 = Cout % 128 == 0, k128 = Cin % 128 == 0;
#define WG_LAUNCH(TNV, TKV)                                                                                              \
  hipLaunchKernelGGL((conv3x3_wgrad_f16_kernel<TNV, TKV>), dim3(Cout / TNV, Cin / TKV, 3 * nsplit), dim3(256), 0, s, dZ, A, \
                     L, H, W, Cin, Cout, nsplit, dW, dzamax)
  if (n128 && k128) WG_LAUNCH(128, 128);
  else if (n128) WG_LAUNCH(128, 64);
  else if (k128) WG_LAUNCH(64, 128);
  else WG_LAUNCH(64, 64);
#undef WG_LAUNCH
  return mm_check(hipGetLastError());
}

// First layer (3 input channels, NCHW crops as delivered): PW[b][co][k] partial sums of dW1[co][k = tap * 3 + colour]
// = sum_p dZ[p][co] * X[crop][colour][p + off(tap)] over block b's pixels (28 columns: 27 + the bias gradient sum_p dZ).
// VALU, like the forward's K = 27 case is not matrix-core shaped; float64 accumulation (the sums cancel: BatchNorm
// backward makes sum_p dZ = 0).  grid = nblocks workgroups, each 256 threads = 64 channels x 4 pixel phases.
__global__ __launch_bounds__(256) void conv3x3_first_wgrad_kernel(const float* __restrict__ dZ, const float* __restrict__ X,
                                                                  int L, int H, int W, float* __restrict__ PW) {
  // A wave = one pixel at a time x 64 channels: the pixel, its image coordinates and the 27 input values of its window are
  // WAVE-UNIFORM (scalar loads, scalar bounds tests, no per-lane address arithmetic); the coordinates advance
  // incrementally (no division in the loop).  The earlier form computed them per lane behind per-tap branches and took
  // 1.4 ms per training step for 2 G fp64 fmas (latency-bound).
  __shared__ float red[4][64][28];  // a wave's fp64 sums, rounded once (28 KB: five workgroups per CU; as doubles two)
  const int c = threadIdx.x & 63;
  const int ph = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int P = L * H * W;  // the launcher checks that this fits an int
  const int p_lo = (int)((long)P * blockIdx.x / gridDim.x), p_hi = (int)((long)P * (blockIdx.x + 1) / gridDim.x);
  double acc[28];
#pragma unroll
  for (int k = 0; k < 28; ++k) acc[k] = 0.0;
  const int HW = H * W;
  int p = p_lo + ph;
  int crop = p / HW, y = (p - crop * HW) / W, x = p - crop * HW - y * W;
  float dn = p < p_hi ? dZ[(long)p * 64 + c] : 0.f;
  for (; p < p_hi; p += 4) {
    const double d = (double)dn;
    if (p + 4 < p_hi) dn = dZ[(long)(p + 4) * 64 + c];  // the next pixel's row is in flight during this one's 55 VALU ops
    float xv[27];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
      const bool in = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
      const int base = (crop * 3 * H + (in ? yy : y)) * W + (in ? xx : x);
#pragma unroll
      for (int col = 0; col < 3; ++col) {
        const float v = X[base + col * HW];
        xv[tap * 3 + col] = in ? v : 0.f;
      }
    }
#pragma unroll
    for (int k = 0; k < 27; ++k) acc[k] = fma(d, (double)xv[k], acc[k]);
    acc[27] += d;
    x += 4;
    while (x >= W) {
      x -= W;
      if (++y == H) {
        y = 0;
        ++crop;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 28; ++k) red[ph][c][k] = (float)acc[k];
  __syncthreads();
  for (int idx = threadIdx.x; idx < 64 * 28; idx += 256) {
    const int cc = idx / 28, k = idx - cc * 28;
    PW[(long)blockIdx.x * 64 * 28 + idx] = (red[0][cc][k] + red[1][cc][k]) + (red[2][cc][k] + red[3][cc][k]);
  }
}

extern "C" int mmmot_conv3x3_first_wgrad(const float* dZ, const float* X, int L, int H, int W, float* PW, int nblocks,
                                         void* stream) {
  if (!dZ || !X || !PW || L <= 0 || H <= 0 || W <= 0 || nblocks <= 0 || nblocks > 65535) return MMMOT_EINVAL;
  if ((long)L * H * W * 3 > 0x7fffffffL) return MMMOT_EINVAL;  // int pixel / input indices in the kernel
  hipLaunchKernelGGL(conv3x3_first_wgrad_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, dZ, X, L, H, W, PW);
  return mm_check(hipGetLastError());
}
