// VGG16-BN trunk layer on gfx950: implicit-GEMM 3x3 convolution on the exact
// fp32 MFMA (v_mfma_f32_32x32x2_f32), eval-BatchNorm folded into weights/bias by
// the host, ReLU and the following 2x2 max-pool fused into the epilogue.
// Replaces conv2d + batch_norm + relu_ (+ max_pool2d) of reference
// modules/vgg.py:67-80 (stage regrouping: modules/appear_net.py:130-157).
//
// GEMM view: rows = output pixels, cols = output channels, K = 9*Cin walked
// tap-major.  A tile row r of workgroup tile t is pixel number m = 128 t + r in
// "quad order": q = m>>2 enumerates the 2x2 pooling windows in (crop, y/2, x/2)
// raster order and m&3 the pixel inside the window.  With the 32x32 MFMA C
// layout (lane owns rows 4*(l>>5)+(e&3)+8*(e>>2)) the four pixels of a window
// land in four consecutive accumulator registers of ONE lane, so the max-pool
// is an in-register max and the pooled NHWC store index is simply q.
#include "common.h"

// OUT16: write the output in the hl16 split-half format consumed by conv3x3_hl16.hip
// (unit u of a pixel = [hi of channels 8u..8u+7 | lo of channels 8u..8u+7], 2-byte stores).
// RAW: bias only - no ReLU, no folded BatchNorm - the pre-BatchNorm tensor of the training-mode trunk (train_vgg.hip) and,
// with flipped / transposed weights, its input gradient.
template <int BN, bool FIRST, bool POOL, bool OUT16 = false, bool RAW = false>
__global__ __launch_bounds__(MM_THREADS, 2) void conv3x3_kernel(
    const float* __restrict__ in, const float* __restrict__ wp, const float* __restrict__ bias,
    float* __restrict__ out, int L, int H, int W, int Cin, int Cout, int Mtot, int ntn) {
  constexpr int WM = (BN == 128) ? 2 : 4;  // waves along rows
  constexpr int WN = 4 / WM;               // waves along cols
  constexpr int TM = MM_BM / (WM * 32);
  constexpr int TN = BN / (WN * 32);
  constexpr int BLD = BN / 32;             // float4 weight loads per thread per stage

  // OUT16 stages the output tile through LDS (fp32 [128][BN+4]) for 16-byte hl16 stores
  constexpr int SMEM_FLOATS =
      (OUT16 && MM_BM * (BN + 4) > (MM_BM + BN) * MM_LDT) ? MM_BM * (BN + 4) : (MM_BM + BN) * MM_LDT;
  __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
  float* As = smem;
  float* Bs = smem + MM_BM * MM_LDT;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  const int lid = mm_xcd_remap(blockIdx.x, gridDim.x);
  const int mt = lid / ntn, nt = lid % ntn;
  const int n0 = nt * BN;

  // quads are enumerated over the map padded to even sides (odd maps: the last row / column of quads is half outside
  // and masked); a pooled layer keeps only the windows that lie inside - the floor of nn.MaxPool2d(2, 2), vgg.py:72
  const int Hq = (H + 1) >> 1, Wq = (W + 1) >> 1;
  const int Hf = H >> 1, Wf = W >> 1;
  const int lrow = tid >> 3;   // 0..31: staging row (plus 32*i)
  const int kq = tid & 7;      // which float4 of the 32-wide k slab

  // per-thread pixel coordinates of its 4 staging rows
  int py[4], px[4];
  long pbase[4];
  bool pval[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = mt * MM_BM + lrow + 32 * i;
    pval[i] = m < Mtot;
    const int q = m >> 2, sub = m & 3;
    const int crop = q / (Hq * Wq);
    const int rem = q - crop * (Hq * Wq);
    const int yq = rem / Wq, xq = rem - yq * Wq;
    py[i] = 2 * yq + (sub >> 1);
    px[i] = 2 * xq + (sub & 1);
    pval[i] = pval[i] && py[i] < H && px[i] < W;
    pbase[i] = FIRST ? (long)crop * 3 * H * W : ((long)crop * H + py[i]) * W + px[i];
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[tm][tn][e] = 0.f;

  f32x4 ra[4], rb[BLD];

  auto load_stage = [&](int it) {
    if constexpr (FIRST) {
      // single stage: k = tap*3 + c for k < 27, zero above
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k = kq * 4 + e;
          const int tap = k / 3, c = k - tap * 3;
          const int yy = py[i] + tap / 3 - 1, xx = px[i] + tap % 3 - 1;
          if (pval[i] && k < 27 && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
            v[e] = in[pbase[i] + ((long)c * H + yy) * W + xx];
        }
        ra[i] = v;
      }
#pragma unroll
      for (int i = 0; i < BLD; ++i)
        rb[i] = *reinterpret_cast<const f32x4*>(&wp[(long)(n0 + lrow + 32 * i) * 32 + kq * 4]);
    } else {
      const int cpt = Cin / MM_BK;  // k slabs per tap
      const int tap = it / cpt;
      const int c0 = (it - tap * cpt) * MM_BK;
      const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int yy = py[i] + dy, xx = px[i] + dx;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (pval[i] && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
          v = *reinterpret_cast<const f32x4*>(&in[(pbase[i] + dy * W + dx) * Cin + c0 + kq * 4]);
        ra[i] = v;
      }
#pragma unroll
      for (int i = 0; i < BLD; ++i)
        rb[i] = *reinterpret_cast<const f32x4*>(
            &wp[((long)tap * Cout + n0 + lrow + 32 * i) * Cin + c0 + kq * 4]);
    }
  };

  const int nk = FIRST ? 1 : 9 * (Cin / MM_BK);
  load_stage(0);
  for (int it = 0; it < nk; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<f32x4*>(&As[(lrow + 32 * i) * MM_LDT + kq * 4]) = ra[i];
#pragma unroll
    for (int i = 0; i < BLD; ++i)
      *reinterpret_cast<f32x4*>(&Bs[(lrow + 32 * i) * MM_LDT + kq * 4]) = rb[i];
    __syncthreads();
    if (it + 1 < nk) load_stage(it + 1);  // global loads fly under the MFMAs
    mm_stage<TM, TN>(As, Bs, acc, wm * TM * 32, wn * TN * 32, lane);
    __syncthreads();
  }

  if constexpr (OUT16 && !POOL) {
    // hl16 output: accumulators -> LDS -> bias + ReLU -> hi/lo split -> two 16-byte stores per 8 channels
    // (per-lane 2-byte stores made this layer store-issue bound: 0.67 ms per 2 frame pairs at cfg3)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    constexpr int CLD = BN + 4;
    float* Cs = smem;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          Cs[(wm * TM * 32 + tm * 32 + mm_acc_row(e, lane)) * CLD + wn * TN * 32 + tn * 32 + (lane & 31)] =
              acc[tm][tn][e];
    __syncthreads();
    constexpr int UN = BN / 8;
    constexpr int RSTEP = MM_THREADS / UN;  // rows per sweep; the thread's channel unit is fixed
    u32x4* out16 = reinterpret_cast<u32x4*>(out);
    const int u = tid % UN, r0 = tid / UN;
    float bv[8];  // loaded once (a load inside the store loop costs an L2 round trip per iteration)
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = bias[n0 + u * 8 + e];
#pragma unroll 4
    for (int r = r0; r < MM_BM; r += RSTEP) {
      const int m = mt * MM_BM + r;
      if (m < Mtot) {
        const int q = m >> 2, sub = m & 3;
        const int crop = q / (Hq * Wq);
        const int rem = q - crop * (Hq * Wq);
        const int yq = rem / Wq, xq = rem - yq * Wq;
        const long pix = ((long)crop * H + 2 * yq + (sub >> 1)) * W + 2 * xq + (sub & 1);
        if (2 * yq + (sub >> 1) >= H || 2 * xq + (sub & 1) >= W) continue;
        f16x8 hh, ll;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float val = fminf(fmaxf(Cs[r * CLD + u * 8 + e] + bv[e], 0.f), 65000.f);
          hh[e] = (_Float16)val;
          ll[e] = (_Float16)(val - (float)hh[e]);
        }
        u32x4* o = out16 + (pix * (Cout >> 3) + (n0 >> 3) + u) * 2;
        o[0] = __builtin_bit_cast(u32x4, hh);
        o[1] = __builtin_bit_cast(u32x4, ll);
      }
    }
    return;
  }
  // epilogue: bias + ReLU (+ 2x2 max-pool), NHWC store
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int n = n0 + wn * TN * 32 + tn * 32 + (lane & 31);
    const float bv = bias[n];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int rbase = wm * TM * 32 + tm * 32;
      if constexpr (POOL) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int r = rbase + 8 * g + 4 * (lane >> 5);  // first row of the quad
          const int m = mt * MM_BM + r;
          const float v = fmaxf(fmaxf(acc[tm][tn][4 * g], acc[tm][tn][4 * g + 1]),
                                fmaxf(acc[tm][tn][4 * g + 2], acc[tm][tn][4 * g + 3]));
          if (m < Mtot) {
            const int q = m >> 2;
            const int crop = q / (Hq * Wq);
            const int rem = q - crop * (Hq * Wq);
            const int yq = rem / Wq, xq = rem - yq * Wq;
            if (yq < Hf && xq < Wf) out[(((long)crop * Hf + yq) * Wf + xq) * Cout + n] = fmaxf(v + bv, 0.f);
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = mt * MM_BM + rbase + mm_acc_row(e, lane);
          if (m < Mtot) {
            const int q = m >> 2, sub = m & 3;
            const int crop = q / (Hq * Wq);
            const int rem = q - crop * (Hq * Wq);
            const int yq = rem / Wq, xq = rem - yq * Wq;
            const long pix = ((long)crop * H + 2 * yq + (sub >> 1)) * W + 2 * xq + (sub & 1);
            if (2 * yq + (sub >> 1) >= H || 2 * xq + (sub & 1) >= W) continue;
            const float val = RAW ? acc[tm][tn][e] + bv : fmaxf(acc[tm][tn][e] + bv, 0.f);
            if constexpr (OUT16) {
              _Float16* o16 = reinterpret_cast<_Float16*>(out) + pix * Cout * 2 + (n >> 3) * 16 + (n & 7);
              const _Float16 h = (_Float16)val;
              o16[0] = h;
              o16[8] = (_Float16)(val - (float)h);
            } else {
              out[pix * Cout + n] = val;
            }
          }
        }
      }
    }
  }
}

template <int BN, bool FIRST, bool POOL, bool OUT16 = false, bool RAW = false>
static int launch_conv(const float* in, const float* wp, const float* bias, float* out, int L, int H,
                       int W, int Cin, int Cout, hipStream_t s) {
  const int Mtot = L * 4 * ((H + 1) >> 1) * ((W + 1) >> 1);  // pixels of the maps padded to even sides (quad order)
  const int ntm = (Mtot + MM_BM - 1) / MM_BM;
  const int ntn = Cout / BN;
  hipLaunchKernelGGL((conv3x3_kernel<BN, FIRST, POOL, OUT16, RAW>), dim3(ntm * ntn), dim3(MM_THREADS), 0, s, in, wp,
                     bias, out, L, H, W, Cin, Cout, Mtot, ntn);
  return mm_check(hipGetLastError());
}

extern "C" int mmmot_conv3x3_bn_relu(const float* in, const float* wp, const float* bias, float* out,
                                     int L, int H, int W, int Cin, int Cout, int first, int pool,
                                     void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!in || !wp || !bias || !out || L <= 0 || H <= 0 || W <= 0) return MMMOT_EINVAL;
  if ((Cout % 64) != 0) return MMMOT_EINVAL;
  if (!mm_al16(in) || !mm_al16(wp) || !mm_al16(out)) return MMMOT_EINVAL;
  if ((long)L * (H + 1) * (W + 1) >= (1L << 31) - MM_BM) return MMMOT_EINVAL;
  if (first) {
    if (Cin != 3) return MMMOT_EINVAL;
    if (Cout % 128 == 0)
      return pool ? launch_conv<128, true, true>(in, wp, bias, out, L, H, W, Cin, Cout, s)
                  : launch_conv<128, true, false>(in, wp, bias, out, L, H, W, Cin, Cout, s);
    return pool ? launch_conv<64, true, true>(in, wp, bias, out, L, H, W, Cin, Cout, s)
                : launch_conv<64, true, false>(in, wp, bias, out, L, H, W, Cin, Cout, s);
  }
  if (Cin % MM_BK != 0) return MMMOT_EINVAL;
  if (Cout % 128 == 0)
    return pool ? launch_conv<128, false, true>(in, wp, bias, out, L, H, W, Cin, Cout, s)
                : launch_conv<128, false, false>(in, wp, bias, out, L, H, W, Cin, Cout, s);
  return pool ? launch_conv<64, false, true>(in, wp, bias, out, L, H, W, Cin, Cout, s)
              : launch_conv<64, false, false>(in, wp, bias, out, L, H, W, Cin, Cout, s);
}

// conv3x3 (pad 1) + bias, nothing else: out NHWC [L][H][W][Cout] = conv(in) + bias.  first / in / wp as in
// mmmot_conv3x3_bn_relu (wp holds the UNFOLDED convolution weights).
extern "C" int mmmot_conv3x3_raw(const float* in, const float* wp, const float* bias, float* out, int L, int H, int W,
                                 int Cin, int Cout, int first, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!in || !wp || !bias || !out || L <= 0 || H <= 0 || W <= 0 || (Cout % 64) != 0) return MMMOT_EINVAL;
  if (!mm_al16(in) || !mm_al16(wp) || !mm_al16(out)) return MMMOT_EINVAL;
  if ((long)L * (H + 1) * (W + 1) >= (1L << 31) - MM_BM) return MMMOT_EINVAL;
  if (first) {
    if (Cin != 3) return MMMOT_EINVAL;
    return (Cout % 128 == 0) ? launch_conv<128, true, false, false, true>(in, wp, bias, out, L, H, W, Cin, Cout, s)
                             : launch_conv<64, true, false, false, true>(in, wp, bias, out, L, H, W, Cin, Cout, s);
  }
  if (Cin % MM_BK != 0) return MMMOT_EINVAL;
  return (Cout % 128 == 0) ? launch_conv<128, false, false, false, true>(in, wp, bias, out, L, H, W, Cin, Cout, s)
                           : launch_conv<64, false, false, false, true>(in, wp, bias, out, L, H, W, Cin, Cout, s);
}

// First trunk layer (NCHW fp32 crops in, K = 27) with hl16 output for the fp16-split trunk.
extern "C" int mmmot_conv3x3_first_hl16(const float* in, const float* wp, const float* bias, void* out, int L,
                                        int H, int W, int Cout, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!in || !wp || !bias || !out || L <= 0 || H <= 0 || W <= 0) return MMMOT_EINVAL;
  if ((H & 1) || (W & 1) || (Cout % 64) != 0) return MMMOT_EINVAL;
  if (!mm_al16(in) || !mm_al16(wp) || !mm_al16(out)) return MMMOT_EINVAL;
  if ((long)L * H * W >= (1L << 31) - MM_BM) return MMMOT_EINVAL;
  if (Cout % 128 == 0) return launch_conv<128, true, false, true>(in, wp, bias, (float*)out, L, H, W, 3, Cout, s);
  return launch_conv<64, true, false, true>(in, wp, bias, (float*)out, L, H, W, 3, Cout, s);
}
