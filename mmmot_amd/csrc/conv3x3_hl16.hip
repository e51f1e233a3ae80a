// VGG trunk layer on the fp16 matrix cores with fp32-class accuracy ("hl16" = hi/lo split-half).
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
// Cout/128 times in the consumer); weights are split on the host after scaling by 2^wshift so
// that their lo parts stay in the fp16 normal range (the epilogue multiplies by 2^-wshift, exact).
//
// Same implicit-GEMM structure as conv3x3.hip: rows = output pixels in quad order (2x2 pooling
// windows contiguous), 128 x BN tile per 4-wave workgroup, K walked tap-major in 64-channel slabs
// through padded LDS (row stride 144 B -> conflict-free ds_read_b128), register prefetch of the
// next slab under the MFMAs.  Epilogue: accumulators -> LDS -> (2x2 max) + bias + ReLU -> split ->
// two 16-byte stores per 8 channels.
#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

__device__ u32x4 mm_zero_page[2];  // zero-initialised: source of out-of-image (padding) taps

#define HL_BK 64   // channels per LDS stage
#define HL_LDT 72  // halves per LDS row: 144 B

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

template <int BN, bool POOL>
__global__ __launch_bounds__(MM_THREADS, 2) void conv3x3_hl16_kernel(
    const u32x4* __restrict__ in, const u32x4* __restrict__ wp, const float* __restrict__ bias,
    u32x4* __restrict__ out, int L, int H, int W, int Cin, int Cout, int Mtot, int ntm, int ntn, float oscale) {
  constexpr int WM = (BN == 128) ? 2 : 4;
  constexpr int WN = 4 / WM;
  constexpr int TM = MM_BM / (WM * 32);
  constexpr int TN = BN / (WN * 32);
  constexpr int BLD = BN / 32;
  constexpr int PLANE_A = MM_BM * HL_LDT;  // halves
  constexpr int PLANE_B = BN * HL_LDT;

  __shared__ __attribute__((aligned(16))) _Float16 smem[2 * PLANE_A + 2 * PLANE_B];
  _Float16* As_hi = smem;
  _Float16* As_lo = smem + PLANE_A;
  _Float16* Bs_hi = smem + 2 * PLANE_A;
  _Float16* Bs_lo = smem + 2 * PLANE_A + PLANE_B;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  // XCD-aware order, channel-tile-major inside an XCD's chunk of pixel tiles: the workgroups that are
  // co-resident on one XCD share ONE weight slab (<= 2.4 MB, L2 resident) and stream distinct
  // activation tiles, instead of cycling all Cout/BN slabs through the 4 MiB L2 (profiles/README.md).
  const int nwg = gridDim.x;
  const int lid = mm_xcd_remap(blockIdx.x, nwg);
  const int xq = nwg >> 3, xr = nwg & 7;
  const int xcd = blockIdx.x & 7;
  const int cbase = (xcd < xr) ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq;  // first lid of this XCD
  const int clen = (xcd < xr) ? xq + 1 : xq;
  int mt, nt;
  {
    // the chunk [cbase, cbase+clen) covers whole pixel tiles when ntn | clen; otherwise fall back to
    // the plain pixel-major order (still correct, just less L2 friendly)
    if (clen % ntn == 0 && cbase % ntn == 0) {
      const int mcount = clen / ntn;
      const int s = lid - cbase;
      nt = s / mcount;
      mt = cbase / ntn + s % mcount;
    } else {
      mt = lid / ntn;
      nt = lid % ntn;
    }
  }
  const int n0 = nt * BN;

  const int Hq = H >> 1, Wq = W >> 1;
  const int lrow = tid >> 3;  // 0..31 (+32 i)
  const int ku = tid & 7;     // 8-channel unit inside the 64-channel slab
  const int cin8 = Cin >> 3;  // units per input pixel

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
    const int yq = rem / Wq, xqq = rem - yq * Wq;
    py[i] = 2 * yq + (sub >> 1);
    px[i] = 2 * xqq + (sub & 1);
    pbase[i] = ((long)crop * H + py[i]) * W + px[i];
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[tm][tn][e] = 0.f;

  u32x4 ra[4][2], rb[BLD][2];
  const u32x4 z4 = {0u, 0u, 0u, 0u};
  const int cpt = Cin / HL_BK;  // slabs per tap

  auto load_stage = [&](int it) {
    const int tap = it / cpt;
    const int u0 = (it - tap * cpt) * (HL_BK / 8) + ku;  // unit index inside the pixel / weight row
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // branch-free and select-free: out-of-image taps read a 32-byte zero page, so the loaded
      // registers are consumed only by the ds_write of the NEXT iteration (the wait sits there).
      // (A conditional load made hipcc keep ra[] in scratch; a select made it wait for the data here.)
      const int yy = py[i] + dy, xx = px[i] + dx;
      const bool ok = pval[i] && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
      const u32x4* p = ok ? in + ((pbase[i] + dy * W + dx) * cin8 + u0) * 2 : mm_zero_page;
      ra[i][0] = p[0];
      ra[i][1] = p[1];
    }
#pragma unroll
    for (int i = 0; i < BLD; ++i) {
      const u32x4* p = wp + (((long)tap * Cout + n0 + lrow + 32 * i) * cin8 + u0) * 2;
      rb[i][0] = p[0];
      rb[i][1] = p[1];
    }
  };

  const int nk = 9 * cpt;
  const int lr = lane & 31;
  const int kh = (lane >> 5) * 8;
  load_stage(0);
  for (int it = 0; it < nk; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<u32x4*>(&As_hi[(lrow + 32 * i) * HL_LDT + ku * 8]) = ra[i][0];
      *reinterpret_cast<u32x4*>(&As_lo[(lrow + 32 * i) * HL_LDT + ku * 8]) = ra[i][1];
    }
#pragma unroll
    for (int i = 0; i < BLD; ++i) {
      *reinterpret_cast<u32x4*>(&Bs_hi[(lrow + 32 * i) * HL_LDT + ku * 8]) = rb[i][0];
      *reinterpret_cast<u32x4*>(&Bs_lo[(lrow + 32 * i) * HL_LDT + ku * 8]) = rb[i][1];
    }
    __syncthreads();
    if (it + 1 < nk) load_stage(it + 1);
#pragma unroll
    for (int k16 = 0; k16 < HL_BK / 16; ++k16) {
      f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const int off = (wm * TM * 32 + tm * 32 + lr) * HL_LDT + k16 * 16 + kh;
        ah[tm] = *reinterpret_cast<const f16x8*>(&As_hi[off]);
        al[tm] = *reinterpret_cast<const f16x8*>(&As_lo[off]);
      }
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int off = (wn * TN * 32 + tn * 32 + lr) * HL_LDT + k16 * 16 + kh;
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
    __syncthreads();
  }

  // ---- epilogue: accumulators -> LDS (fp32 [128][BN+4]) -> pool/bias/relu/split -> 32-byte units ----
  constexpr int CLD = BN + 4;
  static_assert(MM_BM * CLD * 4 <= (int)sizeof(smem), "epilogue staging must fit the K-loop LDS");
  float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        Cs[(wm * TM * 32 + tm * 32 + mm_acc_row(e, lane)) * CLD + wn * TN * 32 + tn * 32 + lr] = acc[tm][tn][e];
  __syncthreads();
  constexpr int UN = BN / 8;  // 8-channel units per tile row
  const int cout8 = Cout >> 3;
  if constexpr (POOL) {
    for (int w = tid; w < (MM_BM / 4) * UN; w += MM_THREADS) {
      const int qd = w / UN, u = w - qd * UN;
      const int m = mt * MM_BM + qd * 4;
      if (m < Mtot) {
        const float* c = &Cs[(qd * 4) * CLD + u * 8];
        f32x8 v = *reinterpret_cast<const f32x8*>(c);
#pragma unroll
        for (int r = 1; r < 4; ++r) {
          const f32x8 w2 = *reinterpret_cast<const f32x8*>(c + r * CLD);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], w2[e]);
        }
        const f32x8 bv = *reinterpret_cast<const f32x8*>(&bias[n0 + u * 8]);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(fmaf(v[e], oscale, bv[e]), 0.f);
        u32x4 hi, lo;
        hl_split8(v, hi, lo);
        u32x4* o = out + ((long)(m >> 2) * cout8 + (n0 >> 3) + u) * 2;
        o[0] = hi;
        o[1] = lo;
      }
    }
  } else {
    for (int w = tid; w < MM_BM * UN; w += MM_THREADS) {
      const int r = w / UN, u = w - r * UN;
      const int m = mt * MM_BM + r;
      if (m < Mtot) {
        f32x8 v = *reinterpret_cast<const f32x8*>(&Cs[r * CLD + u * 8]);
        const f32x8 bv = *reinterpret_cast<const f32x8*>(&bias[n0 + u * 8]);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(fmaf(v[e], oscale, bv[e]), 0.f);
        const int q = m >> 2, sub = m & 3;
        const int crop = q / (Hq * Wq);
        const int rem = q - crop * (Hq * Wq);
        const int yq = rem / Wq, xqq = rem - yq * Wq;
        const long pix = ((long)crop * H + 2 * yq + (sub >> 1)) * W + 2 * xqq + (sub & 1);
        u32x4 hi, lo;
        hl_split8(v, hi, lo);
        u32x4* o = out + (pix * cout8 + (n0 >> 3) + u) * 2;
        o[0] = hi;
        o[1] = lo;
      }
    }
  }
}

template <int BN, bool POOL>
static int launch_hl(const void* in, const void* wp, const float* bias, void* out, int L, int H, int W, int Cin,
                     int Cout, float oscale, hipStream_t s) {
  const int Mtot = L * H * W;
  const int ntm = (Mtot + MM_BM - 1) / MM_BM;
  const int ntn = Cout / BN;
  hipLaunchKernelGGL((conv3x3_hl16_kernel<BN, POOL>), dim3(ntm * ntn), dim3(MM_THREADS), 0, s, (const u32x4*)in,
                     (const u32x4*)wp, bias, (u32x4*)out, L, H, W, Cin, Cout, Mtot, ntm, ntn, oscale);
  return mm_check(hipGetLastError());
}

extern "C" int mmmot_conv3x3_bn_relu_hl16(const void* in, const void* wp, const float* bias, void* out, int L,
                                          int H, int W, int Cin, int Cout, int pool, float oscale, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!in || !wp || !bias || !out || L <= 0 || H <= 0 || W <= 0) return MMMOT_EINVAL;
  if ((H & 1) || (W & 1) || Cin % HL_BK != 0 || Cout % 64 != 0) return MMMOT_EINVAL;
  if (!mm_al16(in) || !mm_al16(wp) || !mm_al16(out)) return MMMOT_EINVAL;
  if ((long)L * H * W >= (1L << 31) - MM_BM) return MMMOT_EINVAL;
  if (Cout % 128 == 0)
    return pool ? launch_hl<128, true>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s)
                : launch_hl<128, false>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
  return pool ? launch_hl<64, true>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s)
              : launch_hl<64, false>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
}

// ---------------------------------------------------------------------------
// fp32 rows <-> hl16 rows (used for the first layer's output handoff in tests and by the host
// weight packer's device-side check; C % 8 == 0).
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
