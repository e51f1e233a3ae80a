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
// Implicit GEMM: rows = output pixels in quad order (2x2 pooling windows contiguous), BM x BN tile per
// workgroup (BM = 128 with 4 waves or 256 with 8 waves, each wave a 64x64 or 32x64 sub-tile), K walked
// channel-slab-major / tap-minor in 64-channel slabs through padded LDS (row stride 144 B ->
// conflict-free ds_read_b128), register prefetch of the next slab.  Epilogue: accumulators -> LDS ->
// (2x2 max) + bias + ReLU -> split -> two 16-byte stores per 8 channels.
//
// What bounds it (phase timers below, rocprofv3 SQ counters in profiles/): the vector-memory pipe.  A
// 128x128 tile pulls 64 KB through the CU's L1/TA per 64-deep stage - 3900 cycles per stage with the
// matrix pipe idle - against 1536 cycles of MFMA; the 256-row tile halves the weight bytes per MFMA.
#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

__device__ unsigned long long mm_dbg[8];  // phase-timing accumulators of the instrumented variant
__device__ u32x4 mm_zero_page[16];        // zero-initialised 256 B: source of out-of-image (padding) taps

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

// TIMED: phase timing instrumentation (shader clock), accumulated per wave into mm_dbg[]:
//   [0] wait for the prefetched loads + ds_write, [1] barrier 1, [2] issue of the next stage's loads,
//   [3] ds_read + MFMA phase, [4] barrier 2, [5] number of (wave, stage) samples
template <int BM, int BN, bool POOL, bool TIMED>
__global__ __launch_bounds__(2 * BM, 2) void conv3x3_hl16_kernel(
    const u32x4* __restrict__ in, const u32x4* __restrict__ wp, const float* __restrict__ bias,
    u32x4* __restrict__ out, int L, int H, int W, int Cin, int Cout, int Mtot, int ntm, int ntn, float oscale) {
  constexpr int NT = 2 * BM;               // threads: 8 per staged row, BM/4 rows per pass, 4 passes for A
  constexpr int NWAVES = NT / 64;
  constexpr int WN = (BN == 128) ? 2 : 1;  // waves along channels
  constexpr int WM = NWAVES / WN;          // waves along pixels
  constexpr int TM = BM / (WM * 32);
  constexpr int TN = BN / (WN * 32);
  constexpr int RPP = NT / 8;              // rows staged per pass
  constexpr int BLD = BN / RPP;            // weight passes
  constexpr int PLANE_A = BM * HL_LDT;     // halves
  constexpr int PLANE_B = BN * HL_LDT;
  static_assert(BM % RPP == 0 && BM / RPP == 4 && BN % RPP == 0, "staging geometry");

  // lo planes start 64 B after a multiple of 128 B: a staging wave writes hi and lo pieces of the same
  // unit in one ds_write_b128 (lane parity selects the plane), and the skew keeps them on different banks
  constexpr int SKEW = 32;  // halves
  __shared__ __attribute__((aligned(128))) _Float16 smem[2 * PLANE_A + 2 * PLANE_B + 4 * SKEW];
  _Float16* As_hi = smem;
  _Float16* As_lo = smem + PLANE_A + SKEW;
  _Float16* Bs_hi = smem + 2 * PLANE_A + 2 * SKEW;
  _Float16* Bs_lo = smem + 2 * PLANE_A + PLANE_B + 3 * SKEW;

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
  if (clen % ntn == 0 && cbase % ntn == 0) {
    // the chunk [cbase, cbase+clen) covers whole pixel tiles: walk it channel-tile-major
    const int mcount = clen / ntn;
    const int s = lid - cbase;
    nt = s / mcount;
    mt = cbase / ntn + s % mcount;
  } else {  // plain pixel-major order (still correct, just less L2 friendly)
    mt = lid / ntn;
    nt = lid % ntn;
  }
  const int n0 = nt * BN;

  const int Hq = H >> 1, Wq = W >> 1;
  const int lrow = tid >> 3;  // 0..RPP-1 (+RPP i)
  const int ku = tid & 7;     // 16-byte piece inside a row's 128-byte half-slab (see load_stage)
  const int cin8 = Cin >> 3;  // units per input pixel
  _Float16* A_pl = (ku & 1) ? As_lo : As_hi;  // plane this thread stages into (piece parity)
  _Float16* B_pl = (ku & 1) ? Bs_lo : Bs_hi;

  int py[4], px[4];
  long pbase[4];
  bool pval[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = mt * BM + lrow + RPP * i;
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
  const int cpt = Cin / HL_BK;  // 64-channel slabs

  // K order: channel-slab major, tap minor.  The 9 taps of one 64-channel slab re-read the same
  // (haloed) pixel neighbourhood back to back, so the re-reads hit L1/L2 instead of being 9 separate
  // sweeps over the activation (tap-major order had every tap miss the 4 MiB L2: profiles/README.md).
  // A row's 64-channel slab is 256 contiguous bytes = 16 pieces of 16 B: [hi u0][lo u0][hi u1]...
  // Thread j = tid&7 of a row takes pieces j and j+8, so each load INSTRUCTION covers a row's 128
  // contiguous bytes with 8 lanes (whole cache lines).  Piece parity = plane (hi / lo), piece>>1 = unit.
  auto load_stage = [&](int it) {
    const int slab = it / 9;
    const int tap = it - slab * 9;
    const int q0 = slab * 16 + ku;  // first piece (u32x4 index inside the pixel / weight row)
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // branch-free and select-free: out-of-image taps read a zero page, so the loaded registers are
      // consumed only by the ds_write of the NEXT iteration (the wait sits there).
      // (A conditional load made hipcc keep ra[] in scratch; a select made it wait for the data here.)
      const int yy = py[i] + dy, xx = px[i] + dx;
      const bool ok = pval[i] && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
      const u32x4* p = ok ? in + (pbase[i] + dy * W + dx) * (cin8 * 2) + q0 : mm_zero_page;
      ra[i][0] = p[0];
      ra[i][1] = p[8];
    }
#pragma unroll
    for (int i = 0; i < BLD; ++i) {
      const u32x4* p = wp + ((long)tap * Cout + n0 + lrow + RPP * i) * (cin8 * 2) + q0;
      rb[i][0] = p[0];
      rb[i][1] = p[8];
    }
  };

  unsigned long long tph[5] = {0, 0, 0, 0, 0};
  unsigned long long tprev = 0;
  auto tick = [&](int ph) {
    if constexpr (TIMED) {
      const unsigned long long now = __builtin_readcyclecounter();
      tph[ph] += now - tprev;
      tprev = now;
    }
  };

  const int nk = 9 * cpt;
  const int lr = lane & 31;
  const int kh = (lane >> 5) * 8;
  load_stage(0);
  if constexpr (TIMED) tprev = __builtin_readcyclecounter();
  for (int it = 0; it < nk; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<u32x4*>(&A_pl[(lrow + RPP * i) * HL_LDT + (ku >> 1) * 8]) = ra[i][0];
      *reinterpret_cast<u32x4*>(&A_pl[(lrow + RPP * i) * HL_LDT + (4 + (ku >> 1)) * 8]) = ra[i][1];
    }
#pragma unroll
    for (int i = 0; i < BLD; ++i) {
      *reinterpret_cast<u32x4*>(&B_pl[(lrow + RPP * i) * HL_LDT + (ku >> 1) * 8]) = rb[i][0];
      *reinterpret_cast<u32x4*>(&B_pl[(lrow + RPP * i) * HL_LDT + (4 + (ku >> 1)) * 8]) = rb[i][1];
    }
    if constexpr (TIMED) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    tick(0);
    __syncthreads();
    tick(1);
    if (it + 1 < nk) load_stage(it + 1);
    tick(2);
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
    tick(3);
    __syncthreads();
    tick(4);
  }
  if constexpr (TIMED) {
    if (lane == 0) {
      for (int i = 0; i < 5; ++i) atomicAdd(&mm_dbg[i], tph[i]);
      atomicAdd(&mm_dbg[5], (unsigned long long)nk);
    }
  }

  // ---- epilogue: accumulators -> LDS (fp32 [128][BN+4], 128 tile rows at a time) ->
  //      pool / bias / relu / split -> 32-byte hl16 units ----
  constexpr int CLD = BN + 4;
  static_assert(128 * CLD * 4 <= (int)sizeof(smem), "epilogue staging must fit the K-loop LDS");
  float* Cs = reinterpret_cast<float*>(smem);
  constexpr int UN = BN / 8;  // 8-channel units per tile row
  const int cout8 = Cout >> 3;
#pragma unroll
  for (int half = 0; half < BM / 128; ++half) {
    if (half > 0) __syncthreads();  // previous half fully consumed
    if ((wm * TM * 32) / 128 == half) {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int e = 0; e < 16; ++e)
            Cs[(wm * TM * 32 - half * 128 + tm * 32 + mm_acc_row(e, lane)) * CLD + wn * TN * 32 + tn * 32 + lr] =
                acc[tm][tn][e];
    }
    __syncthreads();
    const int mbase = mt * BM + half * 128;
    if constexpr (POOL) {
      for (int w = tid; w < 32 * UN; w += NT) {
        const int qd = w / UN, u = w - qd * UN;
        const int m = mbase + qd * 4;
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
      for (int w = tid; w < 128 * UN; w += NT) {
        const int r = w / UN, u = w - r * UN;
        const int m = mbase + r;
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
}

extern "C" int mmmot_debug_read_phase_timers(unsigned long long* out8, int reset) {
  hipError_t e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(mm_dbg), 8 * sizeof(unsigned long long));
  if (e != hipSuccess) return (int)e;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    e = hipMemcpyToSymbol(HIP_SYMBOL(mm_dbg), z, sizeof(z));
  }
  return mm_check(e);
}

// Tuning knob (tools/bench_conv_variants.py): 0 / 1 = 128-row tiles (default), 3 = the same with
// phase-timing instrumentation, 4 = 256-row tiles.  Results are identical.  (A producer/consumer-wave
// variant of this register-staged kernel was measured too and was not faster: the LDS-DMA kernel in
// conv3x3_hl16_dma.hip supersedes it; numbers in profiles/README.md.)
static int g_hl16_variant = 0;
extern "C" int mmmot_set_conv_variant(int v) {
  if (v < 0 || v > 4) return MMMOT_EINVAL;
  g_hl16_variant = v;
  return MMMOT_OK;
}

template <int BM, int BN, bool POOL, bool TIMED>
static int launch_hl_v(const void* in, const void* wp, const float* bias, void* out, int L, int H, int W, int Cin,
                       int Cout, float oscale, hipStream_t s) {
  const int Mtot = L * H * W;
  const int ntm = (Mtot + BM - 1) / BM;
  const int ntn = Cout / BN;
  hipLaunchKernelGGL((conv3x3_hl16_kernel<BM, BN, POOL, TIMED>), dim3(ntm * ntn), dim3(2 * BM), 0, s,
                     (const u32x4*)in, (const u32x4*)wp, bias, (u32x4*)out, L, H, W, Cin, Cout, Mtot, ntm, ntn,
                     oscale);
  return mm_check(hipGetLastError());
}

template <int BN, bool POOL>
static int launch_hl(const void* in, const void* wp, const float* bias, void* out, int L, int H, int W, int Cin,
                     int Cout, float oscale, hipStream_t s) {
  // 256-row tiles (8 waves, one workgroup per CU) halve the weight bytes per MFMA; they need enough
  // pixel tiles to fill 256 CUs, otherwise the 128-row tile (two workgroups per CU) balances better.
  switch (g_hl16_variant) {
    case 1: return launch_hl_v<128, BN, POOL, false>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
    case 3: return launch_hl_v<128, BN, POOL, true>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
    case 4: return launch_hl_v<256, BN, POOL, false>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
    default: return launch_hl_v<128, BN, POOL, false>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
  }
}

extern "C" int mmmot_conv3x3_bn_relu_hl16(const void* in, const void* wp, const float* bias, void* out, int L,
                                          int H, int W, int Cin, int Cout, int pool, float oscale, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!in || !wp || !bias || !out || L <= 0 || H <= 0 || W <= 0) return MMMOT_EINVAL;
  if ((H & 1) || (W & 1) || Cin % HL_BK != 0 || Cout % 64 != 0) return MMMOT_EINVAL;
  if (!mm_al16(in) || !mm_al16(wp) || !mm_al16(out)) return MMMOT_EINVAL;
  if ((long)L * H * W >= (1L << 31) - 256) return MMMOT_EINVAL;
  if (Cout % 128 == 0)
    return pool ? launch_hl<128, true>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s)
                : launch_hl<128, false>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
  return pool ? launch_hl<64, true>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s)
              : launch_hl<64, false>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
}

// ---------------------------------------------------------------------------
// fp32 rows <-> hl16 rows (tests, tools; C % 8 == 0).
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
