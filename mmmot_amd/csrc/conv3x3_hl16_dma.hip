// hl16 trunk layer, LDS-DMA + wave-specialised variant ("dma").
//
// Same arithmetic and storage format as conv3x3_hl16.hip (3-term fp16 hi/lo split on
// v_mfma_f32_32x32x16_f16, fp32 accumulate, hl16 activations and weights).  Different machine mapping,
// driven by two measurements on MI355X (tools/l2bw_probe*.hip, phase timers of conv3x3_hl16.hip):
//   * one CU pulls only ~3.5 B/clk per loading wave from L2 (15 B/clk with 4 waves, 28-33 with 8,
//     43-48 with 16), VGPR loads and LDS-DMA alike;
//   * in the single-role kernel the vector-memory phase (64 KB per 128x128x64 stage) and the MFMA phase
//     strictly alternate: 45 % matrix-pipe utilisation.
// So: (1) 256 x BN tile - the weight tile is shared by twice the pixels, 48 KB instead of 64 KB per
// 128x128x64-equivalent; (2) 8 consumer waves (ds_read + MFMA only) + 8 producer waves (global_load_lds
// only: no VGPR staging, no ds_write, no address math in the loop beyond two adds); (3) a 3-slot LDS
// ring of 32-channel stages so two stages of loads are always in flight across the one barrier per
// stage (counted vmcnt, raw s_barrier); (4) unpadded lane-linear LDS rows (LDS-DMA writes base +
// lane*16) made conflict-free for ds_read_b128 by an XOR swizzle applied on the SOURCE side (which
// global piece a lane fetches) and on the fragment read.
#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

static __device__ u32x4 dma_zero_page[16];  // zero-initialised: source of out-of-image taps

#define D_BM 256
#define D_BK 32               // channels per stage: 4 units = 8 pieces of 16 B = 128 B per row
#define D_ROWB 128            // bytes per LDS row
#define D_NSLOT 3             // ring depth

__device__ __forceinline__ void dma_split8(f32x8 v, u32x4& hi, u32x4& lo) {
  f16x8 h, l;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float x = fminf(fmaxf(v[e], -65000.f), 65000.f);
    h[e] = (_Float16)x;
    l[e] = (_Float16)(x - (float)h[e]);
  }
  hi = __builtin_bit_cast(u32x4, h);
  lo = __builtin_bit_cast(u32x4, l);
}

// swizzle of LDS row r: slot s of the row holds piece s ^ swz(r); with two 128-byte rows per 256-byte
// bank row, (r & 1, (r >> 1) & 7 ^ piece) is distinct for the 16 rows of every ds_read_b128 lane group
__device__ __forceinline__ int dma_swz(int r) { return (r >> 1) & 7; }

// NLW: loading waves (8 = the 8 dedicated producer waves, 16 = consumers load too); ROT: per-workgroup
// rotation of the channel-slab order; ASKIP: timing experiment only (activation tile fetched for the centre
// tap only - WRONG results - to measure what an LDS-resident haloed patch would buy).
template <int BN, bool POOL, int NLW, bool ROT, bool ASKIP>
__global__ __launch_bounds__(1024, 4) void conv3x3_hl16_dma_kernel(
    const u32x4* __restrict__ in, const u32x4* __restrict__ wp, const float* __restrict__ bias,
    u32x4* __restrict__ out, int L, int H, int W, int Cin, int Cout, int Mtot, int ntm, int ntn, float oscale) {
  constexpr int WN = (BN == 128) ? 2 : 1;  // consumer waves along channels
  constexpr int WM = 8 / WN;               // consumer waves along pixels
  constexpr int TM = D_BM / (WM * 32);
  constexpr int TN = BN / (WN * 32);
  constexpr int A_BYTES = D_BM * D_ROWB;   // 32 KB
  constexpr int B_BYTES = BN * D_ROWB;     // 16 / 8 KB
  constexpr int SLOT = A_BYTES + B_BYTES;
  __shared__ __attribute__((aligned(256))) unsigned char smem[D_NSLOT * SLOT];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const bool producer = wave >= 8;

  // XCD-aware, channel-tile-major order (see conv3x3_hl16.hip)
  const int nwg = gridDim.x;
  const int lid = mm_xcd_remap(blockIdx.x, nwg);
  const int xq = nwg >> 3, xr = nwg & 7;
  const int xcd = blockIdx.x & 7;
  const int cbase = (xcd < xr) ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq;
  const int clen = (xcd < xr) ? xq + 1 : xq;
  int mt, nt;
  if (clen % ntn == 0 && cbase % ntn == 0) {
    const int mcount = clen / ntn;
    const int s = lid - cbase;
    nt = s / mcount;
    mt = cbase / ntn + s % mcount;
  } else {
    mt = lid / ntn;
    nt = lid % ntn;
  }
  const int n0 = nt * BN;
  const int Hq = H >> 1, Wq = W >> 1;
  const int cin8 = Cin >> 3;
  const int nk = 9 * (Cin / D_BK);  // stages: channel-slab major, tap minor

  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[tm][tn][e] = 0.f;
  const int cw = wave & 7;
  const int wm = cw / WN, wn = cw % WN;
  const int lr = lane & 31;

  // ---------------- every wave loads: LDS-DMA, two stages ahead ------------------------------------
  // The L2 -> CU path delivers ~3.5 B/clk per loading wave (tools/l2bw_probe*.hip), so all 16 waves of
  // the workgroup issue their share of each stage (2 activation + 1 weight instruction of 8 rows each);
  // the 8 consumer waves additionally run the MFMAs, the other 8 exist only to widen the load path.
  constexpr int AW = D_BM / 8 / NLW;  // activation DMA instructions per loading wave per stage: 2 / 4
  constexpr int BTOT = BN / 8;        // weight DMA instructions per stage: 16 / 8
  constexpr int BW = (BTOT + NLW - 1) / NLW;  // weight instructions per loading wave: 1 / 2
  const bool loader = (NLW == 16) || producer;
  const int lw = (NLW == 16) ? wave : (wave - 8);  // loader index
  const int rsub = lane >> 3;  // row inside an 8-row DMA instruction
  const int slot = lane & 7;   // 16-byte slot inside the 128-byte LDS row
  const u32x4* arow[AW];
  unsigned okmask[AW];
#pragma unroll
  for (int a = 0; a < AW; ++a) {
    const int r = ((lw & (NLW - 1)) * AW + a) * 8 + rsub;  // tile row
    const int m = mt * D_BM + r;
    const bool pv = m < Mtot;
    const int q = m >> 2, sub = m & 3;
    const int crop = q / (Hq * Wq);
    const int rem = q - crop * (Hq * Wq);
    const int yq = rem / Wq, xqq = rem - yq * Wq;
    const int y = 2 * yq + (sub >> 1), x = 2 * xqq + (sub & 1);
    unsigned mk = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
      if (pv && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) mk |= 1u << t;
    }
    okmask[a] = mk;
    arow[a] = in + (pv ? (((long)crop * H + y) * W + x) : 0L) * (cin8 * 2) + (slot ^ dma_swz(r));
  }
  const bool has_b = loader && (lw * BW < BTOT);  // wave-uniform
  const u32x4* brow[BW];
#pragma unroll
  for (int b = 0; b < BW; ++b) {
    const int brw = (((lw & (NLW - 1)) * BW + b) % BTOT) * 8 + rsub;  // weight row inside the tile
    brow[b] = wp + (long)(n0 + brw) * (cin8 * 2) + (slot ^ dma_swz(brw));
  }
  const long tapstride_w = (long)Cout * cin8 * 2;
  const u32x4* zsrc = dma_zero_page + slot;
  // Workgroups start their K walk at different channel slabs: with a power-of-two pixel stride (Cin*4 B)
  // every workgroup of the chip would otherwise read the same 128-byte offset of every 1-KiB row at the
  // same time and camp on a subset of the L2 channels (-30 % load bandwidth in tools/l2bw_probe.hip).
  const int nslab = Cin / D_BK;
  const int slab0 = ROT ? (mt * 3 + nt) % nslab : 0;

  auto issue_stage = [&](int it) {
    int slab = it / 9;
    const int tap = it - slab * 9;
    slab += slab0;
    if (slab >= nslab) slab -= nslab;
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
    const long aoff = (long)(dy * W + dx) * (cin8 * 2) + slab * 8;  // wave-uniform (8 pieces per slab)
    const long boff = (long)tap * tapstride_w + slab * 8;
    unsigned char* sb = smem + (it % D_NSLOT) * SLOT;
    if (!loader) return;
    if (!ASKIP || tap == 4) {
#pragma unroll
      for (int a = 0; a < AW; ++a) {
        const u32x4* src = ((okmask[a] >> tap) & 1u) ? arow[a] + aoff : zsrc;
        unsigned char* dst = sb + ((lw * AW + a) * 8) * D_ROWB;  // wave-uniform: 8 rows = 1 KB
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      }
    }
    if (has_b) {
#pragma unroll
      for (int b = 0; b < BW; ++b) {
        unsigned char* dst = sb + A_BYTES + (((lw * BW + b) % BTOT) * 8) * D_ROWB;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(brow[b] + boff),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      }
    }
  };
  // wait until only the newest stage's loads of THIS wave are still in flight (counted vmcnt needs an
  // immediate: AW + BW, AW, or - in the ASKIP experiment - a full drain)
  auto wait_keep_one_stage = [&]() {
    if (!loader) return;
    if constexpr (ASKIP) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (has_b) {
      if constexpr (AW + BW == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if constexpr (AW + BW == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else if constexpr (AW + BW == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      if constexpr (AW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if constexpr (AW == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };

  issue_stage(0);
  if (nk > 1) {
    issue_stage(1);
    wait_keep_one_stage();  // stage 0 landed
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  const int h = lane >> 5;
  for (int it = 0; it < nk; ++it) {
    // ring slot (it+2)%3 was last read in iteration it-1: free since the barrier that ended it
    if (it + 2 < nk) issue_stage(it + 2);
    if (!producer) {
      const unsigned char* sb = smem + (it % D_NSLOT) * SLOT;
#pragma unroll
      for (int j = 0; j < D_BK / 16; ++j) {
        f16x8 ah[TM], al[TM], bh[TN], bl[TN];
        const int phi = 2 * (2 * j + h);  // hi piece of this lane's 8-channel unit; lo piece = phi + 1
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
          const int r = wm * TM * 32 + tm * 32 + lr;
          const unsigned char* rowp = sb + r * D_ROWB;
          const int sw = dma_swz(r);
          ah[tm] = *reinterpret_cast<const f16x8*>(rowp + ((phi ^ sw) << 4));
          al[tm] = *reinterpret_cast<const f16x8*>(rowp + (((phi + 1) ^ sw) << 4));
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          const int r = wn * TN * 32 + tn * 32 + lr;
          const unsigned char* rowp = sb + A_BYTES + r * D_ROWB;
          const int sw = dma_swz(r);
          bh[tn] = *reinterpret_cast<const f16x8*>(rowp + ((phi ^ sw) << 4));
          bl[tn] = *reinterpret_cast<const f16x8*>(rowp + (((phi + 1) ^ sw) << 4));
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
    }
    // stage it+1 landed (this wave's part), this stage's LDS reads done -> hand the slots over
    if (it + 2 < nk) wait_keep_one_stage();
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  // ---- epilogue (all 16 waves): accumulators -> LDS fp32 [256][BN+4] -> pool/bias/relu/split -> hl16 ----
  constexpr int CLD = BN + 4;
  static_assert(D_BM * CLD * 4 <= (int)sizeof(smem), "epilogue staging must fit the ring");
  float* Cs = reinterpret_cast<float*>(smem);
  constexpr int UN = BN / 8;
  const int cout8 = Cout >> 3;
  if (!producer) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          Cs[(wm * TM * 32 + tm * 32 + mm_acc_row(e, lane)) * CLD + wn * TN * 32 + tn * 32 + lr] = acc[tm][tn][e];
  }
  __syncthreads();
  const int mbase = mt * D_BM;
  if constexpr (POOL) {
    for (int w = tid; w < (D_BM / 4) * UN; w += 1024) {
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
        dma_split8(v, hi, lo);
        u32x4* o = out + ((long)(m >> 2) * cout8 + (n0 >> 3) + u) * 2;
        o[0] = hi;
        o[1] = lo;
      }
    }
  } else {
    for (int w = tid; w < D_BM * UN; w += 1024) {
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
        dma_split8(v, hi, lo);
        u32x4* o = out + (pix * cout8 + (n0 >> 3) + u) * 2;
        o[0] = hi;
        o[1] = lo;
      }
    }
  }
}

static int g_dma_variant = 0;  // 0: 8 producer waves; 1: all 16 waves load; 2: 8 + slab rotation; 3: ASKIP experiment
extern "C" int mmmot_set_dma_variant(int v) {
  if (v < 0 || v > 3) return MMMOT_EINVAL;
  g_dma_variant = v;
  return MMMOT_OK;
}

template <int BN, bool POOL, int NLW, bool ROT, bool ASKIP>
static int launch_dma_v(const void* in, const void* wp, const float* bias, void* out, int L, int H, int W, int Cin,
                      int Cout, float oscale, hipStream_t s) {
  const int Mtot = L * H * W;
  const int ntm = (Mtot + D_BM - 1) / D_BM;
  const int ntn = Cout / BN;
  hipLaunchKernelGGL((conv3x3_hl16_dma_kernel<BN, POOL, NLW, ROT, ASKIP>), dim3(ntm * ntn), dim3(1024), 0, s,
                     (const u32x4*)in, (const u32x4*)wp, bias, (u32x4*)out, L, H, W, Cin, Cout, Mtot, ntm, ntn,
                     oscale);
  return mm_check(hipGetLastError());
}

template <int BN, bool POOL>
static int launch_dma(const void* in, const void* wp, const float* bias, void* out, int L, int H, int W, int Cin,
                      int Cout, float oscale, hipStream_t s) {
  switch (g_dma_variant) {
    case 1: return launch_dma_v<BN, POOL, 16, false, false>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
    case 2: return launch_dma_v<BN, POOL, 8, true, false>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
    case 3: return launch_dma_v<BN, POOL, 8, false, true>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
    default: return launch_dma_v<BN, POOL, 8, false, false>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
  }
}

// Same contract as mmmot_conv3x3_bn_relu_hl16 (Cin % 32 == 0 suffices here).
extern "C" int mmmot_conv3x3_bn_relu_hl16_dma(const void* in, const void* wp, const float* bias, void* out, int L,
                                              int H, int W, int Cin, int Cout, int pool, float oscale,
                                              void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!in || !wp || !bias || !out || L <= 0 || H <= 0 || W <= 0) return MMMOT_EINVAL;
  if ((H & 1) || (W & 1) || Cin % D_BK != 0 || Cout % 64 != 0) return MMMOT_EINVAL;
  if (!mm_al16(in) || !mm_al16(wp) || !mm_al16(out)) return MMMOT_EINVAL;
  if ((long)L * H * W >= (1L << 31) - D_BM) return MMMOT_EINVAL;
  if (Cout % 128 == 0)
    return pool ? launch_dma<128, true>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s)
                : launch_dma<128, false>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
  return pool ? launch_dma<64, true>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s)
              : launch_dma<64, false>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
}
