// A-resident row GEMM for the PointNet layers whose output is only ever reduced
// (see include/mmmot_hip.h: mmmot_gemm_ares).
//
//   v[r][n] = oscale * sum_k relu(X[r][k]*sc[g][k] + sh[g][k]) * W[n][k] + bias[n] + dbias[tile_dbrow[t]][n]
//
// and, instead of storing v ([points][1024] fp32 = 1 GiB per cfg3 frame pair for PointNet conv5),
//   part   [2t+h][0/1][n] = sum / centred M2 of v over the valid rows of 64-row half-tile h   (statistics pass)
//   colsum [2t+h][n]      = sum over the same rows of relu(v*osc[g][n] + osh[g][n])           (consumer pass)
//
// Why a separate kernel: K is 64 or 128 and N is 512 or 1024, so in the generic 128x128-tile kernel
// (gemm_rows.hip) every workgroup stages and splits the same activation rows once per channel tile and runs
// 2 K-stages between a prologue and a three-barrier statistics epilogue - 70-200 TFLOP/s-equivalent.  Here a
// workgroup keeps its 128 activation rows resident in LDS (normalised, ReLU'd and hi/lo split ONCE) and walks
// all N/128 channel tiles: weights stream through a double-buffered LDS stage (one barrier per 64-channel
// stage), the per-tile epilogue is register-only (each wave owns 64 rows x 32 channels: column sums by
// lane-half shuffle, statistics centred on the wave's own 64-row mean), nothing but the partials is stored.
// Arithmetic: fp16 matrix cores with the 3-term hi/lo split (same as gemm_rows F16 / conv3x3_hl16).
//
// Round 3: 2 x 2 register blocking.  With a 64-row x 32-channel wave tile every MFMA needed one 1 KB LDS fragment read
// (4 activation + 2 weight fragments per 6 MFMAs): 192 KB of fragment reads per 1536-cycle stage and CU - the kernel sat
// at 34 % of the ceiling, LDS-read bound (a weight-resident variant, a deferred epilogue and counted-vmcnt role copies
// all left it there).  Now a wave owns 64 rows x 64 channels (channel tile 256, weight stage 256 channels x 32 k = the
// same 32 KB): 4 + 4 fragment reads feed 12 MFMAs.
#include <cstdlib>
#include <type_traits>

#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define AR_BM 128
#define AR_BN 256   // channels per channel tile: a wave owns 64 rows x 64 channels (2 x 2 MFMA blocks)
#define AR_BK 32    // k per weight stage
#define AR_LDT 72   // halves per LDS row of the activation planes (64 k + pad: 144 B, conflict-free ds_read_b128)
#define AR_LDB 40   // halves per LDS row of a weight stage (32 k + pad: 80 B - 16 consecutive rows hit 16 distinct
                    // 16-byte bank groups)
#define AR_THREADS 512

static __device__ float ar_zeros[4096];  // stands in for absent bias / dbias / osc / osh rows (branch-free loads)

// KS = K / 64; MODE bit 0: statistics (part), bit 1: normalise + ReLU + column sums (colsum).
// Every global load of the K loop is unconditional (indices clamped at the end): with a fixed number of
// loads in flight the compiler can wait with counted vmcnt(N) instead of draining the two weight stages
// that are meant to stay in flight.
template <int KS, int MODE>
__global__ __launch_bounds__(AR_THREADS) void gemm_ares_kernel(mmmot_gemm_ares_args a) {
  constexpr int PLANE = AR_BM * AR_LDT;  // halves per [128][72] plane
  constexpr int BPLANE = AR_BN * AR_LDB; // halves per [256][40] weight-stage plane
  constexpr int SPT = 2 * KS;            // 32-k weight stages per channel tile
  __shared__ __attribute__((aligned(16))) _Float16 As[KS][2][PLANE];  // [64-k block][hi, lo]
  __shared__ __attribute__((aligned(16))) _Float16 Bs[2][2][BPLANE];  // [buffer][hi, lo]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;  // 2 x 4 waves: 64 rows x 64 channels each
  const int lr = lane & 31;
  const int kh = (lane >> 5) * 8;

  const int ntn = a.N / AR_BN;
  const int NS = ntn * SPT;  // weight stages per row tile (always even)
  // Persistent workgroups (LDS allows one per CU): tiles blockIdx.x, + gridDim.x, ...  The weight stream
  // wraps around (stage NS continues with stage 0 of the next tile), the next tile's activation rows are
  // fetched into registers during the current tile's last stages, so a tile switch costs one conversion
  // pass + one barrier instead of a cold prologue (measured fixed cost before: 8 us of a 31 us tile).
  struct TileMeta {
    int row0, nrows, grp, dbrow;
  };
  auto meta_of = [&](int tt) {
    TileMeta m;
    m.row0 = a.tile_row0[tt];
    m.nrows = a.tile_nrows[tt];
    // unconditional loads (an absent table reads tile_row0 instead, discarded in scalar()): a conditional load makes
    // the number of loads in flight path-dependent and turns later counted waits into vmcnt(0)
    m.grp = (a.tile_group ? a.tile_group : a.tile_row0)[tt];
    m.dbrow = (a.dbias ? a.tile_dbrow : a.tile_row0)[tt];
    return m;
  };
  // the table words are wave-uniform: held as scalars, and those of tile t + 2 * gridDim.x are requested a whole tile
  // before they are needed (a lookup at the tile switch is two dependent L2 round trips in front of the switch, and
  // its vmcnt(0) also drains the two weight stages that are meant to stay in flight)
  auto scalar = [&](const TileMeta& v) {
    TileMeta m;
    m.row0 = __builtin_amdgcn_readfirstlane(v.row0);
    m.nrows = __builtin_amdgcn_readfirstlane(v.nrows);
    m.grp = a.tile_group ? __builtin_amdgcn_readfirstlane(v.grp) : 0;
    m.dbrow = a.dbias ? __builtin_amdgcn_readfirstlane(v.dbrow) : 0;
    return m;
  };
  int t = blockIdx.x;
  if (t >= a.T) return;
  const int gstep = gridDim.x;
  TileMeta cur = scalar(meta_of(t));

  // ---- wave roles.  vmcnt retires in order PER WAVE: activation rows (HBM, a CU completes only ~30 line misses
  // per microsecond) requested by a wave that also stages weights sit in front of every weight-stage wait, and
  // requested late they arrive late (0.26 ms of 0.99 at N = 1024 was tile-switch time).  So waves 0-3 stream the
  // weights (L2 hits) and waves 4-7 fetch the NEXT tile's activation rows a whole tile ahead, each group with its
  // own queue.  Both views share one set of staging registers (a wave uses one of them).
  const bool wspec = wave < 4;
  u32x4 stg[16];
  // weights: thread tw -> unit q = tw & 3 (of the 4 hl16 units [hi8 | lo8] of a 32-k stage) of channel rows
  // (tw >> 2) + 64 p, p = 0..3: a wave's load instruction covers 16 rows x 64 contiguous bytes
  const int wrow = (tid & 255) >> 2, wq = tid & 3;
  const u32x4* wp = reinterpret_cast<const u32x4*>(a.W);
  const long ku = (long)(a.K >> 3);  // hl16 units per weight row
  auto load_w = [&](int s, auto SLOT) {  // two stages in flight: slot = stage & 1
    constexpr int sl = decltype(SLOT)::value;
    const int nt = s / SPT, ks = s - nt * SPT;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const u32x4* p = wp + ((long)(nt * AR_BN + wrow + 64 * r) * ku + ks * 4 + wq) * 2;
      stg[sl * 8 + 2 * r + 0] = p[0];
      stg[sl * 8 + 2 * r + 1] = p[1];
    }
  };
  auto store_w = [&](int buf, auto SLOT) {
    constexpr int sl = decltype(SLOT)::value;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      *reinterpret_cast<u32x4*>(&Bs[buf][0][(wrow + 64 * r) * AR_LDB + wq * 8]) = stg[sl * 8 + 2 * r + 0];
      *reinterpret_cast<u32x4*>(&Bs[buf][1][(wrow + 64 * r) * AR_LDB + wq * 8]) = stg[sl * 8 + 2 * r + 1];
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  // ---- activation rows (waves 4-7): raw rows -> registers (load_x), normalise + ReLU + hi/lo split -> LDS (stage_a)
  // thread tx -> (rows tx >> 2 and (tx >> 2) + 64, quarter q of the K axis: 16 * KS consecutive channels)
  const int ar = (tid & 255) >> 2, aq = tid & 3;
  auto load_x = [&](const TileMeta& m) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int row = ar + 64 * r;
      const float* px = a.X + (long)(m.row0 + (row < m.nrows ? row : 0)) * a.ldx + aq * (16 * KS);
#pragma unroll
      for (int u = 0; u < 2 * KS; ++u) {
        stg[r * 4 * KS + 2 * u] = *reinterpret_cast<const u32x4*>(px + 8 * u);
        stg[r * 4 * KS + 2 * u + 1] = *reinterpret_cast<const u32x4*>(px + 8 * u + 4);
      }
    }
  };
  auto stage_a = [&](const TileMeta& m) {
    const float* psc = a.sc + (long)m.grp * a.ldsc + aq * (16 * KS);
    const float* psh = a.sh + (long)m.grp * a.ldsc + aq * (16 * KS);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int row = ar + 64 * r;
      const bool rv = row < m.nrows;
#pragma unroll
      for (int u = 0; u < 2 * KS; ++u) {  // 8-channel units of this thread
        const f32x4 x0 = __builtin_bit_cast(f32x4, stg[r * 4 * KS + 2 * u]);
        const f32x4 x1 = __builtin_bit_cast(f32x4, stg[r * 4 * KS + 2 * u + 1]);
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(psc + 8 * u), s1 = *reinterpret_cast<const f32x4*>(psc + 8 * u + 4);
        const f32x4 h0 = *reinterpret_cast<const f32x4*>(psh + 8 * u), h1 = *reinterpret_cast<const f32x4*>(psh + 8 * u + 4);
        u32x4 hi, lo;
        float y[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          y[e] = fminf(fmaxf(fmaf(x0[e], s0[e], h0[e]), 0.f), 65000.f);
          y[4 + e] = fminf(fmaxf(fmaf(x1[e], s1[e], h1[e]), 0.f), 65000.f);
          if (!rv) y[e] = y[4 + e] = 0.f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {  // lo = (f16)(y - (float)hi) as one v_fma_mix per value (common.h)
          unsigned h2, l2;
          mm_split2(y[2 * e], y[2 * e + 1], h2, l2);
          hi[e] = h2;
          lo[e] = l2;
        }
        const int k = aq * (16 * KS) + 8 * u;  // channel of the unit
        const int ks = k / 64, kk = k - ks * 64;  // 64-k block of the activation planes
        *reinterpret_cast<u32x4*>(&As[ks][0][row * AR_LDT + kk]) = hi;
        *reinterpret_cast<u32x4*>(&As[ks][1][row * AR_LDT + kk]) = lo;
      }
    }
  };
  if (!wspec) {
    load_x(cur);
    stage_a(cur);
  }
  // Everything from here on is compiled TWICE, once per wave role (WS: weight-streaming waves 0-3 / row-fetching waves
  // 4-7), and a wave enters its own copy: with the role a compile-time constant every copy has a FIXED number of global
  // loads in flight, so the waits in front of the per-channel constants are counted (vmcnt(16) in the weight waves)
  // instead of vmcnt(0) - which drained both weight stages in flight at the end of every channel tile.  Both copies
  // execute the same sequence of barriers.
  auto run = [&](auto WSC) {
    constexpr bool WS = decltype(WSC)::value;
    // weight stages: s -> register slot s & 1; two stages of global loads are always in flight (one LDS
    // stage of latency is not enough: the L2 round trip under load is longer than a 24-MFMA stage)
    auto prime_w = [&]() {
      if constexpr (WS) {
        load_w(0, S0{});
        load_w(1 % NS, S1{});
        store_w(0, S0{});
        load_w(2 % NS, S0{});
      }
      __syncthreads();
    };
    prime_w();

    f32x16 acc[2][2];  // [row block][channel block]
  #pragma unroll
    for (int tm = 0; tm < 2; ++tm)
  #pragma unroll
      for (int tn = 0; tn < 2; ++tn)
  #pragma unroll
        for (int e = 0; e < 16; ++e) acc[tm][tn][e] = 0.f;

    // per-tile epilogue state (set by begin_tile)
    int nrows = 0, nsub = 0, grp = 0;
    int lim = 0;  // accumulator element (tm, e) is a valid row <=> tm * 32 + (e & 3) + 8 * (e >> 2) < lim (immediates
                  // against one register instead of 32 hoisted row indices)
    float inv_nsub = 0.f;
    long prow = 0;
    bool full = false;
    const float* pbias = a.bias ? a.bias : ar_zeros;
    // per-channel epilogue constants of a channel tile (two 32-channel blocks per wave): combined bias, output scale /
    // shift.  They are loaded one channel tile ahead (for the last channel tile: those of the NEXT row tile) and BEFORE
    // the weight loads of that stage: vmcnt retires in order, so a load issued at epilogue time would also wait
    // for the two weight stages in flight.
    struct EpiC {
      float cb[2], os[2], oh[2];
    };
    auto epi_consts = [&](const TileMeta& m, int nt, EpiC& c) {
      const float* pdb = a.dbias ? a.dbias + (long)m.dbrow * a.lddb : ar_zeros;
      const float* posc = (MODE & 2) ? a.osc + (long)m.grp * a.ldosc : ar_zeros;
      const float* posh = (MODE & 2) ? a.osh + (long)m.grp * a.ldosc : ar_zeros;
  #pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        const int n = nt * AR_BN + wn * 64 + tn * 32 + lr;
        c.cb[tn] = pbias[n] + pdb[n];
        c.os[tn] = posc[n];
        c.oh[tn] = posh[n];
      }
    };
    EpiC ec, en;
    epi_consts(cur, 0, ec);
    en = ec;
    TileMeta nxt = cur, raw2 = cur;
    if (t + gstep < a.T) nxt = scalar(meta_of(t + gstep));
    auto begin_tile = [&](const TileMeta& m, int tt) {
      nrows = m.nrows;
      grp = m.grp;
      nsub = min(max(nrows - 64 * wm, 0), 64);  // valid rows of this wave's half tile
      inv_nsub = nsub > 0 ? 1.f / (float)nsub : 0.f;
      prow = (long)(2 * tt + wm);
      full = (nsub == 64);
      lim = nrows - (wm * 64 + 4 * (lane >> 5));
    };
    begin_tile(cur, t);

    // One stage = 32 k of one channel tile (KQ: which of the tile's SPT stages; its parity is the LDS buffer / register
    // slot parity because SPT is even).
    auto stage = [&](int s, auto KQC) {
      constexpr int kq = decltype(KQC)::value;
      constexpr int odd = kq & 1;
      constexpr int ks = kq >> 1;          // 64-k block of the activation planes
      constexpr int kk0 = (kq & 1) * 32;   // first k inside it
      const int nt = s / SPT;
      if constexpr (kq == 0) {
        const bool last_nt = nt + 1 >= ntn;  // then: first channel tile of the next row tile (same loads, other rows)
        TileMeta m = cur;
        if (last_nt) m = nxt;
        // (handing these constants from waves 0-3 to waves 4-7 through LDS, so that they do not queue behind the
        // activation rows, measured slower than loading them in every wave)
        epi_consts(m, last_nt ? 0 : nt + 1, en);
      }
      // stage s+1 (register slot !odd) -> the other LDS buffer (last read in stage s-1, every wave is past
      // that barrier); then its slot takes the loads of stage s+3
      if constexpr (WS) {
        store_w(1 - odd, std::integral_constant<int, 1 - odd>{});
        load_w((s + 3) % NS, std::integral_constant<int, 1 - odd>{});  // wraps into the next row tile
      }
      // Two 16-channel steps of 12 MFMAs (2 x 2 blocks x 3 terms) fed by 8 fragment reads; the fragments of the second
      // step are read while the MFMAs of the first issue, one read per MFMA (sched_barrier pins the order): the two waves
      // of a SIMD run in lockstep after every barrier, so an LDS round trip that is not covered by this wave's own MFMAs
      // is idle matrix-pipe time.
      const _Float16* ah = &As[ks][0][0];
      const _Float16* al = &As[ks][1][0];
      const _Float16* bh = &Bs[odd][0][0];
      const _Float16* bl = &Bs[odd][1][0];
      const int offa0 = (wm * 64 + lr) * AR_LDT + kk0 + kh, offa1 = offa0 + 32 * AR_LDT;
      const int offb0 = (wn * 64 + lr) * AR_LDB + kh, offb1 = offb0 + 32 * AR_LDB;
      struct Fr {
        f16x8 v[8];  // ah0, al0, ah1, al1, bh0, bl0, bh1, bl1
      };
      auto rd1 = [&](Fr& f, auto JC, auto RC) {
        constexpr int j = decltype(JC)::value, r = decltype(RC)::value;
        if constexpr (r == 0) f.v[0] = *reinterpret_cast<const f16x8*>(ah + offa0 + j * 16);
        if constexpr (r == 1) f.v[1] = *reinterpret_cast<const f16x8*>(al + offa0 + j * 16);
        if constexpr (r == 2) f.v[2] = *reinterpret_cast<const f16x8*>(ah + offa1 + j * 16);
        if constexpr (r == 3) f.v[3] = *reinterpret_cast<const f16x8*>(al + offa1 + j * 16);
        if constexpr (r == 4) f.v[4] = *reinterpret_cast<const f16x8*>(bh + offb0 + j * 16);
        if constexpr (r == 5) f.v[5] = *reinterpret_cast<const f16x8*>(bl + offb0 + j * 16);
        if constexpr (r == 6) f.v[6] = *reinterpret_cast<const f16x8*>(bh + offb1 + j * 16);
        if constexpr (r == 7) f.v[7] = *reinterpret_cast<const f16x8*>(bl + offb1 + j * 16);
      };
      auto mm1 = [&](const Fr& f, auto IC, auto FIRSTC) {  // term-major: consecutive MFMAs walk the four accumulators
        constexpr int i = decltype(IC)::value;
        constexpr int blk = i & 3, term = i >> 2;
        constexpr int tm = blk >> 1, tn = blk & 1;
        // the first MFMA of a channel tile into each accumulator starts from the constant 0: no 64 v_mov per tile and wave
        // to clear them (8 % of the tile's 96 MFMAs in issue time)
        if constexpr (term == 0 && decltype(FIRSTC)::value && kq == 0) {
          const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.v[2 * tm + 1], f.v[4 + 2 * tn], z, 0, 0, 0);
        } else
        if constexpr (term == 0) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.v[2 * tm + 1], f.v[4 + 2 * tn], acc[tm][tn], 0, 0, 0);
        if constexpr (term == 1) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.v[2 * tm], f.v[5 + 2 * tn], acc[tm][tn], 0, 0, 0);
        if constexpr (term == 2) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.v[2 * tm], f.v[4 + 2 * tn], acc[tm][tn], 0, 0, 0);
      };
      Fr f0, f1;
  #define AR_RD(F, J, R) rd1(F, std::integral_constant<int, J>{}, std::integral_constant<int, R>{});
      AR_RD(f0, 0, 0) AR_RD(f0, 0, 1) AR_RD(f0, 0, 2) AR_RD(f0, 0, 3) AR_RD(f0, 0, 4) AR_RD(f0, 0, 5) AR_RD(f0, 0, 6) AR_RD(f0, 0, 7)
      __builtin_amdgcn_sched_barrier(0);
  #define AR_PAIR(I)                                                   \
    mm1(f0, std::integral_constant<int, I>{}, std::true_type{});       \
    __builtin_amdgcn_sched_barrier(0);                                 \
    if constexpr (I < 8) {                                             \
      AR_RD(f1, 1, I)                                                  \
      __builtin_amdgcn_sched_barrier(0);                               \
    }
      AR_PAIR(0) AR_PAIR(1) AR_PAIR(2) AR_PAIR(3) AR_PAIR(4) AR_PAIR(5)
      AR_PAIR(6) AR_PAIR(7) AR_PAIR(8) AR_PAIR(9) AR_PAIR(10) AR_PAIR(11)
  #undef AR_PAIR
  #undef AR_RD
  #define AR_MM(I) mm1(f1, std::integral_constant<int, I>{}, std::false_type{});
      AR_MM(0) AR_MM(1) AR_MM(2) AR_MM(3) AR_MM(4) AR_MM(5) AR_MM(6) AR_MM(7) AR_MM(8) AR_MM(9) AR_MM(10) AR_MM(11)
  #undef AR_MM
      if constexpr (kq == SPT - 1) {
        // ---- channel tile nt finished: register-only epilogue for this wave's 64 rows x 2 x 32 channels ----
        // v = acc*oscale + cb is never formed for full half tiles: the sums are taken on the raw accumulators
        // and rescaled (S = oscale*sum(acc) + n*cb, M2 = oscale^2 * M2(acc), relu(v*os+oh) = relu(acc*(oscale*os)
        // + (cb*os+oh))); 3 VALU ops per value.
  #pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
          const int n = nt * AR_BN + wn * 64 + tn * 32 + lr;
          const float cb = ec.cb[tn], os = ec.os[tn], oh = ec.oh[tn];  // fetched one channel tile ahead
          if constexpr (MODE & 1) {
            float s1 = 0.f;
            if (full) {
              s1 = mm_sum32(acc[0][tn], acc[1][tn]);
            } else {
  #pragma unroll
              for (int tm = 0; tm < 2; ++tm)
  #pragma unroll
                for (int e = 0; e < 16; ++e)
                  if (tm * 32 + (e & 3) + 8 * (e >> 2) < lim) s1 += acc[tm][tn][e];
            }
            s1 = mm_xor32_sum(s1);
            const float mu = s1 * inv_nsub;  // mean of the raw accumulators over the half tile
            float s2 = 0.f;
            if (full) {
              s2 = mm_m2_32(acc[0][tn], acc[1][tn], mu);
            } else {
  #pragma unroll
              for (int tm = 0; tm < 2; ++tm)
  #pragma unroll
                for (int e = 0; e < 16; ++e) {
                  const float d = acc[tm][tn][e] - mu;
                  if (tm * 32 + (e & 3) + 8 * (e >> 2) < lim) s2 = fmaf(d, d, s2);
                }
            }
            s2 = mm_xor32_sum(s2);
            if (lane < 32) {
              a.part[(prow * 2 + 0) * a.N + n] = fmaf(s1, a.oscale, (float)nsub * cb);
              a.part[(prow * 2 + 1) * a.N + n] = s2 * a.oscale * a.oscale;
            }
          }
          if constexpr (MODE & 2) {
            const float m1 = a.oscale * os, m0 = fmaf(cb, os, oh);
            float s3 = 0.f;
            if (full) {
              s3 = mm_relu_sum32(acc[0][tn], acc[1][tn], m1, m0);
            } else {
  #pragma unroll
              for (int tm = 0; tm < 2; ++tm)
  #pragma unroll
                for (int e = 0; e < 16; ++e)
                  if (tm * 32 + (e & 3) + 8 * (e >> 2) < lim) s3 += fmaxf(fmaf(acc[tm][tn][e], m1, m0), 0.f);
            }
            s3 = mm_xor32_sum(s3);
            if (lane < 32) a.colsum[prow * a.N + n] = s3;
          }
        }
        ec = en;
      }
      __syncthreads();
    };
    for (;;) {
      const int tn = t + gstep;
      const bool more = tn < a.T;
      const bool more2 = tn + gstep < a.T;
      // the next tile's rows are requested at the first stage of this tile (waves 4-7: nothing else in their queue
      // but the small per-channel constants)
      // table words of the tile after the next: unconditional (clamped) so that the number of loads in flight stays fixed;
      // consumed a whole tile later
      raw2 = meta_of(more2 ? tn + gstep : tn < a.T ? tn : t);
      for (int s = 0; s < NS; s += SPT) {
        if constexpr (!WS)
          if (more && s == 0) load_x(nxt);
        stage(s, std::integral_constant<int, 0>{});
        stage(s + 1, std::integral_constant<int, 1>{});
        if constexpr (SPT == 4) {
          stage(s + 2, std::integral_constant<int, 2>{});
          stage(s + 3, std::integral_constant<int, 3>{});
        }
      }
      if (!more) break;
      // tile switch: every wave is past the barrier that ended the last stage, As is free
      if constexpr (!WS) stage_a(nxt);
      __syncthreads();  // NS is even: the LDS buffer parity of stage 0 repeats, the weight stream simply continues
      cur = nxt;
      if (more2) nxt = scalar(raw2);
      t = tn;
      begin_tile(cur, t);
    }
  };
  if (wspec) run(std::true_type{});
  else run(std::false_type{});
}

int mmmot_gemm_wres64_launch(const mmmot_gemm_ares_args* a, int mode, int n_cu, hipStream_t s);  // gemm_wres.hip
int mmmot_gemm_wreg128_try(const mmmot_gemm_ares_args* a, int mode, int n_cu, hipStream_t s, int* status);  // gemm_wreg.hip
int mmmot_ares_variant();                                                                                  // gemm_wreg.hip

extern "C" int mmmot_gemm_ares(const mmmot_gemm_ares_args* a, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!a || !a->X || !a->W || !a->sc || !a->sh || !a->tile_row0 || !a->tile_nrows || a->T <= 0) return MMMOT_EINVAL;
  if ((a->K != 64 && a->K != 128) || a->N <= 0 || a->N % 128 != 0) return MMMOT_EINVAL;
  if (a->ldx % 4 != 0 || a->ldsc % 4 != 0 || !mm_al16(a->X) || !mm_al16(a->W) || !mm_al16(a->sc) || !mm_al16(a->sh))
    return MMMOT_EINVAL;
  if (a->dbias && !a->tile_dbrow) return MMMOT_EINVAL;
  if (a->colsum && (!a->osc || !a->osh)) return MMMOT_EINVAL;
  if (!a->part && !a->colsum) return MMMOT_EINVAL;  // nothing to produce
  if (a->N > 4096) return MMMOT_EINVAL;
  const int mode = (a->part ? 1 : 0) | (a->colsum ? 2 : 0);
  const int n_cu = mm_num_cu();
  if (n_cu <= 0) return MMMOT_EINVAL;
  // K = 64 with the whole weight matrix in LDS (N <= 512): the weight-resident kernel; MMMOT_ARES_WRES=0 (read once)
  // keeps the streaming kernel for A/B timing (tools/bench_ares.py)
  static const bool use_wres = [] {
    const char* e = getenv("MMMOT_ARES_WRES");
    return !(e && e[0] == '0');
  }();
  const bool stream_only = mmmot_ares_variant() == 1 && a->N % AR_BN == 0;  // mmmot_set_gemm_ares_variant(1)
  if (use_wres && !stream_only && a->K == 64 && a->N <= 512) return mmmot_gemm_wres64_launch(a, mode, n_cu, s);
  // K = 128 consumer pass on a launch that fills the chip: the wave's weights live in registers (gemm_wreg.hip)
  int wst = MMMOT_OK;
  if (mmmot_gemm_wreg128_try(a, mode, n_cu, s, &wst)) return wst;
  if (a->N % AR_BN != 0) return MMMOT_EINVAL;  // the streaming kernel walks 256-channel tiles
  const int grid = a->T < n_cu ? a->T : n_cu;  // persistent: one workgroup per CU
#define AR_LAUNCH(KSV, MODEV) \
  hipLaunchKernelGGL((gemm_ares_kernel<KSV, MODEV>), dim3(grid), dim3(AR_THREADS), 0, s, *a)
  if (a->K == 128) {
    if (mode == 1) AR_LAUNCH(2, 1); else if (mode == 2) AR_LAUNCH(2, 2); else AR_LAUNCH(2, 3);
  } else {
    if (mode == 1) AR_LAUNCH(1, 1); else if (mode == 2) AR_LAUNCH(1, 2); else AR_LAUNCH(1, 3);
  }
#undef AR_LAUNCH
  return mm_check(hipGetLastError());
}
