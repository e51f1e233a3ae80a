// Weight-resident variant of the A-resident row GEMM (mmmot_gemm_ares, see gemm_ares.hip / include/mmmot_hip.h) for
// K = 64 and N <= 512: PointNet_v1.conv1 64 -> 512 over every LiDAR point (reference modules/point_net.py:30-38), run
// twice per forward (statistics pass, normalise + ReLU + per-detection-sum pass).  Same arguments, same results contract.
//
// Why: in gemm_ares every 32x32x16 MFMA of a wave is fed by one 1 KB LDS fragment read (4 activation + 2 weight fragments
// per 6 MFMAs) - at full matrix-core rate that alone is the whole 128 B/clk of the CU's LDS - plus a weight stage (global
// -> registers -> LDS) and a workgroup barrier every 24 MFMAs; the K = 64 launches sat at 27 % MFMA-busy.  With K = 64 and
// N = 512 the whole hl16 weight matrix is 128 KB, so here
//   * it lives in LDS for the lifetime of a persistent workgroup (one per CU; XOR-swizzled rows instead of padding so that
//     it fits next to a 32 KB activation staging area: exactly the 160 KB of a CU),
//   * a wave keeps the hi / lo fragments of its 64 activation rows in REGISTERS for the whole tile (64 VGPRs): the only
//     LDS reads of the steady state are 2 weight fragments per 6 MFMAs,
//   * there is no weight traffic, no per-stage barrier: two barriers per 128-row tile (around the staging of the next
//     tile's rows, which were prefetched into registers a whole tile earlier).
// Measured (2 M rows, N = 512): statistics pass 0.72 -> 0.44 ms, consumer pass 0.63 -> 0.53 ms; 52 % MFMA-busy at the
// 1.7 GHz the chip holds under this load (profiles/README.md r02 "weight-resident conv1").  Tried without gain: a
// fixed issue priority or an s_sleep skew for one wave of each SIMD, four accumulator chains instead of two.
// Arithmetic and summation order are those of gemm_ares (fp16 matrix cores, 3-term hi/lo split, fp32 accumulation).
#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define WR_THREADS 512
#define WR_NMAX 512
#define WR_BM 128

static __device__ float wr_zeros[WR_NMAX];  // stands in for absent bias / dbias / osc / osh rows (branch-free loads)

// [row][64 halves] planes without padding: 16-byte chunk c of row r lives at chunk c ^ ((r >> 1) & 7).  Sixteen lanes
// reading chunk c of sixteen consecutive rows then touch every bank exactly once (rows alternate between the two bank
// halves, the XOR spreads the eight even / odd rows over the eight chunks of a half).
__device__ __forceinline__ int wr_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3); }

// MODE bit 0: statistics (part), bit 1: normalise + ReLU + column sums (colsum).
template <int MODE>
__global__ __launch_bounds__(WR_THREADS) void gemm_wres64_kernel(mmmot_gemm_ares_args a) {
  __shared__ __attribute__((aligned(16))) _Float16 Wh[WR_NMAX * 64];
  __shared__ __attribute__((aligned(16))) _Float16 Wl[WR_NMAX * 64];
  __shared__ __attribute__((aligned(16))) _Float16 Ah[WR_BM * 64];
  __shared__ __attribute__((aligned(16))) _Float16 Al[WR_BM * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;  // 2 x 4 waves: 64 rows each, 32-channel blocks wn, wn + 4, ...
  const int lr = lane & 31, lh = lane >> 5;
  const int nbw = a.N >> 7;  // 32-channel blocks per wave: 1 .. 4

  // Tile tables: the four words of tile t + 2 * gridDim.x are requested during tile t and taken (as scalars) at the start
  // of tile t + 1, where the loads of tile t + 2's rows need them: a table lookup in front of those loads would be two
  // dependent L2 round trips with the matrix cores idle.
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
  TileMeta cur = scalar(meta_of(t)), nxt = cur, raw2 = cur;
  if (t + gstep < a.T) nxt = scalar(meta_of(t + gstep));

  // ---- the weights: hl16 rows of 8 units [hi8 | lo8] -> swizzled hi / lo planes, once ----
  {
    const u32x4* wp = reinterpret_cast<const u32x4*>(a.W);
    for (int idx = tid; idx < a.N * 8; idx += WR_THREADS) {
      const int n = idx >> 3, u = idx & 7;
      const u32x4 hi = wp[(long)idx * 2], lo = wp[(long)idx * 2 + 1];
      *reinterpret_cast<u32x4*>(&Wh[wr_off(n, u)]) = hi;
      *reinterpret_cast<u32x4*>(&Wl[wr_off(n, u)]) = lo;
    }
  }
  // ---- activation rows: thread -> (rows (tid >> 4) + 32 i, channels 4 (tid & 15) .. + 3); a wave's load instruction
  // covers four whole rows (1 KB contiguous)
  const int sr = tid >> 4, sch = tid & 15;
  f32x4 x[4];
  auto load_x = [&](const TileMeta& m) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = sr + 32 * i;
      x[i] = *reinterpret_cast<const f32x4*>(a.X + (long)(m.row0 + (r < m.nrows ? r : 0)) * a.ldx + sch * 4);
    }
  };
  // prologue scale / shift of this thread's four channels: reloaded only when the group (sample) changes - a load at
  // the tile switch would be an L2 round trip with every wave waiting
  f32x4 s4, h4;
  auto load_norm = [&](int grp) {
    s4 = *reinterpret_cast<const f32x4*>(a.sc + (long)grp * a.ldsc + sch * 4);
    h4 = *reinterpret_cast<const f32x4*>(a.sh + (long)grp * a.ldsc + sch * 4);
  };
  auto stage_a = [&](const TileMeta& m) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = sr + 32 * i;
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      u32x2 hi, lo;
      float y[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        y[e] = fminf(fmaxf(fmaf(x[i][e], s4[e], h4[e]), 0.f), 65000.f);
        if (r >= m.nrows) y[e] = 0.f;
      }
      unsigned h0_, l0_, h1_, l1_;
      mm_split2(y[0], y[1], h0_, l0_);  // lo = (f16)(y - (float)hi) as one v_fma_mix per value (common.h)
      mm_split2(y[2], y[3], h1_, l1_);
      hi = u32x2{h0_, h1_};
      lo = u32x2{l0_, l1_};
      const int off = wr_off(r, sch >> 1) + (sch & 1) * 4;
      *reinterpret_cast<u32x2*>(&Ah[off]) = hi;
      *reinterpret_cast<u32x2*>(&Al[off]) = lo;
    }
  };
  // ---- per-channel epilogue constants of this wave's blocks: bias (fixed), per-tile bias row, output scale / shift.
  // Those of the NEXT tile are requested BEFORE its rows (vmcnt retires in order: a constant load queued behind the
  // row prefetch would make the first epilogue that touches it wait for HBM) into their own registers, and no
  // arithmetic touches them until the tile switch, where they become cb (combined bias) and m1 / m0 (the consumer's
  // relu(acc * m1 + m0)).
  float pb[4], cb[4], m1[4], m0[4], dbn[4], osn[4], ohn[4];
  auto load_consts = [&](const TileMeta& m) {
    const float* pdb = a.dbias ? a.dbias + (long)m.dbrow * a.lddb : wr_zeros;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      dbn[i] = osn[i] = ohn[i] = 0.f;
      if (i < nbw) {
        const int n = (wn + 4 * i) * 32 + lr;
        dbn[i] = pdb[n];
        if constexpr (MODE & 2) {
          osn[i] = a.osc[(long)m.grp * a.ldosc + n];
          ohn[i] = a.osh[(long)m.grp * a.ldosc + n];
        }
      }
    }
  };
  auto take_consts = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      cb[i] = pb[i] + dbn[i];
      m1[i] = a.oscale * osn[i];
      m0[i] = fmaf(cb[i], osn[i], ohn[i]);
    }
  };
#pragma unroll
  for (int i = 0; i < 4; ++i) pb[i] = (a.bias && i < nbw) ? a.bias[(wn + 4 * i) * 32 + lr] : 0.f;
  load_consts(cur);
  load_x(cur);
  load_norm(cur.grp);
  take_consts();

  for (;;) {
    // ---- tile switch: the rows fetched a tile ago -> LDS -> this wave's fragment registers ----
    __syncthreads();  // every wave has read the previous tile's fragments (first pass: the weight planes are complete)
    stage_a(cur);
    __syncthreads();
    f16x8 af[4][4];  // [k16 step][hi rows 0-31, lo rows 0-31, hi rows 32-63, lo rows 32-63]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r0 = wm * 64 + lr, r1 = r0 + 32;
      af[j][0] = *reinterpret_cast<const f16x8*>(&Ah[wr_off(r0, 2 * j + lh)]);
      af[j][1] = *reinterpret_cast<const f16x8*>(&Al[wr_off(r0, 2 * j + lh)]);
      af[j][2] = *reinterpret_cast<const f16x8*>(&Ah[wr_off(r1, 2 * j + lh)]);
      af[j][3] = *reinterpret_cast<const f16x8*>(&Al[wr_off(r1, 2 * j + lh)]);
    }
    const int tn = t + gstep;
    const bool more = tn < a.T;
    const int nrows = cur.nrows;
    const int nsub = min(max(nrows - 64 * wm, 0), 64);  // valid rows of this wave's half tile
    const float inv_nsub = nsub > 0 ? 1.f / (float)nsub : 0.f;
    const long prow = (long)(2 * t + wm);
    const bool full = (nsub == 64);
    // row of accumulator element (tm, e) in the tile = wm * 64 + tm * 32 + mm_acc_row(e, lane); valid <=> the
    // compile-time part < lim (one register instead of 32 hoisted row indices)
    const int lim = nrows - (wm * 64 + 4 * lh);

    // weight fragments: step j + 1 (or step 0 of the next block) is read before the MFMAs of step j issue
    f16x8 bh[2], bl[2];
    auto read_b = [&](int blk, int j, int slot) {
      const int n = (wn + 4 * blk) * 32 + lr;
      bh[slot] = *reinterpret_cast<const f16x8*>(&Wh[wr_off(n, 2 * j + lh)]);
      bl[slot] = *reinterpret_cast<const f16x8*>(&Wl[wr_off(n, 2 * j + lh)]);
    };
    read_b(0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i >= nbw) break;
      const int n = (wn + 4 * i) * 32 + lr;
      f32x16 acc[2];
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[tm][e] = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int sl = j & 1;
        if (j < 3) read_b(i, j + 1, sl ^ 1);
        else if (i + 1 < nbw) read_b(i + 1, 0, sl ^ 1);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[j][1], bh[sl], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[j][3], bh[sl], acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[j][0], bl[sl], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[j][2], bl[sl], acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[j][0], bh[sl], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[j][2], bh[sl], acc[1], 0, 0, 0);
      }
      if (i == 0 && more) {
        // the next tile's requests (constants first, then its rows, then the table words of the tile after it) are
        // issued in the shadow of this block's MFMAs - ~25 VMEM instructions cost ~800 issue cycles (s_memtime) that
        // were part of the tile switch, with the matrix pipe idle - and stay in flight for the rest of the tile
        __builtin_amdgcn_sched_barrier(0);
        load_consts(nxt);
        load_x(nxt);
        if (tn + gstep < a.T) raw2 = meta_of(tn + gstep);
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- register-only epilogue of 64 rows x 32 channels (as gemm_ares: sums on the raw accumulators, rescaled) ----
      const float cbi = cb[i];
      if constexpr (MODE & 1) {
        float s1 = 0.f;
        if (full) {
          s1 = mm_sum32(acc[0], acc[1]);
        } else {
#pragma unroll
          for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int e = 0; e < 16; ++e)
              if (tm * 32 + (e & 3) + 8 * (e >> 2) < lim) s1 += acc[tm][e];
        }
        s1 = mm_xor32_sum(s1);
        const float mu = s1 * inv_nsub;  // mean of the raw accumulators over the half tile
        float s2 = 0.f;
        if (full) {
          s2 = mm_m2_32(acc[0], acc[1], mu);
        } else {
#pragma unroll
          for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const float d = acc[tm][e] - mu;
              if (tm * 32 + (e & 3) + 8 * (e >> 2) < lim) s2 = fmaf(d, d, s2);
            }
        }
        s2 = mm_xor32_sum(s2);
        if (lane < 32) {
          a.part[(prow * 2 + 0) * a.N + n] = fmaf(s1, a.oscale, (float)nsub * cbi);
          a.part[(prow * 2 + 1) * a.N + n] = s2 * a.oscale * a.oscale;
        }
      }
      if constexpr (MODE & 2) {
        const float m1i = m1[i], m0i = m0[i];
        float s3 = 0.f;
        if (full) {
          s3 = mm_relu_sum32(acc[0], acc[1], m1i, m0i);
        } else {
#pragma unroll
          for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int e = 0; e < 16; ++e)
              if (tm * 32 + (e & 3) + 8 * (e >> 2) < lim) s3 += fmaxf(fmaf(acc[tm][e], m1i, m0i), 0.f);
        }
        s3 = mm_xor32_sum(s3);
        if (lane < 32) a.colsum[prow * a.N + n] = s3;
      }
    }
    if (!more) break;
    take_consts();
    if (nxt.grp != cur.grp) load_norm(nxt.grp);
    cur = nxt;
    nxt = scalar(raw2);
    t = tn;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Column-sum pass with INDEPENDENT waves (round 6).  The two-barrier kernel above sat at 52 % MFMA-busy: eight waves meet
// twice per 128-row tile around a staging area, and a wave's register epilogue overlaps nothing of its own.  Here
//   * a wave owns a 64-row half tile for ALL of the layer's 32-channel blocks (up to 16): its A fragments are generated in
//     registers straight from global loads in fragment layout (lane = row, 32 B per 16-k step - the pattern of
//     gemm_wide's source rows), so the rows never pass through LDS and no other wave needs them;
//   * the only shared state is the weight planes, written once: after that barrier a wave never synchronises again, and
//     the two waves of a SIMD drift apart until one's epilogue / conversion VALU runs under the other's MFMAs;
//   * a wave walks a CONTIGUOUS run of half tiles, so the prologue scale / shift (a 512 B LDS slot per wave) and the
//     epilogue constants change once per detection, not once per item.  The epilogue table costs 16 registers: lanes
//     0-31 of tab[i] hold m1 of channel 32 i + lane, lanes 32-63 hold m0, one v_permlane32_swap per block hands both to
//     every lane;
//   * the next item's rows (64 registers) are requested right after the conversion of the current ones and land during
//     its 384 MFMAs; tile-table words come by s_load.
// Same MFMA order per accumulator, same per-lane summation order, same half-wave exchange as above and as gemm_ares:
// the column sums are bit-identical (tests/test_kernels_gpu.py).  Ragged half tiles (a detection's tail) take a compact
// loop with the constants read per block.
struct WiItem {
  int row0, nrows, grp, dbrow;  // of the item's 128-row tile
};

__global__ __launch_bounds__(WR_THREADS) void gemm_wres64i_kernel(mmmot_gemm_ares_args a) {
  __shared__ __attribute__((aligned(16))) _Float16 Wh[WR_NMAX * 64];
  __shared__ __attribute__((aligned(16))) _Float16 Wl[WR_NMAX * 64];
  __shared__ __attribute__((aligned(16))) float Nrm[WR_THREADS / 64][2][64];  // per wave: prologue scale | shift
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int nblk = a.N >> 5;  // 32-channel blocks: 4, 8, 12 or 16
  {
    const u32x4* wp = reinterpret_cast<const u32x4*>(a.W);
    for (int idx = tid; idx < a.N * 8; idx += WR_THREADS) {
      const int n = idx >> 3, u = idx & 7;
      const u32x4 hi = wp[(long)idx * 2], lo = wp[(long)idx * 2 + 1];
      *reinterpret_cast<u32x4*>(&Wh[wr_off(n, u)]) = hi;
      *reinterpret_cast<u32x4*>(&Wl[wr_off(n, u)]) = lo;
    }
  }
  __syncthreads();  // the only one
  // this wave's run of half tiles: [h, hend)
  const int H = 2 * a.T, nw = gridDim.x * (WR_THREADS / 64), gw = blockIdx.x * (WR_THREADS / 64) + wave;
  const int q = H / nw, rem = H % nw;
  int h = gw * q + (gw < rem ? gw : rem);
  const int hend = h + q + (gw < rem ? 1 : 0);
  if (h >= hend) return;

  // tile-table words of item hh by scalar loads (see gemm_wreg.hip: one asm block with its wait, early-clobber outputs)
  auto item = [&](int hh) {
    const int t = hh >> 1;
    const int* p0 = a.tile_row0 + t;
    const int* p1 = a.tile_nrows + t;
    const int* p2 = a.tile_group ? a.tile_group + t : p0;
    const int* p3 = a.dbias ? a.tile_dbrow + t : p0;
    int v0, v1, v2, v3;
    asm volatile(
        "s_load_dword %0, %4, 0x0\n\t"
        "s_load_dword %1, %5, 0x0\n\t"
        "s_load_dword %2, %6, 0x0\n\t"
        "s_load_dword %3, %7, 0x0\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&s"(v0), "=&s"(v1), "=&s"(v2), "=&s"(v3)
        : "s"(p0), "s"(p1), "s"(p2), "s"(p3)
        : "memory");
    WiItem m;
    m.row0 = v0;
    m.nrows = v1;
    m.grp = a.tile_group ? v2 : 0;
    m.dbrow = a.dbias ? v3 : 0;
    return m;
  };
  // the item's 64 rows in fragment layout: lane (lr, lh) holds k = 16 j + 8 lh .. + 7 of rows lr and 32 + lr; rows past
  // the tile read its first row (their fragments are zeroed)
  f32x4 raw[2][4][2];
  auto request = [&](const WiItem& m, int wm) {
    const float* base = a.X + (long)m.row0 * a.ldx;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int rr = wm * 64 + b * 32 + lr;
      const float* p = base + (unsigned)(rr < m.nrows ? rr : 0) * (unsigned)a.ldx + 8 * lh;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        raw[b][j][0] = *reinterpret_cast<const f32x4*>(p + 16 * j);
        raw[b][j][1] = *reinterpret_cast<const f32x4*>(p + 16 * j + 4);
      }
    }
  };
  f16x8 af[4][4];  // [k16 step][hi rows 0-31, lo rows 0-31, hi rows 32-63, lo rows 32-63]
  auto convert = [&](int nsub) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float* ps = &Nrm[wave][0][16 * j + 8 * lh];
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(ps), s1 = *reinterpret_cast<const f32x4*>(ps + 4);
      const f32x4 t0 = *reinterpret_cast<const f32x4*>(ps + 64), t1 = *reinterpret_cast<const f32x4*>(ps + 68);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const bool ok = b * 32 + lr < nsub;
        float y[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          y[e] = fminf(fmaxf(fmaf(raw[b][j][0][e], s0[e], t0[e]), 0.f), 65000.f);
          y[4 + e] = fminf(fmaxf(fmaf(raw[b][j][1][e], s1[e], t1[e]), 0.f), 65000.f);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (!ok) y[e] = 0.f;
        unsigned hi[4], lo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) mm_split2(y[2 * e], y[2 * e + 1], hi[e], lo[e]);
        af[j][2 * b] = __builtin_bit_cast(f16x8, u32x4{hi[0], hi[1], hi[2], hi[3]});
        af[j][2 * b + 1] = __builtin_bit_cast(f16x8, u32x4{lo[0], lo[1], lo[2], lo[3]});
      }
    }
  };
  // epilogue constants of channel n under (grp, dbrow): the consumer's relu(acc * m1 + m0), as in the kernel above
  auto consts_of = [&](const WiItem& m, int n, float& m1v, float& m0v) {
    const float pb = a.bias ? a.bias[n] : 0.f;
    const float db = (a.dbias ? a.dbias + (long)m.dbrow * a.lddb : wr_zeros)[n];
    const float o = a.osc[(long)m.grp * a.ldosc + n];
    const float cb = pb + db;
    m1v = a.oscale * o;
    m0v = fmaf(cb, o, a.osh[(long)m.grp * a.ldosc + n]);
  };
  float tab[16];
  auto build_tab = [&](const WiItem& m) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      tab[i] = 0.f;
      if (i < nblk) {
        float m1v, m0v;
        consts_of(m, i * 32 + lr, m1v, m0v);
        tab[i] = lh ? m0v : m1v;
      }
    }
  };
  f16x8 bh[2], bl[2];
  auto read_b = [&](int n, int j, int slot) {
    bh[slot] = *reinterpret_cast<const f16x8*>(&Wh[wr_off(n, 2 * j + lh)]);
    bl[slot] = *reinterpret_cast<const f16x8*>(&Wl[wr_off(n, 2 * j + lh)]);
  };
  auto mma_block = [&](f32x16 (&acc)[2], int n_next, bool have_next) {
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[tm][e] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int sl = j & 1;
      if (j < 3) read_b(n_next - 32, j + 1, sl ^ 1);
      else if (have_next) read_b(n_next, 0, sl ^ 1);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[j][1], bh[sl], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[j][3], bh[sl], acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[j][0], bl[sl], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[j][2], bl[sl], acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[j][0], bh[sl], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[j][2], bh[sl], acc[1], 0, 0, 0);
    }
  };

  WiItem cur = item(h), nxt = cur;
  if (h + 1 < hend) nxt = item(h + 1);
  request(cur, h & 1);
  int tab_grp = -1, tab_db = -1, nrm_grp = -1;
  for (;;) {
    const int wm = h & 1;
    const int nsub = min(max(cur.nrows - 64 * wm, 0), 64);  // valid rows of the item
    const bool more = h + 1 < hend;
    if (nsub > 0) {
      if (cur.grp != nrm_grp) {
        const float sv = a.sc[(long)cur.grp * a.ldsc + lane], hv = a.sh[(long)cur.grp * a.ldsc + lane];
        Nrm[wave][0][lane] = sv;
        Nrm[wave][1][lane] = hv;
        nrm_grp = cur.grp;
      }
      if (nsub == 64 && (cur.grp != tab_grp || cur.dbrow != tab_db)) {
        build_tab(cur);
        tab_grp = cur.grp;
        tab_db = cur.dbrow;
      }
      convert(nsub);
    }
    WiItem nn = nxt;
    if (more) {
      request(nxt, (h + 1) & 1);
      if (h + 2 < hend) nn = item(h + 2);
    }
    float* out = a.colsum + (long)h * a.N;
    if (nsub == 64) {
      read_b(lr, 0, 0);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (i < nblk) {
          float m1i = tab[i], m0i = tab[i];  // -> (lanes 0-31 of tab[i], lanes 32-63 of tab[i]) on every lane
          asm volatile("s_nop 3\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 3" : "+v"(m1i), "+v"(m0i));
          f32x16 acc[2];
          mma_block(acc, (i + 1) * 32 + lr, i + 1 < nblk);
          float s3 = mm_relu_sum32(acc[0], acc[1], m1i, m0i);
          s3 = mm_xor32_sum(s3);
          if (lane < 32) out[i * 32 + lr] = s3;
        }
      }
    } else if (nsub > 0) {
      const int lim = nsub - 4 * lh;  // row of accumulator element (tm, e) = tm * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh
      read_b(lr, 0, 0);
      for (int i = 0; i < nblk; ++i) {
        float m1i, m0i;
        consts_of(cur, i * 32 + lr, m1i, m0i);
        f32x16 acc[2];
        mma_block(acc, (i + 1) * 32 + lr, i + 1 < nblk);
        float s3 = 0.f;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int e = 0; e < 16; ++e)
            if (tm * 32 + (e & 3) + 8 * (e >> 2) < lim) s3 += fmaxf(fmaf(acc[tm][e], m1i, m0i), 0.f);
        s3 = mm_xor32_sum(s3);
        if (lane < 32) out[i * 32 + lr] = s3;
      }
    } else {
      if (lane < 32)
        for (int i = 0; i < nblk; ++i) out[i * 32 + lr] = 0.f;
    }
    if (!more) break;
    cur = nxt;
    nxt = nn;
    ++h;
  }
}

int mmmot_ares_variant();  // gemm_wreg.hip

// K = 64, N <= 512: called by mmmot_gemm_ares (gemm_ares.hip) after its argument checks
int mmmot_gemm_wres64_launch(const mmmot_gemm_ares_args* a, int mode, int n_cu, hipStream_t s) {
  // column-sum pass: independent waves when every wave gets a run of half tiles (a small launch is better balanced in
  // 128-row tiles over eight waves); variant 2 = whenever eligible, 3 = never (tests, A/B).  Bit-identical either way.
  const int variant = mmmot_ares_variant();
  if (mode == 2 && variant != 3 && (variant == 2 || 2L * a->T >= 4L * 8 * n_cu)) {
    const int wgs = (2 * a->T + 7) / 8;
    hipLaunchKernelGGL(gemm_wres64i_kernel, dim3(wgs < n_cu ? wgs : n_cu), dim3(WR_THREADS), 0, s, *a);
    return mm_check(hipGetLastError());
  }
  const int grid = a->T < n_cu ? a->T : n_cu;  // persistent: one workgroup per CU (LDS: all 160 KB)
  if (mode == 1)
    hipLaunchKernelGGL(gemm_wres64_kernel<1>, dim3(grid), dim3(WR_THREADS), 0, s, *a);
  else if (mode == 2)
    hipLaunchKernelGGL(gemm_wres64_kernel<2>, dim3(grid), dim3(WR_THREADS), 0, s, *a);
  else
    hipLaunchKernelGGL(gemm_wres64_kernel<3>, dim3(grid), dim3(WR_THREADS), 0, s, *a);
  return mm_check(hipGetLastError());
}
