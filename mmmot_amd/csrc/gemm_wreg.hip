// Weight-in-registers variant of the A-resident row GEMM (include/mmmot_hip.h: mmmot_gemm_ares) for PointNet conv5
// 128 -> 1024 + GroupNorm + ReLU + per-detection mean (reference modules/point_net.py:138-148), the consumer pass:
//
//   colsum[2t+h][n] = sum over the valid rows of 64-row half tile h of tile t of
//                     relu((oscale * sum_k relu(X[r][k]*sc[g][k] + sh[g][k]) * W[n][k] + bias[n] + dbias[..][n]) * osc[g][n] + osh[g][n])
//
// Same arithmetic, accumulation order and summation order as gemm_ares_kernel<2, 2> (gemm_ares.hip) - the results are
// bit for bit the same - different data movement.  The streaming kernel keeps the 128 activation rows in LDS and
// streams the weights through LDS too: every 12 MFMAs of a wave need 4 activation + 4 weight fragment reads and the
// kernel sits at 0.38 of the f16x3 ceiling, LDS-read-bound.  Here a workgroup owns 512 of the N output channels for
// the whole launch and every wave keeps the hi / lo fragments of its 64 channels x K = 128 in REGISTERS (128 of its
// 256): the K loop reads only activation fragments - 2 LDS reads per 6 MFMAs instead of 8 per 12 - and there is no
// weight stream.  N / 512 workgroups share a row tile (channel halves): they sit on the same XCD (block b and b + 8),
// so the second reader of a row tile finds it in that XCD's L2.
//
// The unit of work is a 64-row HALF tile; fragment planes (XOR-swizzled 256-byte rows: conflict-free ds_read_b128) and
// raw rows are double-buffered, everything from global memory arrives by LDS-DMA (raw fp32 rows, the group's sc / sh,
// the wave's epilogue vectors: no staging registers, no compiler-visible load whose vmcnt wait would also wait for the
// requests in flight; the tile tables by hand-written scalar loads for the same reason), and a wave's instruction
// stream carries, between the MFMA pairs of the 32-row block it multiplies,
//   * the column sums of the block it finished before (second accumulator set),
//   * its share of the conversion of the NEXT half tile (raw rows -> normalise, ReLU, hi / lo planes), and
//   * the LDS-DMA requests for the tile after,
// pinned in place with scheduling barriers; one __syncthreads per half tile.  Round 4 history (profiles/README.md):
// converting at the tile switch and summing after each block, all waves together, left the matrix pipe idle for a
// quarter of the launch (3.0 ms per 4.2 M points, 46 % MFMA-busy); this form 2.44 ms.
#include <atomic>
#include <cstdlib>

#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define WR_K 128

static __device__ float wr_zeros[4096];  // stands in for absent bias / dbias rows (branch-free loads)

__device__ __forceinline__ void wr_dma16(const void* src, unsigned char* dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

#define WP_ROWS 64
#define WP_PLANE (WP_ROWS * 256)  // one fp16 plane [64 rows][128 k]: 16 KB
#define WP_P (2 * WP_PLANE)       // hi + lo
#define WP_RAW (WP_ROWS * 512)    // raw fp32 rows of a half tile: 32 KB
#define WP_SC 1024                // sc[128], sh[128] of a tile's group

#define WP_PIN() __builtin_amdgcn_sched_barrier(0)

__global__ __launch_bounds__(512, 1) void gemm_wreg128_kernel(mmmot_gemm_ares_args a, int nh, int nseq) {
  // LDS: [P0 | P1 | Raw0 | Raw1 | SC0 | SC1 | CB0 | CB1 | RI]; P = hi plane, lo plane of a half tile; CB = per wave the four
  // epilogue vectors (bias, dbias row, osc row, osh row) of its 64 channels for one tile; RI = a lane's two partial sums
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * WP_P + 2 * WP_RAW + 2 * WP_SC + 2 * 8 * 1024 + 512 * 8];
  unsigned char* const Pl = smem;
  unsigned char* const Raw = smem + 2 * WP_P;
  float* const SC = reinterpret_cast<float*>(smem + 2 * WP_P + 2 * WP_RAW);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* const CB = reinterpret_cast<float*>(smem + 2 * WP_P + 2 * WP_RAW + 2 * WP_SC) + wave * 256;  // + buf * 2048
  f32x2* const RI = reinterpret_cast<f32x2*>(smem + 2 * WP_P + 2 * WP_RAW + 2 * WP_SC + 2 * 8 * 1024) + tid;
  const int lr = lane & 31, hh = lane >> 5;
  const int b = blockIdx.x;
  const int half = (b >> 3) % nh;
  const int seq0 = (b & 7) + 8 * (b / (8 * nh));
  const int nbase = half * 512 + wave * 64;
  if (seq0 >= nseq) return;
  const int stride = (int)(gridDim.x / (8 * nh)) * 8;
  const int ntile = (a.T - seq0 + stride - 1) / stride;  // tiles of this workgroup: seq0 + q * stride < T

  f16x8 wh[8][2], wl[8][2];
  {
    const u32x4* wp = reinterpret_cast<const u32x4*>(a.W);
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const u32x4* row = wp + (long)(nbase + 32 * tn + lr) * (WR_K / 8) * 2;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        wh[s][tn] = __builtin_bit_cast(f16x8, row[(2 * s + hh) * 2]);
        wl[s][tn] = __builtin_bit_cast(f16x8, row[(2 * s + hh) * 2 + 1]);
      }
    }
  }
  const float* pbias = a.bias ? a.bias : wr_zeros;

  // the table words of a tile (wave-uniform: scalar registers), fetched a tile ahead of their first use
  struct Tile {
    int t, row0, nrows, grp, dbrow;
  };
  // (scalar loads written out: the compiler turns these table reads into VECTOR loads - the kernel stores to global
  // memory, so it will not treat them as constant - and then waits vmcnt(0) for them, i.e. for every LDS-DMA request
  // in flight, right after the requests were issued)
  // The four loads and their wait are ONE asm block with early-clobber outputs: with the wait in a separate statement
  // the compiler does not know the destination registers are pending and may schedule a scalar move / select on them in
  // between.  Optional tables read a valid word (the tile's row0) and are replaced by 0 afterwards.
  auto tile_info = [&](int t) {
    Tile w;
    w.t = t;
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
    w.row0 = v0;
    w.nrows = v1;
    w.grp = a.tile_group ? v2 : 0;
    w.dbrow = a.dbias ? v3 : 0;
    return w;
  };
  // raw rows of half h of a tile -> Raw[h]: 4 instructions of 1 KB (2 rows) per wave; rows past the tile read its first row
  // (addresses: a wave-uniform base pointer plus a 32-bit lane offset, nothing 64-bit per lane kept across the loop)
  auto dma_unit = [&](const Tile& w, int h) {
    const int nr = w.nrows - 64 * h;
    const float* base = a.X + (long)w.row0 * a.ldx;
    int lv = lane;  // (opaque: the lane-dependent address arithmetic is redone here instead of living in registers all loop)
    asm volatile("" : "+v"(lv));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = wave * 4 + i;
      const int r = 2 * q + (lv >> 5);
      const unsigned off = (r < nr ? (unsigned)(64 * h + r) * (unsigned)a.ldx : 0u) + (unsigned)(lv & 31) * 4u;
      wr_dma16(base + off, Raw + h * WP_RAW + q * 1024);
    }
  };
  // conversion of one 32-row chunk (cb = 0 / 1) of a half tile: this thread's 8 k of row 32 cb + tid / 16
  const int kc = tid & 15, crow = tid >> 4;
  typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
  struct Conv {
    f32x2 x, s, h;  // two values at a time, written as 4-byte pieces (registers are what this kernel is short of)
    f16x2 hi, lo;
  };
  auto conv_read = [&](Conv& c, int hbuf, int cb, int sci, int part) {  // part 0..3: k = 8 kc + 2 part, + 1
    const int r = 32 * cb + crow;
    c.x = *reinterpret_cast<const f32x2*>(Raw + hbuf * WP_RAW + r * 512 + kc * 32 + 8 * part);
    c.s = *reinterpret_cast<const f32x2*>(SC + sci * 256 + kc * 8 + 2 * part);
    c.h = *reinterpret_cast<const f32x2*>(SC + sci * 256 + 128 + kc * 8 + 2 * part);
  };
  auto conv_val = [&](Conv& c, int e, int part, bool rv) {
    float y = fminf(fmaxf(fmaf(c.x[e], c.s[e], c.h[e]), 0.f), 65000.f);
    if (!rv) y = 0.f;
    c.x[e] = y;  // (the raw value is dead: its register carries the normalised one to the split)
    if (e == 1) {  // both values of the piece are there: hi = one v_cvt_pk, lo = one v_fma_mix per value (common.h)
      unsigned h2, l2;
      mm_split2(c.x[0], c.x[1], h2, l2);
      c.hi = __builtin_bit_cast(f16x2, h2);
      c.lo = __builtin_bit_cast(f16x2, l2);
    }
  };
  auto conv_write = [&](const Conv& c, int hbuf, int cb, int part) {
    const int r = 32 * cb + crow;
    const int off = hbuf * WP_P + r * 256 + ((kc ^ (r & 15)) << 4) + 4 * part;
    *reinterpret_cast<f16x2*>(Pl + off) = c.hi;
    *reinterpret_cast<f16x2*>(Pl + off + WP_PLANE) = c.lo;
  };
  auto convert_plain = [&](int hbuf, int sci, int nr) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      Conv c;
      const bool rv = 32 * cb + crow < nr;
#pragma unroll
      for (int part = 0; part < 4; ++part) {
        conv_read(c, hbuf, cb, sci, part);
#pragma unroll
        for (int e = 0; e < 2; ++e) conv_val(c, e, part, rv);
        conv_write(c, hbuf, cb, part);
      }
    }
  };
  auto dma_piece = [&](const Tile& w, int h, int i) {  // instruction i of dma_unit
    const int nr = w.nrows - 64 * h;
    const float* base = a.X + (long)w.row0 * a.ldx;
    int lv = lane;
    asm volatile("" : "+v"(lv));
    const int q = wave * 4 + i;
    const int r = 2 * q + (lv >> 5);
    const unsigned off = (r < nr ? (unsigned)(64 * h + r) * (unsigned)a.ldx : 0u) + (unsigned)(lv & 31) * 4u;
    wr_dma16(base + off, Raw + h * WP_RAW + q * 1024);
  };
  auto sc_dma = [&](const Tile& w, int sci) {  // wave 0: sc (lanes 0..31) and sh (32..63) of the tile's group -> SC[sci], 1 KB
    unsigned char* dst = reinterpret_cast<unsigned char*>(SC + sci * 256);  // a lane's 16 bytes land at dst + 16 lane
    int lv = lane;
    asm volatile("" : "+v"(lv));
    const unsigned off = (unsigned)(lv & 31) * 4u;
    if (lv < 32) wr_dma16(a.sc + (long)w.grp * a.ldsc + off, dst);
    else wr_dma16(a.sh + (long)w.grp * a.ldsc + off, dst);
  };
  // the wave's epilogue vectors of a tile -> CB[buf]: lanes 0..15 bias, 16..31 the dbias row, 32..47 osc, 48..63 osh (64
  // channels = 256 B each); no registers, no wait - they are read a half tile later
  auto cb_dma = [&](const Tile& w, int buf) {
    unsigned char* dst = reinterpret_cast<unsigned char*>(CB + buf * 2048);
    int lv = lane;
    asm volatile("" : "+v"(lv));
    const unsigned off = (unsigned)(lv & 15) * 4u;
    const int g = lv >> 4;
    if (g == 0) wr_dma16(pbias + nbase + off, dst);
    else if (g == 1) wr_dma16((a.dbias ? a.dbias + (long)w.dbrow * a.lddb + nbase : wr_zeros) + off, dst);
    else if (g == 2) wr_dma16(a.osc + (long)w.grp * a.ldosc + nbase + off, dst);
    else wr_dma16(a.osh + (long)w.grp * a.ldosc + nbase + off, dst);
  };
  const unsigned cl = (unsigned)lr;
  auto tile_consts = [&](int buf, float* m1, float* m0) {  // from CB[buf] (requested a half tile ago, complete at the barrier)
    const float* c = CB + buf * 2048;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const float cb = c[32 * tn + lr] + c[64 + 32 * tn + lr];
      const float os = c[128 + 32 * tn + lr], oh = c[192 + 32 * tn + lr];
      m1[tn] = a.oscale * os;
      m0[tn] = fmaf(cb, os, oh);
    }
  };
  // masked epilogue of a block (rows past the half tile's end do not count): not interleaved, rare
  auto epi_masked = [&](const f32x16* acc, const float* m1, const float* m0, float* run, int lim) {
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      float s = run[tn];
#pragma unroll
      for (int e = 0; e < 16; ++e)
        if ((e & 3) + 8 * (e >> 2) < lim) s += fmaxf(fmaf(acc[tn][e], m1[tn], m0[tn]), 0.f);
      run[tn] = s;
    }
  };
  auto store_sums = [&](int t, int h, float* run) {
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const float tot = mm_xor32_sum(run[tn]);
      float* pc = a.colsum + (long)(2 * t + h) * a.N + nbase;
      if (lane < 32) pc[32 * tn + cl] = tot;
      run[tn] = 0.f;
    }
  };

  // One 32-row block: 48 MFMAs into accC from the planes P[pb] rows 32 blk .., with between the MFMA pairs
  //   E: run[] += relu(accE * mE1 + mE0) over accE's 32 values, in register order, and
  //   C: chunk cb of the conversion Raw[chb] -> P[chb].
  // Both always run (no branches inside the pinned stream): where there is nothing to sum or convert the caller
  // discards the sums / nobody reads the planes.
  auto stream = [&](f32x16* accC, const f32x16* accE, f16x8& ah, f16x8& al, int pb, int blk, bool nextblk, const float* mE1,
                    const float* mE0, float* run, int chb, int cb, int sci, bool crv, int dma, const Tile& dt) {
    const int r = 32 * blk + lr;
    const unsigned char* Ah = Pl + pb * WP_P + r * 256;
    const unsigned char* Al = Ah + WP_PLANE;
    const int sw = r & 15;
    Conv c;
    float e0 = run[0], e1 = run[1];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const int k = 3 * s + p;
        // next step's piece of the row / step 0 of the next 32 rows (same planes)
        const int pn = s < 7 ? (((2 * (s + 1) + hh) ^ sw) << 4) : (((hh ^ sw) << 4) + 32 * 256);
        if (p == 0) {
          if (s == 0) {
            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            accC[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh[0][0], z, 0, 0, 0);
            accC[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh[0][1], z, 0, 0, 0);
          } else {
            accC[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh[s][0], accC[0], 0, 0, 0);
            accC[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh[s][1], accC[1], 0, 0, 0);
          }
          // one fragment set: the lo fragment is free once this pair has issued, the hi fragment after the third pair
          if (s < 7 || nextblk) al = *reinterpret_cast<const f16x8*>(Al + pn);
        } else if (p == 1) {
          accC[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl[s][0], accC[0], 0, 0, 0);
          accC[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl[s][1], accC[1], 0, 0, 0);
        } else {
          accC[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh[s][0], accC[0], 0, 0, 0);
          accC[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh[s][1], accC[1], 0, 0, 0);
          if (s < 7 || nextblk) ah = *reinterpret_cast<const f16x8*>(Ah + pn);
        }
        // E: flat elements [32 k / 24, 32 (k + 1) / 24) of (tn, e)
#pragma unroll
        for (int f = (32 * k) / 24; f < (32 * (k + 1)) / 24; ++f) {
          const float v = fmaxf(fmaf(accE[f >> 4][f & 15], mE1[f >> 4], mE0[f >> 4]), 0.f);
          if (f < 16) e0 += v; else e1 += v;
        }
        // (the sums are used only after the stream, in places under a condition: without this the compiler sinks the
        // whole chain out of the pinned slots to that use - the scheduling barriers do not bind its IR passes)
        asm volatile("" : "+v"(e0), "+v"(e1));
        // C
        if (k % 5 == 1 && k < 20) conv_read(c, chb, cb, sci, k / 5);        // k = 1, 6, 11, 16
        if (k % 5 == 3 && k < 20) conv_val(c, 0, k / 5, crv);               // k = 3, 8, 13, 18
        if (k % 5 == 4 && k < 20) conv_val(c, 1, k / 5, crv);               // k = 4, 9, 14, 19
        if (k % 5 == 0 && k >= 5 && k <= 20) conv_write(c, chb, cb, k / 5 - 1);  // k = 5, 10, 15, 20
        // D: the LDS-DMA requests for the next tile, one per slot (the first stream after a barrier)
        if (dma == 1) {  // half 0 of the next tile -> Raw0, its group's sc / sh -> SC[sci ^ 1]
          if (k >= 16 && k < 20) dma_piece(dt, 0, k - 16);
          if (k == 21 && wave == 0) sc_dma(dt, sci ^ 1);
        }
        if (dma == 2) {  // half 1 of the next tile -> Raw1, its epilogue vectors -> CB[sci] (the caller passes sci ^ 1)
          if (k >= 16 && k < 20) dma_piece(dt, 1, k - 16);
          if (k == 21) cb_dma(dt, sci);
        }
        WP_PIN();
      }
    }
    run[0] = e0;
    run[1] = e1;
  };

  // ---- prologue: both halves of tile 0 requested, its group's sc / sh in SC[0], its epilogue vectors in CB[0], half 0
  // converted ----
  Tile cur = tile_info(seq0);
  Tile nxt = tile_info(ntile > 1 ? seq0 + stride : seq0);
  dma_unit(cur, 0);
  dma_unit(cur, 1);
  if (wave == 0) sc_dma(cur, 0);
  cb_dma(cur, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  convert_plain(0, 0, cur.nrows);

  auto first_frags = [&](f16x8& ah, f16x8& al, int pb) {  // block 0, step 0 of planes P[pb]
    const unsigned char* Ah = Pl + pb * WP_P + lr * 256 + ((hh ^ (lr & 15)) << 4);
    ah = *reinterpret_cast<const f16x8*>(Ah);
    al = *reinterpret_cast<const f16x8*>(Ah + WP_PLANE);
  };
  f16x8 ah, al;
  const f32x16 z16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x16 acc0[2] = {z16, z16}, acc1[2] = {z16, z16};
  float m1[2] = {0.f, 0.f}, m0[2] = {0.f, 0.f};  // the epilogue constants: of the PREVIOUS tile until the first stream of a
                                                 // tile has summed that tile's last block, then of the current one
  float run[2] = {0.f, 0.f};
  int tprev = seq0, nrprev = 0;
  // A block's sums are taken unmasked inside the next stream.  A block that is not full is summed again, masked, from the
  // partial sums the lane parked in LDS before that stream (rare: the last tile of a detection).
  auto park = [&]() { *RI = f32x2{run[0], run[1]}; };
  auto fix = [&](const f32x16* acc, int rows) {  // rows of the block that exist
    if (rows < 32) {
      const f32x2 r = *RI;
      run[0] = r[0];
      run[1] = r[1];
      epi_masked(acc, m1, m0, run, rows - 4 * hh);
    }
  };
  for (int q = 0; q < ntile; ++q) {
    const int nrows = cur.nrows, nrnext = nxt.nrows;
    const int sci = q & 1;
    // ================= half 0: planes P0; converts half 1 of this tile (Raw1 -> P1) =================
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of Raw1 and CB[sci] (requested a half tile ago)
    __syncthreads();
    park();
    first_frags(ah, al, 0);
    stream(acc0, acc1, ah, al, 0, 0, true, m1, m0, run, 1, 0, sci, crow < nrows - 64, 1, nxt);  // + sums of the previous tile's last block
    if (q > 0) {
      fix(acc1, nrprev - 96);
      store_sums(tprev, 1, run);
    }
    run[0] = run[1] = 0.f;
    tile_consts(sci, m1, m0);
    park();
    stream(acc1, acc0, ah, al, 0, 1, false, m1, m0, run, 1, 1, sci, 32 + crow < nrows - 64, 0, nxt);  // + sums of block 0
    fix(acc0, nrows);
    // ================= half 1: planes P1; converts half 0 of the next tile (Raw0 -> P0) =================
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // Raw0 of the next tile
    __syncthreads();
    const Tile nn = tile_info(q + 2 < ntile ? cur.t + 2 * stride : cur.t);  // used from the next half tile on
    park();
    first_frags(ah, al, 1);
    stream(acc0, acc1, ah, al, 1, 0, true, m1, m0, run, 0, 0, sci ^ 1, crow < nrnext, 2, nxt);  // + sums of block 1 of half 0
    fix(acc1, nrows - 32);
    store_sums(cur.t, 0, run);
    park();
    stream(acc1, acc0, ah, al, 1, 1, false, m1, m0, run, 0, 1, sci ^ 1, 32 + crow < nrnext, 0, nxt);  // + sums of block 0 of half 1
    fix(acc0, nrows - 64);
    tprev = cur.t;
    nrprev = nrows;
    cur = nxt;
    nxt = nn;
  }
  // ---- drain: the last block of the last tile ----
  epi_masked(acc1, m1, m0, run, nrprev - 96 - 4 * hh);
  store_sums(tprev, 1, run);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing may still be in flight towards this workgroup's LDS
}

static std::atomic<int> g_ares_variant{0};
// Test knob: 0 = automatic (K = 128: the register-resident kernel for the consumer pass when the launch fills the chip;
// K = 64: the independent-wave kernel of gemm_wres.hip for a column-sum pass that gives every wave a run of half tiles),
// 1 = the streaming kernel only, 2 = the register-resident / independent-wave kernel whenever the layer is eligible,
// 3 = K = 64: the two-barrier weight-resident kernel only (K = 128: as 0).  Results do not depend on it (bit for bit).
extern "C" int mmmot_set_gemm_ares_variant(int v) {
  if (v < 0 || v > 3) return MMMOT_EINVAL;
  g_ares_variant.store(v);
  return MMMOT_OK;
}
int mmmot_ares_variant() { return g_ares_variant.load(); }

// Called by mmmot_gemm_ares after its argument checks; returns 1 when this kernel took the launch.
int mmmot_gemm_wreg128_try(const mmmot_gemm_ares_args* a, int mode, int n_cu, hipStream_t s, int* status) {
  const int variant = g_ares_variant.load();
  if (variant == 1 || a->K != WR_K || mode != 2 || a->N % 512 != 0 || a->N > 4096) return 0;
  if (a->ldx % 4 != 0 || a->ldx < WR_K) return 0;
  if (a->ldosc % 4 != 0 || a->ldsc % 4 != 0 || (a->dbias && a->lddb % 4 != 0)) return 0;  // 16-byte LDS-DMA pieces
  const int nh = a->N / 512;
  // the launch must fill the chip with whole (tile sequence x channel half) groups of 8 workgroups per XCD
  if (variant != 2 && (long)a->T * nh < 2L * n_cu) return 0;
  int groups = n_cu / (8 * nh);                 // sequences of 8 workgroups per channel half
  if (groups < 1) groups = 1;
  const int need = (a->T + 7) / 8;              // sequences of 8 tiles there are
  if (groups > need) groups = need;
  const int grid = groups * 8 * nh;
  const int nseq = a->T < 8 * groups ? a->T : 8 * groups;  // sequence starts that exist
  hipLaunchKernelGGL(gemm_wreg128_kernel, dim3(grid), dim3(512), 0, s, *a, nh, nseq);
  *status = mm_check(hipGetLastError());
  return 1;
}
