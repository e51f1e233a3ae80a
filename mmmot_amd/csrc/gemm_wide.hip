// Wide row GEMM for the K = 512 layers of the pairwise block (reference modules/gcn.py:59-82,
// modules/new_end.py:48-52): the stacked [w_new_end.conv0 | w_link.conv1.0] layer over the on-the-fly pair tensor
// (A_PAIR) and the two GroupNorm-fed layers behind it (A_NORM_RELU).  Same contract and arithmetic as the f16 path of
// gemm_rows.hip (3-term fp16 hi/lo split on v_mfma_f32_32x32x16_f16, fp32 accumulate, identical accumulation order,
// identical order of the statistics sums: Y and part are bit for bit the tile kernel's, so a sample scores the same
// whether it runs alone on the tile kernel or in a batch on this one), different data movement - driven by the round-3
// profile: the tile kernel regenerates the A tile (op(a_i, b_j) or relu(x*sc+sh), clamp, hi/lo split, ds_write: ~7 VALU
// instructions per element) for every one of the N/128 column tiles and spends MORE VALU than MFMA cycles (21 %
// MFMA-busy at full clock).
//
//   * a workgroup = 8 waves = TWO 128-row tiles of the plan x BN = 32 TN columns (TN = 8: 256 columns, 128 accumulator
//     registers per wave); one wave owns 32 rows x all BN columns: its A fragments are generated IN REGISTERS, straight
//     in the MFMA operand layout (lane = row, 8 consecutive k), from 32-byte global loads of the fp32 source rows issued
//     two 16-k steps ahead - no LDS traffic for A at all, and every generated fragment feeds 3 TN MFMAs: VALU work per
//     MFMA drops 4x against the tile kernel and the 256 rows share one copy of the weight stage;
//   * what is the same for all 32 rows of a wave - the a_i row of a pair tile (the caller guarantees M % 32 == 0, so a
//     32-row block never straddles two i), the per-group scale / shift rows of the GroupNorm prologue - is staged in LDS
//     once per item and read as broadcast fragments: the kernel is bound by the RATE of vector-memory instructions a CU
//     gets through (~90 cycles each with eight requesting waves), and those redundant per-lane requests were a third
//     (PAIR) / half (NORM_RELU) of them;
//   * weights stream through a 3-slot LDS ring of (BN rows x 128 B) stages (32 k = 4 hl16 units per row) by LDS-DMA
//     (global_load_lds, no VGPR staging), two stages ahead, XOR-swizzled on the source side like the trunk kernel's
//     weight ring (piece ^ ((row >> 1) & 7): conflict-free ds_read_b128 fragments); one s_barrier per stage with a
//     counted vmcnt (the requests of the stage after next stay in flight across it).  All eight waves load: the L2 -> CU
//     path returns ~3.5 B/clk per requesting wave (profiles/README.md, r01 probe) - a first version of this kernel with
//     four 512-register waves per CU (32 rows x 512 columns each) was load-bound at 27 % MFMA-busy like the tile kernel;
//   * persistent workgroups (one per CU), tiles chained: the first two weight stages and the source rows of the NEXT item
//     are requested during the last two stages of the current one; the output leaves through LDS as 16-byte stores.
// Epilogue: bias, per-tile per-channel sum / tile-centred M2 (input of mmmot_gn_finalize), store - as gemm_rows.hip.
#include <atomic>
#include <type_traits>

#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define GW_BK 32
#define GW_ROWB 128  // bytes of one weight row per stage
#define GW_NSLOT 3   // ring slots: the stage in use, the next one (landed), the one after (in flight)

__device__ __forceinline__ void gw_dma16(const u32x4* src, unsigned char* dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

__device__ __forceinline__ void gw_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int N>
__device__ __forceinline__ void gw_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct GwRaw {  // source values of one 16-k step of this lane's row: 8 consecutive k of b_j (PAIR) / X (NORM_RELU)
  f32x4 x[2];
};

template <int TN, int AMODE, int PAIROP>
__global__ __launch_bounds__(512, 1) void gemm_wide_kernel(mmmot_gemm_args a, int ntn, int nitems) {
  constexpr int BN = 32 * TN;
  constexpr int SLOT = BN * GW_ROWB;            // 32 KB (TN = 8) / 16 KB (TN = 4)
  constexpr int NG = TN / 4;                    // column groups of 4 MFMA tiles: 2 / 1
  constexpr int NDMA = BN / 8 / 8;              // weight DMA instructions per wave and stage: 4 / 2
  constexpr int NRAW = 2;                       // source-row load instructions per wave and 16-k step (b_j / X: 32 B per lane)
  constexpr int SCOLS = 16;                     // output staging: 16 columns at a time
  constexpr int CLD = SCOLS + 4;                // floats per staged row
  constexpr int STG_WAVE = 32 * CLD * 4;        // bytes per wave
  constexpr int RED_OFF = GW_NSLOT * SLOT;      // [8][BN] per-wave column partials, [2][BN] column means
  constexpr int STG_OFF = RED_OFF + 10 * BN * 4;
  // wave-uniform source vectors, double-buffered by item parity: PAIR [2][8 waves][512] (a_i of the wave's rows),
  // NORM_RELU [2][2 halves][2][512] (scale, shift of the half's group)
  constexpr int UNI_OFF = STG_OFF + 8 * STG_WAVE;
  constexpr int UNI_BUF = (AMODE == MMMOT_A_PAIR) ? 8 * 2048 : 2 * 2 * 2048;
  constexpr int SMEM = UNI_OFF + 2 * UNI_BUF;
  static_assert(SMEM <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave >> 2, wq = wave & 3;    // plan tile of the item (0 / 1), 32-row block inside it
  const int lr = lane & 31, h = lane >> 5;
  const int nst = a.K / GW_BK;
  const long wrow = (long)(a.K >> 2);           // 16-byte pieces per weight row
  const u32x4* Wp = reinterpret_cast<const u32x4*>(a.W);

  // weight DMA: instruction q = wave * NDMA + b covers rows 8q .. 8q + 7; lane = (row in 8, slot in row); the swizzle
  // (row >> 1) & 7 = (4 q + (lane >> 4)) & 7 depends on the parity of q only (NDMA is even: parity of q = parity of b)
  unsigned dma_off[2];
#pragma unroll
  for (int par = 0; par < 2; ++par)
    dma_off[par] = (unsigned)((lane >> 3) * wrow + ((lane & 7) ^ ((4 * par + (lane >> 4)) & 7)));
  // weight fragment of column tile tn, k-step j: row r = 32 tn + lr, piece (4 j + 2 h) ^ ((r >> 1) & 7) (hi; lo: ^ 1)
  const int boff = lr * GW_ROWB + (((2 * h) ^ ((lr >> 1) & 7)) << 4);

  struct Item {  // this WAVE's plan tile of the item (the two halves of a workgroup may belong to different groups)
    int t, nt, row0, nrows, grp;
  };
  // item -> (pair of row tiles u, column tile nt): the ntn column tiles of a row-tile pair are items it, it + 8, ... of one
  // group of 8 ntn - with a grid that is a multiple of 8 ntn they run at the same time on workgroups b, b + 8, ... = on the
  // SAME XCD, so the source rows (NORM_RELU: X from HBM; PAIR: a_i / b_j) are fetched once into that XCD's L2 and the
  // other column tiles hit it.  Row-tile pairs beyond the last one (the item count is padded to whole groups) are empty.
  auto item_u = [&](int it) { return (it & 7) + 8 * (it / (8 * ntn)); };
  auto decode = [&](int it, Item& I) {
    const int u = item_u(it);
    I.nt = (it >> 3) % ntn;
    const int t = 2 * u + half;
    const bool ok = t < a.T;
    I.t = ok ? t : a.T - 1;
    I.row0 = a.tile_row0[I.t];
    I.nrows = ok ? a.tile_nrows[I.t] : 0;  // an odd tile count: the last workgroup's second half is empty
    I.grp = a.tile_group ? a.tile_group[I.t] : 0;
  };
  // per-lane source rows of the A operand (row 32 wq + lr of the tile; rows beyond the tile read row 0 and are zeroed)
  struct Src {
    const float* p0;   // per lane: b_j (PAIR) / X row (NORM_RELU), + 8 h
    const float* u0;   // wave-uniform rows for the LDS staging: a_i (PAIR) / scale (NORM_RELU)
    const float* u1;   //                                        shift (NORM_RELU)
    float top;
  };
  auto sources = [&](const Item& I, Src& S) {
    const int r = wq * 32 + lr;
    const bool ok = r < I.nrows;
    const int rr = ok ? r : 0;
    S.top = ok ? 65000.f : 0.f;
    if constexpr (AMODE == MMMOT_A_PAIR) {
      const int q = I.row0 + rr - a.grp_row0[I.grp];
      const int M = a.grp_M[I.grp];
      const int ii = q / M, jj = q - ii * M;
      S.p0 = a.FB + (long)(a.grp_boff[I.grp] + jj) * a.ldf + 8 * h;
      // M % 32 == 0 (pair_uniform32) and tiles start at multiples of 128 rows of their group: the wave's rows share ii;
      // rows beyond the tile stand in for the wave's first row (or, in a wave without valid rows, the tile's first row:
      // it contributes zeros whatever it reads)
      const int iw = __builtin_amdgcn_readfirstlane(ii);
      S.u0 = a.FA + (long)(a.grp_aoff[I.grp] + iw) * a.ldf;
      S.u1 = nullptr;
    } else {
      S.p0 = a.X + (long)(I.row0 + rr) * a.ldx + 8 * h;
      S.u0 = a.sc + (long)I.grp * a.ldsc;
      S.u1 = a.sh + (long)I.grp * a.ldsc;
    }
  };
  auto load_raw = [&](const Src& S, int s, int j, GwRaw& R) {  // k = 32 s + 16 j + 8 h .. + 7
    const int k = s * GW_BK + 16 * j;
    R.x[0] = *reinterpret_cast<const f32x4*>(S.p0 + k);
    R.x[1] = *reinterpret_cast<const f32x4*>(S.p0 + k + 4);
  };
  // the wave-uniform vectors of an item -> LDS buffer `ub` (lane-linear 1 KB pieces: 256 floats per instruction)
  auto dma_uniform = [&](const Src& S, int ub) {
    if constexpr (AMODE == MMMOT_A_PAIR) {
      unsigned char* dst = smem + UNI_OFF + ub * UNI_BUF + wave * 2048;
      gw_dma16(reinterpret_cast<const u32x4*>(S.u0) + lane, dst);
      if (a.K > 256) gw_dma16(reinterpret_cast<const u32x4*>(S.u0) + 64 + lane, dst + 1024);
    } else {
      // wave 1 of each half fetches its group's shift row, the other waves the scale row (the same bytes into the same
      // place): every wave issues the same NUMBER of requests - the counted waits assume it
      const int v = (wq == 1) ? 1 : 0;
      unsigned char* dst = smem + UNI_OFF + ub * UNI_BUF + (half * 2 + v) * 2048;
      const float* src = v ? S.u1 : S.u0;
      gw_dma16(reinterpret_cast<const u32x4*>(src) + lane, dst);
      if (a.K > 256) gw_dma16(reinterpret_cast<const u32x4*>(src) + 64 + lane, dst + 1024);
    }
  };
  // the A fragment of one k-step: op / normalise, fp16 range clamp (rows beyond the tile: bound 0), hi/lo split
  // step: 16-k step inside the item (k = 16 step + 8 h); ub: LDS buffer of the item's uniform vectors
  auto generate = [&](const GwRaw& R, float top, int step, int ub, f16x8& ah, f16x8& al) {
    const int k = 16 * step + 8 * h;
    f32x4 u4[2], w4[2];
    if constexpr (AMODE == MMMOT_A_PAIR) {
      const float* U = reinterpret_cast<const float*>(smem + UNI_OFF + ub * UNI_BUF + wave * 2048) + k;
      u4[0] = *reinterpret_cast<const f32x4*>(U);
      u4[1] = *reinterpret_cast<const f32x4*>(U + 4);
    } else {
      const float* U = reinterpret_cast<const float*>(smem + UNI_OFF + ub * UNI_BUF + half * 2 * 2048) + k;
      u4[0] = *reinterpret_cast<const f32x4*>(U);
      u4[1] = *reinterpret_cast<const f32x4*>(U + 4);
      w4[0] = *reinterpret_cast<const f32x4*>(U + 512);
      w4[1] = *reinterpret_cast<const f32x4*>(U + 516);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      // PAIR: x = b_j (per lane), u = a_i (the wave's row): a_i op b_j with the operand order of the tile kernel
      const float x = R.x[e >> 2][e & 3], u = u4[e >> 2][e & 3];
      float y;
      if constexpr (AMODE == MMMOT_A_PAIR) {
        if constexpr (PAIROP == MMMOT_PAIR_MULTIPLY) y = u * x;
        else if constexpr (PAIROP == MMMOT_PAIR_MINUS_ABS) y = fabsf((u - x) * 0.5f);
        else y = (u - x) * 0.5f;
      } else {
        y = fmaxf(fmaf(x, u, w4[e >> 2][e & 3]), 0.f);
      }
      y = __builtin_amdgcn_fmed3f(y, -top, top);
      ah[e] = (_Float16)y;
      al[e] = (_Float16)(y - (float)ah[e]);
    }
  };
  auto issue_dma = [&](auto BC, const u32x4* wbase, int s, int slot) {  // instruction b of this wave's NDMA
    constexpr int b = decltype(BC)::value;
    if constexpr (b < NDMA) {
      const int q = wave * NDMA + b;
      const unsigned long ubl = (unsigned long)(wbase + ((long)(8 * q) * wrow + 8 * s));
      const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(ubl >> 32));
      const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ubl);
      const u32x4* ub = (const u32x4*)(((unsigned long)hi << 32) | (unsigned long)lo);
      gw_dma16(ub + dma_off[b & 1], smem + slot * SLOT + q * 1024);
    }
  };
  auto dma_stage = [&](const u32x4* wb, int s2, int slot2) {  // this wave's share of one weight stage
    issue_dma(std::integral_constant<int, 0>{}, wb, s2, slot2);
    issue_dma(std::integral_constant<int, 1>{}, wb, s2, slot2);
    issue_dma(std::integral_constant<int, 2>{}, wb, s2, slot2);
    issue_dma(std::integral_constant<int, 3>{}, wb, s2, slot2);
  };

  // ---- persistent loop over items (pair of row tiles x column tile; column tile fastest) -----------------------
  // The K loop is ONE stream of 16-k steps across the items of a workgroup, software-pipelined by hand:
  //   step n uses the A fragment f[n & 1] and walks the NG column groups; the weight fragments of group g + 1 (or of
  //   group 0 of step n + 1) are read BEFORE the 12 MFMAs of group g are issued; fragment f[(n + 1) & 1] is generated
  //   under the MFMAs of group 0 from the source values raw[(n + 1) & 1], which are then reloaded with the values of
  //   step n + 3; the weights of stage s + 2 are requested at the start of stage s and the stage barrier sits in front
  //   of the LAST group of the stage's odd step - behind it the first fragments of the next stage are read, so the
  //   matrix cores never wait for a barrier or a load.
  // Order of the vector-memory instructions of a stage (per wave): [even step] NDMA weight requests, NRAW source loads;
  // [odd step] NRAW source loads - the counted wait at the stage barrier leaves the requests of the two newest steps'
  // source rows and of the newest weight stage in flight.
  int item = blockIdx.x;
  if (item >= nitems) return;
  Item cur, nxt;
  Src scur, snxt;
  decode(item, cur);
  sources(cur, scur);
  const u32x4* wcur = Wp + (long)(cur.nt * BN) * wrow;
  GwRaw raw[2];
  f16x8 fh[2], fl[2];
  f16x8 bh[2][4], bl[2][4];  // weight fragments: [buffer][column tile of the group]
  auto read_b = [&](int buf, int bbase, int j, int g) {  // bbase: boff + slot * SLOT
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int off = (bbase ^ (64 * j)) + (4 * g + q) * 32 * GW_ROWB;
      bh[buf][q] = *reinterpret_cast<const f16x8*>(smem + off);
      bl[buf][q] = *reinterpret_cast<const f16x8*>(smem + (off ^ 16));
    }
  };
  // source values of stream step m of the CURRENT item (m >= 2 nst: the successor's step m - 2 nst)
  auto load_stream = [&](int m, GwRaw& R) {
    const bool nx = m >= 2 * nst;
    const int ml = nx ? m - 2 * nst : m;
    Src sn;
    sn.p0 = nx ? snxt.p0 : scur.p0;
    load_raw(sn, ml >> 1, ml & 1, R);
  };
  int ub = 0;  // LDS buffer of the current item's uniform vectors
  {  // first item of this workgroup: load prologue (K >= 64: two stages exist)
    dma_uniform(scur, 0);
    dma_stage(wcur, 0, 0);
    load_raw(scur, 0, 0, raw[0]);
    load_raw(scur, 0, 1, raw[1]);
    dma_stage(wcur, 1, 1);
    gw_wait_vm<NDMA>();  // stage 0 and the first source values landed; stage 1 may be in flight
    __builtin_amdgcn_s_barrier();
    generate(raw[0], scur.top, 0, 0, fh[0], fl[0]);
    load_raw(scur, 1, 0, raw[0]);
    read_b(0, boff, 0, 0);
  }
  int slot = 0;  // ring slot of the current stage
  for (; item < nitems; item += gridDim.x) {
    const int item_n = item + gridDim.x;
    const bool has_next = item_n < nitems;
    // (no successor: the last stages request this item's first stages again - harmless, and the K loop stays free of
    // branches: one scheduling region per stage)
    decode(has_next ? item_n : item, nxt);
    sources(nxt, snxt);
    const u32x4* wnxt = Wp + (long)(nxt.nt * BN) * wrow;
    dma_uniform(snxt, ub ^ 1);  // read from the last step of this item on: many stage barriers away
    f32x16 acc[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[tn][e] = 0.f;

    for (int s = 0; s < nst; ++s) {
      const int slot1 = (slot == GW_NSLOT - 1) ? 0 : slot + 1;     // next stage
      const int slot2 = (slot1 == GW_NSLOT - 1) ? 0 : slot1 + 1;   // the stage after it: requested now
      const bool wrap = (s + 2 >= nst);                            // it belongs to the successor
      const u32x4* wb2 = wrap ? wnxt : wcur;
      const int s2 = wrap ? s + 2 - nst : s + 2;
      const bool last = (s == nst - 1);
      const int bcur = boff + slot * SLOT, bnxt = boff + slot1 * SLOT;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = 2 * s + j;  // step of this item; f[j] holds its A fragment
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          // fragment buffers: group (step, g) uses buffer (NG * step + g) & 1; NG = 2: g, NG = 1: j
          const int bc = (NG == 1) ? j : (g & 1), bn = bc ^ 1;
          // ---- requests first: the next group's weight fragments; the weight stage after next; the source rows ----
          if (g + 1 < NG) {
            read_b(bn, bcur, j, g + 1);
          } else if (j == 0) {
            read_b(bn, bcur, 1, 0);  // group 0 of the odd step, same slot
          } else {
            // the next stage's weights (requested a stage ago) and every older request have landed; the requests of
            // this stage (weights of the stage after next, source rows of the two steps to come) may stay in flight;
            // every wave is past its last read of this slot
            gw_wait_vm<NDMA + 2 * NRAW>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            read_b(bn, bnxt, 0, 0);
          }
          if (j == 0 && g == 0) dma_stage(wb2, s2, slot2);
          if (g == 0) {  // the A fragment of step n + 1 (the successor's first step behind the last one)
            const bool nx = last && j == 1;
            generate(raw[j ^ 1], nx ? snxt.top : scur.top, nx ? 0 : n + 1, nx ? (ub ^ 1) : ub, fh[j ^ 1], fl[j ^ 1]);
            load_stream(n + 3, raw[j ^ 1]);
          }
          // ---- 12 MFMAs, term-major: consecutive ones hit different accumulators (per accumulator the order of
          // gemm_rows.hip: lo*hi, hi*lo, hi*hi) ----
#pragma unroll
          for (int q = 0; q < 4; ++q)
            acc[4 * g + q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[j], bh[bc][q], acc[4 * g + q], 0, 0, 0);
#pragma unroll
          for (int q = 0; q < 4; ++q)
            acc[4 * g + q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[j], bl[bc][q], acc[4 * g + q], 0, 0, 0);
#pragma unroll
          for (int q = 0; q < 4; ++q)
            acc[4 * g + q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[j], bh[bc][q], acc[4 * g + q], 0, 0, 0);
        }
      }
      slot = slot1;
    }

    // ---------------- epilogue ---------------------------------------------------------------------------------
    // (the ring is untouched: the successor's first two stages are in it / on their way)
    const int n0 = cur.nt * BN;
    const float oscale = a.oscale;
    const int nrows = cur.nrows;
    const int rbase = wq * 32;
    // v = oscale * accumulator + bias is formed where it is used (three times)
    float bv[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) bv[tn] = a.bias ? a.bias[n0 + tn * 32 + lr] : 0.f;
    if (a.Y) {
      float* stg = reinterpret_cast<float*>(smem + STG_OFF + wave * STG_WAVE);  // wave-private
#pragma unroll
      for (int c = 0; c < 2 * TN; ++c) {  // 16 columns at a time: the column halves of the lanes lr < 16 / lr >= 16
        if ((lr >> 4) == (c & 1)) {
#pragma unroll
          for (int e = 0; e < 16; ++e) stg[mm_acc_row(e, lane) * CLD + (lr & 15)] = fmaf(acc[c >> 1][e], oscale, bv[c >> 1]);
        }
        // same wave writes and reads: LDS operations of a wave execute in order
#pragma unroll
        for (int it = 0; it < 2; ++it) {  // 4 lanes per row (64 B), 16 rows per instruction
          const int r = it * 16 + (lane >> 2), p = lane & 3;
          const f32x4 v = *reinterpret_cast<const f32x4*>(&stg[r * CLD + 4 * p]);
          if (rbase + r < nrows)
            *reinterpret_cast<f32x4*>(&a.Y[(long)(cur.row0 + rbase + r) * a.ldy + n0 + c * SCOLS + 4 * p]) = v;
        }
      }
    }
    if (a.part) {
      // per 32-row block: the lane's 16 values in register order, then the two lane halves; per tile (s0 + s1) + (s2 + s3)
      // over its four blocks - the order of gemm_rows.hip, bit for bit
      float* red = reinterpret_cast<float*>(smem + RED_OFF);  // [8 waves][BN]
      float* colmean = red + 8 * BN;                          // [2 halves][BN]
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        float s1 = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (rbase + mm_acc_row(e, lane) < nrows) s1 += fmaf(acc[tn][e], oscale, bv[tn]);
        s1 = mm_xor32_sum(s1);
        if (lane < 32) red[wave * BN + tn * 32 + lr] = s1;
      }
      gw_lds_barrier();
      for (int x = tid; x < 2 * BN; x += 512) {
        const int hf = x / BN, cl = x - hf * BN;
        const float* rp = red + hf * 4 * BN + cl;
        const float sum = (rp[0] + rp[BN]) + (rp[2 * BN] + rp[3 * BN]);
        const int t2 = 2 * item_u(item) + hf;
        if (t2 < a.T) {
          a.part[((long)t2 * 2 + 0) * a.N + n0 + cl] = sum;
          colmean[x] = sum / (float)a.tile_nrows[t2];
        }
      }
      gw_lds_barrier();
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const float mu = colmean[half * BN + tn * 32 + lr];
        float s2 = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (rbase + mm_acc_row(e, lane) < nrows) {
            const float d = fmaf(acc[tn][e], oscale, bv[tn]) - mu;
            s2 = fmaf(d, d, s2);
          }
        s2 = mm_xor32_sum(s2);
        if (lane < 32) red[wave * BN + tn * 32 + lr] = s2;  // red[] was consumed before the barrier above
      }
      gw_lds_barrier();
      for (int x = tid; x < 2 * BN; x += 512) {
        const int hf = x / BN, cl = x - hf * BN;
        const float* rp = red + hf * 4 * BN + cl;
        const float sum = (rp[0] + rp[BN]) + (rp[2 * BN] + rp[3 * BN]);
        const int t2 = 2 * item_u(item) + hf;
        if (t2 < a.T) a.part[((long)t2 * 2 + 1) * a.N + n0 + cl] = sum;
      }
      gw_lds_barrier();  // the partials and means are rewritten by the next item's epilogue
    }
    cur = nxt;
    scur = snxt;
    wcur = wnxt;
    ub ^= 1;
  }
}

static std::atomic<int> g_gemm_variant{0};
// Test knob: 0 = automatic (the wide kernel when the layer is eligible and fills the chip), 1 = the tile kernel of
// gemm_rows.hip only, 2 = the wide kernel whenever the layer is eligible.  Results do not depend on it (bit for bit).
extern "C" int mmmot_set_gemm_rows_variant(int v) {
  if (v < 0 || v > 2) return MMMOT_EINVAL;
  g_gemm_variant.store(v);
  return MMMOT_OK;
}

template <int TN, int AMODE, int PAIROP>
static int gw_launch(const mmmot_gemm_args* a, hipStream_t s, int n_cu) {
  const int ntn = a->N / (32 * TN);
  const int nu = (a->T + 1) / 2;                     // pairs of row tiles
  const int nitems = ((nu + 7) / 8) * 8 * ntn;       // padded to whole groups of 8 row-tile pairs x ntn column tiles
  int grid = (n_cu / (8 * ntn)) * (8 * ntn);         // whole groups: column tiles of a row-tile pair on one XCD
  if (grid < 8 * ntn) grid = 8 * ntn;
  if (grid > nitems) grid = nitems;
  hipLaunchKernelGGL((gemm_wide_kernel<TN, AMODE, PAIROP>), dim3(grid), dim3(512), 0, s, *a, ntn, nitems);
  return mm_check(hipGetLastError());
}

template <int TN>
static int gw_dispatch(const mmmot_gemm_args* a, hipStream_t s, int n_cu) {
  if (a->amode == MMMOT_A_NORM_RELU) return gw_launch<TN, MMMOT_A_NORM_RELU, 0>(a, s, n_cu);
  switch (a->pairop) {
    case MMMOT_PAIR_MULTIPLY: return gw_launch<TN, MMMOT_A_PAIR, MMMOT_PAIR_MULTIPLY>(a, s, n_cu);
    case MMMOT_PAIR_MINUS_ABS: return gw_launch<TN, MMMOT_A_PAIR, MMMOT_PAIR_MINUS_ABS>(a, s, n_cu);
    default: return gw_launch<TN, MMMOT_A_PAIR, MMMOT_PAIR_MINUS>(a, s, n_cu);
  }
}

// Called by mmmot_gemm_rows after its argument checks.  Returns 1 when the wide kernel took the launch (*status = its
// result), 0 when the layer is left to the tile kernel.
int mmmot_gemm_wide_try(const mmmot_gemm_args* a, hipStream_t s, int* status) {
  const int variant = g_gemm_variant.load();
  if (variant == 1) return 0;
  if (!a->w_hl16 || (a->K != 256 && a->K != 512) || a->N % 128 != 0) return 0;  // (whole 256-value pieces of the uniform rows)
  if (a->amode != MMMOT_A_PAIR && a->amode != MMMOT_A_NORM_RELU) return 0;
  if (a->dbias || a->colsum || a->act != MMMOT_ACT_NONE) return 0;
  if (a->amode == MMMOT_A_PAIR && (a->ldf % 4 != 0 || a->K > a->ldf || !a->pair_uniform32)) return 0;
  if (a->amode == MMMOT_A_NORM_RELU && a->ldsc < a->K) return 0;
  if (a->Y && (a->ldy % 4 != 0)) return 0;
  const int n_cu = mm_num_cu();
  if (n_cu <= 0) return 0;
  const bool wide256 = (a->N % 256 == 0);
  const long items = (long)((a->T + 1) / 2) * (a->N / (wide256 ? 256 : 128));
  // small problems (one reference-shaped frame pair) are latency-bound: more, smaller workgroups finish sooner
  if (variant == 0 && (a->K < 256 || !wide256 || items < n_cu / 2)) return 0;
  *status = wide256 ? gw_dispatch<8>(a, s, n_cu) : gw_dispatch<4>(a, s, n_cu);
  return 1;
}
