// Wide row GEMM for the K = 512 layers of the pairwise block (reference modules/gcn.py:59-82,
// modules/new_end.py:48-52): the stacked [w_new_end.conv0 | w_link.conv1.0] layer over the on-the-fly pair tensor
// (A_PAIR) and the two GroupNorm-fed layers behind it (A_NORM_RELU).  Same contract and arithmetic as the f16 path of
// gemm_rows.hip (3-term fp16 hi/lo split on v_mfma_f32_32x32x16_f16, fp32 accumulate, identical accumulation order,
// identical order of the statistics sums: Y and part are bit for bit the tile kernel's, so a sample scores the same
// whether it runs alone on the tile kernel or in a batch on this one), different data movement: the tile kernel
// regenerates the A tile (op(a_i, b_j) or relu(x*sc+sh), clamp, hi/lo split, ds_write) for every one of the N/128
// column tiles and spends more VALU than MFMA cycles.
//
//   * a wave owns 32 rows x 256 columns (128 accumulator registers): its A fragments are generated IN REGISTERS,
//     straight in the MFMA operand layout (lane = row, 8 consecutive k), from 32-byte loads of the fp32 source rows
//     requested three 16-k steps ahead - no LDS traffic for A, every generated fragment feeds 24 MFMAs;
//   * what is the same for all 32 rows of a wave - the a_i row of a pair tile (the caller guarantees M % 32 == 0, so a
//     32-row block never straddles two i), the per-group scale / shift rows of the GroupNorm prologue - is staged in
//     LDS once per item by LDS-DMA and read as broadcast values;
//   * weights stream through a 3-slot LDS ring of 16-k stages (256 rows x 64 B = 16 KB) by LDS-DMA (global_load_lds,
//     no VGPR staging), requested two steps ahead, XOR-swizzled on the source side (piece ^ ((row >> 2) & 3):
//     conflict-free ds_read_b128 fragments); ONE s_barrier per 16-k step with counted vmcnt;
//   * a workgroup is NH row tiles of 128 rows (template parameter): NH = 1 - four waves, TWO workgroups per CU (<= 80 KB
//     of LDS each) that drift out of phase, one's epilogue and waits under the other's K loop; NH = 2 - eight waves, the
//     two tiles share every weight stage (two thirds of the L2 -> CU bytes per MFMA), one workgroup per CU.  The launcher
//     takes NH = 2 when every workgroup walks a long chain of items (cfg4's 32-pair batches), NH = 1 below that;
//   * the K loop is written out as MFMA slots - one MFMA + one piece of other work (a fragment read, two elements of
//     the next A fragment, a weight request) with sched_barrier pinning the order - and the source-row loads are
//     inline assembly with counted waits: left to itself the compiler sinks every fragment read next to its use (an LDS
//     round trip four to six times per step) and answers a register that waits for a load with vmcnt(0) while LDS-DMA
//     is in flight.  No scratch, 253-256 registers;
//   * persistent workgroups, items chained: the first weight stages, source rows and uniform rows of the NEXT item are
//     requested during the last steps of the current one; the output leaves through LDS as 16-byte stores.
// Epilogue: bias, per-tile per-channel sum / tile-centred M2 (input of mmmot_gn_finalize), store - as gemm_rows.hip.
//
// Round 6 history (profiles/r06/gw_experiments.log): the round-5 form of this kernel (256 x 256 tile, 32-k stages,
// compiler-scheduled, 180-228 B of scratch) ran at 0.37 / 0.33 of the f16x3 ceiling on cfg4's two layers; timing
// experiments (no loads -12 %, no operand arithmetic -13 %, no MFMA -34 %) showed no single limiter but eight waves in the
// same phase at the same time.  This form: 0.41 / 0.405 (4.82 / 2.44 ms at 32 pairs).  What is left: the K loop runs at
// 65-70 % of its MFMA time, the epilogue (6.4 GB of fp32 rows per cfg4 launch through LDS staging + two statistics
// passes) adds 20 % that nothing overlaps - 128 accumulator registers per wave leave no room for a second set.
#include <atomic>
#include <type_traits>

#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

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

#define G2_ROWB 64   // bytes of one weight row per 16-k stage
#define G2_NSLOT 3

// NH = row tiles per workgroup: 1 -> four waves, two workgroups per CU; 2 -> eight waves (two tiles share the weight
// stages: two thirds of the L2 -> CU bytes per MFMA), one workgroup per CU
template <int NH, int AMODE, int PAIROP>
__global__ __launch_bounds__(256 * NH, 2 / NH) void gemm_wide_kernel(mmmot_gemm_args a, int ntn, int nitems) {
  constexpr int NW = 4 * NH;                    // waves
  constexpr int TN = 8, BN = 256;
  constexpr int SLOT = BN * G2_ROWB;            // 16 KB
  constexpr int NDMA = 4 / NH;                  // weight DMA instructions per wave and stage (16 rows x 64 B each)
  constexpr int NRAW = 2;
  constexpr int SCOLS = 16, CLD = SCOLS + 4;
  constexpr int STG_WAVE = 32 * CLD * 4;
  constexpr int RED_OFF = G2_NSLOT * SLOT;      // [4 waves][BN] column partials, [BN] column means
  constexpr int STG_OFF = RED_OFF + 5 * NH * BN * 4;
  constexpr int UNI_OFF = STG_OFF + NW * STG_WAVE;
  constexpr int UNI_BUF = (AMODE == MMMOT_A_PAIR) ? NW * 2048 : NH * 2 * 2048;  // PAIR: a_i per wave; NORM: scale, shift per tile
  constexpr int SMEM = UNI_OFF + 2 * UNI_BUF;
  static_assert(SMEM <= 160 * 1024 / (2 / NH), "LDS budget (NH = 1: two workgroups per CU)");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave >> 2, wq = wave & 3;    // row tile of the workgroup, 32-row block of the tile
  const int lr = lane & 31, h = lane >> 5;
  const int nst = a.K >> 4;                     // 16-k steps
  const long wrow = (long)(a.K >> 2);           // 16-byte pieces per weight row
  const u32x4* Wp = reinterpret_cast<const u32x4*>(a.W);

  // weight DMA: instruction q = wave * NDMA + b covers rows 16 q .. 16 q + 15; lane = (row in 16, LDS slot in row)
  const unsigned dma_off = (unsigned)((lane >> 2) * wrow + ((lane & 3) ^ ((lane >> 4) & 3)));
  // weight fragment of column tile tn: row r = 32 tn + lr, hi piece 2 h -> slot (2 h) ^ ((r >> 2) & 3); lo: ^ 1
  const int boff = lr * G2_ROWB + (((2 * h) ^ ((lr >> 2) & 3)) << 4);

  struct Item { int t, nt, row0, nrows, grp; };
  auto item_u = [&](int it) { return (it & 7) + 8 * (it / (8 * ntn)); };  // group of NH row tiles
  auto decode = [&](int it, Item& I) {
    const int t = NH * item_u(it) + half;
    I.nt = (it >> 3) % ntn;
    const bool ok = t < a.T;
    I.t = ok ? t : a.T - 1;
    I.row0 = a.tile_row0[I.t];
    I.nrows = ok ? a.tile_nrows[I.t] : 0;
    I.grp = a.tile_group ? a.tile_group[I.t] : 0;
  };
  struct Src { const float* p0; const float* u0; const float* u1; float top; };
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
      const int iw = __builtin_amdgcn_readfirstlane(ii);
      S.u0 = a.FA + (long)(a.grp_aoff[I.grp] + iw) * a.ldf;
      S.u1 = nullptr;
    } else {
      S.p0 = a.X + (long)(I.row0 + rr) * a.ldx + 8 * h;
      S.u0 = a.sc + (long)I.grp * a.ldsc;
      S.u1 = a.sh + (long)I.grp * a.ldsc;
    }
  };
  auto load_raw = [&](const float* p0, int n, GwRaw& R) {  // k = 16 n + 8 h .. + 7
    // written-out loads: the compiler's waitcnt pass answers a register that waits for a load with vmcnt(0) once
    // LDS-DMA requests are in flight beside it - here that would drain the source rows requested a few instructions
    // earlier (a full HBM round trip every second step).  The loads are invisible to it; the counted waits in front of
    // the arithmetic that consumes them (gw_wait_vm in step / the prologue) are written out too.  The kernel has no
    // scratch: no register of R is copied or parked between the request and that wait (checked in the ISA).
    const float* p = p0 + 16 * n;
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16"
                 : "=&v"(R.x[0]), "=&v"(R.x[1]) : "v"(p) : "memory");
  };
  auto dma_uniform = [&](const Src& S, int ub) {
    if constexpr (AMODE == MMMOT_A_PAIR) {
      unsigned char* dst = smem + UNI_OFF + ub * UNI_BUF + wave * 2048;
      gw_dma16(reinterpret_cast<const u32x4*>(S.u0) + lane, dst);
      if (a.K > 256) gw_dma16(reinterpret_cast<const u32x4*>(S.u0) + 64 + lane, dst + 1024);
    } else {
      const int v = wq & 1;  // waves 1, 3: the shift row; 0, 2: the scale row (same NUMBER of requests per wave)
      unsigned char* dst = smem + UNI_OFF + ub * UNI_BUF + (half * 2 + v) * 2048;
      const float* src = v ? S.u1 : S.u0;
      gw_dma16(reinterpret_cast<const u32x4*>(src) + lane, dst);
      if (a.K > 256) gw_dma16(reinterpret_cast<const u32x4*>(src) + 64 + lane, dst + 1024);
    }
  };
  auto dma_stage = [&](const u32x4* wbase, int n, int slot) {  // this wave's share of the weights of 16-k step n
#pragma unroll
    for (int b = 0; b < NDMA; ++b) {
      const int q = wave * NDMA + b;
      const unsigned long ubl = (unsigned long)(wbase + ((long)(16 * q) * wrow + 4 * n));
      const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(ubl >> 32));
      const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ubl);
      const u32x4* ub = (const u32x4*)(((unsigned long)hi << 32) | (unsigned long)lo);
      gw_dma16(ub + dma_off, smem + slot * SLOT + q * 1024);
    }
  };

  int item = blockIdx.x;
  if (item >= nitems) return;
  Item cur, nxt;
  Src scur, snxt;
  decode(item, cur);
  sources(cur, scur);
  const u32x4* wcur = Wp + (long)(cur.nt * BN) * wrow;
  GwRaw raw[2];
  f16x8 fh[2], fl[2];
  f16x8 wh[2][2], wl[2][2];  // weight fragments: [buffer][tile of the pair] - groups of two column tiles, one group ahead
  f32x4 uv[2], wv[2];        // wave-uniform operand values of the step being generated (wv: NORM_RELU shift)
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  auto read_w = [&](auto BUFC, auto TC, int sbase, int tile) {  // sbase: boff + slot * SLOT
    constexpr int b = decltype(BUFC)::value, t = decltype(TC)::value;
    const int off = sbase + tile * 32 * G2_ROWB;
    wh[b][t] = *reinterpret_cast<const f16x8*>(smem + off);
    wl[b][t] = *reinterpret_cast<const f16x8*>(smem + (off ^ 16));
  };
  auto read_u = [&](int step, int ubuf, bool second) {  // second: the shift row (NORM_RELU)
    const int k = 16 * step + 8 * h;
    const float* U = reinterpret_cast<const float*>(
        smem + UNI_OFF + ubuf * UNI_BUF + ((AMODE == MMMOT_A_PAIR) ? wave * 2048 : half * 2 * 2048)) + k + (second ? 512 : 0);
    f32x4* dst = second ? wv : uv;
    dst[0] = *reinterpret_cast<const f32x4*>(U);
    dst[1] = *reinterpret_cast<const f32x4*>(U + 4);
  };
  // two elements of the A fragment of the next step: op / normalise, clamp, hi/lo split (the arithmetic of generate())
  auto gen_pair = [&](auto JTC, auto PC, float top) {
    constexpr int jt = decltype(JTC)::value, p = decltype(PC)::value;
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    float y2[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      constexpr int e0 = 2 * p;
      const int e = e0 + q;
      const float x = raw[jt].x[e >> 2][e & 3], u = uv[e >> 2][e & 3];
      float y;
      if constexpr (AMODE == MMMOT_A_PAIR) {
        if constexpr (PAIROP == MMMOT_PAIR_MULTIPLY) y = u * x;
        else if constexpr (PAIROP == MMMOT_PAIR_MINUS_ABS) y = fabsf((u - x) * 0.5f);
        else y = (u - x) * 0.5f;
      } else {
        y = fmaxf(fmaf(x, u, wv[e >> 2][e & 3]), 0.f);
      }
      y2[q] = __builtin_amdgcn_fmed3f(y, -top, top);
    }
    const f16x2 h2 = {(_Float16)y2[0], (_Float16)y2[1]};
    const unsigned hb = __builtin_bit_cast(unsigned, h2);
    const f16x2 l2 = __builtin_bit_cast(f16x2, mm_split_lo2(hb, y2[0], y2[1]));
    fh[jt][2 * p] = h2[0];
    fh[jt][2 * p + 1] = h2[1];
    fl[jt][2 * p] = l2[0];
    fl[jt][2 * p + 1] = l2[1];
  };
  auto dma2 = [&](const u32x4* wbase, int n, int slot, int g) {  // half of this wave's NDMA weight requests of step n
    constexpr int PER = NDMA / 2;
#pragma unroll
    for (int b = 0; b < PER; ++b) {
      const int q = wave * NDMA + g * PER + b;
      const unsigned long ubl = (unsigned long)(wbase + ((long)(16 * q) * wrow + 4 * n));
      const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(ubl >> 32));
      const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ubl);
      const u32x4* ub = (const u32x4*)(((unsigned long)hi << 32) | (unsigned long)lo);
      gw_dma16(ub + dma_off, smem + slot * SLOT + q * 1024);
    }
  };
  int ub = 0;
  {  // first item of this workgroup: load prologue (K >= 64: four steps exist)
    dma_uniform(scur, 0);
    dma_stage(wcur, 0, 0);
    load_raw(scur.p0, 0, raw[0]);
    load_raw(scur.p0, 1, raw[1]);
    dma_stage(wcur, 1, 1);
    gw_wait_vm<NDMA>();  // the uniform rows, stage 0 and the first source values landed; stage 1 may be in flight
    __builtin_amdgcn_s_barrier();
    read_u(0, 0, false);
    if constexpr (AMODE != MMMOT_A_PAIR) read_u(0, 0, true);
    gen_pair(I0{}, I0{}, scur.top);
    gen_pair(I0{}, I1{}, scur.top);
    gen_pair(I0{}, std::integral_constant<int, 2>{}, scur.top);
    gen_pair(I0{}, std::integral_constant<int, 3>{}, scur.top);
    __builtin_amdgcn_sched_barrier(0);
    load_raw(scur.p0, 2, raw[0]);
    read_u(1, 0, false);
    if constexpr (AMODE != MMMOT_A_PAIR) read_u(1, 0, true);
    read_w(I0{}, I0{}, boff, 0);
    read_w(I0{}, I1{}, boff, 1);
  }
  int slot = 0;  // ring slot of the current step
  for (; item < nitems; item += gridDim.x) {
    const int item_n = item + gridDim.x;
    const bool has_next = item_n < nitems;
    decode(has_next ? item_n : item, nxt);
    sources(nxt, snxt);
    const u32x4* wnxt = Wp + (long)(nxt.nt * BN) * wrow;
    dma_uniform(snxt, ub ^ 1);
    f32x16 acc[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[tn][e] = 0.f;

    // One 16-k step = 24 MFMAs in four groups of two column tiles (per accumulator the order lo*hi, hi*lo, hi*hi of the
    // tile kernel), written as ONE MFMA + ONE piece of other work at a time with sched_barrier pinning the order (left to
    // itself the scheduler sinks every fragment read next to its use - a wave then sits out an LDS round trip four to
    // six times per step with nothing but its partner to cover it).  Step n carries, beside its MFMAs:
    //   groups 0..2: the fragment reads of the next group (other buffer, same ring slot);
    //   groups 0, 1: the weight requests of step n + 2 (into the slot step n - 1 has finished with);
    //   group  p   : elements 2 p, 2 p + 1 of the A fragment of step n + 1;
    //   end of group 2: counted wait (weights of step n + 1 landed; the newest NRAW + NDMA requests stay in flight),
    //                barrier (every wave is past its reads of this slot);
    //   group 3: the first fragment reads of step n + 1 (next slot), the uniform operand values of step n + 2, the
    //            source-row loads of step n + 3.
    auto step = [&](auto JC, int nn, int sl, int sl1) {
      constexpr int j = decltype(JC)::value, jt = j ^ 1;
      using JT = std::integral_constant<int, jt>;
      const int bcur = boff + sl * SLOT, bnxt = boff + sl1 * SLOT;
      const bool nx1 = (nn + 1 >= nst);                 // the fragment being generated belongs to the successor
      const float top1 = nx1 ? snxt.top : scur.top;
      auto mma = [&](auto BC, auto TC, int tile, int term) {
        constexpr int b = decltype(BC)::value, t = decltype(TC)::value;
        if (term == 0) acc[tile] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[j], wh[b][t], acc[tile], 0, 0, 0);
        else if (term == 1) acc[tile] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[j], wl[b][t], acc[tile], 0, 0, 0);
        else acc[tile] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[j], wh[b][t], acc[tile], 0, 0, 0);
      };
      auto group = [&](auto GC) {
        constexpr int g = decltype(GC)::value;
        using B = std::integral_constant<int, g & 1>;
        using BN_ = std::integral_constant<int, (g & 1) ^ 1>;
        if constexpr (g == 3) {
          gw_wait_vm<NRAW + NDMA>();
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
        }
        // r = 0
        mma(B{}, I0{}, 2 * g, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (g < 3) read_w(BN_{}, I0{}, bcur, 2 * g + 2);
        else read_w(BN_{}, I0{}, bnxt, 0);
        __builtin_amdgcn_sched_barrier(0);
        // r = 1
        mma(B{}, I1{}, 2 * g + 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (g == 0) {
          // the source rows of step nn + 1 (requested two steps ago) have landed: the weight requests of step nn + 1 and
          // the source rows of step nn + 2 behind them may stay in flight
          gw_wait_vm<NDMA + NRAW>();
          __builtin_amdgcn_sched_barrier(0);
        }
        gen_pair(JT{}, GC, top1);
        __builtin_amdgcn_sched_barrier(0);
        // r = 2
        mma(B{}, I0{}, 2 * g, 1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (g < 3) read_w(BN_{}, I1{}, bcur, 2 * g + 3);
        else read_w(BN_{}, I1{}, bnxt, 1);
        __builtin_amdgcn_sched_barrier(0);
        // r = 3
        mma(B{}, I1{}, 2 * g + 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (g < 2) {
          const int m = nn + 2;
          const bool mx = m >= nst;
          const int sl2 = (sl1 == G2_NSLOT - 1) ? 0 : sl1 + 1;
          dma2(mx ? wnxt : wcur, mx ? m - nst : m, sl2, g);
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (g == 3) {  // uniform operand values of step nn + 2 (generated during step nn + 1)
          const int m = nn + 2;
          const bool mx = m >= nst;
          read_u(mx ? m - nst : m, mx ? (ub ^ 1) : ub, false);
          __builtin_amdgcn_sched_barrier(0);
        }
        // r = 4
        mma(B{}, I0{}, 2 * g, 2);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (g == 3 && AMODE != MMMOT_A_PAIR) {
          const int m = nn + 2;
          const bool mx = m >= nst;
          read_u(mx ? m - nst : m, mx ? (ub ^ 1) : ub, true);
          __builtin_amdgcn_sched_barrier(0);
        }
        // r = 5
        mma(B{}, I1{}, 2 * g + 1, 2);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (g == 3) {  // source rows of step nn + 3 (their buffer's last element has just been consumed)
          const int m = nn + 3;
          const bool mx = m >= nst;
          load_raw(mx ? snxt.p0 : scur.p0, mx ? m - nst : m, raw[jt]);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      group(I0{});
      group(I1{});
      group(std::integral_constant<int, 2>{});
      group(std::integral_constant<int, 3>{});
    };
    for (int n = 0; n < nst; n += 2) {
      const int s1 = (slot == G2_NSLOT - 1) ? 0 : slot + 1;
      const int s2 = (s1 == G2_NSLOT - 1) ? 0 : s1 + 1;
      step(I0{}, n, slot, s1);
      step(I1{}, n + 1, s1, s2);
      slot = s2;
    }

    // ---------------- epilogue (the order of every sum is the tile kernel's: bit-identical Y and part) -------------
    const int n0 = cur.nt * BN;
    const float oscale = a.oscale;
    const int nrows = cur.nrows;
    const int rbase = wq * 32;
    float bv[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) bv[tn] = a.bias ? a.bias[n0 + tn * 32 + lr] : 0.f;
    if (a.Y) {
      float* stg = reinterpret_cast<float*>(smem + STG_OFF + wave * STG_WAVE);  // wave-private
#pragma unroll
      for (int c = 0; c < 2 * TN; ++c) {
        if ((lr >> 4) == (c & 1)) {
#pragma unroll
          for (int e = 0; e < 16; ++e) stg[mm_acc_row(e, lane) * CLD + (lr & 15)] = fmaf(acc[c >> 1][e], oscale, bv[c >> 1]);
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int r = it * 16 + (lane >> 2), p = lane & 3;
          const f32x4 v = *reinterpret_cast<const f32x4*>(&stg[r * CLD + 4 * p]);
          if (rbase + r < nrows)
            *reinterpret_cast<f32x4*>(&a.Y[(long)(cur.row0 + rbase + r) * a.ldy + n0 + c * SCOLS + 4 * p]) = v;
        }
      }
    }
    if (a.part) {
      float* red = reinterpret_cast<float*>(smem + RED_OFF) + half * 5 * BN;  // per tile: [4 waves][BN] partials, [BN] means
      float* colmean = red + 4 * BN;
      const int t2 = NH * item_u(item) + half;
      // (a wave whose 32 rows are all inside the tile - every wave of every tile but a group's last - sums without
      // the per-element row test: same values in the same order)
      const bool wfull = __builtin_amdgcn_readfirstlane((int)(nrows - rbase >= 32)) != 0;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        float s1 = 0.f;
        if (wfull) {
#pragma unroll
          for (int e = 0; e < 16; ++e) s1 += fmaf(acc[tn][e], oscale, bv[tn]);
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e)
            if (rbase + mm_acc_row(e, lane) < nrows) s1 += fmaf(acc[tn][e], oscale, bv[tn]);
        }
        s1 = mm_xor32_sum(s1);
        if (lane < 32) red[wq * BN + tn * 32 + lr] = s1;
      }
      gw_lds_barrier();
      {
        const int cl = tid & 255;  // 256 threads of the tile's four waves, 256 columns
        const float sum = (red[cl] + red[BN + cl]) + (red[2 * BN + cl] + red[3 * BN + cl]);
        if (t2 < a.T) {
          a.part[((long)t2 * 2 + 0) * a.N + n0 + cl] = sum;
          colmean[cl] = sum / (float)a.tile_nrows[t2];
        }
      }
      gw_lds_barrier();
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const float mu = colmean[tn * 32 + lr];
        float s2 = 0.f;
        if (wfull) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const float d = fmaf(acc[tn][e], oscale, bv[tn]) - mu;
            s2 = fmaf(d, d, s2);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e)
            if (rbase + mm_acc_row(e, lane) < nrows) {
              const float d = fmaf(acc[tn][e], oscale, bv[tn]) - mu;
              s2 = fmaf(d, d, s2);
            }
        }
        s2 = mm_xor32_sum(s2);
        if (lane < 32) red[wq * BN + tn * 32 + lr] = s2;
      }
      gw_lds_barrier();
      {
        const int cl = tid & 255;
        const float sum = (red[cl] + red[BN + cl]) + (red[2 * BN + cl] + red[3 * BN + cl]);
        if (t2 < a.T) a.part[((long)t2 * 2 + 1) * a.N + n0 + cl] = sum;
      }
      gw_lds_barrier();
    }
    cur = nxt;
    scur = snxt;
    wcur = wnxt;
    ub ^= 1;
  }
}

static std::atomic<int> g_gemm_variant{0};
// Test knob: 0 = automatic (the wide kernel when the layer is eligible and fills the chip; NH by the length of the item
// chains), 1 = the tile kernel of gemm_rows.hip only, 3 / 4 = the wide kernel whenever the layer is eligible, with NH = 1
// (four waves, two workgroups per CU) / NH = 2 (eight waves).  Results do not depend on it (bit for bit).
extern "C" int mmmot_set_gemm_rows_variant(int v) {
  if (v < 0 || v > 4 || v == 2) return MMMOT_EINVAL;
  g_gemm_variant.store(v);
  return MMMOT_OK;
}

template <int NH, int AMODE, int PAIROP>
static int gw_launch(const mmmot_gemm_args* a, hipStream_t s, int n_cu) {
  const int ntn = a->N / 256;
  const int nu = (a->T + NH - 1) / NH;               // groups of NH row tiles
  const int nitems = ((nu + 7) / 8) * 8 * ntn;       // padded to whole groups of 8 x ntn column tiles
  int grid = ((2 / NH) * n_cu / (8 * ntn)) * (8 * ntn);  // NH = 1: two workgroups per CU; whole groups (column tiles of a row tile on one XCD)
  if (grid < 8 * ntn) grid = 8 * ntn;
  if (grid > nitems) grid = nitems;
  hipLaunchKernelGGL((gemm_wide_kernel<NH, AMODE, PAIROP>), dim3(grid), dim3(256 * NH), 0, s, *a, ntn, nitems);
  return mm_check(hipGetLastError());
}

template <int NH>
static int gw_dispatch(const mmmot_gemm_args* a, hipStream_t s, int n_cu) {
  if (a->amode == MMMOT_A_NORM_RELU) return gw_launch<NH, MMMOT_A_NORM_RELU, 0>(a, s, n_cu);
  switch (a->pairop) {
    case MMMOT_PAIR_MULTIPLY: return gw_launch<NH, MMMOT_A_PAIR, MMMOT_PAIR_MULTIPLY>(a, s, n_cu);
    case MMMOT_PAIR_MINUS_ABS: return gw_launch<NH, MMMOT_A_PAIR, MMMOT_PAIR_MINUS_ABS>(a, s, n_cu);
    default: return gw_launch<NH, MMMOT_A_PAIR, MMMOT_PAIR_MINUS>(a, s, n_cu);
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
  if (a->N % 256 != 0) return 0;  // 256-column items (the N = 128 layer of the block is HBM-bound on the tile kernel)
  const long items = (long)((a->T + 1) / 2) * (a->N / 256);
  // small problems (one reference-shaped frame pair) are latency-bound: more, smaller workgroups finish sooner
  if (variant == 0 && items < n_cu / 2) return 0;
  // eight waves sharing the weight stages win once every workgroup walks a long chain of items (cfg4's 32-pair batches:
  // -4 .. -6 % against the four-wave form, tools/bench_rows_gemm.py); below that the two forms are within the box spread
  // of each other and the four-wave form fills the chip with half as many rows
  const bool nh2 = (variant == 4) || (variant == 0 && items >= 24L * n_cu);
  *status = nh2 ? gw_dispatch<2>(a, s, n_cu) : gw_dispatch<1>(a, s, n_cu);
  return 1;
}
