// hl16 trunk layer with an LDS-resident haloed activation patch ("patch" kernel).
//
// Same arithmetic and storage format as conv3x3_hl16.hip (3-term fp16 hi/lo split on
// v_mfma_f32_32x32x16_f16, fp32 accumulate, hl16 activations and weights).  Different data movement,
// driven by the measurements of round 1 (profiles/README.md): the tile kernels re-read the activation tile
// once per tap through L2 (9x) and are bound by the ~3.5 B/clk/wave L2->CU path.  Here
//   * a workgroup owns 256 output pixels as NB blocks of BS x BS pixels (16x16x1 or 8x8x4) and BN output
//     channels; per 32-channel slab it holds the (BS+2)^2 haloed activation patch of every block in LDS
//     (128 B per pixel: 4 hl16 units) and all 9 taps read their shifted A fragments from it: activation
//     loads per slab drop from 9 x 32 KB to 41 KB (x 0.14), total L2->CU bytes per MFMA by 2.6x;
//   * weights stream through a 3-slot ring of (tap, slab) tiles (BN rows x 128 B), three stages ahead;
//   * everything arrives by LDS-DMA (global_load_lds, no VGPR staging, no ds_write): the unpadded
//     lane-linear LDS image is made conflict-free for ds_read_b128 by XOR swizzles applied on the SOURCE
//     side and on the fragment read - weight row r: piece ^ ((r >> 1) & 7); patch pixel (py, px):
//     piece ^ (((px >> 1) + 4 * (py & 1)) & 7), which is conflict-free for every tap shift because the
//     16 lanes of a ds_read_b128 group are 4 pooling quads = 2 pixel rows x 8 distinct px (mod 16);
//   * 8 waves, each wave loads AND computes; fragments are double-buffered in registers (the second
//     16-channel step of a stage is read under the MFMAs of the first, the first step of the next stage
//     under the MFMAs of the second), ONE s_barrier per stage with counted vmcnt.
// Rows of the implicit GEMM are in quad order (4 consecutive rows = one 2x2 pooling window), so max-pool
// stays an in-register/LDS-local max in the epilogue.
#include <atomic>
#include <type_traits>

#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

static __device__ u32x4 pt_zero_page[16];  // zero-initialised: source of out-of-image patch pixels

#define P_BM 256
#define P_ROWB 128  // bytes per LDS record (pixel or weight row): 32 channels hi+lo

__device__ __forceinline__ void pt_split8(f32x8 v, u32x4& hi, u32x4& lo) {
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

// hq8 encode of 16 consecutive channels: fp16 hi (two 16-byte pieces), e4m3(a * 2^-2), e4m3((a - hi) * 2^9)
__device__ __forceinline__ void pt_encode_q8(const float (&v)[16], u32x4& hi0, u32x4& hi1, u32x4& a8, u32x4& l8) {
  f16x8 h0, h1;
  float lo[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const float x = fminf(fmaxf(v[e], -65000.f), 65000.f);
    const _Float16 hh = (_Float16)x;
    if (e < 8) h0[e] = hh; else h1[e - 8] = hh;
    lo[e] = __builtin_amdgcn_fmed3f((x - (float)hh) * 512.f, -448.f, 448.f);  // saturates for |x| > 1792 like the a / 4 copy
  }
  hi0 = __builtin_bit_cast(u32x4, h0);
  hi1 = __builtin_bit_cast(u32x4, h1);
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    int pa = 0, pl = 0;
    const float a0 = __builtin_amdgcn_fmed3f(v[4 * w + 0] * 0.25f, -448.f, 448.f);
    const float a1 = __builtin_amdgcn_fmed3f(v[4 * w + 1] * 0.25f, -448.f, 448.f);
    const float a2 = __builtin_amdgcn_fmed3f(v[4 * w + 2] * 0.25f, -448.f, 448.f);
    const float a3 = __builtin_amdgcn_fmed3f(v[4 * w + 3] * 0.25f, -448.f, 448.f);
    pa = __builtin_amdgcn_cvt_pk_fp8_f32(a0, a1, pa, false);
    pa = __builtin_amdgcn_cvt_pk_fp8_f32(a2, a3, pa, true);
    pl = __builtin_amdgcn_cvt_pk_fp8_f32(lo[4 * w + 0], lo[4 * w + 1], pl, false);
    pl = __builtin_amdgcn_cvt_pk_fp8_f32(lo[4 * w + 2], lo[4 * w + 3], pl, true);
    a8[w] = (unsigned)pa;
    l8[w] = (unsigned)pl;
  }
}

// Range guard of the reduced-range activation formats (read by mmmot_trunk_range_read): [0] activation elements
// written with |a| > 1792 in hq8 mode (the e4m3(a/4) and e4m3(512 a_lo) copies saturate: fp16-class products for that
// element), [1] elements clamped at the fp16 range limit 65000 (either mode: wrong value), [2] the same two events for
// conv1_1 outputs inside the fused first launch (counted once per patch pixel, halo pixels included).
// The common path costs a running maximum (v_max3) and one compare per 16 values; the atomics run only on a hit.
// Every launch gets the counter block as a kernel argument: the caller's own (mmmot_trunk_range_bind - one block per
// engine, so two models on one device, or a captured graph next to an eager forward, never mix their windows) or,
// unbound, this per-device block of the library.
__device__ unsigned int pt_range[4];
#define PT_SAT_E4M3 1792.f
#define PT_SAT_FP16 65000.f

template <int N>
__device__ __forceinline__ void pt_range_guard(const float* v, bool q8, unsigned int* pt_range) {
  float mx = v[0];
#pragma unroll
  for (int e = 1; e < N; ++e) mx = fmaxf(mx, v[e]);  // post-ReLU values: >= 0
  if (mx > (q8 ? PT_SAT_E4M3 : PT_SAT_FP16)) {
    unsigned c = 0, d = 0;
#pragma unroll
    for (int e = 0; e < N; ++e) {
      c += v[e] > PT_SAT_E4M3;
      d += v[e] > PT_SAT_FP16;
    }
    if (q8) atomicAdd(&pt_range[0], c);
    if (d) atomicAdd(&pt_range[1], d);
  }
}

// EXP == 9: phase timestamps of thread 0 (shader clock), summed over items: [0] decode..loads issued, [1] prologue
// wait, [2] K loop, [3] accumulators -> LDS, [4] encode + stores issued, [5] closing barrier, [7] items
__device__ unsigned long long pt_dbg[8];
#define PT_STAMP(i)                                                                  \
  if constexpr (EXP == 9 || EXP == 10) {                                             \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();                    \
    if (threadIdx.x == 0 && (i) >= 0) atomicAdd(&pt_dbg[(i) < 0 ? 0 : (i)], now_ - tprev_); \
    tprev_ = now_;                                                                   \
  }

__device__ __forceinline__ int pt_swz_b(int r) { return (r >> 1) & 7; }
__device__ __forceinline__ int pt_swz_a(int py, int px) { return ((px >> 1) + 4 * (py & 1)) & 7; }

template <int N>
__device__ __forceinline__ void pt_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// workgroup barrier that orders LDS accesses only (no vmcnt drain)
__device__ __forceinline__ void pt_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ void pt_dma16(const u32x4* src, unsigned char* dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

template <int BS>
struct PatchGeom {
  // WHOLE (BS <= 4): a block is a WHOLE feature map of at most BS x BS pixels (conv5 at 64-pixel crops: 4 x 4), so every
  // halo pixel lies outside the image.  The patch is stored WITHOUT halo (256 pixel records = 32 KB per slab instead of
  // the 73 KB sixteen haloed 6 x 6 patches would take) and a tap that leaves the block reads a zero page in LDS:
  // sixteen 4 x 4 maps fill a 256-row tile completely, where the 8 x 8 geometry runs them at 25 % fill.
  static constexpr bool WHOLE = (BS <= 4);
  static constexpr int NB = P_BM / (BS * BS);     // blocks per workgroup tile: 1 / 4 / 16
  static constexpr int PW = WHOLE ? BS : BS + 2;  // patch row length (pixels)
  static constexpr int PP = PW * PW;              // patch pixels per block
  static constexpr int NCH = NB * PP * 8;         // 16-byte pieces per slab patch: 2592 / 3200 / 2048
  static constexpr int FULL = NCH / 512;          // rounds in which all 8 waves move 1 KB each: 5 / 6 / 4
  static constexpr int REM = NCH - FULL * 512;    // pieces of the last, partial round: 32 / 128 / 0
  static constexpr int RL = REM / 8;              // active lanes per wave in the partial round: 4 / 16
  static constexpr int PA = FULL + (REM ? 1 : 0); // DMA rounds per slab patch: 6 / 7 / 4
  // bytes between the patch buffers: the patch itself, or (WHOLE) the 64 x (128 + 4) fp32 rows of an epilogue chunk
  static constexpr int BYTES = WHOLE ? 34 * 1024 : NCH * 16;  // 41 472 / 51 200 / 34 816
  static constexpr int NTB = WHOLE ? 1 : NB;      // per-block origins kept by tile_blocks (WHOLE: only the first crop)
  static_assert(REM % 8 == 0 && PA <= 7, "patch rounds must fit taps 0..6 of the previous slab");
  static_assert(NCH * 16 <= BYTES, "patch buffer");
};

// patch swizzle of the WHOLE geometry: the 16 lanes of a ds_read_b128 group are, per 32-row M-tile, two diagonal
// quads of each of its two blocks = 4 consecutive rows (mod 4 distinct under every tap shift) x 2 blocks x 2 column
// parities (= record parity, the 128-byte half of the 256-byte bank row)
template <int BS>
__device__ __forceinline__ int pt_swz_w(int blk, int py) {
  static_assert(BS == 4, "whole-map geometry: 4 x 4 blocks");
  return ((py & 3) + 4 * (blk & 1)) & 7;
}

// decode row i (0..31) of M-tile g (0..7) of the workgroup tile -> block, pixel inside the block
template <int BS>
__device__ __forceinline__ void pt_row_to_pixel(int g, int i, int& blk, int& y, int& x) {
  const int q = i >> 2;
  if constexpr (BS == 16) {
    blk = 0;
    y = 2 * g + ((i >> 1) & 1);
    x = 2 * q + (i & 1);
  } else if constexpr (BS == 8) {
    blk = g >> 1;
    y = 2 * (2 * (g & 1) + (q >> 2)) + ((i >> 1) & 1);
    x = 2 * (q & 3) + (i & 1);
  } else {  // 4 x 4: two blocks per M-tile, four quads each
    blk = 2 * g + (q >> 2);
    y = 2 * ((q >> 1) & 1) + ((i >> 1) & 1);
    x = 2 * (q & 1) + (i & 1);
  }
}

// EXP: timing experiments only (WRONG results unless 0): 1 = no loads in the K loop, 2 = no barriers in
// the K loop, 3 = no fragment reads in the K loop, 4 = no MFMAs, 5 = no epilogue at all, 6 = epilogue without
// the global stores
// FUSE1: the layer's input is not read from memory but COMPUTED: conv1_1 (3 -> 64, folded BN, ReLU) of the
// raw crops is evaluated for the 18x18 haloed patch of every tile in the prologue (a 336 x 64 x 32 mini-GEMM on
// the matrix cores: K = 27 taps*colours padded to 32) and written straight into the two LDS patch buffers
// (Cin = 64 = both 32-channel slabs).  The [L][H][W][64] conv1_1 tensor (537 MB per cfg3 pair: written once,
// read 1.3x) never exists.  Requires BN = 64, BS = 16, Cin = Cout = 64.
struct Fuse1Args {
  const float* raw;    // crops, NCHW fp32 [L][3][H][W] (normalised: the reference's `dets`), or
  const u32x4* w1;     // conv1_1 weights, hl16 [64][32] (k = tap*3 + colour, zero for k >= 27), scaled by 2^shift
  const float* bias1;  // [64] folded BN bias
  float oscale1;       // 2^-shift
  // raw8 != nullptr: the crops arrive as the 8-bit RGB images of the resize, [L][H][W][3] (what PIL hands to
  // torchvision), and ToTensor / Normalize (x / 255, (x - mean) / std: IEEE divisions, utils/build_util.py:111-112) are
  // applied while the raw window is fetched - the fp32 crop tensor (77 MB per 128 detections at 224 x 224) never exists
  const unsigned char* raw8;
  float mean[3], stdv[3];
};

// Q8 ("hq8" arithmetic and storage): the two CORRECTION terms of the hi/lo split run on the fp8 matrix cores.
// A 32-channel record keeps its 128 bytes but holds [32 x fp16 hi | 32 x e4m3(a * 2^-2) | 32 x e4m3(a_lo * 2^9)]
// (weights: [fp16 w_hi | e4m3(w_lo * 2^5) | e4m3(w_hi * 2^-6)]); per stage and product: two f16 MFMAs (hi*hi,
// K = 2 x 16) + ONE v_mfma_scale_f32_32x32x64_f8f6f4 whose 64 k-slots are [a8 . w_lo8 | a_lo8 . w8] with the
// block scale 2^-3 - 2 instead of 3 f16-MFMA-equivalents per product.  Pieces stay 16 bytes, so loaders, DMA ring
// and swizzles are unchanged.  Accuracy: tools/study_fp8_correction.py.
typedef int i32x8 __attribute__((ext_vector_type(8)));
#define Q8_ASHIFT 2      // fp8 copies of activations carry 2^-2 (range up to 1792)
#define Q8_SCALE_A 124   // E8M0 exponent of the block scale 2^-3 = 2^-11 (lo) * 2^2 (activation copies) * 2^6 (weight copies)
#define Q8_SCALE_B 127

// RAW (training, mmmot_conv3x3_raw_hl16): the output is the plain convolution  acc * oscale + bias  as fp32 rows (8 floats
// where an inference launch writes one [hi8 | lo8] unit: the same bytes at the same place) - no ReLU, no clamp, no split.
template <int BN, int BS, bool POOL, int EXP, bool FUSE1 = false, bool Q8 = false, bool RAW = false>
__global__ __launch_bounds__(512) void conv3x3_hl16_patch_kernel(
    const u32x4* __restrict__ in, const u32x4* __restrict__ wp, const float* __restrict__ bias,
    u32x4* __restrict__ out, int L, int H, int W, int Cin, int Cout, int nby, int nbx, int nblk, int ntm, int ntn,
    const float* __restrict__ oscv, Fuse1Args fz, unsigned int* __restrict__ rng) {
  static_assert(!FUSE1 || (BN == 64 && BS == 16), "the fused first layer exists for 64-channel 16x16 tiles");
  static_assert(!RAW || (!POOL && !FUSE1 && !Q8), "raw fp32 output: plain unpooled f16x3 layers");
  using G = PatchGeom<BS>;
  constexpr int WN = (BN == 128) ? 2 : 1;  // waves along channels
  constexpr int WM = 8 / WN;               // waves along pixels
  constexpr int TM = P_BM / (WM * 32);     // 2 / 1
  constexpr int TN = BN / (WN * 32);       // 2
  constexpr int B_BYTES = BN * P_ROWB;     // 16 / 8 KB per (tap, slab) weight tile
  constexpr int NBL = BN / 64;             // weight DMA instructions per wave per stage: 2 / 1
  constexpr int CLD = BN + 4;
  // LDS map: [patch buffer 0 | patch buffer 1 | weight ring (3 slots) | FUSE1: raw window].  The K loop is ONE
  // stream across the tiles of a workgroup: during the last slab of a tile the first patch slab of the NEXT tile
  // is fetched into the free patch buffer and the ring continues with the next tile's weight stages 0..2
  // ("transition" slab), so a tile has no load prologue.  (A CU pulls only ~5-8 B/clk from HBM - latency x
  // outstanding misses - so the 41 KB first slab plus three weight stages cost 15-20 k cycles when nothing runs
  // beside them: tools/patch_phase_timers.py.)  The epilogue therefore stages the accumulators through the
  // patch buffer the K loop has just finished with, in NCH row chunks, and leaves the other buffer and the ring
  // alone.  FUSE1 computes its patches (both buffers live all along): no transition, staging from offset 0.
  constexpr int P0_OFF = 0;
  constexpr int P1_OFF = G::BYTES;
  constexpr int RING0 = 2 * G::BYTES;
  constexpr int LOOP_BYTES = RING0 + 3 * B_BYTES;
  constexpr int RAW_OFF = LOOP_BYTES;                      // FUSE1: raw input window [3][20][20] fp32
  constexpr int RAW_BYTES = FUSE1 ? 3 * 20 * 20 * 4 + 64 : 0;
  constexpr bool OVL = FUSE1 && !Q8;                       // conv1_1 of the next tile under this tile's K loop (below)
  // pooled layers take the 2x2 maximum in registers (the four pixels of a window are four consecutive accumulator
  // registers of one lane: quad-ordered rows) and stage the 64 pooled rows only - one chunk
  constexpr int SROWS = POOL ? P_BM / 4 : P_BM;            // staged rows per tile
  constexpr int NCH = (FUSE1 || POOL) ? 1 : (SROWS / 2 * CLD * 4 <= G::BYTES) ? 2 : 4;
  constexpr int RCH = SROWS / NCH;                         // staged rows per epilogue chunk
  static_assert((FUSE1 && !OVL) ? (SROWS * CLD * 4 <= LOOP_BYTES) : (RCH * CLD * 4 <= G::BYTES), "epilogue staging does not fit");
  // STREAM: tiles are chained through transition slabs.  The successor's patch source table is made during the
  // tile's own prologue (low register pressure) and parked in LDS: [PA][512] words behind the ring.  The 8x8-block
  // variants have no room for it: they chain only onto a successor with the same pixel tile (same table).
  constexpr bool STREAM = !FUSE1 && (G::NB == 1 || G::WHOLE);
  // OVL (the f16x3 fused first layer): the conv1_1 patches of the NEXT tile are computed underneath the K loop of the
  // current one.  A third patch buffer makes that possible: a tile reads (X, Y) = its two 32-channel slabs; while its
  // second slab runs, the successor's first slab is written to Z and its second one to X (dead by then); the
  // epilogue stages through Y; the successor then works on (Z, X) with Y as its third buffer.  The prologue's VALU
  // work (conversions, ReLU, hi/lo split: ~150 instructions per 16-pixel round and wave, 3 rounds per wave and tile)
  // is cut into MFMA-slot-sized slices and issued in the shadow of the main loop's matrix instructions.
  constexpr int P2_OFF = (LOOP_BYTES + RAW_BYTES + 255) / 256 * 256;
  constexpr int POFF_OFF = OVL ? P2_OFF + G::BYTES : LOOP_BYTES + RAW_BYTES;
  // FUSE1: conv1_1 weights as MFMA A fragments [mb][lane][hi | lo] (8 KB) - read per use, not held in 32 registers
  constexpr int W1_OFF = POFF_OFF;
  // WHOLE: 256 zero bytes (one bank row) that stand in for every pixel outside the block
  constexpr int ZOFF = POFF_OFF + (STREAM ? G::PA * 512 * 4 : 0) + (FUSE1 ? 8192 : 0);
  static_assert(ZOFF % 256 == 0, "zero page: one aligned bank row");
  constexpr int SMEM = ZOFF + (G::WHOLE ? 256 : 0);
  static_assert(SMEM <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: LDS-DMA destinations stay in SGPRs
  if constexpr (G::WHOLE) {  // ordered before the first fragment read by the prologue barrier of the first tile
    if (tid < 16) reinterpret_cast<u32x4*>(smem + ZOFF)[tid] = u32x4{0u, 0u, 0u, 0u};
  }

  // Persistent workgroups: gridDim.x (a multiple of 8, at most one workgroup per CU - LDS allows no more)
  // workgroups walk the nitems = ntm * ntn tiles.  The dispatcher places workgroup b on XCD b % 8; every XCD
  // owns a contiguous chunk of the logical tile order (channel-tile-major inside the chunk, so the weights
  // of a channel tile stay in that XCD's L2) and its workgroups take the chunk's tiles round-robin.  One
  // launch-time workgroup per tile would leave the CU idle for a dispatch latency between tiles: with a
  // single resident workgroup nothing overlaps it (measured: fixed cost of ~2 channel slabs per tile).
  // FUSE1: conv1_1 weight fragments (A operand of v_mfma_f32_16x16x32_f16: channel 16 mb + (lane & 15),
  // k = 8 (lane >> 4) .. + 7 - the whole K = 32 in one step) are staged in LDS once per workgroup ([W1_OFF]; read per
  // use by the serial form; the overlapped form (OVL) keeps them in 32 registers - the K loop of this layer is bound by
  // LDS reads, one per MFMA).  The folded bias / 2^-shift rides in the k = 27 slot of the K = 32 step (27 taps x colours
  // padded to 32): its "window value" is the constant 1, so the matrix cores add it - no fma and no bias read in the
  // epilogue of a round; hi + lo keep 22 bits of it like of every weight (|bias| / 2^-shift must stay below 65504:
  // pack.conv1_weight_shift).  The raw-window offsets of the 8 k values a lane gathers are fixed (k >= 28: offset 0 -
  // the weight is zero and the window value finite, so the gather needs no branch)
  int roff[8];
  f16x8 w1h[4], w1l[4];  // OVL only
  if constexpr (FUSE1) {
    const int kg1 = (threadIdx.x & 63) >> 4;
    {
      const int t = threadIdx.x;
      const int mb = t >> 7, ln = (t >> 1) & 63, hl = t & 1;
      u32x4 wv = fz.w1[((mb * 16 + (ln & 15)) * 4 + (ln >> 4)) * 2 + hl];
      if ((ln >> 4) == 3) {  // k = 24 .. 31 of this channel: element 3 = k 27 takes the bias (the window value there is 1)
        const float bs = fz.bias1[mb * 16 + (ln & 15)] * (1.f / fz.oscale1);  // oscale1 is a power of two
        const _Float16 bh = (_Float16)bs;
        const _Float16 bl = (_Float16)(bs - (float)bh);
        const unsigned bits = (unsigned)__builtin_bit_cast(unsigned short, hl == 0 ? bh : bl);
        wv[1] = (wv[1] & 0x0000ffffu) | (bits << 16);
      }
      reinterpret_cast<u32x4*>(smem + W1_OFF)[t] = wv;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 8 * kg1 + e;
      const int tap = k / 3, c = k - 3 * tap;
      roff[e] = (k < 27) ? c * 400 + (tap / 3) * 20 + (tap % 3) : 0;
    }
  }
  // EXP 15 (correct results): static priority for the later-dispatched half of the workgroup (the arbitration loser
  // of every stage: microarchitecture guide, "two waves per SIMD", item 4)
  if constexpr (EXP == 15) {
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
  }
  const int nitems = ntm * ntn;
  const int xq = nitems >> 3, xr = nitems & 7;
  const int xcd = blockIdx.x & 7;
  const int cbase = (xcd < xr) ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq;
  const int clen = (xcd < xr) ? xq + 1 : xq;
  auto decode_item = [&](int item, int& mt, int& nt) {
    if (clen % ntn == 0 && cbase % ntn == 0) {
      const int mcount = clen / ntn;
      nt = item / mcount;
      mt = cbase / ntn + item % mcount;
    } else {
      mt = (cbase + item) / ntn;
      nt = (cbase + item) % ntn;
    }
  };
  // FUSE1: the raw input window of the NEXT tile is fetched into registers while the current tile's
  // epilogue runs (nothing else hides that round trip: one workgroup per CU)
  float rawv[3] = {0.f, 0.f, 0.f};
  bool raw_ready = false;
  auto fetch_raw = [&](int mtile) {
    const int nbpc_ = nby * nbx;
    const int crop = mtile / nbpc_, br = mtile - crop * nbpc_;
    const int by = br / nbx, bx = br - by * nbx;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int i = threadIdx.x + 512 * k;
      const int c = i / 400, rem = i - c * 400;
      const int wy = rem / 20, wx = rem - wy * 20;
      const int gy = by * 16 - 2 + wy, gx = bx * 16 - 2 + wx;
      // every wave issues exactly one load per k (invalid lanes read element 0 and select 0): the counted vmcnt waits
      // of the overlapped prologue rely on the number of loads in flight
      const bool ok = i < 1200 && mtile < nblk && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      float v;
      if (fz.raw8) {
        const long idx = ok ? (((long)crop * H + gy) * W + gx) * 3 + c : 0;
        const float x = __fdiv_rn((float)fz.raw8[idx], 255.f);                                          // ToTensor
        v = __fdiv_rn(__fsub_rn(x, fz.mean[ok ? c : 0]), fz.stdv[ok ? c : 0]);                          // Normalize
      } else {
        const long idx = ok ? (((long)crop * 3 + c) * H + gy) * W + gx : 0;
        v = fz.raw[idx];
      }
      if (!ok) v = 0.f;
      rawv[k] = v;
    }
  };
  // patch source offsets of one tile (16-byte units, without the slab term; ~0u = zero page).  Round k moves pieces
  // [k*512 + wave*64, +64) (k < FULL) or, in the partial round, [FULL*512 + wave*RL, +RL); piece p = (patch pixel
  // n = p >> 3, LDS slot c' = p & 7) holds the logical piece c' ^ swz_a(py, px) of that pixel's 128-byte slab
  // record.  The block -> (crop, by, bx) divisions are wave-uniform (scalar unit); the table of the NEXT tile is
  // computed while the slower waves still issue their stores, ahead of the closing barrier of the tile.
  unsigned poff[G::PA];
  struct TileBlocks {
    int crop[G::NTB], gy0[G::NTB], gx0[G::NTB];  // wave-uniform: crop (-1: no such block) and patch origin of each block
  };
  auto tile_blocks = [&](int mtile, TileBlocks& tb) {
    if constexpr (G::WHOLE) {  // block q of the tile = crop mtile * NB + q, origin (0, 0), no halo
      tb.crop[0] = mtile * G::NB;
      tb.gy0[0] = 0;
      tb.gx0[0] = 0;
      return;
    }
    const int nbpc_ = nby * nbx;
#pragma unroll
    for (int q = 0; q < G::NTB; ++q) {
      const int b = mtile * G::NB + q;
      const int crop = b / nbpc_, br = b - crop * nbpc_;
      const int by = br / nbx;
      tb.crop[q] = (b < nblk) ? crop : -1;
      tb.gy0[q] = by * BS - 1;
      tb.gx0[q] = (br - by * nbx) * BS - 1;
    }
  };
  auto poff_round = [&](auto KC, const TileBlocks& tb) -> unsigned {
    constexpr int k = decltype(KC)::value;
    const int cin16 = (Cin >> 3) * 2;
    // the lane id is laundered: otherwise the lane-only part of every round (pixel, swizzle) is hoisted out of
    // the tile loop and stays live across the K loop (+20 VGPRs at its register peak)
    int lane_ = lane;
    asm volatile("" : "+v"(lane_));
    const int p = (k < G::FULL) ? (k * 8 + wave) * 64 + lane_ : G::FULL * 512 + wave * G::RL + lane_;
    unsigned off = ~0u;
    if (k < G::FULL || lane_ < G::RL) {
      const int n = p >> 3, c = p & 7;
      const int blk = n / G::PP;
      const int rem = n - blk * G::PP;
      const int py = rem / G::PW, px = rem - py * G::PW;
      int crop = tb.crop[0], gy = tb.gy0[0] + py, gx = tb.gx0[0] + px;
      if constexpr (G::WHOLE) {
        crop += blk;
        if (crop >= nblk) crop = -1;
      }
#pragma unroll
      for (int q = 1; q < G::NTB; ++q)
        if (blk == q) {
          crop = tb.crop[q];
          gy = tb.gy0[q] + py;
          gx = tb.gx0[q] + px;
        }
      int swz;
      if constexpr (G::WHOLE) swz = pt_swz_w<BS>(blk, py);
      else swz = pt_swz_a(py, px);
      if (crop >= 0 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
        off = (unsigned)(((crop * H + gy) * W + gx) * cin16 + (c ^ swz));
    }
    return off;
  };
  auto compute_poff = [&](int mtile) {
    TileBlocks tb;
    tile_blocks(mtile, tb);
    poff[0] = poff_round(std::integral_constant<int, 0>{}, tb);
    poff[1] = poff_round(std::integral_constant<int, 1>{}, tb);
    poff[2] = poff_round(std::integral_constant<int, 2>{}, tb);
    poff[3] = poff_round(std::integral_constant<int, 3>{}, tb);
    if constexpr (G::PA > 4) poff[4] = poff_round(std::integral_constant<int, 4>{}, tb);
    if constexpr (G::PA > 5) poff[5] = poff_round(std::integral_constant<int, 5>{}, tb);
    if constexpr (G::PA > 6) poff[6] = poff_round(std::integral_constant<int, 6>{}, tb);
  };
  int mt = 0, nt = 0;
  if ((int)(blockIdx.x >> 3) < clen) {
    decode_item(blockIdx.x >> 3, mt, nt);
    if constexpr (!FUSE1) compute_poff(mt);
  }
  const u32x4* wbase = wp + (long)(nt * BN) * ((Cin >> 3) * 2);  // weight rows of this tile's channel tile
  const u32x4* wbase_next = wbase;
  bool primed = false;  // this tile's patch slab 0 and weight stages 0..2 were streamed in by the previous tile
  int pcur = P0_OFF;    // patch buffer of the current slab (byte offset in smem)
  int pnext = P1_OFF;   // patch buffer being filled for the next slab (of this tile or the next one)
  int bufX = P0_OFF, bufY = P1_OFF, bufZ = OVL ? P2_OFF : 0;  // OVL: this tile's two slabs and the free buffer
  for (int item = blockIdx.x >> 3; item < clen; item += gridDim.x >> 3) {
  unsigned long long tprev_ = 0;
  PT_STAMP(-1)
  const int n0 = nt * BN;
  const int cin8 = Cin >> 3;
  const int nslab = Cin >> 5;
  const int nbpc = nby * nbx;  // blocks per crop

  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[tm][tn][e] = 0.f;
  const int wm = wave / WN, wn = wave % WN;
  const int lr = lane & 31;
  const int h = lane >> 5;

  // ---------------- loader state ---------------------------------------------------------------------
  // patch: round k moves pieces [k*512 + wave*64, +64) (k < FULL) or, in the partial round,
  // [FULL*512 + wave*RL, +RL); piece p = (patch pixel n = p >> 3, LDS slot c' = p & 7) holds the
  // logical piece c' ^ swz_a(py, px) of that pixel's 128-byte slab record.
  const u32x4* zsrc = pt_zero_page + (lane & 15);
  // weights: instruction (wave * NBL + b) covers 8 rows of the tile; lane = (row in 8, slot in row)
  const int rsub = lane >> 3, slot8 = lane & 7;
  unsigned boffl[NBL];  // lane part of the weight source offset (16-byte units); the rest is wave-uniform
#pragma unroll
  for (int b = 0; b < NBL; ++b) {
    const int brw = (wave * NBL + b) * 8 + rsub;
    boffl[b] = (unsigned)(brw * (cin8 * 2) + (slot8 ^ pt_swz_b(brw)));
  }
  const long tapstride_w = (long)Cout * cin8 * 2;

  auto issue_patch_round = [&](auto KC, int slab, int pbuf) {  // pbuf: byte offset of the patch buffer in smem
    constexpr int k = decltype(KC)::value;
    const u32x4* src = (poff[k] != ~0u) ? in + (poff[k] + (unsigned)(slab * 8)) : zsrc;
    if constexpr (k < G::FULL) {
      pt_dma16(src, smem + pbuf + (k * 8 + wave) * 1024);
    } else {
      if (lane < G::RL) pt_dma16(src, smem + pbuf + G::FULL * 8192 + wave * (G::RL * 16));
    }
  };
  // weight DMA instruction b (0..NBL-1) of stage (tap, slab) into ring slot `ringslot`
  auto issue_b1 = [&](auto BC, const u32x4* base, int tap, int slab, int ringslot) {
    constexpr int b = decltype(BC)::value;
    if constexpr (b < NBL) {
      // wave-uniform base, laundered through readfirstlane so that it stays in scalar registers and is
      // not re-associated with the lane offsets into nine hoisted 64-bit lane values
      const unsigned long ubl = (unsigned long)(base + ((long)tap * tapstride_w + slab * 8));
      const unsigned ub_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(ubl >> 32));
      const unsigned ub_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ubl);
      const u32x4* ub = (const u32x4*)(((unsigned long)ub_hi << 32) | (unsigned long)ub_lo);
      pt_dma16(ub + boffl[b], smem + RING0 + ringslot * B_BYTES + (wave * NBL + b) * 1024);
    }
  };
  auto issue_b = [&](int tap, int slab, int ringslot) {
    issue_b1(std::integral_constant<int, 0>{}, wbase, tap, slab, ringslot);
    issue_b1(std::integral_constant<int, 1>{}, wbase, tap, slab, ringslot);
  };
  // first tile of a workgroup: patch slab 0 -> buffer pbuf, weight stages 0..2 (poff / wbase are current)
  auto issue_tile_head = [&](int pbuf) {
    issue_patch_round(std::integral_constant<int, 0>{}, 0, pbuf);
    issue_patch_round(std::integral_constant<int, 1>{}, 0, pbuf);
    issue_patch_round(std::integral_constant<int, 2>{}, 0, pbuf);
    issue_patch_round(std::integral_constant<int, 3>{}, 0, pbuf);
    if constexpr (G::PA > 4) issue_patch_round(std::integral_constant<int, 4>{}, 0, pbuf);
    if constexpr (G::PA > 5) issue_patch_round(std::integral_constant<int, 5>{}, 0, pbuf);
    if constexpr (G::PA > 6) issue_patch_round(std::integral_constant<int, 6>{}, 0, pbuf);
    issue_b(0, 0, 0);
    issue_b(1, 0, 1);
    issue_b(2, 0, 2);
  };

  // ---------------- fragment read state --------------------------------------------------------------
  // A (patch) read of tap (ty, tx), k-step j, lane half h: pixel n = n0 + ty*PW + tx, piece
  // (4j + 2h) ^ swz(py + ty, px + tx).  swz(.., ty odd) = swz(.., ty even) ^ 4, so per row three lane
  // values a_sx[tx] = ((swz(py, px + tx) ^ 2h) << 4) cover every tap: byte offset inside the record =
  // a_sx[tx] ^ (64 * (j ^ (ty & 1))); the tap's pixel offset is an immediate.
  // WHOLE: no halo in LDS.  The swizzle depends on the row only: a_sx[tm][ty] = ((swz(blk, y + ty - 1) ^ 2h) << 4); the
  // tap's record is a_base + ((ty - 1) * BS + tx - 1) * 128 when the source pixel lies inside the block (bit `tap` of
  // a_ok) and otherwise the zero page, at the slot the pixel WOULD occupy (a_zb: its record parity for tx = 1, flipped
  // for tx = 0 / 2), so that the sixteen lanes of a read group stay on sixteen different slots.
  int a_base[TM], a_sx[TM][3];
  int a_ok[TM], a_zb[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    int blk, y, x;
    pt_row_to_pixel<BS>(wm * TM + tm, lr, blk, y, x);
    a_base[tm] = (blk * G::PP + y * G::PW + x) * P_ROWB;
    if constexpr (G::WHOLE) {
      a_ok[tm] = 0;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int sy = y + t / 3 - 1, sx = x + t % 3 - 1;
        if ((unsigned)sy < (unsigned)BS && (unsigned)sx < (unsigned)BS) a_ok[tm] |= 1 << t;
      }
      a_zb[tm] = ZOFF + ((x & 1) << 7);
#pragma unroll
      for (int ty = 0; ty < 3; ++ty) a_sx[tm][ty] = (pt_swz_w<BS>(blk, y + ty - 1) ^ (Q8 ? h : 2 * h)) << 4;
    } else {
      a_ok[tm] = 0;
      a_zb[tm] = 0;
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) a_sx[tm][tx] = (pt_swz_a(y, x + tx) ^ (Q8 ? h : 2 * h)) << 4;  // Q8: fp16 piece 2j+h
    }
  }
  // record address (without the buffer-independent immediate of the haloed geometries) and swizzled piece offset of
  // k-step 0 of one activation fragment read
  auto a_addr = [&](auto TMC, auto TAPC, int pb, int& rec, int& sxo) {
    constexpr int tm = decltype(TMC)::value;
    constexpr int tap = decltype(TAPC)::value;
    constexpr int ty = tap / 3, tx = tap % 3;
    if constexpr (G::WHOLE) {
      constexpr int imm = ((ty - 1) * BS + (tx - 1)) * P_ROWB;
      const int inside = pb + a_base[tm] + imm;
      const int outside = (tx == 1) ? a_zb[tm] : (a_zb[tm] ^ 128);
      rec = ((a_ok[tm] >> tap) & 1) ? inside : outside;
      sxo = a_sx[tm][ty];
    } else {
      rec = pb + a_base[tm];
      sxo = a_sx[tm][tx] ^ (64 * (ty & 1));
    }
  };
  constexpr auto a_imm = [](int tap) constexpr -> int {  // folded into the ds_read offset field
    return G::WHOLE ? 0 : ((tap / 3) * G::PW + (tap % 3)) * P_ROWB;
  };
  int b_off[TN];  // weight row record + swizzled piece of k-step 0 (k-step 1: ^ 64)
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int r = wn * TN * 32 + tn * 32 + lr;
    b_off[tn] = r * P_ROWB + ((((Q8 ? h : 2 * h)) ^ pt_swz_b(r)) << 4);
  }

  using I0 = std::integral_constant<int, 0>;
  struct Frags {
    f16x8 ah[TM], al[TM], bh[TN], bl[TN];  // Q8: ah/al = fp16 hi of k-step 0/1 (same for b); a8/b8 = fp8 operands
    i32x8 a8[TM], b8[TN];
  };
  constexpr int NPROD = TM * TN;          // products per 16-channel step: 4 / 2
  constexpr int NMMA = 3 * NPROD;         // MFMAs per half stage: 12 / 6
  constexpr int NMMA1 = Q8 ? 2 * NPROD : NMMA;  // Q8: first half = the fp16 hi*hi MFMAs of both k-steps
  constexpr int NMMA2 = Q8 ? NPROD : NMMA;      //     second half = one K=64 fp8 MFMA per product
  constexpr int NRD = 2 * TM + 2 * TN;    // ds_read_b128 per half stage: 8 / 6
  // one fragment read: r < 2*TM -> activation row block r/2 (hi, lo); else weight row block (hi, lo)
  auto read_one = [&](Frags& f, auto RC, auto TAPC, auto JC, int pb) {
    constexpr int r = decltype(RC)::value;
    constexpr int tap = decltype(TAPC)::value;
    constexpr int j = decltype(JC)::value;
    constexpr int ty = tap / 3, tx = tap % 3;
    if constexpr (EXP == 3) {
      if (pb != 0) return;  // only the prologue read (pb == 0) fills the fragments
    }
    if constexpr (Q8) {
      // JC = 0: fp16 fragments of both k-steps (piece 2*js + h); JC = 1: fp8 operands (pieces 4 + 2h + q)
      constexpr int sel = r % 2;  // k-step (fp16) or 16-byte half q of the 32-byte fp8 fragment
      if constexpr (r < 2 * TM) {
        constexpr int tm = r / 2;
        int rec, sxo;
        a_addr(std::integral_constant<int, tm>{}, TAPC, pb, rec, sxo);
        rec += a_imm(tap);
        if constexpr (j == 0) {
          const f16x8 v = *reinterpret_cast<const f16x8*>(smem + rec + (sxo ^ (32 * sel)));
          if constexpr (sel == 0) f.ah[tm] = v; else f.al[tm] = v;
        } else {
          const u32x4 v = *reinterpret_cast<const u32x4*>(smem + rec + (sxo ^ 64 ^ ((h ^ (2 * h)) << 4) ^ (16 * sel)));
#pragma unroll
          for (int e = 0; e < 4; ++e) f.a8[tm][4 * sel + e] = (int)v[e];
        }
      } else {
        constexpr int tn = (r - 2 * TM) / 2;
        constexpr int sb = RING0 + (tap % 3) * B_BYTES;
        if constexpr (j == 0) {
          const f16x8 v = *reinterpret_cast<const f16x8*>(smem + (b_off[tn] ^ (32 * sel)) + sb);
          if constexpr (sel == 0) f.bh[tn] = v; else f.bl[tn] = v;
        } else {
          const u32x4 v = *reinterpret_cast<const u32x4*>(smem + (b_off[tn] ^ 64 ^ ((h ^ (2 * h)) << 4) ^ (16 * sel)) + sb);
#pragma unroll
          for (int e = 0; e < 4; ++e) f.b8[tn][4 * sel + e] = (int)v[e];
        }
      }
    } else if constexpr (r < 2 * TM) {
      constexpr int tm = r / 2;
      // record offsets are multiples of 128 and the swizzled piece offset is < 128: xor 16 flips hi <-> lo
      int rec, sxo;
      a_addr(std::integral_constant<int, tm>{}, TAPC, pb, rec, sxo);
      sxo ^= 64 * j;
      constexpr int imm = a_imm(tap);
      if constexpr (r % 2 == 0) f.ah[tm] = *reinterpret_cast<const f16x8*>(smem + (rec + sxo) + imm);
      else f.al[tm] = *reinterpret_cast<const f16x8*>(smem + (rec + (sxo ^ 16)) + imm);
    } else {
      constexpr int tn = (r - 2 * TM) / 2;
      constexpr int sb = RING0 + (tap % 3) * B_BYTES;
      const int xo = b_off[tn] ^ (64 * j);
      if constexpr (r % 2 == 0) f.bh[tn] = *reinterpret_cast<const f16x8*>(smem + xo + sb);
      else f.bl[tn] = *reinterpret_cast<const f16x8*>(smem + (xo ^ 16) + sb);
    }
  };
  // one MFMA of a half stage, term-major so that consecutive MFMAs hit different accumulators:
  // i -> term i / NPROD (lo*hi, hi*lo, hi*hi), product i % NPROD
  auto mma_one = [&](const Frags& f, auto IC, auto HALFC) {
    constexpr int i = decltype(IC)::value;
    constexpr int half = decltype(HALFC)::value;
    if constexpr (EXP == 4) return;
    // EXP 12: only every fourth MFMA - the matrix-core work per streamed weight byte of a Winograd F(2x2, 3x3) stage
    // (16 positions x [Cout][Cin] transformed weights for 2.25x fewer products, accumulators 4x per output pixel =>
    // a quarter of the MFMAs per 16 KB weight stage) with every load, fragment read and barrier of the direct kernel
    // kept: the time per stage of a Winograd kernel WITHOUT its input / output transforms
    if constexpr (EXP == 12) {
      if constexpr ((i % 4) != 0) return;
    }
    constexpr int term = i / NPROD, p = i % NPROD;
    constexpr int tm = p / TN, tn = p % TN;
    if constexpr (Q8) {
      if constexpr (half == 0) {  // fp16 hi*hi, k-step = term (0 / 1)
        if constexpr (term == 0)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[tm], f.bh[tn], acc[tm][tn], 0, 0, 0);
        else
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[tm], f.bl[tn], acc[tm][tn], 0, 0, 0);
      } else {                    // fp8 correction: K = 64 = [a8 . w_lo8 | a_lo8 . w8], block scale 2^-3
        acc[tm][tn] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(f.a8[tm], f.b8[tn], acc[tm][tn], 0, 0, 0,
                                                                      Q8_SCALE_A, 0, Q8_SCALE_B);
      }
    } else if constexpr (term == 0)
      acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[tm], f.bh[tn], acc[tm][tn], 0, 0, 0);
    else if constexpr (term == 1)
      acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[tm], f.bl[tn], acc[tm][tn], 0, 0, 0);
    else
      acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[tm], f.bh[tn], acc[tm][tn], 0, 0, 0);
  };
  auto read_frags = [&](Frags& f, auto TAPC, auto JC, int pb) {  // all reads of a half stage (prologue)
    read_one(f, std::integral_constant<int, 0>{}, TAPC, JC, pb);
    read_one(f, std::integral_constant<int, 1>{}, TAPC, JC, pb);
    read_one(f, std::integral_constant<int, 2>{}, TAPC, JC, pb);
    read_one(f, std::integral_constant<int, 3>{}, TAPC, JC, pb);
    read_one(f, std::integral_constant<int, 4>{}, TAPC, JC, pb);
    read_one(f, std::integral_constant<int, 5>{}, TAPC, JC, pb);
    if constexpr (NRD > 6) {
      read_one(f, std::integral_constant<int, 6>{}, TAPC, JC, pb);
      read_one(f, std::integral_constant<int, 7>{}, TAPC, JC, pb);
    }
  };

  // ---------------- prologue: patch(slab 0), weight stages 0..2 --------------------------------------
  static_assert(G::BYTES % 256 == 0 && RING0 % 256 == 0 && B_BYTES % 256 == 0, "hi/lo xor addressing");
  const int nitem_ = item + (gridDim.x >> 3);
  const bool has_next = nitem_ < clen;
  int mt_next = 0, nt_next = 0;
  if (has_next) decode_item(nitem_, mt_next, nt_next);
  if constexpr (STREAM) {
    if (has_next) {
      TileBlocks tb;
      tile_blocks(mt_next, tb);
      unsigned* pk = reinterpret_cast<unsigned*>(smem + POFF_OFF) + tid;
      pk[0 * 512] = poff_round(std::integral_constant<int, 0>{}, tb);
      pk[1 * 512] = poff_round(std::integral_constant<int, 1>{}, tb);
      pk[2 * 512] = poff_round(std::integral_constant<int, 2>{}, tb);
      pk[3 * 512] = poff_round(std::integral_constant<int, 3>{}, tb);
      if constexpr (G::PA > 4) pk[4 * 512] = poff_round(std::integral_constant<int, 4>{}, tb);
      if constexpr (G::PA > 5) pk[5 * 512] = poff_round(std::integral_constant<int, 5>{}, tb);
      if constexpr (G::PA > 6) pk[6 * 512] = poff_round(std::integral_constant<int, 6>{}, tb);
    }
  }
  float c11max = 0.f;  // range guard of the conv1_1 outputs (pt_range[2])
  // OVL: the successor's block origin (its conv1_1 patches are made under this tile's K loop) and the tile after it
  // (whose raw window is fetched meanwhile)
  int nx_gy0 = 0, nx_gx0 = 0;
  bool nx_valid = false, has_next2 = false;
  int mt_next2 = 0;
  if constexpr (OVL) {
    if (has_next) {
      const int crop = mt_next / nbpc, br = mt_next - crop * nbpc;
      const int by = br / nbx;
      nx_gy0 = by * 16 - 1;
      nx_gx0 = (br - by * nbx) * 16 - 1;
      nx_valid = mt_next < nblk;
      const int nitem2_ = nitem_ + (gridDim.x >> 3);
      has_next2 = nitem2_ < clen;
      int nt2_ = 0;
      if (has_next2) decode_item(nitem2_, mt_next2, nt2_);
    }
    pcur = bufX;
    pnext = bufY;
  }
  if constexpr (FUSE1) {
  if (!OVL || !primed) {  // OVL: only the first tile of a workgroup computes its own patches up front
    if constexpr (!OVL) {
      pcur = P0_OFF;
      pnext = P1_OFF;
    }
    // ---- raw window: image rows by*16-2 .. +19, columns bx*16-2 .. +19 of the 3 colour planes (zero outside) ----
    float* R = reinterpret_cast<float*>(smem + RAW_OFF);
    const int b0 = mt;  // NB == 1
    const int crop0 = b0 / nbpc, br0 = b0 - crop0 * nbpc;
    const int by0 = br0 / nbx, bx0 = br0 - by0 * nbx;
    if (!raw_ready) fetch_raw(b0);  // first tile of this workgroup
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (tid + 512 * k < 1200) R[tid + 512 * k] = rawv[k];
    if constexpr (OVL) {
      if (has_next) fetch_raw(mt_next);  // older than the ring loads below: landed when the counted wait passes
    }
    issue_b(0, 0, 0);  // conv1_2's weight ring flies while the patch is computed
    issue_b(1, 0, 1);
    issue_b(2, 0, 2);
    __syncthreads();
    if constexpr (OVL) {  // the overlapped rounds take their weight fragments from registers
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        w1h[mb] = *reinterpret_cast<const f16x8*>(smem + W1_OFF + ((mb * 64 + lane) * 2) * 16);
        w1l[mb] = *reinterpret_cast<const f16x8*>(smem + W1_OFF + ((mb * 64 + lane) * 2 + 1) * 16);
      }
    }
    // ---- conv1_1 on the 324 patch pixels as C^T = W1 X^T: MFMA rows = channels, columns = pixels, so a lane
    // ends up with ONE pixel (lane & 15 of pixel tile g) and, per 16-channel block, 4 consecutive channels: their
    // hi and lo halves leave as two 8-byte LDS stores and all per-pixel work (coordinates, swizzle, image
    // mask) is done once per lane.  16-pixel tiles (16x16x32 MFMA, K = 32 in one step): 21 tiles over 8 waves =
    // at most 3 per wave (32-pixel tiles: 11 tiles, two waves' worth of work for waves 0-2 while 3-7 wait). ----
    const int l15 = lane & 15, kg1 = lane >> 4;
    // Three rounds per wave (tiles wave, wave + 8, wave + 16; waves 5-7 have two), written branch-free so that the
    // scheduler overlaps the LDS gather of round i + 1 with the matrix-core chain and the VALU epilogue of round i
    // (two waves per SIMD hide nothing of a dependent ds_read -> cvt -> MFMA x3 -> VALU chain on their own):
    //  * the folded bias is the INITIAL VALUE of the accumulators (binit, 16 registers for the whole kernel) - no LDS
    //    read and no fma in the epilogue, the 2^-shift of the weights is one packed multiply;
    //  * patch pixels outside the image (conv1_2's zero padding applies to conv1_1's OUTPUT) get 0 as the upper bound of
    //    the ReLU / fp16-range clamp - the mask costs no instruction;
    //  * lanes past pixel 323 (last tile) recompute pixel 323 and store the same bytes to the same place.
    auto gather1 = [&](int g, f16x8& xh, f16x8& xl) {
      const int n = g * 16 + l15;
      const int nc = n < 324 ? n : 323;
      const int ppy = nc / 18, ppx = nc - ppy * 18;
      const int rbase = ppy * 20 + ppx;  // window position of tap (0,0)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = R[rbase + roff[e]];
        if (e == 3 && kg1 == 3) v = 1.f;  // k = 27: the bias slot
        xh[e] = (_Float16)v;
        xl[e] = (_Float16)(v - (float)xh[e]);
      }
    };
    auto mma1 = [&](const f16x8& xh, const f16x8& xl, f32x4 (&c1)[4]) {
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        const f16x8 w1h = *reinterpret_cast<const f16x8*>(smem + W1_OFF + ((mb * 64 + lane) * 2) * 16);
        const f16x8 w1l = *reinterpret_cast<const f16x8*>(smem + W1_OFF + ((mb * 64 + lane) * 2 + 1) * 16);
        c1[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1l, xh, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        c1[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1h, xl, c1[mb], 0, 0, 0);
        c1[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1h, xh, c1[mb], 0, 0, 0);
      }
    };
    auto emit1 = [&](int g, const f32x4 (&c1)[4]) {
      const int n0_ = g * 16 + l15;
      const int n = n0_ < 324 ? n0_ : 323;
      const int ppy = n / 18, ppx = n - ppy * 18;
      const int gy = by0 * 16 - 1 + ppy, gx = bx0 * 16 - 1 + ppx;
      const bool inimg = (b0 < nblk) && ((unsigned)gy < (unsigned)H) && ((unsigned)gx < (unsigned)W);
      const float top = inimg ? 65504.f : 0.f;  // fp16 maximum (the guard below reports anything above PT_SAT_FP16)
      const int sw = pt_swz_a(ppy, ppx);
      typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
      const int hh = kg1 & 1;  // which half (4 channels) of an 8-channel unit
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {  // channels 16 mb + 4 kg1 .. + 3 = slab mb >> 1, unit q, half hh
        const int q = (mb & 1) * 2 + (kg1 >> 1);
        const int rb = ((mb >> 1) == 0 ? pcur : pnext) + n * P_ROWB;  // this pixel's record in slab mb >> 1
        f16x4 hi, lo;
        float vv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = c1[mb][r] * fz.oscale1;
          v = __builtin_amdgcn_fmed3f(v, 0.f, top);  // ReLU, the fp16 range clamp and the image mask in one instruction
          c11max = fmaxf(c11max, v);
          vv[r] = v;
          hi[r] = (_Float16)v;
          lo[r] = (_Float16)(v - (float)hi[r]);
        }
        if constexpr (Q8) {
          // record = [fp16 hi: pieces 0..3 | e4m3(a/4): pieces 4,5 | e4m3(a_lo*512): pieces 6,7]
          *reinterpret_cast<f16x4*>(smem + rb + ((q ^ sw) << 4) + 8 * hh) = hi;
          int pa = 0, pl = 0;
          pa = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(vv[0] * 0.25f, 448.f), fminf(vv[1] * 0.25f, 448.f), pa, false);
          pa = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(vv[2] * 0.25f, 448.f), fminf(vv[3] * 0.25f, 448.f), pa, true);
          float ll[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) ll[r] = __builtin_amdgcn_fmed3f((vv[r] - (float)hi[r]) * 512.f, -448.f, 448.f);
          pl = __builtin_amdgcn_cvt_pk_fp8_f32(ll[0], ll[1], pl, false);
          pl = __builtin_amdgcn_cvt_pk_fp8_f32(ll[2], ll[3], pl, true);
          const int bo = 8 * (q & 1) + 4 * hh;  // byte of channel 8q + 4hh inside its 16-channel piece
          *reinterpret_cast<int*>(smem + rb + (((4 + (q >> 1)) ^ sw) << 4) + bo) = pa;
          *reinterpret_cast<int*>(smem + rb + (((6 + (q >> 1)) ^ sw) << 4) + bo) = pl;
        } else {
          *reinterpret_cast<f16x4*>(smem + rb + hh * 8 + (((2 * q) ^ sw) << 4)) = hi;
          *reinterpret_cast<f16x4*>(smem + rb + hh * 8 + (((2 * q + 1) ^ sw) << 4)) = lo;
        }
      }
    };
    if constexpr (OVL) {  // once per workgroup: plain rounds (the register budget belongs to the overlapped form)
#pragma nounroll
      for (int g = wave; g < (EXP == 8 ? 0 : 21); g += 8) {
        f16x8 xh0, xl0;
        f32x4 ca[4];
        gather1(g, xh0, xl0);
        mma1(xh0, xl0, ca);
        emit1(g, ca);
      }
    } else if constexpr (EXP != 8) {  // EXP 8: timing experiment without the conv1_1 prologue
      f16x8 xh0, xl0, xh1, xl1;
      f32x4 ca[4], cb[4];
      gather1(wave, xh0, xl0);
      gather1(wave + 8, xh1, xl1);
      mma1(xh0, xl0, ca);
      mma1(xh1, xl1, cb);
      const bool third = wave + 16 < 21;  // wave-uniform
      if (third) gather1(wave + 16, xh0, xl0);
      emit1(wave, ca);
      if (third) mma1(xh0, xl0, ca);
      emit1(wave + 8, cb);
      if (third) emit1(wave + 16, ca);
    }
    if (c11max > (Q8 ? PT_SAT_E4M3 : PT_SAT_FP16)) atomicAdd(&rng[2], 1u);
    c11max = 0.f;
    pt_wait_vm<2 * NBL>();  // weight stage 0 landed (this wave's part)
    __syncthreads();        // both patch slabs are complete
    if constexpr (OVL) {
      if (has_next) {  // the successor's raw window (every wave is past its gathers from the old one)
#pragma unroll
        for (int k = 0; k < 3; ++k)
          if (tid + 512 * k < 1200) R[tid + 512 * k] = rawv[k];
      }
    }
  }
  PT_STAMP(1)
  } else {
    if (!primed) {  // first tile of this workgroup
      issue_tile_head(pcur);
      PT_STAMP(0)
      pt_wait_vm<2 * NBL>();  // patch + weight stage 0 landed (this wave's part)
      __builtin_amdgcn_s_barrier();
    }
    PT_STAMP(1)
  }
  Frags f0, f1;
  read_frags(f0, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, pcur);

  // One stage = one (slab, tap): 2 k-steps of 16 channels.  Weight stage t+3 is issued after the
  // barrier of stage t (its ring slot was last read by stage t); the patch of the next slab is issued
  // round by round in taps 0..PA-1, BEFORE the weights of the same stage so that the counted vmcnt of
  // a later barrier retires it as well.
  // MODE 0: a slab with something to prefetch - patch slab `pslab` (of the tile poff describes) and, once the
  // weight stream passes tap 8, weights (wb2, slab ws2).  Interior slab: pslab = ws2 = slab + 1, wb2 = wbase.
  // Transition slab (last slab of a tile that has a successor): poff = the successor's table, pslab = ws2 = 0,
  // wb2 = the successor's weights - the same code, so every counted vmcnt is the interior one.
  // MODE 1: last slab of the workgroup's last tile (nothing to prefetch).
  // OVL: one 16-pixel conv1_1 round of the SUCCESSOR tile (see the serial form above) cut into 36 slices = the MFMA
  // slots of three stages; round rr of a wave runs in stages 3 rr .. 3 rr + 2 of the current tile's second slab.
  //   0: pixel of this lane            1-4: window gather (2 values each)     5: image mask, swizzle
  //   6-9: hi/lo split of the gathered values (2 each)                        12-23: the 12 MFMAs (16x16x32)
  //   15-30: one output value per slot (scale, ReLU/clamp/mask, hi/lo split), 8-byte stores after each fourth
  // Slab 0 of the successor (channels 0-31) goes to bufZ, slab 1 to bufX (this tile's first slab: dead).
  int pr_ppy = 0, pr_ppx = 0, pr_rbase = 0, pr_rec = 0, pr_sw = 0;
  float pr_top = 0.f;
  float pr_v[8];
  f16x8 pr_xh, pr_xl;
  f32x4 pr_c[4];
  typedef _Float16 pr_f16x4 __attribute__((ext_vector_type(4)));
  pr_f16x4 pr_hi, pr_lo;
  auto pro_slot = [&](auto SC, auto RC) {
    constexpr int sl = decltype(SC)::value;
    constexpr int rr = decltype(RC)::value;
    if constexpr (OVL && EXP != 8) {
      const float* R = reinterpret_cast<const float*>(smem + RAW_OFF);
      if constexpr (sl == 0) {
        const int g0 = wave + 8 * rr;
        const int g = g0 < 21 ? g0 : 20;  // waves 5-7 repeat tile 20 in their third round (same bytes, same place)
        // laundered lane id: everything below depends on the lane only, would be hoisted out of the tile loop for all
        // three rounds and spilled (scratch reloads with vmcnt(0) inside the K loop)
        int lane_ = lane;
        asm volatile("" : "+v"(lane_));
        const int n = g * 16 + (lane_ & 15);
        const int nc = n < 324 ? n : 323;
        pr_ppy = nc / 18;
        pr_ppx = nc - pr_ppy * 18;
        pr_rbase = pr_ppy * 20 + pr_ppx;
        pr_rec = nc * P_ROWB;
      } else if constexpr (sl >= 1 && sl <= 4) {
        constexpr int e = 2 * (sl - 1);
        pr_v[e] = R[pr_rbase + roff[e]];
        pr_v[e + 1] = R[pr_rbase + roff[e + 1]];
        if constexpr (e + 1 == 3) {
          if ((lane >> 4) == 3) pr_v[3] = 1.f;  // k = 27: the bias slot
        }
      } else if constexpr (sl == 5) {
        const int gy = nx_gy0 + pr_ppy, gx = nx_gx0 + pr_ppx;
        const bool inimg = nx_valid & ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W);  // no branches
        pr_top = inimg ? 65504.f : 0.f;
        pr_sw = pt_swz_a(pr_ppy, pr_ppx);
      } else if constexpr (sl >= 6 && sl <= 9) {
        constexpr int e = 2 * (sl - 6);
#pragma unroll
        for (int k = e; k < e + 2; ++k) {
          pr_xh[k] = (_Float16)pr_v[k];
          pr_xl[k] = (_Float16)(pr_v[k] - (float)pr_xh[k]);
        }
      }
      if constexpr (sl >= 12 && sl <= 23 && EXP != 13) {  // EXP 13: timing experiment without the small MFMAs
        constexpr int mb = (sl - 12) / 3, term = (sl - 12) % 3;
        if constexpr (term == 0)
          pr_c[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1l[mb], pr_xh, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        else if constexpr (term == 1)
          pr_c[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1h[mb], pr_xl, pr_c[mb], 0, 0, 0);
        else
          pr_c[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1h[mb], pr_xh, pr_c[mb], 0, 0, 0);
      }
      if constexpr (sl >= 15 && sl <= 30 && EXP != 14) {  // EXP 14: timing experiment without the value epilogue
        constexpr int mb = (sl - 15) / 4, r = (sl - 15) % 4;
        float v = pr_c[mb][r] * fz.oscale1;
        v = __builtin_amdgcn_fmed3f(v, 0.f, pr_top);
        c11max = fmaxf(c11max, v);
        pr_hi[r] = (_Float16)v;
        pr_lo[r] = (_Float16)(v - (float)pr_hi[r]);
        if constexpr (r == 3) {
          const int kg1 = lane >> 4;
          const int q = (mb & 1) * 2 + (kg1 >> 1);
          const int rb = ((mb >> 1) == 0 ? bufZ : bufX) + pr_rec + (kg1 & 1) * 8;
          *reinterpret_cast<pr_f16x4*>(smem + rb + (((2 * q) ^ pr_sw) << 4)) = pr_hi;
          *reinterpret_cast<pr_f16x4*>(smem + rb + (((2 * q + 1) ^ pr_sw) << 4)) = pr_lo;
        }
      }
    }
  };
  bool rawfly = false;  // OVL: three raw-window loads were issued in stage 0 of the second slab (counted waits below)
  auto stage = [&](auto TAPC, auto MODEC, auto PROC, int slab, int pslab, const u32x4* wb2, int ws2) {
    constexpr int tap = decltype(TAPC)::value;
    constexpr int MODE = decltype(MODEC)::value;
    constexpr bool PRO = decltype(PROC)::value != 0;  // OVL: the successor's conv1_1 slices ride in the MFMA slots
    constexpr bool last = (MODE == 1);
    // loads this wave issued one stage earlier (they may stay in flight across this stage's barrier)
    constexpr int ptap = (tap + 8) % 9;  // tap of the previous stage
    constexpr int prev_issued =
        (tap == 0) ? NBL  // previous stage = tap 8 of a non-last slab (or the prologue's stage-2 weights)
                   : (((last || FUSE1) ? 0 : (ptap < G::PA ? 1 : 0)) + ((last && ptap + 3 > 8) ? 0 : NBL));
    // Half stages are written as ONE MFMA + ONE other instruction at a time (sched_barrier pins the order):
    // the two waves of a SIMD run in lockstep after every barrier, so any run of non-MFMA issue (8 ds_reads,
    // an LDS-DMA with its address math) leaves the matrix pipe idle unless it is cut into MFMA-sized gaps.
    // first half: MFMAs of the first 16-channel step, fragments of the second step read underneath
    auto half1 = [&](auto IC) {
      constexpr int i = decltype(IC)::value;
      if constexpr (i < NMMA1) {
        mma_one(f0, IC, I0{});
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (i < NRD) {
        read_one(f1, IC, TAPC, std::integral_constant<int, 1>{}, pcur);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (PRO && i < 6) {
        pro_slot(std::integral_constant<int, (tap % 3) * 12 + i>{}, std::integral_constant<int, tap / 3>{});
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    half1(std::integral_constant<int, 0>{});
    half1(std::integral_constant<int, 1>{});
    half1(std::integral_constant<int, 2>{});
    half1(std::integral_constant<int, 3>{});
    half1(std::integral_constant<int, 4>{});
    half1(std::integral_constant<int, 5>{});
    half1(std::integral_constant<int, 6>{});
    half1(std::integral_constant<int, 7>{});
    half1(std::integral_constant<int, 8>{});
    half1(std::integral_constant<int, 9>{});
    half1(std::integral_constant<int, 10>{});
    half1(std::integral_constant<int, 11>{});
    if constexpr (PRO && (tap == 1 || tap == 2)) {
      // the three raw-window loads of stage 0 are younger than the weights this barrier needs: they may stay in flight
      if (rawfly) pt_wait_vm<prev_issued + 3>();
      else pt_wait_vm<prev_issued>();
    } else if constexpr (EXP != 1) pt_wait_vm<prev_issued>();  // weights of stage t+1 (and every older load) landed
    // lgkmcnt(0) as a compiler-visible s_waitcnt (vmcnt 63 / expcnt 7 / lgkmcnt 0): the waitcnt pass then
    // knows the second step's fragments have landed and does not re-wait after the next reads are issued
    __builtin_amdgcn_s_waitcnt(0xC07F);
    if constexpr (EXP != 2) __builtin_amdgcn_s_barrier();
    // second half: MFMAs of the second step; first step of the next stage read underneath; this stage's
    // loads (patch round first, then the weights of stage t+3) issued at three spread-out points
    auto half2 = [&](auto IC) {
      constexpr int i = decltype(IC)::value;
      if constexpr (i < NMMA2) {
        mma_one(f1, IC, std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (i < NRD) {
        if constexpr (tap < 8) read_one(f0, IC, std::integral_constant<int, (tap + 1) % 9>{}, I0{}, pcur);
        else if constexpr (MODE == 0) read_one(f0, IC, I0{}, I0{}, pnext);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (EXP != 1) {
        constexpr int at_patch = (BN == 128) ? 2 : 1, at_b0 = (BN == 128) ? 6 : 4, at_b1 = 10;
        if constexpr (i == at_patch) {
          if constexpr (!FUSE1 && !last && tap < G::PA) issue_patch_round(TAPC, pslab, pnext);
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (i == at_b0 || (i == at_b1 && NBL > 1)) {
          using BI = std::integral_constant<int, (i == at_b0) ? 0 : 1>;
          if constexpr (tap + 3 <= 8) issue_b1(BI{}, wbase, tap + 3, slab, tap % 3);
          else if constexpr (MODE == 0) issue_b1(BI{}, wb2, tap + 3 - 9, ws2, tap % 3);
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (PRO && tap == 0 && i == 5) {
          // raw window of the tile after the successor: issued AFTER this stage's weights (so that the waits of the next
          // two stages can leave it in flight), consumed after the K loop
          if (rawfly) fetch_raw(mt_next2);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if constexpr (PRO && i < 6) {
        pro_slot(std::integral_constant<int, (tap % 3) * 12 + 6 + i>{}, std::integral_constant<int, tap / 3>{});
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    half2(std::integral_constant<int, 0>{});
    half2(std::integral_constant<int, 1>{});
    half2(std::integral_constant<int, 2>{});
    half2(std::integral_constant<int, 3>{});
    half2(std::integral_constant<int, 4>{});
    half2(std::integral_constant<int, 5>{});
    half2(std::integral_constant<int, 6>{});
    half2(std::integral_constant<int, 7>{});
    half2(std::integral_constant<int, 8>{});
    half2(std::integral_constant<int, 9>{});
    half2(std::integral_constant<int, 10>{});
    half2(std::integral_constant<int, 11>{});
  };
  auto slab_body = [&](auto LASTC, auto PROC, int slab, int pslab, const u32x4* wb2, int ws2) {
    stage(std::integral_constant<int, 0>{}, LASTC, PROC, slab, pslab, wb2, ws2);
    stage(std::integral_constant<int, 1>{}, LASTC, PROC, slab, pslab, wb2, ws2);
    stage(std::integral_constant<int, 2>{}, LASTC, PROC, slab, pslab, wb2, ws2);
    stage(std::integral_constant<int, 3>{}, LASTC, PROC, slab, pslab, wb2, ws2);
    stage(std::integral_constant<int, 4>{}, LASTC, PROC, slab, pslab, wb2, ws2);
    stage(std::integral_constant<int, 5>{}, LASTC, PROC, slab, pslab, wb2, ws2);
    stage(std::integral_constant<int, 6>{}, LASTC, PROC, slab, pslab, wb2, ws2);
    stage(std::integral_constant<int, 7>{}, LASTC, PROC, slab, pslab, wb2, ws2);
    stage(std::integral_constant<int, 8>{}, LASTC, PROC, slab, pslab, wb2, ws2);
  };
  // slabs 0 .. nslab-2 are interior; the last one is a transition slab when the tile has a successor
  // 8x8-block variants (no LDS room to park a table): chained when the successor works on the same pixel tile
  // (another channel tile of it - the usual order inside an XCD's chunk), whose table is the one in registers
  const bool chain = !FUSE1 && has_next && (STREAM || mt_next == mt);
  if constexpr (OVL) {
    using I1 = std::integral_constant<int, 1>;
    slab_body(I0{}, I0{}, 0, 1, wbase, 1);
    pcur = bufY;
    pnext = bufZ;  // the successor's first slab: its first fragments are read at the end of the transition slab
    if (has_next) {
      rawfly = has_next2;
      slab_body(I0{}, I1{}, 1, 0, wbase, 0);  // transition slab (same weights for every tile) + the successor's conv1_1
      if (c11max > PT_SAT_FP16) atomicAdd(&rng[2], 1u);
      c11max = 0.f;
    } else {
      rawfly = false;
      slab_body(I1{}, I0{}, 1, 0, wbase, 0);
    }
    primed = has_next;
  } else {
  if (chain) wbase_next = wp + (long)(nt_next * BN) * (cin8 * 2);
  const int nloop = chain ? nslab : nslab - 1;
  for (int slab = 0; slab < nloop; ++slab) {
    const bool trans = (slab == nslab - 1);
    if constexpr (STREAM) {
      if (trans) {  // the successor's table replaces this tile's (dead: its last patch slab was requested a slab ago)
        const unsigned* pk = reinterpret_cast<const unsigned*>(smem + POFF_OFF) + tid;
#pragma unroll
        for (int k = 0; k < G::PA; ++k) poff[k] = pk[k * 512];
      }
    }
    slab_body(std::integral_constant<int, 0>{}, I0{}, slab, trans ? 0 : slab + 1, trans ? wbase_next : wbase,
              trans ? 0 : slab + 1);
    if (!trans) {
      const int t = pcur;
      pcur = pnext;
      pnext = t;
    }
  }
  if (!chain) slab_body(std::integral_constant<int, 1>{}, I0{}, nslab - 1, 0, wbase, 0);
  primed = chain;
  }

  if constexpr (FUSE1 && !OVL) {
    raw_ready = has_next;
    if (raw_ready) fetch_raw(mt_next);
  }
  // ---- epilogue: accumulators -> LDS fp32 [256][BN+4] -> pool/bias/relu/split -> hl16 -------------
  // thread -> fixed channel unit u (8 channels) and rows r0, r0 + RSTEP, ...: bias is loaded once, before
  // the barriers (a load inside the store loop costs one L2 round trip per iteration)
  constexpr int UN = BN / 8;
  constexpr int RSTEP = 512 / UN;  // 32 / 64
  const int eu = tid % UN, er0 = tid / UN;
  const f32x8 bv = *reinterpret_cast<const f32x8*>(&bias[n0 + eu * 8]);
  const f32x8 sv = *reinterpret_cast<const f32x8*>(&oscv[n0 + eu * 8]);  // per-output-channel 2^-shift
  // Q8: thread -> 16 channels (hi = two pieces, fp8 copies = one piece each: 4 stores of 16 bytes per 16 channels)
  constexpr int UN16 = BN / 16;
  constexpr int RSTEP16 = 512 / UN16;  // 64 / 128
  const int eu16 = tid % UN16, er16 = tid / UN16;
  float bq[16], sq[16];
  if constexpr (Q8) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      bq[e] = bias[n0 + eu16 * 16 + e];
      sq[e] = oscv[n0 + eu16 * 16 + e];
    }
  }
  // blocks of this tile: validity and first pixel (block-local (0,0)) of each
  int bcrop[G::NTB], bgy0[G::NTB], bgx0[G::NTB];
#pragma unroll
  for (int k = 0; k < G::NTB; ++k) {
    const int b = mt * G::NB + k;
    const int crop = b / nbpc;
    const int br = b - crop * nbpc;
    const int by = br / nbx;
    bcrop[k] = (b < nblk) ? crop : -1;
    bgy0[k] = by * BS;
    bgx0[k] = (br - by * nbx) * BS;
  }
  // pixel (y, x) of block blk of this tile -> crop (-1: no such block), image coordinates (WHOLE: the block is the
  // whole map of crop mt * NB + blk)
#define PT_BLK_PIXEL(blk, y, x)                                   \
  int crop = bcrop[0], gy = bgy0[0] + (y), gx = bgx0[0] + (x);    \
  if constexpr (G::WHOLE) {                                       \
    crop = mt * G::NB + (blk);                                    \
    if (crop >= nblk) crop = -1;                                  \
  } else {                                                        \
    _Pragma("unroll") for (int k = 1; k < G::NTB; ++k) if ((blk) == k) { \
      crop = bcrop[k];                                            \
      gy = bgy0[k] + (y);                                         \
      gx = bgx0[k] + (x);                                         \
    }                                                             \
  }
  if constexpr (EXP != 5) {
  // barriers of the epilogue order LDS traffic only (lgkmcnt): a __syncthreads() would also drain vmcnt, i.e.
  // wait for the successor's loads and for this tile's own global stores
  pt_lds_barrier();  // every wave is past its last read of the buffer that becomes the staging area
  PT_STAMP(2)
  float* Cs = reinterpret_cast<float*>(smem + ((FUSE1 && !OVL) ? 0 : pcur));
  if constexpr (OVL) {
    if (rawfly) {  // every wave is past its gathers from the successor's window: the one after it moves in
      float* R = reinterpret_cast<float*>(smem + RAW_OFF);
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (tid + 512 * k < 1200) R[tid + 512 * k] = rawv[k];
    }
  }
  const int cout8 = Cout >> 3;
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
  if (ch > 0) pt_lds_barrier();  // the previous chunk has been read by everyone
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int rb = wm * TM + tm;  // 32-row block of the tile (wave-uniform)
    if constexpr (POOL) {
      // quad q = 2j + h of the 32-row block holds rows 8j + 4h .. + 3 = accumulator registers 4j .. 4j + 3
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float m = fmaxf(fmaxf(acc[tm][tn][4 * j], acc[tm][tn][4 * j + 1]),
                                fmaxf(acc[tm][tn][4 * j + 2], acc[tm][tn][4 * j + 3]));
          Cs[(rb * 8 + 2 * j + h) * CLD + wn * TN * 32 + tn * 32 + lr] = m;
        }
    } else if (rb / (8 / NCH) == ch) {
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          Cs[((rb % (8 / NCH)) * 32 + mm_acc_row(e, lane)) * CLD + wn * TN * 32 + tn * 32 + lr] = acc[tm][tn][e];
    }
  }
  pt_lds_barrier();
  if (ch == 0) { PT_STAMP(3) }
  if constexpr (Q8) {
    const int Hq = H >> 1, Wq = W >> 1;
    constexpr int NITEM = RCH;  // quads (pooled) or rows of one chunk
#pragma unroll
    for (int i = 0; i < (NITEM + RSTEP16 - 1) / RSTEP16; ++i) {
      const int itl = er16 + i * RSTEP16;  // chunk-local
      const int it = ch * NITEM + itl;
      if (itl < NITEM) {
        int blk, y, x;
        if constexpr (POOL) pt_row_to_pixel<BS>(it >> 3, (it & 7) * 4, blk, y, x);
        else pt_row_to_pixel<BS>(it >> 5, it & 31, blk, y, x);
        PT_BLK_PIXEL(blk, y, x)
        // pooled layers floor odd maps like nn.MaxPool2d(2, 2) (reference modules/vgg.py:72): a window that sticks out
        // of the image (last row / column of an odd map) produces no output
        if (crop >= 0 && gy < H && gx < W && (!POOL || ((gy >> 1) < Hq && (gx >> 1) < Wq))) {
          float v[16];
          const float* c = &Cs[itl * CLD + eu16 * 16];
#pragma unroll
          for (int e = 0; e < 16; e += 4) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(c + e);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[e + k] = fmaxf(fmaf(w4[k], sq[e + k], bq[e + k]), 0.f);
          }
          pt_range_guard<16>(v, true, rng);
          u32x4 hi0, hi1, a8, l8;
          pt_encode_q8(v, hi0, hi1, a8, l8);
          const long pix = POOL ? ((long)crop * Hq + (gy >> 1)) * Wq + (gx >> 1) : ((long)crop * H + gy) * W + gx;
          const int c0 = n0 + eu16 * 16;  // first channel: 32-channel block c0 >> 5, 16-channel half (c0 >> 4) & 1
          u32x4* o = out + (pix * cout8 * 2) + (c0 >> 5) * 8;
          const int hf = (c0 >> 4) & 1;
          if constexpr (EXP == 7) {  // timing experiment: same bytes, 128 contiguous bytes per row and instruction
            u32x4* o7 = out + (pix * cout8 * 2) + (n0 >> 5) * 8 + eu16;
            o7[0] = hi0;
            o7[UN16] = hi1;
            o7[2 * UN16] = a8;
            o7[3 * UN16] = l8;
          } else if constexpr (!POOL && EXP != 6 && EXP != 10 && EXP != 11) {
            // unpooled layers write 131 KB per tile that nothing re-reads before it has left the caches (a layer's
            // output is 1-2 GB per step): streaming stores, -3 % on the 512-channel layers, neutral elsewhere
            // (variant 11 = the same with regular stores)
            __builtin_nontemporal_store(hi0, &o[2 * hf]);
            __builtin_nontemporal_store(hi1, &o[2 * hf + 1]);
            __builtin_nontemporal_store(a8, &o[4 + hf]);
            __builtin_nontemporal_store(l8, &o[6 + hf]);
          } else if constexpr (EXP != 6 && EXP != 10) {
            o[2 * hf] = hi0;
            o[2 * hf + 1] = hi1;
            o[4 + hf] = a8;
            o[6 + hf] = l8;
          } else if (hi0[0] == 0x12345678u && l8[1] == 0x9abcdef0u) {
            o[0] = hi0;
          }
        }
      }
    }
  } else if constexpr (POOL) {
    const int Hq = H >> 1, Wq = W >> 1;
    constexpr int NQ = RCH;  // quads of the (single) chunk
#pragma unroll
    for (int i = 0; i < (NQ + RSTEP - 1) / RSTEP; ++i) {
      const int qdl = er0 + i * RSTEP;  // chunk-local
      const int qd = ch * NQ + qdl;
      if (qdl >= NQ) continue;
      int blk, y, x;
      pt_row_to_pixel<BS>(qd >> 3, (qd & 7) * 4, blk, y, x);
      PT_BLK_PIXEL(blk, y, x)
      if (crop >= 0 && gy < H && gx < W && (gy >> 1) < Hq && (gx >> 1) < Wq) {  // floor pooling: see the hq8 branch
        f32x8 v = *reinterpret_cast<const f32x8*>(&Cs[qdl * CLD + eu * 8]);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(fmaf(v[e], sv[e], bv[e]), 0.f);
        {
          float vg[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) vg[e] = v[e];
          pt_range_guard<8>(vg, false, rng);
        }
        u32x4 hi, lo;
        pt_split8(v, hi, lo);
        const long pix = ((long)crop * Hq + (gy >> 1)) * Wq + (gx >> 1);
        u32x4* o = out + (pix * cout8 + (n0 >> 3) + eu) * 2;
        if constexpr (EXP != 6) {
          o[0] = hi;
          o[1] = lo;
        } else if (hi[0] == 0x12345678u && lo[1] == 0x9abcdef0u) {
          o[0] = hi;  // keeps the computation alive without storing
        }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < RCH / RSTEP; ++i) {
      const int rl = er0 + i * RSTEP;  // chunk-local
      const int r = ch * RCH + rl;
      int blk, y, x;
      pt_row_to_pixel<BS>(r >> 5, r & 31, blk, y, x);
      PT_BLK_PIXEL(blk, y, x)
      if (crop >= 0 && gy < H && gx < W) {
        f32x8 v = *reinterpret_cast<const f32x8*>(&Cs[rl * CLD + eu * 8]);
        u32x4 hi, lo;
        if constexpr (RAW) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], sv[e], bv[e]);
          hi = __builtin_bit_cast(u32x4, f32x4{v[0], v[1], v[2], v[3]});
          lo = __builtin_bit_cast(u32x4, f32x4{v[4], v[5], v[6], v[7]});
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(fmaf(v[e], sv[e], bv[e]), 0.f);
          {
            float vg[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) vg[e] = v[e];
            pt_range_guard<8>(vg, false, rng);
          }
          pt_split8(v, hi, lo);
        }
        const long pix = ((long)crop * H + gy) * W + gx;
        u32x4* o = out + (pix * cout8 + (n0 >> 3) + eu) * 2;
        if constexpr (EXP == 16) {
          // A/B variant: streaming stores, as the hq8 branch uses.  +1.4 / +4.6 / +3.6 % on the 64 -> 128 / 256 -> 256 /
          // 512 -> 512 layers at 4 pairs per launch, -1.7 / -0.9 / 0 % at the 16 pairs per launch of the benchmark
          // (tools/bench_conv_variants.py --crops 2048 --variants 11,20): not adopted for the f16x3 arithmetic
          __builtin_nontemporal_store(hi, &o[0]);
          __builtin_nontemporal_store(lo, &o[1]);
        } else if constexpr (EXP != 6) {
          o[0] = hi;
          o[1] = lo;
        } else if (hi[0] == 0x12345678u && lo[1] == 0x9abcdef0u) {
          o[0] = hi;  // keeps the computation alive without storing
        }
      }
    }
  }
  }  // chunks
  }
  PT_STAMP(4)
  mt = mt_next;
  nt = nt_next;
  wbase = wbase_next;
  if constexpr (OVL) {  // (X, Y, Z) <- (Z, X, Y)
    const int t = bufX;
    bufX = bufZ;
    bufZ = bufY;
    bufY = t;
  }
  if constexpr (!FUSE1) {
    if (chain) {  // the successor's slab 0 sits in pnext
      const int t = pcur;
      pcur = pnext;
      pnext = t;
    } else if (has_next) {  // (8x8-block variants only) another pixel tile: it starts with a load prologue
      wbase = wp + (long)(nt_next * BN) * (cin8 * 2);
      compute_poff(mt_next);
    }
  }
  // LDS-only: the staging area is the successor's next patch buffer (FUSE1: the next tile's raw window / patches);
  // the tile's own stores drain under the next tile, the raw-window registers are waited for where they are used
  // (EXP 17, timing experiment: chained tiles without this barrier - the staging buffer is next written by the patch DMA
  // the successor issues after ITS first stage barrier.  Unverified for correctness; see the roadmap in DESIGN.md section 7)
  if constexpr (EXP == 17) {
    if (!chain) pt_lds_barrier();
  } else {
    pt_lds_barrier();
  }
  PT_STAMP(5)
  if constexpr (EXP == 9 || EXP == 10) {
    if (threadIdx.x == 0) atomicAdd(&pt_dbg[7], 1ull);
  }
  }  // persistent tile loop
}

#ifdef MMMOT_DEBUG
extern "C" int mmmot_debug_read_patch_timers(unsigned long long* out8, int reset) {
  hipError_t e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(pt_dbg), 8 * sizeof(unsigned long long));
  if (e != hipSuccess) return (int)e;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    e = hipMemcpyToSymbol(HIP_SYMBOL(pt_dbg), z, sizeof(z));
  }
  return mm_check(e);
}
#endif

// Counter block the trunk launches of THIS host thread report to (device memory, 4 x uint32, zeroed and read by the
// caller like any other buffer: stream-ordered, capturable); NULL returns to the library's per-device block that
// mmmot_trunk_range_read serves.  The pointer is passed to every launch as a kernel argument, so a launch keeps the
// block it was issued (or captured) with.
static thread_local unsigned int* g_range_bound = nullptr;
extern "C" int mmmot_trunk_range_bind(unsigned int* counters4) {
  if (counters4 && (((uintptr_t)counters4) & 3u)) return MMMOT_EINVAL;
  g_range_bound = counters4;
  return MMMOT_OK;
}
static unsigned int* pt_range_block() {
  if (g_range_bound) return g_range_bound;
  static std::atomic<unsigned int*> cache[16];  // symbol address per device, looked up once
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) return nullptr;
  unsigned int* p = dev < 16 ? cache[dev].load(std::memory_order_relaxed) : nullptr;
  if (!p) {
    if (hipGetSymbolAddress((void**)&p, HIP_SYMBOL(pt_range)) != hipSuccess) return nullptr;
    if (dev < 16) cache[dev].store(p, std::memory_order_relaxed);
  }
  return p;
}

// Range-guard counters (see pt_range): synchronous read of the CURRENT device's library-owned counters into a HOST
// array of 4 (launches issued while a caller's block was bound do not show here); the caller synchronises the launch
// stream first.  reset != 0 clears them.
extern "C" int mmmot_trunk_range_read(unsigned int* out4, int reset) {
  if (!out4) return MMMOT_EINVAL;
  hipError_t e = hipMemcpyFromSymbol(out4, HIP_SYMBOL(pt_range), 4 * sizeof(unsigned int));
  if (e != hipSuccess) return (int)e;
  if (reset) {
    unsigned int z[4] = {0, 0, 0, 0};
    e = hipMemcpyToSymbol(HIP_SYMBOL(pt_range), z, sizeof(z));
  }
  return mm_check(e);
}

static std::atomic<int> g_patch_grid_limit{0};
// Test knob: cap the persistent grid (a multiple of 8; 0 = one workgroup per CU) so that small problems exercise
// the tile chaining (several tiles per workgroup) that production sizes run with.  Results do not depend on it.
extern "C" int mmmot_set_patch_grid_limit(int n) {
  if (n < 0 || n % 8 != 0) return MMMOT_EINVAL;
  g_patch_grid_limit.store(n);
  return MMMOT_OK;
}
#ifdef MMMOT_DEBUG
// Timing experiments (tools/ only; -DMMMOT_DEBUG builds of the library, never the product build): variants 1..8 give
// WRONG results by construction (they remove loads / barriers / MFMAs / stores to time what is left).
static int g_patch_exp = 0;
extern "C" int mmmot_set_patch_variant(int v) {
  if (v < 0 || v > 17) return MMMOT_EINVAL;
  g_patch_exp = v;
  return MMMOT_OK;
}
#endif

static std::atomic<int> g_patch_min_block{0};
// Test knob: smallest block edge the dispatcher may choose (0 / 4 = automatic; 8 = maps of at most 4 x 4 pixels run in the
// haloed 8 x 8 geometry like before round 4).  Results do not depend on it - bit for bit: the zero halo adds exact zeros
// and the order of the accumulation is the same (tests/test_conv_patch_gpu.py).
extern "C" int mmmot_set_patch_min_block(int bs) {
  if (bs != 0 && bs != 4 && bs != 8) return MMMOT_EINVAL;
  g_patch_min_block.store(bs);
  return MMMOT_OK;
}

static int pt_num_cu() { return mm_num_cu(); }

// block edge of a layer: 16 x 16 blocks unless the whole map fits an 8 x 8 block; maps of at most 4 x 4 pixels
// (conv5 at 64-pixel crops) take the halo-free whole-map geometry, 16 maps per tile
static int pt_block_edge(int H, int W) {
  if (H > 8 || W > 8) return 16;
  if (H <= 4 && W <= 4 && g_patch_min_block.load() != 8) return 4;
  return 8;
}

// 128- or 64-channel tiles?  128 halves the LDS traffic per MFMA, but a small problem (one reference-shaped frame pair:
// 22 crops, 14 x 14 maps at conv5 = 88 tiles of 128 channels on 256 CUs) fills the chip better with twice as many
// 64-channel tiles.  Same arithmetic per output element either way (bitwise identical results).
static bool pt_use_bn64(int L, int H, int W, int Cout) {
  if (Cout % 128 != 0) return true;
  const int n_cu = pt_num_cu();
  if (n_cu <= 0) return false;
  const int bs = pt_block_edge(H, W), nb = 256 / (bs * bs);
  const long nblk = (long)L * ((H + bs - 1) / bs) * ((W + bs - 1) / bs);
  const long i128 = ((nblk + nb - 1) / nb) * (Cout / 128), i64 = 2 * i128;
  auto eff = [&](long items) { return (double)items / (double)(((items + n_cu - 1) / n_cu) * n_cu); };
  return 0.85 * eff(i64) > eff(i128);  // a 64-channel tile does half the work of a 128-channel one in ~0.59 of the time
}

template <int BN, int BS, bool POOL, int EXP, bool FUSE1 = false, bool Q8 = false, bool RAW = false>
static int launch_patch_e(const void* in, const void* wp, const float* bias, void* out, int L, int H, int W, int Cin,
                        int Cout, const float* oscale, hipStream_t s, Fuse1Args fz = Fuse1Args{nullptr, nullptr, nullptr, 1.f, nullptr, {0.f, 0.f, 0.f}, {1.f, 1.f, 1.f}}) {
  const int nby = (H + BS - 1) / BS, nbx = (W + BS - 1) / BS;
  const int nblk = L * nby * nbx;
  constexpr int NB = PatchGeom<BS>::NB;
  const int ntm = (nblk + NB - 1) / NB;
  const int ntn = Cout / BN;
  const int n_cu = pt_num_cu();
  if (n_cu <= 0) return MMMOT_EINVAL;
  const int nitems = ntm * ntn;
  int grid = (n_cu / 8) * 8;                       // one persistent workgroup per CU, whole XCDs
  const int glimit = g_patch_grid_limit.load();
  if (glimit > 0 && grid > glimit) grid = glimit;
  if (grid > ((nitems + 7) / 8) * 8) grid = ((nitems + 7) / 8) * 8;
  unsigned int* rng = pt_range_block();
  if (!rng) return MMMOT_EINVAL;
  hipLaunchKernelGGL((conv3x3_hl16_patch_kernel<BN, BS, POOL, EXP, FUSE1, Q8, RAW>), dim3(grid), dim3(512), 0, s,
                     (const u32x4*)in, (const u32x4*)wp, bias, (u32x4*)out, L, H, W, Cin, Cout, nby, nbx, nblk, ntm, ntn,
                     oscale, fz, rng);
  return mm_check(hipGetLastError());
}

template <int BN, int BS, bool POOL>
static int launch_patch(const void* in, const void* wp, const float* bias, void* out, int L, int H, int W, int Cin,
                        int Cout, const float* oscale, hipStream_t s) {
#ifdef MMMOT_DEBUG
  if constexpr (BN == 128 && BS == 16)  // phase timers: pooled and unpooled (tools/patch_phase_timers_f16x3.py)
    if (g_patch_exp == 9) return launch_patch_e<BN, BS, POOL, 9>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
  if constexpr (BN == 128 && BS == 16 && !POOL) {  // the experiments exist for one instantiation only
    switch (g_patch_exp) {
      case 1: return launch_patch_e<BN, BS, POOL, 1>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
      case 2: return launch_patch_e<BN, BS, POOL, 2>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
      case 3: return launch_patch_e<BN, BS, POOL, 3>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
      case 4: return launch_patch_e<BN, BS, POOL, 4>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
      case 5: return launch_patch_e<BN, BS, POOL, 5>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
      case 6: return launch_patch_e<BN, BS, POOL, 6>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
      case 12: return launch_patch_e<BN, BS, POOL, 12>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
      case 15: return launch_patch_e<BN, BS, POOL, 15>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
      case 16: return launch_patch_e<BN, BS, POOL, 16>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
      case 17: return launch_patch_e<BN, BS, POOL, 17>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
      default: break;
    }
  }
#endif
  return launch_patch_e<BN, BS, POOL, 0>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
}

template <int BN, int BS>
static int launch_patch_p(int pool, const void* in, const void* wp, const float* bias, void* out, int L, int H, int W,
                          int Cin, int Cout, const float* oscale, hipStream_t s) {
  return pool ? launch_patch<BN, BS, true>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s)
              : launch_patch<BN, BS, false>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
}

// Same contract as mmmot_conv3x3_bn_relu_hl16 (Cin % 32 == 0, Cout % 64 == 0); any H, W >= 1: pool = 1 on an odd map
// floors like nn.MaxPool2d(2, 2) - the output is (H >> 1) x (W >> 1).
extern "C" int mmmot_conv3x3_bn_relu_hl16_patch(const void* in, const void* wp, const float* bias, void* out, int L,
                                                int H, int W, int Cin, int Cout, int pool, const float* oscale,
                                                void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!in || !wp || !bias || !out || !oscale || L <= 0 || H <= 0 || W <= 0) return MMMOT_EINVAL;
  if (Cin % 32 != 0 || Cout % 64 != 0) return MMMOT_EINVAL;  // odd maps: floor pooling (partial windows dropped)
  if (!mm_al16(in) || !mm_al16(wp) || !mm_al16(out)) return MMMOT_EINVAL;
  if ((long)L * H * W * (Cin / 4) >= (1L << 31) - 64) return MMMOT_EINVAL;  // 32-bit piece offsets
  const int bs = pt_block_edge(H, W);
  if (!pt_use_bn64(L, H, W, Cout))
    return bs == 16 ? launch_patch_p<128, 16>(pool, in, wp, bias, out, L, H, W, Cin, Cout, oscale, s)
           : bs == 8 ? launch_patch_p<128, 8>(pool, in, wp, bias, out, L, H, W, Cin, Cout, oscale, s)
                     : launch_patch_p<128, 4>(pool, in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
  return bs == 16 ? launch_patch_p<64, 16>(pool, in, wp, bias, out, L, H, W, Cin, Cout, oscale, s)
         : bs == 8 ? launch_patch_p<64, 8>(pool, in, wp, bias, out, L, H, W, Cin, Cout, oscale, s)
                   : launch_patch_p<64, 4>(pool, in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
}

// Training (mmmot_amd/train_vgg.py): the plain convolution out[p][n] = oscale[n] * sum in * wp + bias[n] as fp32 rows
// [L * H * W][Cout] - the f16x3 counterpart of mmmot_conv3x3_raw for the training-mode forward (in = hl16 activations)
// and for the input gradient (in = hl16 of the scaled dZ, wp = flipped / transposed weights, oscale carrying both
// power-of-two scales back).  Same geometry dispatch and accumulation order as the inference launch.
extern "C" int mmmot_conv3x3_raw_hl16(const void* in, const void* wp, const float* bias, float* out, int L, int H, int W,
                                      int Cin, int Cout, const float* oscale, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!in || !wp || !bias || !out || !oscale || L <= 0 || H <= 0 || W <= 0) return MMMOT_EINVAL;
  if (Cin % 32 != 0 || Cout % 64 != 0) return MMMOT_EINVAL;
  if (!mm_al16(in) || !mm_al16(wp) || !mm_al16(out)) return MMMOT_EINVAL;
  if ((long)L * H * W * (Cin / 4) >= (1L << 31) - 64) return MMMOT_EINVAL;
  const int bs = pt_block_edge(H, W);
#define PT_RAW(BNV, BSV) launch_patch_e<BNV, BSV, false, 0, false, false, true>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s)
  if (!pt_use_bn64(L, H, W, Cout)) return bs == 16 ? PT_RAW(128, 16) : bs == 8 ? PT_RAW(128, 8) : PT_RAW(128, 4);
  return bs == 16 ? PT_RAW(64, 16) : bs == 8 ? PT_RAW(64, 8) : PT_RAW(64, 4);
#undef PT_RAW
}

// conv1_1 (3 -> 64) + conv1_2 (64 -> 64) + 2x2 max-pool in one kernel: see FUSE1 above.
extern "C" int mmmot_conv1_fused_hl16(const float* crops, const void* w1, const float* bias1, float oscale1,
                                      const void* w2, const float* bias2, const float* oscale2, void* out, int L, int H,
                                      int W, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!crops || !w1 || !bias1 || !w2 || !bias2 || !oscale2 || !out || L <= 0 || H <= 0 || W <= 0) return MMMOT_EINVAL;
  if ((H & 1) || (W & 1) || !mm_al16(w1) || !mm_al16(w2) || !mm_al16(out)) return MMMOT_EINVAL;
  Fuse1Args fz{crops, (const u32x4*)w1, bias1, oscale1, nullptr, {0.f, 0.f, 0.f}, {1.f, 1.f, 1.f}};
#ifdef MMMOT_DEBUG
  if (g_patch_exp == 4) return launch_patch_e<64, 16, true, 4, true>(nullptr, w2, bias2, out, L, H, W, 64, 64, oscale2, s, fz);
  if (g_patch_exp == 8) return launch_patch_e<64, 16, true, 8, true>(nullptr, w2, bias2, out, L, H, W, 64, 64, oscale2, s, fz);
  if (g_patch_exp == 9) return launch_patch_e<64, 16, true, 9, true>(nullptr, w2, bias2, out, L, H, W, 64, 64, oscale2, s, fz);
  if (g_patch_exp == 13) return launch_patch_e<64, 16, true, 13, true>(nullptr, w2, bias2, out, L, H, W, 64, 64, oscale2, s, fz);
  if (g_patch_exp == 14) return launch_patch_e<64, 16, true, 14, true>(nullptr, w2, bias2, out, L, H, W, 64, 64, oscale2, s, fz);
#endif
  return launch_patch_e<64, 16, true, 0, true>(nullptr, w2, bias2, out, L, H, W, 64, 64, oscale2, s, fz);
}

// The same launch fed by the 8-bit crops [L][H][W][3] (RGB) of the resize: ToTensor + Normalize in the loader (mean / std
// by value; q8 != 0: conv1_2 in hq8 arithmetic, w2 / out hq8 - mmmot_conv1_fused_hq8's contract).
extern "C" int mmmot_conv1_fused_u8(const unsigned char* crops_u8, float mean0, float mean1, float mean2, float std0,
                                    float std1, float std2, const void* w1, const float* bias1, float oscale1,
                                    const void* w2, const float* bias2, const float* oscale2, void* out, int L, int H,
                                    int W, int q8, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!crops_u8 || !w1 || !bias1 || !w2 || !bias2 || !oscale2 || !out || L <= 0 || H <= 0 || W <= 0) return MMMOT_EINVAL;
  if ((H & 1) || (W & 1) || !mm_al16(w1) || !mm_al16(w2) || !mm_al16(out)) return MMMOT_EINVAL;
  if (!(std0 != 0.f) || !(std1 != 0.f) || !(std2 != 0.f)) return MMMOT_EINVAL;
  Fuse1Args fz{nullptr, (const u32x4*)w1, bias1, oscale1, crops_u8, {mean0, mean1, mean2}, {std0, std1, std2}};
  if (q8) return launch_patch_e<64, 16, true, 0, true, true>(nullptr, w2, bias2, out, L, H, W, 64, 64, oscale2, s, fz);
  return launch_patch_e<64, 16, true, 0, true>(nullptr, w2, bias2, out, L, H, W, 64, 64, oscale2, s, fz);
}

// ---- hq8 arithmetic: same contracts, activations / weights in the hq8 record format (see Q8 above) ----
template <int BN, int BS>
static int launch_q8_p(int pool, const void* in, const void* wp, const float* bias, void* out, int L, int H, int W,
                       int Cin, int Cout, const float* oscale, hipStream_t s) {
#ifdef MMMOT_DEBUG
  if constexpr (BN == 128 && BS == 16) {  // timing experiments (wrong results), unpooled 128-channel tiles only
    if (!pool && g_patch_exp == 3) return launch_patch_e<BN, BS, false, 3, false, true>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
    if (!pool && g_patch_exp == 4) return launch_patch_e<BN, BS, false, 4, false, true>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
    if (!pool && g_patch_exp == 6) return launch_patch_e<BN, BS, false, 6, false, true>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
    if (!pool && g_patch_exp == 7) return launch_patch_e<BN, BS, false, 7, false, true>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
    if (!pool && g_patch_exp == 9) return launch_patch_e<BN, BS, false, 9, false, true>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
    if (!pool && g_patch_exp == 10) return launch_patch_e<BN, BS, false, 10, false, true>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
    if (!pool && g_patch_exp == 11) return launch_patch_e<BN, BS, false, 11, false, true>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
  }
#endif
  return pool ? launch_patch_e<BN, BS, true, 0, false, true>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s)
              : launch_patch_e<BN, BS, false, 0, false, true>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
}

extern "C" int mmmot_conv3x3_bn_relu_hq8(const void* in, const void* wp, const float* bias, void* out, int L, int H,
                                         int W, int Cin, int Cout, int pool, const float* oscale, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!in || !wp || !bias || !out || !oscale || L <= 0 || H <= 0 || W <= 0) return MMMOT_EINVAL;
  if (Cin % 32 != 0 || Cout % 64 != 0) return MMMOT_EINVAL;  // odd maps: floor pooling (partial windows dropped)
  if (!mm_al16(in) || !mm_al16(wp) || !mm_al16(out)) return MMMOT_EINVAL;
  if ((long)L * H * W * (Cin / 4) >= (1L << 31) - 64) return MMMOT_EINVAL;
  const int bs = pt_block_edge(H, W);
  if (!pt_use_bn64(L, H, W, Cout))
    return bs == 16 ? launch_q8_p<128, 16>(pool, in, wp, bias, out, L, H, W, Cin, Cout, oscale, s)
           : bs == 8 ? launch_q8_p<128, 8>(pool, in, wp, bias, out, L, H, W, Cin, Cout, oscale, s)
                     : launch_q8_p<128, 4>(pool, in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
  return bs == 16 ? launch_q8_p<64, 16>(pool, in, wp, bias, out, L, H, W, Cin, Cout, oscale, s)
         : bs == 8 ? launch_q8_p<64, 8>(pool, in, wp, bias, out, L, H, W, Cin, Cout, oscale, s)
                   : launch_q8_p<64, 4>(pool, in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
}

extern "C" int mmmot_conv1_fused_hq8(const float* crops, const void* w1, const float* bias1, float oscale1,
                                     const void* w2, const float* bias2, const float* oscale2, void* out, int L, int H,
                                     int W, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!crops || !w1 || !bias1 || !w2 || !bias2 || !oscale2 || !out || L <= 0 || H <= 0 || W <= 0) return MMMOT_EINVAL;
  if ((H & 1) || (W & 1) || !mm_al16(w1) || !mm_al16(w2) || !mm_al16(out)) return MMMOT_EINVAL;
  Fuse1Args fz{crops, (const u32x4*)w1, bias1, oscale1, nullptr, {0.f, 0.f, 0.f}, {1.f, 1.f, 1.f}};
#ifdef MMMOT_DEBUG
  if (g_patch_exp == 4) return launch_patch_e<64, 16, true, 4, true, true>(nullptr, w2, bias2, out, L, H, W, 64, 64, oscale2, s, fz);
  if (g_patch_exp == 6) return launch_patch_e<64, 16, true, 6, true, true>(nullptr, w2, bias2, out, L, H, W, 64, 64, oscale2, s, fz);
  if (g_patch_exp == 8) return launch_patch_e<64, 16, true, 8, true, true>(nullptr, w2, bias2, out, L, H, W, 64, 64, oscale2, s, fz);
#endif
  return launch_patch_e<64, 16, true, 0, true, true>(nullptr, w2, bias2, out, L, H, W, 64, 64, oscale2, s, fz);
}

// fp32 rows <-> hq8 rows (tests, tools; n % 32 == 0): one thread per 16 channels
__global__ void hq8_pack_kernel(const float* __restrict__ x, u32x4* __restrict__ y, long n16) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n16) return;
  float v[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) v[e] = x[i * 16 + e];
  u32x4 hi0, hi1, a8, l8;
  pt_encode_q8(v, hi0, hi1, a8, l8);
  u32x4* o = y + (i >> 1) * 8;
  const int hf = (int)(i & 1);
  o[2 * hf] = hi0;
  o[2 * hf + 1] = hi1;
  o[4 + hf] = a8;
  o[6 + hf] = l8;
}

__global__ void hq8_unpack_kernel(const u32x4* __restrict__ x, float* __restrict__ y, long n16) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n16) return;
  const u32x4* r = x + (i >> 1) * 8;
  const int hf = (int)(i & 1);
  const f16x8 h0 = __builtin_bit_cast(f16x8, r[2 * hf]), h1 = __builtin_bit_cast(f16x8, r[2 * hf + 1]);
  const u32x4 l8 = r[6 + hf];
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const auto p0 = __builtin_amdgcn_cvt_pk_f32_fp8((int)l8[w], false);
    const auto p1 = __builtin_amdgcn_cvt_pk_f32_fp8((int)l8[w], true);
    const float lo[4] = {p0[0], p0[1], p1[0], p1[1]};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int e = 4 * w + k;
      y[i * 16 + e] = (float)(e < 8 ? h0[e] : h1[e - 8]) + lo[k] * (1.f / 512.f);
    }
  }
}

extern "C" int mmmot_hq8_pack(const float* x, void* y, long n, void* stream) {
  if (!x || !y || n <= 0 || n % 32 != 0 || !mm_al16(x) || !mm_al16(y)) return MMMOT_EINVAL;
  const long nu = n / 16;
  hipLaunchKernelGGL(hq8_pack_kernel, dim3((unsigned)((nu + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                     (u32x4*)y, nu);
  return mm_check(hipGetLastError());
}

extern "C" int mmmot_hq8_unpack(const void* x, float* y, long n, void* stream) {
  if (!x || !y || n <= 0 || n % 32 != 0 || !mm_al16(x) || !mm_al16(y)) return MMMOT_EINVAL;
  const long nu = n / 16;
  hipLaunchKernelGGL(hq8_unpack_kernel, dim3((unsigned)((nu + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const u32x4*)x, y, nu);
  return mm_check(hipGetLastError());
}
