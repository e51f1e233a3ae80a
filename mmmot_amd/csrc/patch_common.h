// Shared device helpers of the trunk kernel (conv3x3_hl16_patch.hip): activation encoders (hl16 split, hq8 records),
// the range guard of the reduced-range formats, barrier / DMA primitives, LDS swizzles, the block geometry of a tile
// (PatchGeom: 16 x 16, 8 x 8 haloed blocks or whole 4 x 4 maps), the arguments of the fused first layer and the
// constants of the hq8 arithmetic.  Included by that translation unit only.
#pragma once

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

static __device__ u32x4 pt_zero_page[16];  // zero-initialised: source of out-of-image patch pixels

#define P_BM 256
#define P_ROWB 128  // bytes per LDS record (pixel or weight row): 32 channels hi+lo

__device__ __forceinline__ void pt_split8(f32x8 v, u32x4& hi, u32x4& lo) {
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    const float x0 = fminf(fmaxf(v[e], -65000.f), 65000.f), x1 = fminf(fmaxf(v[e + 1], -65000.f), 65000.f);
    unsigned h2, l2;
    mm_split2(x0, x1, h2, l2);  // lo = (f16)(x - (float)hi) as one v_fma_mix per value (common.h)
    hi[e >> 1] = h2;
    lo[e >> 1] = l2;
  }
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

// Phase timers (tools/patch_phase_timers*.py, tools/fused1_phase_timers.py): -DMMMOT_DEBUG builds only.  The product
// translation unit sees empty PT_STAMP / PT_COUNT_ITEM macros and a kernel without the TIMED template parameter.
#ifdef MMMOT_DEBUG
#include "patch_debug.h"
#else
#define PT_TIMED_TPARAM
#define PT_TIMED_TARG(v)
#define PT_STAMP_DECL()
#define PT_STAMP(i)
#define PT_COUNT_ITEM()
#endif

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

// LDS reads of the epilogue's staging area, written out.  The compiler cannot tell the staging buffer from the LDS-DMA
// destinations of the successor tile (both are runtime offsets into one array), so it answers every LDS read it can see
// with s_waitcnt vmcnt(0) - and vmcnt also counts this tile's own global STORES: each iteration of the store loop then
// waited for the previous iteration's stores to reach the L2 (8 round trips per unpooled tile, ~10 % of the tile: the
// "encode + issue stores" phase of profiles/HISTORY.md).  These reads are invisible to that pass; the buffer really is
// disjoint from everything in flight (patch_setup.inc), and the wait for the reads themselves is written out too.
__device__ __forceinline__ unsigned pt_lds_addr(const void* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}
__device__ __forceinline__ void pt_lds_read32(unsigned addr, f32x4& a, f32x4& b) {
  asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16" : "=&v"(a), "=&v"(b) : "v"(addr));
}
__device__ __forceinline__ void pt_lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

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
