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

#include "patch_common.h"

// RAW (training, mmmot_conv3x3_raw_hl16): the output is the plain convolution  acc * oscale + bias  as fp32 rows (8 floats
// where an inference launch writes one [hi8 | lo8] unit: the same bytes at the same place) - no ReLU, no clamp, no split.
template <int BN, int BS, bool POOL, bool FUSE1 = false, bool Q8 = false, bool RAW = false PT_TIMED_TPARAM>
__global__ __launch_bounds__(512) void conv3x3_hl16_patch_kernel(
    const u32x4* __restrict__ in, const u32x4* __restrict__ wp, const float* __restrict__ bias,
    u32x4* __restrict__ out, int L, int H, int W, int Cin, int Cout, int nby, int nbx, int nblk, int ntm, int ntn,
    const float* __restrict__ oscv, Fuse1Args fz, unsigned int* __restrict__ rng) {
#include "patch_setup.inc"
  for (int item = blockIdx.x >> 3; item < clen; item += gridDim.x >> 3) {
#include "patch_loaders.inc"
#include "patch_fragments.inc"
#include "patch_prologue.inc"
#include "patch_kloop.inc"
#include "patch_epilogue.inc"
  }  // persistent tile loop
}


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

template <int BN, int BS, bool POOL, bool FUSE1 = false, bool Q8 = false, bool RAW = false PT_TIMED_TPARAM>
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
  hipLaunchKernelGGL((conv3x3_hl16_patch_kernel<BN, BS, POOL, FUSE1, Q8, RAW PT_TIMED_TARG(TIMED)>), dim3(grid), dim3(512), 0, s,
                     (const u32x4*)in, (const u32x4*)wp, bias, (u32x4*)out, L, H, W, Cin, Cout, nby, nbx, nblk, ntm, ntn,
                     oscale, fz, rng);
  return mm_check(hipGetLastError());
}

template <int BN, int BS, bool POOL>
static int launch_patch(const void* in, const void* wp, const float* bias, void* out, int L, int H, int W, int Cin,
                        int Cout, const float* oscale, hipStream_t s) {
#ifdef MMMOT_DEBUG
  if constexpr (BN == 128 && BS == 16)  // phase timers: pooled and unpooled (tools/patch_phase_timers_f16x3.py)
    if (g_patch_timed) return launch_patch_e<BN, BS, POOL, false, false, false, true>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
#endif
  return launch_patch_e<BN, BS, POOL>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
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
#define PT_RAW(BNV, BSV) launch_patch_e<BNV, BSV, false, false, false, true>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s)
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
  if (g_patch_timed) return launch_patch_e<64, 16, true, true, false, false, true>(nullptr, w2, bias2, out, L, H, W, 64, 64, oscale2, s, fz);
#endif
  return launch_patch_e<64, 16, true, true>(nullptr, w2, bias2, out, L, H, W, 64, 64, oscale2, s, fz);
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
  if (q8) return launch_patch_e<64, 16, true, true, true>(nullptr, w2, bias2, out, L, H, W, 64, 64, oscale2, s, fz);
  return launch_patch_e<64, 16, true, true>(nullptr, w2, bias2, out, L, H, W, 64, 64, oscale2, s, fz);
}

// ---- hq8 arithmetic: same contracts, activations / weights in the hq8 record format (see Q8 above) ----
template <int BN, int BS>
static int launch_q8_p(int pool, const void* in, const void* wp, const float* bias, void* out, int L, int H, int W,
                       int Cin, int Cout, const float* oscale, hipStream_t s) {
#ifdef MMMOT_DEBUG
  if constexpr (BN == 128 && BS == 16)  // phase timers (tools/patch_phase_timers.py)
    if (!pool && g_patch_timed) return launch_patch_e<BN, BS, false, false, true, false, true>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
#endif
  return pool ? launch_patch_e<BN, BS, true, false, true>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s)
              : launch_patch_e<BN, BS, false, false, true>(in, wp, bias, out, L, H, W, Cin, Cout, oscale, s);
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
  return launch_patch_e<64, 16, true, true, true>(nullptr, w2, bias2, out, L, H, W, 64, 64, oscale2, s, fz);
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
