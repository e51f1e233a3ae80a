// Weight-in-registers variant of the A-resident row GEMM (include/mmmot_hip.h: mmmot_gemm_ares) for PointNet conv5
// 128 -> 1024 + GroupNorm + ReLU + per-detection mean (reference modules/point_net.py:138-148), the consumer pass:
//
//   colsum[2t+h][n] = sum over the valid rows of 64-row half tile h of tile t of
//                     relu((oscale * sum_k relu(X[r][k]*sc[g][k] + sh[g][k]) * W[n][k] + bias[n] + dbias[..][n]) * osc[g][n] + osh[g][n])
//
// Same arithmetic, accumulation order and summation order as gemm_ares_kernel<2, 2> (gemm_ares.hip) - the results are
// bit for bit the same - different data movement.  The streaming kernel keeps the 128 activation rows in LDS and
// streams the weights through LDS too: every 12 MFMAs of a wave need 4 activation + 4 weight fragment reads and the
// kernel sits at 0.38 of the f16x3 ceiling, LDS-read-bound (rounds 1-3: weight-resident-in-LDS, deferred epilogue,
// wave roles all left it there).  Here a workgroup owns 512 of the N output channels for the whole launch and every
// wave keeps the hi / lo fragments of its 64 channels x K = 128 in REGISTERS (128 of its 256): the K loop reads only
// activation fragments - 2 LDS reads per 6 MFMAs instead of 8 per 12 -, there is no weight stream at all, and the
// activation tile of the NEXT row tile arrives by LDS-DMA (raw fp32 rows, no staging registers) while the current one
// is multiplied; at the tile switch all eight waves normalise + ReLU + split it into the fragment planes (XOR-swizzled
// 256-byte rows: conflict-free ds_read_b128).  N / 512 workgroups share a row tile (channel halves): they sit on the
// same XCD (block b and b + 8), so the second reader of a row tile finds it in that XCD's L2.
#include <atomic>
#include <cstdlib>

#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define WR_BM 128
#define WR_K 128
#define WR_PLANE (WR_BM * WR_K * 2)      // bytes of one fp16 plane [128 rows][128 k]: 32 KB
#define WR_RAW (WR_BM * WR_K * 4)        // raw fp32 tile: 64 KB

static __device__ float wr_zeros[4096];  // stands in for absent bias / dbias rows (branch-free loads)

__device__ __forceinline__ void wr_dma16(const void* src, unsigned char* dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

__global__ __launch_bounds__(512, 1) void gemm_wreg128_kernel(mmmot_gemm_ares_args a, int nh, int nseq) {
  // LDS: [A hi plane | A lo plane | raw fp32 rows of the next tile]
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * WR_PLANE + WR_RAW];
  unsigned char* Ah = smem;
  unsigned char* Al = smem + WR_PLANE;
  unsigned char* Raw = smem + 2 * WR_PLANE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, hh = lane >> 5;
  // workgroup -> (channel half, position in the tile sequence): blocks b and b + 8 (same XCD) share their row tiles
  const int b = blockIdx.x;
  const int half = (b >> 3) % nh;
  const int seq0 = (b & 7) + 8 * (b / (8 * nh));
  const int nbase = half * 512 + wave * 64;  // this wave's 64 output channels

  // ---- the wave's weights: B fragments of v_mfma_f32_32x32x16_f16 for k-step s, channel block tn: lane (lr, hh) holds
  // the 8 k = 16 s + 8 hh .. + 7 of channel nbase + 32 tn + lr = hl16 unit 2 s + hh of that weight row ([hi8 | lo8]) ----
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

  auto tile_of = [&](int q) { return seq0 + q * (int)(gridDim.x / (8 * nh)) * 8; };
  // raw rows of tile t -> Raw by LDS-DMA: instruction i of this wave covers rows 2 (8 i' ..): lane-linear 1 KB = 2 rows
  auto dma_tile = [&](int t) {
    const int row0 = a.tile_row0[t], nrows = a.tile_nrows[t];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int q = wave * 8 + i;               // 1 KB piece of the tile: rows 2q, 2q + 1
      const int r = 2 * q + (lane >> 5);
      const int rr = r < nrows ? r : 0;         // rows beyond the tile read its first row (mapped) and are zeroed later
      wr_dma16(a.X + (long)(row0 + rr) * a.ldx + (lane & 31) * 4, Raw + q * 1024);
    }
  };
  // Raw -> normalise + ReLU + clamp + hi/lo split -> fragment planes.  Chunk c = i * 512 + tid: row c / 16, 8 k = (c % 16) * 8;
  // plane row stride 256 B, 16-byte piece p of row r lives at slot p ^ (r & 15)
  auto convert_tile = [&](int t) {
    const int nrows = a.tile_nrows[t];
    const int grp = a.tile_group ? a.tile_group[t] : 0;
    const int kc = tid & 15;
    const float* psc = a.sc + (long)grp * a.ldsc + kc * 8;
    const float* psh = a.sh + (long)grp * a.ldsc + kc * 8;
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(psc), s1 = *reinterpret_cast<const f32x4*>(psc + 4);
    const f32x4 h0 = *reinterpret_cast<const f32x4*>(psh), h1 = *reinterpret_cast<const f32x4*>(psh + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = i * 32 + (tid >> 4);
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(Raw + r * 512 + kc * 32);
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(Raw + r * 512 + kc * 32 + 16);
      const bool rv = r < nrows;
      f16x8 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float y0 = fminf(fmaxf(fmaf(x0[e], s0[e], h0[e]), 0.f), 65000.f);
        float y1 = fminf(fmaxf(fmaf(x1[e], s1[e], h1[e]), 0.f), 65000.f);
        if (!rv) y0 = y1 = 0.f;
        hi[e] = (_Float16)y0;
        lo[e] = (_Float16)(y0 - (float)hi[e]);
        hi[4 + e] = (_Float16)y1;
        lo[4 + e] = (_Float16)(y1 - (float)hi[4 + e]);
      }
      const int off = r * 256 + ((kc ^ (r & 15)) << 4);
      *reinterpret_cast<f16x8*>(Ah + off) = hi;
      *reinterpret_cast<f16x8*>(Al + off) = lo;
    }
  };

  if (seq0 >= nseq) return;  // (whole workgroups only: nseq is the same for both halves)
  // ---- prologue: tile 0 -> planes, tile 1 -> Raw ----
  int q = 0;
  int t = tile_of(0);
  dma_tile(t);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  convert_tile(t);
  __syncthreads();
  {
    const int tq = tile_of(1);
    if (tq < a.T) dma_tile(tq);
  }
  for (;;) {
    const int nrows = a.tile_nrows[t];
    const int grp = a.tile_group ? a.tile_group[t] : 0;
    const int dbrow = a.dbias ? a.tile_dbrow[t] : 0;
    // per-channel epilogue constants of this tile
    float m1[2], m0[2];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int n = nbase + 32 * tn + lr;
      const float cb = pbias[n] + (a.dbias ? a.dbias[(long)dbrow * a.lddb + n] : 0.f);
      const float os = a.osc[(long)grp * a.ldosc + n], oh = a.osh[(long)grp * a.ldosc + n];
      m1[tn] = a.oscale * os;
      m0[tn] = fmaf(cb, os, oh);
    }
    float s3[2] = {0.f, 0.f};
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      f32x16 acc[2];
      // A fragment of k-step s: row 32 rb + lr, piece 2 s + hh... the planes hold 16 pieces (8 k each) per row: piece s' = 2 s + hh
      const int r = 32 * rb + lr;
      const int base = r * 256;
      f16x8 ah = *reinterpret_cast<const f16x8*>(Ah + base + ((hh ^ (r & 15)) << 4));
      f16x8 al = *reinterpret_cast<const f16x8*>(Al + base + ((hh ^ (r & 15)) << 4));
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        f16x8 ah2 = ah, al2 = al;
        if (s < 7) {  // next step's fragments under this step's MFMAs
          const int p = 2 * (s + 1) + hh;
          ah2 = *reinterpret_cast<const f16x8*>(Ah + base + ((p ^ (r & 15)) << 4));
          al2 = *reinterpret_cast<const f16x8*>(Al + base + ((p ^ (r & 15)) << 4));
        }
        // term-major over the two channel blocks; per accumulator the order of gemm_ares.hip: lo*hi, hi*lo, hi*hi
        if (s == 0) {
          const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh[0][0], z, 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh[0][1], z, 0, 0, 0);
        } else {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh[s][0], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh[s][1], acc[1], 0, 0, 0);
        }
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl[s][0], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl[s][1], acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh[s][0], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh[s][1], acc[1], 0, 0, 0);
        ah = ah2;
        al = al2;
      }
      // ---- register epilogue: this 32-row block's share of the half tile's column sums (the running sum walks row
      // block 2 h, then 2 h + 1, in register order: the order of mm_relu_sum32 over gemm_ares' 64-row wave tile) ----
      const int lim = nrows - (32 * rb + 4 * hh);  // accumulator element e is a valid row <=> (e & 3) + 8 (e >> 2) < lim
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        float s = s3[tn];
        if (lim >= 32) {
#pragma unroll
          for (int e = 0; e < 16; ++e) s += fmaxf(fmaf(acc[tn][e], m1[tn], m0[tn]), 0.f);
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e)
            if ((e & 3) + 8 * (e >> 2) < lim) s += fmaxf(fmaf(acc[tn][e], m1[tn], m0[tn]), 0.f);
        }
        s3[tn] = s;
      }
      if (rb & 1) {
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
          const float tot = mm_xor32_sum(s3[tn]);
          if (lane < 32) a.colsum[(long)(2 * t + (rb >> 1)) * a.N + nbase + 32 * tn + lr] = tot;
          s3[tn] = 0.f;
        }
      }
    }
    // ---- tile switch ----
    const int tn1 = tile_of(q + 1);
    if (tn1 >= a.T) break;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the next tile's rows (requested a tile ago)
    __syncthreads();                                  // every wave is past its last fragment read; the raw rows are complete
    convert_tile(tn1);
    __syncthreads();
    const int tn2 = tile_of(q + 2);
    if (tn2 < a.T) dma_tile(tn2);
    t = tn1;
    ++q;
  }
}

static std::atomic<int> g_ares_variant{0};
// Test knob: 0 = automatic (the register-resident kernel for the K = 128 consumer pass when the launch fills the chip),
// 1 = the streaming kernel only, 2 = the register-resident kernel whenever the layer is eligible.  Results do not depend
// on it (bit for bit).
extern "C" int mmmot_set_gemm_ares_variant(int v) {
  if (v < 0 || v > 2) return MMMOT_EINVAL;
  g_ares_variant.store(v);
  return MMMOT_OK;
}

// Called by mmmot_gemm_ares after its argument checks; returns 1 when this kernel took the launch.
int mmmot_gemm_wreg128_try(const mmmot_gemm_ares_args* a, int mode, int n_cu, hipStream_t s, int* status) {
  const int variant = g_ares_variant.load();
  if (variant == 1 || a->K != WR_K || mode != 2 || a->N % 512 != 0 || a->N > 4096) return 0;
  if (a->ldx % 4 != 0 || a->ldx < WR_K) return 0;
  const int nh = a->N / 512;
  // the launch must fill the chip with whole (tile sequence x channel half) groups of 8 workgroups per XCD
  if (variant == 0 && (long)a->T * nh < 2L * n_cu) return 0;
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
