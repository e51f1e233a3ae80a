// HBM-bound and small kernels of the mmMOT forward on gfx950: GroupNorm
// statistics combine, ragged/strided segment means, 1-channel output layers,
// per-row LayerNorm, the K=3 PointNet input layer, normalise+activate, fusion
// combine and the dual-softmax.  All use 64-lane wave reductions and 16-byte
// coalesced accesses along the channel axis.
#include "common.h"

// ---------------------------------------------------------------------------
// GroupNorm statistics -> scale/shift.  One workgroup per (group, norm-group).
__global__ __launch_bounds__(256) void gn_finalize_kernel(
    const float* __restrict__ part, const int* __restrict__ grp_tile0, const int* __restrict__ grp_ntiles,
    const int* __restrict__ grp_count, const int* __restrict__ tile_nrows, int ldp, int C, int NG,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float* __restrict__ sc,
    float* __restrict__ sh) {
  __shared__ double red[2][4];
  const int g = blockIdx.x / NG, ng = blockIdx.x % NG;
  const int CG = C / NG, c0 = ng * CG;
  const int tile0 = grp_tile0[g], nt = grp_ntiles[g];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // part[t][0][c] = tile sum S, part[t][1][c] = tile-centred M2 (see gemm_rows.hip epilogue).
  // Tiles of a group are consecutive MM_BM-row chunks, the last one partial - unless tile_nrows says otherwise.
  const int rows = grp_count[g];
  // element idx = tid + 256 k of the group's [nt][CG] statistics: tile idx / CG, channel idx % CG, walked incrementally
  // (a 64-bit division per element made this kernel 0.3 ms on the 128 tiles x 512 channels of a 128 x 128 pair block);
  // every thread adds its elements in the order of the division form: same sums, bit for bit
  const int dq = 256 / CG, dr = 256 % CG;
  const int ti0 = tid / CG, cc0 = tid % CG;
  // U elements per trip: the loop is a chain of HBM round trips on 256 threads per (group, norm group) - with 4 in flight
  // the 128 tiles x 512 channels of a 128 x 128 pair block still took 194 us; the additions stay in element order
  constexpr int U = 16;
  double s1 = 0.0;
  {
    int ti = ti0, cc = cc0;
    while (ti < nt) {
      float v[U];
      int n = 0;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        v[u] = 0.f;
        if (ti < nt) {
          v[u] = part[((long)(tile0 + ti) * 2 + 0) * ldp + c0 + cc];
          n = u + 1;
          ti += dq;
          cc += dr;
          if (cc >= CG) { cc -= CG; ++ti; }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (u < n) s1 += (double)v[u];
    }
  }
  s1 = wave_sum_d(s1);
  if (lane == 0) red[0][wave] = s1;
  __syncthreads();
  const double cnt = (double)rows * (double)CG;
  const double mean = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / cnt;
  // Chan et al.: M2 = sum_chunks [ M2_chunk + n_chunk (mean_chunk - mean)^2 ], chunk = (tile, channel)
  double s2 = 0.0;
  {
    int ti = ti0, cc = cc0;
    while (ti < nt) {
      float vs[U], vm[U];
      float nn[U];
      int n = 0;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        vs[u] = vm[u] = 0.f;
        nn[u] = 0.f;
        if (ti < nt) {
          const int t = tile0 + ti;
          const int left = rows - ti * MM_BM;
          nn[u] = tile_nrows ? (float)tile_nrows[t] : (float)(left < MM_BM ? left : MM_BM);  // a row count: exact in fp32
          vs[u] = part[((long)t * 2 + 0) * ldp + c0 + cc];
          vm[u] = part[((long)t * 2 + 1) * ldp + c0 + cc];
          n = u + 1;
          ti += dq;
          cc += dr;
          if (cc >= CG) { cc -= CG; ++ti; }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (u < n && nn[u] > 0.f) {  // half tiles of the A-resident GEMM may be empty
          const double n_t = (double)nn[u];
          const double d = (double)vs[u] / n_t - mean;
          s2 += (double)vm[u] + n_t * d * d;
        }
    }
  }
  s2 = wave_sum_d(s2);
  if (lane == 0) red[1][wave] = s2;
  __syncthreads();
  const double var = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) / cnt;
  const double rstd = 1.0 / sqrt(var + (double)eps);
  for (int c = c0 + tid; c < c0 + CG; c += 256) {
    const double scv = (double)gamma[c] * rstd;
    sc[(long)g * C + c] = (float)scv;
    sh[(long)g * C + c] = (float)((double)beta[c] - mean * scv);
  }
}

// Per-channel variant (NG == C: nn.GroupNorm(C, C), every PointNet layer) for long tile lists: one
// workgroup per (group, 16 channels); 16 lanes read 64 contiguous bytes of a tile row, 16 tile stripes run
// in parallel, ONE pass: S = sum S_t, Q = sum (M2_t + S_t^2 / n_t) in fp64, var = (Q - S^2/n) / n.
// (The generic kernel walks the tiles once per channel with an 8 KB stride: 0.5 ms for the 16 384 half
// tiles x 1024 channels of PointNet conv5 at 4 cfg3 pairs.)
__global__ __launch_bounds__(1024) void gn_finalize_perchannel_kernel(
    const float* __restrict__ part, const int* __restrict__ grp_tile0, const int* __restrict__ grp_ntiles,
    const int* __restrict__ grp_count, const int* __restrict__ tile_nrows, int ldp, int C,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float* __restrict__ sc,
    float* __restrict__ sh) {
  // 1024 threads = 64 tile stripes x 16 channels: the loop is a chain of dependent-latency trips (2 048 tiles per
  // group for a PointNet layer at cfg3), so the trip count is what matters, not the arithmetic
  __shared__ double red[2][64][16];
  const int g = blockIdx.x, c = blockIdx.y * 16 + (threadIdx.x & 15);
  const int stripe = threadIdx.x >> 4;
  const int tile0 = grp_tile0[g], nt = grp_ntiles[g], rows = grp_count[g];
  double S = 0.0, Q = 0.0;
  auto rows_of = [&](int ti) -> int {
    if (ti >= nt) return 0;
    const int left = rows - ti * MM_BM;
    return tile_nrows ? tile_nrows[tile0 + ti] : (left < MM_BM ? left : MM_BM);
  };
  // 4 tiles per trip: 12 independent loads in flight per lane (the loop is latency-bound otherwise)
  for (int ti = stripe; ti < nt; ti += 256) {
    int n_t[4];
    float s_t[4], m_t[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) n_t[u] = rows_of(ti + 64 * u);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long t = tile0 + min(ti + 64 * u, nt - 1);
      s_t[u] = part[(t * 2 + 0) * ldp + c];
      m_t[u] = part[(t * 2 + 1) * ldp + c];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (n_t[u] > 0) {
        S += (double)s_t[u];
        Q += (double)m_t[u] + (double)s_t[u] * (double)s_t[u] / (double)n_t[u];
      }
  }
  red[0][stripe][threadIdx.x & 15] = S;
  red[1][stripe][threadIdx.x & 15] = Q;
  __syncthreads();
  if (threadIdx.x < 16) {
    double s = 0.0, q = 0.0;
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      s += red[0][i][threadIdx.x];
      q += red[1][i][threadIdx.x];
    }
    const double cnt = (double)rows;
    const double mean = s / cnt;
    double var = (q - s * mean) / cnt;
    var = var > 0.0 ? var : 0.0;
    const double scv = (double)gamma[c] / sqrt(var + (double)eps);
    sc[(long)g * C + c] = (float)scv;
    sh[(long)g * C + c] = (float)((double)beta[c] - mean * scv);
  }
}

extern "C" int mmmot_gn_finalize(const float* part, const int* grp_tile0, const int* grp_ntiles,
                                 const int* grp_count, const int* tile_nrows, int G, int ldp, int C, int NG,
                                 const float* gamma, const float* beta, float eps, float* sc, float* sh,
                                 void* stream) {
  if (!part || !grp_tile0 || !grp_ntiles || !grp_count || !gamma || !beta || !sc || !sh) return MMMOT_EINVAL;
  if (G <= 0 || C <= 0 || NG <= 0 || C % NG != 0 || ldp < C) return MMMOT_EINVAL;
  if (NG == C && C % 16 == 0) {
    hipLaunchKernelGGL(gn_finalize_perchannel_kernel, dim3(G, C / 16), dim3(1024), 0, (hipStream_t)stream, part,
                       grp_tile0, grp_ntiles, grp_count, tile_nrows, ldp, C, gamma, beta, eps, sc, sh);
    return mm_check(hipGetLastError());
  }
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(G * NG), dim3(256), 0, (hipStream_t)stream, part, grp_tile0,
                     grp_ntiles, grp_count, tile_nrows, ldp, C, NG, gamma, beta, eps, sc, sh);
  return mm_check(hipGetLastError());
}

// ---------------------------------------------------------------------------
// Strided segment mean.  grid = (nseg, ceil(C/256)); 4 waves split the rows of
// the segment, every lane owns 4 consecutive channels (1 KiB per wave load).
__global__ __launch_bounds__(256) void segment_mean_kernel(
    const float* __restrict__ X, int ldx, int C, const int* __restrict__ seg_start,
    const int* __restrict__ seg_count, const int* __restrict__ seg_stride, const int* __restrict__ seg_group,
    const int* __restrict__ seg_div, const float* __restrict__ sc, const float* __restrict__ sh, int ldsc, int flags,
    float* __restrict__ out, int ldo, int hl16) {
  const int relu = flags & 1;
  const bool take_max = (flags & 2) != 0;  // maximum over the segment instead of the mean (new_end.py:72-74)
  __shared__ __attribute__((aligned(16))) float red[4][256];
  const int s = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = blockIdx.y * 256 + lane * 4;
  const bool cv = c < C;
  const int start = seg_start[s], count = seg_count[s], stride = seg_stride ? seg_stride[s] : 1;
  const int g = seg_group ? seg_group[s] : 0;
  f32x4 s4 = {1.f, 1.f, 1.f, 1.f}, h4 = {0.f, 0.f, 0.f, 0.f};
  const bool norm = (sc != nullptr);
  if (norm && cv) {
    s4 = *reinterpret_cast<const f32x4*>(&sc[(long)g * ldsc + c]);
    h4 = *reinterpret_cast<const f32x4*>(&sh[(long)g * ldsc + c]);
  }
  const float a0 = take_max ? -3.0e38f : 0.f;
  f32x4 acc0 = {a0, a0, a0, a0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
  auto ld = [&](int t) -> f32x4 {
    f32x4 v;
    if (hl16 == 2) {
      // hq8 row: 128-byte record per 32 channels = [32 x fp16 hi | 32 x e4m3 (unused here) | 32 x e4m3(lo * 2^9)]
      const unsigned char* rec = reinterpret_cast<const unsigned char*>(X + ((long)start + (long)t * stride) * ldx) +
                                 (c >> 5) * 128;
      typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
      const f16x4 h = *reinterpret_cast<const f16x4*>(rec + 2 * (c & 31));
      const int l = *reinterpret_cast<const int*>(rec + 96 + (c & 31));
      const auto p0 = __builtin_amdgcn_cvt_pk_f32_fp8(l, false);
      const auto p1 = __builtin_amdgcn_cvt_pk_f32_fp8(l, true);
      v[0] = (float)h[0] + p0[0] * (1.f / 512.f);
      v[1] = (float)h[1] + p0[1] * (1.f / 512.f);
      v[2] = (float)h[2] + p1[0] * (1.f / 512.f);
      v[3] = (float)h[3] + p1[1] * (1.f / 512.f);
    } else if (hl16) {
      // hl16 row (same bytes as fp32): unit u = c>>3 holds [hi8 | lo8] halves; this lane's 4 channels
      const _Float16* rowp = reinterpret_cast<const _Float16*>(X + ((long)start + (long)t * stride) * ldx);
      typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
      const f16x4 h = *reinterpret_cast<const f16x4*>(rowp + (c >> 3) * 16 + (c & 7));
      const f16x4 l = *reinterpret_cast<const f16x4*>(rowp + (c >> 3) * 16 + 8 + (c & 7));
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (float)h[e] + (float)l[e];
    } else {
      v = *reinterpret_cast<const f32x4*>(&X[((long)start + (long)t * stride) * ldx + c]);
    }
    if (norm) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], s4[e], h4[e]);
    }
    if (relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    return v;
  };
  auto mx4 = [](f32x4 a, f32x4 b) -> f32x4 {
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = fmaxf(a[e], b[e]);
    return r;
  };
  if (cv) {
    int t = wave;
    if (take_max) {
      for (; t < count; t += 4) acc0 = mx4(acc0, ld(t));
    } else {
      for (; t + 12 < count; t += 16) {
        const f32x4 v0 = ld(t), v1 = ld(t + 4), v2 = ld(t + 8), v3 = ld(t + 12);
        acc0 += v0; acc1 += v1; acc2 += v2; acc3 += v3;
      }
      for (; t < count; t += 4) acc0 += ld(t);
    }
  }
  const f32x4 acc = take_max ? acc0 : (acc0 + acc1) + (acc2 + acc3);
  *reinterpret_cast<f32x4*>(&red[wave][lane * 4]) = acc;
  __syncthreads();
  if (wave == 0 && cv) {
    f32x4 r = *reinterpret_cast<const f32x4*>(&red[0][lane * 4]);
    if (take_max) {
      r = mx4(mx4(r, *reinterpret_cast<const f32x4*>(&red[1][lane * 4])),
              mx4(*reinterpret_cast<const f32x4*>(&red[2][lane * 4]), *reinterpret_cast<const f32x4*>(&red[3][lane * 4])));
    } else {
      r += *reinterpret_cast<const f32x4*>(&red[1][lane * 4]);
      r += *reinterpret_cast<const f32x4*>(&red[2][lane * 4]);
      r += *reinterpret_cast<const f32x4*>(&red[3][lane * 4]);
      const float inv = 1.f / (float)(seg_div ? seg_div[s] : count);
#pragma unroll
      for (int e = 0; e < 4; ++e) r[e] *= inv;
    }
    *reinterpret_cast<f32x4*>(&out[(long)s * ldo + c]) = r;
  }
}

// hl16 rows without a normalise / ReLU prologue (the SkipPool global average pools over the trunk's stage outputs, 2 GB
// per 8-pair step): a lane takes a whole 8-channel unit - [hi8 | lo8] = two 16-byte loads - instead of the 4 channels
// (two 8-byte loads) of the general kernel, and all 64 lanes work whatever C is: UPR = C / 8 lanes cover a row, a wave
// load instruction 64 / UPR rows.  Same result contract (mean over the segment, divisor seg_div or the row count).
typedef _Float16 sm_f16x8 __attribute__((ext_vector_type(8)));
template <int UPR>
__global__ __launch_bounds__(256) void segment_mean_hl16_kernel(const float* __restrict__ X, int ldx,
                                                                 const int* __restrict__ seg_start,
                                                                 const int* __restrict__ seg_count,
                                                                 const int* __restrict__ seg_stride,
                                                                 const int* __restrict__ seg_div,
                                                                 float* __restrict__ out, int ldo) {
  constexpr int RPW = 64 / UPR;  // rows per wave load instruction
  __shared__ float red[4][UPR * 8];
  const int s = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int u = lane % UPR, rsub = lane / UPR;
  const int start = seg_start[s], count = seg_count[s], stride = seg_stride ? seg_stride[s] : 1;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  auto add_row = [&](int t) {
    const sm_f16x8* p = reinterpret_cast<const sm_f16x8*>(X + ((long)start + (long)t * stride) * ldx) + 2 * u;
    const sm_f16x8 h = p[0], l = p[1];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += (float)h[e] + (float)l[e];
  };
  constexpr int STEP = 4 * RPW;  // rows per workgroup pass
  int t = wave * RPW + rsub;
  for (; t + 3 * STEP < count; t += 4 * STEP) {  // four independent row loads in flight per lane
    add_row(t);
    add_row(t + STEP);
    add_row(t + 2 * STEP);
    add_row(t + 3 * STEP);
  }
  for (; t < count; t += STEP) add_row(t);
  // lanes u, u + UPR, ... of a wave hold partial sums of the same unit
#pragma unroll
  for (int o = UPR; o < 64; o <<= 1)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += __shfl_xor(acc[e], o);
  if (rsub == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) red[wave][u * 8 + e] = acc[e];
  }
  __syncthreads();
  const float inv = 1.f / (float)(seg_div ? seg_div[s] : count);
  for (int c = tid; c < UPR * 8; c += 256)
    out[(long)s * ldo + c] = ((red[0][c] + red[1][c]) + (red[2][c] + red[3][c])) * inv;
}

extern "C" int mmmot_segment_mean(const float* X, int ldx, int C, const int* seg_start, const int* seg_count,
                                  const int* seg_stride, const int* seg_group, const int* seg_div, int nseg,
                                  const float* sc, const float* sh, int ldsc, int relu, float* out, int ldo,
                                  int hl16, void* stream) {
  if (!X || !seg_start || !seg_count || !out || nseg <= 0 || C <= 0) return MMMOT_EINVAL;
  if (hl16 && (C % 8 != 0 || ldx % 8 != 0)) return MMMOT_EINVAL;
  if (hl16 == 2 && (C % 32 != 0 || ldx % 32 != 0)) return MMMOT_EINVAL;
  if (hl16 < 0 || hl16 > 2 || relu < 0 || relu > 3) return MMMOT_EINVAL;
  if (C % 4 != 0 || ldx % 4 != 0 || ldo % 4 != 0 || !mm_al16(X) || !mm_al16(out)) return MMMOT_EINVAL;
  if ((sc == nullptr) != (sh == nullptr)) return MMMOT_EINVAL;
  if (sc && (ldsc % 4 != 0 || !mm_al16(sc) || !mm_al16(sh))) return MMMOT_EINVAL;
  if (hl16 == 1 && !sc && relu == 0 && (C == 128 || C == 256 || C == 512) && mm_al16(X) && ldx % 8 == 0) {
    hipStream_t st = (hipStream_t)stream;
    if (C == 128)
      hipLaunchKernelGGL(segment_mean_hl16_kernel<16>, dim3(nseg), dim3(256), 0, st, X, ldx, seg_start, seg_count,
                         seg_stride, seg_div, out, ldo);
    else if (C == 256)
      hipLaunchKernelGGL(segment_mean_hl16_kernel<32>, dim3(nseg), dim3(256), 0, st, X, ldx, seg_start, seg_count,
                         seg_stride, seg_div, out, ldo);
    else
      hipLaunchKernelGGL(segment_mean_hl16_kernel<64>, dim3(nseg), dim3(256), 0, st, X, ldx, seg_start, seg_count,
                         seg_stride, seg_div, out, ldo);
    return mm_check(hipGetLastError());
  }
  hipLaunchKernelGGL(segment_mean_kernel, dim3(nseg, (C + 255) / 256), dim3(256), 0, (hipStream_t)stream, X,
                     ldx, C, seg_start, seg_count, seg_stride, seg_group, seg_div, sc, sh, ldsc, relu, out, ldo, hl16);
  return mm_check(hipGetLastError());
}

// ---------------------------------------------------------------------------
// One-channel output layer: one wave per row, lanes stride the channel axis.
__global__ __launch_bounds__(256) void rowdot_kernel(
    const float* __restrict__ X, int ldx, int K, const float* __restrict__ w, float b,
    const float* __restrict__ sc, const float* __restrict__ sh, int ldsc, const int* __restrict__ tile_row0,
    const int* __restrict__ tile_nrows, const int* __restrict__ tile_group, int act, int use_thr, float thr,
    float* __restrict__ out, const int* __restrict__ omap) {
  const int t = blockIdx.x;
  const int row0 = tile_row0[t], nrows = tile_nrows[t];
  const int g = tile_group ? tile_group[t] : 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool norm = (sc != nullptr);
  if (K <= 128) {
    // short rows (the 128 -> 1 layers over the N x M pair rows: 0.8 GB per cfg4 step): a row is 32 lanes x 16 bytes, so
    // the two halves of a wave take two rows at a time and the weight / scale / shift vectors are read once per
    // workgroup, not once per row.  Per row the sum is the one of the general loop below, bit for bit: there the upper
    // half holds zeros and the first butterfly step adds them (x + 0 = x); here that step is skipped.
    const int hl = lane & 31, hi = lane >> 5, k = hl * 4;
    const bool live = k < K;
    f32x4 wv = {0.f, 0.f, 0.f, 0.f}, s4 = wv, h4 = wv;
    if (live) {
      wv = *reinterpret_cast<const f32x4*>(w + k);
      if (norm) {
        s4 = *reinterpret_cast<const f32x4*>(&sc[(long)g * ldsc + k]);
        h4 = *reinterpret_cast<const f32x4*>(&sh[(long)g * ldsc + k]);
      }
    }
    for (int r0 = 2 * wave; r0 < nrows; r0 += 8) {
      const int r = r0 + hi;
      float acc = 0.f;
      if (live && r < nrows) {
        f32x4 v = *reinterpret_cast<const f32x4*>(X + (long)(row0 + r) * ldx + k);
        if (norm) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(v[e], s4[e], h4[e]), 0.f);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = fmaf(v[e], wv[e], acc);
      }
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
      float s = acc + b;
      s = mm_act(s, act);
      if (use_thr && s < thr) s -= 1.f;
      if (hl == 0 && r < nrows) out[omap ? omap[row0 + r] : row0 + r] = s;
    }
    return;
  }
  for (int r = wave; r < nrows; r += 4) {
    const float* xr = X + (long)(row0 + r) * ldx;
    float acc = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
      f32x4 v = *reinterpret_cast<const f32x4*>(xr + k);
      const f32x4 wv = *reinterpret_cast<const f32x4*>(w + k);
      if (norm) {
        const f32x4 s4 = *reinterpret_cast<const f32x4*>(&sc[(long)g * ldsc + k]);
        const f32x4 h4 = *reinterpret_cast<const f32x4*>(&sh[(long)g * ldsc + k]);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(v[e], s4[e], h4[e]), 0.f);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = fmaf(v[e], wv[e], acc);
    }
    float s = wave_sum(acc) + b;
    s = mm_act(s, act);
    if (use_thr && s < thr) s -= 1.f;
    if (lane == 0) out[omap ? omap[row0 + r] : row0 + r] = s;
  }
}

extern "C" int mmmot_rowdot(const float* X, int ldx, int K, const float* w, float b, const float* sc,
                            const float* sh, int ldsc, const int* tile_row0, const int* tile_nrows,
                            const int* tile_group, int T, int act, int use_thr, float thr, float* out,
                            const int* omap, void* stream) {
  if (!X || !w || !tile_row0 || !tile_nrows || !out || T <= 0) return MMMOT_EINVAL;
  if (K % 4 != 0 || ldx % 4 != 0 || !mm_al16(X) || !mm_al16(w)) return MMMOT_EINVAL;
  if ((sc == nullptr) != (sh == nullptr)) return MMMOT_EINVAL;
  if (sc && (ldsc % 4 != 0 || !mm_al16(sc) || !mm_al16(sh))) return MMMOT_EINVAL;
  hipLaunchKernelGGL(rowdot_kernel, dim3(T), dim3(256), 0, (hipStream_t)stream, X, ldx, K, w, b, sc, sh, ldsc,
                     tile_row0, tile_nrows, tile_group, act, use_thr, thr, out, omap);
  return mm_check(hipGetLastError());
}

// ---------------------------------------------------------------------------
// Per-row LayerNorm: one wave per row, the row lives in registers (C <= 1024).
__global__ __launch_bounds__(256) void row_layernorm_kernel(
    const float* __restrict__ X, int ldx, int C, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, int relu, float* __restrict__ Y, int ldy, int R) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + wave;
  if (r >= R) return;
  const int per = C >> 6;  // values per lane, <= 16
  float v[16];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    v[i] = 0.f;
    if (i < per) { v[i] = X[(long)r * ldx + lane + 64 * i]; s += v[i]; }
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i)
    if (i < per) { const float d = v[i] - mean; q += d * d; }
  const float var = wave_sum(q) / (float)C;
  const float rstd = 1.f / sqrtf(var + eps);
#pragma unroll
  for (int i = 0; i < 16; ++i)
    if (i < per) {
      const int c = lane + 64 * i;
      float y = (v[i] - mean) * rstd * gamma[c] + beta[c];
      if (relu) y = fmaxf(y, 0.f);
      Y[(long)r * ldy + c] = y;
    }
}

extern "C" int mmmot_row_layernorm(const float* X, int ldx, int C, const float* gamma, const float* beta,
                                   float eps, int relu, float* Y, int ldy, int R, void* stream) {
  if (!X || !gamma || !beta || !Y || R <= 0) return MMMOT_EINVAL;
  if (C <= 0 || C % 64 != 0 || C > 1024) return MMMOT_EINVAL;
  hipLaunchKernelGGL(row_layernorm_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, X, ldx, C,
                     gamma, beta, eps, relu, Y, ldy, R);
  return mm_check(hipGetLastError());
}

// ---------------------------------------------------------------------------
// One SkipPool head (reference modules/appear_net.py:19-32) for 4 detections per workgroup, in ONE launch:
//   LayerNorm(C) -> 1x1 conv C -> C4 -> LayerNorm + ReLU -> 1x1 conv C4 -> 128 -> LayerNorm + ReLU
// on the pooled [R][C] features of a stage (the global average pool before it stays mmmot_segment_mean).  Seven launches
// (three row LayerNorms, two small-M GEMMs of 4-8 workgroups, ...) per stage were ~0.4 ms of every step and a third of
// the launches of a one-pair forward.  Plain fp32: wave w normalises row w; the matrix-vector products walk the
// output channels (wave w: j = w, w + 4, ...), lanes along k (coalesced 16-byte weight loads, each weight row read
// once for the workgroup's rows), wave reduction per (row, j).  SP_ROWS = 2 rows per workgroup: a one-pair forward
// (~22 detections) still spreads over a dozen CUs, a 2048-detection batch reads the 0.8 MB of head weights 1024 times
// from L2 (0.8 GB: tens of microseconds).
#define SP_ROWS 2
#define SP_JB 4
__device__ __forceinline__ void sp_layernorm_row(const float* __restrict__ x, int C, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, float eps, bool relu, float* y, int lane) {
  float v[8];
  const int per = C >> 6;  // <= 8
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    v[i] = 0.f;
    if (i < per) { v[i] = x[lane + 64 * i]; s += v[i]; }
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (i < per) { const float d = v[i] - mean; q += d * d; }
  const float rstd = 1.f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (i < per) {
      const int c = lane + 64 * i;
      const float o = (v[i] - mean) * rstd * gamma[c] + beta[c];
      y[c] = relu ? fmaxf(o, 0.f) : o;
    }
}

// out[rr][j] = bias[j] + sum_k W[j][k] * x[rr][k] for the workgroup's ROWS rows (x, out in LDS).  Lane l owns
// k = l, l + 64, ... (coalesced 256-byte weight loads); a wave takes JB output channels at a time so that all
// their weight loads are in flight together (with few detections the kernel is a chain of L2 round trips, not work).
// ROWS x JB = 64 (the batch variant, 8 x 8): the 64 lane-partial sums of a step are reduced TOGETHER by a transposing
// butterfly - at offset o a lane keeps the half of its partials whose index bit matches its own lane bit and adds the
// partner's copy of them - 63 exchanges for 64 sums instead of 6 per sum.  Per (row, channel) the additions are those
// of wave_sum (same pairs, same order: the two partners of an exchange add the same two numbers), so the result is
// bit for bit the one of the per-sum butterfly, whatever ROWS is: a detection scores the same alone and in a batch.
template <int ROWS, int JB>
__device__ __forceinline__ void sp_matvec(const float* __restrict__ W, const float* __restrict__ bias, int N, int K,
                                          const float (*x)[512], float (*out)[128], int lane, int wave) {
  const int per = K >> 6;  // <= 8
  for (int j0 = wave * JB; j0 < N; j0 += 4 * JB) {
    float w[JB][8];
#pragma unroll
    for (int jb = 0; jb < JB; ++jb)
#pragma unroll
      for (int i = 0; i < 8; ++i) w[jb][i] = (i < per && j0 + jb < N) ? W[(long)(j0 + jb) * K + lane + 64 * i] : 0.f;
    float acc[JB * ROWS];  // index jb * ROWS + rr
#pragma unroll
    for (int a = 0; a < JB * ROWS; ++a) acc[a] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i < per) {
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) {
          const float xv = x[rr][lane + 64 * i];
#pragma unroll
          for (int jb = 0; jb < JB; ++jb) acc[jb * ROWS + rr] = fmaf(w[jb][i], xv, acc[jb * ROWS + rr]);
        }
      }
    if constexpr (JB * ROWS == 64) {
      // transposing butterfly: after the step at offset o, slot i of a lane holds partial index i + (lane & o ? o : 0) + ...
      // and finally slot 0 of lane l holds the complete sum of partial index l
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const bool up = (lane & o) != 0;
#pragma unroll
        for (int i = 0; i < o; ++i) {
          const float keep = up ? acc[i + o] : acc[i];
          const float send = up ? acc[i] : acc[i + o];
          acc[i] = keep + __shfl_xor(send, o);
        }
      }
      const int jb = lane / ROWS, rr = lane % ROWS;
      if (j0 + jb < N) out[rr][j0 + jb] = acc[0] + bias[j0 + jb];
    } else {
#pragma unroll
      for (int jb = 0; jb < JB; ++jb)
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) {
          const float t = wave_sum(acc[jb * ROWS + rr]);
          if (lane == 0 && j0 + jb < N) out[rr][j0 + jb] = t + bias[j0 + jb];
        }
    }
  }
}

// ROWS = 2 (SP_ROWS): the latency form - a one-pair forward (~22 detections) still spreads over a dozen CUs.  ROWS = 8:
// the batch form (R >= 256 rows): the 0.3 MB of head weights are read once per 8 rows instead of once per 2 and the
// reductions are shared (sp_matvec); every wave normalises ROWS / 4 rows.
template <int ROWS>
__global__ __launch_bounds__(256) void skippool_head_kernel(
    const float* __restrict__ P, int ldp, int C, int C4, const float* __restrict__ g0, const float* __restrict__ b0,
    const float* __restrict__ w1, const float* __restrict__ c1, const float* __restrict__ g2,
    const float* __restrict__ b2, const float* __restrict__ w4, const float* __restrict__ c4,
    const float* __restrict__ g5, const float* __restrict__ b5, float eps, float* __restrict__ out, int ldo, int R) {
  constexpr int JB = ROWS == 8 ? 8 : SP_JB;
  constexpr int RW = ROWS < 4 ? 1 : ROWS / 4;  // rows a wave normalises
  __shared__ __attribute__((aligned(16))) float xs[ROWS][512];
  __shared__ __attribute__((aligned(16))) float hs[ROWS][512];  // normalised hidden rows (C4 <= 128 used)
  __shared__ float h1[ROWS][128];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row0 = blockIdx.x * ROWS;
#pragma unroll
  for (int q = 0; q < RW; ++q) {
    const int lr = wave * RW + q;  // ROWS = 2: waves 0, 1 normalise a row each; all four waves do the products
    if (lr < ROWS) sp_layernorm_row(P + (long)(row0 + lr < R ? row0 + lr : 0) * ldp, C, g0, b0, eps, false, xs[lr], lane);
  }
  __syncthreads();
  sp_matvec<ROWS, JB>(w1, c1, C4, C, xs, h1, lane, wave);
  __syncthreads();
#pragma unroll
  for (int q = 0; q < RW; ++q) {
    const int lr = wave * RW + q;
    if (lr < ROWS) sp_layernorm_row(h1[lr], C4, g2, b2, eps, true, hs[lr], lane);
  }
  __syncthreads();
  sp_matvec<ROWS, JB>(w4, c4, 128, C4, hs, h1, lane, wave);
  __syncthreads();
#pragma unroll
  for (int q = 0; q < RW; ++q) {
    const int lr = wave * RW + q;
    if (lr < ROWS && row0 + lr < R) sp_layernorm_row(h1[lr], 128, g5, b5, eps, true, out + (long)(row0 + lr) * ldo, lane);
  }
}

extern "C" int mmmot_skippool_head(const float* P, int ldp, int C, int C4, const float* g0, const float* b0,
                                   const float* w1, const float* c1, const float* g2, const float* b2, const float* w4,
                                   const float* c4, const float* g5, const float* b5, float eps, float* out, int ldo,
                                   int R, void* stream) {
  if (!P || !g0 || !b0 || !w1 || !c1 || !g2 || !b2 || !w4 || !c4 || !g5 || !b5 || !out || R <= 0) return MMMOT_EINVAL;
  if (C <= 0 || C % 64 != 0 || C > 512 || C4 <= 0 || C4 % 64 != 0 || C4 > 128) return MMMOT_EINVAL;
  if (!mm_al16(w1) || !mm_al16(w4)) return MMMOT_EINVAL;
  if (R >= 256)
    hipLaunchKernelGGL(skippool_head_kernel<8>, dim3((R + 7) / 8), dim3(256), 0, (hipStream_t)stream, P, ldp,
                       C, C4, g0, b0, w1, c1, g2, b2, w4, c4, g5, b5, eps, out, ldo, R);
  else
    hipLaunchKernelGGL(skippool_head_kernel<SP_ROWS>, dim3((R + SP_ROWS - 1) / SP_ROWS), dim3(256), 0, (hipStream_t)stream,
                       P, ldp, C, C4, g0, b0, w1, c1, g2, b2, w4, c4, g5, b5, eps, out, ldo, R);
  return mm_check(hipGetLastError());
}

// ---------------------------------------------------------------------------
// PointNet first shared-MLP layer (K = 3 xyz, or 4 with the reflectivity channel): VALU, output-write bound.
template <int K>
__global__ __launch_bounds__(256) void pointnet_layer1_kernel(
    const float* __restrict__ X, const float* __restrict__ W, const float* __restrict__ bias,
    float* __restrict__ Y, float* __restrict__ part, const int* __restrict__ tile_row0,
    const int* __restrict__ tile_nrows) {
  __shared__ float xs[MM_BM * K];
  __shared__ float red[4][2][64];
  const int t = blockIdx.x;
  const int row0 = tile_row0[t], nrows = tile_nrows[t];
  const int tid = threadIdx.x;
  for (int idx = tid; idx < nrows * K; idx += 256) xs[idx] = X[(long)row0 * K + idx];
  __syncthreads();
  const int c = tid & 63, rq = tid >> 6;
  float w[K];
#pragma unroll
  for (int k = 0; k < K; ++k) w[k] = W[c * K + k];
  const float bv = bias[c];
  auto val = [&](int r) {
    float y = bv;
#pragma unroll
    for (int k = 0; k < K; ++k) y = fmaf(w[k], xs[r * K + k], y);  // same order as the K = 3 kernel of round 1
    return y;
  };
  float s1 = 0.f;
  for (int i = 0; i < 32; ++i) {
    const int r = rq * 32 + i;
    if (r < nrows) {
      const float y = val(r);
      Y[(long)(row0 + r) * 64 + c] = y;
      s1 += y;
    }
  }
  red[rq][0][c] = s1;
  __syncthreads();
  // tile-centred second moment (same statistics contract as mmmot_gemm_rows)
  const float tot = red[0][0][c] + red[1][0][c] + red[2][0][c] + red[3][0][c];
  const float mu = tot / (float)nrows;
  float s2 = 0.f;
  for (int i = 0; i < 32; ++i) {
    const int r = rq * 32 + i;
    if (r < nrows) {
      const float d = val(r) - mu;
      s2 += d * d;
    }
  }
  red[rq][1][c] = s2;
  __syncthreads();
  if (tid < 64) {
    part[((long)t * 2 + 0) * 64 + c] = tot;
    part[((long)t * 2 + 1) * 64 + c] = red[0][1][c] + red[1][1][c] + red[2][1][c] + red[3][1][c];
  }
}

extern "C" int mmmot_pointnet_layer1(const float* X, int K, const float* W, const float* bias, float* Y, float* part,
                                     const int* tile_row0, const int* tile_nrows, int T, void* stream) {
  if (!X || !W || !bias || !Y || !part || !tile_row0 || !tile_nrows || T <= 0 || (K != 3 && K != 4)) return MMMOT_EINVAL;
  if (K == 3)
    hipLaunchKernelGGL(pointnet_layer1_kernel<3>, dim3(T), dim3(256), 0, (hipStream_t)stream, X, W, bias, Y, part,
                       tile_row0, tile_nrows);
  else
    hipLaunchKernelGGL(pointnet_layer1_kernel<4>, dim3(T), dim3(256), 0, (hipStream_t)stream, X, W, bias, Y, part,
                       tile_row0, tile_nrows);
  return mm_check(hipGetLastError());
}

// ---------------------------------------------------------------------------
// Normalise (+ activation) rows with the group's scale/shift.
__global__ __launch_bounds__(256) void affine_act_kernel(
    const float* __restrict__ X, int ldx, int C, const float* __restrict__ sc, const float* __restrict__ sh,
    int ldsc, const int* __restrict__ tile_row0, const int* __restrict__ tile_nrows,
    const int* __restrict__ tile_group, int act, float* __restrict__ Y, int ldy) {
  const int t = blockIdx.x;
  const int row0 = tile_row0[t], nrows = tile_nrows[t];
  const int g = tile_group ? tile_group[t] : 0;
  const int C4 = C >> 2;
  for (int idx = threadIdx.x + 256 * blockIdx.y; idx < nrows * C4; idx += 256 * gridDim.y) {
    const int r = idx / C4, c = (idx - r * C4) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(&X[(long)(row0 + r) * ldx + c]);
    const f32x4 s4 = *reinterpret_cast<const f32x4*>(&sc[(long)g * ldsc + c]);
    const f32x4 h4 = *reinterpret_cast<const f32x4*>(&sh[(long)g * ldsc + c]);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = mm_act(fmaf(v[e], s4[e], h4[e]), act);
    *reinterpret_cast<f32x4*>(&Y[(long)(row0 + r) * ldy + c]) = v;
  }
}

extern "C" int mmmot_affine_act(const float* X, int ldx, int C, const float* sc, const float* sh, int ldsc,
                                const int* tile_row0, const int* tile_nrows, const int* tile_group, int T,
                                int act, float* Y, int ldy, void* stream) {
  if (!X || !sc || !sh || !tile_row0 || !tile_nrows || !Y || T <= 0) return MMMOT_EINVAL;
  if (C % 4 != 0 || ldx % 4 != 0 || ldy % 4 != 0 || ldsc % 4 != 0) return MMMOT_EINVAL;
  if (!mm_al16(X) || !mm_al16(Y) || !mm_al16(sc) || !mm_al16(sh)) return MMMOT_EINVAL;
  hipLaunchKernelGGL(affine_act_kernel, dim3(T, T < 256 ? 8 : 1), dim3(256), 0, (hipStream_t)stream, X, ldx, C, sc, sh, ldsc,
                     tile_row0, tile_nrows, tile_group, act, Y, ldy);
  return mm_check(hipGetLastError());
}

// ---------------------------------------------------------------------------
// Fusion module A/B/C combine -> F[3][Lt][C] (image, lidar, fused).
__global__ __launch_bounds__(256) void fusion_combine_kernel(
    int mode, const float* __restrict__ cat, const float* __restrict__ Y0, int ld0,
    const float* __restrict__ Y1, int ld1, const float* __restrict__ sc0, const float* __restrict__ sh0,
    const float* __restrict__ sc1, const float* __restrict__ sh1, int ldsc, const int* __restrict__ tile_row0,
    const int* __restrict__ tile_nrows, const int* __restrict__ tile_group, float* __restrict__ F, int Lt,
    int C) {
  const int t = blockIdx.x;
  const int row0 = tile_row0[t], nrows = tile_nrows[t];
  const int g = tile_group ? tile_group[t] : 0;
  const int C4 = C >> 2;
  // gridDim.y workgroups share a tile: one workgroup per 128-row tile is 64 dependent-latency trips on 8 CUs
  for (int idx = threadIdx.x + 256 * blockIdx.y; idx < nrows * C4; idx += 256 * gridDim.y) {
    const int r = idx / C4, c = (idx - r * C4) * 4;
    const long d = row0 + r;
    const f32x4 fi = *reinterpret_cast<const f32x4*>(&cat[d * 2 * C + c]);
    const f32x4 fl = *reinterpret_cast<const f32x4*>(&cat[d * 2 * C + C + c]);
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(&sc0[(long)g * ldsc + c]);
    const f32x4 h0 = *reinterpret_cast<const f32x4*>(&sh0[(long)g * ldsc + c]);
    f32x4 fused;
    if (mode == MMMOT_FUSION_A) {
      const f32x4 y0 = *reinterpret_cast<const f32x4*>(&Y0[d * ld0 + c]);
#pragma unroll
      for (int e = 0; e < 4; ++e) fused[e] = fmaf(y0[e], s0[e], h0[e]);
    } else {
      const f32x4 s1 = *reinterpret_cast<const f32x4*>(&sc1[(long)g * ldsc + c]);
      const f32x4 h1 = *reinterpret_cast<const f32x4*>(&sh1[(long)g * ldsc + c]);
      if (mode == MMMOT_FUSION_B) {
        const f32x4 y0 = *reinterpret_cast<const f32x4*>(&Y0[d * ld0 + c]);
        const f32x4 y1 = *reinterpret_cast<const f32x4*>(&Y1[d * ld1 + c]);
#pragma unroll
        for (int e = 0; e < 4; ++e) fused[e] = fmaf(y0[e], s0[e], h0[e]) + fmaf(y1[e], s1[e], h1[e]);
      } else {
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(&Y0[d * ld0 + c]);
        const f32x4 i0 = *reinterpret_cast<const f32x4*>(&Y0[d * ld0 + C + c]);
        const f32x4 g1 = *reinterpret_cast<const f32x4*>(&Y1[d * ld1 + c]);
        const f32x4 i1 = *reinterpret_cast<const f32x4*>(&Y1[d * ld1 + C + c]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a0 = mm_sigmoid(g0[e]), a1 = mm_sigmoid(g1[e]);
          const float num = a0 * fmaf(i0[e], s0[e], h0[e]) + a1 * fmaf(i1[e], s1[e], h1[e]);
          fused[e] = num / (a0 + a1);
        }
      }
    }
    *reinterpret_cast<f32x4*>(&F[(0L * Lt + d) * C + c]) = fi;
    *reinterpret_cast<f32x4*>(&F[(1L * Lt + d) * C + c]) = fl;
    *reinterpret_cast<f32x4*>(&F[(2L * Lt + d) * C + c]) = fused;
  }
}

extern "C" int mmmot_fusion_combine(int mode, const float* cat, const float* Y0, int ld0, const float* Y1,
                                    int ld1, const float* sc0, const float* sh0, const float* sc1,
                                    const float* sh1, int ldsc, const int* tile_row0, const int* tile_nrows,
                                    const int* tile_group, int T, float* F, int Lt, int C, void* stream) {
  if (!cat || !Y0 || !sc0 || !sh0 || !tile_row0 || !tile_nrows || !F || T <= 0 || Lt <= 0) return MMMOT_EINVAL;
  if (mode < MMMOT_FUSION_A || mode > MMMOT_FUSION_C) return MMMOT_EINVAL;
  if (mode != MMMOT_FUSION_A && (!Y1 || !sc1 || !sh1)) return MMMOT_EINVAL;
  if (C % 4 != 0 || ld0 % 4 != 0 || ldsc % 4 != 0 || (Y1 && ld1 % 4 != 0)) return MMMOT_EINVAL;
  hipLaunchKernelGGL(fusion_combine_kernel, dim3(T, 16), dim3(256), 0, (hipStream_t)stream, mode, cat, Y0, ld0, Y1,
                     ld1, sc0, sh0, sc1, sh1, ldsc, tile_row0, tile_nrows, tile_group, F, Lt, C);
  return mm_check(hipGetLastError());
}

// ---------------------------------------------------------------------------
// Softmax modes over each group's N x M logit block (one workgroup per group):
// row statistics by wave reductions, column statistics by one thread per column
// (coalesced across the wave), then a single normalising pass.
__global__ __launch_bounds__(256) void softmax_pairs_kernel(
    const float* __restrict__ logits, float* __restrict__ out, const int* __restrict__ grp_row0,
    const int* __restrict__ grp_N, const int* __restrict__ grp_M, int mode) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int g = blockIdx.x;
  const int N = grp_N[g], M = grp_M[g];
  const float* x = logits + grp_row0[g];
  float* o = out + grp_row0[g];
  float* rmax = sm;
  float* rsum = sm + N;
  float* cmax = sm + 2 * N;
  float* csum = sm + 2 * N + M;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool dual = (mode != MMMOT_SM_SINGLE);
  for (int i = wave; i < N; i += 4) {
    float mx = -INFINITY;
    for (int j = lane; j < M; j += 64) mx = fmaxf(mx, x[(long)i * M + j]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int j = lane; j < M; j += 64) s += expf(x[(long)i * M + j] - mx);
    s = wave_sum(s);
    if (lane == 0) { rmax[i] = mx; rsum[i] = s; }
  }
  if (dual) {
    for (int j = tid; j < M; j += 256) {
      float mx = -INFINITY;
      for (int i = 0; i < N; ++i) mx = fmaxf(mx, x[(long)i * M + j]);
      float s = 0.f;
      for (int i = 0; i < N; ++i) s += expf(x[(long)i * M + j] - mx);
      cmax[j] = mx;
      csum[j] = s;
    }
  }
  __syncthreads();
  const int total = N * M;
  for (int idx = tid; idx < total; idx += 256) {
    const int i = idx / M, j = idx - i * M;
    const float v = x[idx];
    const float p = expf(v - rmax[i]) / rsum[i];
    float r = p;
    if (dual) {
      const float q = expf(v - cmax[j]) / csum[j];
      if (mode == MMMOT_SM_DUAL) r = p * q;
      else if (mode == MMMOT_SM_DUAL_ADD) r = (p + q) / 2.f;
      else r = fmaxf(p, q);
    }
    o[idx] = r;
  }
}

extern "C" int mmmot_softmax_pairs(const float* logits, float* out, const int* grp_row0, const int* grp_N,
                                   const int* grp_M, int G, int max_nm, int mode, void* stream) {
  if (!logits || !out || !grp_row0 || !grp_N || !grp_M || G <= 0 || max_nm <= 0) return MMMOT_EINVAL;
  if (mode < MMMOT_SM_SINGLE || mode > MMMOT_SM_DUAL_MAX) return MMMOT_EINVAL;
  const size_t lds = (size_t)2 * max_nm * sizeof(float);
  if (lds > 64 * 1024) return MMMOT_EINVAL;
  hipLaunchKernelGGL(softmax_pairs_kernel, dim3(G), dim3(256), lds, (hipStream_t)stream, logits, out, grp_row0,
                     grp_N, grp_M, mode);
  return mm_check(hipGetLastError());
}

// ---------------------------------------------------------------------------
extern "C" int mmmot_abi_version(void) { return 10; }

extern "C" int mmmot_device_info(int device, int* cu_count, char* arch, int arch_len) {
  hipDeviceProp_t p;
  hipError_t e = hipGetDeviceProperties(&p, device);
  if (e != hipSuccess) return (int)e;
  if (cu_count) *cu_count = p.multiProcessorCount;
  if (arch && arch_len > 0) {
    int i = 0;
    for (; i < arch_len - 1 && p.gcnArchName[i]; ++i) arch[i] = p.gcnArchName[i];
    arch[i] = 0;
  }
  return MMMOT_OK;
}
