// Point-cloud gather: which LiDAR points lie inside which convex polygon (3D box or image frustum), and the
// per-polygon ordered compaction that produces det_info['points'] / det_info['points_split']
// (see include/mmmot_hip.h: mmmot_points_count / mmmot_points_scatter).
//
// Replaces the numba loop _points_in_convex_polygon_3d_jit (reference point_cloud/geometry.py:96-114) and
// the per-box boolean-index loop of read_and_prep_points (point_cloud/preprocess.py:70-96).  HBM-bound byte
// shuffling: every point is read once by the count pass (the batched form keeps its membership masks for the scatter
// pass, which then reads only the kept points; the single-sweep form tests twice), every kept point written once; the
// membership arithmetic is the reference's - float64, left to right, no FMA contraction - so that the
// inside/outside decision is bit-identical.
#include "common.h"

#define PG_THREADS 256
#define PG_MAX_POLY 256  // polygons per call (24 doubles each in LDS)

__device__ __forceinline__ bool pg_inside(double x, double y, double z, const double* __restrict__ pl) {
  bool in = true;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    // ((x*nx + y*ny) + z*nz) + d, each operation rounded on its own (numba without fastmath)
    double s = __dadd_rn(__dmul_rn(x, pl[4 * k + 0]), __dmul_rn(y, pl[4 * k + 1]));
    s = __dadd_rn(s, __dmul_rn(z, pl[4 * k + 2]));
    s = __dadd_rn(s, pl[4 * k + 3]);
    in = in && !(s >= 0.0);
  }
  return in;
}

__device__ __forceinline__ void pg_load_planes(double* lpl, const double* __restrict__ planes, int NB) {
  for (int i = threadIdx.x; i < NB * 24; i += PG_THREADS) lpl[i] = planes[i];
}

__global__ __launch_bounds__(PG_THREADS) void pg_count_kernel(const float* __restrict__ pts, int P, int F,
                                                              const double* __restrict__ planes, int NB,
                                                              int* __restrict__ cnt, int nblk) {
  __shared__ double lpl[PG_MAX_POLY * 24];
  __shared__ int lcnt[PG_MAX_POLY];
  pg_load_planes(lpl, planes, NB);
  for (int j = threadIdx.x; j < NB; j += PG_THREADS) lcnt[j] = 0;
  __syncthreads();
  const int i = blockIdx.x * PG_THREADS + threadIdx.x;
  const bool valid = i < P;
  double x = 0, y = 0, z = 0;
  if (valid) {
    x = (double)pts[(long)i * F + 0];
    y = (double)pts[(long)i * F + 1];
    z = (double)pts[(long)i * F + 2];
  }
  const int lane = threadIdx.x & 63;
  for (int j = 0; j < NB; ++j) {
    const bool in = valid && pg_inside(x, y, z, &lpl[j * 24]);
    const unsigned long long m = __ballot(in);
    if (lane == 0 && m) atomicAdd(&lcnt[j], __popcll(m));
  }
  __syncthreads();
  for (int j = threadIdx.x; j < NB; j += PG_THREADS) cnt[(long)j * nblk + blockIdx.x] = lcnt[j];
}

// One workgroup: per polygon, the per-block counts become exclusive offsets; split = cumulative rows per
// polygon (an empty polygon occupies one row when pad_empty).
__global__ __launch_bounds__(PG_THREADS) void pg_scan_kernel(int* __restrict__ cnt, int NB, int nblk, int pad_empty,
                                                             int* __restrict__ split) {
  __shared__ int rows[PG_MAX_POLY];
  for (int j = threadIdx.x; j < NB; j += PG_THREADS) {
    int run = 0;
    int* c = cnt + (long)j * nblk;
    for (int b = 0; b < nblk; ++b) {
      const int v = c[b];
      c[b] = run;
      run += v;
    }
    rows[j] = (run == 0 && pad_empty) ? 1 : run;
    cnt[(long)NB * nblk + j] = run;  // polygon totals follow the [NB][nblk] offsets
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int j = 0; j < NB; ++j) {
      split[j] = run;
      run += rows[j];
    }
    split[NB] = run;
  }
}

__global__ __launch_bounds__(PG_THREADS) void pg_scatter_kernel(const float* __restrict__ pts, int P, int F,
                                                                const double* __restrict__ planes, int NB,
                                                                const int* __restrict__ off, int nblk,
                                                                const int* __restrict__ split, float* __restrict__ out,
                                                                int Fo) {
  __shared__ double lpl[PG_MAX_POLY * 24];
  __shared__ int wcnt[PG_MAX_POLY][PG_THREADS / 64];
  pg_load_planes(lpl, planes, NB);
  __syncthreads();
  const int i = blockIdx.x * PG_THREADS + threadIdx.x;
  const bool valid = i < P;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (valid) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (c < F) v[c] = pts[(long)i * F + c];
  }
  const double x = (double)v[0], y = (double)v[1], z = (double)v[2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int j = 0; j < NB; ++j) {
    const bool in = valid && pg_inside(x, y, z, &lpl[j * 24]);
    const unsigned long long m = __ballot(in);
    if (lane == 0) wcnt[j][wave] = __popcll(m);
  }
  __syncthreads();
  for (int j = 0; j < NB; ++j) {
    const bool in = valid && pg_inside(x, y, z, &lpl[j * 24]);
    const unsigned long long m = __ballot(in);
    if (in) {
      int pos = __popcll(m & ((1ull << lane) - 1ull));
      for (int w = 0; w < wave; ++w) pos += wcnt[j][w];
      float* o = out + ((long)split[j] + off[(long)j * nblk + blockIdx.x] + pos) * Fo;
      o[0] = v[0];
      o[1] = v[1];
      o[2] = v[2];
      if (Fo == 4) o[3] = v[3];
    }
  }
}

// An empty polygon owns ONE all-zero row (preprocess.py:80-81): total[j] == 0 while split says one row.
__global__ void pg_pad_kernel(const int* __restrict__ total, const int* __restrict__ split, int NB, float* __restrict__ out,
                              int Fo) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < NB && total[j] == 0 && split[j + 1] - split[j] == 1)
    for (int c = 0; c < Fo; ++c) out[(long)split[j] * Fo + c] = 0.f;
}

extern "C" int mmmot_points_count(const float* pts, int P, int F, const double* planes, int NB, int pad_empty,
                                  int* cnt, int* split, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!pts || !planes || !cnt || !split || P <= 0 || NB <= 0 || NB > PG_MAX_POLY) return MMMOT_EINVAL;
  if (F != 3 && F != 4) return MMMOT_EINVAL;
  const int nblk = (P + PG_THREADS - 1) / PG_THREADS;
  // cnt: [NB][nblk] per-block counts -> exclusive block offsets, followed by [NB] polygon totals
  hipLaunchKernelGGL(pg_count_kernel, dim3(nblk), dim3(PG_THREADS), 0, s, pts, P, F, planes, NB, cnt, nblk);
  hipLaunchKernelGGL(pg_scan_kernel, dim3(1), dim3(PG_THREADS), 0, s, cnt, NB, nblk, pad_empty, split);
  return mm_check(hipGetLastError());
}

extern "C" int mmmot_points_scatter(const float* pts, int P, int F, const double* planes, int NB, const int* cnt,
                                    const int* split, float* out, int Fo, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!pts || !planes || !cnt || !split || !out || P <= 0 || NB <= 0 || NB > PG_MAX_POLY) return MMMOT_EINVAL;
  if ((F != 3 && F != 4) || (Fo != 3 && Fo != F)) return MMMOT_EINVAL;
  const int nblk = (P + PG_THREADS - 1) / PG_THREADS;
  hipLaunchKernelGGL(pg_scatter_kernel, dim3(nblk), dim3(PG_THREADS), 0, s, pts, P, F, planes, NB, cnt, nblk, split,
                     out, Fo);
  hipLaunchKernelGGL(pg_pad_kernel, dim3((NB + 63) / 64), dim3(64), 0, s, cnt + (long)NB * nblk, split, NB, out, Fo);
  return mm_check(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------
// Batched variant: many sweeps per launch, and the image-frustum filter fused into the per-box test (a point
// is emitted for polygon j iff it is inside j AND inside its sweep's filter polygon - the same rows in the same
// order as filtering first and gathering afterwards, without the intermediate array and its read-back).
//   blk_sweep [NBLK] : sweep of every 256-point block       blk_first [NS] : first block of every sweep
//   sweep_row0 [NS+1]: first point of every sweep            poly0 [NS+1]   : first polygon of every sweep
//   filt [NS]        : index (into planes) of the sweep's filter polygon, or -1
//   cnt_off [NPOLY]  : first counter of every polygon (its sweep's blocks), totals follow at cnt_total
//   cnt              : cnt_total counters, NPOLY totals, then (8-byte aligned) four 64-bit membership masks per
//                      counter: ((cnt_total + NPOLY + 1) & ~1) + 8 * cnt_total ints in all
// A sweep may have at most PG_MAX_POLY polygons.
struct PgBatch {
  const float* pts;
  const double* planes;
  const int* blk_sweep;
  const int* blk_first;
  const int* sweep_row0;
  const int* poly0;
  const int* filt;
  const int* cnt_off;
  int F, NS, NPOLY, cnt_total;
};

#define PGB_CHUNK 16  // polygons per pass of a block
#define PGB_K1 2      // planes of the dense first stage (one slab of a box)

__device__ __forceinline__ bool pg_below(double x, double y, double z, const double* __restrict__ pl) {
  double s = __dadd_rn(__dmul_rn(x, pl[0]), __dmul_rn(y, pl[1]));
  s = __dadd_rn(s, __dmul_rn(z, pl[2]));
  s = __dadd_rn(s, pl[3]);
  return !(s >= 0.0);
}

// Membership of one 256-point block in its sweep's polygons, in two stages: every (point, polygon) pair is tested
// against the first PGB_K1 planes; the few pairs that pass (a slab of a box holds a few percent of a sweep) are
// queued in LDS and finish the remaining planes on dense waves.  The decision is the AND of the same six
// comparisons whatever the order, so the masks are the ones the straight loop gives.  Output per (polygon, block):
// the four 64-bit wave masks (read again by the scatter pass, which therefore repeats no arithmetic) and their
// population count.
__global__ __launch_bounds__(PG_THREADS) void pgb_count_kernel(PgBatch a, int* __restrict__ cnt,
                                                               unsigned long long* __restrict__ msk) {
  __shared__ double lpl[(PGB_CHUNK + 1) * 24];
  __shared__ double px[PG_THREADS], py[PG_THREADS], pz[PG_THREADS];
  __shared__ unsigned long long wmask[PGB_CHUNK][PG_THREADS / 64];
  __shared__ unsigned short queue[PGB_CHUNK * PG_THREADS];
  __shared__ int qn;
  const int tid = threadIdx.x, lane = tid & 63;
  const int s = a.blk_sweep[blockIdx.x];
  const int b = blockIdx.x - a.blk_first[s];
  const int p0 = a.poly0[s], np = a.poly0[s + 1] - p0, fi = a.filt[s];
  if (fi >= 0 && tid < 24) lpl[PGB_CHUNK * 24 + tid] = a.planes[(long)fi * 24 + tid];
  const int i = a.sweep_row0[s] + b * PG_THREADS + tid;
  bool valid = i < a.sweep_row0[s + 1];
  double x = 0, y = 0, z = 0;
  if (valid) {
    x = (double)a.pts[(long)i * a.F + 0];
    y = (double)a.pts[(long)i * a.F + 1];
    z = (double)a.pts[(long)i * a.F + 2];
  }
  px[tid] = x;
  py[tid] = y;
  pz[tid] = z;
  __syncthreads();
  if (valid && fi >= 0) valid = pg_inside(x, y, z, &lpl[PGB_CHUNK * 24]);
  for (int j0 = 0; j0 < np; j0 += PGB_CHUNK) {
    const int pc = min(PGB_CHUNK, np - j0);
    __syncthreads();  // the previous pass has finished with lpl / wmask / queue
    for (int t = tid; t < pc * 24; t += PG_THREADS) lpl[t] = a.planes[(long)(p0 + j0) * 24 + t];
    if (tid < PGB_CHUNK * (PG_THREADS / 64)) (&wmask[0][0])[tid] = 0ull;
    if (tid == 0) qn = 0;
    __syncthreads();
    for (int jj = 0; jj < pc; ++jj) {
      bool in = valid;
#pragma unroll
      for (int k = 0; k < PGB_K1; ++k) in = in && pg_below(x, y, z, &lpl[jj * 24 + 4 * k]);
      const unsigned long long m = __ballot(in);
      if (m) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&qn, __popcll(m));
        base = __shfl(base, 0, 64);
        if (in) queue[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)(tid | (jj << 8));
      }
    }
    __syncthreads();
    const int n = qn;
    for (int q = tid; q < n; q += PG_THREADS) {
      const int e = queue[q], pt = e & 255, jj = e >> 8;
      const double qx = px[pt], qy = py[pt], qz = pz[pt];
      bool in = true;
#pragma unroll
      for (int k = PGB_K1; k < 6; ++k) in = in && pg_below(qx, qy, qz, &lpl[jj * 24 + 4 * k]);
      if (in) atomicOr(&wmask[jj][pt >> 6], 1ull << (pt & 63));
    }
    __syncthreads();
    if (tid < pc * (PG_THREADS / 64)) {
      const int jj = tid >> 2, w = tid & 3;
      msk[((long)a.cnt_off[p0 + j0 + jj] + b) * 4 + w] = wmask[jj][w];
    }
    if (tid < pc)
      cnt[a.cnt_off[p0 + j0 + tid] + b] =
          __popcll(wmask[tid][0]) + __popcll(wmask[tid][1]) + __popcll(wmask[tid][2]) + __popcll(wmask[tid][3]);
  }
}

__device__ __forceinline__ int pg_wave_inclusive(int v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int u = __shfl_up(v, d, 64);
    if (lane >= d) v += u;
  }
  return v;
}

// one wave per polygon: its block counts -> exclusive offsets, and the polygon's total
__global__ __launch_bounds__(64) void pgb_scan_kernel(PgBatch a, int* __restrict__ cnt) {
  const int j = blockIdx.x, lane = threadIdx.x;
  int s = 0;  // sweep of polygon j (few sweeps: linear search, wave-uniform)
  while (a.poly0[s + 1] <= j) ++s;
  const int nblk = (a.sweep_row0[s + 1] - a.sweep_row0[s] + PG_THREADS - 1) / PG_THREADS;
  int* c = cnt + a.cnt_off[j];
  int run = 0;
  for (int b0 = 0; b0 < nblk; b0 += 64) {
    const int b = b0 + lane;
    const int v = b < nblk ? c[b] : 0;
    const int inc = pg_wave_inclusive(v, lane);
    if (b < nblk) c[b] = run + inc - v;
    run += __shfl(inc, 63, 64);
  }
  if (lane == 0) cnt[a.cnt_total + j] = run;
}

// one wave: the global split over all polygons (an empty polygon takes one padded row when pad_empty)
__global__ __launch_bounds__(64) void pgb_split_kernel(const int* __restrict__ tot, int NPOLY, int pad_empty,
                                                       int* __restrict__ split) {
  const int lane = threadIdx.x;
  int run = 0;
  for (int j0 = 0; j0 < NPOLY; j0 += 64) {
    const int j = j0 + lane;
    int v = 0;
    if (j < NPOLY) {
      v = tot[j];
      if (v == 0 && pad_empty) v = 1;
    }
    const int inc = pg_wave_inclusive(v, lane);
    if (j < NPOLY) split[j] = run + inc - v;
    run += __shfl(inc, 63, 64);
  }
  if (lane == 0) split[NPOLY] = run;
}

// rows of a block to their places: the masks of the count pass say who goes where; a point is read only if some
// polygon holds it
__global__ __launch_bounds__(PG_THREADS) void pgb_scatter_kernel(PgBatch a, const int* __restrict__ cnt,
                                                                 const unsigned long long* __restrict__ msk,
                                                                 const int* __restrict__ split, float* __restrict__ out,
                                                                 int Fo) {
  __shared__ unsigned long long wm[64][PG_THREADS / 64];
  __shared__ int base[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int s = a.blk_sweep[blockIdx.x];
  const int b = blockIdx.x - a.blk_first[s];
  const int p0 = a.poly0[s], np = a.poly0[s + 1] - p0;
  const long i = a.sweep_row0[s] + b * PG_THREADS + tid;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  bool loaded = false;
  for (int j0 = 0; j0 < np; j0 += 64) {
    const int pc = min(64, np - j0);
    __syncthreads();
    if (tid < pc * 4) wm[tid >> 2][tid & 3] = msk[((long)a.cnt_off[p0 + j0 + (tid >> 2)] + b) * 4 + (tid & 3)];
    if (tid < pc) base[tid] = split[p0 + j0 + tid] + cnt[a.cnt_off[p0 + j0 + tid] + b];
    __syncthreads();
    for (int jj = 0; jj < pc; ++jj) {
      const unsigned long long m = wm[jj][wave];
      if ((m >> lane) & 1ull) {
        if (!loaded) {
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (c < a.F) v[c] = a.pts[i * a.F + c];
          loaded = true;
        }
        int pos = __popcll(m & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; ++w) pos += __popcll(wm[jj][w]);
        float* o = out + ((long)base[jj] + pos) * Fo;
        o[0] = v[0];
        o[1] = v[1];
        o[2] = v[2];
        if (Fo == 4) o[3] = v[3];
      }
    }
  }
}

__device__ __host__ inline long pgb_mask_offset(int cnt_total, int NPOLY) {
  return ((long)cnt_total + NPOLY + 1) & ~1L;  // the 64-bit masks follow the counters and totals, 8-byte aligned
}

extern "C" int mmmot_points_count_batched(const float* pts, int F, int NS, int NPOLY, int NBLK, int cnt_total,
                                          const double* planes, const int* blk_sweep, const int* blk_first,
                                          const int* sweep_row0, const int* poly0, const int* filt,
                                          const int* cnt_off, int pad_empty, int* cnt, int* split, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!pts || !planes || !blk_sweep || !blk_first || !sweep_row0 || !poly0 || !filt || !cnt_off || !cnt || !split)
    return MMMOT_EINVAL;
  if ((F != 3 && F != 4) || NS <= 0 || NPOLY <= 0 || NBLK <= 0) return MMMOT_EINVAL;
  PgBatch a{pts, planes, blk_sweep, blk_first, sweep_row0, poly0, filt, cnt_off, F, NS, NPOLY, cnt_total};
  hipLaunchKernelGGL(pgb_count_kernel, dim3(NBLK), dim3(PG_THREADS), 0, s, a, cnt,
                     (unsigned long long*)(cnt + pgb_mask_offset(cnt_total, NPOLY)));
  hipLaunchKernelGGL(pgb_scan_kernel, dim3(NPOLY), dim3(64), 0, s, a, cnt);
  hipLaunchKernelGGL(pgb_split_kernel, dim3(1), dim3(64), 0, s, cnt + cnt_total, NPOLY, pad_empty, split);
  return mm_check(hipGetLastError());
}

extern "C" int mmmot_points_scatter_batched(const float* pts, int F, int NS, int NPOLY, int NBLK, int cnt_total,
                                            const double* planes, const int* blk_sweep, const int* blk_first,
                                            const int* sweep_row0, const int* poly0, const int* filt,
                                            const int* cnt_off, const int* cnt, const int* split, float* out, int Fo,
                                            void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!pts || !planes || !blk_sweep || !blk_first || !sweep_row0 || !poly0 || !filt || !cnt_off || !cnt || !split || !out)
    return MMMOT_EINVAL;
  if ((F != 3 && F != 4) || (Fo != 3 && Fo != F) || NS <= 0 || NPOLY <= 0 || NBLK <= 0) return MMMOT_EINVAL;
  PgBatch a{pts, planes, blk_sweep, blk_first, sweep_row0, poly0, filt, cnt_off, F, NS, NPOLY, cnt_total};
  hipLaunchKernelGGL(pgb_scatter_kernel, dim3(NBLK), dim3(PG_THREADS), 0, s, a, cnt,
                     (const unsigned long long*)(cnt + pgb_mask_offset(cnt_total, NPOLY)), split, out, Fo);
  hipLaunchKernelGGL(pg_pad_kernel, dim3((NPOLY + 63) / 64), dim3(64), 0, s, cnt + cnt_total, split, NPOLY, out, Fo);
  return mm_check(hipGetLastError());
}
