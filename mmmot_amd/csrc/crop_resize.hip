// Per-detection image preparation on the device: crop (zero padding outside the frame) -> antialiased
// bilinear resize to S x S -> to_tensor -> normalize, i.e. the model input ``dets`` [N][3][S][S]
// (see include/mmmot_hip.h: mmmot_crop_resize_norm).
//
// Replaces, per detection, PIL ``img.crop(box).resize((S, S), Image.BILINEAR)`` + torchvision ToTensor /
// Normalize (reference dataset/test_seq_dataset.py:212-218, utils/build_util.py:111-112,137-142).  The
// arithmetic is Pillow's (src/libImaging/Resample.c): triangle filter with the support stretched by the scale
// when shrinking, coefficients in double precision (every operation rounded on its own, as the C code
// compiled for baseline x86-64), normalised, 22-bit fixed point, two separable passes with an 8-bit
// intermediate - so the uint8 result and the float32 tensor are bit-identical to the reference pipeline.
// Byte work, HBM-bound on the float32 output (3*S*S*4 B per detection); crops are read through L2.
#include "common.h"

#define CR_PREC 22

__device__ __forceinline__ double cr_tri(double x) {
  if (x < 0.0) x = -x;
  return x < 1.0 ? __dsub_rn(1.0, x) : 0.0;
}

// grid (N, 2): axis 0 = horizontal (crop width), 1 = vertical (crop height); thread = output index.
// work layout per (n, axis): bounds [S][2] ints, then kk [S][kmax] ints.
__global__ void cr_coeff_kernel(const int* __restrict__ boxes, int S, int kmax, int* __restrict__ work) {
  const int n = blockIdx.x, axis = blockIdx.y;
  const int xx = blockIdx.z * blockDim.x + threadIdx.x;
  if (xx >= S) return;
  const int* b = boxes + n * 4;
  const int in_size = axis == 0 ? b[2] - b[0] : b[3] - b[1];
  int* base = work + ((long)n * 2 + axis) * S * (2 + kmax);
  int* bounds = base + xx * 2;
  int* kk = base + S * 2 + xx * kmax;
  if (in_size <= 0) {
    bounds[0] = 0;
    bounds[1] = 0;
    return;
  }
  const double scale = __ddiv_rn((double)(float)in_size, (double)S);
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = filterscale;  // 1.0 * filterscale
  const double ss = __ddiv_rn(1.0, filterscale);
  const double center = __dmul_rn(__dadd_rn((double)xx, 0.5), scale);  // in0 = 0
  int xmin = (int)__dadd_rn(__dsub_rn(center, support), 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)__dadd_rn(__dadd_rn(center, support), 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  double ww = 0.0;
  for (int x = 0; x < xmax; ++x) {
    const double w = cr_tri(__dmul_rn(__dadd_rn(__dsub_rn((double)(x + xmin), center), 0.5), ss));
    ww = __dadd_rn(ww, w);
  }
  for (int x = 0; x < xmax; ++x) {
    double w = cr_tri(__dmul_rn(__dadd_rn(__dsub_rn((double)(x + xmin), center), 0.5), ss));
    if (ww != 0.0) w = __ddiv_rn(w, ww);
    kk[x] = (int)__dadd_rn(0.5, __dmul_rn(w, (double)(1 << CR_PREC)));  // triangle weights are >= 0
  }
  for (int x = xmax; x < kmax; ++x) kk[x] = 0;
  bounds[0] = xmin;
  bounds[1] = xmax;
}

__device__ __forceinline__ int cr_clip8(int acc) {
  const int v = acc >> CR_PREC;  // arithmetic shift, like Pillow's lookup on in >> PRECISION_BITS
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// grid (N, S): one output row per workgroup, one output column per thread.
__global__ __launch_bounds__(256) void cr_resize_kernel(const unsigned char* __restrict__ img, int H, int W,
                                                        const int* __restrict__ boxes, int S, int kmax,
                                                        const int* __restrict__ work, const float* __restrict__ ms,
                                                        float* __restrict__ out, unsigned char* __restrict__ out_u8) {
  const int n = blockIdx.x, oy = blockIdx.y, ox = threadIdx.x;
  if (ox >= S) return;
  const int* b = boxes + n * 4;
  const int x1 = b[0], y1 = b[1];
  const int* hb = work + ((long)n * 2 + 0) * S * (2 + kmax);
  const int* vb = work + ((long)n * 2 + 1) * S * (2 + kmax);
  const int xmin = hb[ox * 2], xcnt = hb[ox * 2 + 1];
  const int ymin = vb[oy * 2], ycnt = vb[oy * 2 + 1];
  const int* kh = hb + S * 2 + ox * kmax;
  const int* kv = vb + S * 2 + oy * kmax;
  int acc[3] = {1 << (CR_PREC - 1), 1 << (CR_PREC - 1), 1 << (CR_PREC - 1)};
  for (int j = 0; j < ycnt; ++j) {
    const int gy = y1 + ymin + j;
    int h[3] = {1 << (CR_PREC - 1), 1 << (CR_PREC - 1), 1 << (CR_PREC - 1)};
    if (gy >= 0 && gy < H) {
      const unsigned char* row = img + (long)gy * W * 3;
      for (int i = 0; i < xcnt; ++i) {
        const int gx = x1 + xmin + i;
        if (gx >= 0 && gx < W) {
          const int k = kh[i];
          h[0] += row[gx * 3 + 0] * k;
          h[1] += row[gx * 3 + 1] * k;
          h[2] += row[gx * 3 + 2] * k;
        }
      }
    }
    const int k = kv[j];
    acc[0] += cr_clip8(h[0]) * k;  // the 8-bit intermediate image of the two-pass resize
    acc[1] += cr_clip8(h[1]) * k;
    acc[2] += cr_clip8(h[2]) * k;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int v8 = cr_clip8(acc[c]);
    if (out_u8) out_u8[(((long)n * S + oy) * S + ox) * 3 + c] = (unsigned char)v8;
    if (out) {
      const float x = __fdiv_rn((float)v8, 255.f);                       // to_tensor
      out[(((long)n * 3 + c) * S + oy) * S + ox] = __fdiv_rn(__fsub_rn(x, ms[c]), ms[3 + c]);  // normalize
    }
  }
}

extern "C" int mmmot_crop_resize_norm(const unsigned char* img, int H, int W, const int* boxes, int N, int S, int kmax,
                                      const float* mean_std, int* work, float* out, unsigned char* out_u8,
                                      void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!img || !boxes || !mean_std || !work || (!out && !out_u8) || H <= 0 || W <= 0 || N <= 0) return MMMOT_EINVAL;
  if (S <= 0 || S > 256 || kmax < 3) return MMMOT_EINVAL;
  hipLaunchKernelGGL(cr_coeff_kernel, dim3(N, 2, (S + 63) / 64), dim3(64), 0, s, boxes, S, kmax, work);
  hipLaunchKernelGGL(cr_resize_kernel, dim3(N, S), dim3(256), 0, s, img, H, W, boxes, S, kmax, work, mean_std, out,
                     out_u8);
  return mm_check(hipGetLastError());
}

// ToTensor + Normalize of 8-bit crops [N][S][S][3] -> fp32 [N][3][S][S] (the same IEEE divisions as cr_resize_kernel):
// the model input for the paths that cannot take the bytes directly (exact-fp32 trunk, unfused first layer).
__global__ __launch_bounds__(256) void cr_u8_normalize_kernel(const unsigned char* __restrict__ u8, long npix, int S2,
                                                               const float* __restrict__ ms, float* __restrict__ out) {
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
    const long n = p / S2;
    const int r = (int)(p - n * S2);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float x = __fdiv_rn((float)u8[p * 3 + c], 255.f);
      out[(n * 3 + c) * S2 + r] = __fdiv_rn(__fsub_rn(x, ms[c]), ms[3 + c]);
    }
  }
}

extern "C" int mmmot_u8_normalize(const unsigned char* u8, int N, int S, const float* mean_std, float* out, void* stream) {
  if (!u8 || !mean_std || !out || N <= 0 || S <= 0) return MMMOT_EINVAL;
  const long npix = (long)N * S * S;
  const int grid = (int)((npix + 255) / 256 < 8192 ? (npix + 255) / 256 : 8192);
  hipLaunchKernelGGL(cr_u8_normalize_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, u8, npix, S * S, mean_std, out);
  return mm_check(hipGetLastError());
}
