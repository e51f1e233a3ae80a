/*
 * mmmot_hip.h - C ABI of libmmmot_hip.so, the MI355X (gfx950) device library
 * behind the mmMOT per-frame-pair network forward.
 *
 * The reference (ZwwWayne/mmMOT) is pure Python: its "FFI" for this path is
 * the set of ATen operators that TrackingNet.forward dispatches
 * (modules/tracking_net.py:165-193; operator inventory SURVEY.md section 2b,
 * K1-K18).  Each entry point below replaces a *fused group* of those
 * operator calls; the reference call sites are cited per function.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to contiguous fp32 / int32 data unless
 *    named h_* ; no torch types cross this boundary;
 *  - activations are ROW-major "position-major": one row per position
 *    (pixel / point / detection / detection-pair), channels contiguous.  This
 *    is the transpose of the reference's N,C,L layout and is an internal
 *    choice: MFMA A-fragments and 16-byte coalesced loads both want the
 *    reduction (channel) axis contiguous;
 *  - "row tiles": rows are processed in tiles of <=128 rows that never span
 *    two normalisation groups (a group = one frame-pair, or one
 *    (frame-pair, modality-row)); tile_row0/tile_nrows/tile_group describe
 *    the T tiles, grp_* arrays describe the G groups;
 *  - all launches are asynchronous on `stream` (a hipStream_t passed as
 *    void*); the functions never synchronise and never allocate;
 *  - return value: 0 on success, MMMOT_EINVAL (-1) for a contract violation
 *    (bad alignment / unsupported size), otherwise the positive hipError_t
 *    of the failed launch.
 */
#ifndef MMMOT_HIP_H
#define MMMOT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMMOT_OK 0
#define MMMOT_EINVAL (-1)

/* activation codes */
#define MMMOT_ACT_NONE 0
#define MMMOT_ACT_RELU 1
#define MMMOT_ACT_SIGMOID 2

/* A-operand modes of mmmot_gemm_rows */
#define MMMOT_A_PLAIN 0     /* A = X[r][k]                                  */
#define MMMOT_A_NORM_RELU 1 /* A = max(0, X[r][k]*sc[g][k] + sh[g][k])      */
#define MMMOT_A_PAIR 2      /* A = op(FA[a_off+i][k], FB[b_off+j][k])       */

/* pairwise operators (reference modules/gcn.py:6-41) */
#define MMMOT_PAIR_MULTIPLY 0  /* batch_multiply  gcn.py:6-14  */
#define MMMOT_PAIR_MINUS_ABS 1 /* batch_minus_abs gcn.py:17-28 */
#define MMMOT_PAIR_MINUS 2     /* batch_minus     gcn.py:31-41 */

/* softmax modes (reference modules/tracking_net.py:106-126) */
/* elementwise loss terms of mmmot_score_loss (reference cost.py:97-131) */
#define MMMOT_LOSS_BCE 0        /* F.binary_cross_entropy_with_logits */
#define MMMOT_LOSS_L2 1         /* F.mse_loss(score.mul(mask), gt) */
#define MMMOT_LOSS_SMOOTH_L1 2  /* F.smooth_l1_loss(score.mul(mask), gt) */
#define MMMOT_SM_SINGLE 1
#define MMMOT_SM_DUAL 2
#define MMMOT_SM_DUAL_ADD 3
#define MMMOT_SM_DUAL_MAX 4

/* fusion modes (reference modules/fusion_net.py) */
#define MMMOT_FUSION_A 0
#define MMMOT_FUSION_B 1
#define MMMOT_FUSION_C 2

/* library / device info -------------------------------------------------- */
/* ABI history: 1 = first round-1 cut; 2 = gemm colsum / gemm_ares / gram / points / crops entry points,
 * gn_finalize tile_nrows, segment_mean seg_div; 3 = hq8 arithmetic (mmmot_conv3x3_bn_relu_hq8, mmmot_conv1_fused_hq8,
 * mmmot_hq8_pack/unpack, mmmot_segment_mean hl16 = 2), patch-kernel test / timing knobs;
 * 4 = per-output-channel weight scales (oscale is a [Cout] vector in the hl16 / hq8 trunk entry points),
 * mmmot_trunk_range_read, the tile / LDS-DMA trunk kernels and their knobs removed, timing experiments only in
 * -DMMMOT_DEBUG builds; 5 = training backward of the pairwise block (mmmot_gn_bwd_*, mmmot_gemm_tn,
 * mmmot_pair_bwd, mmmot_pair_expand_bwd, mmmot_rowdot_bwd, mmmot_softmax_pairs_bwd, mmmot_fusion_c_bwd, mmmot_add_rows),
 * mmmot_pointnet_layer1 takes K = 3 | 4; mmmot_pn_mlp64 added (additive, still 5);
 * 6 = mmmot_trunk_range_bind (per-caller range-guard counter blocks); 7 - 9: see the entry points marked so below;
 * 10 = mmmot_gram_rows at K = 128 writes the 32 x 32 blocks on and above the block diagonal of Gout only (the finalize
 * entry point mirrors), mmmot_set_gemm_rows_variant: 2 rejected, 3 / 4 = the two forms of the wide kernel. */
int mmmot_abi_version(void);
/* returns 0 and fills cu_count / gcn arch string (<=32 bytes) of device 0..; */
int mmmot_device_info(int device, int* cu_count, char* arch, int arch_len);

/* ---------------------------------------------------------------------------
 * VGG16-BN trunk layer: 3x3 conv (pad 1) + folded eval-BatchNorm + ReLU, with
 * the following 2x2/s2 max-pool optionally fused into the epilogue.
 * Replaces conv2d + batch_norm + relu_ (+ max_pool2d) of
 * reference modules/vgg.py:67-80 as regrouped by
 * modules/appear_net.py:130-157,166-172.
 *   first=1: `in` is the reference's NCHW crop tensor [L][3][H][W]
 *            (modules/tracking_net.py:132) and wp is [Cout][32]
 *            (k = (ky*3+kx)*3 + c, zero padded 27->32);
 *   first=0: `in` is NHWC [L][H][W][Cin], wp is [9][Cout][Cin].
 *   out: NHWC [L][H][W][Cout], or [L][H>>1][W>>1][Cout] when pool=1 (odd maps are floored like
 *        nn.MaxPool2d(2, 2), modules/vgg.py:72: the last row / column has no window).
 * Implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32).  Any H, W >= 1;
 * Cin % 32 == 0 (first=0); Cout % 64 == 0.
 * ------------------------------------------------------------------------- */
int mmmot_conv3x3_bn_relu(const float* in, const float* wp, const float* bias,
                          float* out, int L, int H, int W, int Cin, int Cout,
                          int first, int pool, void* stream);

/* ---------------------------------------------------------------------------
 * fp16-matrix-core variant of the trunk layer with fp32-class accuracy ("f16x3").
 * "hl16" split-half format (same bytes as fp32): a row of C channels is C/8
 * units of 32 bytes, unit u = [fp16 hi of channels 8u..8u+7 | fp16 lo of the
 * same channels], hi = fp16(x), lo = fp16(x - hi).  Every product is evaluated
 * as a_hi*w_hi + a_hi*w_lo + a_lo*w_hi on v_mfma_f32_32x32x16_f16 with fp32
 * accumulation (3 MFMAs per algorithmic product: ceiling 2.5 PF / 3).
 *   in  : hl16 NHWC [L][H][W][Cin]       wp: hl16 [9][Cout][Cin], output channel n host-scaled by 2^wshift[n]
 *   out : hl16 NHWC (pooled when pool=1)  oscale[Cout] = 2^-wshift[n], applied before the bias
 * Cin % 32 == 0, Cout % 64 == 0, any H, W >= 1 (pool=1 floors odd maps: out is [L][H>>1][W>>1][Cout]).  Same reference lines as above
 * (conv2d + batch_norm + relu_ (+ max_pool2d), modules/vgg.py:67-80).
 * Kernel: LDS-resident haloed activation patch + streamed weight ring, persistent workgroups of 256 pixels
 * (16x16 or 4 x 8x8 blocks) x 64/128 channels (conv3x3_hl16_patch.hip).
 * mmmot_conv3x3_first_hl16 is the Cin=3 layer (fp32 MFMA on the NCHW crops) that
 * emits hl16; mmmot_hl16_pack/unpack convert n fp32 values (n % 8 == 0). */
int mmmot_conv3x3_bn_relu_hl16_patch(const void* in, const void* wp, const float* bias, void* out,
                                     int L, int H, int W, int Cin, int Cout, int pool, const float* oscale,
                                     void* stream);
/* conv1_1 (3->64) + conv1_2 (64->64) + 2x2 max-pool of the VGG trunk in ONE kernel (reference modules/vgg.py:67-80,
 * layers 0-6 of vgg16_bn.features): the patch kernel computes conv1_1 for the haloed 18x18 patch of every tile in
 * its prologue instead of reading it, so the [L][H][W][64] tensor (537 MB per cfg3 pair) is never written.
 *   crops NCHW fp32 [L][3][H][W];  w1 hl16 [64][32] (k = (ky*3+kx)*3 + colour, zero-padded), bias1 [64], oscale1 (scalar);
 *   w2 hl16 [9][64][64], bias2 [64], oscale2 [64];  out hl16 NHWC [L][H/2][W/2][64].  H, W even.
 *   bias1 travels through the matrix cores in the k = 27 slot of w1's records (split hi + lo like a weight):
 *   |bias1[n]| / oscale1 must stay below 65504 (host: pack.conv1_weight_shift picks the shift for weights AND bias). */
int mmmot_conv1_fused_hl16(const float* crops, const void* w1, const float* bias1, float oscale1,
                           const void* w2, const float* bias2, const float* oscale2, void* out, int L, int H, int W,
                           void* stream);
/* ABI 6: mmmot_conv1_fused_hl16 / _hq8 fed by the 8-bit crops of the resize instead of the fp32 model input (SURVEY 8f
 * rank 3 "fuse into the first conv's loader"): crops_u8 [L][H][W][3] RGB (mmmot_crop_resize_norm's out_u8), ToTensor and
 * Normalize - x / 255, (x - mean) / std with IEEE divisions, reference utils/build_util.py:111-112,137-142 - applied while
 * the raw window of a tile is fetched; bit-identical to feeding the fp32 tensor.  q8 != 0: the hq8 contract (w2, out). */
int mmmot_conv1_fused_u8(const unsigned char* crops_u8, float mean0, float mean1, float mean2, float std0, float std1,
                         float std2, const void* w1, const float* bias1, float oscale1, const void* w2,
                         const float* bias2, const float* oscale2, void* out, int L, int H, int W, int q8, void* stream);
/* "hq8" arithmetic (trunk mode 'f16q8', same reference lines): the two correction terms of the hi/lo split run on
 * the fp8 matrix cores (one block-scaled K=64 MFMA), 2 instead of 3 fp16-MFMA equivalents per product.
 *   activation record per 32 channels (128 bytes, like hl16): [32 x fp16 hi | 32 x e4m3(a / 4) | 32 x e4m3((a - hi) * 512)]
 *   weight record per 32 input channels: [32 x fp16 hi(w') | 32 x e4m3((w' - hi) * 32) | 32 x e4m3(hi / 64)],
 *   w' = w * 2^wshift[n] with a shift PER OUTPUT CHANNEL n (max|w'| of every channel in (2^13, 2^14]: the e4m3 copies
 *   of a low-gain channel of a trained, BatchNorm-folded layer keep their precision).
 * e4m3 = OCP FP8 E4M3 (max 448).  Relative error of a product ~2^-15 instead of 2^-21 (hl16); end-to-end score
 * error stays below the 1e-3 budget (tools/study_fp8_correction.py, tests/test_hq8_gpu.py, tests/test_robust_gpu.py).
 * Cin % 32 == 0, Cout % 64 == 0, any H, W >= 1 (pool=1 floors odd maps).  mmmot_hq8_pack/unpack convert n fp32 values (n % 32 == 0). */
int mmmot_conv3x3_bn_relu_hq8(const void* in, const void* wp, const float* bias, void* out,
                              int L, int H, int W, int Cin, int Cout, int pool, const float* oscale, void* stream);
/* mmmot_conv1_fused_hl16 with conv1_2 in hq8 arithmetic: w1 stays hl16, w2 is hq8 [9][64][64], out is hq8 */
int mmmot_conv1_fused_hq8(const float* crops, const void* w1, const float* bias1, float oscale1,
                          const void* w2, const float* bias2, const float* oscale2, void* out, int L, int H, int W,
                          void* stream);
int mmmot_hq8_pack(const float* x, void* y, long n, void* stream);
int mmmot_hq8_unpack(const void* x, float* y, long n, void* stream);
/* Range guard of the reduced-range activation formats.  The trunk epilogues count, on the device,
 *   out4[0]: activation elements written with |a| > 1792 in hq8 mode (their e4m3 copies saturate: the products of
 *            that element fall back to fp16-class accuracy);
 *   out4[1]: elements clamped at +-65000 by the fp16 `hi` half (hl16 and hq8: the value itself is wrong);
 *   out4[2]: workgroup-lanes of the fused first launch whose conv1_1 outputs crossed the mode's limit;  out4[3]: 0.
 * Synchronous read of the current device's counters into a HOST array (the caller synchronises the launch stream
 * first); reset != 0 clears them.  mmmot_amd.Engine reads them on the first forward and periodically, and moves the
 * trunk to f16x3 / f32 when they are hit. */
int mmmot_trunk_range_read(unsigned int* out4, int reset);
/* ABI 6.  Counter block of the CALLER (device memory, 4 x uint32, same layout as above) that the trunk launches issued
 * by this host thread report to from now on; NULL returns to the library's per-device block.  The block travels as a
 * kernel argument: a launch - or a launch captured into a hipGraph - keeps the block it was issued with, so every
 * engine (and every captured forward) has its own window, zeroed and read by its owner with ordinary stream-ordered
 * copies (mmmot_amd.Engine: one asynchronous 16-byte read-back per forward, inspected one step later). */
int mmmot_trunk_range_bind(unsigned int* counters4);
/* tests: cap the persistent grid of the patch kernels (multiple of 8, 0 = one workgroup per CU) so that small
 * problems run several chained tiles per workgroup like production sizes do.  Results do not depend on it. */
int mmmot_set_patch_grid_limit(int n);
/* ABI 7.  tests: smallest block edge the trunk dispatcher may choose - 0 / 4 = automatic (maps of at most 4 x 4 pixels,
 * conv5_x at 64-pixel crops /root/reference/modules/vgg.py:67-80, run 16 whole maps per 256-row tile without halo),
 * 8 = such maps run as one haloed 8 x 8 block at 25 % fill like before ABI 7.  Results do not depend on it, bit for bit. */
int mmmot_set_patch_min_block(int bs);
#ifdef MMMOT_DEBUG
/* -DMMMOT_DEBUG builds only (tools/, never the product library; csrc/patch_debug.h): phase timers of the patch kernel
 * (0 = the product kernels, 9 = the timed instantiations; results are correct either way) and their read-out. */
int mmmot_set_patch_variant(int v);
int mmmot_debug_read_patch_timers(unsigned long long* out8, int reset);
#endif
int mmmot_conv3x3_first_hl16(const float* in, const float* wp, const float* bias, void* out,
                             int L, int H, int W, int Cout, void* stream);
int mmmot_hl16_pack(const float* x, void* y, long n, void* stream);
int mmmot_hl16_unpack(const void* x, float* y, long n, void* stream);
/* ABI 7, training step: y = hl16(x * 2^(target - ex)) with max |x| = amax[0] = f * 2^ex, f in [0.5, 1) - a DEVICE scalar
 * (mmmot_absmax), NULL = unscaled; and the vector out[0..C) = 2^-(shift_a + shift_b) that undoes two such scales in the
 * consumer's epilogue.  No host round trip: gradients (1e-4 .. 1e-7) and freshly stepped weights are scaled where they are. */
int mmmot_hl16_pack_pow2(const float* x, void* y, long n, const float* amax, int target, void* stream);
int mmmot_pow2_oscale(float* out, int C, const float* amax_a, int target_a, const float* amax_b, int target_b, void* stream);
/* ABI 7, training step: the plain 3x3 convolution out[p][n] = oscale[n] * sum_taps in[p + off] . wp[tap][n] + bias[n] as
 * fp32 rows [L*H*W][Cout] (no BatchNorm, no ReLU: the training-mode forward of /root/reference/modules/vgg.py:74-76 keeps
 * the pre-BatchNorm tensor, and the input gradient is the same convolution of dZ with the flipped, transposed weights) on
 * the fp16 matrix cores (f16x3): in = hl16 rows [L*H*W][Cin], wp = hl16 [9][Cout][Cin].  Cin % 32 == 0, Cout % 64 == 0. */
int mmmot_conv3x3_raw_hl16(const void* in, const void* wp, const float* bias, float* out, int L, int H, int W, int Cin,
                           int Cout, const float* oscale, void* stream);

/* ---------------------------------------------------------------------------
 * Row GEMM with fused operand generation and statistics epilogue:
 *   v[r][n] = sum_k A(r,k) * W[n][k] + bias[n] + dbias[rowidx[r]][n]
 *   part[t][0][n] = S  = sum_{r in tile t} v[r][n]
 *   part[t][1][n] = M2 = sum_{r in tile t} (v[r][n] - S/nrows_t)^2   (tile-centred)
 *   Y[r][n] = act(v[r][n])
 * Replaces every conv1d(k=1) / conv2d(1x1) / linear of the path
 * (reference modules/point_net.py:28,40,125,134-138; fusion_net.py:14-29,
 * 53-60,80-83; gcn.py:59-66; new_end.py:48-58; tracking_net.py:92-100;
 * appear_net.py:19-25) together with the producer side of the following
 * GroupNorm (per-tile sums; mmmot_gn_finalize turns them into scale/shift)
 * and the consumer side of the preceding one (A_NORM_RELU prologue).
 * A_PAIR generates the N x M pairwise tensor of modules/gcn.py:6-41 on the
 * fly in LDS - it is never materialised in HBM.
 * K % 32 == 0, N % 64 == 0, all leading dimensions % 4 == 0, pointers 16-byte
 * aligned.  Y, part, bias, dbias may be NULL.
 * ------------------------------------------------------------------------- */
typedef struct mmmot_gemm_args {
  const float* X; int ldx;              /* PLAIN / NORM_RELU source rows      */
  const float* W;                       /* [N][K]                             */
  const float* bias;                    /* [N] or NULL                        */
  const float* dbias; const int* rowidx; int lddb; /* gathered bias or NULL   */
  float* Y; int ldy;                    /* output rows or NULL                */
  float* part;                          /* [T][2][N] or NULL                  */
  const float* sc; const float* sh; int ldsc;      /* [G][ldsc] (NORM_RELU)   */
  const float* FA; const float* FB; int ldf;       /* PAIR feature rows       */
  const int* tile_row0; const int* tile_nrows; const int* tile_group; /* [T]  */
  const int* grp_row0; const int* grp_M; const int* grp_aoff; const int* grp_boff; /* [G] PAIR */
  int T; int N; int K;
  int amode; int pairop; int act;
  /* w_hl16 = 1: W is in the hl16 split-half format ([N][K/8] units of [hi8|lo8] halves, values
   * pre-scaled by 1/oscale) and the contraction runs on the fp16 matrix cores with the 3-term
   * hi/lo split (fp32-class accuracy, see mmmot_conv3x3_bn_relu_hl16); requires K % 64 == 0.
   * The A operand stays fp32 in memory and is split while staged.  oscale multiplies the
   * accumulator before bias. */
  int w_hl16; float oscale;
  /* Fused consumer of the NEXT GroupNorm (v2): when colsum != NULL the epilogue also evaluates
   * colsum[t][n] = sum over the tile's rows of relu(v[r][n] * osc[g][n] + osh[g][n]) (g = tile_group[t]),
   * i.e. the per-tile part of "normalise + ReLU + per-detection mean" (reference
   * modules/point_net.py:28-39,138-148) without the [rows][N] tensor ever reaching HBM: run the GEMM once
   * with Y = NULL, part != NULL (statistics), finalize, and once with Y = NULL, colsum != NULL; tiles
   * must not straddle the segments to be averaged; mmmot_segment_mean over the tile rows (seg_div =
   * rows per segment) finishes the mean. */
  const float* osc; const float* osh; int ldosc;   /* [G][ldosc] or NULL                */
  float* colsum;                        /* [T][N] or NULL                     */
  /* ABI 7 (appended).  PAIR: nonzero = the caller guarantees BOTH grp_M[g] % 32 == 0 for every group AND
   * (tile_row0[t] - grp_row0[tile_group[t]]) % 32 == 0 for every tile (tiles cut every 128 rows from the start of their
   * group satisfy the second condition), i.e. the 32 rows of every 32-row block of a tile share their a_i row; the wide
   * kernel (csrc/gemm_wide.hip) then stages that row in LDS once per wave instead of requesting it per lane (it serves
   * PAIR launches only with this guarantee, takes a_i from the block's first row, and returns WRONG Y - silently - for a
   * caller that sets the flag on ragged or unaligned pair tiles).  0 is always safe: the tile kernel serves the launch. */
  int pair_uniform32;
} mmmot_gemm_args;
int mmmot_gemm_rows(const mmmot_gemm_args* a, void* stream);
/* ABI 7.  tests: which kernel serves mmmot_gemm_rows - 0 = automatic (the wide kernel of csrc/gemm_wide.hip for the
 * K >= 256 layers of the pairwise block, reference modules/gcn.py:59-66 / new_end.py:48-52, when w_hl16 = 1, amode is
 * PAIR or NORM_RELU, N % 256 == 0, no dbias / colsum / activation and the launch fills the chip), 1 = the tile kernel
 * only (the kernel before ABI 7), 3 / 4 = the wide kernel whenever the layer is eligible (N % 256 == 0, K = 256 or 512,
 * any size) with one 128-row tile per four-wave workgroup (two workgroups per CU) / two tiles per eight-wave workgroup;
 * 2 (the round-5 form of the wide kernel, removed in round 6) is rejected.  Y and part do not depend on it, bit for bit. */
int mmmot_set_gemm_rows_variant(int v);

/* A-resident row GEMM for layers whose output is only ever reduced (PointNet conv5 128->1024 and
 * PointNet_v1.conv1 64->512, reference modules/point_net.py:28-39,138-148): the workgroup keeps its 128
 * activation rows (normalised, ReLU'd, hi/lo split once) in LDS and walks all N/128 channel tiles;
 *   v[r][n] = oscale * sum_k relu(X[r][k]*sc[g][k]+sh[g][k]) * W[n][k] + bias[n] + dbias[tile_dbrow[t]][n]
 * is never stored.  Outputs, per 64-row HALF tile h of tile t (row index 2t+h; a half may be empty):
 *   part   [2T][2][N]: sum and half-tile-centred M2 of v   -> mmmot_gn_finalize with tile_nrows = rows per half
 *   colsum [2T][N]   : sum of relu(v*osc[g][n]+osh[g][n])  -> mmmot_segment_mean with seg_div
 * W is hl16 ([N][K/8] units, pre-scaled by 1/oscale); K is 64 or 128; N % 256 == 0 (K = 64 with N <= 512,
 * served by the weight-resident kernel: N % 128 == 0); tiles must not straddle
 * the rows that share a dbias row (detection-aligned tiles).  At least one of part / colsum. */
typedef struct mmmot_gemm_ares_args {
  const float* X; int ldx;
  const float* sc; const float* sh; int ldsc;   /* [G][ldsc] prologue scale / shift         */
  const void* W;                                 /* hl16 weights                             */
  const float* bias;                             /* [N] or NULL                              */
  const float* dbias; const int* tile_dbrow; int lddb;  /* per-tile extra bias row or NULL   */
  const int* tile_row0; const int* tile_nrows; const int* tile_group; /* [T]                 */
  float* part;                                   /* [2T][2][N] or NULL                       */
  const float* osc; const float* osh; int ldosc; /* [G][ldosc], needed with colsum           */
  float* colsum;                                 /* [2T][N] or NULL                          */
  int T; int N; int K;
  float oscale;
} mmmot_gemm_ares_args;
int mmmot_gemm_ares(const mmmot_gemm_ares_args* a, void* stream);
/* ABI 7.  tests: which kernel serves the K = 128 consumer pass (colsum only) of mmmot_gemm_ares - 0 = automatic (the
 * weights-in-registers kernel of csrc/gemm_wreg.hip when N % 512 == 0 and the launch fills the chip), 1 = the streaming
 * kernel only (the kernel before ABI 7), 2 = the register kernel whenever the layer is eligible.  Results do not depend on
 * it, bit for bit.
 * K = 64, N <= 512 (PointNet_v1.conv1, csrc/gemm_wres.hip), since round 6: 0 = the independent-wave kernel for a column-sum
 * pass that gives every wave at least four 64-row half tiles, the two-barrier weight-resident kernel otherwise; 1 = the
 * streaming kernel (N % 256 == 0); 2 = the independent-wave kernel for every column-sum pass; 3 = the two-barrier kernel
 * only (K = 128: as 0).  Bit for bit the same column sums. */
int mmmot_set_gemm_ares_variant(int v);

/* ---------------------------------------------------------------------------
 * GroupNorm statistics of a 1x1-conv output from the second moments of its INPUT (v2):
 * for v = W a + b with a = relu(X*sc + sh) over the rows of a group, per output channel
 *   mean = w.m + b,  var = w^T Cov(a) w  (m = E[a], Cov = E[a a^T] - m m^T)
 * so PointNet conv5 128->1024 + GroupNorm(1024,1024) (reference modules/point_net.py:138) needs ONE pass of
 * the 128->1024 GEMM (the normalise + ReLU + per-detection-sum pass of mmmot_gemm_ares) instead of two.
 * mmmot_gram_rows: per super-tile t (tile_row0/nrows/group, any row count, fp32 accumulation restarts every 128
 *   rows; the 128-row sums are added up in float64 (K = 128) / compensated fp32 pairs (K = 64)) Gout[t][K][K] = sum a a^T
 *   and Sout[t][K] = sum a, float64.  K = 64 or 128.  K = 128 writes the 32 x 32 blocks on and above the block
 *   diagonal only (the matrix is symmetric; mmmot_gn_finalize_gram mirrors) - the other blocks of Gout are not touched.
 * mmmot_gn_finalize_gram: sums the super-tiles of every group (grp_tile0 / grp_ntiles), forms Cov in float64
 *   and writes sc[g][n] = gamma[n]*rstd, sh[g][n] = beta[n] - mean*sc for the N output channels of
 *   W [N][K] (fp32, unscaled) / bias [N] (may be NULL).  work: G*(K*K+K) doubles. */
int mmmot_gram_rows(const float* X, int ldx, int K, const float* sc, const float* sh, int ldsc,
                    const int* tile_row0, const int* tile_nrows, const int* tile_group, int T,
                    double* Gout, double* Sout, void* stream);
/* ABI 10.  tests / A-B: which kernel serves mmmot_gram_rows at K = 128 - 0 = automatic (by the number of super-tiles),
 * 1 = the four-wave form (two workgroups per CU), 2 = the pipelined eight-wave form (one workgroup per CU, both LDS plane
 * pairs, two sub-tiles of rows in flight in registers).  Gout / Sout do not depend on it, bit for bit (same blocks, same
 * summation order per block and per column sum). */
int mmmot_set_gram128_variant(int v);
int mmmot_gn_finalize_gram(const double* Gp, const double* Sp, const int* grp_tile0, const int* grp_ntiles,
                           const int* grp_count, int G, int K, const float* W, const float* bias, int N,
                           const float* gamma, const float* beta, float eps, double* work, float* sc,
                           float* sh, void* stream);
/* ABI 9.  The same route for a layer with a gathered per-detection bias, v[p] = W a[p] + dbias[det(p)]
 * (PointNet_v1.conv1 after the 1088 -> 512 split, reference modules/point_net.py:26-28): the super-tiles handed to
 * mmmot_gram_rows (K = 64) are cut along detections - super-tile t lies inside detection tile_det[t] (a row of dbias
 * [.][lddb]) and holds tile_nrows[t] rows - so that its column sums Sp[t] give the cross term:
 *   mean = w.m + dbar,  var = w^T Cov w + 2 (sum_t d_t (w.Sp[t]) / P - (w.m) dbar) + (sum_t c_t d_t^2 / P - dbar^2),
 * dbar = sum_t c_t d_t / P, float64 throughout.  Replaces the statistics pass of mmmot_gemm_ares over every point.
 * W [N][64] fp32 (unscaled); work: G*(K*K+K) doubles; sc / sh [G][N]. */
int mmmot_gn_finalize_gram_dbias(const double* Gp, const double* Sp, const int* grp_tile0, const int* grp_ntiles,
                                 const int* grp_count, int G, const int* tile_nrows, const int* tile_det, int K,
                                 const float* W, const float* dbias, int lddb, int N, const float* gamma,
                                 const float* beta, float eps, double* work, float* sc, float* sh, void* stream);

/* GroupNorm statistics -> per-channel scale/shift (fp64 combine).
 * part is [T][2][ldp] = per-tile (sum, tile-centred M2) as written by
 * mmmot_gemm_rows / mmmot_pointnet_layer1 (the C channels start at part[0];
 * ldp >= C lets one GEMM's statistics feed several norms over channel
 * sub-ranges); tiles are merged with Chan et al.'s parallel-variance update in
 * fp64, so the result is robust when |mean| >> std.  A group's tiles must be
 * consecutive 128-row chunks of its rows, the last one partial;
 * group g owns tiles [grp_tile0[g], +grp_ntiles[g]) and grp_count[g] rows;
 * the C channels are split into NG normalisation groups (nn.GroupNorm(NG, C),
 * eps, biased variance).  sc[g][c] = gamma[c]*rstd, sh[g][c] = beta[c]-mean*sc.
 * tile_nrows (v2, may be NULL): rows of every tile when the tiles are NOT plain 128-row chunks
 * (detection-aligned tiles of the fused PointNet path).
 * Replaces the statistics half of every F.group_norm on the path. */
int mmmot_gn_finalize(const float* part, const int* grp_tile0, const int* grp_ntiles,
                      const int* grp_count, const int* tile_nrows, int G, int ldp, int C, int NG,
                      const float* gamma, const float* beta, float eps,
                      float* sc, float* sh, void* stream);

/* Strided/ragged segment mean with optional normalise+ReLU prologue:
 *   out[s][c] = mean_{t<count[s]} f(X[start[s] + t*stride[s]][c])
 * Replaces the per-detection Python pooling loops of
 * reference modules/point_net.py:32-39,139-148, adaptive_avg_pool2d of
 * modules/appear_net.py:15,30 and the mean(dim=-2/-1) of
 * modules/new_end.py:70-71.  C % 4 == 0. seg_group / sc / sh may be NULL.
 * seg_div (v2, may be NULL): divide the segment sum by seg_div[s] instead of count[s] (rows of X
 * that are already per-tile partial sums, see mmmot_gemm_args.colsum).
 * relu is a flag word (v4): bit 0 = ReLU after the affine, bit 1 = MAXIMUM over the segment instead of the
 * mean (end_mode 'max' of NewEndIndicator_v2, modules/new_end.py:72-74). */
int mmmot_segment_mean(const float* X, int ldx, int C,
                       const int* seg_start, const int* seg_count, const int* seg_stride,
                       const int* seg_group, const int* seg_div, int nseg,
                       const float* sc, const float* sh, int ldsc, int relu,
                       float* out, int ldo,
                       int hl16 /* 1: X rows are in the hl16 split-half format, 2: hq8 records (C % 32 == 0) */,
                       void* stream);

/* out[omap ? omap[r] : r] = post(act(sum_k f(X[r][k])*w[k] + b)),
 * post(v) = v - (v < thr) when use_thr (reference tracking_net.py:161-162).
 * Final 1-channel layers: gcn.py:66, new_end.py:58, tracking_net.py:99. */
int mmmot_rowdot(const float* X, int ldx, int K, const float* w, float b,
                 const float* sc, const float* sh, int ldsc,
                 const int* tile_row0, const int* tile_nrows, const int* tile_group, int T,
                 int act, int use_thr, float thr,
                 float* out, const int* omap, void* stream);

/* Per-row LayerNorm (= GroupNorm(1,C) on L x C x 1 x 1, appear_net.py:19-25):
 * Y[r][c] = act(gamma[c]*(X[r][c]-mean_r)*rstd_r + beta[c]).  C % 64 == 0, C <= 1024 */
int mmmot_row_layernorm(const float* X, int ldx, int C, const float* gamma, const float* beta,
                        float eps, int relu, float* Y, int ldy, int R, void* stream);

/* One SkipPool head (reference modules/appear_net.py:19-32) after the global average pool, in one launch (ABI 6):
 *   out[r][0:128] = relu(LN_128(w4 . relu(LN_C4(w1 . LN_C(P[r]) + c1)) + c4))        LN = GroupNorm(1, .) of a row
 * P [R][C] pooled stage features (row stride ldp), w1 [C4][C], w4 [128][C4] fp32, C % 64 == 0 <= 512, C4 in {64, 128};
 * out rows have stride ldo (the stage's 128-column slice of the [L][1024] feature matrix). */
int mmmot_skippool_head(const float* P, int ldp, int C, int C4, const float* g0, const float* b0, const float* w1,
                        const float* c1, const float* g2, const float* b2, const float* w4, const float* c4,
                        const float* g5, const float* b5, float eps, float* out, int ldo, int R, void* stream);
/* PointNet first shared-MLP layer with the K x K STN transform folded into the
 * weights (point_net.py:119-125): Y[p][0..63] = W[64][K] x[p] + b, plus
 * per-tile statistics like mmmot_gemm_rows.  K = 3 (xyz; every shipped config sets without_reflectivity) or 4
 * (xyz + reflectivity, tracking_net.py:41); X is [P][K] as delivered in det_info['points']. */
int mmmot_pointnet_layer1(const float* X, int K, const float* W, const float* bias,
                          float* Y, float* part,
                          const int* tile_row0, const int* tile_nrows, int T, void* stream);

/* PointNet shared-MLP layer with 64 input channels (feat.conv2 / conv3 / conv4, point_net.py:134-137):
 *   v[r][n] = sum_k relu(X[r][k]*sc[g][k] + sh[g][k]) * W[n][k] * oscale + bias[n],   k < 64, N = 64 | 128
 * Y[r][n] = v (raw, the next layer normalises it) and part[t] = per-tile (sum, tile-centred M2) of v - exactly what
 * mmmot_gemm_rows computes for amode = MMMOT_A_NORM_RELU, w_hl16 = 1, K = 64 - from persistent workgroups that keep
 * W16 ([N][8] hl16 units, host-scaled by 1/oscale) in LDS and prefetch the next tile's rows.  tile_group may be NULL. */
int mmmot_pn_mlp64(const float* X, int ldx, const float* sc, const float* sh, int ldsc, const void* W16,
                   float oscale, const float* bias, float* Y, int ldy, float* part, const int* tile_row0,
                   const int* tile_nrows, const int* tile_group, int T, int N, void* stream);

/* Y[r][c] = act(X[r][c]*sc[g][c] + sh[g][c]); g from the tile table. C % 4 == 0 */
int mmmot_affine_act(const float* X, int ldx, int C, const float* sc, const float* sh, int ldsc,
                     const int* tile_row0, const int* tile_nrows, const int* tile_group, int T,
                     int act, float* Y, int ldy, void* stream);

/* Fusion module A/B/C combine (fusion_net.py:31-42,62-70,85-92).
 * cat: [Lt][2C] = [image | lidar] features; F: [3][Lt][C] = image, lidar, fused.
 *  A: fused = Y0*sc0+sh0
 *  B: fused = (Y0*sc0+sh0) + (Y1*sc1+sh1)
 *  C: Y0=[gate_p|input_p](image), Y1=[gate_i|input_i](lidar), ld 2C:
 *     fused = (s(g0)*(i0*sc0+sh0) + s(g1)*(i1*sc1+sh1)) / (s(g0)+s(g1)) */
int mmmot_fusion_combine(int mode, const float* cat, const float* Y0, int ld0,
                         const float* Y1, int ld1,
                         const float* sc0, const float* sh0, const float* sc1, const float* sh1, int ldsc,
                         const int* tile_row0, const int* tile_nrows, const int* tile_group, int T,
                         float* F, int Lt, int C, void* stream);

/* Softmax modes over each group's N x M logit block (tracking_net.py:106-126).
 * logits/out rows are ordered (group, i, j); one workgroup per group. */
int mmmot_softmax_pairs(const float* logits, float* out,
                        const int* grp_row0, const int* grp_N, const int* grp_M, int G,
                        int max_nm /* max over groups of N+M (LDS sizing) */,
                        int mode, void* stream);

/* ---------------------------------------------------------------------------
 * Point-cloud gather (the step that produces det_info['points'] /
 * det_info['points_split'] for the path above).  Replaces the numba loop
 * _points_in_convex_polygon_3d_jit (reference point_cloud/geometry.py:96-114) and the per-box
 * boolean-index loop of read_and_prep_points (point_cloud/preprocess.py:70-96).
 *   pts    [P][F] fp32, F = 3 or 4, xyz first
 *   planes [NB][6][4] fp64 rows (nx, ny, nz, d) of convex polygons whose normals point inwards (3D boxes,
 *          image / 2D-box frustums); a point is inside iff ((x*nx + y*ny) + z*nz) + d < 0 for all six,
 *          evaluated in fp64 left to right without FMA contraction - the reference's arithmetic, so the
 *          decision is bit-identical.  NB <= 256 per call.
 * mmmot_points_count : cnt (ints, NB*ceil(P/256) + NB) and split [NB+1] = first output row of every polygon;
 *                      with pad_empty an empty polygon owns ONE all-zero row (preprocess.py:80-81).
 * mmmot_points_scatter: out [split[NB]][Fo] = per polygon its inside points in input order; Fo == F, or
 *                      Fo == 3 with F == 4 (the reflectivity column dropped, preprocess.py:97-100).
 * The caller reads split[NB] (one D2H of the split it hands to the host plan anyway) to size out. */
int mmmot_points_count(const float* pts, int P, int F, const double* planes, int NB, int pad_empty,
                       int* cnt, int* split, void* stream);
int mmmot_points_scatter(const float* pts, int P, int F, const double* planes, int NB, const int* cnt,
                         const int* split, float* out, int Fo, void* stream);
/* Batched form: NS sweeps (points of sweep s = rows [sweep_row0[s], sweep_row0[s+1]) of pts) and NPOLY polygons
 * (polygons of sweep s = [poly0[s], poly0[s+1]), at most 256 per sweep) in one launch; filt[s] >= 0 names a polygon
 * of `planes` every emitted point of sweep s must ALSO be inside (the image frustum of remove_outside_points fused
 * into the per-box test: same rows, same order as filtering first, no intermediate array).  Host-built tables:
 * blk_sweep [NBLK] / blk_first [NS] (256-point blocks per sweep), cnt_off [NPOLY] (first counter of each polygon;
 * polygon totals are stored at cnt[cnt_total + j]; behind them, 8-byte aligned, the count pass leaves four 64-bit
 * membership masks per counter for the scatter pass, so cnt holds ((cnt_total + NPOLY + 1) & ~1) + 8 * cnt_total
 * ints - since ABI 8).  split [NPOLY+1] as above. */
int mmmot_points_count_batched(const float* pts, int F, int NS, int NPOLY, int NBLK, int cnt_total,
                               const double* planes, const int* blk_sweep, const int* blk_first,
                               const int* sweep_row0, const int* poly0, const int* filt, const int* cnt_off,
                               int pad_empty, int* cnt, int* split, void* stream);
int mmmot_points_scatter_batched(const float* pts, int F, int NS, int NPOLY, int NBLK, int cnt_total,
                                 const double* planes, const int* blk_sweep, const int* blk_first,
                                 const int* sweep_row0, const int* poly0, const int* filt, const int* cnt_off,
                                 const int* cnt, const int* split, float* out, int Fo, void* stream);

/* ---------------------------------------------------------------------------
 * Per-detection image preparation: crop with zero padding -> antialiased bilinear
 * resize to S x S -> to_tensor -> normalize = the model input ``dets`` [N][3][S][S].  Replaces, per
 * detection, PIL img.crop(box).resize((S, S), Image.BILINEAR) + torchvision ToTensor/Normalize
 * (reference dataset/test_seq_dataset.py:212-218, utils/build_util.py:111-112,137-142) with Pillow's
 * arithmetic (Resample.c: double-precision triangle coefficients, 22-bit fixed point, two passes with an
 * 8-bit intermediate): the uint8 image and the float32 tensor are bit-identical to that pipeline.
 *   img   [H][W][3] uint8 (RGB frame, device)      boxes [N][4] int (x1, y1, x2, y2), may leave the frame
 *   kmax  >= 2*ceil(max(1, max box extent / S)) + 1 (coefficients per output index)
 *   mean_std [6] floats (mean rgb, std rgb)          work  N*2*S*(2+kmax) ints (scratch)
 *   out   [N][3][S][S] fp32                           out_u8 [N][S][S][3] uint8 or NULL (the resized image)
 * S <= 256. */
/* out (fp32 model input) or out_u8 may be NULL (not both).  mmmot_u8_normalize: ToTensor + Normalize of 8-bit crops
 * [N][S][S][3] -> fp32 [N][3][S][S] for the paths that cannot take bytes (exact-fp32 trunk, unfused first layer). */
int mmmot_u8_normalize(const unsigned char* u8, int N, int S, const float* mean_std, float* out, void* stream);
int mmmot_crop_resize_norm(const unsigned char* img, int H, int W, const int* boxes, int N, int S, int kmax,
                           const float* mean_std, int* work, float* out, unsigned char* out_u8, void* stream);

/* ---------------------------------------------------------------------------
 * Training backward of the pairwise block (first slice; ABI 5): gradients of
 * affinity_module.forward + NewEndIndicator_v2.forward + the softmax modes of TrackingNet.associate
 * (reference modules/gcn.py:68-82, new_end.py:62-82, tracking_net.py:106-126; the step that needs them is
 * tracking_model.py:50-66: forward -> loss -> backward).  Per layer  y = A W^T + b, yhat = (y - mean) rstd,
 * z = yhat gamma + beta, a = relu(z):
 *   dz = dA [z > 0];  dgamma = sum_r dz yhat;  dbeta = sum_r dz;  dy = rstd (gamma dz - m1 - yhat m2);
 *   dA_in = dy W (mmmot_gemm_rows with W^T);  dW = dy^T A_in, db = sum_r dy (mmmot_gemm_tn).
 * yhat is recomputed from the stored pre-norm y with sc1 = rstd, sh1 = -mean rstd, i.e. the output of
 * mmmot_gn_finalize called with gamma = 1, beta = 0.  All fp32 (exact fp32 MFMA in mmmot_gemm_tn).
 * ------------------------------------------------------------------------- */
/* pass 1: P[t][0][c] = sum over the rows of tile t of dz, P[t][1][c] = sum of dz*yhat  (P is [T][2][C]) */
int mmmot_gn_bwd_partial(const float* dA, int ldda, const float* Y, int ldy, int C, const float* sc1,
                         const float* sh1, int ldsc, const float* gamma, const float* beta, int relu,
                         const int* tile_row0, const int* tile_nrows, const int* tile_group, int T, float* P,
                         void* stream);
/* pass 2: S [G][2][C] = sums of P over the tiles of each group (mmmot_segment_mean with divisor 1) ->
 * M [G][2][C]: m1 = sum_{c in norm group} gamma_c S[g][0][c] / cnt, m2 likewise from S[g][1], broadcast to the
 * channels of the norm group; cnt = grp_count[g] * C / NG. */
int mmmot_gn_bwd_finalize(const float* S, const int* grp_count, int G, int C, int NG, const float* gamma, float* M,
                          void* stream);
/* pass 3: dY[r][c] = sc1[g][c] * (gamma_c dz - M[g][0][c] - yhat M[g][1][c]) */
int mmmot_gn_bwd_apply(const float* dA, int ldda, const float* Y, int ldy, int C, const float* sc1, const float* sh1,
                       int ldsc, const float* gamma, const float* beta, int relu, const float* M,
                       const int* tile_row0, const int* tile_nrows, const int* tile_group, int T, float* dY, int lddy,
                       void* stream);
/* dW[n][k] = sum_r dY[r][n] * A(r,k), db[n] = sum_r dY[r][n]; A(r,k) is regenerated like the forward's A operand
 * (MMMOT_A_PLAIN / MMMOT_A_NORM_RELU / MMMOT_A_PAIR).  N % 64 == 0, K % 64 == 0.  Deterministic (row split with
 * per-share partial outputs, no atomics). */
typedef struct mmmot_gemm_tn_args {
  const float* dY; int lddy;            /* [rows][N] gradient of the layer's pre-norm output */
  const float* X; int ldx;              /* PLAIN / NORM_RELU source rows                      */
  const float* sc; const float* sh; int ldsc;
  const float* FA; const float* FB; int ldf;   /* PAIR operands                               */
  const int* tile_row0; const int* tile_nrows; const int* tile_group;
  const int* grp_row0; const int* grp_M; const int* grp_aoff; const int* grp_boff;
  int T; int N; int K; int amode; int pairop;
  int nsplit;                           /* >= 1: the tiles are split into nsplit contiguous shares, share s writes
                                           its partial sums to dW + s*N*K and db + s*N (the caller adds them up)   */
  float* dW;                            /* [nsplit][N][K]                                     */
  float* db;                            /* [nsplit][N] or NULL                                */
} mmmot_gemm_tn_args;
int mmmot_gemm_tn(const mmmot_gemm_tn_args* a, void* stream);
/* mmmot_gemm_tn on the fp16 matrix cores (3-term hi/lo split, fp32 accumulation; csrc/gemm_tn_f16.hip): same arguments
 * and results contract (tiles of at most 128 rows).  dyamax: device pointer to max |dY| (mmmot_absmax below) - dY is
 * scaled by the power of two that puts its maximum at 2^10 before the fp16 split (gradients of 1e-6 would otherwise sit
 * in fp16's subnormals) and the result scaled back exactly; NULL = no scaling. */
int mmmot_gemm_tn_f16(const mmmot_gemm_tn_args* a, const float* dyamax, void* stream);
/* *out (one device float) = max |X[r][c]| over a [R][C] tensor with row stride ld; C % 4 == 0.  Asynchronous. */
int mmmot_absmax(const float* X, int ld, long R, int C, float* out, void* stream);
/* backward of the pairwise operand generation (gcn.py:6-41): side 0 accumulates d op / d a over j into
 * dF[aoff[g] + i], side 1 d op / d b over i into dF[boff[g] + j]; one workgroup per entry of (blk_group, blk_idx)
 * = every (group, i) resp. (group, j); dF is accumulated into (+=), rows of one launch are distinct. */
int mmmot_pair_bwd(const float* dX, int lddx, const float* F, int ldf, float* dF, int lddf, int C,
                   const int* grp_row0, const int* grp_N, const int* grp_M, const int* grp_aoff,
                   const int* grp_boff, const int* blk_group, const int* blk_idx, int nblk, int pairop, int side,
                   void* stream);
/* backward of the strided means that make the new / end vectors (new_end.py:70-71):
 * dA[(g,i,j)][c] = dV[vrow0[g] + j][c] / N + dV[vrow0[g] + M + i][c] / M */
int mmmot_pair_expand_bwd(const float* dV, int lddv, float* dA, int ldda, int C, const int* tile_row0,
                          const int* tile_nrows, const int* tile_group, int T, const int* grp_row0,
                          const int* grp_N, const int* grp_M, const int* grp_vrow0, void* stream);
/* backward of mmmot_rowdot (a = relu(X*sc + sh), act NONE or SIGMOID): gpre[r] = gout[gidx ? gidx[r] : r] * act';
 * dA[r][k] = gpre[r] w[k]; PW[t][k] = sum_{r in tile t} gpre[r] a(r,k) (k < K), PW[t][K] = sum_r gpre[r]. */
int mmmot_rowdot_bwd(const float* X, int ldx, int K, const float* w, float b, const float* sc, const float* sh,
                     int ldsc, const int* tile_row0, const int* tile_nrows, const int* tile_group, int T, int act,
                     const float* gout, const int* gidx, float* dA, int ldda, float* PW, int ldpw, void* stream);
/* backward of mmmot_softmax_pairs: dlogits from dout, the row / column softmaxes are recomputed from the logits */
int mmmot_softmax_pairs_bwd(const float* logits, const float* dout, float* dlogits, const int* grp_row0,
                            const int* grp_N, const int* grp_M, int G, int max_nm, int mode, void* stream);

/* backward of the fusion module C combine (fusion_net.py:31-42; Y_j = [gate_j | input_j] rows as in the forward):
 * DY_j[:, 0:C] = dL/d gate pre-activation, DN_j = dL/d normalised input (goes through mmmot_gn_bwd_* with relu = 0
 * into DY_j[:, C:2C]).  Modes A / B need no kernel (dn = dfused). */
int mmmot_fusion_c_bwd(const float* dFu, const float* Y0, int ld0, const float* Y1, int ld1, const float* sc0,
                       const float* sh0, const float* sc1, const float* sh1, int ldsc, const int* tile_row0,
                       const int* tile_nrows, const int* tile_group, int T, float* DY0, float* DY1, int lddy,
                       float* DN0, float* DN1, int C, void* stream);
/* Y[r][c] = A[r][c] + B[r][c] (two gradient paths into one feature); C % 4 == 0 */
int mmmot_add_rows(const float* A, int lda, const float* B, int ldb, float* Y, int ldy, long R, int C, void* stream);

/* ---------------------------------------------------------------------------
 * Training step, second slice (ABI 6, additive): LiDAR-encoder backward helpers and the loss (csrc/train.hip).
 *
 * mmmot_rows_gather_scale: backward of the per-detection average pools of PointNet (reference
 * modules/point_net.py:32-39 and 139-148: `avg_pool(x[:, :, start:end])`, for conv5 followed by `.repeat` back to the
 * points): X[r][c] = S[rowidx[r]][c] * scale[rowidx[r]]  (scale = 1 / points of the detection; NULL = 1).  C % 4 == 0. */
int mmmot_rows_gather_scale(const float* S, int lds, const int* rowidx, const float* scale, float* X, int ldx, long R,
                            int C, void* stream);
/* ---- third slice: the VGG16-BN trunk in TRAINING mode (reference modules/vgg.py:67-80 under .train(): BatchNorm2d on
 * the statistics of the batch) and its backward; fp32 NHWC, exact fp32 matrix cores (conv3x3.hip, csrc/train_vgg.hip) ----
 * mmmot_conv3x3_raw: out = conv3x3(in, wp) + bias (no BatchNorm fold, no ReLU); layouts of mmmot_conv3x3_bn_relu.  Also
 * the input gradient of a layer: in = dZ, wp = the layer's weights with taps flipped and Cin / Cout swapped, bias = 0. */
int mmmot_conv3x3_raw(const float* in, const float* wp, const float* bias, float* out, int L, int H, int W, int Cin,
                      int Cout, int first, void* stream);
/* per-tile statistics of an existing tensor: part[t][0][c] = sum_r Y[r][c], part[t][1][c] = sum_r (Y[r][c] - tile mean)^2
 * (the `part` contract of mmmot_gemm_rows; input of mmmot_gn_finalize).  C % 4 == 0. */
int mmmot_rows_stats(const float* Y, int ldy, int C, const int* tile_row0, const int* tile_nrows, int T, float* part,
                     void* stream);
/* A = relu(Z * sc + sh) per channel (sc / sh [C]: the batch statistics folded by mmmot_gn_finalize), followed by the
 * 2x2 / stride-2 max-pool with floor semantics when pool != 0.  Z NHWC [L][H][W][C] -> A [L][H >> pool][W >> pool][C]. */
int mmmot_bn_relu_pool(const float* Z, int C, const float* sc, const float* sh, int L, int H, int W, int pool, float* A,
                       void* stream);
/* backward of that max-pool: dA [L][H][W][C] = dP of the window at the window's first maximum of relu(Z * sc + sh)
 * (PyTorch's tie rule), zero elsewhere (odd maps: the last row / column belongs to no window). */
int mmmot_maxpool_bwd(const float* Z, int C, const float* sc, const float* sh, const float* dP, int L, int H, int W,
                      float* dA, void* stream);
/* weight gradient of a 3x3 convolution: dW[s][tap][co][ci] = sum over share s of the pixels of dZ[p][co] * A[p + off(tap)][ci]
 * (zero padding); nsplit shares, the caller adds them.  Cin % 64 == 0, Cout % 64 == 0. */
int mmmot_conv3x3_wgrad(const float* dZ, const float* A, int L, int H, int W, int Cin, int Cout, int nsplit, float* dW,
                        void* stream);
/* ABI 7.  The same weight gradient on the fp16 matrix cores (3-term hi/lo split, the pixel axis as the K of the MFMA like
 * mmmot_gemm_tn_f16): dZ is scaled by the power of two that puts dzamax[0] = max |dZ| (device scalar, mmmot_absmax; NULL =
 * no scaling) at 2^10 and the result scaled back exactly.  Same output layout; fp32-class (products exact to 2^-22).
 * Since round 6 a workgroup owns a (channel-tile pair, tap ROW, share): nsplit such that (Cout/T)(Cin/T) * 3 * nsplit fills
 * the CUs once (T = 128 where the channel count allows, else 64) is the fast choice (mmmot_amd/train_vgg.py:_wgrad_shares);
 * a share sums 64-pixel chunks in fp32 on the matrix cores' accumulators. */
int mmmot_conv3x3_wgrad_f16(const float* dZ, const float* A, int L, int H, int W, int Cin, int Cout, int nsplit, float* dW,
                            const float* dzamax, void* stream);
/* first layer (NCHW crops X [L][3][H][W], Cout = 64): PW[b][co][28] partial sums over block b's pixels of
 * (dW1[co][k = tap * 3 + colour], k < 27 | db1[co]); the caller adds the nblocks partials. */
int mmmot_conv3x3_first_wgrad(const float* dZ, const float* X, int L, int H, int W, float* PW, int nblocks, void* stream);
/* weight / bias gradient of PointNetfeatGN.conv1 with the first transform folded in (forward: mmmot_pointnet_layer1;
 * reference modules/point_net.py:119-125): per-tile partials PW[t][c * (K + 1) + k] = sum_r dY[r][c] X[r][k] (k < K),
 * [..][K] = sum_r dY[r][c]; dY [P][64], X [P][K], K = 3 | 4.  The caller adds the T partial rows. */
int mmmot_pointnet_layer1_bwd(const float* dY, const float* X, int K, const int* tile_row0, const int* tile_nrows, int T,
                              float* PW, void* stream);
/* One term of TrackingLoss (reference cost.py:134-185: DetLoss :97-131 on det / new / end scores, LinkLoss :66-94 on a
 * link block) with its gradient: x [R][C] scores (R modality rows), y [C] target shared by the rows, mask
 * m(c) = mr(mrow[c / M]) * mc(mcol[c % M]) (NULL vectors: 1) with mask_mode 1: v == 1 (LinkLoss' `gt_det == 1`),
 * 2: v != ignore (DetLoss' ignore_index), 0: no mask.  g[r][c] = scale * dl/dx, PL[b] = scale * sum of l over block b's
 * elements (b < nblocks), added to the value PL[b] holds when accumulate != 0 (one block: the terms of a loss add up in
 * PL[0], launch after launch on one stream); scale = ratio / (R * C) reproduces the reference's reduction='mean'. */
int mmmot_score_loss(const float* x, int ldx, const float* y, const float* mrow, const float* mcol, int M, int mask_mode,
                     float ignore, int kind, float scale, int R, int C, float* g, int ldg, float* PL, int nblocks,
                     int accumulate, void* stream);
/* ABI 9.  The 'ghm' type of DetLoss (reference cost.py:105-110,126-128 -> modules/ghm_loss.py:15-61, GHMC_Loss):
 * gradient-harmonised BCE-with-logits of x [R][C] against y [C] (shared by the R rows), entries with y == ignore masked.
 * |sigmoid(x) - y| of the valid elements is histogrammed over `bins` (<= 64) equal bins of [0, 1]; acc_sum [bins]
 * (float64, device) is the module's running state, updated in place: acc <- momentum * acc + (1 - momentum) * count for
 * every non-empty bin (momentum == 0: acc = count, acc_sum untouched); weight = float(tot / acc) / (non-empty bins),
 * tot = max(valid elements, 1).  g[r][c] = scale * w * (sigmoid(x) - y) / tot (weights are constants of the graph),
 * PL[0] (+)= scale * sum(w * bce) / tot (accumulate != 0: added, like mmmot_score_loss).  One workgroup; R * C <= 2^24.
 * momentum travels as fp32 and is widened to float64 on the device: bit-faithful to the
 * reference's Python-float arithmetic for values fp32 represents exactly (the reference's hard-wired 0.75); any other value
 * (0.9, say) differs from it by ~1e-8 relative per step. */
int mmmot_ghm_loss(const float* x, int ldx, const float* y, float ignore, float scale, int R, int C, int bins,
                   float momentum, double* acc_sum, float* g, int ldg, float* PL, int accumulate, void* stream);

/* MFMA fragment-layout self test: C[32][32] = A[32][K] * B[32][K]^T through
 * the same fragment mapping the GEMM kernels use (K % 8 == 0). */
int mmmot_selftest_mfma(const float* A, const float* B, float* C, int K, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MMMOT_HIP_H */
