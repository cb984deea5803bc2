/* cellseg_hip.h -- C ABI of libcellseg_hip.so, the MI355X (gfx950) kernel library behind the
 * per-tile CNN hot path of Newiz430/CellSegmentation.
 *
 * The reference has no FFI: its hot path is Python calling torch.nn modules (model/resnet.py,
 * model/resnext.py, model/efficientnet.py, train/train.py, train/losses.py, metrics/metrics.py,
 * inference.py).  Each entry point below replaces the ATen op(s) that a cited reference line
 * dispatches; the Python host mirror in cellsegmentation_amd/ binds them with ctypes
 * (see INTEGRATION.md for the stub a reference maintainer would add).
 *
 * Conventions
 *  - plain pointers + sizes only; every pointer is a DEVICE pointer unless named host_*;
 *  - activations are NHWC, dtype CS_F32 (parity mode, exact-f32 MFMA) or CS_BF16 (throughput
 *    mode, fp32 accumulate); channel counts are multiples of the 16-byte chunk
 *    (4 for CS_F32, 8 for CS_BF16); parameters/gradients/statistics are fp32 (torch layouts);
 *  - nothing allocates, frees or synchronises: work is enqueued on `stream` (a hipStream_t
 *    passed as void*); workspaces are caller-owned;
 *  - return value: CS_OK or a negative CS_ERR_*; cs_last_error() gives the message.
 */
#ifndef CELLSEG_HIP_H_
#define CELLSEG_HIP_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { CS_F32 = 0, CS_BF16 = 1 };
enum { CS_OK = 0, CS_ERR_INVALID_ARG = -1, CS_ERR_LAUNCH = -2, CS_ERR_UNSUPPORTED = -3 };
enum { CS_ACT_NONE = 0, CS_ACT_RELU = 1, CS_ACT_SILU = 2, CS_ACT_SIGMOID = 3 };

int cs_abi_version(void);
const char* cs_last_error(void);
/* Name of the convolution-family kernel instantiation the calling thread launched last, e.g. "igemm_dma_kernel<bf16,64,128,1,false,true>",
 * "conv2_halo_kernel<9,3,4,1,4,4>", "wgrad_dma_kernel<128,3,true,32>": the names rocprofv3 reports (template arguments without spaces),
 * so that bench.py's roofline entries can be checked against profiles/ (tools/check_bench_vs_profile.py). */
const char* cs_last_conv_variant(void);

/* Geometry of one 2-D convolution (torch.nn.Conv2d as used at model/resnet.py:20,23,51,53,55,
 * 111,183,198,164).  Input N x H x W x C (C = stored/padded channels), output N x P x Q x K. */
typedef struct CsConvGeom {
    int32_t N, H, W, C;
    int32_t K;
    int32_t R, S;       /* kernel height, width */
    int32_t stride, pad;
    int32_t P, Q;       /* output height, width */
    int32_t groups;     /* 0 or 1: dense; > 1: grouped convolution in slab-dense form (ResNeXt, see cs_weight_prep_grouped) */
} CsConvGeom;

/* ---- layout / precision changes at the module boundary ------------------------------------- */
/* x[N][C][H][W] fp32 (the reference's input layout, dataset/dataset.py:78-83 output) ->
 * y[N][H][W][Cp] dtype, channels C..Cp-1 zero-filled. */
int cs_nchw_to_nhwc(const float* x, void* y, int dtype, int N, int C, int H, int W, int Cp, void* stream);
/* y[N][H][W][Cp] dtype -> x[N][C][H][W] fp32 (first C channels).  Used for segmentation logits
 * (model/resnet.py:301-303) and to hand gradients back in torch layout. */
int cs_nhwc_to_nchw(const void* y, int dtype, float* x, int N, int C, int H, int W, int Cp, void* stream);

/* ---- BatchNorm2d in eval mode folded to per-channel scale/shift (resnet.py:254-258: the
 * freeze_bn path runs every BN with running statistics) --------------------------------------
 * scale = gamma*rsqrt(var+eps), shift = beta + (conv_bias - mean)*scale, rstd = rsqrt(var+eps);
 * conv_bias nullable (the biased 3x3 convs of upsample_conv, resnet.py:198). */
int cs_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
               const float* conv_bias, float* scale, float* shift, float* rstd, int C, void* stream);

/* ---- BatchNorm in train mode (batch statistics; model.train(), resnet.py:21,52,112,184,199 and the
 * BatchNorm1d of the image heads, resnet.py:134,138) over NHWC rows z[M][C] ----------------------
 * stats: the per-channel pair (sum, sum of squares) as an EXACT ACCUMULATOR (round 5, ABI 6): cs_bn_accum_words(C) = 6 C + 1
 * zero-initialised 8-byte words -- every contribution is split into three limbs (multiples of 2^0, 2^-40, 2^-80, each below 2^40 of
 * its unit) that are added with one fp64 atomic each: every partial sum of a limb stays exactly representable, so each addition is
 * exact and the totals do not depend on the order the workgroups arrive in (two runs of one step give the same bits; the reference's
 * CPU path is deterministic too) and no bit >= 2^-80 of a contribution is lost; the last word is a sticky
 * "a contribution was NaN / infinite / >= 2^40" flag that makes every total read as NaN.  The words are opaque: kernels of this
 * library produce (cs_bn_stats, cs_conv2d_fwd with `stats`, cs_bn_partial_fold, cs_bn_bwd_reduce) and consume (cs_bn_finalize,
 * cs_bn_apply_stats, cs_bn_bwd_apply) them; cs_bn_accum_read gives the two totals per channel as fp64 [2][C].  The parameters
 * keep their historical `double*` type. */
size_t cs_bn_accum_words(int C);
int cs_bn_accum_read(const double* accum, int C, double* out, void* stream);
int cs_bn_stats(const void* z, int dtype, long long M, int C, double* stats, double* workspace, void* stream);
/* `workspace` of cs_bn_stats / cs_bn_bwd_reduce (nullable, cs_bn_partial_workspace(M, C) bytes -- 0 where the launch has so few
 * row blocks and channels that it adds its sums straight into the accumulator and wants none): per-workgroup partial sums are
 * written there and folded by a second small launch instead of ~1000 atomics per channel (3x faster on large tensors). */
size_t cs_bn_partial_workspace(long long M, int C);
/* mean, rstd = 1/sqrt(biased var+eps); running_* (nullable) updated in place with `momentum` and the
 * unbiased variance, as nn.BatchNorm2d does. */
int cs_bn_finalize(const double* stats, long long M, float eps, float momentum, float* running_mean,
                   float* running_var, float* mean_out, float* rstd_out, int C, void* stream);
/* y = act( gamma*(z-mean)*rstd + beta + residual ); gamma/beta/residual nullable. */
int cs_bn_apply(const void* z, int dtype, const float* mean, const float* rstd, const float* gamma,
                const float* beta, const void* residual, int act, void* y, long long M, int C, void* stream);
/* cs_bn_finalize + cs_bn_apply in ONE launch (train-mode forward of nn.BatchNorm2d, model/resnet.py:150 / efficientnet.py:93 with
 * module.training): every thread derives mean and 1/sqrt(var + eps) of its channels from the accumulator `stats` (above) with the same
 * arithmetic as cs_bn_finalize (bit-identical results), workgroup 0 writes mean_out / rstd_out (kept for backward) and updates the
 * running statistics (nullable). */
int cs_bn_apply_stats(const void* z, int dtype, const double* stats, float eps, float momentum, float* running_mean,
                      float* running_var, const float* gamma, const float* beta, const void* residual, int act, void* y,
                      float* mean_out, float* rstd_out, long long M, int C, void* stream);
/* sums: an exact accumulator like `stats` above (cs_bn_accum_words(C) zeroed words): sum g, sum g*xhat with xhat=(z-mean)*rstd and g = dy, or for
 * act==CS_ACT_SILU g = dy*silu'(gamma*xhat+beta) (the activation that follows the BN; ReLU gradients are
 * already masked by the consumers, see engine.py). gamma/beta nullable (1/0).
 * Two flags may be OR-ed into `act` of cs_bn_bwd_reduce / cs_bn_bwd_apply (the BatchNorm1d layers of the image heads,
 * resnet.py:132-152, whose backward has no convolution epilogue to do the masking):
 *   CS_BN_BWD_OWN_RELU  g = dy * [gamma*xhat + beta > 0]: the ReLU that directly follows this normalisation;
 *   CS_BN_BWD_FROZEN    (apply) the statistics are running statistics (module.eval()): dz = gamma*rstd*g, no batch terms. */
#define CS_BN_BWD_OWN_RELU 0x100
#define CS_BN_BWD_FROZEN 0x200
int cs_bn_bwd_reduce(const void* dy, const void* z, int dtype, const float* mean, const float* rstd,
                     const float* gamma, const float* beta, int act, long long M, int C, double* sums, double* workspace, void* stream);
/* dz = gamma*rstd*( g - sums0/M - xhat*sums1/M ); dgamma=sums1, dbeta=sums0 (fp32, nullable). */
int cs_bn_bwd_apply(const void* dy, const void* z, int dtype, const float* mean, const float* rstd,
                    const float* gamma, const float* beta, int act, const double* sums, long long M, int C, void* dz,
                    float* dgamma, float* dbeta, void* stream);

/* ---- weight staging -------------------------------------------------------------------------
 * w[K][Cin][R][S] fp32 (torch Conv2d.weight) times optional per-K `scale` ->
 *   w_khwc [Kp][R][S][Cp]  (forward operand)   if non-NULL
 *   w_chwk [Cp][R][S][Kp]  (dgrad operand)     if non-NULL
 * Cp/Kp = stored (padded) channel counts; padded rows/columns are zero-filled. */
int cs_weight_prep(const float* w, const float* scale, int dtype, int K, int Cin, int R, int S, int Cp, int Kp,
                   void* w_khwc, void* w_chwk, void* stream);

/* cs_bn_fold + cs_weight_prep in one launch for an eval-mode Conv2d+BatchNorm2d (scale/shift/rstd are [Kp]). */
int cs_stage_conv_bn(const float* w, const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                     const float* conv_bias, int dtype, int K, int Cin, int R, int S, int Cp, int Kp, void* w_khwc,
                     void* w_chwk, float* scale, float* shift, float* rstd, void* stream);

/* The same for EVERY eval-mode Conv2d+BatchNorm2d of a network in ONE launch (a training step re-stages all ~53 layers of a
 * ResNet-50 after each optimizer update: 53 launches of ~7 us otherwise).  `desc` is a DEVICE array of n descriptors, built once
 * by the host for a fixed set of parameter / staging buffers; block0 = prefix sum of cs_stage_conv_bn_blocks() over the layers,
 * total_blocks = the sum. */
typedef struct CsStageDesc {
    const float *w, *gamma, *beta, *mean, *var, *conv_bias;   /* gamma / beta / conv_bias nullable; mean == var == NULL: no BatchNorm is folded
                                                                 * (train-mode BN layers: scale = 1, shift = conv_bias or 0) */
    void *w_khwc, *w_chwk;                                     /* either nullable */
    float *scale, *shift, *rstd;                               /* [Kp] */
    float eps;
    int32_t K, Cin, R, S, Cp, Kp;
    int32_t block0;
    /* nonzero: write that operand in the MFMA-fragment order cs_conv2d_fwd_packed / cs_conv2d_dgrad_packed read (what
     * cs_pack_conv_weights produces from the plain layout; needs Kp % 32 == 0 and Cp % 64 == 0 for fwd, Cp % 32 == 0 and
     * Kp % 64 == 0 for bwd) instead of w_khwc / w_chwk order; the buffer sizes are the same */
    int32_t fwd_packed, bwd_packed;
} CsStageDesc;
int cs_stage_conv_bn_blocks(int K, int Cin, int R, int S, int Cp, int Kp, int want_fwd, int want_bwd);
int cs_stage_conv_bn_multi(const CsStageDesc* desc, int n, int total_blocks, int dtype, void* stream);

/* ---- convolution family (implicit GEMM on MFMA) ---------------------------------------------
 * forward: y = act( scale[k]*conv(x,w) + shift[k] + residual ), any of scale/shift/residual NULL.
 *   Fuses Conv2d+BatchNorm2d(eval)+ReLU(+residual add) of BasicBlock/Bottleneck.forward
 *   (resnet.py:28-43, 60-78) and Conv2d(bias) of upsample_conv/seg_out_conv (resnet.py:195-200,164).
 *   stats (nullable, an exact accumulator of cs_bn_accum_words(K) zeroed words -- see cs_bn_stats --, needs `workspace`): accumulates sum and sum of squares of the STORED output per channel
 *   (BatchNorm2d train-mode batch statistics). */
int cs_conv2d_fwd(const CsConvGeom* g, int dtype, const void* x, const void* w_khwc, const float* scale,
                  const float* shift, const void* residual, int act, void* y, double* stats, void* workspace,
                  void* stream);
/* bytes of `workspace` that cs_conv2d_fwd (stats) / cs_conv2d_dgrad (colsum) need for M destination
 * pixels x n_out destination channels: per-workgroup partial sums, folded without atomics. */
size_t cs_conv2d_stats_workspace(long long M, int n_out);
/* Which BM x BN output tile the fwd/dgrad dispatcher uses for M output pixels x n_out channels
 * (returns BM*1000+BN); lets bench.py / profiles name the kernel instantiation that ran. */
int cs_igemm_tile(long long M, int n_out);
#ifdef CS_AB_SWITCHES
/* A/B flavour only (`make AB=1` -> libcellseg_hip_ab.so; not part of the production ABI).  Staging path of the fwd/dgrad
 * kernel: 0 (default) = LDS-DMA (`buffer_load ... lds`) whenever both operands are < 2 GiB; 1 = register-staged everywhere
 * (the production rule for operands >= 2 GiB, forced here so that small test shapes reach it); 3 = LDS-DMA plus the
 * experimental persistent streaming kernel for short-K 1x1 convolutions.  Returns the previous setting. */
int cs_set_igemm_path(int path);
#endif
/* data gradient: dx = ( conv_transpose(dy, w) + add ) * [mask > 0]; add/mask nullable, both shaped like x.
 *   `mask` is the conv's own input activation when that input came out of a ReLU (the ReLU backward of
 *   resnet.py:41/76 fused here). colsum (nullable fp32 [C]) accumulates per-channel sums of the stored dx. */
int cs_conv2d_dgrad(const CsConvGeom* g, int dtype, const void* dy, const void* w_chwk, const void* add,
                    const void* mask, void* dx, float* colsum, void* workspace, void* stream);
/* One bit per element instead of a 16-bit mask operand (the ReLU masks are 1/6 of a training step's HBM traffic otherwise):
 * cs_conv2d_fwd_bits also writes the plane `positive_bits`: M * K / 8 bytes (M = N*P*Q pixels, K % 32 == 0), CHANNEL-BLOCK-MAJOR since
 * round 5 (ABI 6): the 32-bit word (k / 32) * M + pixel holds channels 32 (k / 32) .. + 31 of that pixel, bit k % 32 = "the stored y is
 * > 0" -- a wave's 32 words of a 32-pixel tile are 128 contiguous bytes (the pixel-major layout of rounds 1-4 put them K / 8 bytes apart:
 * 32 sectors per store instruction).  cs_conv2d_dgrad_bits takes such a plane (of a tensor shaped like x: N*H*W pixels, C channels,
 * C % 32 == 0) in place of `mask`.  Every producer / consumer of bit planes in this library uses this layout (cs_conv2d_fwd_packed,
 * cs_stem_fwd_packed, cs_conv2d_dgrad_packed, cs_positive_bits). */
int cs_conv2d_fwd_bits(const CsConvGeom* g, int dtype, const void* x, const void* w_khwc, const float* scale, const float* shift,
                       const void* residual, int act, void* y, uint8_t* positive_bits, void* stream);
int cs_conv2d_dgrad_bits(const CsConvGeom* g, int dtype, const void* dy, const void* w_chwk, const void* add,
                         const uint8_t* mask_bits, void* dx, float* colsum, void* workspace, void* stream);
/* Deferred column sums: with colsum == NULL and workspace != NULL (ungrouped launches of stride 1, and of stride 2 where
 * cs_conv2d_dgrad_partial_rows(g) > 0: the parity classes then go out as one launch with rows numbered across them) cs_conv2d_dgrad leaves the
 * per-workgroup partial rows in `workspace` -- row r holds the sums of destination-pixel tile r at [r * 2*C + c] -- and skips the
 * fold.  cs_conv2d_dgrad_partial_rows gives the row count; cs_fold_partial_rows folds one such buffer (out[c] += sum over rows, in a
 * fixed order: the unused second half of the first <= 32 partial rows serves as scratch, so `partial` is written to);
 * cs_wgrad_finalize_batched takes the partial rows as they are and folds each channel's column inside its own launch. */
int cs_conv2d_dgrad_partial_rows(const CsConvGeom* g);
int cs_fold_partial_rows(const float* partial, int rows, int n_out, float* out, void* stream);
/* weight gradient, raw split-K partials: dw_khwc[nsplit][K][R][S][Cp] fp32, slab z = sum over pixel slice z of
 *   dy (x) im2col(x), written with plain stores (no zero-fill needed; nsplit = cs_conv2d_wgrad_splits(g, grouped));
 *   cs_wgrad_finalize folds the slabs in a fixed order (bitwise reproducible). */
int cs_conv2d_wgrad_splits(const CsConvGeom* g, int grouped);
int cs_conv2d_wgrad(const CsConvGeom* g, int dtype, const void* x, const void* dy, float* dw_khwc,
                    int use_tr_read, void* stream);
/* Batched weight gradient for n_items layers of IDENTICAL geometry (the repeated blocks of a ResNet stage): one launch,
 * blockIdx.z = item*nsplit + slice, nsplit = cs_conv2d_wgrad_batched_splits(g, dtype, n_items) (fewer slices per layer ->
 * proportionally less partial-slab traffic).  x_tab / dy_tab / dw_tab: HOST arrays of n_items (<= 8) device pointers (they
 * are passed to the kernel by value, no device table, no copy); dw_tab[i] is a [nsplit][K][R][S][Cp] fp32 buffer. */
int cs_conv2d_wgrad_batched_splits(const CsConvGeom* g, int dtype, int n_items);
/* 1 when cs_conv2d_wgrad_batched serves this geometry with the second-generation kernel (wgrad_v2.hip: bf16, stride-1 pad-1 3x3,
 * C and K multiples of 64, image width <= 158): callers with a single layer then prefer the batched entry (n_items = 1). */
int cs_conv2d_wgrad2_supported(const CsConvGeom* g, int dtype);
int cs_conv2d_wgrad_batched(const CsConvGeom* g, int dtype, const void* const* x_tab, const void* const* dy_tab,
                            float* const* dw_tab, int n_items, int use_tr_read, void* stream);
/* Batched cs_wgrad_finalize (eval-BN or plain conv, no bias, not grouped) in ONE launch: `tables` = HOST array of 9*n_items
 * (n <= 8) pointers: [raw | w | scale | rstd | mean | gsum | dw | dgamma | dbeta] x n_items (scale..gsum, dgamma, dbeta used only
 * with want_bn).  gsum[i] is either the [K] vector (gsum_rows[i] == 0) or the per-workgroup partial rows a data-gradient
 * launch left behind ([gsum_rows[i]][gsum_stride] fp32, cs_conv2d_partial_rows / cs_conv2d_packed_partial_rows): the kernel
 * folds them itself.  gsum_rows: HOST array of n_items ints, or NULL (all vectors). */
int cs_wgrad_finalize_batched(const float* const* tables, const int* gsum_rows, int gsum_stride, int n_items, int nsplit, int Kp, int K,
                              int Cin, int R, int S, int Cp, int want_bn, void* stream);
/* dw[K][Cin][R][S] (torch layout, ACCUMULATED into when accumulate!=0) = scale[k]*dw_khwc[k][r][s][c];
 * dbias[k] = scale[k]*gsum[k] (conv bias); and, when dgamma/dbeta non-NULL, the eval-mode BatchNorm parameter gradients
 *   dbeta[k] = gsum[k];  dgamma[k] = rstd[k]*( sum_j w[k][j]*dw_raw[k][j] - mean[k]*gsum[k] ). */
int cs_wgrad_finalize(const float* dw_khwc, int nsplit, int Kp, const float* w, const float* scale, const float* rstd,
                      const float* mean, const float* gsum, int K, int Cin, int R, int S, int Cp,
                      float* dw, float* dbias, float* dgamma, float* dbeta, float* dot_ws /* fp32 [K] scratch ZEROED by the caller, needed with dgamma */,
                      int accumulate, void* stream);
/* ---- grouped 3x3 convolution (ResNeXt, model/resnext.py:16-19,85: groups=32) in slab-dense form ------------
 * C == K, C % 64 == 0, Cg = C/groups divides 64.  Each 64-channel destination tile contracts only over its
 * own 64 source channels with weights that are block-diagonal inside the slab:
 *   cs_weight_prep_grouped: w[K][Cg][R][S] fp32 -> w_khwc[K][R][S][64], w_chwk[C][R][S][64]
 *   cs_conv2d_fwd / _dgrad / _wgrad with CsConvGeom.groups > 1 take these operands
 *     (same signatures; dw buffer of wgrad is [K][R][S][64] fp32)
 *   cs_wgrad_finalize_grouped: picks the group diagonal out of the slab gradient -> dw[K][Cg][R][S] (+BN-eval grads) */
int cs_weight_prep_grouped(const float* w, const float* scale, int dtype, int K, int Cg, int R, int S, void* w_khwc,
                           void* w_chwk, void* stream);
int cs_wgrad_finalize_grouped(const float* dw_slab, int nsplit, const float* w, const float* scale, const float* rstd,
                              const float* mean, const float* gsum, int K, int Cg, int R, int S, float* dw, float* dgamma,
                              float* dbeta, float* dot_ws, void* stream);
/* per-channel column sums: out[c] (+)= sum_m g[m][c]; fp32 out, zeroed by the caller. */
int cs_colsum(const void* g, int dtype, long long M, int C, float* out, void* stream);
/* The same sums left as per-workgroup partial rows (layout and consumers as for cs_conv2d_dgrad's deferred column sums):
 * partial[b * 2*C + c], b < cs_colsum_partial_rows(M); no atomics, no zero-fill. */
int cs_colsum_partial_rows(long long M);
/* ---- optimizer ---------------------------------------------------------------------------------
 * torch.optim.Adam's update (the optimizer the reference's drivers construct: train_tile.py:282, train_image.py:476,
 * train_seg.py:309; L2 weight decay, no amsgrad) over up to cs_adam_max_tensors() fp32 tensors in one launch.
 *   tensors_dev : DEVICE array of {p, m, v, n} for ALL tensors of the optimizer (built once)
 *   grads_host  : HOST array of n_tensors device pointers, the gradients of tensors [t0, t0 + n_tensors) (passed by value)
 *   chunks_dev  : DEVICE array of n_chunks (tensor index, first element / cs_adam_chunk_elems()) int pairs covering those tensors
 *   step        : t >= 1 of this update; hyper-parameters as doubles (the bias corrections and 1 - beta are formed in double, as torch does) */
typedef struct CsAdamTensor { float* p; float* m; float* v; long long n; } CsAdamTensor;
int cs_adam_chunk_elems(void);
int cs_adam_max_tensors(void);
int cs_adam_step(const CsAdamTensor* tensors_dev, const void* const* grads_host, int t0, int n_tensors, const int* chunks_dev,
                 int n_chunks, double lr, double beta1, double beta2, double eps, double weight_decay, double step, void* stream);
/* The same update with the step counts in DEVICE memory (torch.optim.Adam(capturable=True)'s state layout: one fp32 scalar per
 * tensor), so that a training step captured into a HIP graph (cellsegmentation_amd.graphed.GraphedStep over the loop body
 * train/train.py:29-42) advances them at every replay.  Two launches: one thread per tensor does t = ++*steps_dev[row] and leaves
 * (lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t)) -- formed in double -- in coef_dev[row]; the update kernel reads them from there.
 *   steps_dev : DEVICE array, one `float*` per ROW of tensors_dev (rows [t0, t0 + n_tensors) are used)
 *   coef_dev  : DEVICE scratch of 2 floats per row of tensors_dev (caller-owned, overwritten)
 *   lr_dev    : DEVICE double holding the learning rate, or NULL to use `lr` (a captured launch keeps its by-value arguments
 *               for ever: schedulers that change lr between replays write it here) */
int cs_adam_step_dev(const CsAdamTensor* tensors_dev, const void* const* grads_host, int t0, int n_tensors, const int* chunks_dev,
                     int n_chunks, float* const* steps_dev, float* coef_dev, const double* lr_dev, double lr, double beta1, double beta2,
                     double eps, double weight_decay, void* stream);

/* Per-sample, per-channel sums of an NHWC tensor [N][HW][C]: out[n][c] = scale * sum_p a[n][p][c] (* b[n][p][c] when b != NULL),
 * fp32 [N][C], overwritten.  The two reductions of a squeeze-excitation block (torchvision SqueezeExcitation as used by
 * model/efficientnet.py:83,107): the average pool (b = NULL, scale = 1/HW) and ds = sum dy * x of its backward.  Row-strided
 * workgroups leave partial rows in `workspace` (cs_sample_sum_workspace bytes, caller-owned) that are added in a fixed order:
 * bitwise reproducible, no atomics. */
size_t cs_sample_sum_workspace(int N, int HW, int C);
int cs_sample_sum(const void* a, const void* b, int dtype, float scale, float* out, float* workspace, int N, int HW, int C,
                  void* stream);
/* bit plane of a bf16 NHWC tensor x[n_pixels][C] in the layout of cs_conv2d_fwd_bits (channel-block-major, below): the `mask_bits`
 * operand of cs_conv2d_dgrad_packed for a post-ReLU tensor that no convolution epilogue produced, e.g. torch.cat of two ReLU outputs,
 * resnet.py:284-294; C % 32 == 0 */
int cs_positive_bits(const void* x, int dtype, long long n_pixels, int C, uint8_t* bits, void* stream);
int cs_colsum_partial(const void* g, int dtype, long long M, int C, float* partial, void* stream);

/* ---- pooling --------------------------------------------------------------------------------
 * MaxPool2d(3, stride 2, pad 1) (resnet.py:114). */
/* argmax (nullable, uint8 [N][P][Q][C]): window tap kh*3+kw of the first maximum in scan order
 * (ATen's max_pool2d_with_indices tie rule: strictly-greater replaces). */
int cs_maxpool3x3s2_fwd(const void* x, int dtype, void* y, uint8_t* argmax, int N, int H, int W, int C, int P, int Q,
                        void* stream);
/* dx[n][iy][ix][c] = sum over the <=4 windows containing (iy,ix) whose argmax is this pixel of
 * dy * [y_mask > 0] (y_mask nullable: the pooled output when a ReLU precedes the pool and the
 * caller wants the ReLU backward of resnet.py:113 fused). Gather form: no atomics. */
int cs_maxpool3x3s2_bwd(const void* dy, const uint8_t* argmax, const void* y_mask, int dtype, void* dx, int N, int H,
                        int W, int C, int P, int Q, void* stream);
/* AdaptiveAvgPool2d(1)+AdaptiveMaxPool2d(1) summed (resnet.py:266,274): feat[N][C] fp32,
 * argmax[N][C] int32 = first spatial index of the maximum.  with_max==0: average only (the squeeze of
 * torchvision's SqueezeExcitation, efficientnet.py:107). */
int cs_gap_avgmax_fwd(const void* x, int dtype, float* feat, int32_t* argmax, int N, int HW, int C, int with_max,
                      void* stream);
/* dx[n][p][c] = ( dfeat[n][c]/HW + (p==argmax[n][c]) * dfeat[n][c] ) * [x>0 if relu_mask]. */
int cs_gap_avgmax_bwd(const float* dfeat, const int32_t* argmax, const void* x, int dtype, void* dx, int N, int HW,
                      int C, int relu_mask, int with_max, void* stream);

/* ---- segmentation decoder data movement ------------------------------------------------------
 * F.interpolate(mode="bilinear", align_corners=True) (resnet.py:282,287,292,297,300): x[N][H][W][C] -> y[N][P][Q][C] */
int cs_bilinear_ac_fwd(const void* x, int dtype, void* y, int N, int H, int W, int C, int P, int Q, void* stream);
/* dx = interpolate^T(dy) * [mask>0] (mask nullable, shaped like dx: the ReLU output that was upsampled) */
int cs_bilinear_ac_bwd(const void* dy, const void* mask, int dtype, void* dx, int N, int H, int W, int C, int P, int Q,
                       void* stream);
/* torch.cat([a,b], dim=1) in NHWC (resnet.py:284,289,294): out[M][Ca+Cb]; and its inverse (a or b nullable). */
int cs_concat_channels(const void* a, const void* b, int dtype, void* out, long long M, int Ca, int Cb, void* stream);
int cs_split_channels(const void* whole, int dtype, void* a, void* b, long long M, int Ca, int Cb, void* stream);

/* ---- MBConv pieces (model/efficientnet.py:81-122; torchvision 0.11.2 ConvNormActivation with groups=C,
 * SqueezeExcitation, StochasticDepth("row")) -------------------------------------------------------
 * depthwise conv: geometry with K == C, R == S; w_hwc fp32 [R][S][C]; optional fused per-channel
 * scale/shift (eval BN) and activation. */
int cs_dwconv_fwd(const CsConvGeom* g, int dtype, const void* x, const float* w_hwc, const float* scale, const float* shift,
                  int act, void* y, void* stream);
/* Depthwise forward for train-mode BN: y = raw conv output, and the per-channel sum / sum of squares of the stored y as
 * *partial_rows per-workgroup rows partial[r][2][C] (fp64, cs_dwconv_fwd_stats_workspace(g) bytes); cs_bn_partial_fold adds the
 * rows into the exact accumulator `stats` (cs_bn_accum_words(C) zeroed words) -- together they replace cs_dwconv_fwd + cs_bn_stats
 * (one pass over y less). */
size_t cs_dwconv_fwd_stats_workspace(const CsConvGeom* g);
int cs_dwconv_fwd_stats(const CsConvGeom* g, int dtype, const void* x, const float* w_hwc, void* y, double* partial,
                        int* partial_rows, void* stream);
int cs_bn_partial_fold(const double* partial, int rows, int C, double* stats, void* stream);
int cs_dwconv_dgrad(const CsConvGeom* g, int dtype, const void* dy, const float* w_hwc, void* dx, void* stream);
/* dw_hwc[R][S][C] fp32 += ... (zeroed by the caller) */
/* workspace: cs_dwconv_wgrad_workspace(g) bytes of per-workgroup partial rows (folded by a second kernel: no atomics, dw_hwc is
 * overwritten, not accumulated into) */
size_t cs_dwconv_wgrad_workspace(const CsConvGeom* g);
int cs_dwconv_wgrad(const CsConvGeom* g, int dtype, const void* x, const void* dy, float* dw_hwc, float* workspace, void* stream);
/* the same with the result in the parameter's own layout [C][1][R][S] (nn.Conv2d(groups = C).weight, model/efficientnet.py:97-103) */
int cs_dwconv_wgrad_oihw(const CsConvGeom* g, int dtype, const void* x, const void* dy, float* dw_oihw, float* workspace, void* stream);
/* Depthwise filters of MANY layers, parameter layout [C][1][R][S] -> the [R][S][C] the depthwise kernels read, in one launch.
 * desc_dev: n rows in DEVICE memory, `first` (offset of the layer's first element in the launch's flat index space) ascending;
 * total = sum of C * RS. */
typedef struct { const float* src; float* dst; long long C; long long RS; long long first; } CsDwStageDesc;
int cs_dw_weights_hwc_multi(const CsDwStageDesc* desc_dev, int n, long long total, void* stream);
/* y[n][p][c] = x[n][p][c]*s[n][c] */
int cs_se_scale(const void* x, int dtype, const float* s, void* y, int N, int HW, int C, void* stream);
/* phase 0: ds[n][c] = sum_p dy*x ; phase 1: dx = dy*s + davg[n][c]/HW (davg nullable) */
int cs_se_scale_bwd(const void* dy, const void* x, int dtype, const float* s, const float* davg, float* ds, void* dx, int N,
                    int HW, int C, int phase, void* stream);
/* y = a*row_scale[n] + b over N rows of per_row elements (row_scale, b nullable) */
int cs_rowscale_add(const void* a, int dtype, const float* row_scale, const void* b, void* y, int N, long long per_row,
                    void* stream);

/* ---- tile construction on the device (the step before the hot path; dataset/dataset.py:203-214,718-742,78-83) ----
 * images: uint8 [n_images][H][W][3] in HBM; tile t = image tile_img[t], upper-left (row, col) = tile_rc[2t], tile_rc[2t+1];
 * out[n_tiles][size][size][8] dtype = ((u8/255) - mean)/std on channels 0..2, channels 3..7 zero (the NHWC operand of the
 * stem convolution).  mean/std are HOST arrays of 3 floats.  The caller guarantees tiles lie inside the image. */
int cs_tile_gather(const uint8_t* images, int n_images, int H, int W, const int32_t* tile_img, const int32_t* tile_rc,
                   long long n_tiles, int size, const float* host_mean3, const float* host_std3, int dtype, void* out,
                   void* stream);

/* ---- heads: Linear (resnet.py:126,137,140,150), losses (train/train.py:34,80-83) ------------- */
/* y[M][N] = act( x[M][K] @ w[N][K]^T + b[N] )  (fp32; b nullable; preact nullable: the value before act) */
int cs_linear_fwd(const float* x, const float* w, const float* b, float* y, float* preact, int M, int N, int K, int act,
                  void* stream);
/* dx[M][K] = g @ w (nullable dx); dw[N][K] (+)= g^T @ x; db[N] (+)= colsum(g), g = dy through the output
 * activation: `y` = stored output for ReLU / sigmoid, stored PRE-activation for SiLU. */
/* `workspace`: cs_linear_bwd_workspace(M, N, K) bytes (0 up to 512 rows: NULL is fine).  With a long batch axis -- the reference hands its
 * tile loops batches of 40 960 (train_tile.py -b), so the squeeze-excitation layers of an EfficientNet encoder see M in the thousands --
 * the weight gradient is cut into row slices whose partial results land there and are added in slice order (ABI 6). */
size_t cs_linear_bwd_workspace(int M, int N, int K);
int cs_linear_bwd(const float* x, const float* w, const float* dy, const float* y, int act, float* dx, float* dw,
                  float* db, int M, int N, int K, int accumulate, float* workspace, void* stream);
/* CrossEntropyLoss(mean) * gamma on logits[M][C]; dlogits nullable.  `loss` (here and in cs_mse): cs_loss_words() = 16 floats, all
 * overwritten -- the value in loss[0], the rest is the exact accumulator the workgroups' partial sums meet in (round 5: the value of a
 * launch with more than one workgroup no longer depends on their arrival order). */
int cs_loss_words(void);
int cs_softmax_ce(const float* logits, const int64_t* labels, float gamma, float* loss, float* dlogits, int M, int C,
                  void* stream);
/* softmax(logits,1)[:,1] (inference.py:24-27) */
int cs_softmax_prob1(const float* logits, float* p1, int M, int C, void* stream);
/* np.argmax(F.softmax(logits,1), axis=1) (inference.py:72-76, 118-119): arg max over the fp32 probabilities, first index on
 * ties (logits closer than the rounding of exp/÷ tie as probabilities); idx int64 [M]. */
int cs_softmax_argmax(const float* logits, int64_t* idx, int M, int C, void* stream);
/* sum or mean of w_i*(x_i-t_i)^2, w_i = 1 (weighted==0) or the reference's weighted_mse weights
 * (metrics/metrics.py:23-33: ln(t) if t>=20 else t).  dx nullable. scale = upstream grad factor. */
int cs_mse(const float* x, const float* t, int weighted, int mean, float* loss, float* dx, int M, void* stream);

/* Dice loss (train/losses.py:44-62 over metrics/metrics.py:36-53) on p[N][HW], t[N][HW] fp32:
 * sums = (sum p*t, sum p^2, sum t^2) per sample as an exact accumulator (see cs_bn_stats) of cs_bn_accum_words(3 N) 8-byte words,
 * cleared by cs_dice_fwd (the block sums are added in any order with the same bits: the reference's CPU path repeats too);
 * loss = mean|sum over n of 1-(2a+eps)/(b+c+eps). */
int cs_dice_fwd(const float* p, const float* t, int N, long long HW, float eps, int mean, double* sums, float* loss,
                void* stream);
int cs_dice_bwd(const float* p, const float* t, const double* sums, int N, long long HW, float eps, int mean, float* dp,
                void* stream);
/* F.softmax(logits, dim=1)[:, ch] on NCHW fp32 logits[N][C][HW] (train/train.py:189) and its backward. */
int cs_softmax_channel_fwd(const float* logits, float* pc, int N, int C, long long HW, int ch, void* stream);
int cs_softmax_channel_bwd(const float* logits, const float* dpc, float* dlogits, int N, int C, long long HW, int ch,
                           void* stream);

/* ---- adaptive top-k instance selection (inference.py:31-43) ---------------------------------
 * probs[T] fp32, groups[T] int32 non-decreasing, k_per_tile[T] int32 (k of the tile's group).
 * seg_offsets[n_groups+1] int64 (device): start of each group's run; max_run = longest run (host int).
 * Writes the reference's order[index] list: out_idx[0..*out_count) int64 (device), reproducing
 * np.lexsort((probs, groups)) + the wrap-around (i+k)%T comparison bit-exactly.  Runs of any length: up to 8192 tiles are
 * sorted in LDS, longer ones (whole-slide tile grids) in place in global memory -- the reference has no limit either.
 * workspace: >= cs_segmented_topk_workspace(T) bytes. */
size_t cs_segmented_topk_workspace(long long T);
int cs_segmented_topk(const float* probs, const int32_t* groups, const int32_t* k_per_tile,
                      const int64_t* seg_offsets, int n_groups, int max_run, long long T, int64_t* out_idx,
                      int64_t* out_count, void* workspace, size_t workspace_bytes, void* stream);

/* ---- packed-operand bf16 convolutions (csrc/conv_v2.hip): the stride-1 3x3 forward / data gradient of BasicBlock / Bottleneck
 * (model/resnet.py:20,23,53 and their autograd backward) with the source rows of ALL taps staged once per 64-channel chunk (halo
 * tile in LDS), weights streamed to registers in MFMA-fragment order and a register-direct epilogue.  CS_BF16 only.
 *   cs_conv2d_packed_supported : 1 when the geometry is served (3x3, stride 1, channel counts multiples of 64, operand < 2 GiB, ...);
 *                                otherwise use cs_conv2d_fwd / cs_conv2d_dgrad with the plain staged weights.  dgrad: 0 = forward, 1 = data gradient.
 *   cs_pack_conv_weights       : staged weights (w_khwc for forward, w_chwk for the data gradient, as cs_weight_prep /
 *                                cs_stage_conv_bn* write them) -> fragment order [32-row tile][64-ch chunk][tap][16-deep step][lane][8],
 *                                cs_conv2d_packed_weight_bytes() bytes; the data-gradient form mirrors the taps.
 *   cs_conv2d_fwd_packed       : y = act(conv(x, w) + shift[k] + residual); positive_bits (nullable) as cs_conv2d_fwd_bits.
 *   cs_conv2d_dgrad_packed     : dx = (conv_transpose(dy, w) + add) masked by mask_bits (nullable, as cs_conv2d_dgrad_bits);
 *                                partial_rows (nullable): fp32 [cs_conv2d_packed_partial_rows(g, 1)][2][C] per-workgroup column
 *                                sums of the stored dx (first C entries of a row), folded by cs_fold_partial_rows*.
 *                                A STRIDE-2 1x1 geometry (the shortcut of a down-sampling block, model/resnet.py:183) is served in
 *                                COMPACT form: dx is [N][P][Q][C], the values of destination pixels (2y, 2x); every other pixel of
 *                                that gradient is zero and is never written (add, mask_bits, partial_rows must be NULL).  Its
 *                                consumer -- the 1x1 data gradient that accumulates the block input's gradient -- takes it as
 *                                `add` with add_stride = 2: added at even (y, x) only, nothing elsewhere (add_stride = 1: an
 *                                ordinary destination-shaped operand). */
int cs_conv2d_packed_supported(const CsConvGeom* g, int dgrad);
size_t cs_conv2d_packed_weight_bytes(const CsConvGeom* g, int dgrad);
int cs_pack_conv_weights(const CsConvGeom* g, int dgrad, const void* w_staged, void* w_packed, void* stream);
int cs_conv2d_packed_partial_rows(const CsConvGeom* g, int dgrad);
int cs_conv2d_fwd_packed(const CsConvGeom* g, const void* x, const void* w_packed, const float* shift, const void* residual, int act,
                         void* y, uint8_t* positive_bits, void* stream);
int cs_conv2d_dgrad_packed(const CsConvGeom* g, const void* dy, const void* w_packed, const void* add, int add_stride,
                           const uint8_t* mask_bits, void* dx, float* partial_rows, void* stream);

/* ---- stem on a pixel-paired image (Conv2d(3, 64, 7, stride 2, padding 3), model/resnet.py:111) ---------------------------
 * With 3 channels padded to one 16-byte chunk per pixel the implicit GEMM walks 49 chunks of which 147/392 elements are real.
 * Two neighbouring pixels x 4 channels per chunk make it a 7x4-tap convolution over x_pair[N][H][ceil(W/2)][8] (row stride 2,
 * pair stride 1, pad (3, 2)): 28 chunks, 1.75x less MFMA work and DMA traffic, same kernels.
 *   cs_stem_pair_input  : NHWC8 image (channels 3..7 zero) -> x_pair
 *   cs_stem_pair_weights: staged w_khwc[K][7][7][8] (BN scale already folded) -> w_pair[K][7][4][8]
 *   cs_stem_fwd         : as cs_conv2d_fwd (scale/shift/act/stats/workspace), y[N][P][Q][K], P = (H - 1) / 2 + 1
 *   cs_stem_wgrad       : raw split-K slabs dw_pair[cs_stem_wgrad_splits][K][7][4][8] from x_pair and dy
 *   cs_stem_unpair_slabs: those slabs summed into ONE ordinary raw slab dw_khwc[K][7][7][8] (input of cs_wgrad_finalize*) */
int cs_stem_pair_input(const void* x_nhwc8, int dtype, int N, int H, int W, void* x_pair, void* stream);
/* the same x_pair straight from the fp32 NCHW image [N][3][H][W] (cs_nchw_to_nhwc + cs_stem_pair_input in one pass; bf16 only) */
int cs_stem_pair_from_nchw(const float* x_nchw, int dtype, int N, int H, int W, void* x_pair, void* stream);
int cs_stem_pair_weights(const void* w_khwc, int dtype, int K, void* w_pair, void* stream);
int cs_stem_fwd(int N, int H, int W, int K, int dtype, const void* x_pair, const void* w_pair, const float* scale, const float* shift,
                int act, void* y, double* stats, void* workspace, void* stream);
/* bf16 forward on the ring kernel of conv_v2.hip (gathered operand rows): w_packed = cs_pack_conv_weights of the 1x1 geometry
 * (C = 256, K) applied to w_pair padded to [K][8][4][8] with one zero filter row; epilogue as cs_conv2d_fwd_packed. */
int cs_stem_fwd_packed(int N, int H, int W, int K, const void* x_pair, const void* w_packed, const float* shift, int act, void* y,
                       uint8_t* positive_bits, void* stream);
int cs_stem_wgrad_splits(int N, int H, int W, int K);
int cs_stem_wgrad(int N, int H, int W, int K, int dtype, const void* x_pair, const void* dy, float* dw_pair_slabs, int use_tr_read,
                  void* stream);
int cs_stem_unpair_slabs(const float* dw_pair_slabs, int nsplit, int K, float* dw_khwc, void* stream);

/* ---- the steps either side of the top-k -----------------------------------------------------------
 * cs_segmented_order: order = np.lexsort((probs, groups)) alone (train_seg.py:239, evaluate.py:13); order[T] int64.
 * cs_threshold_select: order[[p > threshold for p in probs[order]]] (train_seg.py:243-245), i.e. `rank`'s tile list;
 *   workspace as for cs_segmented_topk.
 * cs_evaluate_tile_counts: evaluate.py:8-27 as four exact counts, counts4 (zeroed by the caller) += {pred != real,
 *   pred & !real, !pred & real, real}; pos_from[g] (int64, host-computed, device-resident) = first sorted position labelled
 *   positive for group g or any later group.
 * cs_paint_tile_masks: utils/image_processing.py:92-98 -- masks[n_images][H][W] uint8 (zeroed by the caller) gets a
 *   tile_size^2 block of ones at tile_xy[t] = (row, col) in image groups[t] for every t in selected[0..n_selected).
 * cs_prune_excess: dataset/dataset.py:190-199 on the shuffled label array: positions kept when the first n_excess entries
 *   with labels[i] == flag are deleted (kept[] int64 ascending, *kept_count). */
int cs_segmented_order(const float* probs, const int64_t* seg_offsets, int n_groups, int max_run, long long T, int64_t* order,
                       void* stream);
int cs_threshold_select(const float* probs, const int64_t* order, long long T, float threshold, int64_t* out_idx,
                        int64_t* out_count, void* workspace, size_t workspace_bytes, void* stream);
int cs_evaluate_tile_counts(const float* probs, const int64_t* order, const int32_t* groups, const int64_t* pos_from,
                            float threshold, long long T, unsigned long long* counts4, void* stream);
int cs_paint_tile_masks(const int64_t* selected, long long n_selected, const int32_t* groups, const int32_t* tile_xy,
                        int tile_size, int H, int W, uint8_t* masks, void* stream);
int cs_prune_excess(const int32_t* labels, long long n, int flag, long long n_excess, int64_t* kept, int64_t* kept_count,
                    void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CELLSEG_HIP_H_ */
