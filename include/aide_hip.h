/* C ABI of libaide_hip.so — the MI355X (gfx950) drop-in for AIDE's FuseUNet/UNet training hot path.
 *
 * The reference (lich0031/AIDE) has no native code and no FFI: the replaceable seam is its Python
 * nn.Module / loss-module API.  Each entry point below is what a Python binding
 * for that seam calls; the "replaces" note cites the reference call site (path:line under the
 * reference tree).  Conventions:
 *   - plain pointers + sizes only; no torch types.  All pointers are DEVICE pointers unless noted.
 *   - tensors are NCHW fp32 planes; `*_bs` is the batch stride in ELEMENTS, so a tensor may be a
 *     channel slice of a larger concatenation buffer (this is how torch.cat is eliminated).
 *   - every call is asynchronous on `stream`, never allocates, never synchronises.
 *   - return value: 0 = ok, <0 = bad argument, >0 = hipError_t of the launch.
 */
#ifndef AIDE_HIP_H
#define AIDE_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* aide_stream_t; /* == hipStream_t */

/* ---- 3x3 convolution, pad 1 (MFMA implicit GEMM) -----------------------------------------------
 * replaces nn.Conv2d(ci, co, 3, padding=1): models_twomodalinputs/netblocks.py:17,24,26 and
 * models_singlemodalinput/UNet.py:12,19,21 (forward) and autograd's convolution_backward. */
int aide_conv3x3_chunk(int Cin);                                  /* channel padding granule */
int aide_conv3x3_pack_weights(const float* w /*[Co][Ci][3][3]*/, float* wf /*[ci_pad][9][Co]*/,
                              float* wd /*[co_pad][9][Ci] or NULL*/, int Co, int Ci, int ci_pad,
                              int co_pad, aide_stream_t stream);
/* all filters of a network in one launch; descs = DEVICE array of n 48-byte records
 * {const float* w; float* wf; float* wd; int32 Co, Ci, ci_pad, co_pad; int64 block_start} */
int aide_conv3x3_pack_weights_multi(const void* descs, int n, int64_t total_blocks, aide_stream_t stream);
int aide_conv3x3_plan(int N, int Cin, int H, int W, int Cout);    /* variant | splitk<<8 */
size_t aide_conv3x3_ws_bytes(int N, int H, int W, int Cout, int splitk);
int aide_conv3x3_igemm(const float* x, int64_t x_bs, const float* wp, int ldw, const float* bias,
                       float* y, int64_t y_bs, int N, int Cin, int H, int W, int Cout, int accumulate,
                       int plan, float* ws, const float* epi_scale, int epi_relu,
                       aide_stream_t stream);   /* forward and dgrad; any Cin, Cout, H, W (partial channel tiles / chunks are
                                                   masked).  epi_scale != NULL (non-split, accumulate = 0, bias given): the
                                                   epilogue writes y = relu?(acc * epi_scale[co] + bias[co]) -- eval-mode
                                                   BatchNorm folded into the convolution (aide_bn_eval_fold) */
/* Winograd F(2x2,3x3) variant of the same convolution (forward / dgrad) for even H, W % 4 == 0,
 * Cout % 64 == 0, Cin % 8 == 0.  Filters are pre-transformed (G g G^T), channel-blocked by 8:
 * uf [ci_pad/8][16][Co][8 ci], ud [co_pad/8][16][Ci][8 co] (the tensors are allocated as [pad][16][C]). */
int aide_conv3x3_wino_supported(int Cin, int H, int W, int Cout);
int aide_conv3x3_wino_splitk(int N, int Cin, int H, int W, int Cout);
/* Winograd F(4x4,3x3) variant for the large layers (36 multiplies per 16 outputs): H % 4 == 0, W % 4 == 0,
 * H >= 16, W >= 16, Cout % 32 == 0 (a trailing half block of 32 is computed and dropped), Cin % 8 == 0; W == 16 needs an even N
 * (two images per workgroup tile).  Workgroup tile = 32 slots of 4x4 outputs: 16 x 32 pixels, or a 20 x 20 canvas (25 slots used)
 * where that covers the plane with fewer tiles (20 x 20, 40 x 40, ... planes).  Filters: uf [Ci/4][36][Co][4 ci], ud [Co/4][36][Ci][4 co]
 * (allocated as [C][36][C']); pack descriptors as above with {w, uf|0, ud|0, Co, Ci, 0, 0, block_start}.
 * splitk must divide Cin / 8. */
int aide_conv3x3_wino4_supported(int Cin, int H, int W, int Cout);
int aide_conv3x3_wino4_splitk(int N, int Cin, int H, int W, int Cout);
int aide_conv3x3_wino4_pack_blocks(int Co, int Ci);
int aide_conv3x3_wino4_pack_multi(const void* descs, int n, int64_t total_blocks, aide_stream_t stream);
/* stats_parts (NULL, or parts[Cout][aide_conv3x3_wino4_stats_parts][2]; only a non-split launch with accumulate = 0 and
 * W >= 32, AIDE_ERR_ARG otherwise): the epilogue also writes, per output channel and workgroup tile, the fp32 sum and sum
 * of squares of its pre-bias outputs -- the BatchNorm statistics for aide_bn_train_fwd_parts (netblocks.py:25,27) without a
 * pass over z.  epi_scale (NULL, or [Cout]; accumulate = 0 and bias given): y = relu?(acc * epi_scale[co] + bias[co]), the
 * folded eval-mode BatchNorm of aide_bn_eval_fold (a split launch applies it in its slab reduce).
 * in_bn_tab (NULL, or [N / in_bn_group_images][Cin][2] = (scale, shift) per image group and INPUT channel; Cin <= 1024, not
 * for W == 16): x is the RAW output z of the layer before and the loader stages relu(x * scale + shift) -- that layer's
 * BatchNorm + ReLU applied on the way in (SURVEY 8b in_prologue{bn_relu}; netblocks.py:25-28), its normalised output never
 * written: the forward-only augmentation passes of the co-teaching loop (trainchaos_proposed_30cases1labeled.py:263-281)
 * keep nothing for a backward pass.  Channels that are already activations get (1, 0).  Table: aide_bn_finalize_groups.
 * All of these are plain arguments of THIS launch: the library keeps no state between calls. */
int aide_conv3x3_wino4(const float* x, int64_t x_bs, const float* u, const float* bias, float* y, int64_t y_bs,
                       int N, int Cin, int H, int W, int Cout, int accumulate, int splitk, float* ws,
                       float* stats_parts, const float* epi_scale, int epi_relu, const float* in_bn_tab,
                       int in_bn_group_images, aide_stream_t stream);
/* descs as above with {w, uf, ud}; an entry occupies aide_conv3x3_wino_pack_blocks(Co, Ci) workgroups */
int aide_conv3x3_wino_pack_blocks(int Co, int Ci);
int aide_conv3x3_wino_pack_multi(const void* descs, int n, int64_t total_blocks, aide_stream_t stream);
int aide_conv3x3_wino(const float* x, int64_t x_bs, const float* u, const float* bias, float* y,
                      int64_t y_bs, int N, int Cin, int H, int W, int Cout, int accumulate, int splitk,
                      float* ws, aide_stream_t stream);
int aide_conv3x3_wgrad_splits(int N, int Co, int Ci, int H, int W);
size_t aide_conv3x3_wgrad_ws_bytes(int N, int Co, int Ci, int H, int W);
/* every weight-gradient entry point: `queue` = NULL (the slab reduce into dw is launched behind the kernel) or a
 * caller-owned aide_wgrad_queue_* handle (the reduce is queued for the caller's batched launch; see below) */
int aide_conv3x3_wgrad(const float* dz, int64_t dz_bs, const float* a, int64_t a_bs,
                       float* dw /*[Co][Ci][3][3]*/, int N, int Co, int Ci, int H, int W, float* ws,
                       void* queue, aide_stream_t stream);

/* Winograd form of the weight gradient (16 instead of 36 MFMA-multiplies per tile, co, ci);
 * even H, W % 4 == 0, Co >= 64, Ci >= 64.  Same outputs as aide_conv3x3_wgrad. */
int aide_conv3x3_wgrad_wino_supported(int Co, int Ci, int H, int W);
int aide_conv3x3_wgrad_wino_splits(int N, int Co, int Ci, int H, int W);
size_t aide_conv3x3_wgrad_wino_ws_bytes(int N, int Co, int Ci, int H, int W);
int aide_conv3x3_wgrad_wino(const float* dz, int64_t dz_bs, const float* a, int64_t a_bs, float* dw, int N,
                            int Co, int Ci, int H, int W, float* ws, void* queue, aide_stream_t stream);
/* stem layers (Ci <= 3; H % 4 == 0, W % 64 == 0): the nine taps folded into the GEMM's N dimension, bound by reading dz once.
 * dz fp32 or bf16-stored (dz_bf16), x fp32; round_bf16: operands rounded to bf16 when staged (the bf16 mode's contract);
 * ws / splits as for aide_conv3x3_wgrad / aide_conv3x3_wgrad_bf16 (which dispatch here themselves).
 * Replaces autograd's weight gradient of nn.Conv2d(3, C, 3, padding=1) (netblocks.py:24 in modal*_downblock1, UNet.py:19). */
int aide_conv3x3_wgrad_stem_supported(int Co, int Ci, int H, int W);
int aide_conv3x3_wgrad_stem_splits(int N, int H, int W);
int aide_conv3x3_wgrad_stem(const void* dz, int dz_bf16, int64_t dz_bs, const float* x, int64_t x_bs, float* dw,
                            int N, int Co, int Ci, int H, int W, float* ws, int splits, int round_bf16,
                            void* queue, aide_stream_t stream);
/* transposed F(4x4,3x3) for the large layers: H % 4 == 0, W % 4 == 0, H >= 8, W >= 16, Co % 32 == 0 (a trailing half tile of 32
 * is computed and dropped), Ci % 32 == 0 */
int aide_conv3x3_wgrad_wino4_supported(int Co, int Ci, int H, int W);
int aide_conv3x3_wgrad_wino4_splits(int N, int Co, int Ci, int H, int W);
size_t aide_conv3x3_wgrad_wino4_ws_bytes(int N, int Co, int Ci, int H, int W);
/* dw [Co][Ci][3][3] = sum over images and pixels; the workgroup count of the launch is an argument (target_wgs <= 0: the
 * default, half of the chip -- the kernel normally shares it with the dependent chain of the backward pass; 256 for a launch
 * that has the chip alone); ws: aide_conv3x3_wgrad_wino4_ws_bytes_t() bytes; queue: NULL or a batched-reduce queue */
int aide_conv3x3_wgrad_wino4_splits_t(int N, int Co, int Ci, int H, int W, int target_wgs);
size_t aide_conv3x3_wgrad_wino4_ws_bytes_t(int N, int Co, int Ci, int H, int W, int target_wgs);
int aide_conv3x3_wgrad_wino4_t(const float* dz, int64_t dz_bs, const float* a, int64_t a_bs, float* dw, int N,
                               int Co, int Ci, int H, int W, float* ws, int target_wgs, void* queue,
                               aide_stream_t stream);

/* ---- bf16-MFMA mode of the same convolution (BASELINE config 5: "FuseUNet bf16 MFMA path") -------------------
 * replaces the same nn.Conv2d call sites (netblocks.py:17,24,26; UNet.py:12,19,21) when the engine runs with
 * precision='bf16': operands are rounded to bf16 (RNE) while they are staged into LDS, products accumulate in fp32
 * (v_mfma_f32_32x32x16_bf16); tensors in HBM, master weights, bias, BatchNorm statistics and the loss stay fp32.
 * Filters are pre-packed to bf16: uf [ceil(Ci/16)][9][2][Co][8], ud [ceil(Co/16)][9 reversed][2][Ci][8].
 * Supported: W % 32 == 0, Cout % 32 == 0 (forward / dgrad); Co % 32 == 0, W % 32 == 0, H % 4 == 0 (weight gradient;
 * a bf16-stored x operand additionally needs Ci % 8 == 0).  The engine keeps the fp32 kernels for every other layer. */
int aide_conv3x3_bf16_supported(int Cin, int H, int W, int Cout);
int aide_conv3x3_bf16_splitk(int N, int Cin, int H, int W, int Cout);
size_t aide_conv3x3_bf16_pack_elems(int Cout, int Cin);          /* bf16 elements of one direction's pack */
/* descs = DEVICE array of n 48-byte records {const float* w; uint16_t* uf (or 0); uint16_t* ud (or 0); int32 Co, Ci, 0, 0;
 * int64 block_start}; an entry occupies aide_conv3x3_bf16_pack_blocks(Co, Ci) workgroups (64 co x 16 ci blocks through LDS) */
int aide_conv3x3_bf16_pack_blocks(int Cout, int Cin);
int aide_conv3x3_bf16_pack_multi(const void* descs, int n, int64_t total_blocks, aide_stream_t stream);
/* forward and dgrad; x_bf16 / y_bf16 = 0: fp32 tensors; 1: bf16 STORAGE of the conv output z (forward: y_bf16) or of its
 * gradient dz (dgrad: x_bf16) -- the engine's precision='bf16' mode keeps z and dz in HBM as bf16 (they are only read by
 * BatchNorm / by these kernels) */
int aide_conv3x3_bf16_mixed(const void* x, int x_bf16, int64_t x_bs, const uint16_t* u, const float* bias, void* y,
                            int y_bf16, int64_t y_bs, int N, int Cin, int H, int W, int Cout, int accumulate, int splitk,
                            float* ws, aide_stream_t stream);
int aide_conv3x3_wgrad_bf16_supported(int Co, int Ci, int H, int W);
/* co_blocks: output-channel blocks of 32 per workgroup tile -- 1 (32 co x 64 ci, 2 waves, two workgroups per CU: the
 * 32-channel first-level layers), 2 (64 co x 64 ci, 4 waves), 4 (128 co x 64 ci, 8 waves; needs Co % 128 == 0, else 2 is
 * used) or 0 = the built-in rule (1 for Co <= 32, 4 from 150 GFLOP per launch); the split count depends on it */
int aide_conv3x3_wgrad_bf16_splits(int N, int Co, int Ci, int H, int W, int co_blocks);
size_t aide_conv3x3_wgrad_bf16_ws_bytes(int N, int Co, int Ci, int H, int W, int co_blocks);
int aide_conv3x3_wgrad_bf16_mixed(const void* dz, int dz_bf16, int64_t dz_bs, const void* a, int a_bf16, int64_t a_bs,
                                  float* dw, int N, int Co, int Ci, int H, int W, float* ws, int co_blocks, void* queue,
                                  aide_stream_t stream);

/* ---- ConvTranspose2d(k=2, s=2) (learned_bilinear=True up path) ---------------------------------
 * replaces nn.ConvTranspose2d: netblocks.py:12, UNet.py:7 */
int aide_convT2x2_fwd(const float* x, int64_t x_bs, const float* w /*[Ci][Co][2][2]*/, const float* b,
                      float* y, int64_t y_bs, int N, int Ci, int Co, int H, int W, aide_stream_t stream);
int aide_convT2x2_dgrad(const float* dy, int64_t dy_bs, const float* w, float* dx, int64_t dx_bs, int N,
                        int Ci, int Co, int H, int W, aide_stream_t stream);
size_t aide_convT2x2_wgrad_ws_bytes(int N, int Ci, int Co, int H, int W);
int aide_convT2x2_wgrad(const float* x, int64_t x_bs, const float* dy, int64_t dy_bs, float* dw, int N,
                        int Ci, int Co, int H, int W, float* ws, aide_stream_t stream);

/* ---- BatchNorm2d (+ReLU) -----------------------------------------------------------------------
 * replaces nn.BatchNorm2d + nn.ReLU: netblocks.py:25,27,28,18 ; UNet.py:20,22,23,13 */
/* Workspace of the training-mode calls below: ZERO-FILLED once by the caller (it holds the arrival counters of the one-pass
 * kernels, which every launch leaves at zero) and used by ONE stream at a time. */
size_t aide_bn_ws_bytes(int C);
/* 1: a channel of N * H * W values runs in the one-pass form -- S workgroups per channel hold the values in registers, exchange
 * fp64 partial sums through the workspace (summed in slot order: bit-reproducible) and normalise from the registers: forward
 * 8 B / element, backward 12, one launch each (units of 8 values when H * W and every batch stride are multiples of 8, whatever
 * the storage types: a bf16-stored call has the statistics of the fp32-stored call on the widened tensor, bit for bit).
 * 0: the two-pass fallbacks (planes that are no multiple of 4, channels of more than 2 M values). */
int aide_bn_one_pass(int N, int C, int H, int W);
/* Eval-mode BatchNorm folded into the convolution before it (the per-case inference loop,
 * trainchaos_comparison_1case.py:233-273: net.eval(), running statistics): aide_bn_eval_fold also writes
 * fbias = conv_bias * scale + shift; an aide_conv3x3_wino4 or aide_conv3x3_igemm launch given epi_scale = scale,
 * bias = fbias (accumulate = 0) writes y = relu?(acc * scale[co] + bias[co]) straight into the activation -- no separate
 * BatchNorm pass over the conv output (a split-K aide_conv3x3_wino4 launch applies it in its slab reduce;
 * aide_conv3x3_igemm only non-split).  A launch that cannot honour the epilogue returns AIDE_ERR_ARG. */
int aide_bn_eval_fold(int C, const float* gamma, const float* beta, const float* running_mean,
                      const float* running_var, float eps, const float* conv_bias, float* scale, float* shift,
                      float* fbias, aide_stream_t stream);
/* `done` (the three aide_bn_relu_bwd* calls): NULL, or an event of aide_event_create that is recorded when dz is complete --
 * attached to the call's last dispatch instead of a record packet of its own: aide_stream_wait_event(other, done) then
 * orders another stream behind dz at ~1.4 us of this queue's time (aide_stream_order: ~5 us; tools/ubench/handover_cost.hip) */
/* aide_bn_relu_bwd with dA still in the split-K slabs [splitk][N][C][H][W] (fp32, split_stride elements apart) of the
 * data-gradient convolution that produced it (aide_conv3x3_wino4 / aide_conv3x3_wino launched with accumulate = 2): the
 * kernel sums the slabs itself, in the order of the split reduce, so that launch and one pass over dA disappear.  One-pass
 * shapes only (aide_bn_one_pass(N, C, H, W) == 1); AIDE_ERR_ARG otherwise. */
int aide_bn_relu_bwd_slabs(const float* slabs, int splitk, int64_t split_stride, const float* z, int64_t z_bs, float* dz,
                           int64_t dz_bs, int N, int C, int H, int W, const float* mean, const float* rstd,
                           const float* scale, const float* shift, int relu, float* dgamma, float* dbeta, float* dbias,
                           void* ws, void* done, aide_stream_t stream);
/* Training-mode forward a = relu?(bn(z)) (batch statistics, running-statistics update; mean / rstd / scale / shift [C] are kept
 * by the caller for the backward), the plain apply a = relu?(z * scale + shift), and the backward dA -> dz, dgamma, dbeta (+ the
 * mathematically zero conv-bias gradient).  z_bf16 / a_bf16 / dz_bf16 give the element type behind the untyped pointers (0 = fp32;
 * 1 = bf16 STORAGE, precision='bf16'); the arithmetic (fp32 per element, fp64 reductions) is the same, widening is exact, dz is
 * narrowed RNE */
int aide_bn_train_fwd_mixed(const void* z, int z_bf16, int64_t z_bs, void* a, int a_bf16, int64_t a_bs, int N, int C,
                            int H, int W, const float* gamma, const float* beta, float eps, float momentum,
                            float* running_mean, float* running_var, long long* num_batches_tracked, float* mean,
                            float* rstd, float* scale, float* shift, int relu, void* ws, aide_stream_t stream);
/* BatchNorm(train)+ReLU fed directly by the split-K slabs of the convolution before it: a conv launched with
 * accumulate = 2 leaves its partial results in ws as [splitk][N][C][H][W] fp32 (no reduce launch); this entry sums them in
 * split order (+ bias), writes z for the backward pass and normalises -- one launch and one pass over z less per layer. */
int aide_bn_train_fwd_slabs(const float* slabs, int splitk, int64_t split_stride, const float* bias, void* z, int z_bf16,
                            int64_t z_bs, void* a, int a_bf16, int64_t a_bs, int N, int C, int H, int W,
                            const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                            float* running_var, long long* num_batches_tracked, float* mean, float* rstd, float* scale,
                            float* shift, int relu, void* ws, aide_stream_t stream);
/* BatchNorm statistics from the convolution's epilogue: a forward aide_conv3x3_wino4 launch given stats_parts writes, per
 * output channel and workgroup tile, the fp32 sum and sum of squares of its pre-bias outputs to parts[Cout][nparts][2]
 * (nparts = aide_conv3x3_wino4_stats_parts); aide_bn_train_fwd_parts then normalises with ONE pass over z (replaces the
 * statistics pass of nn.BatchNorm2d in train mode, netblocks.py:25,27).  aide_bn_two_pass: 1 for the planes where that pays
 * (more than 16384 values per channel, or fewer than 64 channels), 0 for the small deep-level planes. */
int aide_conv3x3_wino4_stats_parts(int N, int H, int W);
int aide_bn_two_pass(int N, int C, int H, int W);
/* the same for ONE group of a stacked batch (Engine.run_groups): `parts` points at the group's first entry of channel 0,
 * nparts counts the group's entries, parts_stride the entries per channel of the whole launch */
int aide_bn_train_fwd_parts_strided(const void* z, int z_bf16, int64_t z_bs, void* a, int a_bf16, int64_t a_bs, int N, int C,
                                    int H, int W, const float* parts, int nparts, int parts_stride, const float* conv_bias,
                                    const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                                    float* running_var, long long* num_batches_tracked, float* mean, float* rstd,
                                    float* scale, float* shift, int relu, aide_stream_t stream);
/* BatchNorm(train)+ReLU of a STACKED batch: `groups` independent batches of N images each, stacked along the batch dimension
 * (the four detached augmentation forwards of the co-teaching loop as one pass, trainchaos_proposed_30cases1labeled.py:265-269).
 * Every group is normalised with its own batch statistics; running statistics / num_batches_tracked are updated once per
 * group, in order -- what `groups` sequential train-mode forwards leave (bit for bit in the fused single-launch and the
 * conv-epilogue-parts forms; the two-pass form sums at most 96 / groups partials per channel instead of a launch's own split
 * count, so its statistics agree to rounding, <= 1e-6 relative) -- in ONE launch sequence; mean / rstd /
 * scale / shift receive the last group's values.  Input, one of: z as it is (slabs == parts == NULL); the split-K slabs
 * [splitk][N * groups][C][H][W] of the conv before it (as aide_bn_train_fwd_slabs; z is written); the conv epilogue's
 * statistics parts[C][parts_stride][2], group g's nparts entries at g * nparts (as aide_bn_train_fwd_parts). */
int aide_bn_train_fwd_groups(const void* z, int z_bf16, int64_t z_bs, void* a, int a_bf16, int64_t a_bs, int N, int groups,
                             int C, int H, int W, const float* slabs, int splitk, int64_t split_stride,
                             const float* slab_bias, const float* parts, int nparts, int parts_stride,
                             const float* conv_bias, const float* gamma, const float* beta, float eps, float momentum,
                             float* running_mean, float* running_var, long long* num_batches_tracked, float* mean,
                             float* rstd, float* scale, float* shift, int relu, void* ws, aide_stream_t stream);
/* BatchNorm(train) WITHOUT its pass over z: the conv epilogue's statistics of a stacked batch (parts / nparts / parts_stride /
 * conv_bias as above; N = images per group) become the per-group (scale, shift) entries [tab_c0, tab_c0 + C) of
 * tab[groups][tab_C][2] -- the table the consumer convolution's loader applies (in_bn_tab of aide_conv3x3_wino4) -- and the
 * running statistics / num_batches_tracked move once per group, in order. */
int aide_bn_finalize_groups(int N, int groups, int C, int H, int W, const float* parts, int nparts, int parts_stride,
                            const float* conv_bias, const float* gamma, const float* beta, float eps, float momentum,
                            float* running_mean, float* running_var, long long* num_batches_tracked, float* mean,
                            float* rstd, float* scale, float* shift, float* tab, int tab_C, int tab_c0, aide_stream_t stream);
int aide_bn_relu_apply_mixed(const void* z, int z_bf16, int64_t z_bs, void* a, int a_bf16, int64_t a_bs, int N, int C,
                             int H, int W, const float* scale, const float* shift, int relu, aide_stream_t stream);
/* Backward of relu?(bn(z)) of a layer whose activation ALSO fed an nn.MaxPool2d(2, 2) (fuseunet.py:13-31 / :51-78, UNet.py:114): dA is
 * the gradient from the activation's other readers (the decoder's skip path), pdy [N][C][H/2][W/2] (batch stride pdy_bs) the gradient of
 * the pooled tensor; the kernel routes pdy to the arg-max of every window (activation recomputed from z exactly as the forward pass
 * computed it; first maximum in row-major window order, as nn.MaxPool2d's backward) and adds it to dA on the way in -- bit-identical to
 * aide_maxpool2x2_bwd(accumulate = 1) followed by aide_bn_relu_bwd_mixed, without the pooling backward's pass.  fp32 storage, shapes of
 * aide_bn_relu_bwd_pool_supported (one-pass form with units of 8: H even, W % 8 == 0). */
int aide_bn_relu_bwd_pool_supported(int N, int C, int H, int W);
/* ... and the forward half: BatchNorm(train)+ReLU of such a layer writes the activation AND the pooled tensor [N * groups][C][H/2][W/2]
 * (at the layer's channel 0 of the pooled buffer) -- aide_maxpool2x2_fwd's pass does not run; `groups` as aide_bn_train_fwd_groups.  fp32,
 * z as it is, the shapes of aide_bn_relu_bwd_pool_supported(N, C, H, W) with N = images per group. */
int aide_bn_train_fwd_pool(const float* z, int64_t z_bs, float* a, int64_t a_bs, float* pooled, int64_t pooled_bs, int N, int groups,
                           int C, int H, int W, const float* gamma, const float* beta, float eps, float momentum,
                           float* running_mean, float* running_var, long long* num_batches_tracked, float* mean, float* rstd,
                           float* scale, float* shift, int relu, void* ws, aide_stream_t stream);
int aide_bn_relu_bwd_pool(const float* dA, int64_t d_bs, const float* pdy, int64_t pdy_bs, const float* z, int64_t z_bs, float* dz,
                          int64_t dz_bs, int N, int C, int H, int W, const float* mean, const float* rstd, const float* scale,
                          const float* shift, int relu, float* dgamma, float* dbeta, float* dbias, void* ws, void* done,
                          aide_stream_t stream);
/* Backward of relu?(bn(z)) of the layer whose activation feeds the 1x1 head (last_conv1: fuseunet.py:41 / :88, UNet.py:120): dA =
 * sum_k head_w[k][c] * dlogits[n][k][p] is formed inside the kernel (k ascending, fused multiply-adds: the sums of aide_head1x1_bwd's dx,
 * bit for bit) -- the head's data-gradient pass does not run.  fp32 storage, shapes of aide_bn_one_pass, 1 <= K <= 8. */
int aide_bn_relu_bwd_head(const float* dlogits, int64_t dl_bs, const float* head_w, int K, const float* z, int64_t z_bs, float* dz,
                          int64_t dz_bs, int N, int C, int H, int W, const float* mean, const float* rstd, const float* scale,
                          const float* shift, int relu, float* dgamma, float* dbeta, float* dbias, void* ws, void* done,
                          aide_stream_t stream);
int aide_bn_relu_bwd_mixed(const void* dA, int dA_bf16, int64_t d_bs, const void* z, int z_bf16, int64_t z_bs, void* dz,
                           int dz_bf16, int64_t dz_bs, int N, int C, int H, int W, const float* mean, const float* rstd,
                           const float* scale, const float* shift, int relu, float* dgamma, float* dbeta, float* dbias,
                           void* ws, void* done, aide_stream_t stream);

/* ---- MaxPool2d(2,2), bilinear x2 (align_corners=True) -------------------------------------------
 * replaces nn.MaxPool2d: fuseunet.py:13-31 (calls :51-78), UNet.py:114 ;
 *          nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True): netblocks.py:16 */
int aide_maxpool2x2_fwd(const float* x, int64_t x_bs, float* y, int64_t y_bs, int N, int C, int H, int W,
                        aide_stream_t stream);
int aide_maxpool2x2_bwd(const float* x, int64_t x_bs, const float* dy, int64_t dy_bs, float* dx,
                        int64_t dx_bs, int N, int C, int H, int W, int accumulate, aide_stream_t stream);
int aide_upsample2x_bilinear_fwd(const float* x, int64_t x_bs, float* y, int64_t y_bs, int N, int C, int H,
                                 int W, aide_stream_t stream);
/* the same on bf16-stored activations and activation gradients (precision='bf16'); arithmetic in fp32 */
int aide_maxpool2x2_fwd_mixed(const void* x, int x_bf16, int64_t x_bs, void* y, int y_bf16, int64_t y_bs, int N, int C,
                              int H, int W, aide_stream_t stream);
int aide_maxpool2x2_bwd_mixed(const void* x, int x_bf16, int64_t x_bs, const void* dy, int dy_bf16, int64_t dy_bs, void* dx,
                              int dx_bf16, int64_t dx_bs, int N, int C, int H, int W, int accumulate, aide_stream_t stream);
int aide_upsample2x_bilinear_bwd_mixed(const void* dy, int dy_bf16, int64_t dy_bs, void* dx, int dx_bf16, int64_t dx_bs,
                                       int N, int C, int H, int W, int accumulate, aide_stream_t stream);
int aide_upsample2x_bilinear_fwd_mixed(const void* x, int x_bf16, int64_t x_bs, void* y, int y_bf16, int64_t y_bs, int N,
                                       int C, int H, int W, aide_stream_t stream);
int aide_upsample2x_bilinear_bwd(const float* dy, int64_t dy_bs, float* dx, int64_t dx_bs, int N, int C,
                                 int H, int W, int accumulate, aide_stream_t stream);
/* inverse augmentation of logit planes (flip + PIL-exact bilinear rotation); replaces the D2H -> PIL ->
 * H2D round trip of reverseaug(): train_files/trainchaos_proposed_30cases1labeled.py:81-95.
 * par: DEVICE [N][8] doubles {a,b,c,d,e,f (PIL inverse affine), flip, mode (0 affine,1 copy,2 r180,3 r90,4 r270)} */
int aide_reverse_aug(const float* x, int64_t x_bs, float* y, int64_t y_bs, const double* par, int N, int C,
                     int H, int W, aide_stream_t stream);
int aide_fill_zero(float* p, int64_t bs, int N, int C, int H, int W, aide_stream_t stream);

/* ---- loss-module branches off the hot path ----------------------------------------------------------------------
 * aide_onehot_argmax: targets given one-hot [N][C][HW] -> int64 class indices (utils/loss2d.py:11-12).
 * aide_dice_terms_{fwd,bwd}: per image sum_k w_k (1 - (2 sum p_k t_k + s) / (sum p_k + sum t_k + s)), reduction 0 mean
 * (/N), 1 sum, 2 none.  K = 1: DiceLoss on a PROBABILITY input x [N][HW] (loss2d.py:47-61); K = 2: MulticlassDiceLoss with
 * one-hot targets t [N][2][HW] and class weights (w0, w1) on logits x [N][2][HW] (loss2d.py:96-104; softmax inside).
 * ws: aide_dice_terms_ws_bytes (kept by the caller between fwd and bwd).  g: [1] or [N] upstream gradient. */
int aide_onehot_argmax(const float* t, int64_t t_bs, int N, int C, int HW, long long* idx, aide_stream_t stream);
size_t aide_dice_terms_ws_bytes(int N, int HW);
int aide_dice_terms_fwd(const float* x, int64_t x_bs, const float* t, int64_t t_bs, int N, int HW, int K, float w0,
                        float w1, float smooth, int reduction, double* ws, float* per_image, float* out,
                        aide_stream_t stream);
int aide_dice_terms_bwd(const float* x, int64_t x_bs, const float* t, int64_t t_bs, int N, int HW, int K, float w0,
                        float w1, float smooth, int reduction, const double* ws, const float* g, float* dx,
                        int64_t dx_bs, aide_stream_t stream);

/* ---- stream ordering (host pointers): fork / join between the main stream and the weight-gradient stream as plain C
 * calls, so that a recorded launch sequence (aide_amd/tape.py) can re-issue them.  aide_stream_order(ev, from, to):
 * work enqueued on `to` afterwards waits for the work enqueued on `from` so far (hipEventRecord + hipStreamWaitEvent). */
int aide_event_create(void** ev);
int aide_event_destroy(void* ev);
int aide_stream_order(void* ev, aide_stream_t from, aide_stream_t to);
/* the two halves as separate calls (record first): the free-running second lane of the forward pass records an event
 * behind each of its pooling halves and the main stream waits for it only in front of the first reader */
int aide_event_record(void* ev, aide_stream_t from);
int aide_stream_wait_event(aide_stream_t to, void* ev);

/* ---- batched slab reduce of the weight gradients (host pointers) ---------------------------------------------------
 * Every aide_conv3x3_wgrad* call leaves per-split partial results ("slabs") in its workspace; with queue = NULL it reduces
 * them into dw with a small launch of its own.  Given a caller-owned queue (aide_wgrad_queue_create) the reduce is only
 * recorded there -- the call then needs a workspace region of its own that stays untouched until the flush -- and
 * aide_wgrad_queue_flush(queue, stream) runs ONE launch over everything recorded (<= 512 layers; fixed summation order per
 * element: deterministic, identical results) on `stream`, which must be ordered after the weight-gradient kernels.
 * aide_wgrad_queue_pending: entries recorded and not yet flushed.  aide_wgrad_queue_discard: error path, drops them (they
 * point into a pass that did not finish) and returns how many.  The queue is plain host memory owned by the caller: one
 * per launch sequence (the engine keeps one per plan); the library has no global state besides the kernel timer. */
int aide_wgrad_queue_create(void** queue);
int aide_wgrad_queue_destroy(void* queue);
int aide_wgrad_queue_pending(const void* queue);
int aide_wgrad_queue_flush(void* queue, aide_stream_t stream);
int aide_wgrad_queue_discard(void* queue);

/* ---- kernel timer (measurement only; bench.py `roofline`, `critical`, `streaming`) ----------------------------------
 * While armed for a family, every launch of that family's MAIN kernel carries a start / stop HIP event pair on its own
 * stream (hipExtLaunchKernelGGL): the pair records the dispatch's begin / end timestamps -- the duration rocprofv3
 * --kernel-trace reports -- without extra packets in the queue, so the two-stream schedule of the timed steps is
 * measured as it runs.  Families (bit ids of `family_mask`): 0 conv3x3_mfma_kernel, 1 conv3x3_wino_kernel,
 * 2 conv3x3_wino4_kernel, 3 conv3x3_wgrad_kernel, 4 conv3x3_wgrad_wino_kernel, 5 conv3x3_wgrad4_kernel,
 * 6 wgrad_stem_kernel, 7 conv3x3_bf16_kernel, 8 conv3x3_wgrad_bf16_kernel, 9 convT kernels (work = algorithmic
 * direct-convolution flop, 2 N H W Co Ci 9 per launch); the streaming families 10 BatchNorm forward, 11 BatchNorm backward,
 * 12 max-pooling, 13 bilinear up-sampling (work = algorithmic bytes of the launch: every operand read once, the result written
 * once), 14 split-K / slab reduces, 15 head, loss, Adam, filter packs (work 0).  A timed launch that also carries a hand-over
 * event (`done`) records that event with a packet of its own while the timer is armed.
 * aide_ktimer_start creates the events (call it outside the timed region); aide_ktimer_read / aide_ktimer_dump need an idle
 * device; aide_ktimer_read returns the number of launches that found no free slot (>= 0) or an error (< 0). */
int aide_ktimer_start(int family_mask, int capacity);
/* the launchers' own hook (not for callers): claims an event pair (hipEvent_t*) for one launch of `family` on `stream`; 0 = not armed */
int aide_ktimer_slot(int family, double work, aide_stream_t stream, void** e0, void** e1);
int aide_ktimer_stop(void);
int aide_ktimer_arm(int family_mask);        /* re-arm after a stop, keeping what was recorded */
int aide_ktimer_read(int family, int64_t* launches, double* ms, double* flops, double* max_ms);
/* every recorded launch in launch order: family, work, begin / end of the dispatch in ms after the begin of the first recorded
 * launch, launch stream (arrays of `capacity` entries) -> number of launches written, or an error (< 0) */
int aide_ktimer_dump(int capacity, int* family, double* work, double* begin_ms, double* end_ms, uint64_t* stream);

/* ---- Spatial_Attention branch of the attention variants (fuseunetsa / UNetsa) -----------------------------------
 * replaces Spatial_Attention.forward (models_twomodalinputs/netblocks.py:68-89, models_singlemodalinput/UNet.py:85-107)
 * and `y = sa(y) * y` (fuseunet.py:139-141, UNet.py:191-200) with their autograd backward:
 *   t1 = conv1x1(y; C -> R)   t2, t3 = dilated conv3x3 (R -> R, dilation = padding)   t4 = conv1x1(t3; R -> 1)
 *   gate = sigmoid(BatchNorm2d(1)(t4))   out[c] = gate * y[c]
 * Small-channel tensors (t1..t4, gate) are dense; C-channel tensors carry a batch stride (concat slices). */
int aide_pwconv_fwd(const float* x, int64_t x_bs, const float* w /*[R][C]*/, const float* b, float* y, int64_t y_bs,
                    int N, int C, int R, int HW, aide_stream_t stream);
/* dx (+)= gate * dout + W^T dt; gate [N][HW] and dout are both NULL for a plain 1x1 dgrad */
int aide_pwconv_dgrad(const float* dt, int64_t dt_bs, const float* w, const float* gate, const float* dout,
                      int64_t dout_bs, float* dx, int64_t dx_bs, int N, int C, int R, int HW, int accumulate,
                      aide_stream_t stream);
int aide_pwconv_wgrad(const float* dt, int64_t dt_bs, const float* x, int64_t x_bs, float* dw /*[R][C]*/,
                      float* db /*[R] or NULL*/, int N, int C, int R, int HW, aide_stream_t stream);
/* dilated 3x3 (padding = dilation), dense [N][C][H][W]; transposed != 0 computes the dgrad (x = dy, y = dx, b NULL) */
int aide_dconv3x3_small(const float* x, const float* w /*[Cout][Cin][3][3]*/, const float* b, float* y, int N, int Cin,
                        int Cout, int H, int W, int dilation, int transposed, aide_stream_t stream);
int aide_dconv3x3_small_wgrad(const float* dy, const float* x, float* dw, float* db, int N, int Cout, int Cin, int H,
                              int W, int dilation, aide_stream_t stream);
/* gate = sigmoid(bn(t4)) over M = N*H*W values of the single channel; stat[2] <- {mean, rstd} */
int aide_sa_gate_fwd(const float* t4, const float* gamma, const float* beta, float* running_mean, float* running_var,
                     int64_t* num_batches_tracked, float eps, float momentum, int training, float* stat, float* gate,
                     int64_t M, aide_stream_t stream);
int aide_sa_mul(const float* gate, const float* y, int64_t y_bs, float* out, int64_t out_bs, int N, int C, int HW,
                aide_stream_t stream);
/* dt4, dgamma, dbeta from dout (gradient of out) ; ws: N*HW floats + 16 bytes */
int aide_sa_gate_bwd(const float* dout, int64_t dout_bs, const float* y, int64_t y_bs, const float* gate,
                     const float* t4, const float* stat, const float* gamma, float* dgamma, float* dbeta, float* dt4,
                     int N, int C, int HW, float* ws, aide_stream_t stream);

/* ---- 1x1 head convolution -----------------------------------------------------------------------
 * replaces last_conv1 = nn.Conv2d(64, num_classes, 1): fuseunet.py:41,89 ; UNet.py:150,164.  K = num_classes, 1 .. 8 */
size_t aide_head1x1_ws_bytes(int C, int K);
/* x_bf16 / dx_bf16 = 1: the head on a bf16-stored feature map (precision='bf16'); logits and every gradient stay fp32 */
/* The head on the RAW output z of the conv layer under it (round 6): that layer's training-mode BatchNorm + ReLU
 * (netblocks.py:27-28 in front of last_conv1, fuseunet.py:88-89) is applied while z is read -- a = max(fma(z, scale[c], shift[c]), 0), the
 * expression of the BatchNorm apply kernels -- so the layer's normalising pass never runs and its activation is never stored.  in_scale /
 * in_shift [C]: e.g. from aide_bn_finalize_groups (statistics of the conv epilogue).  aide_head1x1_wgrad_bn: dw [K][C], db [K] from dlogits
 * and the recomputed activation; the data gradient belongs to aide_bn_relu_bwd_head.  fp32; ws: aide_head1x1_ws_bytes. */
int aide_head1x1_fwd_bn(const float* z, int64_t z_bs, const float* in_scale, const float* in_shift, const float* w, const float* b,
                        float* y, int64_t y_bs, int N, int C, int K, int H, int W, aide_stream_t stream);
int aide_head1x1_wgrad_bn(const float* dy, int64_t dy_bs, const float* z, int64_t z_bs, const float* in_scale, const float* in_shift,
                          float* dw, float* db, int N, int C, int K, int H, int W, void* ws, aide_stream_t stream);
int aide_head1x1_fwd_mixed(const void* x, int x_bf16, int64_t x_bs, const float* w, const float* b, float* y,
                           int64_t y_bs, int N, int C, int K, int H, int W, aide_stream_t stream);
int aide_head1x1_bwd_mixed(const float* dy, int64_t dy_bs, const void* x, int x_bf16, int64_t x_bs, const float* w,
                           void* dx, int dx_bf16, int64_t dx_bs, float* dw, float* db, int N, int C, int K, int H, int W,
                           void* ws, aide_stream_t stream);
/* ---- fused segmentation losses / co-teaching selection -----------------------------------------
 * replaces utils/loss2d.py:5-154, utils/coteach_loss.py:94-161, utils/metrics2d.py:8-29 and the
 * inline selection of train_files/trainchaos_proposed_30cases1labeled.py:274-321 */
int aide_seg_loss_blocks(int HW);
size_t aide_seg_loss_ws_bytes(int N, int HW);
int aide_seg_stats(const float* logits, int64_t l_bs, const long long* targets, int64_t t_bs, float w0,
                   float w1, int ignore_index, const float* pseudo, int64_t p_bs, const float* wmap,
                   int64_t w_bs, int N, int HW, double* partials, aide_stream_t stream);
int aide_seg_loss_finalize(const double* partials, int N, int HW, int reduction, float w_ce, float w_dice,
                           float smooth, double* stats, float* out, float* per_image, long long* idx,
                           float* coef, float* hard_dice, aide_stream_t stream);
int aide_coteach_finalize(const double* partials1, const double* partials2, int N, int HW, int variant,
                          int keep, float w_ce, float w_dice, float smooth, float rate, float w_seg,
                          float w_cor, double* stats1, double* stats2, float* loss, float* per_image1,
                          float* per_image2, long long* idx1, long long* idx2, float* coef1, float* coef2,
                          float* hard_dice, aide_stream_t stream);
int aide_seg_loss_bwd(const float* logits, int64_t l_bs, const long long* targets, int64_t t_bs, float w0,
                      float w1, int ignore_index, const float* pseudo, int64_t p_bs, const float* wmap,
                      int64_t w_bs, int N, int HW, const double* stats, const float* coef, float smooth,
                      const float* gout, int g_stride, float* dlogits, int64_t d_bs, aide_stream_t stream);
int aide_ce_map(const float* logits, int64_t l_bs, const long long* targets, int64_t t_bs, float w0,
                float w1, int ignore_index, int N, int HW, float* out, const float* gout, float* dlogits,
                int64_t d_bs, aide_stream_t stream);
int aide_mse_map(const float* logits, int64_t l_bs, const float* target, int64_t q_bs, int N, int HW,
                 float* out, const float* gout, float* dlogits, int64_t d_bs, aide_stream_t stream);
int aide_pseudo_label(const float* const* logits /* HOST array of K device pointers */, int K,
                      int64_t l_bs, int N, int HW, float temperature, float* pl, float* wm,
                      aide_stream_t stream);

/* ---- the same losses for num_classes = 3 .. aide_seg_max_classes() (csrc/loss_mc.hip) -------------------------------
 * The reference's modules are written for any class count (fuseunet(num_classes=...), models_twomodalinputs/
 * fuseunet.py:7,41; nn.CrossEntropyLoss(weight), utils/loss2d.py:8; softmax(...)[:, 1] in every Dice form,
 * utils/loss2d.py:44-46,65-66,96,106; MulticlassMSELoss :115-117; sharpen, trainchaos_proposed_30cases1labeled.py:97-101;
 * argmax(softmax), trainchaos_comparison_1case.py:262-264); its shipped scripts run two.  Logits / pseudo labels /
 * gradients are [N][C][HW] planes, class_w is a HOST array of C floats (NULL = ones).  The statistics (partials, stats)
 * have the layout of aide_seg_stats, so aide_seg_loss_finalize serves both; aide_coteach_finalize_mc takes C for the
 * mean of the consistency map (aide_coteach_finalize = C 2). */
int aide_seg_max_classes(void);
int aide_seg_stats_mc(const float* logits, int64_t l_bs, const long long* targets, int64_t t_bs,
                      const float* class_w /* HOST */, int C, int ignore_index, const float* pseudo, int64_t p_bs,
                      const float* wmap, int64_t w_bs, int N, int HW, double* partials, aide_stream_t stream);
int aide_seg_loss_bwd_mc(const float* logits, int64_t l_bs, const long long* targets, int64_t t_bs,
                         const float* class_w /* HOST */, int C, int ignore_index, const float* pseudo, int64_t p_bs,
                         const float* wmap, int64_t w_bs, int N, int HW, const double* stats, const float* coef,
                         float smooth, const float* gout, int g_stride, float* dlogits, int64_t d_bs,
                         aide_stream_t stream);
int aide_coteach_finalize_mc(const double* partials1, const double* partials2, int N, int HW, int C, int variant,
                             int keep, float w_ce, float w_dice, float smooth, float rate, float w_seg,
                             float w_cor, double* stats1, double* stats2, float* loss, float* per_image1,
                             float* per_image2, long long* idx1, long long* idx2, float* coef1, float* coef2,
                             float* hard_dice, aide_stream_t stream);
int aide_ce_map_mc(const float* logits, int64_t l_bs, const long long* targets, int64_t t_bs,
                   const float* class_w /* HOST */, int C, int ignore_index, int N, int HW, float* out,
                   const float* gout, float* dlogits, int64_t d_bs, aide_stream_t stream);
int aide_mse_map_mc(const float* logits, int64_t l_bs, const float* target, int64_t q_bs, int C, int N, int HW,
                    float* out, const float* gout, float* dlogits, int64_t d_bs, aide_stream_t stream);
int aide_label_map_mc(const float* logits, int64_t l_bs, int C, int N, int HW, long long* labels,
                      aide_stream_t stream);
int aide_pseudo_label_mc(const float* const* logits /* HOST array of K device pointers */, int K, int C,
                         int64_t l_bs, int N, int HW, float temperature, float* pl, float* wm,
                         aide_stream_t stream);
/* KLbidirection (utils/coteach_loss.py:85-92), the region cross entropy of Coteachingloss_dropregionce (:163-196; aux is one
 * 32-bit word per 2x2 region here) and the pixel term of Coteachingloss_dropimagedroppixel (:221-252) for C classes; the
 * selections are aide_select_smallest as in the two-class forms.  (Pixelcoreg_Focalloss reads channels 0 and 1 only in the
 * reference itself, utils/reg_loss.py:70-99: it has no C-class form.) */
int aide_kl_map_mc(const float* z1, int64_t b1, const float* z2, int64_t b2, int C, int N, int HW, float* out,
                   const float* gout, float* g1, int64_t gb1, float* g2, int64_t gb2, aide_stream_t stream);
int aide_region_ce_fwd_mc(const float* z, int64_t zb, const long long* t, int64_t tb, int C, int N, int H, int W,
                          int ignore_index, float* loss, unsigned* aux, aide_stream_t stream);
int aide_region_ce_bwd_mc(const float* z, int64_t zb, const unsigned* aux, const unsigned char* mask, const float* coeff,
                          int C, int N, int H, int W, float* dz, int64_t db, aide_stream_t stream);
int aide_droppixel_map_mc(const float* z1, int64_t b1, const float* z2, int64_t b2, const long long* t, int64_t tb,
                          const long long* idx, int ndrop, int C, int HW, int which, int ignore_index, float* v,
                          aide_stream_t stream);
int aide_droppixel_bwd_mc(const float* z1, int64_t b1, const float* z2, int64_t b2, const long long* t, int64_t tb,
                          const long long* idx, int ndrop, int C, int HW, int which, int ignore_index,
                          const unsigned char* mask, const float* coeff, float* g1, float* g2, aide_stream_t stream);
/* MulticlassDiceLoss with one-hot targets [N][C][HW] (utils/loss2d.py:98-104), one weighted Dice term per class */
size_t aide_dice_terms_mc_ws_bytes(int N, int HW, int C);
int aide_dice_terms_mc_fwd(const float* x, int64_t x_bs, const float* t, int64_t t_bs, int N, int HW, int C,
                           const float* class_w /* HOST */, float smooth, int reduction, double* ws,
                           float* per_image, float* out, aide_stream_t stream);
int aide_dice_terms_mc_bwd(const float* x, int64_t x_bs, const float* t, int64_t t_bs, int N, int HW, int C,
                           const float* class_w /* HOST */, float smooth, int reduction, const double* ws,
                           const float* g, float* dx, int64_t dx_bs, aide_stream_t stream);

/* ---- remaining co-teaching operators (utils/coteach_loss.py:85-92, 163-196, 198-254), two classes ----
 * aide_kl_map: KLbidirection, out[n][p] = KL(p1||p2) + KL(p2||p1); with gout (per-pixel upstream gradient) also
 *   the gradients w.r.t. both logit tensors.
 * aide_region_ce_fwd/_bwd: cross entropy on 2x2 max-pooled logits (per class) and targets (:171-179); aux keeps
 *   the arg-max window positions and the pooled target; bwd scatters coeff[0] * mask * dCE to those positions.
 * aide_select_smallest: per segment of M values the k smallest candidates (all values, or only those > 0), ties
 *   by lower index (stable argsort, :180-186 / :228-232); k = *k_in | (int64)(rr * candidates) if rr >= 0 | k_host;
 *   mask[M] marks the selection, sums[seg] = fp64 sum of sum_vals over it, ks[seg] = k.
 * aide_droppixel_map/_bwd: v = target * (KL(z1, z2) + CE(z_which, target)) on the images idx[0..ndrop) (:221-227,
 *   :240-246) and its gradient w.r.t. both logit tensors for the selected pixels (g1, g2 pre-zeroed). */
int aide_kl_map(const float* z1, int64_t b1, const float* z2, int64_t b2, int N, int HW, float* out,
                const float* gout, float* g1, int64_t gb1, float* g2, int64_t gb2, aide_stream_t stream);
int aide_region_ce_fwd(const float* z, int64_t zb, const long long* t, int64_t tb, int N, int H, int W,
                       int ignore_index, float* loss, unsigned char* aux, aide_stream_t stream);
int aide_region_ce_bwd(const float* z, int64_t zb, const unsigned char* aux, const unsigned char* mask,
                       const float* coeff, int N, int H, int W, float* dz, int64_t db, aide_stream_t stream);
/* the region cross entropy for any pooling window and class count: Coteachingloss_dropregionce(scale) pools with
 * MaxPool2d(kernel = stride = (KH, KW), ceil_mode=True), KH = int(H / int(H * scale)) (utils/coteach_loss.py:171-177) --
 * ceil(H / KH) x ceil(W / KW) regions, border windows clipped; C = 2 .. 8; aux: C + 1 words per region */
int aide_region_ce_fwd_win(const float* z, int64_t zb, const long long* t, int64_t tb, int C, int N, int H, int W, int KH,
                           int KW, int ignore_index, float* loss, int* aux, aide_stream_t stream);
int aide_region_ce_bwd_win(const float* z, int64_t zb, const int* aux, const unsigned char* mask, const float* coeff, int C,
                           int N, int H, int W, int KH, int KW, float* dz, int64_t db, aide_stream_t stream);
int aide_select_smallest(const float* sel_vals, const float* sum_vals, int64_t seg_stride, int nseg, int M,
                         int64_t k_host, double rr, const long long* k_in, int only_positive, unsigned char* mask,
                         double* sums, long long* ks, aide_stream_t stream);
int aide_droppixel_map(const float* z1, int64_t b1, const float* z2, int64_t b2, const long long* t, int64_t tb,
                       const long long* idx, int ndrop, int HW, int which, float* v, aide_stream_t stream);
int aide_droppixel_bwd(const float* z1, int64_t b1, const float* z2, int64_t b2, const long long* t, int64_t tb,
                       const long long* idx, int ndrop, int HW, int which, const unsigned char* mask,
                       const float* coeff, float* g1, float* g2, aide_stream_t stream);

/* Pixelcoreg_Focalloss / _twomodel (utils/reg_loss.py:58-193): per pixel key = (1-kd)(focal1 + focal2 [+ focal3]) +
 * kd KL(1,2) (focal: gamma 2, lossweight 1), val = focal3 with three nets (z3 != NULL), tf = target as float; the
 * per-image "k smallest keys" selection is aide_select_smallest; bwd: coeff[0] * mask * d val / d logits (two nets:
 * g1, g2 of the key; three nets: g3 only, as in the reference where the ranking is not differentiated). */
int aide_pixelcoreg_map(const float* z1, const float* z2, const float* z3, const long long* t, int64_t tb, int N,
                        int HW, float kd, float* key, float* val, float* tf, aide_stream_t stream);
int aide_pixelcoreg_bwd(const float* z1, const float* z2, const float* z3, const long long* t, int64_t tb, int N,
                        int HW, float kd, const unsigned char* mask, const float* coeff, float* g1, float* g2,
                        float* g3, aide_stream_t stream);

/* per-case inference (trainchaos_comparison_1case.py:262-264): labels[n][p] = argmax(softmax(logits[n][:,p]))
 * for two classes, int64 like torch.argmax; ties (also those created by the softmax rounding) -> 0 */
int aide_label_map(const float* logits, int64_t l_bs, int N, int HW, long long* labels, aide_stream_t stream);

/* ---- Adam(amsgrad), one launch for all parameter tensors ----------------------------------------
 * replaces torch.optim.Adam(net.parameters(), lr, amsgrad=True): trainchaos_comparison_1case.py:170 */
int aide_adam_amsgrad_multi(float* const* p, const float* const* g, float* const* m, float* const* v,
                            float* const* vmax, const int64_t* sizes, const int64_t* block_start,
                            int ntensors, int64_t total_blocks, float lr, float beta1, float beta2,
                            float eps, float weight_decay, int amsgrad, int64_t step,
                            aide_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* AIDE_HIP_H */
