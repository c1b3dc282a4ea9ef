/* satrn_hip.h -- C ABI of libsatrn_hip.so: the MI355X (gfx950) hot path of EfficientSATRN / LiteSATRN.
 *
 * The reference (bcaitech1/p4-fr-sorry-math-but-love-you) has no FFI: its "plugin API" for this path is the
 * Python nn.Module interface of networks/EfficientSATRN.py and networks/LiteSATRN.py.  Every entry point
 * below names the reference code it replaces (paths relative to the reference repo).  The Python mirror
 * that binds these symbols is p4-fr-sorry-math-but-love-you_amd/ (see INTEGRATION.md).
 *
 * Conventions
 *  - all pointers are DEVICE pointers borrowed from the caller (PyTorch tensors); the library allocates no
 *    device memory, never frees or retains a pointer beyond what is documented for satrn_model_bind /
 *    satrn_model_set_workspace;
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*);
 *  - return 0 on success, a negative code on error; satrn_last_error() gives the message (thread-local);
 *    nothing throws across the ABI;
 *  - dtype: 0 = f32 (exact-f32 MFMA; parity mode), 1 = bf16 storage with f32 accumulation;
 *  - activations are NHWC ("channels last"): a [B,C,H,W] reference tensor is stored as [B,H,W,C];
 *    for C == 1 inputs and for the encoder output [b, hw, c] the two layouts coincide;
 *  - channel counts must be multiples of 8 (bf16) / 4 (f32) except the image channels and the vocabulary.
 */
#ifndef SATRN_HIP_H
#define SATRN_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SATRN_F32 0
#define SATRN_BF16 1
#define SATRN_ACT_NONE 0
#define SATRN_ACT_RELU 1
#define SATRN_ACT_SILU 2
#define SATRN_ACT_SIGMOID 3

const char* satrn_last_error(void);
int satrn_abi_version(void);

/* ---------------------------------------------------------------------------------------------------
 * Operator level (one fwd/bwd pair per hot-path row of SURVEY.md section 8a)
 * ------------------------------------------------------------------------------------------------- */

/* Re-layout fp32 master weights into the compute dtype.
 * dense: w [N][K] -> fwd [N][K], bwd [K][ldb] (transposed, zero padded to ldb >= N).  nn.Linear / 1x1 conv.
 * conv3x3: w [Co][Ci][3][3] -> fwd [Co][9][Ci], bwd [Ci][9][Co].   depthwise: w [C][1][3][3] -> [9][C]. */
int satrn_pack_dense(int dtype, const float* w, void* fwd, void* bwd, int N, int K, int ldb, void* stream);
int satrn_pack_conv3x3(int dtype, const float* w, void* fwd, void* bwd, int Co, int Ci, void* stream);
int satrn_pack_dwconv3x3(int dtype, const float* w, void* out, int C, void* stream);

/* y[M][N] = act(x[M][K] * Wfwd[N][K]^T + bias) with optional dropout.  Replaces nn.Linear / 1x1 nn.Conv2d
 * (+ReLU/Sigmoid/Dropout that follow it): networks/EfficientSATRN.py:145-150 (PositionalEncoding dense0/1),
 * :200-214,225 (q/k/v/out_linear), :243,249 (conv0/conv1), :339-346 (Feedforward), :461 (generator).
 * out_f32 != 0 writes fp32 (the logits).  seed: device uint32 (may be NULL when drop_p == 0). */
int satrn_linear_fwd(int dtype, const void* x, const void* w_fwd, const float* bias, void* y, int M, int N, int K,
                     int act, int out_f32, float drop_p, const uint32_t* seed, uint32_t site, void* stream);
/* The 1x1 convolution / linear product whose epilogue also feeds the BatchNorm beside it (no bias; timm's conv_pw / conv_pwl + bn of
 * the EfficientNetV2 blocks, networks/EfficientSATRN.py:66,74,84, and the encoder's conv0 / norm0, conv1 / norm1, :243-252):
 *   y[M][N] (+)= x[M][K] * Wfwd[N][K]^T, and stats[r][0][n] += sum_m v, stats[r][1][n] += sum_m v^2 over the rows of replica r
 *   (stats: fp32 [stats_rep][2][N], ZERO on entry; the consumer adds the replicas) -- the batch statistics of the BatchNorm that follows;
 * with bnb_y != NULL (y is then the GRADIENT wrt a BatchNorm's output z = act(bn(bnb_y)), bnb_y [M][N]): the two sums are that BatchNorm's
 * backward reductions  sum g  and  sum g * xhat,  g = y_total * act'(bnb_y * scale + shift), xhat = (bnb_y - mean) * rstd, with
 * bnb_ss = [scale | shift] and bnb_mr = [mean | rstd] (fp32 [2N] each, as written by satrn_batchnorm_act_fwd). */
int satrn_linear_fwd_stats(int dtype, const void* x, const void* w_fwd, void* y, int M, int N, int K, float* stats, int stats_rep,
                           const void* bnb_y, const float* bnb_ss, const float* bnb_mr, int bnb_act, int accumulate, void* stream);
/* The encoder layer's self-attention region in one launch + the LayerNorm behind it (bf16; EncoderLayer.forward,
 * networks/EfficientSATRN.py:260-268: `out = self.norm(input); out = attention(out, out, out); out = self.norm(out + input)` with the
 * SHARED LayerNorm, MultiHeadAttention :198-228, temperature sqrt(heads * head_dim) :187-189).  x [B*L][D] (L = h*w tokens <= 64,
 * D = 256 or 512, head_dim 64, even head count); wqkv = the packed fused projection [3D][D] (q | k | v rows), wo [D][D].
 * Outputs = everything the step-wise operators would have produced (the unfused backward reads them): y1 = norm(x) with mean | rstd,
 * qkv [B*L][3D], the attention output [B*L][D] and its log-sum-exp [B][heads][L], o = dropout(out_linear(att) + bo) and y2 = norm(o + x)
 * with its mean | rstd.  parts_scratch: (heads / 2) * B*L*D elements.  Dropout sites / indices are those of satrn_attention_fwd and
 * satrn_linear_fwd, so both forms draw the same masks. */
int satrn_enc_attn_region_fwd(const void* x, const float* ln_w, const float* ln_b, const void* wqkv, const float* bqkv, const void* wo, const float* bo,
                              int B, int L, int D, int heads, float attn_drop, float out_drop, const uint32_t* seed, uint32_t site_attn, uint32_t site_out,
                              void* y1, float* mean_rstd1, void* qkv, void* att, float* lse, void* parts_scratch, void* o, void* y2, float* mean_rstd2,
                              void* stream);
/* dx[M][K] (+)= dy[M][ldy(>=N)] * W  using Wbwd [K][ldb]; `accumulate` adds into dx. */
int satrn_linear_bwd_data(int dtype, const void* dy, int ldy, const void* w_bwd, int ldb, void* dx, int M, int N,
                          int K, int accumulate, void* stream);
/* dw[N][K] += dy^T x  (fp32, atomics: zero dw first);  db[N] += column sums of dy (db may be NULL). */
int satrn_linear_bwd_weight(int dtype, const void* dy, int ldy, const void* x, float* dw, float* db, int M, int N,
                            int K, void* stream);
/* The same with a caller-provided partial-tile slab (fp32, ws_floats >= slices * N * K; 320 * 16384 always suffices): the large-product
 * kernel then stores each item's partial tile [slice][N][K] with plain stores and a fold launch adds the slices in slice order --
 * no float atomics into dw (deterministic; inside a training step, where the gradient buffers are cold, the atomics cost as much
 * as the product).  This is the form the engine's steps use.  Shapes the large-product kernel does not take behave like
 * satrn_linear_bwd_weight. */
int satrn_linear_bwd_weight_ws(int dtype, const void* dy, int ldy, const void* x, float* dw, float* db, int M, int N, int K,
                               float* ws, size_t ws_floats, void* stream);
/* Linear + activation for a training forward: y = act(x W^T + b) and dact = act'(x W^T + b), both [M][N] in the compute dtype --
 * the backward needs the pre-activation only through that derivative, which the epilogue has beside the activation (GELU: one erf / exp
 * evaluation for both).  SwinTRN's Mlp.fc1 + nn.GELU (networks/SWIN.py:24-47).  act: 1 ReLU, 2 SiLU, 3 sigmoid, 4 GELU (exact erf form). */
int satrn_linear_act_fwd(int dtype, const void* x, const void* w_fwd, const float* bias, void* y, void* dact, int M, int N, int K, int act,
                         void* stream);
/* dx[M][K] = (dy[M][ldy(>=N)] * W) (.) dact[M][K] * scale: the data gradient of the linear layer BEHIND an activation, leaving as the
 * gradient of that activation's input (dact = satrn_linear_act_fwd's second output of the layer in front; kind 5).  kind 1: dact is
 * the stored OUTPUT of a ReLU (+ dropout with keep scale `scale`): factor = scale where it is positive, else 0.  scale 0 means 1. */
int satrn_linear_bwd_data_act(int dtype, const void* dy, int ldy, const void* w_bwd, int ldb, const void* dact, int kind, float scale,
                              void* dx, int M, int N, int K, void* stream);

/* 3x3 convolution, NHWC, im2col-free implicit GEMM on MFMA.  Replaces nn.Conv2d(k=3) of
 * networks/LiteSATRN.py:50-70 (ShallowCNN conv1..3) and the timm EfficientNetV2-S 3x3 convs behind
 * networks/EfficientSATRN.py:74,84.  x [B,H,W,Ci], y [B,OH,OW,Co]; pt/pl = top/left padding (TF "SAME"
 * strided convs pad bottom/right more). */
int satrn_conv3x3_fwd(int dtype, const void* x, const void* w_fwd, void* y, int B, int H, int W, int Ci, int Co, int OH,
                      int OW, int stride, int pt, int pl, void* stream);
/* Inference forms of the backbone's convolutions (model.eval(): timm ConvBnAct / FusedMBConv / MBConv run by
 * networks/EfficientSATRN.py:74-76,82-87 and the shallow CNN of networks/LiteSATRN.py:51-69): the product with the eval-mode BatchNorm
 * that follows it (escale = weight/sqrt(running_var+eps), eshift = bias - running_mean*escale), its activation and the block's
 * residual (res, [rows][Co] in the compute dtype, may be NULL) in the epilogue:  y = act(conv(x)*escale + eshift) + res.
 * The 1x1 convolutions are the linear form over NHWC rows. */
int satrn_conv3x3_bn_eval_act_fwd(int dtype, const void* x, const void* w_packed, const float* escale, const float* eshift, int act,
                                  const void* res, void* y, int B, int H, int W, int Ci, int Co, int OH, int OW, int stride, int pt,
                                  int pl, void* stream);
int satrn_linear_bn_eval_act_fwd(int dtype, const void* x, const void* w, const float* escale, const float* eshift, int act,
                                 const void* res, void* y, int M, int N, int K, void* stream);
int satrn_conv3x3_bwd_data(int dtype, const void* dy, const void* w_bwd, void* dx, int B, int H, int W, int Ci, int Co,
                           int OH, int OW, int stride, int pt, int pl, int accumulate, void* stream);
/* dw in the torch layout [Co][Ci][3][3], fp32 atomics (zero first) */
int satrn_conv3x3_bwd_weight(int dtype, const void* dy, const void* x, float* dw, int B, int H, int W, int Ci, int Co,
                             int OH, int OW, int stride, int pt, int pl, void* stream);
/* first conv on the fp32 NCHW image (Cin = 1 or 3): networks/EfficientSATRN.py:67-69,82 (conv_stem, stride 2
 * pad 0) and networks/LiteSATRN.py:24-26,51 (conv0, stride 1 pad 1).  w is the fp32 master [Co][Cin][3][3]. */
int satrn_stem_conv_fwd(int dtype, const float* img, const float* w, void* y, int B, int Cin, int H, int W, int Co,
                        int stride, int pad, void* stream);
int satrn_stem_conv_bwd_weight(int dtype, const float* img, const void* dy, float* dw, int B, int Cin, int H, int W,
                               int Co, int stride, int pad, void* stream);
/* depthwise 3x3 (+bias): networks/EfficientSATRN.py:245-247,274 and the timm MBConv conv_dw. */
int satrn_dwconv3x3_fwd(int dtype, const void* x, const void* w_packed, const float* bias, void* y, int B, int H, int W,
                        int C, int OH, int OW, int stride, int pt, int pl, void* stream);
/* Inference form of the MBConv depthwise seam (timm DepthwiseSeparable / InvertedResidual: conv_dw -> bn2 -> SiLU -> se pool, as
 * run by networks/EfficientSATRN.py:74-76 under model.eval()): stride-1 "same" depthwise 3x3 (+bias) of an already activated x,
 * eval-mode BatchNorm handed over as per-channel escale = weight/sqrt(running_var+eps), eshift = bias - running_mean*escale,
 * activation, and -- optional -- pool[b][c] = sum over the image of y (the squeeze-and-excite mean is pool/(H*W)).  One launch on
 * the small maps (bf16, C % 64 == 0, W % 3 == 0, H*W/3*8 <= 512 threads); other shapes run the generic kernel + a pooling pass. */
int satrn_dwconv3x3_bn_eval_act_pool_fwd(int dtype, const void* x, const void* w_packed, const float* bias, const float* escale,
                                         const float* eshift, int act, void* y, float* pool, int B, int H, int W, int C, void* stream);
int satrn_dwconv3x3_bwd_data(int dtype, const void* dy, const void* w_packed, void* dx, int B, int H, int W, int C, int OH,
                             int OW, int stride, int pt, int pl, int accumulate, void* stream);
int satrn_dwconv3x3_bwd_weight(int dtype, const void* x, const void* dy, float* dw, float* dbias, int B, int H, int W,
                               int C, int OH, int OW, int stride, int pt, int pl, void* stream);

/* BatchNorm2d (batch statistics when train != 0, running statistics otherwise) fused with the activation
 * that follows it and an optional residual added AFTER the activation: z = act(bn(y)) (+ res).
 * networks/LiteSATRN.py:52-69, networks/EfficientSATRN.py:83,86,272-279 and the timm blocks.
 * scratch: 6*C floats; the first 2*C must be ZERO on entry; after the call scratch[2C..4C) = scale/shift and
 * scratch[4C..6C) = mean/rstd, which satrn_batchnorm_act_bwd needs.  running_* are updated in place
 * (momentum 0.1, unbiased variance) when train != 0. */
int satrn_batchnorm_act_fwd(int dtype, const void* y, const float* weight, const float* bias, float* running_mean,
                            float* running_var, int64_t* num_batches_tracked, float eps, int train, int act,
                            const void* res, void* z, long M, int C, float* scratch, void* stream);
/* Training-mode BatchNorm2d + activation + the squeeze-and-excite block behind it (timm MBConv: bn2 -> SiLU -> SqueezeExcite, run by
 * networks/EfficientSATRN.py:74-76): z = act(bn(y)), pooled = mean over the image, u1 = W1 pooled + b1, s1 = silu(u1),
 * gate = sigmoid(W2 s1 + b2), out = z * gate.  One launch on the small maps (bf16, C % 64 == 0, C <= 1536, S <= 64, S % 8 == 0,
 * B * C / 64 workgroups all resident at once -- asked of the occupancy query for the instantiation that would run): the workgroups of an image hand their shares of the hidden layer to each
 * other through `mailbox` -- mailbox_images * 1600 8-byte words that are ZERO before the first call and that nothing else writes
 * (a launch number tags every word, so the mailbox is never cleared between calls).  Other shapes / dtypes: the separate kernels.
 * keep_z == 0: z need not hold the activated tensor afterwards (the engine's backward recomputes it from y); z must be a valid
 * [B][HW][C] buffer either way.  scratch as in satrn_batchnorm_act_fwd (6*C floats, first 2*C ZERO on entry).  A hand-off that
 * does not complete within 2 s sets bit 2 of the device error word (satrn_device_error) instead of hanging. */
int satrn_batchnorm_act_se_fwd(int dtype, const void* y, const float* weight, const float* bias, float* running_mean, float* running_var,
                               int64_t* num_batches_tracked, float eps, int act, void* z, int keep_z, const void* W1, const float* b1,
                               const void* W2, const float* b2, float* pooled, float* u1, float* s1, void* gate, void* out, int B, int HW,
                               int C, int S, float* scratch, unsigned long long* mailbox, int mailbox_images, void* stream);
/* The FRONT of a timm MBConv block (SURVEY Appendix B stages 3-5; networks/EfficientSATRN.py:74,84 -> timm `.blocks`: conv_pw -> bn1 -> SiLU
 * -> conv_dw -> bn2 -> SiLU -> se) in training mode as ONE launch, bf16 only, on the small maps of the late stages (H * W = 48 with
 * Cin = 256, or 192 with Cin = 160 / 128; W % 3 == 0, C % 64 == 0, C <= 1536, S <= 64, S % 8 == 0, B <= 64):
 *   y1 = x W0^T, z1 = SiLU(BatchNorm_1(y1)), y2 = depthwise3x3(z1) (stride 1, "same", no bias), z2 = SiLU(BatchNorm_2(y2)),
 *   pooled = mean_hw z2, u1 = W1 pooled + b1, s1 = SiLU(u1), gate = sigmoid(W2 s1 + b2), z3 = z2 * gate.
 * Both BatchNorms use BATCH statistics: the workgroups (image, 64-channel slab) exchange their per-channel sums through `mailbox`
 * (3 * (C / 64) * B * 128 + B * (C / 64) * 64 8-byte words, ZERO before the first call, written by nothing else; a launch number tags
 * every word, so it is never cleared) and add them in image order (deterministic statistics).  Everything the separate operators
 * (satrn_linear_fwd_stats, satrn_batchnorm_act_dwconv3x3_fwd, satrn_batchnorm_act_se_fwd) leave behind is written: y1, z1, y2, z3
 * [B][H][W][C], z2 when keep_z2, bn*_coef = scale | shift | mean | rstd (4 * C floats each), running statistics (momentum 0.1) and
 * num_batches_tracked, pooled [B][C], u1 / s1 [B][S], gate [B][C].  x [B][H][W][Cin]; W0 [C][Cin] (satrn_pack_dense fwd); dw_packed
 * [9][C] (satrn_pack_dwconv3x3); W1 [S][C], W2 [C][S] bf16.  Returns -1 for shapes the one launch does not take (incl. a grid that
 * would not be resident at once: the workgroups wait for each other); a wait that times out (2 s) sets device error bit 2. */
int satrn_mbconv_front_fwd(const void* x, const void* W0, void* y1, const float* bn1_weight, const float* bn1_bias, float* bn1_running_mean,
                           float* bn1_running_var, int64_t* bn1_num_batches_tracked, float* bn1_coef, void* z1, const void* dw_packed, void* y2,
                           const float* bn2_weight, const float* bn2_bias, float* bn2_running_mean, float* bn2_running_var,
                           int64_t* bn2_num_batches_tracked, float* bn2_coef, void* z2, int keep_z2, const void* W1, const float* b1, const void* W2,
                           const float* b2, float* pooled, float* u1, float* s1, void* gate, void* z3, int B, int H, int W, int Cin, int C, int S,
                           float eps, unsigned long long* mailbox, long mailbox_words, void* stream);
/* The same with the block INPUT taken as the output of the BatchNorm that ends the block in front (timm InvertedResidual: bn3, no activation,
 * + that block's residual): x = BatchNorm(in_y) (+ in_res), batch statistics from in_sums (in_sums_rep replicas of [sum | sum of squares],
 * 2 * Cin floats each, as satrn_linear_fwd_stats leaves them).  Every workgroup normalises its image's rows while it stages them, x is also
 * written to x_out [B][H][W][Cin], in_coef receives scale | shift | mean | rstd (4 * Cin floats) and the running statistics are updated --
 * what satrn_batchnorm_act_fwd(act = none, residual) does, without its launch. */
int satrn_mbconv_front_fwd_bn_in(const void* in_y, const void* in_res, const float* in_sums, int in_sums_rep, const float* in_weight, const float* in_bias,
                                 float* in_running_mean, float* in_running_var, int64_t* in_num_batches_tracked, float* in_coef, void* x_out,
                                 const void* W0, void* y1, const float* bn1_weight, const float* bn1_bias, float* bn1_running_mean,
                                 float* bn1_running_var, int64_t* bn1_num_batches_tracked, float* bn1_coef, void* z1, const void* dw_packed, void* y2,
                                 const float* bn2_weight, const float* bn2_bias, float* bn2_running_mean, float* bn2_running_var,
                                 int64_t* bn2_num_batches_tracked, float* bn2_coef, void* z2, int keep_z2, const void* W1, const float* b1,
                                 const void* W2, const float* b2, float* pooled, float* u1, float* s1, void* gate, void* z3, int B, int H, int W,
                                 int Cin, int C, int S, float eps, unsigned long long* mailbox, long mailbox_words, void* stream);
/* Backward of the same block, first half of its middle, in ONE launch (bf16, the shapes of satrn_mbconv_front_fwd): the projection's data
 * gradient dz3 = dy3 W_proj (dy3 [B][H][W][Cout] = the gradient at conv_pwl's output; w_bwd = satrn_pack_dense's backward pack of W_proj,
 * [C][ldb]) and the squeeze-and-excite backward with z2 RECOMPUTED from BatchNorm 2's input bn2_y and coefficients bn2_coef (scale | shift |
 * mean | rstd, 4 * C floats):  dgate = sum_hw dz3 * z2, dz2 = dgate * gate * (1 - gate) [B][C] fp32, ds1 = W2^T dz2 and du1 = ds1 * SiLU'(u1)
 * [B][S] fp32, dpooled = W1^T du1 [B][C] bf16, and bn2_sums[2C] += this batch's BatchNorm-2 backward sums (sum g | sum g * xhat with
 * g = (dz3 * gate + dpooled / HW) * SiLU'(.)): what satrn_linear_bwd_data + satrn_se_bwd_bnred produce.  dz3 [B][H][W][C] is written for the
 * depthwise backward (satrn_bn_bwd_apply_dwconv3x3_bwd_data_bnred takes it with gate / dpooled).  Only the C / 64 workgroups of one image
 * exchange data (mailbox: B * (C / 64) * 64 8-byte words, ZERO before the first call; shareable with the other mailbox operators), so the
 * launch needs no whole-grid residency.  Returns -1 for shapes it does not take. */
int satrn_mbconv_bwd_se(const void* dy3, const void* w_bwd, int ldb, void* dz3, const void* bn2_y, const float* bn2_coef, const void* gate,
                        const float* u1, const void* W1, const void* W2, float* dz2, float* ds1, float* du1, void* dpooled, float* bn2_sums, int B,
                        int H, int W, int Cout, int C, int S, unsigned long long* mailbox, long mailbox_words, void* stream);
/* The same with dy3 taken as the BACKWARD of the block-ending BatchNorm (timm InvertedResidual.bn3: batch statistics, no activation) applied to
 * the gradient dz at the block's output: bn3_y = that BatchNorm's input, bn3_coef = its scale | shift | mean | rstd (4 * Cout floats),
 * bn3_sums = [sum dz | sum dz * xhat] (2 * Cout floats; what satrn_linear_fwd_stats's bnb form or satrn_batchnorm_act_bwd's reduction leaves).
 * dy3_out [B][H][W][Cout] receives the result (the projection's weight gradient reads it), bn3_dweight / bn3_dbias accumulate the BatchNorm's
 * parameter gradients -- satrn_batchnorm_act_bwd_apply's job, without its launch. */
int satrn_mbconv_bwd_se_bn_in(const void* dz, const void* bn3_y, const float* bn3_coef, const float* bn3_weight, const float* bn3_sums, void* dy3_out,
                              float* bn3_dweight, float* bn3_dbias, const void* w_bwd, int ldb, void* dz3, const void* bn2_y, const float* bn2_coef,
                              const void* gate, const float* u1, const void* W1, const void* W2, float* dz2, float* ds1, float* du1, void* dpooled,
                              float* bn2_sums, int B, int H, int W, int Cout, int C, int S, unsigned long long* mailbox, long mailbox_words, void* stream);
/* Training-mode BatchNorm2d + activation of y[B][H][W][C] -> z, followed by the stride-1 "same" depthwise 3x3 (+bias) of z -> out,
 * in one launch where the shape allows (the expand-BN-SiLU-depthwise seam of the timm MBConv block in the 8x24 / 4x12 stages;
 * networks/EfficientSATRN.py:74-76 runs those blocks).  Results equal satrn_batchnorm_act_fwd + satrn_dwconv3x3_fwd bit for bit
 * (same operand rounding, same accumulation order); other shapes / dtypes run exactly those two.  scratch as in
 * satrn_batchnorm_act_fwd (6*C floats, first 2*C ZERO on entry); out_stats: 2*C floats, ZERO on entry, receives the column sums
 * and sums of squares of `out` (what the next BatchNorm needs) -- may be NULL. */
int satrn_batchnorm_act_dwconv3x3_fwd(int dtype, const void* y, const float* weight, const float* bias, float* running_mean,
                                      float* running_var, int64_t* num_batches_tracked, float eps, int act, void* z,
                                      const void* dw_packed, const float* dw_bias, void* out, float* out_stats, int B, int H,
                                      int W, int C, float* scratch, void* stream);
/* Backward of the same seam: dz (+)= data gradient of the stride-1 "same" depthwise 3x3 given d_out, AND the two column sums the
 * BatchNorm backward of z = act(bn(y)) needs (scratch2 of satrn_batchnorm_act_bwd: sum g, sum g*xhat with g = dz*act'), one
 * launch where the shape allows; otherwise satrn_dwconv3x3_bwd_data + the reduction pass.  `scratch` is the forward's
 * (scale/shift at [2C,4C), mean/rstd at [4C,6C)); scratch2: 2*C floats, ZERO on entry.  Follow with
 * satrn_batchnorm_act_bwd_apply. */
int satrn_dwconv3x3_bwd_data_bnred(int dtype, const void* d_out, const void* dw_packed, void* dz, int accumulate, const void* y,
                                   const float* scratch, int act, float* scratch2, int B, int H, int W, int C, void* stream);
/* satrn_batchnorm_act_bwd_apply of the BatchNorm BEHIND the depthwise convolution (its output gradient dz2, raw input y2 = the
 * convolution's output, column sums in scratch2_b) + satrn_dwconv3x3_bwd_data_bnred, one launch where the shape allows: the
 * apply result dy2 is written (the depthwise weight gradient reads it) and consumed from LDS.  Same results as the two calls. */
int satrn_bn_bwd_apply_dwconv3x3_bwd_data_bnred(int dtype, const void* dz2, const void* y2, const float* weight_b, const float* scratch_b, int act_b,
                                                const float* scratch2_b, void* dy2, float* dweight_b, float* dbias_b, const void* dw_packed,
                                                void* dz, int accumulate, const void* y, const float* scratch, int act, float* scratch2, int B,
                                                int H, int W, int C, void* stream);
/* ... and the WHOLE backward of the BatchNorm in front of the convolution as well (bf16; z = act(bn_a(y)) is what the convolution read):
 * satrn_bn_bwd_apply_dwconv3x3_bwd_data_bnred + satrn_batchnorm_act_bwd_apply(dz, y, weight_a, scratch, act, ...) in one launch.  The B
 * workgroups of a 64-channel slab exchange their shares of that BatchNorm's two column sums through `mailbox` (B * (C / 64) * 128 8-byte
 * words, ZERO before the first call, shareable with the other mailbox operators) and write dy1 = the gradient of bn_a's input straight from
 * their registers; dz (the gradient of z) is never stored.  dweight_a / dbias_a (+)= bn_a's parameter gradients.  The launch needs the
 * whole grid resident: returns -1 for shapes / batch sizes it does not take (call the two operators instead).
 * Reference: the backward of timm's InvertedResidual bn1 -> act1 -> conv_dw -> bn2 (networks/EfficientSATRN.py:74,84). */
int satrn_bn_bwd_apply_dwconv3x3_bwd_data_bn_bwd(const void* dz2, const void* y2, const float* weight_b, const float* scratch_b, int act_b,
                                                 const float* scratch2_b, void* dy2, float* dweight_b, float* dbias_b, const void* dw_packed,
                                                 const void* y, const float* weight_a, const float* scratch, int act, void* dy1, float* dweight_a,
                                                 float* dbias_a, int B, int H, int W, int C, unsigned long long* mailbox, long mailbox_words,
                                                 void* stream);
/* the second half of satrn_batchnorm_act_bwd for callers that already hold the column sums in scratch2 */
int satrn_batchnorm_act_bwd_apply(int dtype, const void* dz, const void* y, const float* weight, const float* scratch, int act,
                                  void* dy, float* dweight, float* dbias, long M, int C, const float* scratch2, void* stream);
/* dy = d(loss)/d(y) given dz; dweight/dbias accumulate (+=).  scratch2: 2*C floats, ZERO on entry. */
int satrn_batchnorm_act_bwd(int dtype, const void* dz, const void* y, const float* weight, const float* scratch,
                            int act, void* dy, float* dweight, float* dbias, long M, int C, float* scratch2,
                            void* stream);

/* MaxPool2d(2,2): networks/LiteSATRN.py:53,58,63,68.  bwd recomputes the argmax (first maximum wins). */
int satrn_maxpool2x2_fwd(int dtype, const void* x, void* y, int B, int H, int W, int C, void* stream);
int satrn_maxpool2x2_bwd(int dtype, const void* x, const void* dy, void* dx, int B, int H, int W, int C, void* stream);

/* LayerNorm over the last dim of (a + b) (b may be NULL): networks/EfficientSATRN.py:265,268,378,382,385.
 * mean_rstd: 2*R floats saved for the backward. */
int satrn_layernorm_fwd(int dtype, const void* a, const void* b, const float* weight, const float* bias, void* out,
                        float* mean_rstd, long R, int C, float eps, void* stream);
int satrn_layernorm_bwd(int dtype, const void* dout, const void* a, const void* b, const float* weight,
                        const float* mean_rstd, void* da, void* db, int acc_a, int acc_b, float* dweight, float* dbias,
                        long R, int C, void* stream);

/* Squeeze-and-excite of the EfficientNetV2-S MBConv blocks (timm SqueezeExcite behind networks/EfficientSATRN.py:74,84):
 * gate[b] = sigmoid(W2 * silu(W1 * mean_hw(x[b]) + b1) + b2), y = x * gate.  x, y: [B][HW][C]; W1 [S][C] and W2 [C][S] in the
 * compute dtype (the packed copies); pooled [B][C], u1 / s1 [B][S] fp32 (saved for the backward); gate [B][C].
 * pool_sums: optional [B][C] fp32 sums over HW of x (what the BatchNorm pass in front accumulates in the training step); when
 * given (bf16, S <= 64, C <= 1536) the MLP + scale run as one launch over B x 8 channel groups, otherwise pool + MLP per image
 * and the scale as a second launch.
 * backward (data path): dz2 [B][C], du1 [B][S] fp32 (inputs of the weight-gradient products), dpooled [B][C] in the compute
 * dtype; the gradient wrt x is dy*gate + dpooled/HW (folded into the BatchNorm backward that follows in the engine).
 * ds1_zeroed: [B][S] fp32 zeros (scratch of the two-launch wide form; NULL: the per-image form is used, which needs
 * dgate_scratch [B][C] in the compute dtype instead). */
int satrn_se_fwd(int dtype, const void* x, const void* W1, const float* b1, const void* W2, const float* b2, const float* pool_sums,
                 float* pooled, float* u1, float* s1, void* gate, void* y, int B, int HW, int C, int S, void* stream);
int satrn_se_bwd(int dtype, const void* dy, const void* x, const void* gate, const float* u1, const void* W1, const void* W2,
                 float* dz2, float* du1, float* ds1_zeroed, void* dgate_scratch, void* dpooled, int B, int HW, int C, int S,
                 void* stream);
/* satrn_se_bwd for the timm MBConv seam BatchNorm -> SiLU -> SqueezeExcite, where x = act(bn(bn_y)) feeds the SE block only: x is
 * recomputed from bn_y (not read), and bn_scratch2 (2*C floats, ZERO on entry) receives the column sums that BatchNorm's backward
 * needs for the gradient dy*gate + dpooled/HW reaching its output (so satrn_batchnorm_act_bwd's reduction pass is not needed;
 * follow with the apply pass).  bn_scratch = the BatchNorm forward's scratch (scale/shift at [2C,4C), mean/rstd at [4C,6C));
 * P_scratch: 4*B*C floats.  bf16 wide form only: returns -1 for shapes / dtypes the wide form does not take. */
int satrn_se_bwd_bnred(int dtype, const void* dy, const void* bn_y, const float* bn_scratch, int act, const void* gate, const float* u1,
                       const void* W1, const void* W2, float* dz2, float* du1, float* ds1_zeroed, void* dpooled, float* P_scratch,
                       float* bn_scratch2, int B, int HW, int C, int S, void* stream);
/* The same in ONE launch, given a mailbox (mailbox_images * 1600 8-byte words, ZERO before the first use, written by nothing but the
 * mailbox operators -- the one of satrn_batchnorm_act_se_fwd can be shared): the workgroups of an image exchange their shares of
 * W2^T dz2 through it and add them in a fixed order (no float atomics on ds1: that part is deterministic); ds1_zeroed then receives
 * the sums (it need not be zero).  B > mailbox_images, B * 8 workgroups beyond what the device holds resident at once (occupancy query for the kernel), or mailbox == NULL: the two
 * launches of satrn_se_bwd_bnred.  A hand-off that times out sets device error bit 2.  (Alone on the device the one launch is
 * shorter; inside the engine's training step, beside the weight-gradient stream, it measured slower and the engine keeps the two.) */
int satrn_se_bwd_bnred_mbox(int dtype, const void* dy, const void* bn_y, const float* bn_scratch, int act, const void* gate, const float* u1,
                            const void* W1, const void* W2, float* dz2, float* du1, float* ds1_zeroed, void* dpooled, float* P_scratch,
                            float* bn_scratch2, int B, int HW, int C, int S, unsigned long long* mailbox, int mailbox_images, void* stream);

/* Weight gradients of the two SqueezeExcite matrices (timm SqueezeExcite conv_reduce / conv_expand, networks/EfficientSATRN.py:74,84)
 * from what satrn_se_fwd / satrn_se_bwd left: dW2 [C][S] += dz2^T s1, db2 [C] += colsum(dz2), dW1 [S][C] += du1^T pooled,
 * db1 [S] += colsum(du1).  All operands fp32 (dz2, pooled [B][C]; du1, s1 [B][S]); the outputs are ACCUMULATED into (the engine's
 * fp32 gradient buffers).  The sum over the batch is taken in a fixed order (no atomics). */
int satrn_se_bwd_weights(const float* dz2, const float* du1, const float* s1, const float* pooled, float* dW1, float* db1, float* dW2,
                         float* db2, int B, int C, int S, void* stream);

/* Adaptive 2D positional encoding, networks/EfficientSATRN.py:135-154: out = x + g0*hpos[h] + g1*wpos[w]
 * with gate [B][2C] = sigmoid(dense1(relu(dense0(mean_hw x)))) computed by satrn_pool_hw + satrn_linear_fwd. */
int satrn_pool_hw(int dtype, const void* x, void* out, int B, int HW, int C, void* stream);
int satrn_posenc2d_fwd(int dtype, const void* x, const void* gate, const float* hpos, const float* wpos, void* out,
                       int B, int H, int W, int C, void* stream);
int satrn_posenc2d_bwd_gate(int dtype, const void* dout, const float* hpos, const float* wpos, void* dgate, int B, int H,
                            int W, int C, void* stream);

/* The EncoderLayer raw reshape (networks/EfficientSATRN.py:269): the [b,hw,c] buffer reinterpreted as
 * [b,c,h,w], delivered in NHWC.  inverse != 0 routes gradients back. */
int satrn_encoder_reshape(int dtype, int inverse, const void* in, void* out, int B, int HW, int C, int accumulate,
                          void* stream);

/* Scaled dot-product attention with the reference's temperature sqrt(heads*head_dim), -inf masks and
 * attention-probability dropout: networks/EfficientSATRN.py:164-172 inside :198-228.
 * q [B,Lq,ldq] / k,v [B,Lk,ldk] are column slices (head h at columns h*hd) of projection outputs; o [B,Lq,ldo].
 * text (int64 [B][ld_text], may be NULL): key j>0 is masked when text[b][j] == pad_id (:469-473);
 * causal != 0 masks key j > query i (:475-478).  lse [B,heads,Lq] is saved for the backward. */
int satrn_attention_fwd(int dtype, const void* q, const void* k, const void* v, void* o, float* lse, int B, int heads,
                        int Lq, int Lk, int hd, int ldq, int ldk, int ldv, int ldo, int causal, const int64_t* text,
                        int ld_text, int pad_id, float temperature, float drop_p, const uint32_t* seed, uint32_t site,
                        void* stream);
/* ws: 2 * B*heads*Lq*roundup(Lk,32) elements of the compute dtype.  dq/dk/dv have the strides of q/k/v. */
int satrn_attention_bwd(int dtype, const void* q, const void* k, const void* v, const void* o, const float* lse,
                        const void* d_o, void* dq, void* dk, void* dv, void* ws, int B, int heads, int Lq, int Lk, int hd,
                        int ldq, int ldk, int ldv, int ldo, int causal, const int64_t* text, int ld_text, int pad_id,
                        float temperature, float drop_p, const uint32_t* seed, uint32_t site, void* stream);

/* Token embedding * sqrt(D) + interleaved sin/cos positional table (+dropout):
 * networks/EfficientSATRN.py:480-483 and :400-426.  ids int64 [B][ld_ids]; pe fp32 [max_len][D]. */
int satrn_embedding_fwd(int dtype, const int64_t* ids, int ld_ids, const float* table, const float* pe, void* out, int B,
                        int L, int D, int pos0, float drop_p, const uint32_t* seed, uint32_t site, void* stream);
int satrn_embedding_bwd(int dtype, const int64_t* ids, int ld_ids, const void* dout, float* dtable, int B, int L, int D,
                        float drop_p, const uint32_t* seed, uint32_t site, void* stream);

/* CrossEntropyLoss(ignore_index) on fp32 logits [B*T][V] against targets = &expected[0][1] (int64, row stride ld):
 * networks/EfficientSATRN.py:690-692 as used by train_modules/train_single_opt.py:82,86.
 * loss_out[0..2] = sum, count, mean.  dlogits [B*T][Vp] (compute dtype, zero padded) = d(mean loss)/d(logits).
 * lse_ws: B*T floats. */
int satrn_cross_entropy(int dtype, const float* logits, const int64_t* targets, int ld, int B, int T, int V, int Vp,
                        int pad_id, float* loss_out, float* lse_ws, void* dlogits, void* stream);

/* Per-step training metrics without host round trips (train_modules/train_single_opt.py:101-109 = id_to_string(do_eval=1)
 * of utils/utils.py:134-164 + word_error_rate / sentence_acc of utils/metrics.py:9-34 + the symbol counts), on token ids:
 * sequence int64 [B][ld_seq] (T predictions per sample), expected int64 [B][ld_exp] (L entries: <SOS> ... <EOS> <PAD>|-1...).
 * acc double [5] (device) += {sum over samples of Levenshtein/max(len), samples, exactly-equal samples, correct symbols,
 * non-pad symbols}; empty_id = id of the "" token (or -1).  Sequences up to 512 tokens. */
int satrn_step_metrics(const int64_t* sequence, int ld_seq, int T, const int64_t* expected, int ld_exp, int L, int B, int pad_id,
                       int sos_id, int eos_id, int empty_id, double* acc, void* stream);
/* Knowledge-distillation loss of train_modules/train_distillation.py:49-55 (loss_fn_kd), forward and gradient in one
 * pass: student / teacher fp32 logits [B][T][V], labels int64 [B][ld] (NOT ignored when PAD, as in the reference).
 * loss = alpha*temperature^2 * KLDiv_batchmean(log_softmax(s/temperature), softmax(t/temperature)) + (1-alpha)*CE(s, labels);
 * loss_out [1], dlogits fp32 [B][T][V] = d loss / d student. */
int satrn_kd_loss(const float* student, const float* teacher, const int64_t* labels, int ld, int B, int T, int V,
                  float temperature, float alpha, float* loss_out, float* dlogits, void* stream);
/* clip_grad_norm_(max_norm) + AdamW.step over flat fp32 buffers: train_modules/train_single_opt.py:95-98.
 * gnorm_sq: device float, ZERO on entry (receives sum of squares, reduced in a fixed order so that data-parallel
 * replicas stay bit-identical); scratch1024: 1024 device floats.  hyper (device, 9 floats):
 * lr, beta1, beta2, eps, weight_decay, max_norm, 1-beta1^t, 1-beta2^t, grad_scale. */
int satrn_clip_adamw(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n, float* gnorm_sq,
                     float* scratch1024, const float* hyper, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Model level: the whole nn.Module path behind one handle.
 * Replaces EfficientSATRN / LiteSATRN .forward (networks/EfficientSATRN.py:697-706,
 * networks/LiteSATRN.py:581-590), loss.backward() + clip + optimizer.step
 * (train_modules/train_single_opt.py:86-98) and the greedy branch (networks/EfficientSATRN.py:528-561).
 * ------------------------------------------------------------------------------------------------- */
typedef struct satrn_model satrn_model;

typedef struct satrn_config {
  int network; /* 0 LiteSATRN, 1 EfficientSATRN, 2 SwinTRN */
  int rgb;     /* FLAGS.data.rgb */
  int height, width;
  int enc_hidden, enc_filter, enc_heads, enc_layers;
  int dec_src, dec_hidden, dec_filter, dec_heads, dec_layers;
  int num_classes;
  int pad_id, sos_id;
  float dropout; /* FLAGS.dropout_rate */
  int dtype;
  /* network == 2, SwinTRN (networks/SWIN.py:1024-1031; the reference hard-codes img 384, patch 4, embed 128, depths 2/2/18/2,
   * heads 4/8/16/32, window 12, drop_path 0.5, 21841 head classes; dec_src = enc_hidden = 8 * embed).  Ignored otherwise. */
  int swin_embed, swin_depths[4], swin_heads[4], swin_window, swin_patch, swin_head_classes;
  float swin_drop_path;
} satrn_config;

satrn_model* satrn_model_create(const satrn_config* cfg);
void satrn_model_destroy(satrn_model* m);
/* state table = the reference's state_dict keys, in the engine's flat order */
int satrn_model_num_state(const satrn_model* m);
const char* satrn_model_state_name(const satrn_model* m, int i);
/* kind: 0 parameter (fp32 flat), 1 fp32 buffer (BN running stats), 2 int64 buffer (num_batches_tracked);
 * init: 0 xavier_normal, 1 conv default, 2 linear default, 3 bias default (fan_in), 4 ones, 5 zeros, 6 N(0,1) */
int satrn_model_state_info(const satrn_model* m, int i, int* kind, int* ndim, int64_t* shape4, int64_t* offset,
                           int* init, int* fan_in, int* fan_out);
int64_t satrn_model_flat_size(const satrn_model* m, int kind);
/* borrow flat buffers until the next bind / destroy (grads may be NULL for inference) */
int satrn_model_bind(satrn_model* m, float* params, float* grads, float* buf_f32, int64_t* buf_i64);
size_t satrn_model_workspace_bytes(satrn_model* m, int B, int L);
int satrn_model_set_workspace(satrn_model* m, void* ws, size_t bytes, void* stream);
int satrn_model_pack_weights(satrn_model* m, void* stream);
/* training-graph forward; logits fp32 [B][L-1][V]; record != 0 keeps the tape for a backward.
 * teacher_forced != 0: networks/EfficientSATRN.py:490-495 (expected[:, :-1] is the decoder input);
 * teacher_forced == 0: the autoregressive branch with gradients, :496-525 (expected only gives the length). */
int satrn_model_forward(satrn_model* m, const float* images, const int64_t* expected, int B, int L, int train, int record,
                        int teacher_forced, float* logits, void* stream);
/* backward from d(loss)/d(logits) (fp32 [B][L-1][V]); parameter grads ACCUMULATE into the bound grads */
int satrn_model_backward(satrn_model* m, const float* dlogits, void* stream);
/* fused CE on the last forward's logits + backward (loss readable with satrn_model_read_loss) */
int satrn_model_loss_backward(satrn_model* m, const int64_t* expected, int B, int L, void* stream);
/* forward + CE + backward + clip + AdamW + re-pack; hyper9 is a HOST array (see satrn_clip_adamw; entries 6,7
 * are filled in by the library).  use_graph != 0 replays one captured hipGraph per (B, L): images / expected must
 * then stay at the same device addresses from call to call.  phase: bit 0 = zero grads + forward + CE + backward,
 * bit 1 = clip + AdamW + re-pack (data-parallel callers all-reduce the flat gradient between the two).
 * phase = 16 + k (k = 0..3, in order, eager only): the same work as bit 0 cut into four backward segments -- k = 0 zeroes
 * the gradients, runs forward + CE and the decoder's backward; 1 = encoder transformer + positional encoding; 2 = last
 * backbone stage; 3 = the rest.  When call k returns, the flat-gradient range satrn_model_segment_range(k) is final on
 * `stream`, so its all-reduce can run (on another stream) while the following segments execute.  16 + k + 4*k_to runs
 * segments k..k_to in one call (one side-stream join at the end); the data-parallel driver uses 16 + 0 + 4*2 (segments
 * 0-2: 74 % of the parameters) followed by 16 + 3.
 * + 32 (any of the above, eager): module.eval() semantics WITH gradients -- BatchNorm uses (and does not update) its running
 * statistics, dropout is off; every sample is then independent of the rest of the batch, which is the mode the data-parallel
 * equivalence test runs in (N ranks' averaged gradient == one rank's gradient on the concatenated batch).
 * + 64 (any of the above, eager): the NON-teacher-forced training branch, networks/EfficientSATRN.py:496-525 (the decoder feeds on its
 * own argmax, `expected` only gives the length; gradients flow through every step) -- the branch the reference's per-batch coin
 * (`random.random() < teacher_forcing_ratio`, :489) takes on 20-70 % of the training batches of the shipped schedule. */
int satrn_model_train_step(satrn_model* m, const float* images, const int64_t* expected, int B, int L,
                           const float* hyper9, int use_graph, int phase, void* stream);
/* The dual-optimizer iteration of train_modules/train_dual_opt.py:87-113: as satrn_model_train_step (eager), but phase
 * bit 1 clips the encoder.* and the decoder.* gradients SEPARATELY (two clip_grad_norm_ calls there) and steps each group
 * with its own hyper9 (own learning rate / weight decay; Adam == entry 4 set to 0).  The two squared gradient norms are
 * readable with satrn_model_read_grad_norms (host float[2]: encoder, decoder; synchronises). */
int satrn_model_train_step_dual(satrn_model* m, const float* images, const int64_t* expected, int B, int L,
                                const float* hyper9_enc, const float* hyper9_dec, int phase, void* stream);
int satrn_model_read_grad_norms(satrn_model* m, float* out2_host, void* stream);
/* argmax over the vocabulary of the last forward's / train step's logits -> ids int64 [B][L-1] (device): the `sequence`
 * of train_modules/train_single_opt.py:82-84 for the per-step metrics, without materialising logits for the caller.
 * Valid until the next call that runs the model (any forward, decode or train step). */
int satrn_model_last_sequence(satrn_model* m, int64_t* ids, int B, int L, void* stream);
int satrn_model_segment_range(satrn_model* m, int seg, int64_t* lo, int64_t* hi); /* [lo, hi) in flat fp32 elements */
int satrn_model_read_loss(satrn_model* m, float* out4_host, void* stream); /* sum, count, mean, gnorm^2; syncs */
int satrn_model_encode(satrn_model* m, const float* images, int B, float* src_out, void* stream);
/* KV-cached greedy decode; images may be NULL when src (fp32 [B][N][dec_src]) is given.
 * logits fp32 [B][steps][V]; ids int64 [B][steps].  use_graph != 0 captures the whole decode (encoder + all steps)
 * once per (B, steps, buffer addresses) and replays it: the four buffers must then stay at the same addresses. */
int satrn_model_greedy(satrn_model* m, const float* images, const float* src, int B, int steps, float* logits,
                       int64_t* ids, int use_graph, void* stream);
/* Forced replay of the same decode (the step loop of networks/EfficientSATRN.py:528-561 with `target` supplied instead of
 * `torch.argmax(o[:, -1, :])` at :557): the token fed to step t + 1 of image b is forced_ids[b][t] (int64 [B][steps], device);
 * logits are every step's outputs and ids their argmax, as in satrn_model_greedy.  With a reference decode's ids this compares
 * EVERY step's logits of a kernel against that reference, not only the prefix before the first near-tie flips a token. */
int satrn_model_greedy_forced(satrn_model* m, const float* images, const float* src, int B, int steps,
                              const int64_t* forced_ids, float* logits, int64_t* ids, void* stream);
/* Which kernel produced the last greedy result: 0 none yet, 1 the role pipeline (bf16, D = 256, F = 1024, B <= 112: one persistent
 * workgroup per decoder role), 2 the one-workgroup-per-image kernel, 3 step-wise launches.  *giveups (may be NULL) = pipelines of
 * this model that ran into a bounded wait and were re-run on path 2 (the pipeline is then disabled for the process; with
 * SATRN_PIPE_STRICT in the environment the decode fails with -7 instead).  satrn_model_decode_note: why path 1 was not taken. */
int satrn_model_last_decode_path(satrn_model* m, int* giveups);
const char* satrn_model_decode_note(satrn_model* m);
/* DecodingManager on the device (postprocessing/postprocessing.py:180-388).  rules: int32 [V + 8] in device memory, the
 * manager's rule lists compiled by satrn_amd.decoding.compile_rules: per token (flag bits: 1 next must be "_", 2 next must
 * be "{", 4 next cannot be "_", 8 next cannot be "{", 16 cannot follow <SOS>) | (run-length limit << 8), then the ids of
 * <SOS>, <EOS>, "", "_", "{", "}" and two reserved words.  state: int32 [B][4] = last token, run length, #"{", #"}".
 * satrn_sift = DecodingManager.sift (:189-246): probs = softmax(x) with the blacklisted tokens zeroed, targets = argmax,
 * state advanced (MemoryNode.record, :317-335).  satrn_sift_reset = DecodingManager.reset (:248-255).
 * satrn_model_greedy_rules = the greedy decode of satrn_model_greedy with the manager inside the decode kernel
 * (networks/EfficientSATRN.py:536-554): `probs` receives the masked probabilities [B][steps][V], not logits. */
int satrn_sift(const float* x, int ld, int32_t* state, const int32_t* rules, int B, int V, int64_t* targets, float* probs,
               int ldp, void* stream);
int satrn_sift_reset(int32_t* state, int B, int sos_id, void* stream);
int satrn_model_greedy_rules(satrn_model* m, const float* images, const float* src, int B, int steps, const int32_t* rules,
                             float* probs, int64_t* ids, void* stream);
/* EfficientSATRN.beam_search (networks/EfficientSATRN.py:708-867; caller postprocessing/decoding.py:42-48) for topk = 1:
 * per image a best-first search over a priority queue of (score = -sum(log p)/len, node), stopping at the first popped
 * <EOS> node or after max_sequence-1 expansions of beam_width children each; the whole search of every image runs in one
 * launch (priority queue, decoder steps with the ancestors' KV rows, log-softmax, top-k, back-trace; no host sync).
 * sequences: int64 [B][max_sequence] in device memory -- the utterance INCLUDING <SOS> (as the reference returns it),
 * padded with pad_id.  beam_width 1..16, max_sequence 1..500. */
int satrn_model_beam_search(satrn_model* m, const float* images, int B, int beam_width, int max_sequence, int eos_id, int pad_id,
                            int64_t* sequences, void* stream);
/* Step-wise decoding session = EfficientSATRN_decoder.step_forward / reset_status (networks/EfficientSATRN.py:932-952),
 * the interface the ensemble driver uses (utils/ensemble_utils.py:84-96: softmax-average the models' step logits, pick the
 * next token outside the model).  begin: src fp32 [B][N][dec_src] (an encoder output, satrn_model_encode) -> cross-
 * attention K/V of every layer + empty self-attention caches, step index 0.  step: target int64 [B] (this step's input
 * tokens; <SOS> first) -> logits fp32 [B][V]; at most max_steps (<= 500) steps.  The session lives in the workspace:
 * any other call on the same model ends it (the next satrn_model_step then fails). */
int satrn_model_step_begin(satrn_model* m, const float* src, int B, int max_steps, void* stream);
int satrn_model_step(satrn_model* m, const int64_t* target, float* logits, void* stream);
/* One EAGER forward + CE + backward with every kernel launch bracketed by HIP events on `stream`; writes a JSON array
 * [{"kernel", "launches", "ms", "flops", "bytes"}...] (per kernel family, algorithmic flops / bytes) to json_out. */
int satrn_model_profile_step(satrn_model* m, const float* images, const int64_t* expected, int B, int L, char* json_out,
                             size_t cap, void* stream);
/* Diagnostics: stage-boundary probes.  With probes enabled, a forward remembers the activation at the end of the stem, of every backbone
 * stage, of the backbone, of the positional encoding, of every encoder / decoder layer (satrn_model_probe_count / _info: name, rows,
 * columns; row-major [rows][cols], NHWC for feature maps).  _read casts an activation to fp32 into a caller buffer; a buffer registered
 * with _set_grad BEFORE the backward receives the gradient wrt that activation (fp32).  tools/bf16_grad_error.py uses this to locate where
 * the bf16 mode's error against the f32 mode enters. */
int satrn_model_probe_enable(satrn_model* m, int on);
int satrn_model_probe_count(satrn_model* m);
int satrn_model_probe_info(satrn_model* m, int i, const char** name, int64_t* rows, int* cols);
int satrn_model_probe_read(satrn_model* m, int i, float* out_f32, void* stream);
int satrn_model_probe_set_grad(satrn_model* m, int i, float* grad_out_f32);
/* Evaluation-time image transform for a batch of variable-size images in one launch (image half):
 * data/dataset.py:76-81 ([h / w > 2: rotate(90, expand=True)]) + data/augmentations.py:28-44 (A.Resize(H, W) = cv2 INTER_LINEAR
 * on uint8, A.Normalize(mean, std, max_pixel_value = 255), ToTensorV2).  descs: DEVICE array of B records
 * { const uint8_t* pixels (device, HWC, C interleaved); int32 h, w, stride_bytes, pad } (24 bytes each); out fp32 [B][C][H][W];
 * mean3 / std3: HOST float[3] (the first C entries are used).  C is 1 or 3. */
int satrn_image_preprocess(const void* descs, int B, int C, int H, int W, float* out, const float* mean3, const float* std3,
                           void* stream);

/* Optimizer state of the fused step (torch.optim.AdamW's exp_avg / exp_avg_sq / step, which the reference checkpoints:
 * train_modules/train_single_opt.py:497-512 `optimizer.state_dict()`).  exp_avg / exp_avg_sq are caller-owned flat fp32
 * DEVICE buffers of satrn_model_flat_size(m, 0) elements, borrowed until the next bind / destroy -- they are NOT part of the
 * workspace, so replacing the workspace (satrn_model_set_workspace with a larger one for a longer batch) neither clears
 * the moments nor restarts the bias correction; it also carries the dropout RNG state over from the old workspace, which
 * must stay allocated until that call returns.  satrn_model_rng_state reads (set == 0) or writes the RNG word. */
int satrn_model_bind_optimizer(satrn_model* m, float* exp_avg, float* exp_avg_sq);
/* Device error word (read and cleared; synchronises `stream`): bit 0 = a decoder-input token id outside the embedding table,
 * bit 1 = a loss target outside [0, V) other than ignore_index, bit 2 = a hand-off of satrn_batchnorm_act_se_fwd timed out.  nn.Embedding / nn.CrossEntropyLoss raise on such input
 * (the reference's loader pads `expected` with -1, data/loader.py:12-16, rewritten to <PAD> at
 * train_modules/train_single_opt.py:78); the kernels skip the element and set the bit instead of reading out of bounds.
 * satrn_model_read_loss checks the word too and fails with -6. */
int satrn_device_error(void* stream);
/* Diagnostics: how many launches took each kernel ROUTE since the last reset (host-side counters, no synchronisation).  Tests use it
 * to assert that a shape really ran on the kernel they mean to cover (the large-shape routes are chosen by size thresholds inside the
 * launchers).  out[i], i < n: 0 persistent GEMM (dense), 1 persistent GEMM (3x3 convolution / data gradient), 2 persistent weight
 * gradient, 3 tile GEMM (gemm_kernel family incl. halo convolution and skinny), 4 tile weight gradient, 5 BatchNorm + squeeze-and-excite
 * in one launch, 6 MBConv block front in one launch (expand .. squeeze-and-excite), 7 MBConv backward (projection data gradient + squeeze-and-excite) in
 * one launch, 8 row-streaming kernel for tall, thin dense products, 9 autoregressive training branch as one forward launch
 * (kernels_ar.hip); the rest 0.
 * reset != 0 clears them after the read.  Returns the number of defined routes. */
int satrn_route_counts(long long* out, int n, int reset);
float* satrn_model_adam_state(satrn_model* m, int which /*0 exp_avg, 1 exp_avg_sq*/);
int satrn_model_set_step(satrn_model* m, long t);
long satrn_model_get_step(satrn_model* m);
int satrn_model_rng_state(satrn_model* m, uint32_t* seed_io_host, int set, void* stream);

#ifdef __cplusplus
}
#endif
#endif
