// Host-side launch API of the SATRN gfx950 kernels (internal; the public C-ABI is include/satrn_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

enum { DT_F32 = 0, DT_BF16 = 1 };

// Deterministic reductions (the f32 parity mode): every cross-workgroup float reduction that normally ends in global float
// atomics (BatchNorm statistics in GEMM epilogues, column reductions, split-M weight gradients, bias / LayerNorm / embedding
// gradients, the loss sum) instead stores per-workgroup partials into a scratch slab and a second small launch folds them in
// a FIXED order.  Results are then bit-identical from run to run and independent of how the backward is cut into
// segments / streams.  The engine owns two slabs (main chain, weight-gradient side stream) and switches the mode on for
// dtype f32; operator-level C-ABI calls without an engine keep the atomic forms.
struct DetCtx { int on = 0; float* scratch[2] = {nullptr, nullptr}; size_t cap = 0 /*floats per slab*/; hipStream_t side = nullptr; };
extern DetCtx g_det;
// workgroups a dense weight-gradient launch aims for on the side stream (0 = the default, 96); set per engine call like g_det:
// SwinTRN's products are 10-40x larger than EfficientSATRN's and its chain leaves more of the chip free (160: 31.8 -> 30.7 ms/step)
extern int g_wgrad_dense_blocks;
// smallest dense weight gradient (GFLOP) the persistent kernel takes; set per engine call: 1.0 for SwinTRN (16.25 -> 16.05-16.12 ms per step
// against 2.0), 2.0 otherwise (EfficientSATRN: 10.40 -> 10.51-10.56 ms at 1.0)
extern float g_wgrad_big_min_gflop;
// Partial-tile slabs of the persistent weight-gradient kernel (kernels_gemm_big.hip): with a slab its items store their fp32 partial
// tiles [slice][N][K] with plain stores and a fold launch adds the slices in order -- instead of items x 16 K float atomics, which
// inside a training step (gradient buffers cold) cost as much as the product itself.  Two slabs like g_det (chain, side stream);
// the engine points them into its workspace per call, operator-level C-ABI calls keep the atomic form (cap = 0).
struct WgPartCtx { float* scratch[2] = {nullptr, nullptr}; size_t cap = 0 /*floats per slab*/; hipStream_t side = nullptr; };
extern WgPartCtx g_wgpart;
// mailbox for kernels whose workgroups hand small vectors to each other inside ONE launch (launch_bn_pool_se, launch_se_bwd_wide): `images`
// x 1600 8-byte words that start zeroed and are only written by those kernels, each launch with its own tag (se_next_tag).  Set per
// engine call (the model's workspace); null for operator calls that bring none.
struct SeBoxCtx { unsigned long long* box = nullptr; int images = 0; bool bwd = false /*the squeeze-and-excite backward may use it too*/; };
extern SeBoxCtx g_sebox;
// mailbox of the MBConv block kernels (kernels_mbconv.hip): `words` 8-byte words that start zeroed and that only those kernels write (every
// launch with its own tag); holds the per-(slab, image) BatchNorm sums of up to three exchanges: 3 x (C / 64) x B x 128 words.
struct MbBoxCtx { unsigned long long* box = nullptr; size_t words = 0; int images = 0; };
extern MbBoxCtx g_mbbox;
unsigned se_next_tag();   // (atomic: host threads driving several models never share a tag)
bool se_box_usable(hipStream_t s);   // false while `s` is being captured into a hipGraph
// Workgroups of `kernel` (threads per workgroup, dynamic LDS bytes) the device holds at ONCE: the upper bound for the grid of a launch
// whose workgroups wait for each other (a workgroup the dispatcher cannot place never arrives, and the resident ones spin until their
// timeout).  From hipOccupancyMaxActiveBlocksPerMultiprocessor for that very instantiation (cached), one workgroup per CU less than the
// API's answer when that answer is register-file-independent (>= 7: the hardware admits fewer than the API says at high SGPR counts,
// MI355X_MICROARCH.md "Residency and cooperative launch"); 0 when the query fails.
long resident_capacity(const void* kernel, int threads, size_t lds_bytes);
// launches per kernel route (satrn_route_counts, include/satrn_hip.h): host-side diagnostics for tests
enum { RT_GEMM_BIG = 0, RT_GEMM_BIG_CONV = 1, RT_WGRAD_BIG = 2, RT_GEMM_TILE = 3, RT_WGRAD_TILE = 4, RT_BN_POOL_SE = 5, RT_MBCONV_FWD = 6, RT_MBCONV_BWD = 7, RT_GEMM_TALL = 8, RT_AR_FUSED = 9, RT_COUNT = 10 };
extern long long g_route[RT_COUNT];
unsigned* device_error_word();   // device address of the error word (bit 2: a mailbox wait timed out)
void launch_fold4(const float* part, int nrep, long stride, long n, float* out, hipStream_t s);   // launch_fold with 16-byte accesses (n, stride % 4 == 0)
void det_overflow_warn(size_t need_floats);
// ---- run-time switches: FOUR environment variables (round 4; the library used to read ~70 SATRN_* names) -------------------------------
//   SATRN_OFF    = comma-separated features whose current form is switched OFF, i.e. the form it replaced runs (A/B levers of tests and
//                  tools; re-read at every C-ABI call), e.g. SATRN_OFF=mbconv_front,gemm_g2
//   SATRN_KNOBS  = name=value,...  tuning knobs, opt-in forms and the tri-state routes (gemm_big / wgrad_big / conv_big / gemm_tall = 0 off,
//                  1 by size, 2 every shape that fits)
//   SATRN_PROF   = comma-separated diagnostics: stage, host, join, shapes, pipe, dec, mb
//   SATRN_TIMING = name[=value],...  timing experiments that SKIP work (results are WRONG; announced once on stderr): skip_wgrad,
//                  no_stats_atomics, big_dbg, a2_dbg, ea_dbg
// (modes keep their own names: SATRN_DETERMINISTIC, SATRN_NONDET, SATRN_PIPE_STRICT)
void sw_refresh();   // re-read the four variables (every C-ABI entry point calls it; launches consult the snapshot)
bool sw_off(const char* name);
bool sw_knob_set(const char* name);
long sw_knob(const char* name, long dflt);
double sw_knobf(const char* name, double dflt);
const char* sw_knob_str(const char* name);   // pointer to the value inside the environment string (ends at ',' / ' ' / 0), or null
bool sw_prof(const char* name);
int sw_timing(const char* name);   // 0 = not set; value (1 if none given) otherwise
static inline float* det_scratch(hipStream_t s, size_t need_floats) {
  if (!g_det.on) return nullptr;
  if (need_floats > g_det.cap) { det_overflow_warn(need_floats); return nullptr; }
  return g_det.scratch[(g_det.side && s == g_det.side) ? 1 : 0];
}
// out[i] += sum_{r < nrep} part[r * stride + i]  (r ascending), i < n
void launch_fold(const float* part, int nrep, long stride, long n, float* out, hipStream_t s);
enum { AM_DENSE = 0, AM_CONV = 1, AM_DGRAD = 2 };
// Cyclic shift + window partition as a row map (networks/SWIN.py:338-371): token (b, y, x) of a [B][H][W] map lives at window row
// ((b * nWh + wy) * nWw + wx) * ws * ws + py * ws + px with (wy * ws + py, wx * ws + px) = ((y - shift) mod H, (x - shift) mod W).
// ws == 0: identity.  Kernels that take one read or write their WINDOW-ordered operand through it, which removes the separate
// permutation passes around the attention (launch_window_perm: four per block and step).
struct RowMap { int H = 0, W = 0, ws = 0, shift = 0; };
// Residual add in front of a LayerNorm, folded into it (SwinTRN: x = shortcut + DropPath(branch), then norm(x); networks/SWIN.py:283-300):
// forward  sum[r] = a[r] + scale(sample of r) * b[bmap(r)], stored to sum_out, and normalised;
// backward d sum = LayerNorm backward + gsum (the gradient other consumers of the sum left there), then a's gradient (+)= d sum and
//          b's gradient [bmap(r)] = scale * d sum.
// scale = stochastic-depth keep / (1 - p) per SAMPLE (rows_per_sample rows each), 1 when drop_p == 0.  Inactive when sum_out / on == 0.
struct LnAdd {
  int on = 0;
  void* sum_out = nullptr;        // forward: where the sum goes
  const void* gsum = nullptr;     // backward: incoming gradient of the sum (may be null)
  RowMap bmap;                    // b (and its gradient) are in window order
  float drop_p = 0.f; const uint32_t* seed = nullptr; uint32_t site = 0; long rows_per_sample = 1;
};
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SILU = 2, ACT_SIGMOID = 3, ACT_GELU = 4 /*exact erf form, nn.GELU (networks/SWIN.py:29)*/,
       ACT_DFACTOR = 5 /*backward only: the stored tensor already IS act'(u) (GemmP::pre_grad)*/ };

// C[M,N] = act(gatherA[M,K] * Bw[N,K]^T + bias) (+dropout) (+= if beta)
struct GemmP {
  const void* A; const void* Bw; void* C; const float* bias;
  int M, N, K, lda, ldc;
  // conv geometry: source tensor [Bn, H, W, Ci] (AM_CONV: the input; AM_DGRAD: dY), output rows = Bn*OH*OW
  int H, W, Ci, OH, OW, KW, stride, pt, pl;
  int act, beta, out_f32;
  float drop_p; const uint32_t* seed; uint32_t site;
  float* stats_part;  // deterministic mode (set by launch_gemm): per-row-tile partial sums [slot][2][N], folded into stats afterwards
  float* stats;  // optional [stats_rep][2N] (zeroed): per-column sum / sum of squares of the output (for the BatchNorm that follows)
  int stats_rep; // replicas (>= 1) the row tiles spread their atomics over; the consumer sums them
  // optional (dgrad of a BatchNorm output, N == the BN's C): stats become the BN backward's column sums
  // [sum g, sum g*xhat] with g = dz_total*act'(y*scale+shift), so the separate reduction pass disappears
  const void* bnb_y; const float* bnb_ss; const float* bnb_mr; int bnb_act;
  // optional inference epilogue (eval-mode BatchNorm folded in): out = act(acc * escale[col] + eshift[col]) + eres[row][col]
  // (eres: residual in the compute dtype with row stride ldc, may be null)
  const float* escale; const float* eshift; const void* eres;
  // optional (dense products): pre_out != null -> the value BEFORE the activation is stored there (rounded to the compute dtype) and C
  // gets act(that rounded value): the product and the activation pass behind it in one launch, both tensors kept for the backward.
  // bact_u != null -> the output is multiplied by act'(bact_u[row][col]) (bact = its kind): a data gradient that arrives at an
  // activation's output leaves as the gradient of its input (the separate act-backward pass disappears).  Row stride of both = ldc.
  void* pre_out; const void* bact_u; int bact;
  // pre_grad: pre_out receives act'(u) instead of u -- the backward needs u only through that factor, and the forward epilogue has the
  // erf / exp of it in hand already (GELU: the dgrad epilogue's derivative pass, 55 us of VALU work on four waves per CU at
  // [9216][1536], becomes one multiply); the consumer then passes bact = ACT_DFACTOR
  int pre_grad;
  int dbg_no_stats_atomics;   // timing experiment (SATRN_TIMING_NO_STATS_ATOMICS; wrong statistics)
  int no_stage_y;             // A/B (SATRN_GEMM_NO_STAGE_Y, set by launch_gemm): the BatchNorm-backward operand read element-wise from global
  // AM_DGRAD with stride 2 (set by launch_gemm): rows are dealt to the workgroups by PARITY CLASS of the output pixel ((oy + pt) & 1,
  // (ox + pl) & 1) -- a class meets only 4 / 2 / 2 / 1 of the nine taps, so a workgroup's k loop visits those taps only (2.25 on
  // average instead of 9 of which 6.75 staged zeros).  Needs OH, OW even and M / 4 a multiple of the tile height.
  int dgrad_classes;
  float bact_scale;   // times this (ReLU + dropout: bact_u is the stored OUTPUT, whose zeros cover both, and 1/(1-p) the kept ones' scale); 0 = 1
};
// eval-mode BatchNorm scale / shift of every BatchNorm of a model in ONE launch: out[0..C) = w * rsqrt(rv + eps),
// out[C..2C) = b - rm * scale
struct BnEvalDesc { const float* w; const float* b; const float* rm; const float* rv; float* out; float eps; int C; };
void launch_bn_eval_prepare(const BnEvalDesc* descs_dev, int n, hipStream_t s);
void launch_gemm(int dt, int amode, const GemmP& p, hipStream_t s);
// large dense bf16 products on the persistent 8-wave direct-to-LDS kernel (kernels_gemm_big.hip); false = shape / epilogue not taken
bool gemm_big_launch(const GemmP& p, hipStream_t s);
// tall, thin dense bf16 products (M >= 16 384 rows, N and K <= 256: the 1x1 projections of the fused-MBConv stages and their data gradients) on the
// row-streaming kernel (kernels_gemm_tall.hip); false = shape / epilogue not taken
bool gemm_tall_launch(const GemmP& p, hipStream_t s);
bool gemm_big_conv_launch(int amode, const GemmP& p, hipStream_t s);   // 3x3 stride-1 'same' convolution / its data gradient as a shifted GEMM

// dW[n][k] (+)= sum_m dY[m][n] * gatherA[m][k]
struct WgradP {
  const void* dY; const void* A; void* dW;
  int M, N, K, ldy, lda;
  int H, W, Ci, OH, OW, KW, stride, pt, pl;  // AM_CONV: A is the conv input [Bn,H,W,Ci], rows m = Bn*OH*OW
  int conv;        // 0 dense, 1 conv (im2col gather of A; dW written in [N][Ci][KH][KW] torch layout)
  int conv_packed_out;  // conv: write dW as [N][taps][Ci] (contiguous atomics) for launch_conv_grad_unpack
  int out_t;       // 0: fp32 atomicAdd into dW (zeroed by caller); 1: store as T (batched attention use)
  int out_accum;   // out_t == 1: add to the existing values (K/V shared by several attention calls)
  int full_grid;   // 1: size the grid to fill the chip (no concurrent data-gradient chain to stay out of the way of)
  int nbatch, nb_inner;                    // batched: z -> (z / nb_inner, z % nb_inner)
  long sY_o, sY_i, sA_o, sA_i, sW_o, sW_i;  // element strides per outer/inner batch index
  int ldw;                                 // out_t==1: row stride of dW
  float* det_part;                         // deterministic mode (set by launch_wgrad): [split][N][K] partial slabs
  int launch_order;                        // 1: (tile, slice) in launch order instead of slice-major per XCD (A/B: SATRN_WGRAD_LAUNCH_ORDER, set by launch_wgrad)
  int dbg_no_atomics;                      // timing experiment (SATRN_TIMING=wgrad_no_atomics)
  float* dbias;                            // optional (dense, !out_t): dbias[n] += sum_m dY[m][n], accumulated by the k-tile-0 workgroups from the dY
                                           // chunks they stage anyway (was a separate launch_colsum pass over dY)
};
void launch_wgrad(int dt, const WgradP& p, hipStream_t s);
// large dense bf16 weight gradients on the persistent direct-to-LDS kernel (kernels_gemm_big.hip); false = not taken
bool wgrad_big_launch(const WgradP& p, hipStream_t s);
void launch_conv_grad_unpack(const float* tmp /*[N][taps][Ci]*/, float* dw /*[N][Ci][taps] +=*/, int N, int Ci, int taps, hipStream_t s);

// ---- attention ---------------------------------------------------------------------------
struct AttnP {
  const void* Q; const void* K; const void* V; void* O; float* lse;   // lse [B,H,Lq]
  const void* dO; void* dQ; void* dS; void* Pd;                       // backward (mode 1): dS,Pd [B,H,Lq,LkP] as T
  const int64_t* text; int ld_text;  // token ids [B][ld_text] for the pad mask (nullptr = none)
  int B, H, Lq, Lk, hd;
  int ldq, ldk, ldv, ldo;  // row strides (elements)
  int causal, pad_id;
  float inv_temp, drop_p; const uint32_t* seed; uint32_t site;
  // window attention (networks/SWIN.py:163-183): scores += bias[h][i][j] (relative position bias, shared by all windows) and
  // += wmask[b % nW][i][j] (0 / -100 shifted-window mask); both fp32 with row stride Lk, either may be null
  const float* bias; const float* wmask; int nW;
  // the same two terms computed INSIDE the kernel (no [H][N][N] / [nW][N][N] tensors are read): rel_table = the
  // relative_position_bias_table parameter [(2*rel_ws-1)^2][H] (fp32), token i of a window sits at (i / rel_ws, i % rel_ws);
  // labels [nW][Lq] = region id of every token of every window of the shifted map (mask = -100 where two ids differ; null = none)
  const float* rel_table; int rel_ws; const unsigned char* labels;
  // fused backward (kernels_attn2.hip, launch_attn2_bwd): dK / dV are finished inside the workgroup (same strides as K / V; kv_accum: add to the
  // existing values), drel (optional, fp32 [(2 rel_ws - 1)^2][H]) receives the relative-position-table gradient with atomics
  void* dK; void* dV; int kv_accum; float* drel; int dbg;
  int q_pos0;              // step mode: absolute position of query row 0 (causal uses q_pos0 + i)
  long sq_b, sk_b, sv_b, so_b;  // batch strides (elements) of Q / K / V / O(dO,dQ use sq_b/so_b)
};
void launch_attn(int dt, int mode, const AttnP& p, hipStream_t s);
size_t attn_lkp(int Lk);  // padded key count used for dS/Pd workspaces
// register-resident attention for short sequences (kernels_attn2.hip; bf16, Lq <= 144, Lk <= 160, head_dim 32 / 64); false = not taken
bool attn2_ok(int dt, const AttnP& p);
bool launch_attn2_fwd(const AttnP& p, hipStream_t s);
bool launch_attn2_bwd(const AttnP& p, hipStream_t s);   // needs Q, K, V, O, lse, dO -> dQ, dK, dV (+ drel)

// ---- the encoder's self-attention region as one launch (kernels_encattn.hip): LayerNorm -> q|k|v -> attention -> output projection partials
struct EncAttnP {
  const void* x;                                  // [B*L][D] layer input (bf16)
  const float* ln_w; const float* ln_b;           // the layer's shared LayerNorm
  const void* wqkv; const float* bqkv;            // fused projection [3D][D] (compute copy), bias [3D]
  const void* wo;                                 // output projection [D][D]
  void* y1; float* mr;                            // saved: LayerNorm output [B*L][D], mean | rstd [2 * B*L]
  void* qkv; void* att; float* lse;               // saved: q|k|v [B*L][3D], attention output [B*L][D], log-sum-exp [B][H][L]
  void* parts;                                    // [H/2][B*L][D] partial output projections (bias / dropout / sum: launch_layernorm_parts)
  int B, L, D, H, LkP;
  float inv_temp, drop_p; const uint32_t* seed; uint32_t site;   // attention-probability dropout (same counter hash as attn_kernel)
  int dbg;
};
bool enc_attn_fused_ok(int dt, int L, int D, int H);
bool launch_enc_attn_fwd(const EncAttnP& p, hipStream_t s);
void launch_layernorm_parts(const void* parts, int nparts, long pstride, const float* abias, float drop_p, const uint32_t* seed, uint32_t site, void* a_out,
                            const void* b, const float* w, const float* bias, void* out, float* mr, long R, int C, hipStream_t s);

// ---- elementwise / reductions (all NHWC, C % (16/sizeof(T)) == 0 unless stated) ------------
void launch_colstats(int dt, const void* y, long M, int C, float* sums /*[2C], zeroed*/, hipStream_t s);
void launch_bn_finalize(const float* sums, long M, int C, const float* w, const float* b, float* rm, float* rv,
                        int64_t* nbt, float eps, float mom, int train, float* scale_shift /*[2C]*/,
                        float* mean_rstd /*[2C]*/, hipStream_t s);
// finalize folded in: sums != null -> batch statistics (running stats updated, momentum mom), else running statistics;
// writes scale/shift and mean/rstd for the backward
void launch_bn_act(int dt, const void* y, const float* sums, int sums_rep, const float* w, const float* b, float* rm,
                   float* rv, int64_t* nbt, float eps, float mom, float* scale_shift, float* mean_rstd, const void* res,
                   void* z, long M, int C, int act, hipStream_t s);
void launch_bn_bwd_reduce(int dt, const void* dz, const void* y, const float* scale_shift, const float* mean_rstd,
                          long M, int C, int act, float* red /*[2C] zeroed*/, hipStream_t s,
                          const void* se_gate = nullptr /*[B][C] T*/, const void* se_dpool = nullptr /*[B][C] T*/, int se_hw = 0);
void launch_bn_bwd_apply(int dt, const void* dz, const void* y, const float* scale_shift, const float* mean_rstd,
                         const float* w, const float* red, long M, int C, int act, void* dy, float* dw, float* db,
                         hipStream_t s, int red_rep = 1, const void* se_gate = nullptr, const void* se_dpool = nullptr, int se_hw = 0,
                         int eval_stats = 0 /*1: the forward used running statistics: no batch-mean terms in dy*/);
// se_gate/se_dpool/se_hw: the gradient fed to the two BatchNorm-backward passes is dz*gate[b][c] + dpool[b][c]/se_hw (the
// squeeze-and-excite backward wrt its input, b = row / se_hw) computed on the fly instead of materialised by se_bwd_x
void launch_stem_conv(int dt, const float* img, const float* w, void* y, int B, int Cin, int H, int W, int Co, int OH,
                      int OW, int stride, int pad, hipStream_t s);
void launch_stem_wgrad(int dt, const float* img, const void* dy, float* dw, int B, int Cin, int H, int W, int Co,
                       int OH, int OW, int stride, int pad, hipStream_t s);
void launch_dwconv(int dt, int mode /*0 fwd,1 dgrad*/, const void* x, const void* wp /*[9][C] as T*/, const float* bias,
                   void* y, int B, int H, int W, int C, int OH, int OW, int stride, int pt, int pl, int beta,
                   float* stats /*optional [2C] zeroed: column sums of y (mode 0)*/, hipStream_t s,
                   const float* escale = nullptr, const float* eshift = nullptr, int eact = 0 /*inference (mode 0): y = act(conv*escale[c] + eshift[c])*/);
// inference, stride 1, whole image x 64 channels per workgroup: out = act((dw3x3(x) + bias)*escale + eshift), pool[b][c] = sum over the
// image of out (optional; complete, no atomics).  false = shape not taken
// se != null: the squeeze-and-excite block behind it in the same launch -- out = act(...) * sigmoid(W2 silu(W1 mean + b1) + b2); false when the
// grid would not be resident at once (the image's workgroups exchange the hidden layer through se->box, see SeBoxCtx) or the shape is not taken
struct SeEvalArgs { unsigned long long* box; int box_images; const void* W1; const float* b1; const void* W2; const float* b2; int S; };
bool launch_dwconv_eval_img(int dt, const void* x, const void* wp, const float* dwbias, const float* escale, const float* eshift, int act, void* out,
                            float* pool /*[B][C] or null*/, int B, int H, int W, int C, hipStream_t s, const SeEvalArgs* se = nullptr);
void launch_image_pool(int dt, const void* x /*[B][HW][C]*/, float* pool /*[B][C] sums over HW*/, int B, int HW, int C, hipStream_t s);
void launch_dwconv_wgrad(int dt, const void* x, const void* dy, float* dw /*[C][9] torch layout*/, float* dbias,
                         float* scratch10C /*optional zeroed [10][C]: contiguous atomics + scatter*/, int B, int H, int W,
                         int C, int OH, int OW, int stride, int pt, int pl, hipStream_t s);
void launch_maxpool(int dt, int bwd, const void* x, const void* dy_or_null, void* out, int B, int H, int W, int C,
                    hipStream_t s);
void launch_pool_hw(int dt, const void* x, void* out /*[B,C] as T*/, int B, int HW, int C, hipStream_t s);
// BatchNorm(batch statistics)+activation that also accumulates poolsum[b][c] += sum_hw z (zeroed [B][C]); bn_act_pool_ok() says
// whether the shape / mode takes it (HW % 48 == 0, not the deterministic mode)
bool bn_act_pool_ok(long M, int C, int HW);
// data gradient of the stride-1 depthwise 3x3 (dz (+)= conv^T(dy)) AND the BatchNorm-backward column sums of z = act(bn(y)) in red
// (zeroed [2C]), one launch (bf16, whole image x 64 channels per workgroup); false = not taken, nothing launched
// a BatchNorm backward-apply pass handed to the kernel that consumes its result (launch_dwconv_bwd_bn): the operands of launch_bn_bwd_apply
struct BnBwdHold {
  bool armed = false;
  const void* dz = nullptr; const void* y = nullptr; const float* ss = nullptr; const float* mr = nullptr; const float* w = nullptr; const float* red = nullptr;
  long M = 0; int C = 0, act = 0; void* dy = nullptr; float* dwp = nullptr; float* dbp = nullptr;
  const void* se_gate = nullptr; const void* se_dpool = nullptr; int se_hw = 0; int rep = 1 /*replicas of red*/;
};
// y == nullptr: no BatchNorm sums (plain data gradient).  ap != nullptr: dy is NOT read -- it is first produced as the BatchNorm
// backward-apply result of *ap (written to ap->dy = dy for the weight-gradient pass, parameter gradients accumulated) and consumed
// from LDS: launch_bn_bwd_apply + the data gradient + the next BatchNorm's sums in one launch
// tail != nullptr (needs ap, y, beta == 0): the backward-apply pass of the BatchNorm in FRONT too -- the slab's workgroups exchange their
// shares of its column sums (g_mbbox) and tail->dy receives what launch_bn_bwd_apply(dz, y, red) would have written; dz and red are not
// written.  false when the shape / residency / mode does not allow it (the caller retries without tail).
struct BnBwdTail { void* dy = nullptr; const float* w = nullptr; float* dwp = nullptr; float* dbp = nullptr; };
bool launch_dwconv_bwd_bn(int dt, const void* dy, const void* wp, void* dz, int beta, const void* y, const float* ss, const float* mr, int act,
                          float* red, int B, int H, int W, int C, hipStream_t s, const BnBwdHold* ap = nullptr, const BnBwdTail* tail = nullptr);
bool dwconv_img_ok(int dt, int H, int W, int C);   // shape / dtype / mode test of the two image-tile depthwise kernels
// BatchNorm(batch statistics)+activation of y -> z AND the stride-1 depthwise 3x3 of z -> out with out's column sums in red (zeroed
// [2C]), one launch (bf16, whole image x 64 channels per workgroup); false = shape / mode not taken, nothing launched
bool launch_bn_dwconv(int dt, const void* y, const float* sums, int sums_rep, const float* w, const float* b, float* rm, float* rv, int64_t* nbt,
                      float eps, float mom, float* ss, float* mr, void* z, const void* wp, const float* dwbias, void* out, float* red, int B, int H,
                      int W, int C, int act, hipStream_t s);
void launch_bn_act_pool(int dt, const void* y, const float* sums, int sums_rep, const float* w, const float* b, float* rm, float* rv,
                        int64_t* nbt, float eps, float mom, float* ss, float* mr, void* z, float* poolsum, long M, int C, int HW, int act,
                        hipStream_t s);
void launch_se_scale(int dt, const void* x, const void* gate /*[B,C] T*/, void* out, int B, int HW, int C, hipStream_t s);
// dx (+)= dout*gate + dpool[b,c]/HW ; dgate[b,c] = sum_hw dout*x   (two kernels)
// data path of the squeeze-and-excite backward in two wide launches (dgate + dz2 + ds1 | du1 + dpooled); false = not taken
bool launch_se_bwd_wide(int dt, const void* dy, const void* x, const void* gate, const float* u1, const void* W1, const void* W2, float* dz2,
                        float* du1, float* ds1_zeroed /*[B][S]*/, void* dpooled, int B, int HW, int C, int S, hipStream_t s,
                        // optional: x is act(bn(bn_y)) and read by this op only -> x is recomputed from bn_y, and bn_red (zeroed [2C]) receives that
                        // BatchNorm's backward column sums (bn_P: [4][B][C] floats of scratch); launch_bn_bwd_reduce is then not needed
                        const void* bn_y = nullptr, const float* bn_ss = nullptr, const float* bn_mr = nullptr, int bn_act = 0, float* bn_P = nullptr,
                        float* bn_red = nullptr);
void launch_se_bwd_gate(int dt, const void* dout, const void* x, void* dgate /*[B,C] T*/, int B, int HW, int C, hipStream_t s);
void launch_se_bwd_x(int dt, const void* dout, const void* gate, const void* dpool /*[B,C] T or null*/, void* dx, int B,
                     int HW, int C, int beta, hipStream_t s);
void launch_bcast_add_hw(int dt, const void* dpool /*[B,C]*/, void* dx, int B, int HW, int C, float scale, hipStream_t s);
void launch_posenc2d(int dt, const void* x, const void* gate /*[B,2C] T*/, const float* hpos, const float* wpos, void* out,
                     int B, int H, int W, int C, hipStream_t s);
void launch_posenc2d_bwd(int dt, const void* dout, const float* hpos, const float* wpos, void* dgate /*[B,2C] T*/, int B,
                         int H, int W, int C, hipStream_t s);
void launch_layernorm(int dt, const void* a, const void* b_or_null, const float* w, const float* bias, void* out,
                      float* mean_rstd /*[2R]*/, long R, int C, float eps, float drop_p, const uint32_t* seed,
                      uint32_t site, hipStream_t s, RowMap out_map = RowMap(), LnAdd add = LnAdd());   // out_map: token row r is written at its window row
void launch_layernorm_bwd(int dt, const void* dout, const void* a, const void* b_or_null, const float* w,
                          const float* mean_rstd, void* da, void* db_or_null, int beta_a, int beta_b, float* dw,
                          float* dbias, long R, int C, float drop_p, const uint32_t* seed, uint32_t site, hipStream_t s,
                          float* part_ws = nullptr, RowMap dout_map = RowMap(), LnAdd add = LnAdd());   // dout_map: dout is in window order
// part_ws != null: [layernorm_bwd_blocks(R)][2][C] floats receive per-block (dw | dbias) partials instead of atomics; fold them with
// launch_layernorm_fold (any stream ordered after the backward kernel)
int layernorm_bwd_blocks(long R);
void launch_layernorm_fold(const float* part, int nblocks, int C, float* dw, float* dbias, hipStream_t s);
void launch_reshape_quirk(int dt, int inverse, const void* in, void* out, int B, int HW, int C, int beta, hipStream_t s);
void launch_embed(int dt, const int64_t* ids, const float* table, const float* pe, void* out, int B, int L, int ld_ids,
                  int D, int pos0, float drop_p, const uint32_t* seed, uint32_t site, hipStream_t s, int nrows = 0);
void launch_embed_bwd(int dt, const int64_t* ids, const void* dout, float* dtable, int B, int L, int ld_ids, int D,
                      float drop_p, const uint32_t* seed, uint32_t site, hipStream_t s, int nrows = 0);
// nrows > 0: ids outside [0, nrows) are skipped and flagged in the device error word instead of followed
unsigned device_error_read_clear(hipStream_t s);  // bit 0: embedding id out of range, bit 1: CE target out of range; synchronises
void launch_colsum(int dt, const void* x, long M, int C, int ld, float* out /*[C] +=*/, hipStream_t s);
void launch_act_bwd(int dt, const void* dz, const void* z_post, void* du, long n, int act, float drop_p, hipStream_t s);
void launch_dropout_bwd(int dt, const void* dz, void* du, long M, int N, float drop_p, const uint32_t* seed, uint32_t site,
                        hipStream_t s);
// loss_out [4]: sum, count, mean; lse_ws [B*T]; dlogits [B*T][Vp] written as dt_out (zero padded), scaled by *upstream
void launch_ce_full(int dt_out, const float* logits, const int64_t* tgt, int ld_tgt, int tgt_off, int B, int T, int V,
                    int Vp, int pad_id, float* loss_out, float* lse_ws, void* dlogits, const float* upstream,
                    hipStream_t s);
void launch_step_metrics(const int64_t* seq, int ld_seq, int T, const int64_t* expected, int ld_exp, int L, int B, int pad_id,
                         int sos_id, int eos_id, int empty_id, double* acc /*[5] +=*/, hipStream_t s);
void launch_kd_loss(const float* student, const float* teacher, const int64_t* labels, int ld_labels, int B, int T, int V,
                    float temperature, float alpha, float* loss_out /*[1]*/, float* dlogits /*[B*T][V]*/, hipStream_t s);
void launch_act_fwd(int dt, const void* u, void* z, long n, int act, hipStream_t s);
void launch_cast_pad(int dt_out, const float* in, void* out, long R, int C, int Cp, hipStream_t s);
void launch_pack_dense_ld(int dt, const float* w, void* fwd, void* bwd, int N, int K, int ldb, hipStream_t s);
int launch_attn_checked(int dt, int mode, const AttnP& p, hipStream_t s);
void launch_cast(int dt_in, int dt_out, const void* in, void* out, long n, hipStream_t s);
void launch_fill(void* p, int value_byte, size_t bytes, hipStream_t s);
void launch_set_scalars(float* dst, const float* src /*host, n <= 12*/, int n, hipStream_t s);   // by kernel argument, no copy engine
void launch_add(int dt, const void* a, const void* b, void* out, long n, hipStream_t s);
void launch_argmax(const float* logits, int64_t* ids, int R, int V, int ld_in, int ld_out, hipStream_t s);
void launch_seed_advance(uint32_t* seed, hipStream_t s);
void launch_copy_rows(int dt, const void* src, void* dst, int B, int n, int C, long src_bstride, long src_off,
                      long dst_bstride, long dst_off, int beta, hipStream_t s);
void launch_fill_i64(int64_t* p, int64_t v, long n, hipStream_t s);

// ---- weight packing + optimizer --------------------------------------------------------------
void launch_pack_dense(int dt, const float* w, void* fwd /*[N][K]*/, void* bwd /*[K][N]*/, int N, int K, hipStream_t s);
void launch_pack_conv(int dt, const float* w /*[Co][Ci][T]*/, void* fwd /*[Co][T][Ci]*/, void* bwd /*[Ci][T][Co]*/, int Co,
                      int Ci, int taps, hipStream_t s);
void launch_pack_dw(int dt, const float* w /*[C][9]*/, void* out /*[9][C]*/, int C, hipStream_t s);
// kind 0 dense (N,K,ldb), 1 conv3x3 (N=Co, K=Ci), 2 depthwise (N=C)
struct PackDesc { const float* src; void* fwd; void* bwd; long start; int kind, N, K, ldb; };
#define PACK_BLK 4096
void launch_pack_all(int dt, const PackDesc* d, const void* blk_table /*int2 (desc, chunk) per block*/, int nblk, hipStream_t s);
void launch_sumsq(const float* g, long n, float* out /*[1] +=*/, float* partial /*[1024] scratch*/, hipStream_t s);
void launch_adamw(float* p, const float* g, float* m, float* v, long n, const float* gnorm_sq, const float* hyper,
                  hipStream_t s);

// fused squeeze-and-excite (kernels_se.hip): W1 [S][C], W2 [C][S] are the fp32 masters
// W1 [S][C], W2 [C][S]: the packed compute-dtype copies (dt)
// squeeze-and-excite MLP + x*gate from pool SUMS (launch_bn_act_pool): grid (B, 4 channel groups); false = shape / dtype not
// taken (the caller then runs launch_se_fwd + launch_se_scale)
// BatchNorm (batch statistics) + activation + squeeze-and-excite (pool, MLP, x*gate) in one launch on the small maps; box: a persistent
// mailbox of (box_images * (1536 + 64)) 8-byte words that starts zeroed and is only ever touched by this kernel.  z may be null (the
// activated tensor is then not stored).  false = shape / mode not taken.  A wait that times out sets device error bit 2.
bool launch_bn_pool_se(int dt, const void* y, const float* sums, int sums_rep, const float* w, const float* b, float* rm, float* rv, int64_t* nbt,
                       float eps, float mom, float* ss, float* mr, void* z, const void* W1, const float* b1, const void* W2, const float* b2,
                       float* pooled, float* u1, float* s1, void* gate, void* out, unsigned long long* box, int box_images, int B, int HW, int C,
                       int S, int act, hipStream_t s);
// The front of an MBConv block in ONE launch (kernels_mbconv.hip): expand product -> BatchNorm (batch statistics) + SiLU -> depthwise 3x3 ->
// BatchNorm + SiLU -> squeeze-and-excite -> z3 = z2 * gate, for x [B][H][W][Cin] bf16 with H * W = 48 or 192 (the 4x12 / 8x24 maps).  Needs
// g_mbbox and g_sebox; mbconv_front_ok says beforehand whether a shape is taken (incl. the residency of the whole grid) -- a launch that
// follows a true answer cannot refuse.  Writes everything the separate kernels wrote (see the file header); z2 may be null.
bool mbconv_front_ok(int dt, int B, int H, int W, int Cin, int C, int S, hipStream_t s);
// xin != null (then x == null): the block input is the output of a BatchNorm (batch statistics from `sums`, `rep` replicas of [2 Cin]; no
// activation) plus an optional residual -- normalised while the kernel stages it and written to `out` for later readers (launch_bn_act's job)
struct MbXinArgs { const void* y; const void* res; const float* sums; int rep; const float* w; const float* b; float* rm; float* rv; int64_t* nbt;
                   float* ss; float* mr; float eps; void* out; };
bool launch_mbconv_front(int dt, const void* x, const MbXinArgs* xin, const void* W0, void* y1, const float* bn1_w, const float* bn1_b, float* bn1_rm, float* bn1_rv, int64_t* bn1_nbt,
                         float* bn1_ss, float* bn1_mr, float bn1_eps, void* z1, const void* wdw, void* y2, const float* bn2_w, const float* bn2_b, float* bn2_rm,
                         float* bn2_rv, int64_t* bn2_nbt, float* bn2_ss, float* bn2_mr, float bn2_eps, void* z2, const void* Wr, const float* br,
                         const void* We, const float* be, float* pooled, float* u1, float* s1, void* gate, void* z3, int B, int H, int W, int Cin, int C, int S,
                         float mom, hipStream_t s);
// Backward: the projection's data gradient dz3 = dy3 W1 and the squeeze-and-excite backward (incl. the sums BatchNorm 2's backward needs) in
// ONE launch (kernels_mbconv.hip); same shapes as the front.  Wb = the projection's backward pack [C][ldb]; red [2C] is accumulated into.
// Only the workgroups of one image wait for each other (g_sebox).  false = not taken.
// din != null (then dy3 == null): dy3 = the backward of the block-ending BatchNorm (batch statistics, no activation; column sums in `red`)
// applied to dz while the kernel stages it, written to dy_out for the weight gradient; dw / db (may be null) receive the BatchNorm's
// parameter gradients -- launch_bn_bwd_apply's job
struct MbDinArgs { const void* dz; const void* y; const float* ss; const float* mr; const float* w; const float* red; int rep; void* dy_out; float* dw; float* db; };
bool launch_mbconv_bwd_se(int dt, const void* dy3, const MbDinArgs* din, const void* Wb, int ldb, void* dz3, const void* y2, const float* ss2, const float* mr2, const void* gate,
                          const float* u1, const void* We, const void* Wr, float* dz2, float* ds1, float* du1, void* dpooled, float* red, int B, int H, int W,
                          int CN, int C, int S, hipStream_t s);
bool launch_se_mlp_scale(int dt, const void* x, const float* poolsum, const void* W1, const float* b1, const void* W2, const float* b2,
                         float* pooled, float* u1, float* s1, void* gate, void* y, int B, int HW, int C, int S, hipStream_t s);
void launch_se_fwd(int dt, const void* x, const void* W1, const float* b1, const void* W2, const float* b2, float* pooled,
                   float* u1, float* s1, void* gate, int B, int HW, int C, int S, hipStream_t s);
void launch_se_bwd(int dt, const void* dgate, const void* gate, const float* u1, const float* s1, const float* pooled,
                   const void* W1, const void* W2, float* dz2, float* du1, void* dpooled, float* dW1, float* db1, float* dW2,
                   float* db2, int B, int C, int S, hipStream_t s, int parts = 3 /*1: data path, 2: weight gradients*/);

// persistent greedy decoder (kernels_decode.hip): weights are the packed [N][K] compute copies, biases / LN fp32
struct DecLayerW {
  const void *wqkv, *wo, *wq2, *wo2, *w0, *w1;  // k-panel-major [K/32][N][32] (launch_repack_kpanel); k|v of the history = rows D..3D of wqkv
  const float *bqkv, *bo, *bq2, *bo2, *b0, *b1, *bkv, *ln1w, *ln1b, *ln2w, *ln2b, *ln3w, *ln3b;
  const void* crossKV;  // [B][Nsrc][2D]
  void* cache;          // [B][steps][2D] scratch
};
struct DecodeP {
  DecLayerW L[4];
  int nlayers;
  const float* embed; const float* pe; const void* wgen; const float* bgen;
  float* logits; int64_t* ids;
  int B, steps, D, F, V, H, Nsrc, sos;
  int dbg;  // timing ablation bits (SATRN_DEC_DBG), 0 in production
  const int32_t* rules;  // optional compiled DecodingManager rules [V + 8]: outputs become masked probabilities
  long long* prof;       // optional [16] cycle counters per phase family (SATRN_DEC_PROF, workgroup 0 only)
  // forced replay (parity tests, ensemble-style teacher forcing): when non-null the token fed to step t + 1 of image b is
  // forced[b * ld_forced + t] instead of the step's own argmax (logits / ids outputs are unchanged: ids stay the argmax)
  const int64_t* forced; int ld_forced;
};
int launch_decode_greedy(int dt, const DecodeP& p, hipStream_t s);
// pipelined weight-stationary greedy decoder (kernels_decode.hip): one persistent workgroup per ROLE (a slice of the decoder's
// weights resident in LDS), images flow through the roles over tagged 8-byte granule mailboxes
struct PipeRole { int type, layer, sub, img0, img1, istep /*serves images img0, img0 + istep, ... < img1*/, N, K, Ntot, row0, kp0; const void* w; const float* bias; const float* lnw; const float* lnb;
                  const void* w2; int N2, K2, Ntot2, row02, kp02; /*second LDS-resident matrix (feed-forward role: the K-slab of the output projection)*/
                  const float* pre_bias; const float* pre_lnw; const float* pre_lnb; /*non-null: recompute the previous layer's FFN combine + LayerNorm from its partials*/ };
struct PipeP {
  DecLayerW L[4];
  int nlayers;
  const float* embed; const float* pe;
  float* logits; int64_t* ids;
  int B, steps, D, F, V, H, Nsrc, sos;
  const PipeRole* roles; unsigned long long* mail; int* err; long long timeout_ticks;
  const int32_t* rules;  // optional DecodingManager table (sift.h): applied by the generator role, which keeps each of its images' memory
  long long* prof;  // optional [2 * nroles]: wall-clock ticks spent waiting / in total (SATRN_PIPE_PROF)
  const int64_t* forced; int ld_forced;   // see DecodeP
};
size_t decode_pipe_scratch_bytes(const DecodeP& p);
int launch_decode_pipe(int dt, const DecodeP& p, void* scratch, size_t scratch_bytes, hipStream_t s);  // 0 launched, -1 shape / device not supported
// why the last launch_decode_pipe call returned -1 ("" after a launch): shape, device too small, co-residency not provable, disabled
const char* decode_pipe_reason();
// a pipeline that timed out is not tried again in this process (every later decode would burn the same timeout): sticky
void decode_pipe_disable(const char* why);
int decode_pipe_error(void* scratch, hipStream_t s);  // weights in DecLayerW / wgen: k-panel-major copies
// training-time autoregressive branch (kernels_ar.hip): one workgroup per image, all T steps of a direction in one launch.  Slabs are
// [B*T][C] in the compute dtype, row b*T + t.  Weights: k-panel-major copies ([K/32][N][32], launch_repack_kpanel) of W and of W^T.
struct ArLayer {
  const void *wqkv, *wo, *wq2, *wo2, *w0, *w1;          // forward: [3D][D], [D][D], [D][D], [D][D], [F][D], [D][F]
  const void *wqkvT, *woT, *wq2T, *wo2T, *w0T, *w1T;    // the transposes (backward)
  const float *bqkv, *bo, *bq2, *bo2, *b0, *b1, *ln1w, *ln1b, *ln2w, *ln2b, *ln3w, *ln3b;
  float *dln1w, *dln1b, *dln2w, *dln2b, *dln3w, *dln3b; // LayerNorm parameter gradients (+=)
  const void* crossKV;   // [B][Nsrc][2D]
  void* cache;           // [B][T][2D]: k|v of the layer outputs (of the inputs while a step runs)
  void *q, *kvin, *att, *s1, *t1, *q2, *a2, *s2, *t2, *f0, *f1d;   // saved by the forward: widths D, 2D, D, D, D, D, D, D, D, F, D
  void *dqkvi, *dkvo, *dout, *dq2, *dout2, *df0, *df1;             // written by the backward: widths 3D, 2D, D, D, D, F, D
  float* dkvacc;         // [B][T][2D] f32, ZERO before the backward: gradient of the history entries
  float* dcross;         // [B][Nsrc][2D] f32, ZERO before the backward: gradient of crossKV
};
struct ArP {
  ArLayer L[4];
  ArLayer* Ltab;         // device memory, 4 entries: the launchers copy L[] there (the kernels index it with the runtime layer number)
  int nlayers;
  void* xs[5];           // xs[0] = embedding + PE, xs[l + 1] = output of layer l
  const float* embed; const float* pe; const void* wgen; const float* bgen;
  float* logits /*[B*T][V] f32*/; int64_t* ids /*[B][T] argmax*/; int64_t* in_ids /*[B][T] the token fed to each step*/;
  const void* dxtop /*[B*T][D] gradient of xs[nlayers]*/; void* dx0 /*[B*T][D] gradient of xs[0]*/;
  float* lnpart /*[B][nlayers][6][D] scratch of the backward: per-image LayerNorm parameter gradients*/;
  unsigned long long* gbox /*ar_bwd_box_bytes(): hand-off of the backward's layer pipeline*/; unsigned tag; unsigned* err; long long timeout_ticks;
  int B, T, D, F, V, H, Nsrc, sos;
  float p_att, p_res, p_ff; const uint32_t* seed; uint32_t site;
  long long* prof;   // optional [16] wall-clock ticks per phase family of workgroup 0 (SATRN_PROF=ar)
  int G;             // forward: workgroups (weight slices) per image, ar_fwd_slices()
  int kv_lds;        // forward (set by the launcher): history and cross-attention keys / values of the slice held in LDS
  unsigned long long* fbox;   // forward, G > 1: ar_fwd_box_bytes() mailbox of the slices' exchanges
};
bool ar_train_ok(int dt, int B, int D, int F, int V, int H, int T, int Nsrc, int nlayers);
int launch_ar_fwd(int dt, const ArP& p, hipStream_t s);   // -1: not launched
int ar_fwd_slices(int dt, int D, int F, int H);
size_t ar_fwd_box_bytes(int B, int G, int D);
int launch_ar_bwd(int dt, const ArP& p, hipStream_t s);   // -1: not launched (shape / residency / no mailbox)
size_t ar_bwd_box_bytes(int B, int T, int D, int nlayers);
// best-first beam search (networks/EfficientSATRN.py:708-867): DecodeP.steps = max_sequence - 1 expansions (= cache rows per
// image); node tables are per image [NN], NN >= 1 + bw*steps; path [steps][pstride] uint16; out int64 [B][max_seq]
struct BeamP {
  int bw, max_seq, eos, pad, NN, pstride;
  int32_t *parent, *tok, *len, *slot;
  double *logp, *score;
  uint16_t* path;
  int64_t* out;
};
int launch_beam_search(int dt, const DecodeP& p, const BeamP& q, hipStream_t s);
void launch_repack_kpanel(int dt, const void* src /*[N][K]*/, void* dst /*[K/32][N][32]*/, int N, int K, hipStream_t s);
// DecodingManager.sift / reset as launches: x [B][ld] logits (or probabilities), state int32 [B][4], targets int64 [B], probs [B][ldp]
void launch_sift(const float* x, int ld, int32_t* state, const int32_t* rules, int B, int V, int64_t* targets, float* probs,
                 int ldp, hipStream_t s);
void launch_sift_reset(int32_t* state, int B, int sos, hipStream_t s);
void launch_sift_strided(float* x, int ld, int32_t* state, const int32_t* rules, int B, int V, int64_t* targets, int ldt,
                         hipStream_t s);

// ---- evaluation-time image transform (kernels_image.hip): one descriptor per image (device pointer to uint8 HWC pixels, row stride in bytes)
struct ImageDesc { const unsigned char* data; int h, w, stride, pad_; };
void launch_image_preprocess(const ImageDesc* descs_dev, int B, int C, int H, int W, float* out /*[B][C][H][W]*/, const float* mean3,
                             const float* std3, hipStream_t s);

// ---- SwinTRN-specific data movement (kernels_swin.hip) ----------------------------------------------------------------
void launch_patchify(int dt, const float* img, void* out /*[B*(H/P)*(W/P)][Cin*P*P]*/, int B, int Cin, int H, int W, int P, hipStream_t s);
void launch_add_rows_table(int dt, const void* x, const float* table /*[LC] fp32*/, void* out, int B, long LC, hipStream_t s);
void launch_window_perm(int dt, const void* in, void* out, int B, int H, int W, int C, int ws, int shift, int reverse, int beta, hipStream_t s);
void launch_patch_merge(int dt, const void* in, void* out, int B, int H, int W, int C /*input channels*/, int reverse, int beta, hipStream_t s);
void launch_relpos_bias(const float* table /*[(2ws-1)^2][heads]*/, float* bias /*[heads][N][N]*/, int ws, int heads, hipStream_t s);
void launch_relpos_bias_bwd(const float* dbias /*[heads][N][ld]*/, float* dtable, int ws, int heads, int ld, float scale, hipStream_t s);
// map (ws != 0, C = channels): the BRANCH side is in window order -- mode 0 reads b at the window row of each token, mode 1 writes out there
void launch_droppath(int dt, int mode, const void* a, const void* b, void* out, int B, long per_sample, float p, const uint32_t* seed,
                     uint32_t site, hipStream_t s, RowMap map = RowMap(), int C = 0);
