// SATRN engine: a static-plan executor for the EfficientSATRN / LiteSATRN hot path.
// The engine owns no device memory: parameters, gradients, BN buffers and one workspace are borrowed
// from the caller (PyTorch tensors).  A forward pass records a tape of backward closures; a whole
// training step (forward + CE + backward + clip + AdamW + weight re-pack) is captured into one hipGraph.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "kernels.h"

struct SatrnConfig {
  int network;  // 0 LiteSATRN, 1 EfficientSATRN
  int rgb;
  int height, width;
  int enc_hidden, enc_filter, enc_heads, enc_layers;
  int dec_src, dec_hidden, dec_filter, dec_heads, dec_layers;
  int num_classes;
  int pad_id, sos_id;
  float dropout;
  int dtype;  // DT_F32 / DT_BF16
  // network == 2 (SwinTRN, networks/SWIN.py:1024-1031): the reference hard-codes 384 / patch 4 / 128 / [2,2,18,2] / [4,8,16,32] /
  // window 12 / drop_path 0.5 / ape / 21841 head classes; smaller values only serve the parity tests
  int swin_embed = 0, swin_depths[4] = {0, 0, 0, 0}, swin_heads[4] = {0, 0, 0, 0}, swin_window = 0, swin_patch = 0, swin_head_classes = 0;
  float swin_drop_path = 0.f;
};

enum { ST_PARAM = 0, ST_BUF_F32 = 1, ST_BUF_I64 = 2 };
struct StateEntry {
  std::string name;
  std::vector<int64_t> shape;
  int kind;
  int64_t offset;  // element offset inside the flat buffer of its kind
  int64_t numel;
  int init;  // 0 xavier, 1 conv(kaiming-uniform), 2 linear_w, 3 linear_b(fan_in given), 4 ones, 5 zeros, 6 normal
  int fan_in, fan_out;
};

struct Vec { float* p = nullptr; float* g = nullptr; int n = 0; int64_t off = -1; };
enum { WK_DENSE = 0, WK_CONV3 = 1, WK_DW = 2, WK_STEM = 3 };
struct Wt {
  int kind = WK_DENSE;
  int N = 0, K = 0;        // GEMM view: N outputs, K = taps*Ci contraction
  int Co = 0, Ci = 0, taps = 1;
  int ldb = 0;             // row length of the transposed (backward) copy
  int64_t off = -1;        // element offset in flat params / grads
  float* p = nullptr; float* g = nullptr;
  void* fwd = nullptr; void* bwd = nullptr;  // packed compute copies
  int64_t pk_fwd_off = -1, pk_bwd_off = -1;  // byte offsets inside the persistent region
};
struct BNp { Vec w, b; int64_t rm_off = -1, rv_off = -1, nbt_off = -1; float* rm = nullptr; float* rv = nullptr; int64_t* nbt = nullptr; int C = 0; float eps = 1e-5f;
             size_t eval_off = 0; /* floats into the eval scale/shift table [2C] */ };
struct LNp { Vec w, b; int C = 0; };
struct MHAp { Wt qkv;   /* fused [3D][K] when q and kv share the input width, else q only */
              Wt kv;    /* cross attention: [2D][Ksrc] */
              Vec bqkv; /* fused bias [3D] (or [D] for q when cross) */
              Vec bkv; Wt out; Vec bout; Wt qonly; Vec bq; int D = 0, heads = 0; bool cross = false; };

struct Tensor {
  void* p = nullptr; void* g = nullptr;
  long rows = 0; int C = 0;
  int B = 0, H = 0, W = 0;
  bool g_init = false;
  bool f32 = false;  // logits
  int stats_rep = 1;
  float* stats = nullptr;  // [2C] column sum / sum of squares written by the producing kernel (BatchNorm input)
  // BatchNorm outputs (training): what the LAST gradient contributor needs to also produce the BN backward's column
  // sums in its own epilogue (bn_red, zeroed, [bn_red_rep][2C]); ncons counts forward consumers so far -- the first
  // consumer in forward order is the last contributor in backward order
  int ncons = 0;
  const void* bn_y = nullptr; const float* bn_ss = nullptr; const float* bn_mr = nullptr; int bn_act = 0;
  float* bn_red = nullptr; int bn_red_rep = 1;
  // ... or runs that BatchNorm's whole backward itself (launch_dwconv_bwd_bn with a tail): bn_src = the BatchNorm's input tensor, bn_p its
  // parameters; bn_applied tells the BatchNorm's closure that its input gradient and parameter gradients are already there
  Tensor* bn_src = nullptr; BNp* bn_p = nullptr; bool bn_applied = false;
  // squeeze-and-excite backward folded into this BatchNorm output's backward: g holds the SE OUTPUT's gradient and the
  // true gradient is g*se_gate[b] + se_dpool[b]/se_hw
  const void* se_gate = nullptr; const void* se_dpool = nullptr; int se_hw = 0;
  bool bn_has_res = false;  // the BatchNorm that produced this tensor also added a residual (its gradient = this tensor's)
  // output of a product whose epilogue applied an activation that needs its INPUT for the derivative (GELU): act_pre = the stored
  // pre-activation values.  A consumer whose data gradient is the only contribution multiplies by act'(act_pre) in its own epilogue
  // and sets g_preact: g then already holds the gradient of the pre-activation
  void* act_pre = nullptr; int act_kind = 0; bool g_preact = false; float act_scale = 0.f;   // ReLU (+dropout): act_pre = the output itself, act_scale = 1/(1-p)
  // output of a depthwise convolution whose backward kernel can also run the backward-apply pass of the BatchNorm that consumes this
  // tensor: that BatchNorm's closure leaves its operands in bhold instead of launching (op_bn_act -> op_dwconv, launch_dwconv_bwd_bn)
  bool dw_bwd_fuse = false; BnBwdHold bhold;
  // the output of a squeeze-and-excite block (op_se): the dense product that consumes it leaves its data gradient here instead of launching,
  // and op_se's backward runs both in one launch where the shape allows (launch_mbconv_bwd_se), else launches the held product first
  // ... and, one step earlier: the output y3 of the block's projection is marked (bn_bwd_hold_ok) so that the backward-apply pass of the BatchNorm
  // behind it (bn3) is not launched either (bhold); the projection's closure moves it -- with its own weight-gradient launch, which reads the dy3
  // that does not exist yet -- to the squeeze-and-excite output (after_fused), whose closure runs all of it in one launch or in the old order
  bool bn_bwd_hold_ok = false; std::function<void()> after_fused;
  bool se_out = false; std::shared_ptr<GemmP> dgrad_hold; double dgrad_hold_flops = 0, dgrad_hold_bytes = 0; int dgrad_hold_ldb = 0; const void* dgrad_hold_w = nullptr;
  // inference: a product whose launch is postponed until the BatchNorm that consumes it is known, so that BatchNorm (eval
  // statistics) + activation + residual run in its epilogue and the raw output is never written (op_gemm -> op_bn_act)
  std::shared_ptr<GemmP> pend; int pend_mode = 0;
  std::function<int(const float* escale, const float* eshift, int act, void* out, float* pool, void* se_out, const SeEvalArgs* se)> pend_dw;  // same for a depthwise conv
};

struct SwinBlock { LNp n1, n2; Wt qkv, proj, fc1, fc2; Vec bqkv, bproj, b1, b2, rpb; int dim = 0, heads = 0, res = 0, ws = 0, shift = 0; float drop_path = 0.f; int geo = -1; };
struct SwinStage { std::vector<SwinBlock> blocks; bool down = false; LNp dnorm; Wt dred; int dim = 0, res = 0; };
struct EffBlock { int type, cin, cout, mid, stride, se; bool skip; Wt c0, c1, dw, se_r, se_e; Vec se_rb, se_eb; BNp bn1, bn2, bn3; };
struct EncLayer { LNp norm; MHAp att; Wt conv0, conv1, dw; Vec dwb; BNp norm0, dwnorm, norm1; };
struct DecLayer { MHAp self_att, cross_att; LNp ln1, ln2, ln3; Wt lin0, lin1; Vec b0, b1; };

struct Model {
  SatrnConfig cfg;
  std::vector<StateEntry> state;
  int64_t n_params = 0, n_buf_f32 = 0, n_buf_i64 = 0;
  // architecture
  Wt stem; BNp stem_bn;
  std::vector<Wt> lite_conv; std::vector<BNp> lite_bn;
  std::vector<EffBlock> blocks;
  Wt conv_last; BNp bn_last;
  // SwinTRN encoder
  Wt sw_patch; Vec sw_patch_b, sw_ape; LNp sw_patch_norm, sw_norm; std::vector<SwinStage> swin; Wt sw_head; Vec sw_head_b;
  struct SwinGeo { int res, ws, shift; size_t off; size_t off_lab; }; std::vector<SwinGeo> sw_geo;  // shifted-window mask tables ([nW][N][N] fp32) and region labels ([nW][N] bytes) in the persistent region
  Wt pe_d0, pe_d1; Vec pe_b0, pe_b1;
  std::vector<EncLayer> enc;
  Wt embed;  // [V+1][Dd] (gathered directly from the fp32 master)
  std::vector<DecLayer> dec;
  Wt gen; Vec gen_b;
  std::vector<Wt*> all_w; std::vector<Vec*> all_v; std::vector<BNp*> all_bn;
  // bound memory
  float* params = nullptr; float* grads = nullptr; float* buf_f32 = nullptr; int64_t* buf_i64 = nullptr;
  char* ws = nullptr; size_t ws_bytes = 0;
  // persistent region layout (byte offsets inside ws)
  size_t persist_bytes = 0;
  size_t off_packed = 0, off_scalars = 0, off_pe1d = 0, off_hpos = 0, off_wpos = 0,
         off_stage_img = 0, off_stage_tgt = 0, off_zero = 0;
  size_t zero_bytes = 0, zero_hwm = 0;
  size_t off_sumsq = 0;
  static constexpr int SEBOX_IMAGES = 128; size_t off_sebox = 0;
  static constexpr int MBBOX_IMAGES = 64; static constexpr size_t MBBOX_WORDS = (size_t)3 * 24 * MBBOX_IMAGES * 128; size_t off_mbbox = 0;
  size_t off_wgpart = 0, wgpart_floats = 0;   // two partial-tile slabs of the persistent weight-gradient kernel (bf16)
  size_t off_det = 0, det_floats = 0;  // two scratch slabs of the deterministic reductions (f32 parity mode), 0 = atomics
  size_t off_bn_eval = 0, off_bn_desc = 0; std::vector<BnEvalDesc> bn_desc_host; bool bn_desc_dirty = true;
  size_t off_packdesc = 0, packdesc_bytes = 0, off_packblk = 0, packblk_bytes = 0; int pack_n = 0; long pack_total = 0; bool pack_dirty = true;
  int stage_B = 0, stage_L = 0;
  int feat_h = 0, feat_w = 0;
  bool bound = false, ws_set = false, tables_ready = false;
  // execution state
  struct Exec* ex = nullptr;
  hipGraphExec_t graphs[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; int graph_B = 0, graph_L = 0;
  long adam_t = 0;
  // optimizer state lives OUTSIDE the resizable workspace (caller-owned flat fp32 buffers, one element per parameter):
  // a workspace regrown for a longer batch must not restart Adam
  float* adam_m = nullptr; float* adam_v = nullptr;
  hipGraphExec_t decode_graph = nullptr; const void* decode_key[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  float* hy_pinned = nullptr; unsigned long hy_seq = 0;
  // step-wise decoding session (satrn_model_step_begin / satrn_model_step)
  std::vector<struct Tensor*> step_cross, step_cache; int step_B = 0, step_max = 0, step_t = 0; size_t step_mark = 0, step_keep = 0;
  // backward segments for overlapping the gradient exchange with the rest of the backward pass: tape marks recorded
  // in forward (early/late backbone, end of backbone, end of encoder) and the flat-gradient range each segment completes
  size_t seg_mark[3] = {0, 0, 0}; int64_t seg_lo[4] = {0, 0, 0, 0}, seg_hi[4] = {0, 0, 0, 0}; int late_block = 0; int seg_next = 0;
  long logits_epoch = -1;  // epoch of the arena that holds the last forward's logits
  long epoch = 0, step_epoch = -1;  // every arena reset bumps epoch: a session from an older epoch is dead
  // which decoder produced the last greedy result: 0 none yet, 1 role pipeline, 2 one workgroup per image, 3 step-wise launches;
  // pipe_giveups counts pipelines that timed out and were re-run on another kernel (never silently: see decode_note)
  int last_decode_path = 0, pipe_giveups = 0; std::string decode_note;
  bool probe_on = false;
  std::string err;
};

// scalars block (floats unless noted), all in device memory inside the persistent region
enum { SC_SEED = 0 /*uint32*/, SC_GNORM = 1, SC_GNORM2 = 2 /*decoder group, dual-optimizer step*/, SC_LOSS = 4 /*4 floats*/, SC_HYPER = 8 /*9 floats*/, SC_ONE = 20,
       SC_HYPER2 = 21 /*9 floats: decoder group*/, SC_COUNT = 32 };

struct ProfRec { std::string name; double flops = 0, bytes = 0; hipEvent_t a = nullptr, b = nullptr; };

struct Exec {
  Model* m = nullptr;
  std::vector<ProfRec>* prof = nullptr; double nflops = 0, nbytes = 0;
  void prof_begin(const char* call); void prof_end();
  hipStream_t s = nullptr;
  int dt = 0;
  bool train = false, rec = false, dry = false;
  // nolaunch: ops allocate their outputs and record their backward closures but launch nothing -- the caller produces the outputs with
  // ONE fused kernel (encoder self-attention region); the slots below hand it what the ops allocated internally
  // probes (diagnostics, satrn_model_probe_*): tensors at the stage boundaries of the last forward; gout != null -> the backward casts
  // the tensor's gradient into it (fp32) at the moment the tape passes the probe point (later the buffer may be aliased by another
  // tensor's gradient)
  struct Probe { std::string name; Tensor* t; float* gout; };
  std::vector<Probe> probes; bool probe_on = false;
  bool nolaunch = false;
  float* last_mr = nullptr; float* last_lse = nullptr; uint32_t last_site = 0; float last_drop = 0.f;
  // what the last op_bn_act / op_se allocated (scale|shift and mean|rstd; pooled means, hidden layer, gate; whether the activated tensor must
  // be stored): the MBConv block launch (eff_block) reads them after a nolaunch pass over the ops
  float* last_bn_ss = nullptr; float* last_bn_mr = nullptr;
  // the BatchNorm (+ residual) that ends an MBConv block, not launched: the next block's one-launch front normalises its input while staging it
  // (MbXinArgs, kernels.h); whoever cannot take it launches it (flush_xhold)
  struct FwdBnHold { bool armed = false; const void* y = nullptr; const void* res = nullptr; const float* sums = nullptr; int rep = 1; BNp* bn = nullptr;
                     float* ss = nullptr; float* mr = nullptr; void* z = nullptr; long M = 0; int C = 0; } xhold;
  struct LastSe { float* pooled = nullptr; float* u1 = nullptr; float* s1 = nullptr; void* gate = nullptr; bool need_x = true; } last_se;
  bool serial = false;  // no concurrent side stream (hipGraph capture / profiling): side kernels may fill the chip
  float drop = 0.f;
  char* base = nullptr; size_t cap = 0, off = 0, peak = 0;  // bump arena
  char* zbase = nullptr; size_t zcap = 0, zoff = 0;          // zero pool
  uint32_t site = 1;
  std::vector<std::function<void()>> tape;
  std::vector<std::unique_ptr<Tensor>> tens;
  Tensor* logits = nullptr; Tensor* src = nullptr;
  bool oom = false;
  hipStream_t s2 = nullptr; std::vector<hipEvent_t> evs; hipEvent_t evj = nullptr; int nfork = 0; bool forked = false;
  hipStream_t side(); void join();
  std::vector<std::function<void(hipStream_t)>> pending; void defer(std::function<void(hipStream_t)> fn); void flush_side();


  // SATRN_STAGE_PROF=1: events on the main stream at stage boundaries of an ordinary eager step (forward and, through tape
  // closures, backward) -> per-stage wall time of the critical chain, printed by the next step
  std::vector<std::pair<std::string, hipEvent_t>> marks; void mark(const char* name); void mark_report();
  void* alloc(size_t bytes);
  float* zalloc(size_t nfloats);
  Tensor* newt(long rows, int C, int B = 0, int H = 0, int W = 0, bool f32 = false);
  size_t esz() const { return dt == DT_BF16 ? 2 : 4; }
  void* grad(Tensor* t, int* beta);
  void reset(char* b, size_t c, char* zb, size_t zc);
};

Model* model_create(const SatrnConfig& cfg);
void model_destroy(Model* m);
size_t model_workspace_bytes(Model* m, int B, int L);
int model_bind(Model* m, float* params, float* grads, float* buf_f32, int64_t* buf_i64);
int model_bind_optimizer(Model* m, float* exp_avg, float* exp_avg_sq);
int model_rng_state(Model* m, uint32_t* seed_io, int set, hipStream_t s);
int model_set_workspace(Model* m, void* ws, size_t bytes, hipStream_t s);
int model_pack_weights(Model* m, hipStream_t s);
int model_forward(Model* m, const float* img, const int64_t* expected, int B, int L, bool train, bool record,
                  float* logits_out, hipStream_t s, bool teacher_forced = true);
int model_backward(Model* m, const float* dlogits, hipStream_t s);
int model_loss_backward(Model* m, const int64_t* expected, int B, int L, hipStream_t s);
int model_train_step(Model* m, const float* img, const int64_t* expected, int B, int L, const float* hyper9,
                     int use_graph, int phase, hipStream_t s, const float* hyper9_dec = nullptr);
int model_read_grad_norms(Model* m, float* out2, hipStream_t s);
int model_last_sequence(Model* m, int64_t* ids_out, int B, int L, hipStream_t s);
int model_read_loss(Model* m, float* out4, hipStream_t s);
int model_encode(Model* m, const float* img, int B, float* src_out, hipStream_t s);
int model_greedy(Model* m, const float* img, const float* src_or_null, int B, int steps, float* logits_out,
                 int64_t* ids_out, int use_graph, hipStream_t s, const int32_t* rules = nullptr, const int64_t* forced = nullptr);
int model_beam_search(Model* m, const float* img, int B, int beam_width, int max_sequence, int eos_id, int pad_id,
                      int64_t* sequences, hipStream_t s);
int model_backward_segment(Model* m, const int64_t* expected, int B, int L, int seg, hipStream_t s, int seg_to = -1);
int model_step_begin(Model* m, const float* src, int B, int max_steps, hipStream_t s);
int model_step(Model* m, const int64_t* target, float* logits_out, hipStream_t s);
int model_profile_step(Model* m, const float* img, const int64_t* expected, int B, int L, char* out, size_t out_cap,
                       hipStream_t s);
int model_probe_count(Model* m);
int model_probe_info(Model* m, int i, const char** name, int64_t* rows, int* cols);
int model_probe_read(Model* m, int i, float* out_f32, hipStream_t s);
int model_probe_set_grad(Model* m, int i, float* gout_f32);
