// extern "C" boundary of libsatrn_hip.so (declarations + reference citations: include/satrn_hip.h).
#include <math.h>
#include <string.h>

#include <string>

#include "../../include/satrn_hip.h"
#include "engine.h"

void launch_act_fwd(int dt, const void* u, void* z, long n, int act, hipStream_t s);

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
static int done(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(-10, std::string(what) + ": " + hipGetErrorString(e));
  return 0;
}
#define S(x) ((hipStream_t)(x))
// operator-level calls have no engine workspace behind them: they always use the atomic reduction forms
#define CHK_DT(dt) do { sw_refresh(); g_det.on = 0; g_sebox.box = nullptr; g_sebox.images = 0; g_sebox.bwd = false; g_mbbox.box = nullptr; g_mbbox.words = 0; g_mbbox.images = 0; g_wgpart.cap = 0; if ((dt) != 0 && (dt) != 1) return fail(-1, "dtype must be 0 (f32) or 1 (bf16)"); } while (0)
static int chk_c(int dt, int c, const char* what) {
  int ch = dt == DT_BF16 ? 8 : 4;
  if (c <= 0 || c % ch) return fail(-1, std::string(what) + " must be a positive multiple of " + std::to_string(ch));
  return 0;
}

extern "C" {

const char* satrn_last_error(void) { return g_err.c_str(); }
int satrn_abi_version(void) { return 1; }

int satrn_pack_dense(int dt, const float* w, void* fwd, void* bwd, int N, int K, int ldb, void* st) {
  CHK_DT(dt);
  launch_pack_dense_ld(dt, w, fwd, bwd, N, K, ldb, S(st));
  return done("pack_dense");
}
int satrn_pack_conv3x3(int dt, const float* w, void* fwd, void* bwd, int Co, int Ci, void* st) {
  CHK_DT(dt);
  launch_pack_conv(dt, w, fwd, bwd, Co, Ci, 9, S(st));
  return done("pack_conv3x3");
}
int satrn_pack_dwconv3x3(int dt, const float* w, void* out, int C, void* st) {
  CHK_DT(dt);
  launch_pack_dw(dt, w, out, C, S(st));
  return done("pack_dwconv3x3");
}

int satrn_linear_fwd(int dt, const void* x, const void* w, const float* bias, void* y, int M, int N, int K, int act,
                     int out_f32, float drop_p, const uint32_t* seed, uint32_t site, void* st) {
  CHK_DT(dt);
  if (chk_c(dt, K, "K")) return -1;
  if (drop_p > 0.f && !seed) return fail(-1, "dropout needs a device seed");
  GemmP p;
  memset(&p, 0, sizeof(p));
  p.A = x; p.Bw = w; p.C = y; p.bias = bias; p.M = M; p.N = N; p.K = K; p.lda = K; p.ldc = N; p.act = act;
  p.out_f32 = out_f32; p.drop_p = drop_p; p.seed = seed; p.site = site;
  launch_gemm(dt, AM_DENSE, p, S(st));
  return done("linear_fwd");
}
int satrn_linear_fwd_stats(int dt, const void* x, const void* w, void* y, int M, int N, int K, float* stats, int stats_rep, const void* bnb_y,
                           const float* bnb_ss, const float* bnb_mr, int bnb_act, int accumulate, void* st) {
  CHK_DT(dt);
  if (chk_c(dt, K, "K") || chk_c(dt, N, "N")) return -1;
  if (!stats || stats_rep < 1) return fail(-1, "satrn_linear_fwd_stats: stats [stats_rep][2N] (zeroed) is required");
  if (bnb_y && (!bnb_ss || !bnb_mr)) return fail(-1, "satrn_linear_fwd_stats: bnb_y needs bnb_ss and bnb_mr");
  GemmP p;
  memset(&p, 0, sizeof(p));
  p.A = x; p.Bw = w; p.C = y; p.M = M; p.N = N; p.K = K; p.lda = K; p.ldc = N; p.beta = accumulate;
  p.stats = stats; p.stats_rep = stats_rep; p.bnb_y = bnb_y; p.bnb_ss = bnb_ss; p.bnb_mr = bnb_mr; p.bnb_act = bnb_act;
  launch_gemm(dt, AM_DENSE, p, S(st));
  return done("linear_fwd_stats");
}
int satrn_enc_attn_region_fwd(const void* x, const float* ln_w, const float* ln_b, const void* wqkv, const float* bqkv, const void* wo, const float* bo,
                              int B, int L, int D, int heads, float attn_drop, float out_drop, const uint32_t* seed, uint32_t site_attn, uint32_t site_out,
                              void* y1, float* mean_rstd1, void* qkv, void* att, float* lse, void* parts_scratch, void* o, void* y2, float* mean_rstd2,
                              void* st) {
  if (!enc_attn_fused_ok(DT_BF16, L, D, heads)) return fail(-1, "satrn_enc_attn_region_fwd: bf16, L <= 64, D in {256, 512}, head_dim 64, even head count");
  if ((attn_drop > 0.f || out_drop > 0.f) && !seed) return fail(-1, "dropout needs a device seed");
  EncAttnP q;
  memset(&q, 0, sizeof(q));
  q.x = x; q.ln_w = ln_w; q.ln_b = ln_b; q.wqkv = wqkv; q.bqkv = bqkv; q.wo = wo; q.y1 = y1; q.mr = mean_rstd1; q.qkv = qkv; q.att = att; q.lse = lse;
  q.parts = parts_scratch; q.B = B; q.L = L; q.D = D; q.H = heads; q.LkP = (int)attn_lkp(L); q.inv_temp = 1.0f / sqrtf((float)D);
  q.drop_p = attn_drop; q.seed = seed; q.site = site_attn;
  if (!launch_enc_attn_fwd(q, S(st))) return fail(-1, "satrn_enc_attn_region_fwd: launch refused");
  launch_layernorm_parts(parts_scratch, heads / 2, (long)B * L * D, bo, out_drop, seed, site_out, o, x, ln_w, ln_b, y2, mean_rstd2, (long)B * L, D, S(st));
  return done("enc_attn_region_fwd");
}
int satrn_linear_bwd_data(int dt, const void* dy, int ldy, const void* wb, int ldb, void* dx, int M, int N, int K,
                          int accumulate, void* st) {
  CHK_DT(dt);
  if (chk_c(dt, ldb, "ldb") || chk_c(dt, ldy, "ldy")) return -1;
  (void)N;
  GemmP p;
  memset(&p, 0, sizeof(p));
  p.A = dy; p.Bw = wb; p.C = dx; p.M = M; p.N = K; p.K = ldb; p.lda = ldy; p.ldc = K; p.beta = accumulate;
  launch_gemm(dt, AM_DENSE, p, S(st));
  return done("linear_bwd_data");
}
int satrn_linear_bwd_weight_ws(int dt, const void* dy, int ldy, const void* x, float* dw, float* db, int M, int N, int K, float* ws,
                               size_t ws_floats, void* st) {
  CHK_DT(dt);
  if (chk_c(dt, K, "K") || chk_c(dt, ldy, "ldy")) return -1;
  if (!ws || !ws_floats) return fail(-1, "satrn_linear_bwd_weight_ws: a partial-tile slab is required");
  WgradP q;
  memset(&q, 0, sizeof(q));
  q.dY = dy; q.A = x; q.dW = dw; q.M = M; q.N = N; q.K = K; q.ldy = ldy; q.lda = K; q.nbatch = 1; q.nb_inner = 1;
  if (db && !sw_off("wgrad_bias")) q.dbias = db;
  g_wgpart.cap = ws_floats; g_wgpart.scratch[0] = g_wgpart.scratch[1] = ws; g_wgpart.side = nullptr;
  launch_wgrad(dt, q, S(st));
  g_wgpart.cap = 0;
  if (db && !q.dbias) launch_colsum(dt, dy, M, N, ldy, db, S(st));
  return done("linear_bwd_weight_ws");
}
int satrn_linear_act_fwd(int dt, const void* x, const void* w, const float* bias, void* y, void* dact, int M, int N, int K, int act, void* st) {
  CHK_DT(dt);
  if (chk_c(dt, K, "K")) return -1;
  if (act < ACT_RELU || act > ACT_GELU) return fail(-1, "satrn_linear_act_fwd: act must be 1 (ReLU) .. 4 (GELU)");
  if (!dact) return fail(-1, "satrn_linear_act_fwd: dact is required (satrn_linear_fwd is the form without it)");
  GemmP p;
  memset(&p, 0, sizeof(p));
  p.A = x; p.Bw = w; p.C = y; p.bias = bias; p.M = M; p.N = N; p.K = K; p.lda = K; p.ldc = N; p.act = act;
  p.pre_out = dact; p.pre_grad = 1;
  launch_gemm(dt, AM_DENSE, p, S(st));
  return done("linear_act_fwd");
}
int satrn_linear_bwd_data_act(int dt, const void* dy, int ldy, const void* wb, int ldb, const void* dact, int kind, float scale, void* dx, int M,
                              int N, int K, void* st) {
  CHK_DT(dt);
  if (chk_c(dt, ldb, "ldb") || chk_c(dt, ldy, "ldy")) return -1;
  if (kind != ACT_DFACTOR && kind != ACT_RELU) return fail(-1, "satrn_linear_bwd_data_act: kind must be 5 (stored derivative) or 1 (ReLU output)");
  if (!dact) return fail(-1, "satrn_linear_bwd_data_act: dact is required");
  (void)N;
  GemmP p;
  memset(&p, 0, sizeof(p));
  p.A = dy; p.Bw = wb; p.C = dx; p.M = M; p.N = K; p.K = ldb; p.lda = ldy; p.ldc = K;
  p.bact_u = dact; p.bact = kind; p.bact_scale = scale;
  launch_gemm(dt, AM_DENSE, p, S(st));
  return done("linear_bwd_data_act");
}
int satrn_linear_bwd_weight(int dt, const void* dy, int ldy, const void* x, float* dw, float* db, int M, int N, int K,
                            void* st) {
  CHK_DT(dt);
  if (chk_c(dt, K, "K") || chk_c(dt, ldy, "ldy")) return -1;
  WgradP q;
  memset(&q, 0, sizeof(q));
  q.dY = dy; q.A = x; q.dW = dw; q.M = M; q.N = N; q.K = K; q.ldy = ldy; q.lda = K; q.nbatch = 1; q.nb_inner = 1;
  const bool fold = db && !g_det.on && !sw_off("wgrad_bias");   // bias gradient summed inside the weight-gradient kernel
  if (fold) q.dbias = db;
  launch_wgrad(dt, q, S(st));
  if (db && !fold) launch_colsum(dt, dy, M, N, ldy, db, S(st));
  return done("linear_bwd_weight");
}

static void conv_geo(GemmP& p, int H, int W, int Ci, int OH, int OW, int stride, int pt, int pl) {
  p.H = H; p.W = W; p.Ci = Ci; p.OH = OH; p.OW = OW; p.KW = 3; p.stride = stride; p.pt = pt; p.pl = pl;
}
int satrn_conv3x3_fwd(int dt, const void* x, const void* w, void* y, int B, int H, int W, int Ci, int Co, int OH, int OW,
                      int stride, int pt, int pl, void* st) {
  CHK_DT(dt);
  if (chk_c(dt, Ci, "Ci") || chk_c(dt, Co, "Co")) return -1;
  GemmP p;
  memset(&p, 0, sizeof(p));
  p.A = x; p.Bw = w; p.C = y; p.M = B * OH * OW; p.N = Co; p.K = 9 * Ci; p.ldc = Co;
  conv_geo(p, H, W, Ci, OH, OW, stride, pt, pl);
  launch_gemm(dt, AM_CONV, p, S(st));
  return done("conv3x3_fwd");
}
int satrn_conv3x3_bn_eval_act_fwd(int dt, const void* x, const void* w, const float* escale, const float* eshift, int act, const void* res, void* y,
                                  int B, int H, int W, int Ci, int Co, int OH, int OW, int stride, int pt, int pl, void* st) {
  CHK_DT(dt);
  if (chk_c(dt, Ci, "Ci") || chk_c(dt, Co, "Co")) return -1;
  if (!escale || !eshift) return fail(-1, "conv3x3_bn_eval_act_fwd: escale / eshift are required");
  GemmP p;
  memset(&p, 0, sizeof(p));
  p.A = x; p.Bw = w; p.C = y; p.M = B * OH * OW; p.N = Co; p.K = 9 * Ci; p.ldc = Co;
  p.escale = escale; p.eshift = eshift; p.act = act; p.eres = res;
  conv_geo(p, H, W, Ci, OH, OW, stride, pt, pl);
  launch_gemm(dt, AM_CONV, p, S(st));
  return done("conv3x3_bn_eval_act_fwd");
}
int satrn_linear_bn_eval_act_fwd(int dt, const void* x, const void* w, const float* escale, const float* eshift, int act, const void* res, void* y,
                                 int M, int N, int K, void* st) {
  CHK_DT(dt);
  if (chk_c(dt, K, "K") || chk_c(dt, N, "N")) return -1;
  if (!escale || !eshift) return fail(-1, "linear_bn_eval_act_fwd: escale / eshift are required");
  GemmP p;
  memset(&p, 0, sizeof(p));
  p.A = x; p.Bw = w; p.C = y; p.M = M; p.N = N; p.K = K; p.lda = K; p.ldc = N;
  p.escale = escale; p.eshift = eshift; p.act = act; p.eres = res;
  launch_gemm(dt, AM_DENSE, p, S(st));
  return done("linear_bn_eval_act_fwd");
}
int satrn_conv3x3_bwd_data(int dt, const void* dy, const void* wb, void* dx, int B, int H, int W, int Ci, int Co, int OH,
                           int OW, int stride, int pt, int pl, int accumulate, void* st) {
  CHK_DT(dt);
  if (chk_c(dt, Ci, "Ci") || chk_c(dt, Co, "Co")) return -1;
  GemmP p;
  memset(&p, 0, sizeof(p));
  p.A = dy; p.Bw = wb; p.C = dx; p.M = B * H * W; p.N = Ci; p.K = 9 * Co; p.ldc = Ci; p.beta = accumulate;
  conv_geo(p, OH, OW, Co, H, W, stride, pt, pl);
  launch_gemm(dt, AM_DGRAD, p, S(st));
  return done("conv3x3_bwd_data");
}
int satrn_conv3x3_bwd_weight(int dt, const void* dy, const void* x, float* dw, int B, int H, int W, int Ci, int Co,
                             int OH, int OW, int stride, int pt, int pl, void* st) {
  CHK_DT(dt);
  if (chk_c(dt, Ci, "Ci") || chk_c(dt, Co, "Co")) return -1;
  WgradP q;
  memset(&q, 0, sizeof(q));
  q.dY = dy; q.A = x; q.dW = dw; q.M = B * OH * OW; q.N = Co; q.K = 9 * Ci; q.ldy = Co; q.conv = 1; q.nbatch = 1; q.nb_inner = 1;
  q.H = H; q.W = W; q.Ci = Ci; q.OH = OH; q.OW = OW; q.KW = 3; q.stride = stride; q.pt = pt; q.pl = pl;
  launch_wgrad(dt, q, S(st));
  return done("conv3x3_bwd_weight");
}
int satrn_stem_conv_fwd(int dt, const float* img, const float* w, void* y, int B, int Cin, int H, int W, int Co, int stride,
                        int pad, void* st) {
  CHK_DT(dt);
  if (chk_c(dt, Co, "Co")) return -1;
  int OH = (H + 2 * pad - 3) / stride + 1, OW = (W + 2 * pad - 3) / stride + 1;
  launch_stem_conv(dt, img, w, y, B, Cin, H, W, Co, OH, OW, stride, pad, S(st));
  return done("stem_conv_fwd");
}
int satrn_stem_conv_bwd_weight(int dt, const float* img, const void* dy, float* dw, int B, int Cin, int H, int W, int Co,
                               int stride, int pad, void* st) {
  CHK_DT(dt);
  if (chk_c(dt, Co, "Co")) return -1;
  int OH = (H + 2 * pad - 3) / stride + 1, OW = (W + 2 * pad - 3) / stride + 1;
  launch_stem_wgrad(dt, img, dy, dw, B, Cin, H, W, Co, OH, OW, stride, pad, S(st));
  return done("stem_conv_bwd_weight");
}
int satrn_dwconv3x3_fwd(int dt, const void* x, const void* wp, const float* bias, void* y, int B, int H, int W, int C,
                        int OH, int OW, int stride, int pt, int pl, void* st) {
  CHK_DT(dt);
  if (chk_c(dt, C, "C")) return -1;
  launch_dwconv(dt, 0, x, wp, bias, y, B, H, W, C, OH, OW, stride, pt, pl, 0, nullptr, S(st));
  return done("dwconv3x3_fwd");
}
int satrn_dwconv3x3_bn_eval_act_pool_fwd(int dt, const void* x, const void* wp, const float* bias, const float* escale, const float* eshift, int act,
                                         void* y, float* pool, int B, int H, int W, int C, void* st) {
  CHK_DT(dt);
  if (chk_c(dt, C, "C")) return -1;
  if (!escale || !eshift) return fail(-1, "dwconv3x3_bn_eval_act_pool_fwd: escale / eshift are required");
  if (!launch_dwconv_eval_img(dt, x, wp, bias, escale, eshift, act, y, pool, B, H, W, C, S(st))) {
    launch_dwconv(dt, 0, x, wp, bias, y, B, H, W, C, H, W, 1, 1, 1, 0, nullptr, S(st), escale, eshift, act);
    if (pool) launch_image_pool(dt, y, pool, B, H * W, C, S(st));
  }
  return done("dwconv3x3_bn_eval_act_pool_fwd");
}
int satrn_dwconv3x3_bwd_data(int dt, const void* dy, const void* wp, void* dx, int B, int H, int W, int C, int OH, int OW,
                             int stride, int pt, int pl, int accumulate, void* st) {
  CHK_DT(dt);
  if (chk_c(dt, C, "C")) return -1;
  launch_dwconv(dt, 1, dy, wp, nullptr, dx, B, OH, OW, C, H, W, stride, pt, pl, accumulate, nullptr, S(st));
  return done("dwconv3x3_bwd_data");
}
int satrn_dwconv3x3_bwd_weight(int dt, const void* x, const void* dy, float* dw, float* dbias, int B, int H, int W, int C,
                               int OH, int OW, int stride, int pt, int pl, void* st) {
  CHK_DT(dt);
  if (chk_c(dt, C, "C")) return -1;
  launch_dwconv_wgrad(dt, x, dy, dw, dbias, nullptr, B, H, W, C, OH, OW, stride, pt, pl, S(st));
  return done("dwconv3x3_bwd_weight");
}

int satrn_batchnorm_act_fwd(int dt, const void* y, const float* w, const float* b, float* rm, float* rv, int64_t* nbt,
                            float eps, int train, int act, const void* res, void* z, long M, int C, float* scratch,
                            void* st) {
  CHK_DT(dt);
  if (chk_c(dt, C, "C")) return -1;
  if (train) launch_colstats(dt, y, M, C, scratch, S(st));
  launch_bn_act(dt, y, train ? scratch : nullptr, 1, w, b, rm, rv, train ? nbt : nullptr, eps, 0.1f, scratch + 2 * C, scratch + 4 * C,
                res, z, M, C, act, S(st));
  return done("batchnorm_act_fwd");
}
int satrn_batchnorm_act_se_fwd(int dt, const void* y, const float* w, const float* b, float* rm, float* rv, int64_t* nbt, float eps, int act, void* z,
                               int keep_z, const void* W1, const float* b1, const void* W2, const float* b2, float* pooled, float* u1, float* s1, void* gate,
                               void* out, int B, int HW, int C, int S, float* scratch, unsigned long long* mailbox, int mailbox_images, void* st) {
  CHK_DT(dt);
  if (chk_c(dt, C, "C")) return -1;
  if (!z || !scratch) return fail(-1, "batchnorm_act_se_fwd: z and scratch are required");
  const long M = (long)B * HW;
  launch_colstats(dt, y, M, C, scratch, S(st));
  if (!launch_bn_pool_se(dt, y, scratch, 1, w, b, rm, rv, nbt, eps, 0.1f, scratch + 2 * C, scratch + 4 * C, keep_z ? z : nullptr, W1, b1, W2, b2, pooled, u1, s1,
                         gate, out, mailbox, mailbox_images, B, HW, C, S, act, S(st))) {
    // the separate kernels (the squeeze-and-excite kernel pools by itself)
    launch_bn_act(dt, y, scratch, 1, w, b, rm, rv, nbt, eps, 0.1f, scratch + 2 * C, scratch + 4 * C, nullptr, z, M, C, act, S(st));
    launch_se_fwd(dt, z, W1, b1, W2, b2, pooled, u1, s1, gate, B, HW, C, S, S(st));
    launch_se_scale(dt, z, gate, out, B, HW, C, S(st));
  }
  return done("batchnorm_act_se_fwd");
}
static int mbconv_front_impl(const void* x, const MbXinArgs* xin, const void* W0, void* y1, const float* w1, const float* b1n, float* rm1, float* rv1, int64_t* nbt1,
                             float* coef1, void* z1, const void* dwp, void* y2, const float* w2, const float* b2n, float* rm2, float* rv2, int64_t* nbt2, float* coef2,
                             void* z2, int keep_z2, const void* W1, const float* b1, const void* W2, const float* b2, float* pooled, float* u1, float* s1,
                             void* gate, void* z3, int B, int H, int W, int Cin, int C, int S, float eps, unsigned long long* mailbox, long mailbox_words,
                             void* st) {
  CHK_DT(DT_BF16);
  if (B < 1 || C < 64 || (C % 64) || !mailbox) return fail(-1, "mbconv_front_fwd: B >= 1, C a multiple of 64 and a mailbox are required");
  const long bn_words = 3L * (C / 64) * B * 128, se_words = (long)B * (C / 64) * 64;
  if (mailbox_words < bn_words + se_words) return fail(-1, "mbconv_front_fwd: mailbox too small (3 * (C / 64) * B * 128 + B * (C / 64) * 64 words)");
  g_mbbox.box = mailbox; g_mbbox.words = (size_t)bn_words; g_mbbox.images = 64;
  g_sebox.box = mailbox + bn_words; g_sebox.images = B;
  const bool okk = mbconv_front_ok(DT_BF16, B, H, W, Cin, C, S, S(st)) &&
                   launch_mbconv_front(DT_BF16, x, xin, W0, y1, w1, b1n, rm1, rv1, nbt1, coef1, coef1 + 2 * C, eps, z1, dwp, y2, w2, b2n, rm2, rv2, nbt2, coef2,
                                       coef2 + 2 * C, eps, keep_z2 ? z2 : nullptr, W1, b1, W2, b2, pooled, u1, s1, gate, z3, B, H, W, Cin, C, S, 0.1f, S(st));
  g_mbbox.box = nullptr; g_mbbox.words = 0; g_mbbox.images = 0; g_sebox.box = nullptr; g_sebox.images = 0;
  if (!okk) return fail(-1, "mbconv_front_fwd: shape not taken by the one-launch form (use the separate operators)");
  return done("mbconv_front_fwd");
}
int satrn_mbconv_front_fwd(const void* x, const void* W0, void* y1, const float* w1, const float* b1n, float* rm1, float* rv1, int64_t* nbt1, float* coef1,
                           void* z1, const void* dwp, void* y2, const float* w2, const float* b2n, float* rm2, float* rv2, int64_t* nbt2, float* coef2,
                           void* z2, int keep_z2, const void* W1, const float* b1, const void* W2, const float* b2, float* pooled, float* u1, float* s1,
                           void* gate, void* z3, int B, int H, int W, int Cin, int C, int S, float eps, unsigned long long* mailbox, long mailbox_words,
                           void* st) {
  if (!x) return fail(-1, "mbconv_front_fwd: x is required");
  return mbconv_front_impl(x, nullptr, W0, y1, w1, b1n, rm1, rv1, nbt1, coef1, z1, dwp, y2, w2, b2n, rm2, rv2, nbt2, coef2, z2, keep_z2, W1, b1, W2, b2, pooled, u1,
                           s1, gate, z3, B, H, W, Cin, C, S, eps, mailbox, mailbox_words, st);
}
int satrn_mbconv_front_fwd_bn_in(const void* in_y, const void* in_res, const float* in_sums, int in_sums_rep, const float* in_weight, const float* in_bias,
                                 float* in_rm, float* in_rv, int64_t* in_nbt, float* in_coef, void* x_out, const void* W0, void* y1, const float* w1,
                                 const float* b1n, float* rm1, float* rv1, int64_t* nbt1, float* coef1, void* z1, const void* dwp, void* y2, const float* w2,
                                 const float* b2n, float* rm2, float* rv2, int64_t* nbt2, float* coef2, void* z2, int keep_z2, const void* W1, const float* b1,
                                 const void* W2, const float* b2, float* pooled, float* u1, float* s1, void* gate, void* z3, int B, int H, int W, int Cin, int C,
                                 int S, float eps, unsigned long long* mailbox, long mailbox_words, void* st) {
  if (!in_y || !in_sums || !in_coef || !x_out) return fail(-1, "mbconv_front_fwd_bn_in: in_y, in_sums, in_coef and x_out are required");
  MbXinArgs xa{in_y, in_res, in_sums, in_sums_rep, in_weight, in_bias, in_rm, in_rv, in_nbt, in_coef, in_coef + 2 * Cin, eps, x_out};
  return mbconv_front_impl(nullptr, &xa, W0, y1, w1, b1n, rm1, rv1, nbt1, coef1, z1, dwp, y2, w2, b2n, rm2, rv2, nbt2, coef2, z2, keep_z2, W1, b1, W2, b2, pooled, u1,
                           s1, gate, z3, B, H, W, Cin, C, S, eps, mailbox, mailbox_words, st);
}
int satrn_mbconv_bwd_se(const void* dy3, const void* w_bwd, int ldb, void* dz3, const void* bn2_y, const float* coef2, const void* gate, const float* u1,
                        const void* W1, const void* W2, float* dz2, float* ds1, float* du1, void* dpooled, float* bn2_sums, int B, int H, int W, int Cout,
                        int C, int S, unsigned long long* mailbox, long mailbox_words, void* st) {
  CHK_DT(DT_BF16);
  if (B < 1 || C < 64 || (C % 64) || !mailbox || mailbox_words < (long)B * (C / 64) * 64) return fail(-1, "mbconv_bwd_se: a mailbox of B * (C / 64) * 64 words is required");
  g_sebox.box = mailbox; g_sebox.images = B;
  const bool okk = launch_mbconv_bwd_se(DT_BF16, dy3, nullptr, w_bwd, ldb, dz3, bn2_y, coef2, coef2 + 2 * C, gate, u1, W2, W1, dz2, ds1, du1, dpooled, bn2_sums, B, H, W,
                                        Cout, C, S, S(st));
  g_sebox.box = nullptr; g_sebox.images = 0;
  if (!okk) return fail(-1, "mbconv_bwd_se: shape not taken by the one-launch form (use satrn_linear_bwd_data + satrn_se_bwd_bnred)");
  return done("mbconv_bwd_se");
}
int satrn_mbconv_bwd_se_bn_in(const void* dz, const void* bn3_y, const float* coef3, const float* w3, const float* sums3, void* dy3_out, float* dw3, float* db3,
                              const void* w_bwd, int ldb, void* dz3, const void* bn2_y, const float* coef2, const void* gate, const float* u1, const void* W1,
                              const void* W2, float* dz2, float* ds1, float* du1, void* dpooled, float* bn2_sums, int B, int H, int W, int Cout, int C, int S,
                              unsigned long long* mailbox, long mailbox_words, void* st) {
  CHK_DT(DT_BF16);
  if (B < 1 || C < 64 || (C % 64) || !mailbox || mailbox_words < (long)B * (C / 64) * 64) return fail(-1, "mbconv_bwd_se_bn_in: a mailbox of B * (C / 64) * 64 words is required");
  if (!dz || !bn3_y || !coef3 || !w3 || !sums3 || !dy3_out) return fail(-1, "mbconv_bwd_se_bn_in: dz, bn3_y, bn3_coef, bn3_weight, bn3_sums and dy3_out are required");
  g_sebox.box = mailbox; g_sebox.images = B;
  MbDinArgs da{dz, bn3_y, coef3, coef3 + 2 * Cout, w3, sums3, 1, dy3_out, dw3, db3};
  const bool okk = launch_mbconv_bwd_se(DT_BF16, nullptr, &da, w_bwd, ldb, dz3, bn2_y, coef2, coef2 + 2 * C, gate, u1, W2, W1, dz2, ds1, du1, dpooled, bn2_sums, B, H, W,
                                        Cout, C, S, S(st));
  g_sebox.box = nullptr; g_sebox.images = 0;
  if (!okk) return fail(-1, "mbconv_bwd_se_bn_in: shape not taken by the one-launch form");
  return done("mbconv_bwd_se_bn_in");
}
int satrn_batchnorm_act_dwconv3x3_fwd(int dt, const void* y, const float* w, const float* b, float* rm, float* rv, int64_t* nbt,
                                      float eps, int act, void* z, const void* dwp, const float* dwb, void* out, float* out_stats,
                                      int B, int H, int W, int C, float* scratch, void* st) {
  CHK_DT(dt);
  if (chk_c(dt, C, "C")) return -1;
  const long M = (long)B * H * W;
  launch_colstats(dt, y, M, C, scratch, S(st));
  if (!launch_bn_dwconv(dt, y, scratch, 1, w, b, rm, rv, nbt, eps, 0.1f, scratch + 2 * C, scratch + 4 * C, z, dwp, dwb, out, out_stats, B, H, W, C,
                        act, S(st))) {
    launch_bn_act(dt, y, scratch, 1, w, b, rm, rv, nbt, eps, 0.1f, scratch + 2 * C, scratch + 4 * C, nullptr, z, M, C, act, S(st));
    launch_dwconv(dt, 0, z, dwp, dwb, out, B, H, W, C, H, W, 1, 1, 1, 0, out_stats, S(st));
  }
  return done("batchnorm_act_dwconv3x3_fwd");
}
int satrn_dwconv3x3_bwd_data_bnred(int dt, const void* dout, const void* dwp, void* dz, int accumulate, const void* y, const float* scratch,
                                   int act, float* scratch2, int B, int H, int W, int C, void* st) {
  CHK_DT(dt);
  if (chk_c(dt, C, "C")) return -1;
  if (!launch_dwconv_bwd_bn(dt, dout, dwp, dz, accumulate, y, scratch + 2 * C, scratch + 4 * C, act, scratch2, B, H, W, C, S(st))) {
    launch_dwconv(dt, 1, dout, dwp, nullptr, dz, B, H, W, C, H, W, 1, 1, 1, accumulate, nullptr, S(st));
    launch_bn_bwd_reduce(dt, dz, y, scratch + 2 * C, scratch + 4 * C, (long)B * H * W, C, act, scratch2, S(st));
  }
  return done("dwconv3x3_bwd_data_bnred");
}
int satrn_bn_bwd_apply_dwconv3x3_bwd_data_bnred(int dt, const void* dz2, const void* y2, const float* wb, const float* scratch_b, int act_b,
                                                const float* scratch2_b, void* dy2, float* dwb, float* dbb, const void* dwp, void* dz, int accumulate,
                                                const void* y, const float* scratch, int act, float* scratch2, int B, int H, int W, int C, void* st) {
  CHK_DT(dt);
  if (chk_c(dt, C, "C")) return -1;
  const long M = (long)B * H * W;
  BnBwdHold h;
  h.armed = true; h.dz = dz2; h.y = y2; h.ss = scratch_b + 2 * C; h.mr = scratch_b + 4 * C; h.w = wb; h.red = scratch2_b; h.M = M; h.C = C; h.act = act_b;
  h.dy = dy2; h.dwp = dwb; h.dbp = dbb;
  if (!launch_dwconv_bwd_bn(dt, dy2, dwp, dz, accumulate, y, scratch + 2 * C, scratch + 4 * C, act, scratch2, B, H, W, C, S(st), &h)) {
    launch_bn_bwd_apply(dt, dz2, y2, scratch_b + 2 * C, scratch_b + 4 * C, wb, scratch2_b, M, C, act_b, dy2, dwb, dbb, S(st));
    if (!launch_dwconv_bwd_bn(dt, dy2, dwp, dz, accumulate, y, scratch + 2 * C, scratch + 4 * C, act, scratch2, B, H, W, C, S(st))) {
      launch_dwconv(dt, 1, dy2, dwp, nullptr, dz, B, H, W, C, H, W, 1, 1, 1, accumulate, nullptr, S(st));
      launch_bn_bwd_reduce(dt, dz, y, scratch + 2 * C, scratch + 4 * C, M, C, act, scratch2, S(st));
    }
  }
  return done("bn_bwd_apply_dwconv3x3_bwd_data_bnred");
}
int satrn_bn_bwd_apply_dwconv3x3_bwd_data_bn_bwd(const void* dz2, const void* y2, const float* wb, const float* scratch_b, int act_b, const float* scratch2_b,
                                                 void* dy2, float* dwb, float* dbb, const void* dwp, const void* y, const float* wa, const float* scratch,
                                                 int act, void* dy1, float* dwa, float* dba, int B, int H, int W, int C, unsigned long long* mailbox,
                                                 long mailbox_words, void* st) {
  CHK_DT(DT_BF16);
  if (chk_c(DT_BF16, C, "C")) return -1;
  if (B < 1 || (C % 64) || !mailbox || mailbox_words < (long)B * (C / 64) * 128)
    return fail(-1, "bn_bwd_apply_dwconv3x3_bwd_data_bn_bwd: a mailbox of B * (C / 64) * 128 words is required");
  const long M = (long)B * H * W;
  BnBwdHold h;
  h.armed = true; h.dz = dz2; h.y = y2; h.ss = scratch_b + 2 * C; h.mr = scratch_b + 4 * C; h.w = wb; h.red = scratch2_b; h.M = M; h.C = C; h.act = act_b;
  h.dy = dy2; h.dwp = dwb; h.dbp = dbb;
  BnBwdTail tl;
  tl.dy = dy1; tl.w = wa; tl.dwp = dwa; tl.dbp = dba;
  g_mbbox.box = mailbox; g_mbbox.words = (size_t)mailbox_words; g_mbbox.images = B;
  const bool okk = launch_dwconv_bwd_bn(DT_BF16, dy2, dwp, nullptr, 0, y, scratch + 2 * C, scratch + 4 * C, act, nullptr, B, H, W, C, S(st), &h, &tl);
  g_mbbox.box = nullptr; g_mbbox.words = 0; g_mbbox.images = 0;
  if (!okk) return fail(-1, "bn_bwd_apply_dwconv3x3_bwd_data_bn_bwd: shape not taken by the one-launch form (use satrn_bn_bwd_apply_dwconv3x3_bwd_data_bnred + satrn_batchnorm_act_bwd_apply)");
  return done("bn_bwd_apply_dwconv3x3_bwd_data_bn_bwd");
}
int satrn_batchnorm_act_bwd_apply(int dt, const void* dz, const void* y, const float* w, const float* scratch, int act, void* dy, float* dw,
                                  float* db, long M, int C, const float* scratch2, void* st) {
  CHK_DT(dt);
  if (chk_c(dt, C, "C")) return -1;
  launch_bn_bwd_apply(dt, dz, y, scratch + 2 * C, scratch + 4 * C, w, scratch2, M, C, act, dy, dw, db, S(st));
  return done("batchnorm_act_bwd_apply");
}
int satrn_batchnorm_act_bwd(int dt, const void* dz, const void* y, const float* w, const float* scratch, int act,
                            void* dy, float* dw, float* db, long M, int C, float* scratch2, void* st) {
  CHK_DT(dt);
  if (chk_c(dt, C, "C")) return -1;
  launch_bn_bwd_reduce(dt, dz, y, scratch + 2 * C, scratch + 4 * C, M, C, act, scratch2, S(st));
  launch_bn_bwd_apply(dt, dz, y, scratch + 2 * C, scratch + 4 * C, w, scratch2, M, C, act, dy, dw, db, S(st));
  return done("batchnorm_act_bwd");
}

int satrn_maxpool2x2_fwd(int dt, const void* x, void* y, int B, int H, int W, int C, void* st) {
  CHK_DT(dt);
  if (chk_c(dt, C, "C")) return -1;
  launch_maxpool(dt, 0, x, nullptr, y, B, H, W, C, S(st));
  return done("maxpool_fwd");
}
int satrn_maxpool2x2_bwd(int dt, const void* x, const void* dy, void* dx, int B, int H, int W, int C, void* st) {
  CHK_DT(dt);
  if (chk_c(dt, C, "C")) return -1;
  launch_maxpool(dt, 1, x, dy, dx, B, H, W, C, S(st));
  return done("maxpool_bwd");
}

int satrn_layernorm_fwd(int dt, const void* a, const void* b, const float* w, const float* bias, void* out, float* mr,
                        long R, int C, float eps, void* st) {
  CHK_DT(dt);
  if (chk_c(dt, C, "C")) return -1;
  launch_layernorm(dt, a, b, w, bias, out, mr, R, C, eps, 0.f, nullptr, 0, S(st));
  return done("layernorm_fwd");
}
int satrn_layernorm_bwd(int dt, const void* dout, const void* a, const void* b, const float* w, const float* mr, void* da,
                        void* db, int acc_a, int acc_b, float* dw, float* dbias, long R, int C, void* st) {
  CHK_DT(dt);
  if (chk_c(dt, C, "C")) return -1;
  launch_layernorm_bwd(dt, dout, a, b, w, mr, da, db, acc_a, acc_b, dw, dbias, R, C, 0.f, nullptr, 0, S(st));
  return done("layernorm_bwd");
}

int satrn_se_fwd(int dt, const void* x, const void* W1, const float* b1, const void* W2, const float* b2, const float* pool_sums,
                 float* pooled, float* u1, float* s1, void* gate, void* y, int B, int HW, int C, int S, void* st) {
  CHK_DT(dt);
  if (B <= 0 || HW <= 0 || C <= 0 || S <= 0 || (C % 8) != 0) return fail(-1, "satrn_se_fwd: bad shape");
  if (!(pool_sums && launch_se_mlp_scale(dt, x, pool_sums, W1, b1, W2, b2, pooled, u1, s1, gate, y, B, HW, C, S, S(st)))) {
    launch_se_fwd(dt, x, W1, b1, W2, b2, pooled, u1, s1, gate, B, HW, C, S, S(st));
    launch_se_scale(dt, x, gate, y, B, HW, C, S(st));
  }
  return done("se_fwd");
}
int satrn_se_bwd(int dt, const void* dy, const void* x, const void* gate, const float* u1, const void* W1, const void* W2,
                 float* dz2, float* du1, float* ds1_zeroed, void* dgate_scratch, void* dpooled, int B, int HW, int C, int S, void* st) {
  CHK_DT(dt);
  if (B <= 0 || HW <= 0 || C <= 0 || S <= 0 || (C % 8) != 0) return fail(-1, "satrn_se_bwd: bad shape");
  if (!(ds1_zeroed && launch_se_bwd_wide(dt, dy, x, gate, u1, W1, W2, dz2, du1, ds1_zeroed, dpooled, B, HW, C, S, S(st)))) {
    if (!dgate_scratch) return fail(-1, "satrn_se_bwd: the per-image form needs dgate_scratch [B][C]");
    launch_se_bwd_gate(dt, dy, x, dgate_scratch, B, HW, C, S(st));
    launch_se_bwd(dt, dgate_scratch, gate, u1, nullptr, nullptr, W1, W2, dz2, du1, dpooled, nullptr, nullptr, nullptr, nullptr, B, C, S, S(st), 1);
  }
  return done("se_bwd");
}
int satrn_se_bwd_bnred(int dt, const void* dy, const void* bn_y, const float* bn_scratch, int act, const void* gate, const float* u1, const void* W1,
                       const void* W2, float* dz2, float* du1, float* ds1_zeroed, void* dpooled, float* P_scratch, float* bn_scratch2, int B, int HW,
                       int C, int S, void* st) {
  CHK_DT(dt);
  if (B <= 0 || HW <= 0 || C <= 0 || S <= 0 || (C % 8) != 0) return fail(-1, "satrn_se_bwd_bnred: bad shape");
  if (!bn_y || !bn_scratch || !P_scratch || !bn_scratch2 || !ds1_zeroed) return fail(-1, "satrn_se_bwd_bnred: null operand");
  if (!launch_se_bwd_wide(dt, dy, nullptr, gate, u1, W1, W2, dz2, du1, ds1_zeroed, dpooled, B, HW, C, S, S(st), bn_y, bn_scratch + 2 * C, bn_scratch + 4 * C, act,
                          P_scratch, bn_scratch2))
    return fail(-1, "satrn_se_bwd_bnred: the wide form does not take this dtype / shape (use satrn_se_bwd + satrn_batchnorm_act_bwd)");
  return done("se_bwd_bnred");
}
int satrn_se_bwd_bnred_mbox(int dt, const void* dy, const void* bn_y, const float* bn_scratch, int act, const void* gate, const float* u1, const void* W1,
                            const void* W2, float* dz2, float* du1, float* ds1_zeroed, void* dpooled, float* P_scratch, float* bn_scratch2, int B, int HW,
                            int C, int S, unsigned long long* mailbox, int mailbox_images, void* st) {
  CHK_DT(dt);
  if (B <= 0 || HW <= 0 || C <= 0 || S <= 0 || (C % 8) != 0) return fail(-1, "satrn_se_bwd_bnred_mbox: bad shape");
  if (!bn_y || !bn_scratch || !P_scratch || !bn_scratch2 || !ds1_zeroed) return fail(-1, "satrn_se_bwd_bnred_mbox: null operand");
  g_sebox.box = mailbox; g_sebox.images = mailbox ? mailbox_images : 0; g_sebox.bwd = mailbox != nullptr;
  const bool okw = launch_se_bwd_wide(dt, dy, nullptr, gate, u1, W1, W2, dz2, du1, ds1_zeroed, dpooled, B, HW, C, S, S(st), bn_y, bn_scratch + 2 * C,
                                      bn_scratch + 4 * C, act, P_scratch, bn_scratch2);
  g_sebox.box = nullptr; g_sebox.images = 0; g_sebox.bwd = false;
  if (!okw) return fail(-1, "satrn_se_bwd_bnred_mbox: the wide form does not take this dtype / shape (use satrn_se_bwd + satrn_batchnorm_act_bwd)");
  return done("se_bwd_bnred_mbox");
}
int satrn_se_bwd_weights(const float* dz2, const float* du1, const float* s1, const float* pooled, float* dW1, float* db1, float* dW2, float* db2,
                         int B, int C, int S, void* st) {
  if (B <= 0 || C <= 0 || S <= 0) return fail(-1, "satrn_se_bwd_weights: bad shape");
  if (!dz2 || !du1 || !s1 || !pooled || !dW1 || !db1 || !dW2 || !db2) return fail(-1, "satrn_se_bwd_weights: null operand");
  launch_se_bwd(DT_F32, nullptr, nullptr, nullptr, s1, pooled, nullptr, nullptr, const_cast<float*>(dz2), const_cast<float*>(du1), nullptr, dW1, db1, dW2, db2,
                B, C, S, S(st), 2);
  return done("se_bwd_weights");
}
int satrn_pool_hw(int dt, const void* x, void* out, int B, int HW, int C, void* st) {
  CHK_DT(dt);
  if (chk_c(dt, C, "C")) return -1;
  launch_pool_hw(dt, x, out, B, HW, C, S(st));
  return done("pool_hw");
}
int satrn_posenc2d_fwd(int dt, const void* x, const void* gate, const float* hpos, const float* wpos, void* out, int B,
                       int H, int W, int C, void* st) {
  CHK_DT(dt);
  if (chk_c(dt, C, "C")) return -1;
  launch_posenc2d(dt, x, gate, hpos, wpos, out, B, H, W, C, S(st));
  return done("posenc2d_fwd");
}
int satrn_posenc2d_bwd_gate(int dt, const void* dout, const float* hpos, const float* wpos, void* dgate, int B, int H,
                            int W, int C, void* st) {
  CHK_DT(dt);
  if (chk_c(dt, C, "C")) return -1;
  launch_posenc2d_bwd(dt, dout, hpos, wpos, dgate, B, H, W, C, S(st));
  return done("posenc2d_bwd_gate");
}
int satrn_encoder_reshape(int dt, int inverse, const void* in, void* out, int B, int HW, int C, int accumulate, void* st) {
  CHK_DT(dt);
  launch_reshape_quirk(dt, inverse, in, out, B, HW, C, accumulate, S(st));
  return done("encoder_reshape");
}

static void fill_attn(AttnP& p, const void* q, const void* k, const void* v, int B, int heads, int Lq, int Lk, int hd,
                      int ldq, int ldk, int ldv, int ldo, int causal, const int64_t* text, int ld_text, int pad_id,
                      float temperature, float drop_p, const uint32_t* seed, uint32_t site) {
  memset(&p, 0, sizeof(p));
  p.Q = q; p.K = k; p.V = v; p.B = B; p.H = heads; p.Lq = Lq; p.Lk = Lk; p.hd = hd; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv;
  p.ldo = ldo; p.sq_b = (long)Lq * ldq; p.sk_b = (long)Lk * ldk; p.sv_b = (long)Lk * ldv; p.so_b = (long)Lq * ldo;
  p.causal = causal; p.text = text; p.ld_text = ld_text; p.pad_id = pad_id; p.inv_temp = 1.0f / temperature;
  p.drop_p = drop_p; p.seed = seed; p.site = site;
}
int satrn_attention_fwd(int dt, const void* q, const void* k, const void* v, void* o, float* lse, int B, int heads, int Lq,
                        int Lk, int hd, int ldq, int ldk, int ldv, int ldo, int causal, const int64_t* text, int ld_text,
                        int pad_id, float temperature, float drop_p, const uint32_t* seed, uint32_t site, void* st) {
  CHK_DT(dt);
  AttnP p;
  fill_attn(p, q, k, v, B, heads, Lq, Lk, hd, ldq, ldk, ldv, ldo, causal, text, ld_text, pad_id, temperature, drop_p, seed, site);
  p.O = o; p.lse = lse;
  if (launch_attn_checked(dt, 0, p, S(st))) return fail(-1, "attention: unsupported shape (Lk <= 256, head_dim <= 64, LDS fit)");
  return done("attention_fwd");
}
int satrn_attention_bwd(int dt, const void* q, const void* k, const void* v, const void* o, const float* lse,
                        const void* d_o, void* dq, void* dk, void* dv, void* ws, int B, int heads, int Lq, int Lk, int hd,
                        int ldq, int ldk, int ldv, int ldo, int causal, const int64_t* text, int ld_text, int pad_id,
                        float temperature, float drop_p, const uint32_t* seed, uint32_t site, void* st) {
  CHK_DT(dt);
  AttnP p;
  fill_attn(p, q, k, v, B, heads, Lq, Lk, hd, ldq, ldk, ldv, ldo, causal, text, ld_text, pad_id, temperature, drop_p, seed, site);
  const size_t es = dt == DT_BF16 ? 2 : 4;
  const size_t LkP = attn_lkp(Lk);
  p.O = (void*)o; p.lse = (float*)lse; p.dO = d_o; p.dQ = dq; p.dS = ws; p.Pd = (char*)ws + (size_t)B * heads * Lq * LkP * es;
  // short sequences (bf16): dQ, dK and dV in ONE launch, no dS / Pd workspace traffic (kernels_attn2.hip)
  p.dK = dk; p.dV = dv; p.kv_accum = 0;
  if (dt == DT_BF16 && launch_attn2_bwd(p, S(st))) return done("attention_bwd");
  if (launch_attn_checked(dt, 1, p, S(st))) return fail(-1, "attention: unsupported shape (Lk <= 256, head_dim <= 64, LDS fit)");
  WgradP w;
  memset(&w, 0, sizeof(w));
  w.M = Lq; w.N = Lk; w.K = hd; w.ldy = (int)LkP; w.out_t = 1; w.nbatch = B * heads; w.nb_inner = heads;
  w.sY_o = (long)heads * Lq * LkP; w.sY_i = (long)Lq * LkP;
  w.dY = p.Pd; w.A = d_o; w.lda = ldo; w.sA_o = (long)Lq * ldo; w.sA_i = hd; w.dW = dv; w.sW_o = (long)Lk * ldv; w.sW_i = hd; w.ldw = ldv;
  launch_wgrad(dt, w, S(st));
  w.dY = p.dS; w.A = q; w.lda = ldq; w.sA_o = (long)Lq * ldq; w.sA_i = hd; w.dW = dk; w.sW_o = (long)Lk * ldk; w.ldw = ldk;
  launch_wgrad(dt, w, S(st));
  return done("attention_bwd");
}

int satrn_embedding_fwd(int dt, const int64_t* ids, int ld_ids, const float* table, const float* pe, void* out, int B,
                        int L, int D, int pos0, float drop_p, const uint32_t* seed, uint32_t site, void* st) {
  CHK_DT(dt);
  launch_embed(dt, ids, table, pe, out, B, L, ld_ids, D, pos0, drop_p, seed, site, S(st));
  return done("embedding_fwd");
}
int satrn_embedding_bwd(int dt, const int64_t* ids, int ld_ids, const void* dout, float* dtable, int B, int L, int D,
                        float drop_p, const uint32_t* seed, uint32_t site, void* st) {
  CHK_DT(dt);
  launch_embed_bwd(dt, ids, dout, dtable, B, L, ld_ids, D, drop_p, seed, site, S(st));
  return done("embedding_bwd");
}
int satrn_cross_entropy(int dt, const float* logits, const int64_t* targets, int ld, int B, int T, int V, int Vp,
                        int pad_id, float* loss_out, float* lse_ws, void* dlogits, void* st) {
  CHK_DT(dt);
  launch_ce_full(dt, logits, targets, ld, 0, B, T, V, Vp, pad_id, loss_out, lse_ws, dlogits, nullptr, S(st));
  return done("cross_entropy");
}
int satrn_step_metrics(const int64_t* sequence, int ld_seq, int T, const int64_t* expected, int ld_exp, int L, int B, int pad_id,
                       int sos_id, int eos_id, int empty_id, double* acc, void* st) {
  if (!sequence || !expected || !acc || B <= 0 || T <= 0 || L <= 0 || T > 512 || L > 512) return fail(-1, "satrn_step_metrics: bad argument (sequences up to 512 tokens)");
  launch_step_metrics(sequence, ld_seq, T, expected, ld_exp, L, B, pad_id, sos_id, eos_id, empty_id, acc, S(st));
  return done("step_metrics");
}
int satrn_kd_loss(const float* student, const float* teacher, const int64_t* labels, int ld, int B, int T, int V,
                  float temperature, float alpha, float* loss_out, float* dlogits, void* st) {
  if (!student || !teacher || !labels || !loss_out || !dlogits || B <= 0 || T <= 0 || V <= 0 || !(temperature > 0.f))
    return fail(-1, "satrn_kd_loss: bad argument");
  launch_kd_loss(student, teacher, labels, ld, B, T, V, temperature, alpha, loss_out, dlogits, S(st));
  return done("kd_loss");
}
int satrn_clip_adamw(float* p, const float* g, float* m, float* v, long n, float* gnorm_sq, float* scratch1024,
                     const float* hyper, void* st) {
  launch_sumsq(g, n, gnorm_sq, scratch1024, S(st));
  launch_adamw(p, g, m, v, n, gnorm_sq, hyper, S(st));
  return done("clip_adamw");
}

// ---- model level -------------------------------------------------------------------------------------------
struct satrn_model { Model* m; };

satrn_model* satrn_model_create(const satrn_config* c) {
  if (!c) { g_err = "null config"; return nullptr; }
  if (c->dtype != 0 && c->dtype != 1) { g_err = "dtype must be 0 or 1"; return nullptr; }
  const int ch = c->dtype == DT_BF16 ? 8 : 4;
  if (c->enc_hidden % 16 || c->dec_hidden % ch || c->enc_filter % ch || c->dec_filter % ch || c->dec_src != c->enc_hidden) {
    g_err = "hidden sizes must be multiples of 16 (encoder) / 8 and dec_src == enc_hidden";
    return nullptr;
  }
  if (c->enc_hidden % c->enc_heads || c->dec_hidden % c->dec_heads) { g_err = "hidden_dim must divide by head_num"; return nullptr; }
  SatrnConfig k;
  k.network = c->network; k.rgb = c->rgb; k.height = c->height; k.width = c->width;
  k.enc_hidden = c->enc_hidden; k.enc_filter = c->enc_filter; k.enc_heads = c->enc_heads; k.enc_layers = c->enc_layers;
  k.dec_src = c->dec_src; k.dec_hidden = c->dec_hidden; k.dec_filter = c->dec_filter; k.dec_heads = c->dec_heads;
  k.dec_layers = c->dec_layers; k.num_classes = c->num_classes; k.pad_id = c->pad_id; k.sos_id = c->sos_id;
  k.dropout = c->dropout; k.dtype = c->dtype;
  if (c->network == 2) {
    k.swin_embed = c->swin_embed; k.swin_window = c->swin_window; k.swin_patch = c->swin_patch; k.swin_head_classes = c->swin_head_classes;
    k.swin_drop_path = c->swin_drop_path;
    for (int i = 0; i < 4; ++i) { k.swin_depths[i] = c->swin_depths[i]; k.swin_heads[i] = c->swin_heads[i]; }
    // networks/SWIN.py:559-572 (square image, whole patches), :59 (windows tile every stage), head_dim as the attention kernel takes it
    if (c->height != c->width || c->swin_patch <= 0 || c->height % c->swin_patch) { g_err = "SwinTRN: square input, side a multiple of the patch size"; return nullptr; }
    if (c->swin_embed <= 0 || c->swin_embed % 32 || c->dec_src != c->swin_embed * 8) { g_err = "SwinTRN: embed_dim a multiple of 32 and dec_src == 8 * embed_dim"; return nullptr; }
    if ((c->rgb * c->swin_patch * c->swin_patch) % ch) { g_err = "SwinTRN: rgb * patch^2 must be a multiple of 8"; return nullptr; }
    int res = c->height / c->swin_patch;
    for (int i = 0; i < 4; ++i, res /= 2) {
      const int dim = c->swin_embed << i, ws = res <= c->swin_window ? res : c->swin_window;
      if (c->swin_depths[i] <= 0 || c->swin_heads[i] <= 0 || dim % c->swin_heads[i] || (dim / c->swin_heads[i]) % ch || dim / c->swin_heads[i] > 64) { g_err = "SwinTRN: bad depth / head count (head_dim <= 64, multiple of 8)"; return nullptr; }
      if (res < 1 || res % ws || (i < 3 && res % 2) || ws * ws > 256) { g_err = "SwinTRN: the window must tile every stage's resolution (and ws*ws <= 256 keys)"; return nullptr; }
    }
    if (c->swin_head_classes <= 0) { g_err = "SwinTRN: head classes"; return nullptr; }
  } else if (c->network != 0 && c->network != 1) { g_err = "network must be 0 (LiteSATRN), 1 (EfficientSATRN) or 2 (SwinTRN)"; return nullptr; }
  satrn_model* h = new satrn_model();
  h->m = model_create(k);
  return h;
}
void satrn_model_destroy(satrn_model* h) { if (h) { model_destroy(h->m); delete h; } }
int satrn_model_num_state(const satrn_model* h) { return (int)h->m->state.size(); }
const char* satrn_model_state_name(const satrn_model* h, int i) { return h->m->state[i].name.c_str(); }
int satrn_model_state_info(const satrn_model* h, int i, int* kind, int* ndim, int64_t* shape4, int64_t* offset, int* init,
                           int* fan_in, int* fan_out) {
  if (i < 0 || i >= (int)h->m->state.size()) return fail(-1, "state index out of range");
  const StateEntry& e = h->m->state[i];
  *kind = e.kind; *ndim = (int)e.shape.size(); *offset = e.offset; *init = e.init; *fan_in = e.fan_in; *fan_out = e.fan_out;
  for (int d = 0; d < 4; ++d) shape4[d] = d < (int)e.shape.size() ? e.shape[d] : 1;
  return 0;
}
int64_t satrn_model_flat_size(const satrn_model* h, int kind) {
  return kind == 0 ? h->m->n_params : kind == 1 ? h->m->n_buf_f32 : h->m->n_buf_i64;
}
static int mret(satrn_model* h, int rc, const char* what) {
  if (rc) return fail(rc, std::string(what) + ": " + h->m->err);
  return done(what);
}
int satrn_model_bind(satrn_model* h, float* p, float* g, float* bf, int64_t* bi) { return mret(h, model_bind(h->m, p, g, bf, bi), "bind"); }
size_t satrn_model_workspace_bytes(satrn_model* h, int B, int L) { return model_workspace_bytes(h->m, B, L); }
int satrn_model_set_workspace(satrn_model* h, void* ws, size_t bytes, void* st) { return mret(h, model_set_workspace(h->m, ws, bytes, S(st)), "set_workspace"); }
int satrn_model_pack_weights(satrn_model* h, void* st) { return mret(h, model_pack_weights(h->m, S(st)), "pack_weights"); }
int satrn_model_forward(satrn_model* h, const float* img, const int64_t* exp, int B, int L, int train, int record,
                        int teacher_forced, float* logits, void* st) {
  return mret(h, model_forward(h->m, img, exp, B, L, train != 0, record != 0, logits, S(st), teacher_forced != 0), "forward");
}
int satrn_model_backward(satrn_model* h, const float* dl, void* st) { return mret(h, model_backward(h->m, dl, S(st)), "backward"); }
int satrn_model_loss_backward(satrn_model* h, const int64_t* exp, int B, int L, void* st) {
  return mret(h, model_loss_backward(h->m, exp, B, L, S(st)), "loss_backward");
}
int satrn_model_train_step(satrn_model* h, const float* img, const int64_t* exp, int B, int L, const float* hyper9,
                           int use_graph, int phase, void* st) {
  return mret(h, model_train_step(h->m, img, exp, B, L, hyper9, use_graph, phase, S(st)), "train_step");
}
int satrn_model_train_step_dual(satrn_model* h, const float* img, const int64_t* exp, int B, int L, const float* hyper9_enc,
                                const float* hyper9_dec, int phase, void* st) {
  if (!hyper9_enc || !hyper9_dec) return fail(-1, "satrn_model_train_step_dual: both hyper-parameter arrays are required");
  return mret(h, model_train_step(h->m, img, exp, B, L, hyper9_enc, 0, phase, S(st), hyper9_dec), "train_step_dual");
}
int satrn_model_last_sequence(satrn_model* h, int64_t* ids, int B, int L, void* st) {
  if (!ids || B <= 0 || L < 2) return fail(-1, "satrn_model_last_sequence: bad argument");
  return mret(h, model_last_sequence(h->m, ids, B, L, S(st)), "last_sequence");
}
int satrn_model_read_grad_norms(satrn_model* h, float* out2, void* st) {
  if (!out2) return fail(-1, "satrn_model_read_grad_norms: out2 is null");
  return mret(h, model_read_grad_norms(h->m, out2, S(st)), "read_grad_norms");
}
int satrn_model_read_loss(satrn_model* h, float* out4, void* st) { return mret(h, model_read_loss(h->m, out4, S(st)), "read_loss"); }
int satrn_model_encode(satrn_model* h, const float* img, int B, float* src, void* st) { return mret(h, model_encode(h->m, img, B, src, S(st)), "encode"); }
int satrn_model_greedy(satrn_model* h, const float* img, const float* src, int B, int steps, float* logits, int64_t* ids,
                       int use_graph, void* st) {
  return mret(h, model_greedy(h->m, img, src, B, steps, logits, ids, use_graph, S(st)), "greedy");
}
int satrn_model_greedy_rules(satrn_model* h, const float* img, const float* src, int B, int steps, const int32_t* rules,
                             float* probs, int64_t* ids, void* st) {
  if (!rules) return fail(-1, "satrn_model_greedy_rules: rules is null");
  return mret(h, model_greedy(h->m, img, src, B, steps, probs, ids, 0, S(st), rules), "greedy_rules");
}
int satrn_model_greedy_forced(satrn_model* h, const float* img, const float* src, int B, int steps, const int64_t* forced_ids,
                              float* logits, int64_t* ids, void* st) {
  if (!forced_ids) return fail(-1, "satrn_model_greedy_forced: forced_ids is null");
  return mret(h, model_greedy(h->m, img, src, B, steps, logits, ids, 0, S(st), nullptr, forced_ids), "greedy_forced");
}
int satrn_model_probe_enable(satrn_model* h, int on) { if (!h || !h->m) return -1; h->m->probe_on = on != 0; return 0; }
int satrn_model_probe_count(satrn_model* h) { return (h && h->m) ? model_probe_count(h->m) : -1; }
int satrn_model_probe_info(satrn_model* h, int i, const char** name, int64_t* rows, int* cols) { return mret(h, model_probe_info(h->m, i, name, rows, cols), "probe_info"); }
int satrn_model_probe_read(satrn_model* h, int i, float* out_f32, void* st) { return mret(h, model_probe_read(h->m, i, out_f32, S(st)), "probe_read"); }
int satrn_model_probe_set_grad(satrn_model* h, int i, float* gout_f32) { return mret(h, model_probe_set_grad(h->m, i, gout_f32), "probe_set_grad"); }
int satrn_model_last_decode_path(satrn_model* h, int* giveups_out) {
  if (!h || !h->m) return -1;
  if (giveups_out) *giveups_out = h->m->pipe_giveups;
  return h->m->last_decode_path;
}
const char* satrn_model_decode_note(satrn_model* h) { return (h && h->m) ? h->m->decode_note.c_str() : ""; }
int satrn_model_beam_search(satrn_model* h, const float* img, int B, int beam_width, int max_sequence, int eos_id, int pad_id,
                            int64_t* sequences, void* st) {
  if (!img || !sequences || B <= 0) return fail(-1, "satrn_model_beam_search: bad argument");
  return mret(h, model_beam_search(h->m, img, B, beam_width, max_sequence, eos_id, pad_id, sequences, S(st)), "beam_search");
}
int satrn_sift(const float* x, int ld, int32_t* state, const int32_t* rules, int B, int V, int64_t* targets, float* probs,
               int ldp, void* st) {
  if (!x || !state || !rules || !targets || !probs || B <= 0 || V <= 0 || ld < V || ldp < V) return fail(-1, "satrn_sift: bad argument");
  launch_sift(x, ld, state, rules, B, V, targets, probs, ldp, S(st));
  return 0;
}
int satrn_sift_reset(int32_t* state, int B, int sos_id, void* st) {
  if (!state || B <= 0) return fail(-1, "satrn_sift_reset: bad argument");
  launch_sift_reset(state, B, sos_id, S(st));
  return 0;
}
int satrn_model_segment_range(satrn_model* h, int seg, int64_t* lo, int64_t* hi) {
  if (!h || seg < 0 || seg > 3 || !lo || !hi) return fail(-1, "satrn_model_segment_range: bad argument");
  *lo = h->m->seg_lo[seg]; *hi = h->m->seg_hi[seg];
  return 0;
}
int satrn_model_step_begin(satrn_model* h, const float* src, int B, int max_steps, void* st) {
  if (!src || B <= 0) return fail(-1, "satrn_model_step_begin: src is null or B <= 0");
  return mret(h, model_step_begin(h->m, src, B, max_steps, S(st)), "step_begin");
}
int satrn_model_step(satrn_model* h, const int64_t* target, float* logits, void* st) {
  if (!target || !logits) return fail(-1, "satrn_model_step: null pointer");
  return mret(h, model_step(h->m, target, logits, S(st)), "step");
}
int satrn_model_profile_step(satrn_model* h, const float* img, const int64_t* exp, int B, int L, char* json_out,
                             size_t cap, void* st) {
  return mret(h, model_profile_step(h->m, img, exp, B, L, json_out, cap, S(st)), "profile_step");
}
float* satrn_model_adam_state(satrn_model* h, int which) {
  Model* m = h->m;
  return which == 0 ? m->adam_m : m->adam_v;
}
int satrn_image_preprocess(const void* descs, int B, int C, int H, int W, float* out, const float* mean3, const float* std3, void* st) {
  if (!descs || !out || !mean3 || !std3 || B <= 0 || (C != 1 && C != 3) || H <= 0 || W <= 0) return fail(-1, "satrn_image_preprocess: bad argument (C must be 1 or 3)");
  launch_image_preprocess((const ImageDesc*)descs, B, C, H, W, out, mean3, std3, S(st));
  return done("image_preprocess");
}
int satrn_device_error(void* st) { return (int)device_error_read_clear(S(st)); }
int satrn_route_counts(long long* out, int n, int reset) {
  for (int i = 0; i < n; ++i) out[i] = i < RT_COUNT ? g_route[i] : 0;
  if (reset) for (int i = 0; i < RT_COUNT; ++i) g_route[i] = 0;
  return RT_COUNT;
}
int satrn_model_bind_optimizer(satrn_model* h, float* exp_avg, float* exp_avg_sq) { return mret(h, model_bind_optimizer(h->m, exp_avg, exp_avg_sq), "bind_optimizer"); }
long satrn_model_get_step(satrn_model* h) { return h->m->adam_t; }
int satrn_model_rng_state(satrn_model* h, uint32_t* seed_io_host, int set, void* st) { return mret(h, model_rng_state(h->m, seed_io_host, set, S(st)), "rng_state"); }
int satrn_model_set_step(satrn_model* h, long t) { h->m->adam_t = t; return 0; }

}  // extern "C"
