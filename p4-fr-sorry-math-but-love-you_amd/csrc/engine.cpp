// SATRN engine: model tables, op tape (forward + recorded backward), training step, greedy decode.
// Reference behaviour followed (paths relative to the reference repo):
//   networks/EfficientSATRN.py:63-87 (EfficientNet), :90-154 (PositionalEncoding), :157-228 (attention),
//   :231-281 (EncoderLayer, shared LayerNorm + raw reshape), :326-397 (Feedforward / TransformerDecoderLayer),
//   :400-426 (PositionEncoder1D), :429-566 (SATRNDecoder), networks/LiteSATRN.py:21-70 (ShallowCNN),
//   train_modules/train_single_opt.py:80-98 (loss, backward, clip, AdamW).
#include "engine.h"
#include <chrono>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>

void launch_act_fwd(int dt, const void* u, void* z, long n, int act, hipStream_t s);

// launch wrapper: skipped in dry (sizing) runs; with profiling on, each launch is bracketed by HIP events on the
// engine's stream and attributed to a kernel family (the launcher's name) with its algorithmic flops / bytes.
#define LCH(e, call) do { if (!(e).dry && !(e).nolaunch) { if ((e).prof) (e).prof_begin(#call); call; if ((e).prof) (e).prof_end(); } (e).nflops = 0; (e).nbytes = 0; } while (0)

// algorithmic work of the next launch, attributed by the profiler (satrn_model_profile_step): bytes = every operand read
// once + every result written once in its storage dtype, flops = 2 x multiply-accumulates
#define WORK(e, fl, by) do { (e).nflops = (double)(fl); (e).nbytes = (double)(by); } while (0)

// =====================================================================================================
// Exec: arena + tape
// =====================================================================================================
void Exec::prof_begin(const char* call) {
  ProfRec r;
  const char* p = strstr(call, "launch_");
  std::string nm = p ? p : call;
  size_t q = nm.find('(');
  if (q != std::string::npos) nm = nm.substr(0, q);
  static const bool shapes = sw_prof("shapes");  // split families by problem size
  if (shapes) { char b[64]; snprintf(b, sizeof(b), " B%.0f F%.0f", nbytes, nflops); nm += b; }
  r.name = nm; r.flops = nflops; r.bytes = nbytes;
  (void)hipEventCreate(&r.a); (void)hipEventCreate(&r.b);
  (void)hipEventRecord(r.a, s);
  prof->push_back(r);
}
void Exec::prof_end() { (void)hipEventRecord(prof->back().b, s); }

// Fork: everything launched on side() after this call may run concurrently with the main chain from this point on.
// Weight gradients are off the critical path (they are only needed by the optimizer), and the small late-stage kernels
// leave most CUs idle, so they go to the side stream; join() makes the main stream wait for all of them.
hipStream_t Exec::side() {
  if (!s2 || dry) return s;
  forked = true;
  return s2;
}
// Side launches are queued (they only need their inputs, which the main stream has already been asked to produce) and
// flushed a few closures later behind ONE event: keeps the host cost of forking at ~1/8 of an event pair per launch.
void Exec::defer(std::function<void(hipStream_t)> fn) {
  // timing experiment only (the chain alone; WRONG gradients): announced loudly, once, so that a stray variable cannot give a
  // silently non-training model
  static const bool skip = [] {
    return sw_timing("skip_wgrad") != 0;   // (weight / bias gradients are NOT computed: the data-gradient chain alone; never for training)
  }();
  if (skip && !dry) return;
  if (!s2 || dry) { fn(s); return; }
  pending.push_back(std::move(fn));
  // hand the batch to the side stream every 8 launches (round 2, tools/ab_flush.sh: 4 -> 12.35, 8 -> 12.14, 16 -> 12.15,
  // never -> 14.6 ms/step): every hand-over costs the chain 11-15 us (tools/micro/fork_cost.hip: an event record on the
  // chain's queue + a wait on the side queue; the record alone is 2 us).  Larger batches early in the backward and small
  // ones near the join were tried as well (32 / 8): no change -- what the fewer forks save, the later start of the side work costs
  // (round 4, with the chain at 421 launches: 4 -> 8.55, 8 -> 8.68, 2 / 3 -> 8.64-8.69 ms with 160 weight-gradient workgroups)
  static const int thr = (int)sw_knob("flush", 4);
  if ((int)pending.size() >= thr) flush_side();
}
void Exec::flush_side() {
  if (pending.empty()) return;
  // fork events order device work only (nobody on the host reads them): no system-scope fence at the record
  const unsigned evflags = (unsigned)(hipEventDisableTiming | hipEventDisableSystemFence);
  if (nfork >= (int)evs.size()) { hipEvent_t ev; (void)hipEventCreateWithFlags(&ev, evflags); evs.push_back(ev); }
  hipEvent_t ev = evs[nfork++];
  (void)hipEventRecord(ev, s);
  (void)hipStreamWaitEvent(s2, ev, 0);
  for (auto& f : pending) f(s2);
  pending.clear();
  forked = true;
}
void Exec::join() {
  flush_side();
  if (!s2 || dry || !forked) return;
  if (!evj) (void)hipEventCreateWithFlags(&evj, hipEventDisableTiming);
  static const bool jprof = sw_prof("join");  // how long the main chain waits for the side stream
  static hipEvent_t ja = nullptr, jb = nullptr;
  if (jprof) {
    if (!ja) { (void)hipEventCreate(&ja); (void)hipEventCreate(&jb); }
    else { float ms = 0.f; if (hipEventElapsedTime(&ms, ja, jb) == hipSuccess) fprintf(stderr, "[join] main waited %.3f ms for the side stream\n", ms); }
    (void)hipEventRecord(ja, s);
  }
  (void)hipEventRecord(evj, s2);
  (void)hipStreamWaitEvent(s, evj, 0);
  if (jprof) (void)hipEventRecord(jb, s);
  forked = false;
}

static const bool g_stage_prof = sw_prof("stage");
// replicas of a GEMM epilogue's column-sum target (BatchNorm statistics / BatchNorm-backward sums): tall products with few columns put
// thousands of same-address float atomics on 2 x C words; the row tiles spread them over `rep` copies which the consumer adds.
// SATRN_STATS_REP_SCALE (knob, read once) multiplies the replica count of the shapes that have replicas (tools/ab_bench.sh).
static int stats_rep_for(int C, long rows) {
  if (g_det.on) return 1;
  const int base = (C <= 64 && rows >= 65536) ? 16 : ((C <= 256 && rows >= 16384) ? 4 : 1);
  static const int sc = (int)sw_knob("stats_rep_scale", 1);   // n: times n; -n: divided by n
  if (base == 1 || sc == 0 || sc == 1) return base;
  return sc > 0 ? std::min(base * sc, 64) : std::max(base / -sc, 1);
}
void Exec::mark(const char* name) {
  if (!g_stage_prof || dry) return;
  hipEvent_t ev; (void)hipEventCreate(&ev);
  (void)hipEventRecord(ev, s);
  marks.emplace_back(name, ev);
}
void Exec::mark_report() {
  if (marks.size() < 2) { marks.clear(); return; }
  (void)hipEventSynchronize(marks.back().second);
  std::string out = "[stage]";
  float tot = 0.f;
  for (size_t i = 1; i < marks.size(); ++i) {
    float ms = 0.f; (void)hipEventElapsedTime(&ms, marks[i - 1].second, marks[i].second);
    char b[96]; snprintf(b, sizeof(b), " %s=%.3f", marks[i].first.c_str(), ms); out += b; tot += ms;
  }
  fprintf(stderr, "%s total=%.3f\n", out.c_str(), tot);
  for (auto& m_ : marks) (void)hipEventDestroy(m_.second);
  marks.clear();
}
void Exec::reset(char* b, size_t c, char* zb, size_t zc) {
  base = b; cap = c; off = 0; zbase = zb; zcap = zc; zoff = 0; site = 1;
  tape.clear(); tens.clear(); logits = nullptr; src = nullptr; oom = false; nfork = 0; forked = false;
}
void* Exec::alloc(size_t bytes) {
  size_t a = (off + 255) & ~(size_t)255;
  off = a + bytes;
  if (off > peak) peak = off;
  if (!dry && off > cap) { oom = true; return base; }
  return base + a;
}
float* Exec::zalloc(size_t n) {
  size_t a = (zoff + 63) & ~(size_t)63;
  zoff = a + n * sizeof(float);
  if (zoff > zcap) { oom = true; return (float*)zbase; }
  return (float*)(zbase + a);
}
Tensor* Exec::newt(long rows, int C, int B, int H, int W, bool f32) {
  tens.emplace_back(new Tensor());
  Tensor* t = tens.back().get();
  t->rows = rows; t->C = C; t->B = B; t->H = H; t->W = W; t->f32 = f32;
  t->p = alloc((size_t)rows * C * (f32 ? 4 : esz()));
  return t;
}
void* Exec::grad(Tensor* t, int* beta) {
  if (!t->g) t->g = alloc((size_t)t->rows * t->C * esz());
  if (beta) *beta = t->g_init ? 1 : 0;
  t->g_init = true;
  return t->g;
}

// =====================================================================================================
// model tables
// =====================================================================================================
namespace {
struct Builder {
  Model* m;
  int winit = 0;  // attention / feed-forward Linear weights: 0 xavier_normal_ (SATRN, networks/EfficientSATRN.py:193-196,336-337), 2 torch default (SWIN.py:776-793)
  int64_t addp(const std::string& name, std::vector<int64_t> shape, int init, int fan_in, int fan_out) {
    StateEntry e;
    e.name = name; e.shape = shape; e.kind = ST_PARAM; e.offset = m->n_params; e.init = init;
    e.fan_in = fan_in; e.fan_out = fan_out;
    e.numel = 1;
    for (auto d : shape) e.numel *= d;
    m->n_params += e.numel;
    m->state.push_back(e);
    return e.offset;
  }
  int64_t addb(const std::string& name, std::vector<int64_t> shape, int kind, int init) {
    StateEntry e;
    e.name = name; e.shape = shape; e.kind = kind; e.init = init; e.fan_in = e.fan_out = 0;
    e.numel = 1;
    for (auto d : shape) e.numel *= d;
    if (kind == ST_BUF_F32) { e.offset = m->n_buf_f32; m->n_buf_f32 += e.numel; }
    else { e.offset = m->n_buf_i64; m->n_buf_i64 += e.numel; }
    m->state.push_back(e);
    return e.offset;
  }
  Vec vec(const std::string& name, int n, int init, int fan_in = 0) {
    Vec v; v.n = n; v.off = addp(name, {n}, init, fan_in, 0);
    return v;
  }
  Wt dense(const std::string& name, int N, int K, int init, bool conv_shape = false) {
    Wt w; w.kind = WK_DENSE; w.N = N; w.K = K; w.Co = N; w.Ci = K; w.taps = 1; w.ldb = (N + 7) & ~7;
    std::vector<int64_t> shp = conv_shape ? std::vector<int64_t>{N, K, 1, 1} : std::vector<int64_t>{N, K};
    w.off = addp(name, shp, init, K, N);
    return w;
  }
  Wt conv3(const std::string& name, int Co, int Ci, int init, bool stem = false) {
    Wt w; w.kind = stem ? WK_STEM : WK_CONV3; w.N = Co; w.K = 9 * Ci; w.Co = Co; w.Ci = Ci; w.taps = 9; w.ldb = Co;
    w.off = addp(name, {Co, Ci, 3, 3}, init, Ci * 9, Co * 9);
    return w;
  }
  Wt dwc(const std::string& name, int C, int init) {
    Wt w; w.kind = WK_DW; w.N = C; w.K = 9; w.Co = C; w.Ci = 1; w.taps = 9;
    w.off = addp(name, {C, 1, 3, 3}, init, 9, C * 9);
    return w;
  }
  BNp bn(const std::string& name, int C, float eps) {
    BNp b; b.C = C; b.eps = eps;
    b.w = vec(name + ".weight", C, 4);
    b.b = vec(name + ".bias", C, 5);
    b.rm_off = addb(name + ".running_mean", {C}, ST_BUF_F32, 5);
    b.rv_off = addb(name + ".running_var", {C}, ST_BUF_F32, 4);
    b.nbt_off = addb(name + ".num_batches_tracked", {}, ST_BUF_I64, 5);
    return b;
  }
  LNp ln(const std::string& name, int C) {
    LNp l; l.C = C;
    l.w = vec(name + ".weight", C, 4);
    l.b = vec(name + ".bias", C, 5);
    return l;
  }
  // flat order: q.w k.w v.w q.b k.b v.b out.w out.b  (so [q;k;v] and [k;v] are contiguous fused operands)
  MHAp mha(const std::string& name, int D, int Ksrc, int heads) {
    MHAp a; a.D = D; a.heads = heads; a.cross = (Ksrc != D);
    int64_t qo = addp(name + ".q_linear.weight", {D, D}, winit, D, D);
    int64_t ko = addp(name + ".k_linear.weight", {D, Ksrc}, winit, Ksrc, D);
    addp(name + ".v_linear.weight", {D, Ksrc}, winit, Ksrc, D);
    int64_t qb = addp(name + ".q_linear.bias", {D}, 3, D, 0);
    int64_t kb = addp(name + ".k_linear.bias", {D}, 3, Ksrc, 0);
    addp(name + ".v_linear.bias", {D}, 3, Ksrc, 0);
    a.qkv.kind = WK_DENSE; a.qkv.off = qo; a.qkv.K = D; a.qkv.N = a.cross ? D : 3 * D;
    a.qkv.ldb = a.qkv.N; a.qkv.Co = a.qkv.N; a.qkv.Ci = D;
    a.bqkv.off = qb; a.bqkv.n = a.qkv.N;
    // kv view always exists (the step decoder projects K/V separately from Q)
    a.kv.kind = WK_DENSE; a.kv.off = ko; a.kv.K = Ksrc; a.kv.N = 2 * D; a.kv.ldb = 2 * D; a.kv.Co = 2 * D; a.kv.Ci = Ksrc;
    a.bkv.off = kb; a.bkv.n = 2 * D;
    // q-only view (the step decoder projects the single query row separately from the history's K/V)
    a.qonly.kind = WK_DENSE; a.qonly.off = qo; a.qonly.K = D; a.qonly.N = D; a.qonly.ldb = D; a.qonly.Co = D; a.qonly.Ci = D;
    a.bq.off = qb; a.bq.n = D;
    a.out = dense(name + ".out_linear.weight", D, D, winit);
    a.bout = vec(name + ".out_linear.bias", D, 3, D);
    return a;
  }
};

// EfficientNetV2-S block table (timm==0.4.9 tf_efficientnetv2_s; third-party)
struct StageDef { int type, rep, stride, exp, cout; float se; };
const StageDef kEffV2S[6] = {{0, 2, 1, 1, 24, 0.f},   {1, 4, 2, 4, 48, 0.f},   {1, 4, 2, 4, 64, 0.f},
                             {2, 6, 2, 4, 128, .25f}, {2, 9, 1, 6, 160, .25f}, {2, 15, 2, 6, 256, .25f}};
}  // namespace

static void segment_ranges(Model* m);

// ---- the side stream ----------------------------------------------------------------------------------------------------------------
// Optimizer-only work (weight gradients) runs on a second, low-priority stream beside the data-gradient chain.  ONE side stream per device
// and process, shared by every model.  Whether two streams really run side by side is decided by the runtime: it deals streams onto
// hardware queues, and a chain and a side stream whose queues share a dispatch pipe slow each other down instead of overlapping (seen
// in bench.py's process: with a chain stream and a side stream per MODEL the f32 leg ran at 43 ms per step against 23 ms alone, and with
// only the side stream shared the SwinTRN leg at 32 against 16: its chain sat on hardware queue 5, the side stream on queue 1; alone: 3
// and 1 -- tools/f32_in_process.py, tools/swin_in_process.py).  The Python module shares its chain stream the same way (networks.py
// _chain_stream), so every model of a process runs on the pair the first one got.  A timed probe (two spin kernels, one per stream)
// was tried as a detector and dropped: it called a pair good that then ran the step at 28 ms.
static std::mutex g_side_mu;
static std::map<int, hipStream_t> g_side;
static hipStream_t side_stream_get() {
  std::lock_guard<std::mutex> lock(g_side_mu);
  int dev = 0;
  (void)hipGetDevice(&dev);
  auto it = g_side.find(dev);
  if (it == g_side.end()) {
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);  // lo = numerically greatest = lowest priority
    hipStream_t s2 = nullptr;
    if (hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, lo) != hipSuccess)
      (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    it = g_side.emplace(dev, s2).first;
  }
  return it->second;
}

Model* model_create(const SatrnConfig& cfg) {
  sw_refresh();
  Model* m = new Model();
  m->cfg = cfg;
  Builder b{m};
  const int D = cfg.enc_hidden;
  if (cfg.network == 2) {
    // built below (after the shared decoder dims are known)
  } else if (cfg.network == 0) {
    int ch[5] = {cfg.rgb, D / 2, D, D, D};
    for (int i = 0; i < 4; ++i) {
      std::string n = "encoder.shallow_cnn.conv" + std::to_string(i) + ".weight";
      m->lite_conv.push_back(b.conv3(n, ch[i + 1], ch[i], i < 3 ? 0 : 1, i == 0));
      m->lite_bn.push_back(b.bn("encoder.shallow_cnn.batch_norm" + std::to_string(i), ch[i + 1], 1e-5f));
    }
  } else {
    const std::string p = "encoder.shallow_cnn.";
    m->stem = b.conv3(p + "conv_stem.weight", 24, cfg.rgb, 1, true);
    m->stem_bn = b.bn(p + "bn1", 24, 1e-3f);
    int cin = 24;
    for (int s = 0; s < 6; ++s) {
      for (int i = 0; i < kEffV2S[s].rep; ++i) {
        const StageDef& sd = kEffV2S[s];
        EffBlock eb;
        eb.type = sd.type; eb.cin = cin; eb.cout = sd.cout; eb.mid = cin * sd.exp; eb.stride = i == 0 ? sd.stride : 1;
        eb.se = sd.se > 0 ? (int)(cin * sd.se) : 0;
        eb.skip = eb.stride == 1 && cin == sd.cout;
        std::string q = p + "eff_block." + std::to_string(s) + "." + std::to_string(i) + ".";
        if (sd.type == 0) {
          eb.c0 = b.conv3(q + "conv.weight", eb.cout, cin, 1);
          eb.bn1 = b.bn(q + "bn1", eb.cout, 1e-3f);
        } else if (sd.type == 1) {
          eb.c0 = b.conv3(q + "conv_exp.weight", eb.mid, cin, 1);
          eb.bn1 = b.bn(q + "bn1", eb.mid, 1e-3f);
          eb.c1 = b.dense(q + "conv_pwl.weight", eb.cout, eb.mid, 1, true);
          eb.bn2 = b.bn(q + "bn2", eb.cout, 1e-3f);
        } else {
          eb.c0 = b.dense(q + "conv_pw.weight", eb.mid, cin, 1, true);
          eb.bn1 = b.bn(q + "bn1", eb.mid, 1e-3f);
          eb.dw = b.dwc(q + "conv_dw.weight", eb.mid, 1);
          eb.bn2 = b.bn(q + "bn2", eb.mid, 1e-3f);
          eb.se_r = b.dense(q + "se.conv_reduce.weight", eb.se, eb.mid, 1, true);
          eb.se_rb = b.vec(q + "se.conv_reduce.bias", eb.se, 3, eb.mid);
          eb.se_e = b.dense(q + "se.conv_expand.weight", eb.mid, eb.se, 1, true);
          eb.se_eb = b.vec(q + "se.conv_expand.bias", eb.mid, 3, eb.se);
          eb.c1 = b.dense(q + "conv_pwl.weight", eb.cout, eb.mid, 1, true);
          eb.bn3 = b.bn(q + "bn3", eb.cout, 1e-3f);
        }
        m->blocks.push_back(eb);
        cin = sd.cout;
      }
    }
    m->conv_last = b.dense(p + "conv_last.weight", D, 256, 1, true);
    m->bn_last = b.bn(p + "bn2", D, 1e-5f);
  }
  if (cfg.network == 2) {
    // SwinTransformer (networks/SWIN.py:590-755) in torch's state_dict order: a module's own parameters, then its buffers,
    // then its children.  init 9 = trunc_normal_(std 0.02) (every nn.Linear weight, the bias tables, the position embedding:
    // :137,:665,:707-710), 7 / 8 = the relative_position_index / attn_mask buffers (filled by the host mirror; the kernels
    // derive both from the geometry)
    const std::string p = "encoder.";
    const int E = cfg.swin_embed, P = cfg.swin_patch, Cin = cfg.rgb, R0 = cfg.height / P;
    auto lin = [&](const std::string& n, int N, int K, bool bias, Wt* w, Vec* bv) {
      *w = b.dense(n + ".weight", N, K, 9);
      if (bias) *bv = b.vec(n + ".bias", N, 5);
    };
    m->sw_patch.kind = WK_DENSE; m->sw_patch.N = E; m->sw_patch.K = Cin * P * P; m->sw_patch.Co = E; m->sw_patch.Ci = m->sw_patch.K; m->sw_patch.taps = 1; m->sw_patch.ldb = (E + 7) & ~7;
    m->sw_patch.off = b.addp(p + "patch_embed.proj.weight", {E, Cin, P, P}, 1, Cin * P * P, E);
    m->sw_patch_b = b.vec(p + "patch_embed.proj.bias", E, 3, Cin * P * P);
    m->sw_patch_norm = b.ln(p + "patch_embed.norm", E);
    m->sw_ape.n = R0 * R0 * E; m->sw_ape.off = b.addp(p + "absolute_pos_embed", {1, R0 * R0, E}, 9, 0, 0);
    int total_blocks = 0, bi = 0;
    for (int i = 0; i < 4; ++i) total_blocks += cfg.swin_depths[i];
    for (int i = 0; i < 4; ++i) {
      SwinStage st;
      st.dim = E << i; st.res = R0 >> i; st.down = i < 3;
      for (int j = 0; j < cfg.swin_depths[i]; ++j, ++bi) {
        SwinBlock sb;
        sb.dim = st.dim; sb.heads = cfg.swin_heads[i]; sb.res = st.res;
        sb.ws = cfg.swin_window; sb.shift = (j % 2 == 0) ? 0 : cfg.swin_window / 2;
        if (st.res <= sb.ws) { sb.ws = st.res; sb.shift = 0; }  // :253-256
        sb.drop_path = total_blocks > 1 ? cfg.swin_drop_path * (float)bi / (float)(total_blocks - 1) : 0.f;  // torch.linspace (:669-671)
        const std::string q = p + "layers." + std::to_string(i) + ".blocks." + std::to_string(j) + ".";
        const int N = sb.ws * sb.ws, C = sb.dim;
        if (sb.shift > 0) b.addb(q + "attn_mask", {(st.res / sb.ws) * (st.res / sb.ws), N, N}, ST_BUF_F32, 8);
        sb.n1 = b.ln(q + "norm1", C);
        sb.rpb.n = (2 * sb.ws - 1) * (2 * sb.ws - 1) * sb.heads;
        sb.rpb.off = b.addp(q + "attn.relative_position_bias_table", {(2 * sb.ws - 1) * (2 * sb.ws - 1), sb.heads}, 9, 0, 0);
        b.addb(q + "attn.relative_position_index", {N, N}, ST_BUF_I64, 7);
        lin(q + "attn.qkv", 3 * C, C, true, &sb.qkv, &sb.bqkv);
        lin(q + "attn.proj", C, C, true, &sb.proj, &sb.bproj);
        sb.n2 = b.ln(q + "norm2", C);
        lin(q + "mlp.fc1", 4 * C, C, true, &sb.fc1, &sb.b1);
        lin(q + "mlp.fc2", C, 4 * C, true, &sb.fc2, &sb.b2);
        st.blocks.push_back(sb);
      }
      if (st.down) {
        const std::string q = p + "layers." + std::to_string(i) + ".downsample.";
        Vec none;
        lin(q + "reduction", 2 * st.dim, 4 * st.dim, false, &st.dred, &none);
        st.dnorm = b.ln(q + "norm", 4 * st.dim);
      }
      m->swin.push_back(st);
    }
    m->sw_norm = b.ln(p + "norm", E << 3);
    lin(p + "head", cfg.swin_head_classes, E << 3, true, &m->sw_head, &m->sw_head_b);
    m->sw_head.kind = WK_STEM;  // the classification head is never applied (:732-739): a parameter for state_dict parity only, not packed
  }
  if (cfg.network != 2) {
  m->pe_d0 = b.dense("encoder.positional_encoding.dense0.weight", D / 2, D, 0);
  m->pe_b0 = b.vec("encoder.positional_encoding.dense0.bias", D / 2, 3, D);
  m->pe_d1 = b.dense("encoder.positional_encoding.dense1.weight", 2 * D, D / 2, 0);
  m->pe_b1 = b.vec("encoder.positional_encoding.dense1.bias", 2 * D, 3, D / 2);
  }
  const int Fe = cfg.enc_filter;
  for (int l = 0; l < (cfg.network == 2 ? 0 : cfg.enc_layers); ++l) {
    std::string q = "encoder.attention_layers." + std::to_string(l) + ".";
    EncLayer el;
    el.norm = b.ln(q + "norm", D);
    el.att = b.mha(q + "attention_layer", D, D, cfg.enc_heads);
    el.conv0 = b.dense(q + "conv0.weight", Fe, D, 0, true);
    el.norm0 = b.bn(q + "norm0", Fe, 1e-5f);
    el.dw = b.dwc(q + "depthwise.weight", Fe, 0);
    el.dwb = b.vec(q + "depthwise.bias", Fe, 3, 9);
    el.dwnorm = b.bn(q + "depthwise_norm", Fe, 1e-5f);
    el.conv1 = b.dense(q + "conv1.weight", D, Fe, 0, true);
    el.norm1 = b.bn(q + "norm1", D, 1e-5f);
    m->enc.push_back(el);
  }
  const int Dd = cfg.dec_hidden, Ds = cfg.dec_src, Ff = cfg.dec_filter, V = cfg.num_classes;
  if (cfg.network == 2) b.winit = 2;
  m->embed.kind = WK_STEM; m->embed.N = V + 1; m->embed.K = Dd;
  m->embed.off = b.addp("decoder.embedding.weight", {V + 1, Dd}, 6, 0, 0);
  for (int l = 0; l < cfg.dec_layers; ++l) {
    std::string q = "decoder.attention_layers." + std::to_string(l) + ".";
    DecLayer dl;
    dl.self_att = b.mha(q + "self_attention_layer", Dd, Dd, cfg.dec_heads);
    dl.ln1 = b.ln(q + "self_attention_norm", Dd);
    dl.cross_att = b.mha(q + "attention_layer", Dd, Ds, cfg.dec_heads);
    dl.cross_att.cross = true;
    dl.cross_att.qkv.N = Dd; dl.cross_att.qkv.ldb = Dd; dl.cross_att.qkv.Co = Dd; dl.cross_att.bqkv.n = Dd;
    dl.ln2 = b.ln(q + "attention_norm", Dd);
    // SWIN.py's Feedforward is an nn.Sequential (:826-838): state_dict names layers.0 / layers.3 instead of linear0 / linear1
    const std::string f0 = cfg.network == 2 ? "feedforward_layer.layers.0" : "feedforward_layer.linear0";
    const std::string f1 = cfg.network == 2 ? "feedforward_layer.layers.3" : "feedforward_layer.linear1";
    dl.lin0 = b.dense(q + f0 + ".weight", Ff, Dd, b.winit);
    dl.b0 = b.vec(q + f0 + ".bias", Ff, 3, Dd);
    dl.lin1 = b.dense(q + f1 + ".weight", Dd, Ff, b.winit);
    dl.b1 = b.vec(q + f1 + ".bias", Dd, 3, Ff);
    dl.ln3 = b.ln(q + "feedforward_norm", Dd);
    m->dec.push_back(dl);
  }
  m->gen = b.dense("decoder.generator.weight", V, Dd, 2);
  m->gen_b = b.vec("decoder.generator.bias", V, 3, Dd);

  // registries (pointers into the now-stable containers)
  auto regw = [&](Wt& w) { m->all_w.push_back(&w); };
  auto regv = [&](Vec& v) { m->all_v.push_back(&v); };
  auto regbn = [&](BNp& x) { m->all_bn.push_back(&x); regv(x.w); regv(x.b); };
  auto regln = [&](LNp& x) { regv(x.w); regv(x.b); };
  auto regmha = [&](MHAp& a) { regw(a.qkv); regw(a.kv); regw(a.qonly); regv(a.bqkv); regv(a.bkv); regv(a.bq); regw(a.out); regv(a.bout); };
  if (cfg.network == 2) {
    regw(m->sw_patch); regv(m->sw_patch_b); regln(m->sw_patch_norm); regv(m->sw_ape);
    for (auto& st : m->swin) {
      for (auto& sb : st.blocks) {
        regln(sb.n1); regln(sb.n2); regv(sb.rpb); regw(sb.qkv); regv(sb.bqkv); regw(sb.proj); regv(sb.bproj);
        regw(sb.fc1); regv(sb.b1); regw(sb.fc2); regv(sb.b2);
      }
      if (st.down) { regw(st.dred); regln(st.dnorm); }
    }
    regln(m->sw_norm); regw(m->sw_head); regv(m->sw_head_b);
  } else if (cfg.network == 0) {
    for (auto& w : m->lite_conv) regw(w);
    for (auto& x : m->lite_bn) regbn(x);
  } else {
    regw(m->stem); regbn(m->stem_bn);
    for (auto& eb : m->blocks) {
      regw(eb.c0); regbn(eb.bn1);
      if (eb.type >= 1) { regw(eb.c1); regbn(eb.bn2); }
      if (eb.type == 2) { regw(eb.dw); regw(eb.se_r); regw(eb.se_e); regv(eb.se_rb); regv(eb.se_eb); regbn(eb.bn3); }
    }
    regw(m->conv_last); regbn(m->bn_last);
  }
  if (cfg.network != 2) { regw(m->pe_d0); regw(m->pe_d1); regv(m->pe_b0); regv(m->pe_b1); }
  for (auto& el : m->enc) {
    regln(el.norm); regmha(el.att); regw(el.conv0); regw(el.conv1); regw(el.dw); regv(el.dwb);
    regbn(el.norm0); regbn(el.dwnorm); regbn(el.norm1);
  }
  regw(m->embed);
  for (auto& dl : m->dec) {
    regmha(dl.self_att); regmha(dl.cross_att); regln(dl.ln1); regln(dl.ln2); regln(dl.ln3);
    regw(dl.lin0); regw(dl.lin1); regv(dl.b0); regv(dl.b1);
  }
  regw(m->gen); regv(m->gen_b);

  // persistent-region layout: packed weights first
  size_t esz = cfg.dtype == DT_BF16 ? 2 : 4;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t a = (o + 255) & ~(size_t)255; o = a + bytes; return a; };
  for (Wt* w : m->all_w) {
    if (w->kind == WK_DENSE) {
      w->pk_fwd_off = take((size_t)w->N * w->K * esz);
      w->pk_bwd_off = take((size_t)w->K * w->ldb * esz);
    } else if (w->kind == WK_CONV3) {
      w->pk_fwd_off = take((size_t)w->N * w->K * esz);
      w->pk_bwd_off = take((size_t)w->Ci * 9 * w->Co * esz);
    } else if (w->kind == WK_DW) {
      w->pk_fwd_off = take((size_t)9 * w->Co * esz);
    }
  }
  m->off_packed = 0;
  m->off_scalars = take(SC_COUNT * 4);
  m->off_sumsq = take(1024 * 4);
  m->off_pe1d = take((size_t)500 * Dd * 4);
  if (cfg.network == 0) { m->feat_h = cfg.height / 16; m->feat_w = cfg.width / 16; }
  else if (cfg.network == 2) { m->feat_h = m->feat_w = (cfg.height / cfg.swin_patch) >> 3; }
  else { m->feat_h = cfg.height / 32; m->feat_w = cfg.width / 32; }
  for (auto& st : m->swin)
    for (auto& sb : st.blocks) {
      if (sb.shift <= 0) continue;
      for (size_t g = 0; g < m->sw_geo.size(); ++g)
        if (m->sw_geo[g].res == sb.res && m->sw_geo[g].ws == sb.ws && m->sw_geo[g].shift == sb.shift) sb.geo = (int)g;
      if (sb.geo < 0) {
        const size_t nW = (size_t)(sb.res / sb.ws) * (sb.res / sb.ws), N = (size_t)sb.ws * sb.ws;
        { const size_t o_mask = take(nW * N * N * 4); m->sw_geo.push_back({sb.res, sb.ws, sb.shift, o_mask, take(nW * N)}); }
        sb.geo = (int)m->sw_geo.size() - 1;
      }
    }
  m->off_hpos = take((size_t)std::max(m->feat_h, 1) * D * 4);
  m->off_wpos = take((size_t)std::max(m->feat_w, 1) * D * 4);
  m->packdesc_bytes = (m->all_w.size() + 8) * sizeof(PackDesc);
  m->off_packdesc = take(m->packdesc_bytes);
  {
    size_t nb = 0;
    for (Wt* w : m->all_w) nb += std::max(((size_t)w->N * w->K + PACK_BLK - 1) / PACK_BLK + 1, (size_t)((w->N + 63) / 64) * ((w->K + 63) / 64));
    m->packblk_bytes = nb * 2 * sizeof(int);
    m->off_packblk = take(m->packblk_bytes);
  }
  {  // eval-mode BatchNorm scale/shift table + its descriptor array (one prepare launch per inference forward)
    size_t tot = 0;
    for (BNp* b : m->all_bn) { b->eval_off = tot; tot += (size_t)2 * b->C; }
    m->off_bn_eval = take(tot * 4);
    m->off_bn_desc = take(m->all_bn.size() * sizeof(BnEvalDesc) + 16);
  }
  // f32 is the parity mode: cross-workgroup float reductions go through fixed-order partial slabs instead of atomics
  // (SATRN_DETERMINISTIC=1 forces it for bf16 too, SATRN_NONDET=1 switches it off)
  if ((cfg.dtype == DT_F32 || getenv("SATRN_DETERMINISTIC")) && !getenv("SATRN_NONDET")) {
    m->det_floats = (size_t)8 << 20;
    m->off_det = take(2 * m->det_floats * sizeof(float));
  }
  if (cfg.dtype == DT_BF16 && !m->det_floats) {
    // partial tiles of one launch of the persistent weight-gradient kernel: at most one 128 x 128 fp32 tile per item, CU-count items (+ slack)
    m->wgpart_floats = (size_t)320 * 128 * 128;
    m->off_wgpart = take(2 * m->wgpart_floats * sizeof(float));
    // mailbox of launch_bn_pool_se (starts zeroed with the workspace; only that kernel writes it, with a new tag per launch)
    m->off_sebox = take((size_t)Model::SEBOX_IMAGES * (1536 + 64) * 8);
    // mailbox of the MBConv block kernels (kernels_mbconv.hip): three exchanges of (C / 64 <= 24 slabs) x images x 128 sums
    m->off_mbbox = take(Model::MBBOX_WORDS * 8);
  }
  m->zero_bytes = 40u << 20;
  m->off_zero = take(m->zero_bytes);
  m->persist_bytes = (o + 255) & ~(size_t)255;
  m->ex = new Exec();
  m->ex->m = m;
  if (!sw_off("side_stream")) m->ex->s2 = side_stream_get();

  segment_ranges(m);
  return m;
}

void model_destroy(Model* m) {
  if (!m) return;
  for (int i = 0; i < 8; ++i) if (m->graphs[i]) (void)hipGraphExecDestroy(m->graphs[i]);
  if (m->hy_pinned) (void)hipHostFree(m->hy_pinned);
  delete m->ex;
  delete m;
}

int model_bind(Model* m, float* params, float* grads, float* buf_f32, int64_t* buf_i64) {
  m->params = params; m->grads = grads; m->buf_f32 = buf_f32; m->buf_i64 = buf_i64;
  for (Wt* w : m->all_w) { w->p = params + w->off; w->g = grads ? grads + w->off : nullptr; }
  for (Vec* v : m->all_v) { v->p = params + v->off; v->g = grads ? grads + v->off : nullptr; }
  for (BNp* b : m->all_bn) { b->rm = buf_f32 + b->rm_off; b->rv = buf_f32 + b->rv_off; b->nbt = buf_i64 ? buf_i64 + b->nbt_off : nullptr; }
  m->bound = true;
  m->pack_dirty = true;
  m->bn_desc_dirty = true;
  for (int i = 0; i < 8; ++i) if (m->graphs[i]) { (void)hipGraphExecDestroy(m->graphs[i]); m->graphs[i] = nullptr; }
  if (m->decode_graph) { (void)hipGraphExecDestroy(m->decode_graph); m->decode_graph = nullptr; }
  return 0;
}

static float* scal(Model* m) { return (float*)(m->ws + m->off_scalars); }

int model_bind_optimizer(Model* m, float* exp_avg, float* exp_avg_sq) {
  m->adam_m = exp_avg; m->adam_v = exp_avg_sq;
  for (int i = 0; i < 8; ++i) if (m->graphs[i]) { (void)hipGraphExecDestroy(m->graphs[i]); m->graphs[i] = nullptr; }
  return 0;
}

// dropout RNG state (one uint32 on the device, advanced once per training step): read (set == 0) or written
int model_rng_state(Model* m, uint32_t* seed_io, int set, hipStream_t s) {
  if (!m->ws_set) { m->err = "set a workspace first"; return -1; }
  if (set) (void)hipMemcpyAsync(m->ws + m->off_scalars + SC_SEED * 4, seed_io, 4, hipMemcpyHostToDevice, s);
  else (void)hipMemcpyAsync(seed_io, m->ws + m->off_scalars + SC_SEED * 4, 4, hipMemcpyDeviceToHost, s);
  (void)hipStreamSynchronize(s);
  return 0;
}

int model_set_workspace(Model* m, void* ws, size_t bytes, hipStream_t s) {
  if (bytes < m->persist_bytes + (1u << 20)) { m->err = "workspace too small"; return -2; }
  // a REPLACEMENT workspace (grown for a larger batch / longer sequence) continues the run: the dropout seed is carried
  // over from the old one (the caller keeps it alive until this call returns); Adam's moments and step count are not in
  // the workspace at all
  uint32_t seed0 = 0x1234567u;
  if (m->ws_set && m->ws && m->ws != (char*)ws) {
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(&seed0, m->ws + m->off_scalars + SC_SEED * 4, 4, hipMemcpyDeviceToHost);
  }
  m->ws = (char*)ws; m->ws_bytes = bytes;
  // packed weights carry zero padding (generator: 245 -> 256 columns) that kernels multiply with zero gradients:
  // the padding must be finite, so clear everything once
  (void)hipMemsetAsync(m->ws, 0, bytes, s);
  m->zero_hwm = 0; m->ex->zoff = 0;  // fresh, all-zero workspace
  m->bn_desc_dirty = true;
  (void)hipStreamSynchronize(s);
  for (Wt* w : m->all_w) {
    w->fwd = w->pk_fwd_off >= 0 ? m->ws + w->pk_fwd_off : nullptr;
    w->bwd = w->pk_bwd_off >= 0 ? m->ws + w->pk_bwd_off : nullptr;
  }
  for (int i = 0; i < 8; ++i) if (m->graphs[i]) { (void)hipGraphExecDestroy(m->graphs[i]); m->graphs[i] = nullptr; }
  if (m->decode_graph) { (void)hipGraphExecDestroy(m->decode_graph); m->decode_graph = nullptr; }
  // tables: 1-D PE (networks/EfficientSATRN.py:408-418) and 2-D PE (:111-127), built in fp32 like the reference
  const int Dd = m->cfg.dec_hidden, D = m->cfg.enc_hidden;
  std::vector<float> pe((size_t)500 * Dd);
  for (int pos = 0; pos < 500; ++pos)
    for (int i = 0; i < Dd; ++i) {
      float rate = 1.0f / powf(10000.0f, (float)(2 * (i / 2)) / (float)Dd);
      float a = (float)pos * rate;
      pe[(size_t)pos * Dd + i] = (i & 1) ? cosf(a) : sinf(a);
    }
  (void)hipMemcpyAsync(m->ws + m->off_pe1d, pe.data(), pe.size() * 4, hipMemcpyHostToDevice, s);
  auto tab2d = [&](int len, size_t off) {
    std::vector<float> t((size_t)std::max(len, 1) * D);
    int nts = D / 2;
    float inc = logf(1.0e4f / 1.0f) / ((float)nts - 1.0f);
    for (int pos = 0; pos < len; ++pos)
      for (int k = 0; k < nts; ++k) {
        float inv = expf((float)k * -inc);
        float st = (float)pos * inv;
        t[(size_t)pos * D + k] = sinf(st);
        t[(size_t)pos * D + nts + k] = cosf(st);
      }
    (void)hipMemcpyAsync(m->ws + off, t.data(), t.size() * 4, hipMemcpyHostToDevice, s);
    (void)hipStreamSynchronize(s);
  };
  (void)hipStreamSynchronize(s);
  tab2d(m->feat_h, m->off_hpos);
  tab2d(m->feat_w, m->off_wpos);
  for (auto& g : m->sw_geo) {
    // networks/SWIN.py:288-309: region ids 0..8 of the SHIFTED map (h / w slices [0, -ws), [-ws, -shift), [-shift, end)),
    // window-partitioned; mask[w][i][j] = -100 where the two tokens come from different regions
    const int nWw = g.res / g.ws, N = g.ws * g.ws;
    std::vector<float> mk((size_t)nWw * nWw * N * N);
    auto region = [&](int v) { return v < g.res - g.ws ? 0 : (v < g.res - g.shift ? 1 : 2); };
    for (int wy = 0; wy < nWw; ++wy)
      for (int wx = 0; wx < nWw; ++wx)
        for (int i = 0; i < N; ++i)
          for (int j = 0; j < N; ++j) {
            const int idi = 3 * region(wy * g.ws + i / g.ws) + region(wx * g.ws + i % g.ws);
            const int idj = 3 * region(wy * g.ws + j / g.ws) + region(wx * g.ws + j % g.ws);
            mk[(((size_t)wy * nWw + wx) * N + i) * N + j] = idi != idj ? -100.0f : 0.0f;
          }
    (void)hipMemcpyAsync(m->ws + g.off, mk.data(), mk.size() * 4, hipMemcpyHostToDevice, s);
    // the same information as one byte per token: its region id (the attention kernel compares two ids instead of reading mk)
    std::vector<unsigned char> lab((size_t)nWw * nWw * N);
    for (int wy = 0; wy < nWw; ++wy)
      for (int wx = 0; wx < nWw; ++wx)
        for (int i = 0; i < N; ++i) lab[((size_t)wy * nWw + wx) * N + i] = (unsigned char)(3 * region(wy * g.ws + i / g.ws) + region(wx * g.ws + i % g.ws));
    (void)hipMemcpyAsync(m->ws + g.off_lab, lab.data(), lab.size(), hipMemcpyHostToDevice, s);
    (void)hipStreamSynchronize(s);
  }
  (void)hipMemsetAsync(m->ws + m->off_scalars, 0, SC_COUNT * 4, s);
  float one = 1.0f;
  (void)hipMemcpyAsync(scal(m) + SC_ONE, &one, 4, hipMemcpyHostToDevice, s);
  (void)hipMemcpyAsync(scal(m) + SC_SEED, &seed0, 4, hipMemcpyHostToDevice, s);
  (void)hipStreamSynchronize(s);
  m->ws_set = true;
  m->pack_dirty = true;
  return 0;
}

int model_pack_weights(Model* m, hipStream_t s) {
  if (!m->bound || !m->ws_set) { m->err = "bind + workspace first"; return -1; }
  const int dt = m->cfg.dtype;
  if (m->pack_dirty) {  // (re)build the descriptor table: one entry per packed weight
    std::vector<PackDesc> d;
    long total = 0;
    for (Wt* w : m->all_w) {
      PackDesc e;
      e.src = w->p; e.fwd = w->fwd; e.bwd = w->bwd; e.start = total; e.ldb = w->ldb;
      if (w->kind == WK_DENSE) { e.kind = 0; e.N = w->N; e.K = w->K; total += (long)w->N * w->K; }
      else if (w->kind == WK_CONV3) { e.kind = 1; e.N = w->Co; e.K = w->Ci; total += (long)w->Co * w->Ci * 9; }
      else if (w->kind == WK_DW) { e.kind = 2; e.N = w->Co; e.K = 1; total += (long)w->Co * 9; }
      else continue;
      d.push_back(e);
    }
    if (d.size() * sizeof(PackDesc) > m->packdesc_bytes) { m->err = "pack descriptor table overflow"; return -1; }
    std::vector<int> blk;  // (descriptor, chunk) pairs
    for (size_t i = 0; i < d.size(); ++i) {
      if (d[i].kind == 0) {  // 64x64 tiles
        long nt = (long)((d[i].N + 63) / 64) * ((d[i].K + 63) / 64);
        for (long c = 0; c < nt; ++c) { blk.push_back((int)i); blk.push_back((int)c); }
      } else {
        long n = d[i].kind == 1 ? (long)d[i].N * d[i].K * 9 : (long)d[i].N * 9;
        for (long c = 0; c * PACK_BLK < n; ++c) { blk.push_back((int)i); blk.push_back((int)c); }
      }
    }
    if (blk.size() * sizeof(int) > m->packblk_bytes) { m->err = "pack block table overflow"; return -1; }
    (void)hipMemcpyAsync(m->ws + m->off_packdesc, d.data(), d.size() * sizeof(PackDesc), hipMemcpyHostToDevice, s);
    (void)hipMemcpyAsync(m->ws + m->off_packblk, blk.data(), blk.size() * sizeof(int), hipMemcpyHostToDevice, s);
    (void)hipStreamSynchronize(s);
    m->pack_n = (int)(blk.size() / 2); m->pack_total = total; m->pack_dirty = false;
  }
  launch_pack_all(dt, (const PackDesc*)(m->ws + m->off_packdesc), m->ws + m->off_packblk, m->pack_n, s);
  return 0;
}

// =====================================================================================================
// ops (forward launch + recorded backward)
// =====================================================================================================
namespace {
struct Geo { int H, W, Ci, OH, OW, KW, stride, pt, pl; };

static inline void used(Tensor* t) { if (t) t->ncons++; }
static constexpr bool g_fuse_bnb = true, g_fuse_actb = true, g_fuse_bn_eval = true;   // (round-2 A/B switches, retired)

static void acc_grad(Exec& e, Tensor* t, const void* src) {
  // first contribution: alias the producer's gradient buffer (it has no reader left once its own backward ran)
  if (!t->g) { t->g = const_cast<void*>(src); t->g_init = true; return; }
  int beta;
  void* g = e.grad(t, &beta);
  long n = t->rows * t->C;
  e.nbytes = (double)n * e.esz() * (beta ? 3 : 2);
  if (beta) LCH(e, launch_add(e.dt, g, src, g, n, e.s));
  else LCH(e, (void)hipMemcpyAsync(g, src, (size_t)n * e.esz(), hipMemcpyDeviceToDevice, e.s));
}

// y[M][N] = act(gather(x) * W^T + bias) (+dropout).  geo == nullptr: dense over x rows.
Tensor* op_gemm(Exec& e, Tensor* x, Wt* w, Vec* bias, int act, float drop_p, const Geo* geo, int B = 0, bool out_f32 = false,
                void* out_ptr = nullptr, bool want_stats = false) {
  const long M = geo ? (long)B * geo->OH * geo->OW : x->rows;
  const int N = w->N;
  Tensor* y;
  if (out_ptr) {
    e.tens.emplace_back(new Tensor());
    y = e.tens.back().get();
    y->rows = M; y->C = N; y->f32 = out_f32; y->p = out_ptr;
  } else {
    y = e.newt(M, N, B, geo ? geo->OH : 0, geo ? geo->OW : 0, out_f32);
  }
  if (!e.train) drop_p = 0.f;
  const uint32_t site = drop_p > 0.f ? e.site++ : 0;
  e.last_site = site; e.last_drop = drop_p;
  const uint32_t* seed = (const uint32_t*)(scal(e.m) + SC_SEED);
  GemmP p;
  memset(&p, 0, sizeof(p));
  p.A = x->p; p.Bw = w->fwd; p.C = y->p; p.bias = bias ? bias->p : nullptr;
  p.M = (int)M; p.N = N; p.K = w->K; p.lda = x->C; p.ldc = N;
  p.act = act; p.out_f32 = out_f32 ? 1 : 0; p.drop_p = drop_p; p.seed = seed; p.site = site;
  if (act == ACT_GELU && e.rec && !geo && !out_f32 && !out_ptr) {
    // GELU's derivative needs the pre-activation -- but only through act'(u), which the epilogue has almost for free beside the
    // activation itself: that factor is what is kept (product + activation pass in one launch, the backward multiplies)
    y->act_pre = e.alloc((size_t)M * N * e.esz()); y->act_kind = ACT_DFACTOR;
    p.pre_out = y->act_pre; p.pre_grad = 1;
  }
  if (act == ACT_RELU && e.rec && !geo && !out_f32 && !out_ptr) {
    // ReLU (+dropout): the derivative is read off the stored output (zeros = clipped or dropped), no second tensor needed
    y->act_pre = y->p; y->act_kind = ACT_RELU; y->act_scale = 1.0f / (1.0f - drop_p);
  }
  if (want_stats && e.train) {
    // tall, narrow outputs (early backbone stages): many row tiles hit the same 2N addresses -> spread over replicas
    const int rep = stats_rep_for(N, M);  // deterministic mode folds into ONE [2N]
    y->stats = e.zalloc((size_t)rep * 2 * N); y->stats_rep = rep; p.stats = y->stats; p.stats_rep = rep;
  }
  if (geo) { p.H = geo->H; p.W = geo->W; p.Ci = geo->Ci; p.OH = geo->OH; p.OW = geo->OW; p.KW = geo->KW; p.stride = geo->stride; p.pt = geo->pt; p.pl = geo->pl; }
  e.nflops = 2.0 * (double)M * N * w->K; e.nbytes = ((double)x->rows * x->C + (double)M * N + (double)N * w->K) * e.esz();
  if (want_stats && !e.train && !e.rec && g_fuse_bn_eval && !bias && act == ACT_NONE && !out_f32 && !out_ptr) {
    // inference: the BatchNorm that follows (eval statistics) runs in this product's epilogue -- op_bn_act launches it
    y->pend = std::make_shared<GemmP>(p); y->pend_mode = geo ? AM_CONV : AM_DENSE;
    used(x);
    return y;
  }
  LCH(e, launch_gemm(e.dt, geo ? AM_CONV : AM_DENSE, p, e.s));
  // x is a BatchNorm output and this is its first consumer: our dgrad is the last writer of x's gradient, so its
  // epilogue can also produce that BatchNorm's backward column sums
  const bool fuse_bnb = g_fuse_bnb && e.rec && x->bn_y && x->ncons == 0 && (geo ? geo->Ci : w->K) == x->C;
  used(x);
  if (e.rec) {
    Geo g{};
    if (geo) g = *geo;
    const bool hasgeo = geo != nullptr;
    e.tape.push_back([&e, x, y, w, bias, act, drop_p, site, seed, g, hasgeo, M, N, B, out_f32, fuse_bnb]() {
      if (!y->g) return;
      const int ldy = out_f32 ? w->ldb : N;
      void* dY = y->g;
      WORK(e, 0, (double)M * N * e.esz() * (act == ACT_NONE ? 2 : 3));
      if (act == ACT_RELU) { if (!y->g_preact) LCH(e, launch_act_bwd(e.dt, dY, y->p, dY, M * N, ACT_RELU, drop_p, e.s)); }
      else if (act == ACT_GELU) {
        if (!y->act_pre) { e.m->err = "internal: GELU epilogue without its pre-activation"; e.oom = true; return; }
        if (!y->g_preact) LCH(e, launch_act_bwd(e.dt, dY, y->act_pre, dY, M * N, ACT_DFACTOR, 0.f, e.s));   // else: the consumer's dgrad epilogue did it
      }
      else if (act == ACT_SIGMOID) LCH(e, launch_act_bwd(e.dt, dY, y->p, dY, M * N, ACT_SIGMOID, 0.f, e.s));
      else if (drop_p > 0.f) LCH(e, launch_dropout_bwd(e.dt, dY, dY, M, N, drop_p, seed, site, e.s));
      // y's BatchNorm backward-apply held back (MBConv projection, see Tensor::bn_bwd_hold_ok): dY does not exist yet
      const bool bnheld = y->bhold.armed && y->bn_bwd_hold_ok;
      auto flush_bn = [&e, y]() {
        BnBwdHold& h = y->bhold;
        if (!h.armed) return;
        h.armed = false;
        WORK(e, 0, (double)h.M * h.C * e.esz() * 3);
        LCH(e, launch_bn_bwd_apply(e.dt, h.dz, h.y, h.ss, h.mr, h.w, h.red, h.M, h.C, h.act, h.dy, h.dwp, h.dbp, e.s, h.rep, nullptr, nullptr, 0, 0));
      };
      const bool can_hold_dgrad = !hasgeo && x->se_out && !fuse_bnb && !out_f32 && ldy == N && !e.dry && e.dt == DT_BF16 && !x->g && x->ncons == 1;
      if (bnheld && !can_hold_dgrad) flush_bn();
      WgradP q;
      memset(&q, 0, sizeof(q));
      q.dY = dY; q.A = x->p; q.dW = w->g; q.M = (int)M; q.N = N; q.K = w->K; q.ldy = ldy; q.lda = x->C;
      q.nbatch = 1; q.nb_inner = 1;
      q.full_grid = (e.serial || !e.s2 || e.prof) ? 1 : 0;
      float* tmp = nullptr;
      if (hasgeo) {
        q.conv = 1; q.H = g.H; q.W = g.W; q.Ci = g.Ci; q.OH = g.OH; q.OW = g.OW; q.KW = g.KW; q.stride = g.stride; q.pt = g.pt; q.pl = g.pl;
        // contiguous fp32 atomics into a zeroed [N][9][Ci] scratch, then one small pass into the torch layout
        tmp = e.zalloc((size_t)N * w->K);
        q.conv_packed_out = 1; q.dW = tmp;
      }
      auto do_wgrad = [&e, q, tmp, bias, w, g, hasgeo, M, N, dY, ldy, x]() mutable {
        const int dt = e.dt; const bool dry = e.dry;
        float* bg = bias ? bias->g : nullptr; float* wg = w->g; const int Ci = g.Ci;
        // dense products: the weight-gradient kernel sums the bias gradient from the dY chunks it stages anyway (not in the deterministic mode,
        // whose fixed-order column sums stay a pass of their own)
        const bool fold_db = !sw_off("wgrad_bias");   // read per call (A/B in one process)
        if (bg && fold_db && !hasgeo && !g_det.on) { q.dbias = bg; bg = nullptr; }
        if (e.prof || dry) {
          WORK(e, 0, (double)M * N * e.esz());
          if (bg) LCH(e, launch_colsum(dt, dY, M, N, ldy, bg, e.s));
          WORK(e, 2.0 * (double)M * N * w->K, ((double)M * N + (double)x->rows * x->C) * e.esz() + (double)N * w->K * 4);
          LCH(e, launch_wgrad(dt, q, e.s));
          if (tmp) { WORK(e, 0, (double)N * w->K * 12); LCH(e, launch_conv_grad_unpack(tmp, wg, N, Ci, 9, e.s)); }
        } else {
          e.defer([=](hipStream_t ws) {
            if (bg) launch_colsum(dt, dY, M, N, ldy, bg, ws);
            launch_wgrad(dt, q, ws);
            if (tmp) launch_conv_grad_unpack(tmp, wg, N, Ci, 9, ws);
          });
          if (tmp) e.flush_side();  // the heavy 3x3 weight gradients of the last backward stages start at once (join wait 0.19 -> 0.08 ms)
        }
      };
      const bool defer_wgrad = bnheld && y->bhold.armed;   // dY is produced by the block's backward launch: the weight gradient is issued behind it
      if (!defer_wgrad) do_wgrad();
      int beta;
      void* dx = e.grad(x, &beta);
      GemmP d;
      memset(&d, 0, sizeof(d));
      d.A = dY; d.Bw = w->bwd; d.C = dx; d.beta = beta;
      if (fuse_bnb) {
        const int rep = stats_rep_for(x->C, x->rows);
        x->bn_red = e.zalloc((size_t)rep * 2 * x->C); x->bn_red_rep = rep;
        d.stats = x->bn_red; d.stats_rep = rep; d.bnb_y = x->bn_y; d.bnb_ss = x->bn_ss; d.bnb_mr = x->bn_mr; d.bnb_act = x->bn_act;
      }
      if (!hasgeo && !fuse_bnb && !beta && x->act_pre && x->ncons == 1 && !x->g_preact && g_fuse_actb) {
        // x = act(u) feeds this product only: dx leaves the epilogue as du = dx * act'(u)
        d.bact_u = x->act_pre; d.bact = x->act_kind; d.bact_scale = x->act_scale; x->g_preact = true;
      }
      if (!hasgeo) {
        d.M = (int)M; d.N = w->K; d.K = w->ldb; d.lda = ldy; d.ldc = x->C;
        e.nflops = 2.0 * (double)d.M * d.N * w->N;
        e.nbytes = ((double)d.M * w->N + (double)d.M * d.N * (beta ? 2 : 1) + (double)w->N * d.N) * e.esz();  // dY + dX (+old dX) + W
        if (x->se_out && !beta && !fuse_bnb && !d.bact_u && !out_f32 && ldy == N && !e.dry && e.dt == DT_BF16) {
          // x is a squeeze-and-excite output: its closure (next) runs this product together with its own kernels where it can
          x->dgrad_hold = std::make_shared<GemmP>(d); x->dgrad_hold_flops = e.nflops; x->dgrad_hold_bytes = e.nbytes;
          if (defer_wgrad) { x->bhold = y->bhold; y->bhold.armed = false; x->after_fused = do_wgrad; }
          e.nflops = 0; e.nbytes = 0;
          return;
        }
        if (defer_wgrad) { const double fl = e.nflops, by = e.nbytes; flush_bn(); do_wgrad(); WORK(e, fl, by); }
        LCH(e, launch_gemm(e.dt, AM_DENSE, d, e.s));
      } else {
        d.M = (int)((long)B * g.H * g.W); d.N = g.Ci; d.K = 9 * w->Co; d.ldc = g.Ci;
        d.H = g.OH; d.W = g.OW; d.Ci = w->Co; d.OH = g.H; d.OW = g.W; d.KW = g.KW; d.stride = g.stride; d.pt = g.pt; d.pl = g.pl;
        e.nflops = 2.0 * (double)M * N * w->K;
        e.nbytes = ((double)M * N + (double)d.M * d.N * (beta ? 2 : 1) + (double)N * w->K) * e.esz();
        LCH(e, launch_gemm(e.dt, AM_DGRAD, d, e.s));
      }
    });
  }
  return y;
}

// a BatchNorm+activation whose launch is taken over by the depthwise convolution that consumes it (launch_bn_dwconv): op_bn_act
// does all its bookkeeping (tensors, tape) and leaves the operands here instead of launching
struct BnHold {
  bool armed = false;
  const void* y = nullptr; const float* sums = nullptr; int rep = 1; BNp* bn = nullptr; float* ss = nullptr; float* mr = nullptr; void* z = nullptr;
  long M = 0; int C = 0, act = 0;
};
// the BatchNorm in front of a squeeze-and-excite block, held back like BnHold: op_se launches both as one kernel (launch_bn_pool_se) or, where
// that does not take the shape, this pass (launch_bn_act_pool) followed by its own kernels
struct SeHold {
  bool armed = false;
  const void* y = nullptr; const float* sums = nullptr; int rep = 1; BNp* bn = nullptr; float* ss = nullptr; float* mr = nullptr; void* z = nullptr;
  float* pool = nullptr; long M = 0; int C = 0, HW = 0, act = 0;
  // inference: the depthwise convolution in front (Tensor::pend_dw), held back with the eval-mode BatchNorm's scale / shift
  std::function<int(const float*, const float*, int, void*, float*, void*, const SeEvalArgs*)> dwfn; const float* esc = nullptr; const float* esh = nullptr;
};
Tensor* op_bn_act(Exec& e, Tensor* y, BNp* bn, int act, Tensor* res, float** pool_out = nullptr, BnHold* hold = nullptr, SeHold* sehold = nullptr,
                  Exec::FwdBnHold* fhold = nullptr) {
  const int C = bn->C;
  const long M = y->rows;
  if (y->pend) {
    Tensor* z = e.newt(M, C, y->B, y->H, y->W);
    GemmP p = *y->pend;
    const float* ev = (const float*)(e.m->ws + e.m->off_bn_eval) + bn->eval_off;
    p.C = z->p; p.ldc = C; p.escale = ev; p.eshift = ev + C; p.act = act; p.eres = res ? res->p : nullptr;
    e.nflops = 2.0 * (double)p.M * p.N * p.K; e.nbytes = ((double)M * C * (res ? 2 : 1) + (double)p.N * p.K) * e.esz();
    LCH(e, launch_gemm(e.dt, y->pend_mode, p, e.s));
    y->pend.reset();
    used(res);
    return z;
  }
  if (y->pend_dw) {
    auto fn = y->pend_dw;
    y->pend_dw = nullptr;
    if (!res) {
      Tensor* z = e.newt(M, C, y->B, y->H, y->W);
      const float* ev = (const float*)(e.m->ws + e.m->off_bn_eval) + bn->eval_off;
      e.nbytes = (double)M * C * e.esz() * 2;
      float* pool = pool_out && y->B > 0 ? (float*)e.alloc((size_t)y->B * C * 4) : nullptr;
      if (sehold && pool && !e.dry) {   // a squeeze-and-excite block follows: op_se launches the depthwise kernel, with its own part where it can
        sehold->armed = true; sehold->dwfn = fn; sehold->esc = ev; sehold->esh = ev + C; sehold->act = act; sehold->z = z->p; sehold->pool = pool;
        *pool_out = pool;
        return z;
      }
      if (fn(ev, ev + C, act, z->p, pool, nullptr, nullptr) == 1) *pool_out = pool;
      return z;
    }
    fn(nullptr, nullptr, 0, y->p, nullptr, nullptr, nullptr);  // a residual is added by the separate pass below: run the plain convolution first
  }
  float* ss = (float*)e.alloc((size_t)2 * C * 4);
  float* mr = (float*)e.alloc((size_t)2 * C * 4);
  e.last_bn_ss = ss; e.last_bn_mr = mr;
  float* sums = nullptr;
  if (e.train) {
    sums = y->stats;
    if (!sums) {  // producer without a fused statistics epilogue (the stem conv)
      sums = e.zalloc(2 * C);
      e.nbytes = (double)M * C * e.esz();
      LCH(e, launch_colstats(e.dt, y->p, M, C, sums, e.s));
    }
  }
  Tensor* z = e.newt(M, C, y->B, y->H, y->W);
  e.nbytes = (double)M * C * e.esz() * (res ? 3 : 2);
  if (pool_out && e.train && !res && sums && y->B > 0 && bn_act_pool_ok(M, C, y->H * y->W)) {
    // a squeeze-and-excite block follows: its average pool is accumulated here (zeroed [B][C] sums)
    *pool_out = e.zalloc((size_t)y->B * C);
    if (sehold && !e.dry) {   // op_se launches this pass, fused with its own where the shape allows
      sehold->armed = true; sehold->y = y->p; sehold->sums = sums; sehold->rep = y->stats ? y->stats_rep : 1; sehold->bn = bn; sehold->ss = ss; sehold->mr = mr;
      sehold->z = z->p; sehold->pool = *pool_out; sehold->M = M; sehold->C = C; sehold->HW = y->H * y->W; sehold->act = act;
      e.nflops = 0; e.nbytes = 0;
    } else
    LCH(e, launch_bn_act_pool(e.dt, y->p, sums, y->stats ? y->stats_rep : 1, bn->w.p, bn->b.p, bn->rm, bn->rv, bn->nbt, bn->eps, 0.1f, ss, mr, z->p,
                              *pool_out, M, C, y->H * y->W, act, e.s));
  } else if (hold && e.train && sums && !res) {
    hold->armed = true; hold->y = y->p; hold->sums = sums; hold->rep = y->stats ? y->stats_rep : 1; hold->bn = bn; hold->ss = ss; hold->mr = mr;
    hold->z = z->p; hold->M = M; hold->C = C; hold->act = act;
    e.nflops = 0; e.nbytes = 0;
  } else if (fhold && e.train && sums && act == ACT_NONE && !e.dry && !e.nolaunch) {
    // the consumer (the next MBConv block's front) normalises while it stages its input
    *fhold = Exec::FwdBnHold{true, y->p, res ? res->p : nullptr, sums, y->stats ? y->stats_rep : 1, bn, ss, mr, z->p, M, C};
    e.nflops = 0; e.nbytes = 0;
  } else
  LCH(e, launch_bn_act(e.dt, y->p, sums, y->stats ? y->stats_rep : 1, bn->w.p, bn->b.p, bn->rm, bn->rv, e.train ? bn->nbt : nullptr, bn->eps, 0.1f, ss, mr,
                       res ? res->p : nullptr, z->p, M, C, act, e.s));
  used(y); used(res);
  if (e.rec) {
    if (e.train) { z->bn_y = y->p; z->bn_ss = ss; z->bn_mr = mr; z->bn_act = act; z->bn_has_res = res != nullptr; z->bn_src = y; z->bn_p = bn; }
    const int eval_stats = e.train ? 0 : 1;  // recorded forward with running statistics: dy = dz * act' * w * rstd, no batch terms
    e.tape.push_back([&e, y, z, bn, act, res, ss, mr, M, C, eval_stats]() {
      if (!z->g) return;
      if (z->bn_applied) return;   // the kernel that produced z's gradient went on through this BatchNorm (op_dwconv's closure, no residual)
      float* red = z->bn_red;  // already produced by the epilogue of the last kernel that wrote z's gradient?
      if (!red) {
        red = e.zalloc(2 * C);
        e.nbytes = (double)M * C * e.esz() * 2;
        LCH(e, launch_bn_bwd_reduce(e.dt, z->g, y->p, ss, mr, M, C, act, red, e.s, z->se_gate, z->se_dpool, z->se_hw));
      }
      void* dy = e.grad(y, nullptr);
      if (y->dw_bwd_fuse && !eval_stats && (z->bn_red ? z->bn_red_rep : 1) == 1 && !e.dry) {
        // the depthwise convolution that produced y runs this pass inside its data-gradient kernel (its closure is next)
        BnBwdHold& h = y->bhold;
        h.armed = true; h.dz = z->g; h.y = y->p; h.ss = ss; h.mr = mr; h.w = bn->w.p; h.red = red; h.M = M; h.C = C; h.act = act; h.dy = dy;
        h.dwp = bn->w.g; h.dbp = bn->b.g; h.se_gate = z->se_gate; h.se_dpool = z->se_dpool; h.se_hw = z->se_hw;
      } else if (y->bn_bwd_hold_ok && !eval_stats && act == ACT_NONE && !z->se_gate && !e.dry && e.dt == DT_BF16 && sw_knob("mbconv_dfold", 0) != 0) {
        // (OFF by default, SATRN_KNOBS=mbconv_dfold=1 switches it on: measured flat -- 8.72 vs 8.70-8.72 ms per step, same box -- the 24 / 15 workgroups of an
        // image each repeat the image's pass: the block launch grows by 5-9 us where the 5 us launch disappears)
        // y is an MBConv block's projection output: its closure (next) and the squeeze-and-excite closure behind it run this pass inside the
        // block's backward launch (launch_mbconv_bwd_se with MbDinArgs), or launch it themselves
        BnBwdHold& h = y->bhold;
        h.armed = true; h.dz = z->g; h.y = y->p; h.ss = ss; h.mr = mr; h.w = bn->w.p; h.red = red; h.M = M; h.C = C; h.act = act; h.dy = dy;
        h.dwp = bn->w.g; h.dbp = bn->b.g; h.se_gate = nullptr; h.se_dpool = nullptr; h.se_hw = 0; h.rep = z->bn_red ? z->bn_red_rep : 1;
      } else {
        e.nbytes = (double)M * C * e.esz() * 3;
        LCH(e, launch_bn_bwd_apply(e.dt, z->g, y->p, ss, mr, bn->w.p, red, M, C, act, dy, bn->w.g, bn->b.g, e.s, z->bn_red ? z->bn_red_rep : 1,
                                   z->se_gate, z->se_dpool, z->se_hw, eval_stats));
      }
      if (res) acc_grad(e, res, z->g);
    });
  }
  return z;
}

Tensor* op_stem(Exec& e, const float* img, Wt* w, int B, int Cin, int H, int W, int stride, int pad) {
  const int OH = (H + 2 * pad - 3) / stride + 1, OW = (W + 2 * pad - 3) / stride + 1;
  Tensor* y = e.newt((long)B * OH * OW, w->Co, B, OH, OW);
  WORK(e, 2.0 * (double)B * OH * OW * w->Co * 9 * Cin, (double)B * Cin * H * W * 4 + (double)B * OH * OW * w->Co * e.esz());
  LCH(e, launch_stem_conv(e.dt, img, w->p, y->p, B, Cin, H, W, w->Co, OH, OW, stride, pad, e.s));
  if (e.rec)
    e.tape.push_back([&e, img, w, y, B, Cin, H, W, OH, OW, stride, pad]() {
      if (!y->g) return;
      {
        const int dt = e.dt; void* yg = y->g; float* wg = w->g; const int Co = w->Co;
        WORK(e, 2.0 * (double)B * OH * OW * Co * 9 * Cin, (double)B * Cin * H * W * 4 + (double)B * OH * OW * Co * e.esz());
        if (e.prof || e.dry) LCH(e, launch_stem_wgrad(dt, img, yg, wg, B, Cin, H, W, Co, OH, OW, stride, pad, e.s));
        else e.defer([=](hipStream_t ws) { launch_stem_wgrad(dt, img, yg, wg, B, Cin, H, W, Co, OH, OW, stride, pad, ws); });
      }
    });
  return y;
}

Tensor* op_dwconv(Exec& e, Tensor* x, Wt* w, Vec* bias, int stride, int OH, int OW, int pt, int pl, bool want_stats = true, BnHold* hold = nullptr) {
  // x is a BatchNorm output and this is its first consumer: the data gradient below is the last writer of x's gradient and can
  // reduce that BatchNorm's backward column sums on the way (as the dgrad GEMM epilogue does, op_gemm)
  const bool fuse_bnb = g_fuse_bnb && e.rec && x->bn_y && !x->bn_has_res && x->ncons == 0 && stride == 1 && pt == 1 && pl == 1 && OH == x->H && OW == x->W;
  used(x);
  const int B = x->B, H = x->H, W = x->W, C = x->C;
  Tensor* y = e.newt((long)B * OH * OW, C, B, OH, OW);
  if (want_stats && e.train) y->stats = e.zalloc(2 * C);
  const bool hold_bwd = !sw_off("bn_apply_dw");   // read per call: tests toggle it in one process
  y->dw_bwd_fuse = hold_bwd && e.rec && e.train && stride == 1 && pt == 1 && pl == 1 && OH == H && OW == W && dwconv_img_ok(e.dt, H, W, C);
  if (want_stats && !e.train && !e.rec && g_fuse_bn_eval) {
    // inference: the BatchNorm that follows runs in this kernel's epilogue (launched by op_bn_act)
    Exec* ep = &e;
    const void* xp = x->p; const void* wf = w->fwd; const float* bp = bias ? bias->p : nullptr;
    // -> 0: out written, no pool; 1: out + pool sums; 2: se_out = out * gate written (the squeeze-and-excite block ran in the same launch)
    y->pend_dw = [=](const float* esc, const float* esh, int act, void* out, float* pool, void* se_out, const SeEvalArgs* se) -> int {
      WORK((*ep), 18.0 * (double)B * OH * OW * C, ((double)B * H * W + (double)B * OH * OW) * C * ep->esz());
      // small maps: whole image x 64 channels per workgroup, which also leaves the squeeze-and-excite pool complete
      bool img = false;
      if (esc && stride == 1 && pt == 1 && pl == 1 && OH == H && OW == W) {
        if (se && se_out) {
          LCH((*ep), img = launch_dwconv_eval_img(ep->dt, xp, wf, bp, esc, esh, act, se_out, nullptr, B, H, W, C, ep->s, se));
          if (img) return 2;
        }
        LCH((*ep), img = launch_dwconv_eval_img(ep->dt, xp, wf, bp, esc, esh, act, out, pool, B, H, W, C, ep->s));
      }
      if (img) return pool != nullptr ? 1 : 0;
      LCH((*ep), launch_dwconv(ep->dt, 0, xp, wf, bp, out, B, H, W, C, OH, OW, stride, pt, pl, 0, nullptr, ep->s, esc, esh, act));
      return 0;
    };
    return y;
  }
  bool fused = false;
  if (hold && hold->armed) {
    // the BatchNorm in front was held back: both in one launch where the shape allows (whole image x 64 channels per workgroup)
    BNp* bn = hold->bn;
    if (y->stats && stride == 1 && pt == 1 && pl == 1 && OH == H && OW == W && !e.dry) {
      WORK(e, 18.0 * (double)B * OH * OW * C, (double)B * H * W * C * e.esz() * 3);
      LCH(e, fused = launch_bn_dwconv(e.dt, hold->y, hold->sums, hold->rep, bn->w.p, bn->b.p, bn->rm, bn->rv, bn->nbt, bn->eps, 0.1f, hold->ss, hold->mr,
                                      hold->z, w->fwd, bias ? bias->p : nullptr, y->p, y->stats, B, H, W, C, hold->act, e.s));
    }
    if (!fused) {
      WORK(e, 0, (double)hold->M * hold->C * e.esz() * 2);
      LCH(e, launch_bn_act(e.dt, hold->y, hold->sums, hold->rep, bn->w.p, bn->b.p, bn->rm, bn->rv, bn->nbt, bn->eps, 0.1f, hold->ss, hold->mr, nullptr,
                           hold->z, hold->M, hold->C, hold->act, e.s));
    }
    hold->armed = false;
  }
  if (!fused) {
    // the statistics pass behind it reads y once more
    WORK(e, 18.0 * (double)B * OH * OW * C, ((double)B * H * W + (double)B * OH * OW * (y->stats ? 2 : 1)) * C * e.esz());
    LCH(e, launch_dwconv(e.dt, 0, x->p, w->fwd, bias ? bias->p : nullptr, y->p, B, H, W, C, OH, OW, stride, pt, pl, 0, y->stats, e.s));
  }
  if (e.rec)
    e.tape.push_back([&e, x, y, w, bias, B, H, W, C, OH, OW, stride, pt, pl, fuse_bnb]() {
      if (!y->g) return;
      float* scr = e.zalloc((size_t)10 * C);
      // data gradient first: with a held BatchNorm pass (y->bhold) this launch is what PRODUCES y->g, which the weight gradient reads
      int beta;
      void* dx = e.grad(x, &beta);
      BnBwdHold* hold = y->bhold.armed ? &y->bhold : nullptr;
      bool fused = false;
      if (fuse_bnb && hold && !beta && !e.dry && !x->bn_red && x->bn_src && x->bn_p && e.train) {
        // ... and through the BatchNorm in front as well: x's gradient never leaves the kernel, the gradient of that BatchNorm's input does
        Tensor* src = x->bn_src;
        const bool had = src->g != nullptr;
        BnBwdTail tl;
        tl.dy = e.grad(src, nullptr); tl.w = x->bn_p->w.p; tl.dwp = x->bn_p->w.g; tl.dbp = x->bn_p->b.g;
        WORK(e, 18.0 * (double)B * OH * OW * C, ((double)B * OH * OW * 4 + (double)B * H * W * 2) * C * e.esz());
        if (!had) LCH(e, fused = launch_dwconv_bwd_bn(e.dt, y->g, w->fwd, dx, 0, x->bn_y, x->bn_ss, x->bn_mr, x->bn_act, nullptr, B, H, W, C, e.s, hold, &tl));
        if (fused) x->bn_applied = true;
      }
      if (!fused && (fuse_bnb || hold) && !e.dry && !x->bn_red) {
        float* red = fuse_bnb ? e.zalloc((size_t)2 * C) : nullptr;
        WORK(e, 18.0 * (double)B * OH * OW * C, ((double)B * OH * OW * (hold ? 4 : 2) + (double)B * H * W * (beta ? 2 : 1)) * C * e.esz());
        LCH(e, fused = launch_dwconv_bwd_bn(e.dt, y->g, w->fwd, dx, beta, fuse_bnb ? x->bn_y : nullptr, x->bn_ss, x->bn_mr, x->bn_act, red, B, H, W, C, e.s, hold));
        if (fused && fuse_bnb) { x->bn_red = red; x->bn_red_rep = 1; }
      }
      if (!fused && hold) {
        WORK(e, 0, (double)hold->M * hold->C * e.esz() * 3);
        LCH(e, launch_bn_bwd_apply(e.dt, hold->dz, hold->y, hold->ss, hold->mr, hold->w, hold->red, hold->M, hold->C, hold->act, hold->dy, hold->dwp, hold->dbp, e.s, 1,
                                   hold->se_gate, hold->se_dpool, hold->se_hw, 0));
        if (fuse_bnb && !e.dry && !x->bn_red) {
          float* red = e.zalloc((size_t)2 * C);
          WORK(e, 18.0 * (double)B * OH * OW * C, ((double)B * OH * OW * 2 + (double)B * H * W * (beta ? 2 : 1)) * C * e.esz());
          LCH(e, fused = launch_dwconv_bwd_bn(e.dt, y->g, w->fwd, dx, beta, x->bn_y, x->bn_ss, x->bn_mr, x->bn_act, red, B, H, W, C, e.s));
          if (fused) { x->bn_red = red; x->bn_red_rep = 1; }
        }
      }
      y->bhold.armed = false;
      if (!fused) {
        WORK(e, 18.0 * (double)B * OH * OW * C, ((double)B * OH * OW + (double)B * H * W * (beta ? 2 : 1)) * C * e.esz());
        LCH(e, launch_dwconv(e.dt, 1, y->g, w->fwd, nullptr, dx, B, OH, OW, C, H, W, stride, pt, pl, beta, nullptr, e.s));
      }
      {
        const int dt = e.dt; void* xp = x->p; void* yg = y->g; float* wg = w->g; float* bg = bias ? bias->g : nullptr;
        WORK(e, 18.0 * (double)B * OH * OW * C, ((double)B * H * W + (double)B * OH * OW) * C * e.esz());
        if (e.prof || e.dry) LCH(e, launch_dwconv_wgrad(dt, xp, yg, wg, bg, scr, B, H, W, C, OH, OW, stride, pt, pl, e.s));
        else e.defer([=](hipStream_t ws) { launch_dwconv_wgrad(dt, xp, yg, wg, bg, scr, B, H, W, C, OH, OW, stride, pt, pl, ws); });
      }
    });
  return y;
}

Tensor* op_maxpool(Exec& e, Tensor* x) {
  used(x);
  const int B = x->B, H = x->H, W = x->W, C = x->C;
  Tensor* y = e.newt((long)B * (H / 2) * (W / 2), C, B, H / 2, W / 2);
  WORK(e, 0, (double)(x->rows + y->rows) * C * e.esz());
  LCH(e, launch_maxpool(e.dt, 0, x->p, nullptr, y->p, B, H, W, C, e.s));
  if (e.rec)
    e.tape.push_back([&e, x, y, B, H, W, C]() {
      if (!y->g) return;
      void* dx = e.grad(x, nullptr);
      if ((H & 1) || (W & 1)) LCH(e, launch_fill(dx, 0, (size_t)x->rows * C * e.esz(), e.s));
      WORK(e, 0, (double)(2 * x->rows + y->rows) * C * e.esz());
      LCH(e, launch_maxpool(e.dt, 1, x->p, y->g, dx, B, H, W, C, e.s));
    });
  return y;
}

Tensor* op_pool(Exec& e, Tensor* x) {  // mean over HW -> [B][C]
  used(x);
  const int B = x->B, HW = x->H * x->W, C = x->C;
  Tensor* y = e.newt(B, C);
  WORK(e, 0, (double)x->rows * C * e.esz());
  LCH(e, launch_pool_hw(e.dt, x->p, y->p, B, HW, C, e.s));
  if (e.rec)
    e.tape.push_back([&e, x, y, B, HW, C]() {
      if (!y->g) return;
      int beta;
      void* dx = e.grad(x, &beta);
      if (!beta) LCH(e, launch_fill(dx, 0, (size_t)x->rows * C * e.esz(), e.s));
      WORK(e, 0, (double)x->rows * C * e.esz() * 2);
      LCH(e, launch_bcast_add_hw(e.dt, y->g, dx, B, HW, C, 1.0f / (float)HW, e.s));
    });
  return y;
}

Tensor* op_act(Exec& e, Tensor* u, int act) {
  used(u);
  Tensor* z = e.newt(u->rows, u->C);
  WORK(e, 0, (double)u->rows * u->C * e.esz() * 2);
  LCH(e, launch_act_fwd(e.dt, u->p, z->p, u->rows * u->C, act, e.s));
  if (e.rec)
    e.tape.push_back([&e, u, z, act]() {
      if (!z->g) return;
      void* du = e.grad(u, nullptr);
      WORK(e, 0, (double)u->rows * u->C * e.esz() * 3);
      LCH(e, launch_act_bwd(e.dt, z->g, u->p, du, u->rows * u->C, act, 0.f, e.s));
    });
  return z;
}

Tensor* op_se_scale(Exec& e, Tensor* x, Tensor* gate) {
  used(x); used(gate);
  const int B = x->B, HW = x->H * x->W, C = x->C;
  Tensor* y = e.newt(x->rows, C, B, x->H, x->W);
  WORK(e, 0, (double)x->rows * C * e.esz() * 2);
  LCH(e, launch_se_scale(e.dt, x->p, gate->p, y->p, B, HW, C, e.s));
  if (e.rec)
    e.tape.push_back([&e, x, gate, y, B, HW, C]() {
      if (!y->g) return;
      void* dg = e.grad(gate, nullptr);
      WORK(e, 0, (double)x->rows * C * e.esz() * 2);
      LCH(e, launch_se_bwd_gate(e.dt, y->g, x->p, dg, B, HW, C, e.s));
      int beta;
      void* dx = e.grad(x, &beta);
      WORK(e, 0, (double)x->rows * C * e.esz() * (beta ? 3 : 2));
      LCH(e, launch_se_bwd_x(e.dt, y->g, gate->p, nullptr, dx, B, HW, C, beta, e.s));
    });
  return y;
}

Tensor* op_posenc_apply(Exec& e, Tensor* x, Tensor* gate) {
  used(x); used(gate);
  Model* m = e.m;
  const int B = x->B, H = x->H, W = x->W, C = x->C;
  const float* hpos = (const float*)(m->ws + m->off_hpos);
  const float* wpos = (const float*)(m->ws + m->off_wpos);
  Tensor* y = e.newt(x->rows, C, B, H, W);
  WORK(e, 0, (double)x->rows * C * e.esz() * 2);
  LCH(e, launch_posenc2d(e.dt, x->p, gate->p, hpos, wpos, y->p, B, H, W, C, e.s));
  if (e.rec)
    e.tape.push_back([&e, x, gate, y, hpos, wpos, B, H, W, C]() {
      if (!y->g) return;
      void* dg = e.grad(gate, nullptr);
      WORK(e, 0, (double)x->rows * C * e.esz());
      LCH(e, launch_posenc2d_bwd(e.dt, y->g, hpos, wpos, dg, B, H, W, C, e.s));
      acc_grad(e, x, y->g);
    });
  return y;
}

// LayerNorm backward: dx on the main stream; the per-block (dw | dbias) partials are folded on the side stream (optimizer-only sums)
static void ln_backward(Exec& e, const void* dy, const void* a, const void* b, LNp* ln, const float* mr, void* da, void* db, int ba, int bb,
                        long R, int C, RowMap dmap = RowMap(), LnAdd add = LnAdd()) {
  const int g = layernorm_bwd_blocks(R);
  float* part = (float*)e.alloc((size_t)g * 2 * C * sizeof(float));
  LCH(e, launch_layernorm_bwd(e.dt, dy, a, b, ln->w.p, mr, da, db, ba, bb, ln->w.g, ln->b.g, R, C, 0.f, nullptr, 0, e.s, part, dmap, add));
  float* dw = ln->w.g; float* dbias = ln->b.g;
  if (e.prof || e.dry) {
    LCH(e, launch_layernorm_fold(part, g, C, dw, dbias, e.s));
  } else {
    e.defer([=](hipStream_t s2) { launch_layernorm_fold(part, g, C, dw, dbias, s2); });
  }
}

// map: the output (and its gradient) are kept in window order (RowMap): LayerNorm + shift / window partition in one pass
Tensor* op_ln(Exec& e, Tensor* a, Tensor* b, LNp* ln, RowMap map = RowMap()) {
  used(a); used(b);
  const long R = a->rows;
  const int C = a->C;
  Tensor* y = e.newt(R, C, a->B, a->H, a->W);
  float* mr = (float*)e.alloc((size_t)2 * R * 4);
  e.last_mr = mr;
  WORK(e, 0, (double)R * C * e.esz() * (b ? 3 : 2));
  LCH(e, launch_layernorm(e.dt, a->p, b ? b->p : nullptr, ln->w.p, ln->b.p, y->p, mr, R, C, 1e-5f, 0.f, nullptr, 0, e.s, map));
  if (e.rec)
    e.tape.push_back([&e, a, b, y, ln, mr, R, C, map]() {
      if (!y->g) return;
      int ba = 0, bb = 0;
      void* da = e.grad(a, &ba);
      void* db = b ? e.grad(b, &bb) : nullptr;
      WORK(e, 0, (double)R * C * e.esz() * ((b ? 5 : 3) + ba + (b ? bb : 0)));
      ln_backward(e, y->g, a->p, b ? b->p : nullptr, ln, mr, da, db, ba, bb, R, C, map);
    });
  return y;
}

// sum = a + DropPath(b) (b read through bmap: window order), y = LayerNorm(sum) written through out_map -- the residual add folded into the
// LayerNorm behind it (networks/SWIN.py:283-300: `x = shortcut + self.drop_path(x)` then `self.norm2(x)`; likewise the MLP's residual and the
// next block's norm1).  The backward hands d sum (its own LayerNorm backward + what later consumers of the sum left in sum->g) to a and,
// scaled and re-ordered, to b.  Same stochastic-depth site sequence as op_droppath_add.
struct AddLn { Tensor* sum; Tensor* y; };
AddLn op_add_ln(Exec& e, Tensor* a, Tensor* b, int B, float p, RowMap bmap, LNp* ln, RowMap out_map) {
  used(a); used(b);
  if (!e.train) p = 0.f;
  const uint32_t site = p > 0.f ? e.site++ : 0;
  const uint32_t* seed = (const uint32_t*)(scal(e.m) + SC_SEED);
  const long R = a->rows;
  const int C = a->C;
  Tensor* sum = e.newt(R, C, B, a->H, a->W);
  Tensor* y = e.newt(R, C, B, a->H, a->W);
  float* mr = (float*)e.alloc((size_t)2 * R * 4);
  e.last_mr = mr;
  LnAdd add;
  add.on = 1; add.sum_out = sum->p; add.bmap = bmap; add.drop_p = p; add.seed = seed; add.site = site; add.rows_per_sample = R / B;
  WORK(e, 0, (double)R * C * e.esz() * 4);
  LCH(e, launch_layernorm(e.dt, a->p, b->p, ln->w.p, ln->b.p, y->p, mr, R, C, 1e-5f, 0.f, nullptr, 0, e.s, out_map, add));
  if (e.rec)
    e.tape.push_back([&e, a, b, sum, y, ln, mr, R, C, out_map, add]() {
      if (!y->g) {
        if (sum->g) { e.m->err = "internal: add+LayerNorm whose normalised output has no gradient"; e.oom = true; }
        return;
      }
      int ba = 0, bb = 0;
      void* da = e.grad(a, &ba);
      void* db = e.grad(b, &bb);
      if (bb) { e.m->err = "internal: add+LayerNorm branch has two consumers"; e.oom = true; return; }
      LnAdd q = add;
      q.sum_out = nullptr; q.gsum = sum->g;
      WORK(e, 0, (double)R * C * e.esz() * (5 + ba + (sum->g ? 1 : 0)));
      ln_backward(e, y->g, sum->p, nullptr, ln, mr, da, db, ba, 0, R, C, out_map, q);
    });
  return {sum, y};
}

Tensor* op_quirk(Exec& e, Tensor* yv) {  // networks/EfficientSATRN.py:269
  used(yv);
  const int B = yv->B, HW = yv->H * yv->W, C = yv->C;
  Tensor* z = e.newt(yv->rows, C, B, yv->H, yv->W);
  WORK(e, 0, (double)yv->rows * C * e.esz() * 2);
  LCH(e, launch_reshape_quirk(e.dt, 0, yv->p, z->p, B, HW, C, 0, e.s));
  if (e.rec)
    e.tape.push_back([&e, yv, z, B, HW, C]() {
      if (!z->g) return;
      int beta;
      void* dy = e.grad(yv, &beta);
      WORK(e, 0, (double)yv->rows * C * e.esz() * (beta ? 3 : 2));
      LCH(e, launch_reshape_quirk(e.dt, 1, z->g, dy, B, HW, C, beta, e.s));
    });
  return z;
}

// attention over column slices of projection outputs: Q = qt[:, qoff:qoff+D], K = kvt[:, koff:], V = kvt[:, voff:]
Tensor* op_attn(Exec& e, Tensor* qt, int qoff, Tensor* kvt, int koff, int voff, int B, int Lq, int Lk, int heads, int D,
                int causal, const int64_t* text, int ld_text, float drop_p) {
  used(qt); used(kvt);
  const int hd = D / heads;
  Tensor* o = e.newt((long)B * Lq, D, B);
  float* lse = (float*)e.alloc((size_t)B * heads * Lq * 4);
  if (!e.train) drop_p = 0.f;
  const uint32_t site = drop_p > 0.f ? e.site++ : 0;
  e.last_lse = lse; e.last_site = site; e.last_drop = drop_p;
  const uint32_t* seed = (const uint32_t*)(scal(e.m) + SC_SEED);
  const size_t es = e.esz();
  AttnP p;
  memset(&p, 0, sizeof(p));
  p.Q = (char*)qt->p + qoff * es; p.K = (char*)kvt->p + koff * es; p.V = (char*)kvt->p + voff * es; p.O = o->p; p.lse = lse;
  p.text = text; p.ld_text = ld_text; p.B = B; p.H = heads; p.Lq = Lq; p.Lk = Lk; p.hd = hd;
  p.ldq = qt->C; p.ldk = kvt->C; p.ldv = kvt->C; p.ldo = D;
  p.sq_b = (long)Lq * qt->C; p.sk_b = (long)Lk * kvt->C; p.sv_b = (long)Lk * kvt->C; p.so_b = (long)Lq * D;
  p.causal = causal; p.pad_id = e.m->cfg.pad_id; p.inv_temp = 1.0f / sqrtf((float)D); p.drop_p = drop_p; p.seed = seed; p.site = site;
  WORK(e, 4.0 * (double)B * heads * Lq * Lk * hd, ((double)B * Lq * D * 2 + (double)B * Lk * D * 2) * e.esz());
  LCH(e, launch_attn(e.dt, 0, p, e.s));
  if (e.rec)
    e.tape.push_back([&e, qt, qoff, kvt, koff, voff, o, p, B, Lq, Lk, heads, hd, D, es]() {
      if (!o->g) return;
      const size_t LkP = attn_lkp(Lk);
      int bq = 0, bkv = 0;
      void* dq = e.grad(qt, &bq);
      void* dkv = qt == kvt ? dq : e.grad(kvt, &bkv);  // K/V shared by several attention calls (step decoder): accumulate
      (void)bq;
      if (attn2_ok(e.dt, p)) {
        // short sequences: dQ, dK, dV finished inside one workgroup per (batch, head) -- no dS / Pd tensors, no batched products
        AttnP q = p;
        q.dO = o->g; q.dQ = (char*)dq + qoff * es; q.dK = (char*)dkv + koff * es; q.dV = (char*)dkv + voff * es; q.kv_accum = bkv;
        WORK(e, 14.0 * (double)B * heads * Lq * Lk * hd, ((double)B * Lq * D * 4 + (double)B * Lk * D * 4) * es);
        LCH(e, launch_attn2_bwd(q, e.s));
        return;
      }
      void* dS = e.alloc((size_t)B * heads * Lq * LkP * es);
      void* Pd = e.alloc((size_t)B * heads * Lq * LkP * es);
      AttnP q = p;
      q.dO = o->g; q.dQ = (char*)dq + qoff * es; q.dS = dS; q.Pd = Pd;
      // reads q, k, v, o, dO; writes dQ and the two [B, heads, Lq, LkP] probability / score-gradient tensors
      WORK(e, 6.0 * (double)B * heads * Lq * Lk * hd, ((double)B * Lq * D * 4 + (double)B * Lk * D * 2 + 2.0 * B * heads * Lq * LkP) * es);
      LCH(e, launch_attn(e.dt, 1, q, e.s));
      WgradP w;
      memset(&w, 0, sizeof(w));
      w.M = Lq; w.N = Lk; w.K = hd; w.ldy = (int)LkP; w.out_t = 1; w.out_accum = bkv; w.nbatch = B * heads; w.nb_inner = heads;
      w.sY_o = (long)heads * Lq * LkP; w.sY_i = (long)Lq * LkP;
      w.sW_o = (long)Lk * kvt->C; w.sW_i = hd; w.ldw = kvt->C;
      // dV = Pd^T dO
      WORK(e, 2.0 * (double)B * heads * Lq * Lk * hd, ((double)B * heads * Lq * LkP + (double)B * Lq * D + (double)B * Lk * D) * es);
      w.dY = Pd; w.A = o->g; w.lda = D; w.sA_o = (long)Lq * D; w.sA_i = hd; w.dW = (char*)dkv + voff * es;
      LCH(e, launch_wgrad(e.dt, w, e.s));
      // dK = dS^T Q
      WORK(e, 2.0 * (double)B * heads * Lq * Lk * hd, ((double)B * heads * Lq * LkP + (double)B * Lq * D + (double)B * Lk * D) * es);
      w.dY = dS; w.A = (char*)qt->p + qoff * es; w.lda = qt->C; w.sA_o = (long)Lq * qt->C; w.sA_i = hd; w.dW = (char*)dkv + koff * es;
      LCH(e, launch_wgrad(e.dt, w, e.s));
    });
  return o;
}

Tensor* op_embed(Exec& e, const int64_t* ids, int ld_ids, int B, int L, int pos0, float drop_p) {
  Model* m = e.m;
  const int D = m->cfg.dec_hidden;
  Tensor* y = e.newt((long)B * L, D, B);
  if (!e.train) drop_p = 0.f;
  const uint32_t site = drop_p > 0.f ? e.site++ : 0;
  const uint32_t* seed = (const uint32_t*)(scal(m) + SC_SEED);
  const float* pe = (const float*)(m->ws + m->off_pe1d);
  WORK(e, 0, (double)B * L * D * (4 + e.esz()));
  LCH(e, launch_embed(e.dt, ids, m->embed.p, pe, y->p, B, L, ld_ids, D, pos0, drop_p, seed, site, e.s, m->embed.N));
  if (e.rec)
    e.tape.push_back([&e, m, ids, ld_ids, y, B, L, D, drop_p, seed, site]() {
      if (!y->g) return;
      WORK(e, 0, (double)B * L * D * (4 + e.esz()));
      LCH(e, launch_embed_bwd(e.dt, ids, y->g, m->embed.g, B, L, ld_ids, D, drop_p, seed, site, e.s, m->embed.N));
    });
  return y;
}

// ---- SwinTRN ops (networks/SWIN.py) ---------------------------------------------------------------------
// rows permutation: cyclic shift + window partition (reverse = window_reverse + roll back); backward = the inverse permutation
Tensor* op_window_perm(Exec& e, Tensor* x, int B, int H, int W, int ws, int shift, int reverse) {
  used(x);
  const int C = x->C;
  Tensor* y = e.newt(x->rows, C, B, H, W);
  WORK(e, 0, (double)x->rows * C * e.esz() * 2);
  LCH(e, launch_window_perm(e.dt, x->p, y->p, B, H, W, C, ws, shift, reverse, 0, e.s));
  if (e.rec)
    e.tape.push_back([&e, x, y, B, H, W, C, ws, shift, reverse]() {
      if (!y->g) return;
      int beta;
      void* dx = e.grad(x, &beta);
      WORK(e, 0, (double)x->rows * C * e.esz() * (beta ? 3 : 2));
      LCH(e, launch_window_perm(e.dt, y->g, dx, B, H, W, C, ws, shift, !reverse, beta, e.s));
    });
  return y;
}

Tensor* op_patch_merge(Exec& e, Tensor* x, int B, int H, int W) {
  used(x);
  const int C = x->C;
  Tensor* y = e.newt(x->rows / 4, 4 * C, B, H / 2, W / 2);
  WORK(e, 0, (double)x->rows * C * e.esz() * 2);
  LCH(e, launch_patch_merge(e.dt, x->p, y->p, B, H, W, C, 0, 0, e.s));
  if (e.rec)
    e.tape.push_back([&e, x, y, B, H, W, C]() {
      if (!y->g) return;
      int beta;
      void* dx = e.grad(x, &beta);
      WORK(e, 0, (double)x->rows * C * e.esz() * (beta ? 3 : 2));
      LCH(e, launch_patch_merge(e.dt, y->g, dx, B, H, W, C, 1, beta, e.s));
    });
  return y;
}

// out = shortcut + DropPath(branch): per-sample keep mask, scale 1/(1-p) (timm DropPath; identity in eval)
// map: the branch (and its gradient) are in window order (RowMap): window reverse / shift back folded into the residual add
Tensor* op_droppath_add(Exec& e, Tensor* shortcut, Tensor* branch, int B, float p, RowMap map = RowMap()) {
  used(shortcut); used(branch);
  if (!e.train) p = 0.f;
  const uint32_t site = p > 0.f ? e.site++ : 0;
  const uint32_t* seed = (const uint32_t*)(scal(e.m) + SC_SEED);
  const int C = branch->C;
  const long per = branch->rows / B * C;
  Tensor* y = e.newt(branch->rows, C, B, shortcut->H, shortcut->W);
  WORK(e, 0, (double)branch->rows * C * e.esz() * 3);
  LCH(e, launch_droppath(e.dt, 0, shortcut->p, branch->p, y->p, B, per, p, seed, site, e.s, map, C));
  if (e.rec)
    e.tape.push_back([&e, shortcut, branch, y, B, per, C, p, seed, site, map]() {
      if (!y->g) return;
      int beta;
      void* db = e.grad(branch, &beta);
      if (beta) { e.m->err = "internal: drop-path branch has two consumers"; e.oom = true; return; }
      WORK(e, 0, (double)branch->rows * C * e.esz() * 2);
      LCH(e, launch_droppath(e.dt, 1, y->g, nullptr, db, B, per, p, seed, site, e.s, map, C));
      acc_grad(e, shortcut, y->g);
    });
  return y;
}

// x [B][L][C] + table [L][C] (absolute position embedding, fp32 parameter); d table = sum over the batch
Tensor* op_add_table(Exec& e, Tensor* x, Vec* table, int B) {
  used(x);
  const int C = x->C;
  const long LC = x->rows / B * C;
  Tensor* y = e.newt(x->rows, C, B, x->H, x->W);
  WORK(e, 0, (double)x->rows * C * e.esz() * 2);
  LCH(e, launch_add_rows_table(e.dt, x->p, table->p, y->p, B, LC, e.s));
  if (e.rec)
    e.tape.push_back([&e, x, y, table, B, LC]() {
      if (!y->g) return;
      const int dt = e.dt; void* yg = y->g; float* tg = table->g;
      if (e.prof || e.dry) { WORK(e, 0, (double)B * LC * e.esz()); LCH(e, launch_colsum(dt, yg, B, (int)LC, (int)LC, tg, e.s)); }
      else e.defer([=](hipStream_t ws) { launch_colsum(dt, yg, B, (int)LC, (int)LC, tg, ws); });
      acc_grad(e, x, y->g);
    });
  return y;
}

// window attention over a fused qkv tensor [B_*N][3C] (q | k | v, head h at columns h*hd): softmax(q k^T * hd^-0.5 + bias[h] +
// mask[window]) v  (networks/SWIN.py:152-190); the relative-position-bias table is gathered to [heads][N][N] per call
Tensor* op_window_attn(Exec& e, Tensor* qkv, SwinBlock* sb, int B_, const float* wmask, int nW) {
  used(qkv);
  const int N = sb->ws * sb->ws, C = sb->dim, heads = sb->heads, hd = C / heads;
  Tensor* o = e.newt((long)B_ * N, C, B_);
  float* lse = (float*)e.alloc((size_t)B_ * heads * N * 4);
  // relative position bias + shifted-window mask: computed inside the attention kernel from the table parameter and one region
  // label per token (the first form gathered the bias into [heads][N][N] and read it and the [nW][N][N] mask per score: 10.4 vs
  // 7.1 ms of attention per step in round 2)
  constexpr bool bias_tensor = false;
  float* bias = nullptr;
  const size_t es = e.esz();
  AttnP p;
  memset(&p, 0, sizeof(p));
  p.Q = qkv->p; p.K = (char*)qkv->p + (size_t)C * es; p.V = (char*)qkv->p + (size_t)2 * C * es; p.O = o->p; p.lse = lse;
  p.B = B_; p.H = heads; p.Lq = N; p.Lk = N; p.hd = hd;
  p.ldq = p.ldk = p.ldv = 3 * C; p.ldo = C;
  p.sq_b = p.sk_b = p.sv_b = (long)N * 3 * C; p.so_b = (long)N * C;
  p.inv_temp = 1.0f / sqrtf((float)hd); p.pad_id = e.m->cfg.pad_id;
  p.nW = nW > 0 ? nW : 1;
  if (bias_tensor) { p.bias = bias; p.wmask = wmask; }
  else { p.rel_table = sb->rpb.p; p.rel_ws = sb->ws; p.labels = wmask ? (const unsigned char*)(e.m->ws + e.m->sw_geo[sb->geo].off_lab) : nullptr; }
  WORK(e, 4.0 * (double)B_ * heads * N * N * hd, (double)B_ * N * C * 4 * es);
  LCH(e, launch_attn(e.dt, 0, p, e.s));
  if (e.rec)
    e.tape.push_back([&e, qkv, o, sb, p, B_, N, C, heads, hd, es]() {
      if (!o->g) return;
      const size_t LkP = attn_lkp(N);
      int bq = 0;
      void* dq = e.grad(qkv, &bq);
      if (bq) { e.m->err = "internal: qkv tensor has two consumers"; e.oom = true; return; }
      void* dS = e.alloc((size_t)B_ * heads * N * LkP * es);
      const bool fused = attn2_ok(e.dt, p);
      if (fused) {
        // dQ, dK and dV in one launch (kernels_attn2.hip); it also leaves the raw-score gradient for the table gradient below
        AttnP q = p;
        q.dO = o->g; q.dQ = dq; q.dK = (char*)dq + (size_t)C * es; q.dV = (char*)dq + (size_t)2 * C * es; q.kv_accum = 0; q.dS = dS;
        WORK(e, 14.0 * (double)B_ * heads * N * N * hd, ((double)B_ * N * C * 8 + (double)B_ * heads * N * LkP) * es);
        LCH(e, launch_attn2_bwd(q, e.s));
      }
      void* Pd = fused ? nullptr : e.alloc((size_t)B_ * heads * N * LkP * es);
      AttnP q = p;
      if (!fused) {
      q.dO = o->g; q.dQ = dq; q.dS = dS; q.Pd = Pd;
      WORK(e, 6.0 * (double)B_ * heads * N * N * hd, ((double)B_ * N * C * 6 + 2.0 * B_ * heads * N * LkP) * es);
      LCH(e, launch_attn(e.dt, 1, q, e.s));
      WgradP w;
      memset(&w, 0, sizeof(w));
      w.M = N; w.N = N; w.K = hd; w.ldy = (int)LkP; w.out_t = 1; w.out_accum = 0; w.nbatch = B_ * heads; w.nb_inner = heads;
      w.sY_o = (long)heads * N * LkP; w.sY_i = (long)N * LkP;
      w.sW_o = (long)N * 3 * C; w.sW_i = hd; w.ldw = 3 * C;
      WORK(e, 2.0 * (double)B_ * heads * N * N * hd, ((double)B_ * heads * N * LkP + 2.0 * B_ * N * C) * es);
      w.dY = Pd; w.A = o->g; w.lda = C; w.sA_o = (long)N * C; w.sA_i = hd; w.dW = (char*)dq + (size_t)2 * C * es;  // dV = Pd^T dO
      LCH(e, launch_wgrad(e.dt, w, e.s));
      WORK(e, 2.0 * (double)B_ * heads * N * N * hd, ((double)B_ * heads * N * LkP + 2.0 * B_ * N * C) * es);
      w.dY = dS; w.A = qkv->p; w.lda = 3 * C; w.sA_o = (long)N * 3 * C; w.sA_i = hd; w.dW = (char*)dq + (size_t)C * es;    // dK = dS^T Q
      LCH(e, launch_wgrad(e.dt, w, e.s));
      }
      // d table: dS (gradient of the raw q k^T product) summed over the windows = inv_temp * d bias; optimizer-only -> side stream
      float* dbias = e.zalloc((size_t)heads * N * LkP);
      {
        const int dt = e.dt, ws_ = sb->ws, ld = (int)LkP, ncol = (int)(heads * N * LkP);
        float* tg = sb->rpb.g; const float sc = sqrtf((float)hd);
        if (e.prof || e.dry) {
          WORK(e, 0, (double)B_ * ncol * es);
          LCH(e, launch_colsum(dt, dS, B_, ncol, ncol, dbias, e.s));
          LCH(e, launch_relpos_bias_bwd(dbias, tg, ws_, heads, ld, sc, e.s));
        } else {
          e.defer([=](hipStream_t s2) { launch_colsum(dt, dS, B_, ncol, ncol, dbias, s2); launch_relpos_bias_bwd(dbias, tg, ws_, heads, ld, sc, s2); });
        }
      }
    });
  return o;
}

// One Swin block.  `pend` = the MLP branch of the block in front whose residual add has not been materialised yet: it is folded into this
// block's first LayerNorm (op_add_ln), as this block's attention branch is folded into its second one; `keep_pending` leaves this block's own
// MLP branch pending for the next block of the stage (the last block of a stage adds it with op_droppath_add: patch merging needs the sum).
// SATRN_SWIN_NO_ADD_LN=1 (read per call, tests) keeps the separate residual adds.
struct SwinPend { Tensor* o = nullptr; float p = 0.f; };
Tensor* swin_block(Exec& e, Tensor* x, SwinBlock* sb, int B, SwinPend* pend, bool keep_pending) {
  Model* m = e.m;
  const int R = sb->res, nWw = R / sb->ws, nW = nWw * nWw;
  // shift + window partition ride on the LayerNorm in front of the attention (it writes window order) and on the residual add behind
  // it (it reads window order): no permutation passes (SATRN_SWIN_PERM_PASS=1 keeps the four separate ones, for tests)
  RowMap wmap;
  const bool perm_pass = sw_off("swin_rowmap");   // read per call
  const bool fuse_add = !sw_off("swin_add_ln");
  if ((nW > 1 || sb->shift) && !perm_pass) { wmap.H = R; wmap.W = R; wmap.ws = sb->ws; wmap.shift = sb->shift; }
  Tensor* y;
  if (pend->o) {
    AddLn r = op_add_ln(e, x, pend->o, B, pend->p, RowMap(), &sb->n1, wmap);
    x = r.sum; y = r.y;
    x->B = B; x->H = R; x->W = R;
    pend->o = nullptr;
  } else {
    y = op_ln(e, x, nullptr, &sb->n1, wmap);
  }
  Tensor* yw = ((nW > 1 || sb->shift) && perm_pass) ? op_window_perm(e, y, B, R, R, sb->ws, sb->shift, 0) : y;
  Tensor* qkv = op_gemm(e, yw, &sb->qkv, &sb->bqkv, ACT_NONE, 0.f, nullptr);
  const float* mask = sb->geo >= 0 ? (const float*)(m->ws + m->sw_geo[sb->geo].off) : nullptr;
  Tensor* att = op_window_attn(e, qkv, sb, B * nW, mask, nW);
  Tensor* pr = op_gemm(e, att, &sb->proj, &sb->bproj, ACT_NONE, 0.f, nullptr);
  Tensor* prt = ((nW > 1 || sb->shift) && perm_pass) ? op_window_perm(e, pr, B, R, R, sb->ws, sb->shift, 1) : pr;
  Tensor* x1; Tensor* y2;
  if (fuse_add) {
    AddLn r = op_add_ln(e, x, prt, B, sb->drop_path, wmap, &sb->n2, RowMap());
    x1 = r.sum; y2 = r.y;
    x1->B = B; x1->H = R; x1->W = R;
  } else {
    x1 = op_droppath_add(e, x, prt, B, sb->drop_path, wmap);
    x1->B = B; x1->H = R; x1->W = R;
    y2 = op_ln(e, x1, nullptr, &sb->n2);
  }
  Tensor* g;
  if (!sw_off("swin_gelu_epilogue")) {   // read per call (tests compare both forms in one process)
    g = op_gemm(e, y2, &sb->fc1, &sb->b1, ACT_GELU, 0.f, nullptr);   // GELU in the product's epilogue, its derivative kept beside it
  } else {
    Tensor* h = op_gemm(e, y2, &sb->fc1, &sb->b1, ACT_NONE, 0.f, nullptr);
    g = op_act(e, h, ACT_GELU);
  }
  Tensor* o = op_gemm(e, g, &sb->fc2, &sb->b2, ACT_NONE, 0.f, nullptr);
  if (fuse_add && keep_pending) { pend->o = o; pend->p = sb->drop_path; return x1; }
  Tensor* x2 = op_droppath_add(e, x1, o, B, sb->drop_path);
  x2->B = B; x2->H = R; x2->W = R;
  return x2;
}

// SwinTransformer.forward_features (networks/SWIN.py:722-735): patch embedding (4x4 stride-4 conv = GEMM over extracted patches) +
// LayerNorm + absolute position embedding, four stages of (shifted-)window blocks with patch merging between them, final LayerNorm
Tensor* swin_encoder_forward(Exec& e, const float* img, int B) {
  Model* m = e.m;
  const SatrnConfig& c = m->cfg;
  const int P = c.swin_patch, R0 = c.height / P, E = c.swin_embed;
  Tensor* pt = e.newt((long)B * R0 * R0, c.rgb * P * P, B, R0, R0);
  WORK(e, 0, (double)B * c.rgb * c.height * c.width * 4 + (double)pt->rows * pt->C * e.esz());
  LCH(e, launch_patchify(e.dt, img, pt->p, B, c.rgb, c.height, c.width, P, e.s));
  Tensor* x = op_gemm(e, pt, &m->sw_patch, &m->sw_patch_b, ACT_NONE, 0.f, nullptr);
  x->B = B; x->H = R0; x->W = R0;
  x = op_ln(e, x, nullptr, &m->sw_patch_norm);
  x = op_add_table(e, x, &m->sw_ape, B);
  (void)E;
  for (auto& st : m->swin) {
    SwinPend pend;
    for (size_t bi = 0; bi < st.blocks.size(); ++bi) x = swin_block(e, x, &st.blocks[bi], B, &pend, bi + 1 < st.blocks.size());
    if (st.down) {
      Tensor* mg = op_patch_merge(e, x, B, st.res, st.res);
      Tensor* mn = op_ln(e, mg, nullptr, &st.dnorm);
      x = op_gemm(e, mn, &st.dred, nullptr, ACT_NONE, 0.f, nullptr);
      x->B = B; x->H = st.res / 2; x->W = st.res / 2;
    }
  }
  x = op_ln(e, x, nullptr, &m->sw_norm);
  m->seg_mark[0] = 0; m->seg_mark[1] = e.tape.size();
  if (x->H != m->feat_h || x->W != m->feat_w) { m->err = "SwinTRN: final resolution does not match the configuration"; e.oom = true; }
  return x;  // [B * 144][8E]
}

// ---- composite blocks ---------------------------------------------------------------------------------
static void same_geo(int H, int W, int Ci, int stride, Geo* g) {
  g->H = H; g->W = W; g->Ci = Ci; g->KW = 3; g->stride = stride;
  if (stride == 1) { g->OH = H; g->OW = W; g->pt = 1; g->pl = 1; }
  else {
    g->OH = (H + stride - 1) / stride; g->OW = (W + stride - 1) / stride;
    int ph = std::max((g->OH - 1) * stride + 3 - H, 0), pw = std::max((g->OW - 1) * stride + 3 - W, 0);
    g->pt = ph / 2; g->pl = pw / 2;
  }
}

Tensor* mha_self(Exec& e, Tensor* x, MHAp* a, int B, int L, int causal, const int64_t* text, int ld_text, float out_drop) {
  const int D = a->D;
  Tensor* qkv = op_gemm(e, x, &a->qkv, &a->bqkv, ACT_NONE, 0.f, nullptr);
  Tensor* att = op_attn(e, qkv, 0, qkv, D, 2 * D, B, L, L, a->heads, D, causal, text, ld_text, e.drop);
  return op_gemm(e, att, &a->out, &a->bout, ACT_NONE, out_drop, nullptr);
}

Tensor* encoder_layer(Exec& e, Tensor* x, EncLayer* el) {
  const int B = x->B, H = x->H, W = x->W;
  const float p = e.drop;
  // MultiHeadAttention.dropout followed by EncoderLayer.dropout0: two independent masks == one mask with 1-(1-p)^2
  const float out_drop = 1.f - (1.f - p) * (1.f - p);
  Tensor* y2;
  // the one-launch region re-streams the weights per (image, head pair): it wins up to ~96 images (B = 64: 43.9 vs 50.6 us for the four
  // launches) and loses beyond (128: 78.9 vs 58.9; 512: 301 vs 129; 1024: 600 vs 254 -- tools/enc_attn_region.py --batch)
  static const long ea_max_b = sw_knob("enc_attn_max_b", 96);
  if (enc_attn_fused_ok(e.dt, H * W, el->att.D, el->att.heads) && !g_det.on && B <= ea_max_b) {
    // ONE launch for LayerNorm -> q|k|v -> attention -> output-projection partials (kernels_encattn.hip).  The four ops are run with
    // nolaunch: they allocate their outputs and record their (unfused) backward closures exactly as before; the fused kernel then fills
    // those outputs, and the LayerNorm behind the block adds the partial projections (+ bias, dropout) in a fixed order.
    MHAp* a = &el->att;
    const int L = H * W, D = a->D;
    e.nolaunch = true;
    Tensor* y1 = op_ln(e, x, nullptr, &el->norm);
    float* mr1 = e.last_mr;
    Tensor* qkv = op_gemm(e, y1, &a->qkv, &a->bqkv, ACT_NONE, 0.f, nullptr);
    Tensor* att = op_attn(e, qkv, 0, qkv, D, 2 * D, B, L, L, a->heads, D, 0, nullptr, 0, e.drop);
    float* lse = e.last_lse; const uint32_t site_att = e.last_site; const float drop_att = e.last_drop;
    Tensor* o = op_gemm(e, att, &a->out, &a->bout, ACT_NONE, out_drop, nullptr);
    const uint32_t site_out = e.last_site; const float drop_out = e.last_drop;
    e.nolaunch = false;
    const int HP = a->heads / 2;
    void* parts = e.alloc((size_t)HP * B * L * D * e.esz());
    EncAttnP q;
    memset(&q, 0, sizeof(q));
    q.x = x->p; q.ln_w = el->norm.w.p; q.ln_b = el->norm.b.p; q.wqkv = a->qkv.fwd; q.bqkv = a->bqkv.p; q.wo = a->out.fwd;
    q.y1 = y1->p; q.mr = mr1; q.qkv = qkv->p; q.att = att->p; q.lse = lse; q.parts = parts;
    q.B = B; q.L = L; q.D = D; q.H = a->heads; q.LkP = (int)attn_lkp(L);
    q.inv_temp = 1.0f / sqrtf((float)D); q.drop_p = drop_att; q.seed = (const uint32_t*)(scal(e.m) + SC_SEED); q.site = site_att;
    WORK(e, 2.0 * (double)B * L * D * 4 * D + 4.0 * (double)B * L * L * D, ((double)B * L * D * 11 + 4.0 * D * D) * e.esz());
    LCH(e, launch_enc_attn_fwd(q, e.s));
    // y2 = norm(x + attention) with the attention output formed from the partial projections
    used(o); used(x);
    y2 = e.newt(o->rows, D, B, H, W);
    float* mr2 = (float*)e.alloc((size_t)2 * o->rows * 4);
    WORK(e, 0, (double)o->rows * D * e.esz() * (3 + HP));
    LCH(e, launch_layernorm_parts(parts, HP, (long)B * L * D, a->bout.p, drop_out, (const uint32_t*)(scal(e.m) + SC_SEED), site_out, o->p, x->p, el->norm.w.p,
                                  el->norm.b.p, y2->p, mr2, o->rows, D, e.s));
    if (e.rec) {
      LNp* ln = &el->norm;
      const long R = o->rows;
      e.tape.push_back([&e, o, x, y2, ln, mr2, R, D]() {
        if (!y2->g) return;
        int ba = 0, bb = 0;
        void* da = e.grad(o, &ba);
        void* db = e.grad(x, &bb);
        WORK(e, 0, (double)R * D * e.esz() * (5 + ba + bb));
        ln_backward(e, y2->g, o->p, x->p, ln, mr2, da, db, ba, bb, R, D);
      });
    }
  } else {
    Tensor* y1 = op_ln(e, x, nullptr, &el->norm);
    Tensor* o = mha_self(e, y1, &el->att, B, H * W, 0, nullptr, 0, out_drop);
    y2 = op_ln(e, o, x, &el->norm);
  }
  y2->B = B; y2->H = H; y2->W = W;
  Tensor* z = op_quirk(e, y2);
  Tensor* c0 = op_gemm(e, z, &el->conv0, nullptr, ACT_NONE, 0.f, nullptr, 0, false, nullptr, true);
  c0->B = B; c0->H = H; c0->W = W;
  BnHold hold;   // BatchNorm + ReLU + depthwise 3x3 (+bias) in one launch where the shape allows (launch_bn_dwconv)
  Tensor* b0 = op_bn_act(e, c0, &el->norm0, ACT_RELU, nullptr, nullptr, &hold);
  Tensor* d = op_dwconv(e, b0, &el->dw, &el->dwb, 1, H, W, 1, 1, true, &hold);
  Tensor* b1 = op_bn_act(e, d, &el->dwnorm, ACT_RELU, nullptr);
  Tensor* c1 = op_gemm(e, b1, &el->conv1, nullptr, ACT_NONE, 0.f, nullptr, 0, false, nullptr, true);
  c1->B = B; c1->H = H; c1->W = W;
  return op_bn_act(e, c1, &el->norm1, ACT_RELU, x);
}

// squeeze-and-excite: pool + MLP in one kernel, x*gate in a second; backward = dgate reduction, two SE kernels, dx
Tensor* op_se(Exec& e, Tensor* x, EffBlock* eb, float* poolsum = nullptr, SeHold* sh = nullptr) {
  used(x);
  const int B = x->B, HW = x->H * x->W, C = x->C, S = eb->se;
  float* pooled = (float*)e.alloc((size_t)B * C * 4);
  float* u1 = (float*)e.alloc((size_t)B * S * 4);
  float* s1 = (float*)e.alloc((size_t)B * S * 4);
  Tensor* gate = e.newt(B, C);
  Tensor* y = e.newt(x->rows, C, B, x->H, x->W);
  e.last_se = Exec::LastSe{pooled, u1, s1, gate->p, true};
  y->se_out = true;
  bool fused = false;
  if (sh && sh->armed && sh->dwfn) {
    // inference: depthwise 3x3 + eval BatchNorm + SiLU + pool + MLP + x*gate in ONE launch where the grid is resident at once
    SeEvalArgs a{g_sebox.box, g_sebox.images, eb->se_r.fwd, eb->se_rb.p, eb->se_e.fwd, eb->se_eb.p, S};
    const int r = sh->dwfn(sh->esc, sh->esh, sh->act, sh->z, sh->pool, y->p, g_sebox.box ? &a : nullptr);
    fused = r == 2;
    if (r == 0) poolsum = nullptr;   // the generic depthwise kernel left no pool: the squeeze-and-excite kernel pools by itself
    sh->armed = false; sh->dwfn = nullptr;
  }
  if (sh && sh->armed) {
    // BatchNorm + activation + pool + MLP + x*gate in ONE launch (the image's workgroups hand the pool and the hidden layer to each other).
    // The activated tensor x is stored only if something will read it: the backward recomputes it from the BatchNorm's input whenever
    // the wide squeeze-and-excite backward with the folded BatchNorm sums applies (the closure below decides the same way).
    BNp* bn = sh->bn;
    const bool bwd_recomputes = g_fuse_bnb && e.dt == DT_BF16 && !g_det.on && S <= 64 && (S % 8) == 0 && (C % 8) == 0 && ((C / 8 + 7) / 8) <= 24 &&
                                !sw_off("se_wide_bwd") && !sw_off("se_bn_sums");
    const bool need_x = e.rec && !(e.train && bwd_recomputes);
    e.last_se.need_x = need_x;
    // (not under hipGraph capture: the per-launch mailbox tag would be replayed)
    unsigned long long* box = g_sebox.box;
    WORK(e, 4.0 * (double)B * C * S, (double)x->rows * C * e.esz() * (need_x ? 3 : 2) + (double)C * S * e.esz() * 2);
    LCH(e, fused = launch_bn_pool_se(e.dt, sh->y, sh->sums, sh->rep, bn->w.p, bn->b.p, bn->rm, bn->rv, bn->nbt, bn->eps, 0.1f, sh->ss, sh->mr,
                                     need_x ? sh->z : nullptr, eb->se_r.fwd, eb->se_rb.p, eb->se_e.fwd, eb->se_eb.p, pooled, u1, s1, gate->p, y->p, box,
                                     g_sebox.images, B, HW, C, S, sh->act, e.s));
    if (!fused) {
      WORK(e, 0, (double)sh->M * sh->C * e.esz() * 2);
      LCH(e, launch_bn_act_pool(e.dt, sh->y, sh->sums, sh->rep, bn->w.p, bn->b.p, bn->rm, bn->rv, bn->nbt, bn->eps, 0.1f, sh->ss, sh->mr, sh->z, sh->pool,
                                sh->M, sh->C, sh->HW, sh->act, e.s));
    }
    sh->armed = false;
  }
  if (!fused && poolsum) {  // the pool was accumulated by the BatchNorm pass in front: MLP + x*gate in one launch
    WORK(e, 4.0 * (double)B * C * S, (double)x->rows * C * e.esz() * 2 + (double)C * S * e.esz() * 2);
    LCH(e, fused = launch_se_mlp_scale(e.dt, x->p, poolsum, eb->se_r.fwd, eb->se_rb.p, eb->se_e.fwd, eb->se_eb.p, pooled, u1, s1, gate->p, y->p, B, HW, C, S, e.s));
    if (e.dry) fused = e.dt == DT_BF16 && S <= 64 && (S % 8) == 0 && C <= 1536 && (C % 8) == 0;
  }
  if (!fused) {
    WORK(e, 4.0 * (double)B * C * S, (double)x->rows * C * e.esz() + (double)C * S * e.esz() * 2);
    LCH(e, launch_se_fwd(e.dt, x->p, eb->se_r.fwd, eb->se_rb.p, eb->se_e.fwd, eb->se_eb.p, pooled, u1, s1, gate->p, B, HW, C, S, e.s));
    WORK(e, 0, (double)x->rows * C * e.esz() * 2);
    LCH(e, launch_se_scale(e.dt, x->p, gate->p, y->p, B, HW, C, e.s));
  }
  if (e.rec)
    e.tape.push_back([&e, x, y, gate, eb, pooled, u1, s1, B, HW, C, S]() {
      std::shared_ptr<GemmP> held = y->dgrad_hold;   // the data gradient of the product that consumed y, not launched yet (op_gemm)
      y->dgrad_hold.reset();
      BnBwdHold bb = y->bhold;                       // the backward-apply pass of the BatchNorm behind that product, not launched yet either
      y->bhold.armed = false;
      std::function<void()> after = std::move(y->after_fused);   // ... and that product's weight gradient, which reads the pass's output
      y->after_fused = nullptr;
      auto run_held = [&]() {
        if (bb.armed) {
          bb.armed = false;
          WORK(e, 0, (double)bb.M * bb.C * e.esz() * 3);
          LCH(e, launch_bn_bwd_apply(e.dt, bb.dz, bb.y, bb.ss, bb.mr, bb.w, bb.red, bb.M, bb.C, bb.act, bb.dy, bb.dwp, bb.dbp, e.s, bb.rep, nullptr, nullptr, 0, 0));
        }
        if (after) { after(); after = nullptr; }
        if (!held) return;
        WORK(e, y->dgrad_hold_flops, y->dgrad_hold_bytes);
        LCH(e, launch_gemm(e.dt, AM_DENSE, *held, e.s));
        held.reset();
      };
      if (!y->g) { run_held(); return; }
      void* dgate = e.alloc((size_t)B * C * e.esz());
      void* dpooled = e.alloc((size_t)B * C * e.esz());
      float* dz2 = (float*)e.alloc((size_t)B * C * 4);
      float* du1 = (float*)e.alloc((size_t)B * S * 4);
      const int dt = e.dt; void* gp = gate->p; const void* w1 = eb->se_r.fwd; const void* w2 = eb->se_e.fwd;
      float* g1 = eb->se_r.g; float* gb1 = eb->se_rb.g; float* g2 = eb->se_e.g; float* gb2 = eb->se_eb.g;
      float* ds1 = e.zalloc((size_t)B * S);
      bool wide = false;
      constexpr bool fold = true;
      // x is a BatchNorm output read by this op only: its backward computes y->g*gate + dpooled/HW on the fly, so the gradient
      // tensor se_bwd_x would write is never materialised -- and (round 2) that BatchNorm's backward column sums come out of the
      // two SE kernels as well
      const bool folds = fold && x->bn_y && !x->bn_has_res && x->ncons == 1 && !x->g;
      const bool bnred = folds && g_fuse_bnb && !x->bn_red && !sw_off("se_bn_sums");   // read per call (tests)
      float* bnP = bnred ? (float*)e.alloc((size_t)4 * B * C * 4) : nullptr;
      float* bnR = bnred ? e.zalloc((size_t)2 * C) : nullptr;
      if (held && bnred && x->bn_act == ACT_SILU && !e.dry) {
        // the projection's data gradient + this backward in ONE launch (only the image's own workgroups wait for each other)
        bool one = false;
        WORK(e, y->dgrad_hold_flops + 8.0 * (double)B * C * S, y->dgrad_hold_bytes + (double)x->rows * C * e.esz() * 2 + (double)C * S * e.esz() * 2);
        MbDinArgs da{bb.dz, bb.y, bb.ss, bb.mr, bb.w, bb.red, bb.rep, bb.dy, bb.dwp, bb.dbp};
        LCH(e, one = launch_mbconv_bwd_se(dt, bb.armed ? nullptr : held->A, bb.armed ? &da : nullptr, held->Bw, held->K, held->C, x->bn_y, x->bn_ss, x->bn_mr, gp, u1,
                                          w2, w1, dz2, ds1, du1, dpooled, bnR, B, x->H, x->W, held->lda, C, S, e.s));
        if (one) { held.reset(); wide = true; bb.armed = false; if (after) { after(); after = nullptr; } }
      }
      run_held();
      if (!wide) {
      WORK(e, 8.0 * (double)B * C * S, (double)x->rows * C * e.esz() * 2 + (double)C * S * e.esz() * 2);
      LCH(e, wide = launch_se_bwd_wide(dt, y->g, x->p, gp, u1, w1, w2, dz2, du1, ds1, dpooled, B, HW, C, S, e.s, bnred ? x->bn_y : nullptr, x->bn_ss, x->bn_mr,
                                       x->bn_act, bnP, bnR));
      }
      if (wide && bnred && !e.dry) { x->bn_red = bnR; x->bn_red_rep = 1; }
      if (e.dry) wide = true;   // planning pass: same allocations either way
      if (wide) {
        // weight gradients of the two SE matrices: optimizer-only -> side stream
        if (e.prof) { WORK(e, 8.0 * (double)B * C * S, (double)C * S * 8 + (double)B * C * 8); LCH(e, launch_se_bwd(dt, dgate, gp, u1, s1, pooled, w1, w2, dz2, du1, dpooled, g1, gb1, g2, gb2, B, C, S, e.s, 2)); }
        else if (!e.dry) e.defer([=](hipStream_t ws) { launch_se_bwd(dt, dgate, gp, u1, s1, pooled, w1, w2, dz2, du1, dpooled, g1, gb1, g2, gb2, B, C, S, ws, 2); });
      } else {
        WORK(e, 0, (double)x->rows * C * e.esz() * 2);
        LCH(e, launch_se_bwd_gate(e.dt, y->g, x->p, dgate, B, HW, C, e.s));
        // data path on the main chain; the two weight-gradient products only feed the optimizer -> side stream
        if (e.prof || e.dry) {
          WORK(e, 8.0 * (double)B * C * S, (double)C * S * e.esz() * 2 + (double)B * C * 16);
          LCH(e, launch_se_bwd(dt, dgate, gp, u1, s1, pooled, w1, w2, dz2, du1, dpooled, g1, gb1, g2, gb2, B, C, S, e.s, 3));
        } else {
          launch_se_bwd(dt, dgate, gp, u1, s1, pooled, w1, w2, dz2, du1, dpooled, g1, gb1, g2, gb2, B, C, S, e.s, 1);
          e.defer([=](hipStream_t ws) { launch_se_bwd(dt, dgate, gp, u1, s1, pooled, w1, w2, dz2, du1, dpooled, g1, gb1, g2, gb2, B, C, S, ws, 2); });
        }
      }
      if (folds) {
        x->g = y->g; x->g_init = true;
        x->se_gate = gate->p; x->se_dpool = dpooled; x->se_hw = HW;
      } else {
        int beta;
        void* dx = e.grad(x, &beta);
        WORK(e, 0, (double)x->rows * C * e.esz() * (beta ? 3 : 2));
        LCH(e, launch_se_bwd_x(e.dt, y->g, gate->p, dpooled, dx, B, HW, C, beta, e.s));
      }
    });
  return y;
}

// a held block-ending BatchNorm (Exec::xhold) that its consumer does not take: the launch it stood for
static void flush_xhold(Exec& e) {
  Exec::FwdBnHold& h = e.xhold;
  if (!h.armed) return;
  h.armed = false;
  WORK(e, 0, (double)h.M * h.C * e.esz() * (h.res ? 3 : 2));
  LCH(e, launch_bn_act(e.dt, h.y, h.sums, h.rep, h.bn->w.p, h.bn->b.p, h.bn->rm, h.bn->rv, h.bn->nbt, h.bn->eps, 0.1f, h.ss, h.mr, h.res, h.z, h.M, h.C, ACT_NONE, e.s));
}

Tensor* eff_block(Exec& e, Tensor* x, EffBlock* eb, bool hold_out = false) {
  const int B = x->B, H = x->H, W = x->W;
  Geo g;
  same_geo(H, W, eb->cin, eb->stride, &g);
  Tensor* skip = eb->skip ? x : nullptr;
  if (eb->type != 2) flush_xhold(e);
  if (eb->type == 0) {
    Tensor* y = op_gemm(e, x, &eb->c0, nullptr, ACT_NONE, 0.f, &g, B, false, nullptr, true);
    return op_bn_act(e, y, &eb->bn1, ACT_SILU, skip);
  }
  if (eb->type == 1) {
    Tensor* y = op_gemm(e, x, &eb->c0, nullptr, ACT_NONE, 0.f, &g, B, false, nullptr, true);
    Tensor* z = op_bn_act(e, y, &eb->bn1, ACT_SILU, nullptr);
    Tensor* y2 = op_gemm(e, z, &eb->c1, nullptr, ACT_NONE, 0.f, nullptr, 0, false, nullptr, true);
    y2->B = B; y2->H = g.OH; y2->W = g.OW;
    return op_bn_act(e, y2, &eb->bn2, ACT_NONE, skip);
  }
  // The front of the block -- expand product, BatchNorm + SiLU, depthwise 3x3, BatchNorm + SiLU, squeeze-and-excite -- as ONE launch on the
  // small maps of the late stages (kernels_mbconv.hip: the batch statistics travel between the workgroups of the launch instead of
  // across two kernel boundaries).  The ops below then only do their bookkeeping (tensors, tape): same allocations, same backward.
  const bool front = eb->stride == 1 && e.train && !e.nolaunch && x->C == eb->cin &&
                     (e.dry ? (e.dt == DT_BF16 && !g_det.on) : mbconv_front_ok(e.dt, B, H, W, eb->cin, eb->c0.N, eb->se, e.s));
  // a BatchNorm (+ residual) held back by the block in front (Exec::xhold): this launch computes x from it; no one-launch front -> launch it now
  const bool xfold = front && e.xhold.armed && e.xhold.z == x->p && e.xhold.C == eb->cin;
  Exec::FwdBnHold xh = e.xhold;
  if (xfold) e.xhold.armed = false; else flush_xhold(e);
  if (front) e.nolaunch = true;
  Tensor* y = op_gemm(e, x, &eb->c0, nullptr, ACT_NONE, 0.f, nullptr, 0, false, nullptr, true);
  y->B = B; y->H = H; y->W = W;
  BnHold hold;
  Tensor* z = op_bn_act(e, y, &eb->bn1, ACT_SILU, nullptr, nullptr, &hold);
  float* ss1 = e.last_bn_ss; float* mr1 = e.last_bn_mr;
  Tensor* y2 = op_dwconv(e, z, &eb->dw, nullptr, eb->stride, g.OH, g.OW, g.pt, g.pl, true, &hold);
  float* poolsum = nullptr;
  SeHold sehold;
  Tensor* z2 = op_bn_act(e, y2, &eb->bn2, ACT_SILU, nullptr, &poolsum, nullptr, &sehold);
  float* ss2 = e.last_bn_ss; float* mr2 = e.last_bn_mr;
  Tensor* z3 = op_se(e, z2, eb, poolsum, &sehold);
  if (front) {
    e.nolaunch = false;
    const int C = eb->c0.N;
    const Exec::LastSe se = e.last_se;
    bool ok = false;
    WORK(e, 2.0 * (double)x->rows * C * eb->cin + 18.0 * (double)x->rows * C + 4.0 * (double)B * C * eb->se,
         ((double)x->rows * eb->cin + (double)x->rows * C * (se.need_x ? 5 : 4) + (double)C * eb->cin + 2.0 * (double)C * eb->se) * e.esz());
    MbXinArgs xa{xh.y, xh.res, xh.sums, xh.rep, xfold ? xh.bn->w.p : nullptr, xfold ? xh.bn->b.p : nullptr, xfold ? xh.bn->rm : nullptr, xfold ? xh.bn->rv : nullptr,
                 xfold ? xh.bn->nbt : nullptr, xh.ss, xh.mr, xfold ? xh.bn->eps : 0.f, xh.z};
    LCH(e, ok = launch_mbconv_front(e.dt, xfold ? nullptr : x->p, xfold ? &xa : nullptr, eb->c0.fwd, y->p, eb->bn1.w.p, eb->bn1.b.p, eb->bn1.rm, eb->bn1.rv, eb->bn1.nbt, ss1, mr1, eb->bn1.eps, z->p,
                                    eb->dw.fwd, y2->p, eb->bn2.w.p, eb->bn2.b.p, eb->bn2.rm, eb->bn2.rv, eb->bn2.nbt, ss2, mr2, eb->bn2.eps,
                                    se.need_x ? z2->p : nullptr, eb->se_r.fwd, eb->se_rb.p, eb->se_e.fwd, eb->se_eb.p, se.pooled, se.u1, se.s1, se.gate, z3->p,
                                    B, H, W, eb->cin, C, eb->se, 0.1f, e.s));
    if (!ok && !e.dry) { e.m->err = "internal: the MBConv block launch refused a shape its own check accepted"; e.oom = true; }
  }
  Tensor* y3 = op_gemm(e, z3, &eb->c1, nullptr, ACT_NONE, 0.f, nullptr, 0, false, nullptr, true);
  y3->B = B; y3->H = g.OH; y3->W = g.OW;
  y3->bn_bwd_hold_ok = e.train && e.rec && e.dt == DT_BF16 && !g_det.on;   // (the backward closures decide; every refusal launches the held pass itself)
  return op_bn_act(e, y3, &eb->bn3, ACT_NONE, skip, nullptr, nullptr, nullptr, hold_out ? &e.xhold : nullptr);
}

// eval-mode scale/shift of every BatchNorm (running statistics may have changed since the last call: one launch)
static void bn_eval_prepare(Exec& e) {
  Model* m = e.m;
  if (e.dry || !g_fuse_bn_eval) return;
  BnEvalDesc* dev = (BnEvalDesc*)(m->ws + m->off_bn_desc);
  if (m->bn_desc_dirty) {
    m->bn_desc_host.clear();
    float* tab = (float*)(m->ws + m->off_bn_eval);
    for (BNp* b : m->all_bn) m->bn_desc_host.push_back(BnEvalDesc{b->w.p, b->b.p, b->rm, b->rv, tab + b->eval_off, b->eps, b->C});
    // the host vector is a member: it outlives the asynchronous copy
    (void)hipMemcpyAsync(dev, m->bn_desc_host.data(), m->bn_desc_host.size() * sizeof(BnEvalDesc), hipMemcpyHostToDevice, e.s);
    m->bn_desc_dirty = false;
  }
  launch_bn_eval_prepare(dev, (int)m->all_bn.size(), e.s);
}

Tensor* swin_encoder_forward(Exec& e, const float* img, int B);
// diagnostics: remember a stage-boundary tensor (satrn_model_probe_enable); its gradient is cast out when the backward passes this point
static void probe(Exec& e, const std::string& name, Tensor* t) {
  if (!e.probe_on || e.dry || t->f32) return;
  e.probes.push_back({name, t, nullptr});
  if (e.rec) {
    const size_t idx = e.probes.size() - 1;
    e.tape.push_back([&e, idx]() {
      Exec::Probe& pr = e.probes[idx];
      if (pr.gout && pr.t->g) launch_cast(e.dt, DT_F32, pr.t->g, pr.gout, pr.t->rows * pr.t->C, e.s);
    });
  }
}

Tensor* encoder_forward(Exec& e, const float* img, int B) {
  if (e.m->cfg.network == 2) return swin_encoder_forward(e, img, B);
  if (!e.train && !e.rec) bn_eval_prepare(e);
  Model* m = e.m;
  const SatrnConfig& c = m->cfg;
  Tensor* x;
  if (c.network == 0) {
    x = op_stem(e, img, &m->lite_conv[0], B, c.rgb, c.height, c.width, 1, 1);
    x = op_bn_act(e, x, &m->lite_bn[0], ACT_RELU, nullptr);
    x = op_maxpool(e, x);
    for (int i = 1; i < 4; ++i) {
      Geo g;
      same_geo(x->H, x->W, x->C, 1, &g);
      x = op_gemm(e, x, &m->lite_conv[i], nullptr, ACT_NONE, 0.f, &g, B, false, nullptr, true);
      x = op_bn_act(e, x, &m->lite_bn[i], ACT_RELU, nullptr);
      x = op_maxpool(e, x);
    }
  } else {
    auto stage_mark = [&e](const std::string& nm) {  // forward mark now, backward mark when the tape reaches this point
      if (!g_stage_prof || e.dry) return;
      e.mark(("f:" + nm).c_str());
      if (e.rec) { Exec* ep = &e; e.tape.push_back([ep, nm]() { ep->mark(("b:" + nm).c_str()); }); }
    };
    stage_mark("begin");
    x = op_stem(e, img, &m->stem, B, c.rgb, c.height, c.width, 2, 0);
    x = op_bn_act(e, x, &m->stem_bn, ACT_SILU, nullptr);
    stage_mark("stem");
    probe(e, "stem", x);
    for (size_t bi = 0; bi < m->blocks.size(); ++bi) {
      if ((int)bi == m->late_block) m->seg_mark[0] = e.tape.size();
      // this block's closing BatchNorm (+ residual) rides in the next block's launch when that one is a one-launch MBConv front
      bool hold_out = false;
      if (bi + 1 < m->blocks.size() && e.train && !e.dry && !sw_off("mbconv_xfold")) {
        const EffBlock& cbk = m->blocks[bi]; const EffBlock& nb = m->blocks[bi + 1];
        Geo og;
        same_geo(x->H, x->W, cbk.cin, cbk.stride, &og);
        hold_out = cbk.type == 2 && nb.type == 2 && nb.stride == 1 && nb.cin == cbk.cout && mbconv_front_ok(e.dt, B, og.OH, og.OW, nb.cin, nb.c0.N, nb.se, e.s);
      }
      x = eff_block(e, x, &m->blocks[bi], hold_out);
      if (bi + 1 == m->blocks.size() || m->blocks[bi + 1].cout != m->blocks[bi].cout) {
        stage_mark("cout" + std::to_string(m->blocks[bi].cout));
        probe(e, "backbone_c" + std::to_string(m->blocks[bi].cout), x);
      }
    }
    flush_xhold(e);
    int H = x->H, W = x->W;
    x = op_gemm(e, x, &m->conv_last, nullptr, ACT_NONE, 0.f, nullptr, 0, false, nullptr, true);
    x->B = B; x->H = H; x->W = W;
    x = op_bn_act(e, x, &m->bn_last, ACT_SILU, nullptr);
  }
  if (x->H != m->feat_h || x->W != m->feat_w) { m->err = "feature map size does not match input_size/32 (or /16)"; e.oom = true; }
  probe(e, "backbone_out", x);
  m->seg_mark[1] = e.tape.size();  // end of the backbone
  if (c.network == 0) m->seg_mark[0] = 0;
  // adaptive 2D positional encoding (networks/EfficientSATRN.py:135-154)
  Tensor* pooled = op_pool(e, x);
  Tensor* h0 = op_gemm(e, pooled, &m->pe_d0, &m->pe_b0, ACT_RELU, e.drop, nullptr);
  Tensor* gate = op_gemm(e, h0, &m->pe_d1, &m->pe_b1, ACT_SIGMOID, 0.f, nullptr);
  x = op_posenc_apply(e, x, gate);
  if (g_stage_prof && !e.dry) { e.mark("f:posenc"); if (e.rec) { Exec* ep = &e; e.tape.push_back([ep]() { ep->mark("b:enc_layers"); }); } }
  probe(e, "posenc", x);
  { int li = 0; for (auto& el : m->enc) { x = encoder_layer(e, x, &el); probe(e, "enc_layer" + std::to_string(li++), x); } }
  if (g_stage_prof && !e.dry) { e.mark("f:enc_layers"); if (e.rec) { Exec* ep = &e; e.tape.push_back([ep]() { ep->mark("b:decoder"); }); } }
  // inference postpones products until their BatchNorm is known (op_gemm / op_dwconv -> op_bn_act): none may be left over
  for (auto& t : e.tens)
    if (t->pend || t->pend_dw) { m->err = "internal: a postponed product was never launched"; e.oom = true; }
  return x;  // [B*HW][D] == [b, hw, c]
}

// out rows (b, j<n) = src rows (b*sbs + soff + j); backward scatters (accumulating) into src's gradient
Tensor* op_rows(Exec& e, Tensor* src, int B, int n, long sbs, long soff) {
  used(src);
  const int C = src->C;
  Tensor* y = e.newt((long)B * n, C, B);
  LCH(e, launch_copy_rows(e.dt, src->p, y->p, B, n, C, sbs, soff, n, 0, 0, e.s));
  if (e.rec)
    e.tape.push_back([&e, src, y, B, n, C, sbs, soff]() {
      if (!y->g) return;
      int beta;
      void* g = e.grad(src, &beta);
      if (!beta) LCH(e, launch_fill(g, 0, (size_t)src->rows * C * e.esz(), e.s));
      LCH(e, launch_copy_rows(e.dt, y->g, g, B, n, C, n, 0, sbs, soff, 1, e.s));
    });
  return y;
}
// dst rows (b*dbs + doff + j) = x rows (b*n + j), in place inside an existing tensor; backward routes dst's gradient
// rows back to x (x is the only writer of those rows)
void op_store_rows(Exec& e, Tensor* dst, Tensor* x, int B, int n, long dbs, long doff) {
  used(dst); used(x);
  const int C = x->C;
  LCH(e, launch_copy_rows(e.dt, x->p, dst->p, B, n, C, n, 0, dbs, doff, 0, e.s));
  if (e.rec)
    e.tape.push_back([&e, dst, x, B, n, C, dbs, doff]() {
      if (!dst->g) return;
      int beta;
      void* gx = e.grad(x, &beta);
      LCH(e, launch_copy_rows(e.dt, dst->g, gx, B, n, C, dbs, doff, n, 0, beta, e.s));
    });
}
// hist = [F rows (b, 0..t-1) ; x row b]  ->  [B*(t+1)][C]
Tensor* op_hist(Exec& e, Tensor* F, Tensor* x, int B, int T, int t) {
  used(F); used(x);
  const int C = x->C;
  Tensor* h = e.newt((long)B * (t + 1), C, B);
  if (t > 0) LCH(e, launch_copy_rows(e.dt, F->p, h->p, B, t, C, T, 0, t + 1, 0, 0, e.s));
  LCH(e, launch_copy_rows(e.dt, x->p, h->p, B, 1, C, 1, 0, t + 1, t, 0, e.s));
  if (e.rec)
    e.tape.push_back([&e, F, x, h, B, T, t, C]() {
      if (!h->g) return;
      if (t > 0) {
        int bf;
        void* gF = e.grad(F, &bf);
        if (!bf) LCH(e, launch_fill(gF, 0, (size_t)F->rows * C * e.esz(), e.s));
        LCH(e, launch_copy_rows(e.dt, h->g, gF, B, t, C, t + 1, 0, T, 0, 1, e.s));
      }
      int bx;
      void* gx = e.grad(x, &bx);
      LCH(e, launch_copy_rows(e.dt, h->g, gx, B, 1, C, t + 1, t, 1, 0, bx, e.s));
    });
  return h;
}

// Train-time autoregressive branch WITH gradients (networks/EfficientSATRN.py:496-525): every step feeds the argmax of
// the previous step, the self-attention history of a layer is k/v_linear over [its previous outputs ; current input]
// (recomputed per step like the reference, so gradients reach every earlier output), dropout stays active.
Tensor* decoder_ar(Exec& e, Tensor* src, int B, int L, float* logits_out) {
  Model* m = e.m;
  const SatrnConfig& c = m->cfg;
  const int T = L - 1, Dd = c.dec_hidden, V = c.num_classes, Nsrc = (int)(src->rows / B);
  const int nl = (int)m->dec.size();
  int64_t* ids = (int64_t*)e.alloc((size_t)B * T * 8);
  int64_t* sos = (int64_t*)e.alloc((size_t)B * 8);
  LCH(e, launch_fill_i64(sos, c.sos_id, B, e.s));
  std::vector<Tensor*> F(nl), crossKV(nl);
  for (int l = 0; l < nl; ++l) {
    F[l] = e.newt((long)B * T, Dd, B);
    crossKV[l] = op_gemm(e, src, &m->dec[l].cross_att.kv, &m->dec[l].cross_att.bkv, ACT_NONE, 0.f, nullptr);
  }
  // full logits tensor [B*T][V] (fp32, caller's buffer); its gradient arrives as [B*T][Vp] in the compute dtype
  e.tens.emplace_back(new Tensor());
  Tensor* full = e.tens.back().get();
  // (the fused training step passes no buffer: the logits then live in the arena, like decoder_tf's)
  if (!logits_out) logits_out = (float*)e.alloc((size_t)B * T * V * sizeof(float));
  full->rows = (long)B * T; full->C = V; full->f32 = true; full->p = logits_out;
  const int Vp = m->gen.ldb;
  const float fp = (e.train && e.drop > 0.f) ? 0.1f : 0.f;
  for (int t = 0; t < T; ++t) {
    Tensor* x = op_embed(e, t == 0 ? sos : ids + (t - 1), t == 0 ? 1 : T, B, 1, t, 0.f);
    for (int l = 0; l < nl; ++l) {
      DecLayer& dl = m->dec[l];
      Tensor* hist = op_hist(e, F[l], x, B, T, t);
      Tensor* q = op_gemm(e, x, &dl.self_att.qonly, &dl.self_att.bq, ACT_NONE, 0.f, nullptr);
      Tensor* kv = op_gemm(e, hist, &dl.self_att.kv, &dl.self_att.bkv, ACT_NONE, 0.f, nullptr);
      Tensor* att = op_attn(e, q, 0, kv, 0, Dd, B, 1, t + 1, c.dec_heads, Dd, 0, nullptr, 0, e.drop);
      Tensor* o = op_gemm(e, att, &dl.self_att.out, &dl.self_att.bout, ACT_NONE, e.drop, nullptr);
      Tensor* t1 = op_ln(e, o, x, &dl.ln1);
      Tensor* q2 = op_gemm(e, t1, &dl.cross_att.qkv, &dl.cross_att.bqkv, ACT_NONE, 0.f, nullptr);
      Tensor* a2 = op_attn(e, q2, 0, crossKV[l], 0, Dd, B, 1, Nsrc, c.dec_heads, Dd, 0, nullptr, 0, e.drop);
      Tensor* o2 = op_gemm(e, a2, &dl.cross_att.out, &dl.cross_att.bout, ACT_NONE, e.drop, nullptr);
      Tensor* t2 = op_ln(e, o2, t1, &dl.ln2);
      Tensor* f0 = op_gemm(e, t2, &dl.lin0, &dl.b0, ACT_RELU, fp, nullptr);
      Tensor* f1 = op_gemm(e, f0, &dl.lin1, &dl.b1, ACT_RELU, fp, nullptr);
      x = op_ln(e, f1, t2, &dl.ln3);
      op_store_rows(e, F[l], x, B, 1, T, t);
    }
    Tensor* lt = op_gemm(e, x, &m->gen, &m->gen_b, ACT_NONE, 0.f, nullptr, 0, true);  // [B][V] fp32
    LCH(e, launch_copy_rows(DT_F32, lt->p, logits_out, B, 1, V, 1, 0, T, t, 0, e.s));
    LCH(e, launch_argmax((const float*)lt->p, ids + t, B, V, V, T, e.s));
    if (e.rec)
      e.tape.push_back([&e, full, lt, B, T, t, Vp]() {
        if (!full->g) return;
        lt->g = e.alloc((size_t)B * Vp * e.esz());
        lt->g_init = true;
        LCH(e, launch_copy_rows(e.dt, full->g, lt->g, B, 1, Vp, T, t, 1, 0, 0, e.s));
      });
  }
  return full;
}

// one weight gradient over a [M][N] slab of output gradients and the [M][K] slab of the product's inputs (the side stream's work, as in
// op_gemm's closure; the bias gradient rides in the same launch)
static void wgrad_slab(Exec& e, Wt* w, Vec* bias, const void* dY, int ldy, const void* A, int lda, long M, int N) {
  WgradP q;
  memset(&q, 0, sizeof(q));
  q.dY = dY; q.A = A; q.dW = w->g; q.M = (int)M; q.N = N; q.K = w->K; q.ldy = ldy; q.lda = lda;
  q.nbatch = 1; q.nb_inner = 1;
  q.full_grid = (e.serial || !e.s2 || e.prof) ? 1 : 0;
  float* bg = bias ? bias->g : nullptr;
  if (bg && !sw_off("wgrad_bias") && !g_det.on) { q.dbias = bg; bg = nullptr; }
  const int dt = e.dt;
  if (e.prof || e.dry) {
    if (bg) { WORK(e, 0, (double)M * N * e.esz()); LCH(e, launch_colsum(dt, dY, M, N, ldy, bg, e.s)); }
    WORK(e, 2.0 * (double)M * N * w->K, ((double)M * N + (double)M * w->K) * e.esz() + (double)N * w->K * 4);
    LCH(e, launch_wgrad(dt, q, e.s));
  } else {
    e.defer([=](hipStream_t ws) {
      if (bg) launch_colsum(dt, dY, M, N, ldy, bg, ws);
      launch_wgrad(dt, q, ws);
    });
  }
}

// The same branch as decoder_ar in two launches (kernels_ar.hip: one workgroup per image runs all T steps of a direction) plus one
// product per weight over [B*T]-row slabs: 2 + 26 launches instead of ~126 per step.
Tensor* decoder_ar_fused(Exec& e, Tensor* src, int B, int L, float* logits_out) {
  Model* m = e.m;
  const SatrnConfig& c = m->cfg;
  const int T = L - 1, D = c.dec_hidden, F = c.dec_filter, V = c.num_classes, Nsrc = (int)(src->rows / B), H = c.dec_heads;
  const int nl = (int)m->dec.size();
  const size_t es = e.esz();
  const long R = (long)B * T;
  std::vector<Tensor*> crossKV(nl);
  for (int l = 0; l < nl; ++l) crossKV[l] = op_gemm(e, src, &m->dec[l].cross_att.kv, &m->dec[l].cross_att.bkv, ACT_NONE, 0.f, nullptr);
  e.tens.emplace_back(new Tensor());
  Tensor* full = e.tens.back().get();
  if (!logits_out) logits_out = (float*)e.alloc((size_t)R * V * sizeof(float));
  full->rows = R; full->C = V; full->f32 = true; full->p = logits_out;
  auto ap = std::make_shared<ArP>();
  memset(ap.get(), 0, sizeof(ArP));
  auto slab = [&](int C) { return e.alloc((size_t)R * C * es); };
  // k-panel-major copies of every weight and of its transpose (the packed backward copy is W^T row-major): re-made per call, the
  // weights change every step
  auto kp = [&](const void* w, int N, int K) -> const void* {
    void* d = e.alloc((size_t)N * K * es);
    WORK(e, 0, (double)N * K * es * 2);
    if (d) LCH(e, launch_repack_kpanel(e.dt, w, d, N, K, e.s));
    return d;
  };
  for (int l = 0; l < nl; ++l) {
    DecLayer& dl = m->dec[l];
    ArLayer& w = ap->L[l];
    w.wqkv = kp(dl.self_att.qkv.fwd, 3 * D, D); w.wqkvT = kp(dl.self_att.qkv.bwd, D, 3 * D);
    w.wo = kp(dl.self_att.out.fwd, D, D); w.woT = kp(dl.self_att.out.bwd, D, D);
    w.wq2 = kp(dl.cross_att.qkv.fwd, D, D); w.wq2T = kp(dl.cross_att.qkv.bwd, D, D);
    w.wo2 = kp(dl.cross_att.out.fwd, D, D); w.wo2T = kp(dl.cross_att.out.bwd, D, D);
    w.w0 = kp(dl.lin0.fwd, F, D); w.w0T = kp(dl.lin0.bwd, D, F);
    w.w1 = kp(dl.lin1.fwd, D, F); w.w1T = kp(dl.lin1.bwd, F, D);
    w.bqkv = dl.self_att.bqkv.p; w.bo = dl.self_att.bout.p; w.bq2 = dl.cross_att.bqkv.p; w.bo2 = dl.cross_att.bout.p; w.b0 = dl.b0.p; w.b1 = dl.b1.p;
    w.ln1w = dl.ln1.w.p; w.ln1b = dl.ln1.b.p; w.ln2w = dl.ln2.w.p; w.ln2b = dl.ln2.b.p; w.ln3w = dl.ln3.w.p; w.ln3b = dl.ln3.b.p;
    w.dln1w = dl.ln1.w.g; w.dln1b = dl.ln1.b.g; w.dln2w = dl.ln2.w.g; w.dln2b = dl.ln2.b.g; w.dln3w = dl.ln3.w.g; w.dln3b = dl.ln3.b.g;
    w.crossKV = crossKV[l]->p;
    w.cache = slab(2 * D);
    w.q = slab(D); w.kvin = slab(2 * D); w.att = slab(D); w.s1 = slab(D); w.t1 = slab(D); w.q2 = slab(D); w.a2 = slab(D); w.s2 = slab(D);
    w.t2 = slab(D); w.f0 = slab(F); w.f1d = slab(D);
  }
  for (int l = 0; l <= nl; ++l) ap->xs[l] = slab(D);
  ap->Ltab = (ArLayer*)e.alloc(4 * sizeof(ArLayer));
  ap->nlayers = nl; ap->embed = m->embed.p; ap->pe = (const float*)(m->ws + m->off_pe1d); ap->wgen = kp(m->gen.fwd, V, D); ap->bgen = m->gen_b.p;
  ap->logits = logits_out;
  ap->ids = (int64_t*)e.alloc((size_t)R * 8); ap->in_ids = (int64_t*)e.alloc((size_t)R * 8);
  ap->B = B; ap->T = T; ap->D = D; ap->F = F; ap->V = V; ap->H = H; ap->Nsrc = Nsrc; ap->sos = c.sos_id;
  const bool drop = e.train && e.drop > 0.f;
  ap->p_att = drop ? e.drop : 0.f; ap->p_res = drop ? e.drop : 0.f;
  ap->p_ff = drop ? 0.1f : 0.f;   // Feedforward dropout is hard-wired to 0.1 (networks/EfficientSATRN.py:327)
  ap->seed = (const uint32_t*)(scal(m) + SC_SEED); ap->site = drop ? e.site++ : 0;
  // per step and image: 15 D^2 + 2 D F multiply-accumulates per layer + the generator; the weights are streamed once per step and image
  WORK(e, 2.0 * (double)R * (nl * (7.0 * D * D + 2.0 * D * F) + (double)D * V), (double)R * (nl * (7.0 * D * D + 2.0 * D * F) + (double)D * V) * es);
  ap->G = ar_fwd_slices(e.dt, D, F, H);
  ap->fbox = ap->G > 1 ? (unsigned long long*)e.alloc(ar_fwd_box_bytes(B, ap->G, D)) : nullptr;
  {
    int rc = 0;
    LCH(e, rc = launch_ar_fwd(e.dt, *ap, e.s));
    if (rc && ap->G > 1) { ap->G = 1; LCH(e, rc = launch_ar_fwd(e.dt, *ap, e.s)); }   // the sliced form could not be resident: one workgroup per image
    if (rc) { m->err = "autoregressive forward: launch refused"; e.oom = true; }
  }
  if (e.rec)
    e.tape.push_back([&e, m, ap, full, crossKV, B, T, D, F, V, Nsrc, nl, R, es]() {
      if (!full->g) return;
      const int Vp = m->gen.ldb;
      // generator: its data gradient for all steps in one product, its weight gradient over the slab
      void* dxtop = e.alloc((size_t)R * D * es);
      {
        GemmP d;
        memset(&d, 0, sizeof(d));
        d.A = full->g; d.Bw = m->gen.bwd; d.C = dxtop; d.M = (int)R; d.N = D; d.K = Vp; d.lda = Vp; d.ldc = D;
        WORK(e, 2.0 * (double)R * D * V, ((double)R * Vp + (double)R * D + (double)V * D) * es);
        LCH(e, launch_gemm(e.dt, AM_DENSE, d, e.s));
      }
      wgrad_slab(e, &m->gen, &m->gen_b, full->g, Vp, ap->xs[nl], D, R, V);
      ap->dxtop = dxtop; ap->dx0 = e.alloc((size_t)R * D * es);
      ap->lnpart = (float*)e.alloc((size_t)B * nl * 6 * D * 4);
      ap->gbox = (unsigned long long*)e.alloc(ar_bwd_box_bytes(B, T, D, nl));
      for (int l = 0; l < nl; ++l) {
        ArLayer& w = ap->L[l];
        w.dqkvi = e.alloc((size_t)R * 3 * D * es); w.dkvo = e.alloc((size_t)R * 2 * D * es); w.dout = e.alloc((size_t)R * D * es);
        w.dq2 = e.alloc((size_t)R * D * es); w.dout2 = e.alloc((size_t)R * D * es); w.df0 = e.alloc((size_t)R * F * es); w.df1 = e.alloc((size_t)R * D * es);
        w.dkvacc = (float*)e.alloc((size_t)R * 2 * D * 4);
        w.dcross = (float*)e.alloc((size_t)B * Nsrc * 2 * D * 4);
        WORK(e, 0, (double)R * 2 * D * 4);
        if (w.dkvacc) LCH(e, launch_fill(w.dkvacc, 0, (size_t)R * 2 * D * 4, e.s));
        WORK(e, 0, (double)B * Nsrc * 2 * D * 4);
        if (w.dcross) LCH(e, launch_fill(w.dcross, 0, (size_t)B * Nsrc * 2 * D * 4, e.s));
      }
      WORK(e, 2.0 * (double)R * nl * (8.0 * D * D + 2.0 * D * F), (double)R * nl * (8.0 * D * D + 2.0 * D * F) * es);
      int rc = 0;
      LCH(e, rc = launch_ar_bwd(e.dt, *ap, e.s));
      if (rc) { m->err = "autoregressive backward: the layer pipeline cannot be resident on this device (SATRN_OFF=ar_fused runs the operator form)"; e.oom = true; }
      for (int l = 0; l < nl; ++l) {
        DecLayer& dl = m->dec[l];
        ArLayer& w = ap->L[l];
        wgrad_slab(e, &dl.self_att.qkv, &dl.self_att.bqkv, w.dqkvi, 3 * D, ap->xs[l], D, R, 3 * D);
        wgrad_slab(e, &dl.self_att.kv, &dl.self_att.bkv, w.dkvo, 2 * D, ap->xs[l + 1], D, R, 2 * D);
        wgrad_slab(e, &dl.self_att.out, &dl.self_att.bout, w.dout, D, w.att, D, R, D);
        wgrad_slab(e, &dl.cross_att.qkv, &dl.cross_att.bqkv, w.dq2, D, w.t1, D, R, D);
        wgrad_slab(e, &dl.cross_att.out, &dl.cross_att.bout, w.dout2, D, w.a2, D, R, D);
        wgrad_slab(e, &dl.lin0, &dl.b0, w.df0, F, w.t2, D, R, F);
        wgrad_slab(e, &dl.lin1, &dl.b1, w.df1, D, w.f0, F, R, D);
        // the cross-attention keys / values: their product's closure (recorded before this one) takes it from here
        Tensor* kvt = crossKV[l];
        kvt->g = e.alloc((size_t)kvt->rows * kvt->C * es);
        kvt->g_init = true;
        WORK(e, 0, (double)kvt->rows * kvt->C * (4 + es));
        if (kvt->g && w.dcross) LCH(e, launch_cast(DT_F32, e.dt, w.dcross, kvt->g, kvt->rows * kvt->C, e.s));
      }
      WORK(e, 0, (double)R * D * (4 + es));
      LCH(e, launch_embed_bwd(e.dt, ap->in_ids, ap->dx0, m->embed.g, B, T, T, D, 0.f, ap->seed, 0, e.s, m->embed.N));
    });
  return full;
}

static bool ar_fused_ok(Exec& e, Tensor* src, int B, int L) {
  const SatrnConfig& c = e.m->cfg;
  return ar_train_ok(e.dt, B, c.dec_hidden, c.dec_filter, c.num_classes, c.dec_heads, L - 1, (int)(src->rows / B), (int)e.m->dec.size());
}

Tensor* decoder_tf(Exec& e, Tensor* src, const int64_t* expected, int B, int L, float* logits_out) {
  Model* m = e.m;
  const SatrnConfig& c = m->cfg;
  const int T = L - 1, Dd = c.dec_hidden, Nsrc = (int)(src->rows / B);
  Tensor* t = op_embed(e, expected, L, B, T, 0, e.drop);
  probe(e, "dec_embed", t);
  for (auto& dl : m->dec) {
    Tensor* o = mha_self(e, t, &dl.self_att, B, T, 1, expected, L, e.drop);
    Tensor* t1 = op_ln(e, o, t, &dl.ln1);
    Tensor* q = op_gemm(e, t1, &dl.cross_att.qkv, &dl.cross_att.bqkv, ACT_NONE, 0.f, nullptr);
    Tensor* kv = op_gemm(e, src, &dl.cross_att.kv, &dl.cross_att.bkv, ACT_NONE, 0.f, nullptr);
    Tensor* a2 = op_attn(e, q, 0, kv, 0, Dd, B, T, Nsrc, c.dec_heads, Dd, 0, nullptr, 0, e.drop);
    Tensor* o2 = op_gemm(e, a2, &dl.cross_att.out, &dl.cross_att.bout, ACT_NONE, e.drop, nullptr);
    Tensor* t2 = op_ln(e, o2, t1, &dl.ln2);
    const float fp = e.train ? 0.1f : 0.f;  // Feedforward dropout is hard-wired to 0.1 (networks/EfficientSATRN.py:327)
    Tensor* f0 = op_gemm(e, t2, &dl.lin0, &dl.b0, ACT_RELU, e.drop > 0.f ? fp : 0.f, nullptr);
    Tensor* f1 = op_gemm(e, f0, &dl.lin1, &dl.b1, ACT_RELU, e.drop > 0.f ? fp : 0.f, nullptr);
    t = op_ln(e, f1, t2, &dl.ln3);
    probe(e, "dec_layer" + std::to_string(&dl - &m->dec[0]), t);
  }
  return op_gemm(e, t, &m->gen, &m->gen_b, ACT_NONE, 0.f, nullptr, 0, true, logits_out);
}
}  // namespace

// =====================================================================================================
// model-level entry points
// =====================================================================================================
// the reduction mode is process-global state of the kernel launchers: every engine entry point that launches work selects
// its own model's mode first
static void det_activate(Model* m) {
  sw_refresh();
  g_det.on = (m->det_floats && m->ws) ? 1 : 0;
  g_det.cap = m->det_floats;
  g_det.scratch[0] = g_det.on ? (float*)(m->ws + m->off_det) : nullptr;
  g_det.scratch[1] = g_det.on ? (float*)(m->ws + m->off_det) + m->det_floats : nullptr;
  g_det.side = m->ex ? m->ex->s2 : nullptr;
  g_wgrad_dense_blocks = 160;   // (SATRN: 96 until round 4; 160-192 measured best once the chain got shorter: 8.55 vs 8.60 ms; 224 -> 8.59, 256 -> 8.63)
  g_wgrad_big_min_gflop = m->cfg.network == 2 ? 1.0f : 2.0f;
  const bool wp = m->wgpart_floats && m->ws;
  g_wgpart.cap = wp ? m->wgpart_floats : 0;
  g_wgpart.scratch[0] = wp ? (float*)(m->ws + m->off_wgpart) : nullptr;
  g_wgpart.scratch[1] = wp ? (float*)(m->ws + m->off_wgpart) + m->wgpart_floats : nullptr;
  g_wgpart.side = m->ex ? m->ex->s2 : nullptr;
  // (not under hipGraph capture -- Exec::serial: a captured launch would replay its mailbox tag)
  const bool mbox = wp && m->off_sebox && !(m->ex && m->ex->serial);
  g_sebox.box = mbox ? (unsigned long long*)(m->ws + m->off_sebox) : nullptr;
  g_sebox.images = mbox ? Model::SEBOX_IMAGES : 0;
  g_sebox.bwd = mbox && sw_knob("se_bwd_one_launch", 0) != 0;   // (measured slower inside the step: off by default)
  g_mbbox.box = (mbox && m->off_mbbox) ? (unsigned long long*)(m->ws + m->off_mbbox) : nullptr;
  g_mbbox.words = g_mbbox.box ? Model::MBBOX_WORDS : 0;
  g_mbbox.images = g_mbbox.box ? Model::MBBOX_IMAGES : 0;
}

static void exec_begin(Model* m, hipStream_t s, bool train, bool rec, bool dry) {
  Exec& e = *m->ex;
  det_activate(m);
  e.s = s; e.dt = m->cfg.dtype; e.train = train; e.rec = rec; e.dry = dry;
  e.drop = train ? m->cfg.dropout : 0.f;
  e.probes.clear(); e.probe_on = m->probe_on; e.xhold.armed = false;
  if (!e.dry && e.zbase && e.zoff > m->zero_hwm) m->zero_hwm = e.zoff;  // what the previous call dirtied at most
  e.reset(m->ws + m->persist_bytes, m->ws_bytes > m->persist_bytes ? m->ws_bytes - m->persist_bytes : 0,
          m->ws + m->off_zero, m->zero_bytes);
  e.peak = 0;
  ++m->epoch;  // whatever lived in the arena (e.g. a step-wise decoding session) is gone
  // the pool is all-zero after set_workspace and only ever dirtied up to the high-water mark: clear just that much
  if (!dry && m->zero_hwm) launch_fill(m->ws + m->off_zero, 0, std::min(m->zero_bytes, (m->zero_hwm + 4095) & ~(size_t)4095), s);
}

size_t model_workspace_bytes(Model* m, int B, int L) {
  // dry-run the training forward + backward (and the greedy decoder's buffers) with a counting arena
  Exec& e = *m->ex;
  char* save_ws = m->ws;
  m->ws = nullptr;
  exec_begin(m, nullptr, true, true, true);
  e.cap = (size_t)1 << 60; e.zcap = m->zero_bytes;
  Tensor* src = encoder_forward(e, nullptr, B);
  Tensor* lg = decoder_tf(e, src, nullptr, B, L, nullptr);
  lg->g = e.alloc((size_t)lg->rows * m->gen.ldb * e.esz());
  e.alloc((size_t)lg->rows * 4);
  for (auto it = e.tape.rbegin(); it != e.tape.rend(); ++it) (*it)();
  size_t train_peak = e.peak;
  e.tape.clear(); e.tens.clear();
  // the autoregressive training branch keeps O(T^2) history tensors: size it as well
  {
    exec_begin(m, nullptr, true, true, true);
    e.cap = (size_t)1 << 60; e.zcap = m->zero_bytes;
    Tensor* s2 = encoder_forward(e, nullptr, B);
    Tensor* l2 = decoder_ar(e, s2, B, L, nullptr);
    l2->g = e.alloc((size_t)l2->rows * m->gen.ldb * e.esz());
    e.alloc((size_t)l2->rows * 4);
    for (auto it = e.tape.rbegin(); it != e.tape.rend(); ++it) (*it)();
    if (e.peak > train_peak) train_peak = e.peak;
    e.tape.clear(); e.tens.clear();
  }
  {   // ... and its two-launch form (kernels_ar.hip): slabs instead of per-step tensors
    exec_begin(m, nullptr, true, true, true);
    e.cap = (size_t)1 << 60; e.zcap = m->zero_bytes;
    Tensor* s2 = encoder_forward(e, nullptr, B);
    if (ar_fused_ok(e, s2, B, L)) {
      Tensor* l2 = decoder_ar_fused(e, s2, B, L, nullptr);
      l2->g = e.alloc((size_t)l2->rows * m->gen.ldb * e.esz());
      e.alloc((size_t)l2->rows * 4);
      for (auto it = e.tape.rbegin(); it != e.tape.rend(); ++it) (*it)();
      if (e.peak > train_peak) train_peak = e.peak;
    }
    e.tape.clear(); e.tens.clear();
  }
  // module.eval() semantics WITH gradients (train_step phase + 32: BatchNorm running statistics, the unfused BatchNorm-backward forms):
  // its backward allocates differently from the training-mode tape (found at bs4 128x384 bf16: "workspace exhausted in backward")
  {
    exec_begin(m, nullptr, false, true, true);
    e.cap = (size_t)1 << 60; e.zcap = m->zero_bytes;
    Tensor* s3 = encoder_forward(e, nullptr, B);
    Tensor* l3 = decoder_tf(e, s3, nullptr, B, L, nullptr);
    l3->g = e.alloc((size_t)l3->rows * m->gen.ldb * e.esz());
    e.alloc((size_t)l3->rows * 4);
    for (auto it = e.tape.rbegin(); it != e.tape.rend(); ++it) (*it)();
    if (e.peak > train_peak) train_peak = e.peak;
    e.tape.clear(); e.tens.clear();
  }
  // greedy: encoder (eval) + caches
  size_t es = e.esz();
  size_t dec = (size_t)m->cfg.dec_layers * ((size_t)B * 512 * 2 * m->cfg.dec_hidden * es + (size_t)B * m->feat_h * m->feat_w * 2 * m->cfg.dec_hidden * es) +
               (size_t)64 * B * std::max(m->cfg.dec_filter, 3 * m->cfg.dec_hidden) * 4 + (1u << 20);
  {  // k-panel-major weight copies of the persistent decoder
    const size_t D = m->cfg.dec_hidden, F = m->cfg.dec_filter;
    dec += ((size_t)m->cfg.dec_layers * (6 * D * D + 2 * D * F) + (size_t)m->cfg.num_classes * D) * es + (64u << 10);
  }
  // pipelined decoder: role table + one 2 KB granule mailbox per (edge, image)
  dec += 256 + 256 * sizeof(PipeRole) + ((size_t)(m->cfg.dec_layers + 1) + (size_t)m->cfg.dec_layers * 18) * B * 256 * 8 + (64u << 10);
  // beam search: node tables (1 + 16*499 nodes x 32 B) and ancestor-row lists (499 x 504 x 2 B) per image
  dec += (size_t)B * ((size_t)(1 + 16 * 499) * 32 + (size_t)499 * 504 * 2) + (64u << 10);
  m->ws = save_ws;
  size_t need = m->persist_bytes + std::max(train_peak, train_peak / 2 + dec) + (16u << 20);
  return need;
}

int model_forward(Model* m, const float* img, const int64_t* expected, int B, int L, bool train, bool record,
                  float* logits_out, hipStream_t s, bool teacher_forced) {
  if (!m->bound || !m->ws_set) { m->err = "bind parameters and set a workspace first"; return -1; }
  Exec& e = *m->ex;
  // every TRAINING forward draws fresh dropout / stochastic-depth masks (its backward reads the same RNG word); the module-API
  // path (model(...) + loss.backward(), the reference's loop) advances here just like the fused train_step
  if (train) launch_seed_advance((uint32_t*)(scal(m) + SC_SEED), s);
  exec_begin(m, s, train, record, false);
  e.src = encoder_forward(e, img, B);
  m->seg_mark[2] = e.tape.size();  // end of the encoder
  m->seg_next = 0;
  e.logits = teacher_forced ? decoder_tf(e, e.src, expected, B, L, logits_out)
                            : (ar_fused_ok(e, e.src, B, L) ? decoder_ar_fused(e, e.src, B, L, logits_out) : decoder_ar(e, e.src, B, L, logits_out));
  m->logits_epoch = m->epoch;
  if (e.oom) { if (m->err.empty()) m->err = "workspace exhausted"; return -2; }
  return 0;
}

// Backward in four segments (decoder | encoder transformer + positional encoding | late backbone | early backbone): after
// segment k, the flat-gradient range seg_lo[k]..seg_hi[k] is final on stream s (side-stream weight gradients joined), so
// a data-parallel caller can start reducing it while the next segments run.
static void segment_ranges(Model* m) {
  const int64_t off_dec = m->embed.off, off_pe = m->cfg.network == 2 ? m->embed.off : m->pe_d0.off;
  int64_t off_late = 0;
  m->late_block = 0;
  if (m->cfg.network == 2 && m->swin.size() == 4 && !m->swin[2].blocks.empty()) {
    // SwinTRN: the deep third stage onwards (18 of 24 blocks, ~85 % of the encoder parameters) is the "late" part
    off_late = m->swin[2].blocks[0].n1.w.off;
  } else if (m->cfg.network != 0 && !m->blocks.empty()) {
    // last stage of the backbone (15 blocks at 4x12, ~47 % of all parameters) + conv_last
    size_t k = m->blocks.size();
    while (k > 0 && m->blocks[k - 1].cout == m->blocks.back().cout) --k;
    m->late_block = (int)k;
    off_late = m->blocks[k].c0.off;
  }
  m->seg_lo[0] = off_dec;  m->seg_hi[0] = m->n_params;
  m->seg_lo[1] = off_pe;   m->seg_hi[1] = off_dec;
  m->seg_lo[2] = off_late; m->seg_hi[2] = off_pe;
  m->seg_lo[3] = 0;        m->seg_hi[3] = off_late;
}

// runs segments seg..seg_to in one go (one side-stream join, at the end)
int model_backward_segment(Model* m, const int64_t* expected, int B, int L, int seg, hipStream_t s, int seg_to) {
  Exec& e = *m->ex;
  det_activate(m);
  if (!e.logits) { m->err = "no forward"; return -1; }
  if (seg_to < seg) seg_to = seg;
  if (seg != m->seg_next || seg < 0 || seg_to > 3) { m->err = "backward segments must run in order 0..3 after a recorded forward"; return -1; }
  e.s = s;
  if (seg == 0) {
    Tensor* lg = e.logits;
    const int Vp = m->gen.ldb, V = m->cfg.num_classes;
    float* lse = (float*)e.alloc((size_t)lg->rows * 4);
    lg->g = e.alloc((size_t)lg->rows * Vp * e.esz());
    lg->g_init = true;
    launch_ce_full(e.dt, (const float*)lg->p, expected, L, 1, B, L - 1, V, Vp, m->cfg.pad_id, scal(m) + SC_LOSS, lse, lg->g, nullptr, s);
  }
  const size_t hi = seg == 0 ? e.tape.size() : m->seg_mark[3 - seg];
  const size_t lo = seg_to == 3 ? 0 : m->seg_mark[2 - seg_to];
  for (size_t i = hi; i > lo; --i) e.tape[i - 1]();
  e.join();
  m->seg_next = seg_to + 1;
  if (seg_to == 3) e.tape.clear();
  if (e.oom) { m->err = "workspace exhausted in backward"; return -2; }
  return 0;
}

static void run_tape(Exec& e) {
  for (auto it = e.tape.rbegin(); it != e.tape.rend(); ++it) (*it)();
  e.tape.clear();
  e.join();
}

int model_backward(Model* m, const float* dlogits, hipStream_t s) {
  Exec& e = *m->ex;
  det_activate(m);
  if (!e.logits || e.tape.empty()) { m->err = "no recorded forward"; return -1; }
  e.s = s;
  Tensor* lg = e.logits;
  const int Vp = m->gen.ldb;
  lg->g = e.alloc((size_t)lg->rows * Vp * e.esz());
  lg->g_init = true;
  launch_cast_pad(e.dt, dlogits, lg->g, lg->rows, m->cfg.num_classes, Vp, s);
  run_tape(e);
  if (e.oom) { m->err = "workspace exhausted in backward"; return -2; }
  return 0;
}

int model_loss_backward(Model* m, const int64_t* expected, int B, int L, hipStream_t s) {
  Exec& e = *m->ex;
  det_activate(m);
  if (!e.logits) { m->err = "no forward"; return -1; }
  e.s = s;
  Tensor* lg = e.logits;
  const int Vp = m->gen.ldb, V = m->cfg.num_classes;
  float* lse = (float*)e.alloc((size_t)lg->rows * 4);
  void* dl = nullptr;
  if (!e.tape.empty()) { lg->g = e.alloc((size_t)lg->rows * Vp * e.esz()); lg->g_init = true; dl = lg->g; }
  else dl = e.alloc((size_t)lg->rows * Vp * e.esz());
  launch_ce_full(e.dt, (const float*)lg->p, expected, L, 1, B, L - 1, V, Vp, m->cfg.pad_id, scal(m) + SC_LOSS, lse, dl,
                 nullptr, s);
  run_tape(e);
  if (e.oom) { m->err = "workspace exhausted in backward"; return -2; }
  return 0;
}


int model_train_step(Model* m, const float* img, const int64_t* expected, int B, int L, const float* hyper9,
                     int use_graph, int phase, hipStream_t s, const float* hyper9_dec) {
  // hyper9_dec != null: the reference's dual-optimizer iteration (train_modules/train_dual_opt.py:87-113) -- encoder and
  // decoder parameters are clipped SEPARATELY (clip_grad_norm_ per group) and stepped with their own learning rates
  if (hyper9_dec && use_graph) { m->err = "the dual-optimizer step runs eagerly (use_graph must be 0)"; return -1; }
  // 16 + k (+ 4 * k_to): backward segments k..k_to (k == 0 also zeroes the gradients and runs forward + CE)
  // + 32: forward with BatchNorm RUNNING statistics and no dropout (module.eval() semantics) but gradients recorded -- the
  // per-sample-independent mode in which N ranks' averaged gradients equal one rank's on the concatenated batch
  const bool train_mode = !(phase & 32);
  // + 64: the reference's NON-teacher-forced training branch (networks/EfficientSATRN.py:496-525: the decoder feeds on its own argmax,
  // gradients flow through every step) instead of the teacher-forced one -- the branch its coin takes on 20-70 % of the training batches
  // (train_modules/train_single_opt.py:75, tf ratio 0.8 -> 0.3).
  const bool teacher_forced = !(phase & 64);
  const int seg = (phase & 16) ? (phase & 3) : -1;
  const int seg_to = (phase >> 2) & 3;
  if (seg >= 0) {
    if (!m->bound || !m->ws_set || !m->grads) { m->err = "bind parameters/grads and set a workspace first"; return -1; }
    if (seg == 0) {
      launch_fill(m->grads, 0, (size_t)m->n_params * 4, s);
      int rc = model_forward(m, img, expected, B, L, train_mode, true, nullptr, s, teacher_forced);
      if (rc) return rc;
    }
    return model_backward_segment(m, expected, B, L, seg, s, seg_to);
  }
  phase &= 3;
  if (!phase) return 0;
  if (!m->bound || !m->ws_set || !m->grads) { m->err = "bind parameters/grads and set a workspace first"; return -1; }
  if ((phase & 2) && (!m->adam_m || !m->adam_v)) { m->err = "bind the optimizer state first (satrn_model_bind_optimizer)"; return -1; }
  if (!train_mode) use_graph = 0;  // the captured graphs are the training-mode step
  // hyper-parameters for this step (lr changes every iteration in the reference's scheduler)
  if (phase & 2) m->adam_t += 1;
  // (as kernel arguments: a pinned-memory copy of 36 bytes ran as a blit on another queue, ~50 us of idle chain per step boundary)
  {
    float hy[9];
    memcpy(hy, hyper9, 9 * sizeof(float));
    hy[6] = 1.0f - powf(hy[1], (float)m->adam_t);
    hy[7] = 1.0f - powf(hy[2], (float)m->adam_t);
    launch_set_scalars(scal(m) + SC_HYPER, hy, 9, s);
    if (hyper9_dec) {
      float hd[9];
      memcpy(hd, hyper9_dec, 9 * sizeof(float));
      hd[6] = 1.0f - powf(hd[1], (float)m->adam_t);
      hd[7] = 1.0f - powf(hd[2], (float)m->adam_t);
      launch_set_scalars(scal(m) + SC_HYPER2, hd, 9, s);
    }
  }
  static const bool host_prof = sw_prof("host");  // host time spent ISSUING the forward / backward / optimizer
  auto hnow = []() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  auto body = [&]() -> int {
    const double h0 = host_prof ? hnow() : 0.0;
    double h1 = h0, h2 = h0;
    struct HostRep { const bool on; const double& a; const double& b; const double& c; decltype(hnow)& now;
                     ~HostRep() { if (on) { double d = now(); fprintf(stderr, "[host] issue: forward %.0f us, loss+backward %.0f us, optimizer %.0f us\n", b - a, c - b, d - c); } } } rep{host_prof, h0, h1, h2, hnow};
    if (phase & 1) {
      if (g_stage_prof) { m->ex->s = s; m->ex->mark_report(); m->ex->mark("start"); }
      // clearing the 109 MB gradient buffer is not on the forward's path: eager two-stream steps do it on the side stream
      // (behind everything the previous step queued on `s`, i.e. its optimizer), beside the forward; the backward waits for it
      Exec& ex = *m->ex;
      const bool side_zero = ex.s2 && !ex.serial && !use_graph;
      static hipEvent_t evz0 = nullptr, evz1 = nullptr;
      if (side_zero) {
        if (!evz0) { (void)hipEventCreateWithFlags(&evz0, hipEventDisableTiming); (void)hipEventCreateWithFlags(&evz1, hipEventDisableTiming); }
        (void)hipEventRecord(evz0, s);
        (void)hipStreamWaitEvent(ex.s2, evz0, 0);
        launch_fill(m->grads, 0, (size_t)m->n_params * 4, ex.s2);
        (void)hipEventRecord(evz1, ex.s2);
      } else {
        launch_fill(m->grads, 0, (size_t)m->n_params * 4, s);
      }
      int rc = model_forward(m, img, expected, B, L, train_mode, true, nullptr, s, teacher_forced);
      if (rc) return rc;
      if (side_zero) (void)hipStreamWaitEvent(s, evz1, 0);
      if (g_stage_prof) m->ex->mark("f:decoder");
      if (host_prof) h1 = hnow();
      rc = model_loss_backward(m, expected, B, L, s);
      if (rc) return rc;
      if (g_stage_prof) m->ex->mark("b:join");
    }
    if (host_prof) { h2 = hnow(); if (!(phase & 1)) h1 = h2; }
    if (phase & 2) {
      float* am = m->adam_m;
      float* av = m->adam_v;
      launch_fill(scal(m) + SC_GNORM, 0, 8, s);  // both norm slots
      if (!hyper9_dec) {
        launch_sumsq(m->grads, m->n_params, scal(m) + SC_GNORM, (float*)(m->ws + m->off_sumsq), s);
        launch_adamw(m->params, m->grads, am, av, m->n_params, scal(m) + SC_GNORM, scal(m) + SC_HYPER, s);
      } else {
        // flat order: encoder.* first, decoder.* from the embedding on (model.encoder.parameters() / model.decoder.parameters())
        const int64_t ne = m->embed.off, nd = m->n_params - ne;
        launch_sumsq(m->grads, ne, scal(m) + SC_GNORM, (float*)(m->ws + m->off_sumsq), s);
        launch_sumsq(m->grads + ne, nd, scal(m) + SC_GNORM2, (float*)(m->ws + m->off_sumsq), s);
        launch_adamw(m->params, m->grads, am, av, ne, scal(m) + SC_GNORM, scal(m) + SC_HYPER, s);
        launch_adamw(m->params + ne, m->grads + ne, am + ne, av + ne, nd, scal(m) + SC_GNORM2, scal(m) + SC_HYPER2, s);
      }
      int rc = model_pack_weights(m, s);
      if (g_stage_prof) m->ex->mark("opt+pack");
      return rc;
    }
    return 0;
  };
  if (!use_graph) return body();
  if (m->graph_B != B || m->graph_L != L) {
    for (int i = 0; i < 8; ++i) if (m->graphs[i]) { (void)hipGraphExecDestroy(m->graphs[i]); m->graphs[i] = nullptr; }
  if (m->decode_graph) { (void)hipGraphExecDestroy(m->decode_graph); m->decode_graph = nullptr; }
    m->graph_B = B; m->graph_L = L;
  }
  hipGraphExec_t& gx = m->graphs[phase + (teacher_forced ? 0 : 4)];   // the two decoder branches are two graphs
  if (!gx) {
    // the caller must keep img / expected at the same addresses across replays (bench + trainer use staging buffers)
    hipGraph_t g = nullptr;
    if (hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) != hipSuccess) { m->err = "stream capture failed"; return -3; }
    // captured as ONE chain: the replay does not overlap a forked branch well on ROCm 7.2, and without a concurrent chain
    // the weight-gradient kernels may use chip-filling grids
    hipStream_t keep_s2 = m->ex->s2;
    m->ex->s2 = nullptr;
    m->ex->serial = true;
    int rc = body();
    m->ex->serial = false;
    m->ex->s2 = keep_s2;
    hipError_t er = hipStreamEndCapture(s, &g);
    if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
    if (er != hipSuccess || !g) { m->err = "graph capture failed"; return -3; }
    if (hipGraphInstantiate(&gx, g, nullptr, nullptr, 0) != hipSuccess) { (void)hipGraphDestroy(g); m->err = "graph instantiate failed"; return -3; }
    (void)hipGraphDestroy(g);
  }
  // ROCm 7.2: replays that contained hipMemsetAsync nodes went wrong when launched back to back (accumulators not
  // cleared).  Every clear is now an ordinary kernel; draining the stream before a replay is kept as a cheap
  // belt-and-braces measure (one host sync per ~20 ms step).
  (void)hipStreamSynchronize(s);
  if (hipGraphLaunch(gx, s) != hipSuccess) { m->err = "graph launch failed"; return -3; }
  return 0;
}

int model_read_loss(Model* m, float* out4, hipStream_t s) {
  (void)hipMemcpyAsync(out4, scal(m) + SC_LOSS, 16, hipMemcpyDeviceToHost, s);
  (void)hipMemcpyAsync(out4 + 3, scal(m) + SC_GNORM, 4, hipMemcpyDeviceToHost, s);
  (void)hipStreamSynchronize(s);
  if (unsigned ef = device_error_read_clear(s)) {
    if (ef & 4u) { m->err = "a hand-off inside the fused BatchNorm + squeeze-and-excite kernel timed out (results of that step are invalid; SATRN_NO_FUSED_POOL_SE=1 selects the separate kernels)"; return -7; }
    m->err = std::string("token ids out of range reached the model (") + ((ef & 1) ? "decoder input outside the embedding table; " : "") +
             ((ef & 2) ? "loss target outside the vocabulary; " : "") + "rewrite the loader's -1 padding to <PAD> first, train_modules/train_single_opt.py:78)";
    return -6;
  }
  return 0;
}

// argmax over the vocabulary of the LAST forward's logits -> ids [B][L-1]: the `sequence` the reference's training loop
// derives from the model output for its per-step metrics (train_modules/train_single_opt.py:82-84)
int model_last_sequence(Model* m, int64_t* ids_out, int B, int L, hipStream_t s) {
  Exec& e = *m->ex;
  if (!e.logits || m->logits_epoch != m->epoch) { m->err = "no live forward (the last call on this model was not a forward / train step)"; return -1; }
  if ((long)B * (L - 1) != e.logits->rows) { m->err = "last_sequence: B, L differ from the last forward"; return -1; }
  launch_argmax((const float*)e.logits->p, ids_out, (int)e.logits->rows, m->cfg.num_classes, m->cfg.num_classes, 1, s);
  return 0;
}

int model_read_grad_norms(Model* m, float* out2, hipStream_t s) {
  (void)hipMemcpyAsync(out2, scal(m) + SC_GNORM, 8, hipMemcpyDeviceToHost, s);
  (void)hipStreamSynchronize(s);
  return 0;
}

int model_encode(Model* m, const float* img, int B, float* src_out, hipStream_t s) {
  if (!m->bound || !m->ws_set) { m->err = "bind parameters and set a workspace first"; return -1; }
  Exec& e = *m->ex;
  exec_begin(m, s, false, false, false);
  e.src = encoder_forward(e, img, B);
  if (e.oom) { if (m->err.empty()) m->err = "workspace exhausted"; return -2; }
  if (src_out) launch_cast(e.dt, DT_F32, e.src->p, src_out, e.src->rows * e.src->C, s);
  return 0;
}

// One KV-cached decoder step for all B sequences: ids[b*ld_ids] are the input tokens of step t, logits -> [B][ld_logits].
static int decode_one_step(Model* m, const int64_t* ids, int ld_ids, int t, int steps, std::vector<Tensor*>& crossKV,
                           std::vector<Tensor*>& cache, float* logits, int ld_logits) {
  Exec& e = *m->ex;
  hipStream_t s = e.s;
  const SatrnConfig& c = m->cfg;
  const int Dd = c.dec_hidden, V = c.num_classes, L = (int)m->dec.size();
  const int B = cache[0]->B;
  const int Nsrc = (int)(crossKV[0]->rows / B);
  const size_t es = e.esz();
  const float inv_temp = 1.0f / sqrtf((float)Dd);
  Tensor* x = op_embed(e, ids, ld_ids, B, 1, t, 0.f);
  for (int l = 0; l < L; ++l) {
    DecLayer& dl = m->dec[l];
    MHAp& sa = dl.self_att;
    // q = x Wq ; [k v](x) -> cache slot t
    Wt wq = sa.qkv; wq.N = Dd;
    Vec bq = sa.bqkv; bq.n = Dd;
    Tensor* q = op_gemm(e, x, &wq, &bq, ACT_NONE, 0.f, nullptr);
    GemmP g;
    memset(&g, 0, sizeof(g));
    g.A = x->p; g.Bw = (char*)sa.qkv.fwd + (size_t)Dd * Dd * es; g.bias = sa.bqkv.p + Dd;
    g.C = (char*)cache[l]->p + (size_t)t * 2 * Dd * es; g.M = B; g.N = 2 * Dd; g.K = Dd; g.lda = Dd; g.ldc = steps * 2 * Dd;
    launch_gemm(e.dt, AM_DENSE, g, s);
    Tensor* att = e.newt(B, Dd, B);
    AttnP p;
    memset(&p, 0, sizeof(p));
    p.Q = q->p; p.K = cache[l]->p; p.V = (char*)cache[l]->p + (size_t)Dd * es; p.O = att->p;
    p.B = B; p.H = c.dec_heads; p.Lq = 1; p.Lk = t + 1; p.hd = Dd / c.dec_heads;
    p.ldq = Dd; p.ldk = 2 * Dd; p.ldv = 2 * Dd; p.ldo = Dd;
    p.sq_b = Dd; p.sk_b = (long)steps * 2 * Dd; p.sv_b = p.sk_b; p.so_b = Dd;
    p.inv_temp = inv_temp; p.pad_id = c.pad_id;
    if (launch_attn_checked(e.dt, 0, p, s)) { m->err = "attention shape unsupported (Lk > 512?)"; return -4; }
    Tensor* o = op_gemm(e, att, &sa.out, &sa.bout, ACT_NONE, 0.f, nullptr);
    Tensor* t1 = op_ln(e, o, x, &dl.ln1);
    Tensor* q2 = op_gemm(e, t1, &dl.cross_att.qkv, &dl.cross_att.bqkv, ACT_NONE, 0.f, nullptr);
    Tensor* a2 = op_attn(e, q2, 0, crossKV[l], 0, Dd, B, 1, Nsrc, c.dec_heads, Dd, 0, nullptr, 0, 0.f);
    Tensor* o2 = op_gemm(e, a2, &dl.cross_att.out, &dl.cross_att.bout, ACT_NONE, 0.f, nullptr);
    Tensor* t2 = op_ln(e, o2, t1, &dl.ln2);
    Tensor* f0 = op_gemm(e, t2, &dl.lin0, &dl.b0, ACT_RELU, 0.f, nullptr);
    Tensor* f1 = op_gemm(e, f0, &dl.lin1, &dl.b1, ACT_RELU, 0.f, nullptr);
    x = op_ln(e, f1, t2, &dl.ln3);
    // history entry for the following steps: k/v of this layer's OUTPUT
    g.A = x->p;
    launch_gemm(e.dt, AM_DENSE, g, s);
  }
  GemmP g;
  memset(&g, 0, sizeof(g));
  g.A = x->p; g.Bw = m->gen.fwd; g.bias = m->gen_b.p; g.C = logits; g.M = B; g.N = V; g.K = Dd;
  g.lda = Dd; g.ldc = ld_logits; g.out_f32 = 1;
  launch_gemm(e.dt, AM_DENSE, g, s);
  return 0;
}

// Parameter block of the persistent decode kernels (greedy / beam search): k-panel-major weight copies in the arena,
// per-layer cross K/V and self-attention caches ([B][steps][2D]).
static void fill_decode_params(Model* m, DecodeP& dp, const std::vector<Tensor*>& crossKV, const std::vector<Tensor*>& cache, int B,
                               int steps, int Nsrc, hipStream_t s) {
  Exec& e = *m->ex;
  const SatrnConfig& c = m->cfg;
  const int Dd = c.dec_hidden, V = c.num_classes, L = (int)m->dec.size();
  const size_t es = e.esz();
  memset(&dp, 0, sizeof(dp));
  // the kernel streams every weight once per step and image: k-panel-major copies (one coalesced load per MFMA operand)
  auto kp = [&](const void* fwd, int N, int K) -> const void* {
    void* d = e.alloc((size_t)N * K * es);
    if (!e.dry && d) launch_repack_kpanel(e.dt, fwd, d, N, K, s);
    return d;
  };
  for (int l = 0; l < L; ++l) {
    DecLayer& dl = m->dec[l];
    DecLayerW& w = dp.L[l];
    w.wqkv = kp(dl.self_att.qkv.fwd, 3 * Dd, Dd); w.bqkv = dl.self_att.bqkv.p; w.wo = kp(dl.self_att.out.fwd, Dd, Dd); w.bo = dl.self_att.bout.p;
    w.wq2 = kp(dl.cross_att.qkv.fwd, Dd, Dd); w.bq2 = dl.cross_att.bqkv.p; w.wo2 = kp(dl.cross_att.out.fwd, Dd, Dd); w.bo2 = dl.cross_att.bout.p;
    w.w0 = kp(dl.lin0.fwd, c.dec_filter, Dd); w.b0 = dl.b0.p; w.w1 = kp(dl.lin1.fwd, Dd, c.dec_filter); w.b1 = dl.b1.p;
    w.bkv = dl.self_att.bqkv.p + Dd;
    w.ln1w = dl.ln1.w.p; w.ln1b = dl.ln1.b.p; w.ln2w = dl.ln2.w.p; w.ln2b = dl.ln2.b.p; w.ln3w = dl.ln3.w.p; w.ln3b = dl.ln3.b.p;
    w.crossKV = crossKV[l]->p; w.cache = cache[l]->p;
  }
  dp.nlayers = L; dp.embed = m->embed.p; dp.pe = (const float*)(m->ws + m->off_pe1d); dp.wgen = kp(m->gen.fwd, V, Dd); dp.bgen = m->gen_b.p;
  dp.B = B; dp.steps = steps; dp.D = Dd; dp.F = c.dec_filter; dp.V = V;
  dp.H = c.dec_heads; dp.Nsrc = Nsrc; dp.sos = c.sos_id;
}

// Greedy decode with the reference's step semantics (networks/EfficientSATRN.py:528-561, :386-396): the
// self-attention history of a layer is k/v_linear of that layer's previous OUTPUTS plus the current INPUT.
// KV-cached: slot t first holds k/v(input_t), is attended, then is overwritten with k/v(output_t).
static int greedy_body(Model* m, const float* img, const float* src_in, int B, int steps, float* logits_out, int64_t* ids_out,
                       hipStream_t s, const int32_t* rules, const int64_t* forced) {
  Exec& e = *m->ex;
  const SatrnConfig& c = m->cfg;
  const int Dd = c.dec_hidden, V = c.num_classes;
  exec_begin(m, s, false, false, false);
  Tensor* src;
  if (src_in) {
    const int N = m->feat_h * m->feat_w;
    src = e.newt((long)B * N, c.dec_src, B);
    launch_cast(DT_F32, e.dt, src_in, src->p, src->rows * src->C, s);
  } else {
    src = encoder_forward(e, img, B);
  }
  const int Nsrc = (int)(src->rows / B);
  const size_t es = e.esz();
  const int L = (int)m->dec.size();
  std::vector<Tensor*> crossKV(L), cache(L);
  for (int l = 0; l < L; ++l) {
    crossKV[l] = op_gemm(e, src, &m->dec[l].cross_att.kv, &m->dec[l].cross_att.bkv, ACT_NONE, 0.f, nullptr);
    cache[l] = e.newt((long)B * steps, 2 * Dd, B);
  }
  int64_t* sos = (int64_t*)e.alloc((size_t)B * 8);
  launch_fill_i64(sos, c.sos_id, B, s);
  // ---- fast path: the persistent one-launch decoders
  if (!sw_off("decode_kernel") && L <= 4) {
    DecodeP dp;
    fill_decode_params(m, dp, crossKV, cache, B, steps, Nsrc, s);
    dp.logits = logits_out; dp.ids = ids_out; dp.rules = rules; dp.forced = forced; dp.ld_forced = steps;
    // (1) bf16: the pipelined weight-stationary decoder (one workgroup per role, weights resident in LDS).  It is checked
    //     synchronously -- a decode is tens of milliseconds and its caller reads the result next -- and a pipeline that gave up
    //     (bounded waits) is re-run on the one-workgroup-per-image kernel, so a result is always the decoder's.  Not inside a
    //     stream capture (the check synchronises).
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cap);
    if (cap == hipStreamCaptureStatusNone) {
      const size_t sb = decode_pipe_scratch_bytes(dp);
      void* scratch = e.alloc(sb);
      if (!e.oom && launch_decode_pipe(e.dt, dp, scratch, sb, s) == 0) {
        const int perr = decode_pipe_error(scratch, s);
        if (perr == 0) { m->last_decode_path = 1; m->decode_note.clear(); return 0; }
        // A give-up (a bounded wait ran out: some role workgroup was not resident, or the device is wedged) is never silent: it is
        // counted, reported through satrn_model_last_decode_path, and the pipeline is not tried again in this process.  With
        // SATRN_PIPE_STRICT set it is an error instead of a re-run.
        char why[96];
        snprintf(why, sizeof(why), "role %d timed out", perr - 1);
        m->pipe_giveups += 1;
        m->decode_note = std::string("pipelined decoder gave up (") + why + ")";
        decode_pipe_disable(why);
        if (getenv("SATRN_PIPE_STRICT")) { m->err = m->decode_note; return -7; }
      } else {
        m->decode_note = decode_pipe_reason();
      }
      if (e.oom) { m->err = "workspace exhausted"; return -2; }
    } else {
      m->decode_note = "inside a stream capture";
    }
    if (launch_decode_greedy(e.dt, dp, s) == 0) {
      if (e.oom) { m->err = "workspace exhausted"; return -2; }
      m->last_decode_path = 2;
      return 0;
    }
  }
  m->last_decode_path = 3;
  int32_t* sift_state = nullptr;
  if (rules) {
    sift_state = (int32_t*)e.alloc((size_t)B * 16);
    launch_sift_reset(sift_state, B, c.sos_id, s);
  }
  const size_t mark = e.off;
  const size_t keep = e.tens.size();
  for (int t = 0; t < steps; ++t) {
    e.off = mark;  // per-step scratch is reused
    int rc = decode_one_step(m, t == 0 ? sos : (forced ? forced : ids_out) + (t - 1), t == 0 ? 1 : steps, t, steps, crossKV, cache,
                             logits_out + (size_t)t * V, steps * V);
    if (rc) return rc;
    if (rules) launch_sift_strided(logits_out + (size_t)t * V, steps * V, sift_state, rules, B, V, ids_out + t, steps, s);
    else launch_argmax(logits_out + (size_t)t * V, ids_out + t, B, V, steps * V, steps, s);
    e.tens.resize(keep);
  }
  if (e.oom) { m->err = "workspace exhausted"; return -2; }
  return 0;
}

// Best-first beam search (EfficientSATRN.beam_search, networks/EfficientSATRN.py:708-867, topk = 1): encoder, cross K/V,
// then ONE launch that runs every image's whole search (kernels_decode.hip).  sequences: int64 [B][max_sequence] (device).
int model_beam_search(Model* m, const float* img, int B, int beam_width, int max_sequence, int eos_id, int pad_id,
                      int64_t* sequences, hipStream_t s) {
  if (!m->bound || !m->ws_set) { m->err = "bind parameters and set a workspace first"; return -1; }
  if (beam_width < 1 || beam_width > 16) { m->err = "beam_width must be in 1..16"; return -1; }
  if (max_sequence < 1 || max_sequence > 500) { m->err = "max_sequence must be in 1..500 (PositionEncoder1D max_len)"; return -1; }
  Exec& e = *m->ex;
  const SatrnConfig& c = m->cfg;
  const int L = (int)m->dec.size();
  if (L > 4) { m->err = "beam search: at most 4 decoder layers"; return -1; }
  exec_begin(m, s, false, false, false);
  Tensor* src = encoder_forward(e, img, B);
  const int Nsrc = (int)(src->rows / B);
  const int E = max_sequence - 1;  // expansions per image at most (:754)
  std::vector<Tensor*> crossKV(L), cache(L);
  for (int l = 0; l < L; ++l) {
    crossKV[l] = op_gemm(e, src, &m->dec[l].cross_att.kv, &m->dec[l].cross_att.bkv, ACT_NONE, 0.f, nullptr);
    cache[l] = e.newt((long)B * std::max(E, 1), 2 * c.dec_hidden, B);
  }
  DecodeP dp;
  fill_decode_params(m, dp, crossKV, cache, B, E, Nsrc, s);
  BeamP q;
  memset(&q, 0, sizeof(q));
  q.bw = beam_width; q.max_seq = max_sequence; q.eos = eos_id; q.pad = pad_id;
  q.NN = 1 + beam_width * E;
  q.pstride = (E + 7) & ~7;
  const size_t nn = (size_t)B * q.NN;
  q.logp = (double*)e.alloc(nn * 8); q.score = (double*)e.alloc(nn * 8);
  q.parent = (int32_t*)e.alloc(nn * 4); q.tok = (int32_t*)e.alloc(nn * 4); q.len = (int32_t*)e.alloc(nn * 4); q.slot = (int32_t*)e.alloc(nn * 4);
  q.path = (uint16_t*)e.alloc((size_t)B * std::max(E, 1) * std::max(q.pstride, 8) * 2);
  q.out = sequences;
  if (e.oom) { m->err = "workspace exhausted"; return -2; }
  if (launch_beam_search(e.dt, dp, q, s) != 0) { m->err = "beam search: unsupported decoder shape for the persistent kernel"; return -1; }
  return 0;
}

// ---- step-wise decoding session (the ensemble driver's interface: networks/EfficientSATRN.py:932-952 step_forward /
// reset_status, called from utils/ensemble_utils.py:84-96).  begin = reset_status + the per-sequence work (cross K/V
// of src, empty self-attention caches); each step consumes the caller's token ids and returns logits [B][V].  The
// session lives in the workspace arena, so any other call on the same model ends it.
int model_step_begin(Model* m, const float* src_in, int B, int max_steps, hipStream_t s) {
  if (!m->bound || !m->ws_set) { m->err = "bind parameters and set a workspace first"; return -1; }
  if (max_steps < 1 || max_steps > 500) { m->err = "max 500 decode steps (PositionEncoder1D max_len)"; return -1; }
  Exec& e = *m->ex;
  const SatrnConfig& c = m->cfg;
  exec_begin(m, s, false, false, false);
  const int N = m->feat_h * m->feat_w;
  Tensor* src = e.newt((long)B * N, c.dec_src, B);
  launch_cast(DT_F32, e.dt, src_in, src->p, src->rows * src->C, s);
  const int L = (int)m->dec.size();
  m->step_cross.assign(L, nullptr); m->step_cache.assign(L, nullptr);
  for (int l = 0; l < L; ++l) {
    m->step_cross[l] = op_gemm(e, src, &m->dec[l].cross_att.kv, &m->dec[l].cross_att.bkv, ACT_NONE, 0.f, nullptr);
    m->step_cache[l] = e.newt((long)B * max_steps, 2 * c.dec_hidden, B);
  }
  if (e.oom) { m->err = "workspace exhausted"; m->step_B = 0; return -2; }
  m->step_B = B; m->step_max = max_steps; m->step_t = 0; m->step_mark = e.off; m->step_keep = e.tens.size();
  m->step_epoch = ++m->epoch;
  return 0;
}

int model_step(Model* m, const int64_t* target, float* logits_out, hipStream_t s) {
  Exec& e = *m->ex;
  det_activate(m);
  if (!m->step_B || m->step_epoch != m->epoch) { m->err = "no live step session (call satrn_model_step_begin; other calls on the model end a session)"; return -1; }
  if (m->step_t >= m->step_max) { m->err = "step session exhausted (max_steps reached)"; return -1; }
  e.s = s;
  e.off = m->step_mark;
  e.tens.resize(m->step_keep);
  int rc = decode_one_step(m, target, 1, m->step_t, m->step_max, m->step_cross, m->step_cache, logits_out, m->cfg.num_classes);
  if (rc) return rc;
  if (e.oom) { m->err = "workspace exhausted"; return -2; }
  m->step_t += 1;
  return 0;
}

// ---- per-family event profile of ONE eager training step (forward + CE + backward; no optimizer) -----------------
int model_profile_step(Model* m, const float* img, const int64_t* expected, int B, int L, char* out, size_t out_cap,
                       hipStream_t s) {
  if (!m->bound || !m->ws_set || !m->grads) { m->err = "bind parameters/grads and set a workspace first"; return -1; }
  Exec& e = *m->ex;
  std::vector<ProfRec> recs;
  e.prof = &recs;
  hipStream_t keep_s2 = e.s2;
  e.s2 = nullptr;  // per-launch event timing needs a single stream
  (void)hipMemsetAsync(m->grads, 0, (size_t)m->n_params * 4, s);
  int rc = model_forward(m, img, expected, B, L, true, true, nullptr, s);
  if (!rc) rc = model_loss_backward(m, expected, B, L, s);
  e.prof = nullptr;
  e.s2 = keep_s2;
  (void)hipStreamSynchronize(s);
  struct Agg { double ms = 0, flops = 0, bytes = 0; long n = 0; };
  std::vector<std::pair<std::string, Agg>> agg;
  for (auto& r : recs) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, r.a, r.b);
    (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
    size_t i = 0;
    for (; i < agg.size(); ++i) if (agg[i].first == r.name) break;
    if (i == agg.size()) agg.push_back({r.name, Agg()});
    agg[i].second.ms += ms; agg[i].second.flops += r.flops; agg[i].second.bytes += r.bytes; agg[i].second.n += 1;
  }
  std::sort(agg.begin(), agg.end(), [](const std::pair<std::string, Agg>& a, const std::pair<std::string, Agg>& b) { return a.second.ms > b.second.ms; });
  std::string js = "[";
  for (size_t i = 0; i < agg.size(); ++i) {
    char buf[256];
    snprintf(buf, sizeof(buf), "%s{\"kernel\": \"%s\", \"launches\": %ld, \"ms\": %.4f, \"flops\": %.6e, \"bytes\": %.6e}", i ? ", " : "",
             agg[i].first.c_str(), agg[i].second.n, agg[i].second.ms, agg[i].second.flops, agg[i].second.bytes);
    js += buf;
  }
  js += "]";
  if (out && out_cap) { strncpy(out, js.c_str(), out_cap - 1); out[out_cap - 1] = 0; }
  return rc;
}

// Greedy decode entry: eager, or (use_graph) the whole decode -- encoder + every step's ~45 launches -- captured once
// per (B, steps, buffer addresses) and replayed, which removes the host launch cost of ~10^4 kernels per batch.
int model_greedy(Model* m, const float* img, const float* src_in, int B, int steps, float* logits_out, int64_t* ids_out,
                 int use_graph, hipStream_t s, const int32_t* rules, const int64_t* forced) {
  if (!m->bound || !m->ws_set) { m->err = "bind parameters and set a workspace first"; return -1; }
  if (steps > 500) { m->err = "max 500 decode steps (PositionEncoder1D max_len)"; return -1; }
  if (!use_graph || rules || forced) return greedy_body(m, img, src_in, B, steps, logits_out, ids_out, s, rules, forced);
  const void* key[6] = {img, src_in, logits_out, ids_out, (void*)(intptr_t)B, (void*)(intptr_t)steps};
  if (m->decode_graph && memcmp(key, m->decode_key, sizeof(key)) != 0) { (void)hipGraphExecDestroy(m->decode_graph); m->decode_graph = nullptr; }
  if (!m->decode_graph) {
    hipGraph_t g = nullptr;
    if (hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) != hipSuccess) { m->err = "stream capture failed"; return -3; }
    int rc = greedy_body(m, img, src_in, B, steps, logits_out, ids_out, s, nullptr, nullptr);
    hipError_t er = hipStreamEndCapture(s, &g);
    if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
    if (er != hipSuccess || !g) { m->err = "decode graph capture failed"; return -3; }
    if (hipGraphInstantiate(&m->decode_graph, g, nullptr, nullptr, 0) != hipSuccess) { (void)hipGraphDestroy(g); m->err = "decode graph instantiate failed"; return -3; }
    (void)hipGraphDestroy(g);
    memcpy(m->decode_key, key, sizeof(key));
  }
  (void)hipStreamSynchronize(s);
  if (hipGraphLaunch(m->decode_graph, s) != hipSuccess) { m->err = "decode graph launch failed"; return -3; }
  return 0;
}

// ---- probes (diagnostics) ------------------------------------------------------------------------------------------------
int model_probe_count(Model* m) { return (int)m->ex->probes.size(); }
int model_probe_info(Model* m, int i, const char** name, int64_t* rows, int* cols) {
  Exec& e = *m->ex;
  if (i < 0 || i >= (int)e.probes.size()) { m->err = "probe index out of range"; return -1; }
  if (name) *name = e.probes[i].name.c_str();
  if (rows) *rows = e.probes[i].t->rows;
  if (cols) *cols = e.probes[i].t->C;
  return 0;
}
int model_probe_read(Model* m, int i, float* out_f32, hipStream_t s) {
  Exec& e = *m->ex;
  if (i < 0 || i >= (int)e.probes.size() || !out_f32) { m->err = "probe index out of range"; return -1; }
  Tensor* t = e.probes[i].t;
  launch_cast(e.dt, DT_F32, t->p, out_f32, t->rows * t->C, s);
  return 0;
}
int model_probe_set_grad(Model* m, int i, float* gout_f32) {
  Exec& e = *m->ex;
  if (i < 0 || i >= (int)e.probes.size()) { m->err = "probe index out of range"; return -1; }
  e.probes[i].gout = gout_f32;
  return 0;
}
