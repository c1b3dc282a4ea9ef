// HBM-bound kernels of the SATRN path (NHWC activations, 16-byte vector access along channels):
// BatchNorm (batch-stat) forward/backward, activations, depthwise 3x3, stem conv, max-pool, SE gate,
// adaptive 2D positional encoding, LayerNorm, embedding, cross-entropy, packing, clip + AdamW.
#include "common.h"
#include "kernels.h"
#include "tile_dev.h"
#include <atomic>
#include <cstring>
#include <string>
#include <map>
#include <mutex>
#include <tuple>

#define DISPATCH_T(dt, ...)                      \
  do {                                           \
    if ((dt) == DT_BF16) { typedef bf16_t T; __VA_ARGS__; } \
    else { typedef float T; __VA_ARGS__; }       \
  } while (0)



DetCtx g_det;
WgPartCtx g_wgpart;
SeBoxCtx g_sebox;
long long g_route[RT_COUNT] = {0};
// a mailbox launch must not be captured into a hipGraph: its tag would be replayed and the stale granules of the previous replay would match
bool se_box_usable(hipStream_t s) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  return hipStreamIsCapturing(s, &st) == hipSuccess && st == hipStreamCaptureStatusNone;
}
unsigned se_next_tag() {   // one number per launch that uses a mailbox, whichever kernel (0 is what a fresh mailbox holds)
  static std::atomic<unsigned> tag{0};
  unsigned t = ++tag;
  while (t == 0) t = ++tag;
  return t;
}
long resident_capacity(const void* kernel, int threads, size_t lds_bytes) {
  static std::mutex mu;
  static std::map<std::tuple<const void*, int, size_t>, long> cache;
  static int cus = 0;
  std::lock_guard<std::mutex> lock(mu);
  if (!cus) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  }
  const auto key = std::make_tuple(kernel, threads, lds_bytes);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds_bytes) != hipSuccess || per_cu < 1) per_cu = 0;
  if (per_cu >= 7) per_cu -= 1;
  if (per_cu > 8) per_cu = 8;
  const long cap = (long)per_cu * cus;
  cache[key] = cap;
  return cap;
}
// The four lists are read from the environment by sw_refresh() -- once per C-ABI call (operator entry points, every engine entry point) --
// and kept parsed-ready here: a launch consults the snapshot (an empty list answers at once) instead of scanning `environ` several times
// per launch (6-8 getenv calls of ~0.3 us each per product: a third of the host's time per launch on the launch-bound paths, e.g. the
// 16 000 launches of the autoregressive training branch).
static std::string g_sw[4];   // SATRN_OFF, SATRN_KNOBS, SATRN_PROF, SATRN_TIMING
void sw_refresh() {
  static const char* names[4] = {"SATRN_OFF", "SATRN_KNOBS", "SATRN_PROF", "SATRN_TIMING"};
  for (int i = 0; i < 4; ++i) {
    const char* v = getenv(names[i]);
    if (!v) { if (!g_sw[i].empty()) g_sw[i].clear(); }
    else if (g_sw[i] != v) g_sw[i] = v;
  }
}
static const bool g_sw_init = (sw_refresh(), true);
// -> just behind `name` inside list `which` (at '=', a separator or the end), or null
static const char* sw_find(int which, const char* name) {
  if (g_sw[which].empty()) return nullptr;
  const char* s = g_sw[which].c_str();
  const size_t n = strlen(name);
  while (*s) {
    while (*s == ',' || *s == ' ') ++s;
    if (!strncmp(s, name, n) && (s[n] == 0 || s[n] == ',' || s[n] == ' ' || s[n] == '=')) return s + n;
    while (*s && *s != ',' && *s != ' ') ++s;
  }
  return nullptr;
}
bool sw_off(const char* name) { return sw_find(0, name) != nullptr; }
bool sw_knob_set(const char* name) { const char* p = sw_find(1, name); return p && *p == '='; }
long sw_knob(const char* name, long dflt) { const char* p = sw_find(1, name); return (p && *p == '=') ? atol(p + 1) : dflt; }
double sw_knobf(const char* name, double dflt) { const char* p = sw_find(1, name); return (p && *p == '=') ? atof(p + 1) : dflt; }
const char* sw_knob_str(const char* name) { const char* p = sw_find(1, name); return (p && *p == '=') ? p + 1 : nullptr; }
bool sw_prof(const char* name) { return sw_find(2, name) != nullptr; }
int sw_timing(const char* name) {
  const char* p = sw_find(3, name);
  if (!p) return 0;
  const int x = *p == '=' ? atoi(p + 1) : 1;
  if (x) fprintf(stderr, "[satrn] WARNING: SATRN_TIMING %s=%d -- a timing experiment: work is skipped and results are WRONG\n", name, x);
  return x;
}
void det_overflow_warn(size_t need_floats) {
  static bool once = false;
  if (!once) { once = true; fprintf(stderr, "[satrn] deterministic-reduction scratch too small (%zu floats needed, %zu available): this reduction falls back to float atomics\n", need_floats, g_det.cap); }
}
__global__ void fold_kernel(const float* part, int nrep, long stride, long n, float* out) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float a = 0.f;
    for (int r = 0; r < nrep; ++r) a += part[(size_t)r * stride + i];
    out[i] += a;
  }
}
// Many slots (the deterministic mode's BatchNorm statistics of a tall product: one slot per 64 rows, up to 6 144 of them, 2 N <= 3 072 columns):
// fold_kernel walked them with ONE thread per column -- thousands of serial loads, 0.1-1.4 ms per fold and a third of the f32 step's kernel time
// (profiles/r04_f32_kernel_stats.csv).  Here a block owns FOLD_CB columns x a range of R slots: 256 threads = FOLD_CB column lanes x row lanes,
// every row lane adds its slots in ascending order, the row lanes are added in ascending order through LDS.  Level 1 (grid.y = ceil(nrep / R)
// row ranges) leaves each range's sum IN PLACE in the range's first slot; level 2 (one range over those first slots) adds them to out.  The
// order of every addition depends on (nrep, n) only: bit-identical from run to run.
#define FOLD_CB 32
__global__ __launch_bounds__(256) void fold_par_kernel(float* part, int nrows, long row_step, long stride, long n, int R, float* out) {
  constexpr int RL = 256 / FOLD_CB;
  __shared__ float red[RL][FOLD_CB];
  const int cq = threadIdx.x % FOLD_CB, rl = threadIdx.x / FOLD_CB;
  const long c = (long)blockIdx.x * FOLD_CB + cq;
  const int r0 = blockIdx.y * R, r1 = min(nrows, r0 + R);
  float a = 0.f;
  if (c < n) {
    int r = r0 + rl;
    for (; r + 3 * RL < r1; r += 4 * RL) {   // four slots requested together, added in slot order
      const float v0 = part[(size_t)r * row_step * stride + c], v1 = part[(size_t)(r + RL) * row_step * stride + c];
      const float v2 = part[(size_t)(r + 2 * RL) * row_step * stride + c], v3 = part[(size_t)(r + 3 * RL) * row_step * stride + c];
      a = (((a + v0) + v1) + v2) + v3;
    }
    for (; r < r1; r += RL) a += part[(size_t)r * row_step * stride + c];
  }
  red[rl][cq] = a;
  __syncthreads();
  if (rl == 0 && c < n) {
    float t = red[0][cq];
#pragma unroll
    for (int l = 1; l < RL; ++l) t += red[l][cq];
    if (out) out[c] += t;
    else part[(size_t)r0 * row_step * stride + c] = t;
  }
}
void launch_fold(const float* part, int nrep, long stride, long n, float* out, hipStream_t s) {
  if (n <= 0 || nrep <= 0) return;
  if (nrep < 32) {
    long g = (n + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(fold_kernel, dim3((int)g), dim3(256), 0, s, part, nrep, stride, n, out);
    return;
  }
  float* pw = const_cast<float*>(part);   // (scratch slabs: level 1 overwrites the first slot of every range with the range's sum)
  const int gx = (int)((n + FOLD_CB - 1) / FOLD_CB);
  constexpr int R = 256;
  if (nrep <= R) {
    hipLaunchKernelGGL(fold_par_kernel, dim3(gx, 1), dim3(256), 0, s, pw, nrep, 1L, stride, n, nrep, out);
    return;
  }
  const int S = (nrep + R - 1) / R;
  hipLaunchKernelGGL(fold_par_kernel, dim3(gx, S), dim3(256), 0, s, pw, nrep, 1L, stride, n, R, (float*)nullptr);
  hipLaunchKernelGGL(fold_par_kernel, dim3(gx, 1), dim3(256), 0, s, pw, S, (long)R, stride, n, S, out);
}

// G slice groups per column quad: many slices of a small matrix (SwinTRN stage 1: 256 slices of 96 x 96) would otherwise be a handful of
// threads walking hundreds of dependent-free but serial loads each; the groups' sums meet in LDS in group order (fixed order: deterministic)
template <int G>
__global__ __launch_bounds__(256) void fold4_kernel(const float4* part, int nrep, long stride4, long n4, float4* out) {
  constexpr int CPB = 256 / G;   // column quads per block
  __shared__ float4 red[G][CPB];
  const int c = threadIdx.x % CPB, g = threadIdx.x / CPB;
  const long i = (long)blockIdx.x * CPB + c;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4) {
    const int per = (nrep + G - 1) / G, r0 = g * per, r1 = min(nrep, r0 + per);
    int r = r0;
    for (; r + 3 < r1; r += 4) {   // four slices requested together (one dependent round trip per slice otherwise); summed in slice order
      const float4 v0 = part[(size_t)r * stride4 + i], v1 = part[(size_t)(r + 1) * stride4 + i], v2 = part[(size_t)(r + 2) * stride4 + i], v3 = part[(size_t)(r + 3) * stride4 + i];
      a.x = (((a.x + v0.x) + v1.x) + v2.x) + v3.x; a.y = (((a.y + v0.y) + v1.y) + v2.y) + v3.y;
      a.z = (((a.z + v0.z) + v1.z) + v2.z) + v3.z; a.w = (((a.w + v0.w) + v1.w) + v2.w) + v3.w;
    }
    for (; r < r1; ++r) { const float4 v = part[(size_t)r * stride4 + i]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
  }
  if (G > 1) {
    red[g][c] = a;
    __syncthreads();
    if (g == 0) {
#pragma unroll
      for (int q = 1; q < G; ++q) { const float4 v = red[q][c]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
    }
  }
  if (g == 0 && i < n4) {
    float4 o = out[i];
    o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
    out[i] = o;
  }
}
void launch_fold4(const float* part, int nrep, long stride, long n, float* out, hipStream_t s) {
  if (n <= 0 || nrep <= 0) return;
  const long n4 = n / 4;
  // enough blocks for the chip, few enough slices per thread
  const int G = (nrep >= 32 || n4 < 32768) ? (nrep >= 8 ? 16 : 4) : (nrep >= 8 && n4 < 131072 ? 4 : 1);
  if (G == 16) hipLaunchKernelGGL(fold4_kernel<16>, dim3((int)((n4 + 15) / 16)), dim3(256), 0, s, (const float4*)part, nrep, stride / 4, n4, (float4*)out);
  else if (G == 4) hipLaunchKernelGGL(fold4_kernel<4>, dim3((int)((n4 + 63) / 64)), dim3(256), 0, s, (const float4*)part, nrep, stride / 4, n4, (float4*)out);
  else hipLaunchKernelGGL(fold4_kernel<1>, dim3((int)((n4 + 255) / 256)), dim3(256), 0, s, (const float4*)part, nrep, stride / 4, n4, (float4*)out);
}

static inline int grid_for(long work, int per_block = 256, int cap = 4096) {
  long g = (work + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

// =============================================================================================
// generic column reduction over a [M][C] matrix (C % CH == 0): up to NO sums per column.
// block = TX column-chunk threads x TY row threads; partials go through LDS atomics, then one global
// atomicAdd per column per block.  F::operator()(row, col0, float out[NO][CH]) adds the row's terms.
// =============================================================================================
// UNR: rows a thread requests together.  It must DIVIDE the per-thread trip count: an unrolled loop whose trip count is
// smaller than its unroll factor runs only the sequential remainder loop -- one dependent memory round trip per row.
template <typename T, int NO, typename F, int UNR = 4>
__global__ __launch_bounds__(256) void colreduce_kernel(F f, long M, int C, int rows_per_block, int tx_log2,
                                                        float* o0, float* o1, int nmain, float* part) {
  constexpr int CH = TT<T>::CH;
  __shared__ float red[256 * CH];  // [TY][TX*CH]
  const int TX = 1 << tx_log2, TY = 256 >> tx_log2;
  const int tx = threadIdx.x & (TX - 1), ty = threadIdx.x >> tx_log2;
  const int CC = C / CH;
  // consecutive row blocks of one channel group on one XCD (its L2): the depthwise weight gradient's row blocks share their halo image rows
  const int lin = xcd_remap((int)(blockIdx.x + gridDim.x * blockIdx.y), (int)(gridDim.x * gridDim.y));
  const int bx = lin % (int)gridDim.x, by = lin / (int)gridDim.x;
  const int cbase = by * TX;
  const int c = cbase + tx;
  float acc[NO][CH];
#pragma unroll
  for (int k = 0; k < NO; ++k)
#pragma unroll
    for (int j = 0; j < CH; ++j) acc[k][j] = 0.f;
  const long r0 = (long)bx * rows_per_block;
  long r1 = r0 + rows_per_block;
  if (r1 > M) r1 = M;
  if (c < CC) {
    f.prep(c * CH);
#pragma unroll UNR
    for (long r = r0 + ty; r < r1; r += TY) f(r, c * CH, acc);
  }
#pragma unroll
  for (int k = 0; k < NO; ++k) {
#pragma unroll
    for (int j = 0; j < CH; ++j) red[(ty * TX + tx) * CH + j] = acc[k][j];
    __syncthreads();
    for (int i = threadIdx.x; i < TX * CH; i += 256) {
      float sum = 0.f;
      for (int y = 0; y < TY; ++y) sum += red[y * TX * CH + i];
      int col = cbase * CH + i;
      if (col < C) {
        // deterministic mode: this row block's partial goes to its own slot [row block][k][col]; colreduce_fold_kernel
        // adds the row blocks in ascending order
        if (part) { part[((size_t)bx * NO + k) * C + col] = sum; continue; }
        // nmain < 0: k-major o0[k*C + col] (contiguous atomics).  Else sums k < nmain go to o0[col*nmain + k] and the
        // remaining one to o1[col]
        if (nmain < 0) atomicAdd(o0 + (long)k * C + col, sum);
        else if (k < nmain) atomicAdd(o0 + (long)col * nmain + k, sum);
        else if (o1) atomicAdd(o1 + col, sum);
      }
    }
    __syncthreads();
  }
}

__global__ void colreduce_fold_kernel(const float* part, int gx, int row_step, int NO, int C, float* o0, float* o1, int nmain) {
  const long n = (long)NO * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i / C), col = (int)(i - (long)k * C);
    float a = 0.f;
    for (int r = 0; r < gx; ++r) a += part[(size_t)r * row_step * n + i];
    if (nmain < 0) o0[(long)k * C + col] += a;
    else if (k < nmain) o0[(long)col * nmain + k] += a;
    else if (o1) o1[col] += a;
  }
}

template <typename T, int NO, typename F>
static void launch_colreduce(F f, long M, int C, float* o0, float* o1, int nmain, hipStream_t s, int rows_per_thread = 8,
                             int target_blocks = 512, int max_blocks = 1024) {
  constexpr int CH = TT<T>::CH;
  int CC = C / CH;
  // at most 32 column chunks per block (>= 8 row lanes), >= 32 rows per thread when M allows: keeps the number of
  // global atomics (NO * TX * CH per block) small next to the streamed bytes
  int txl = 0;
  while ((1 << txl) < CC && txl < 5) ++txl;
  int TX = 1 << txl, TY = 256 >> txl;
  int gy = (CC + TX - 1) / TX;
  // ~4 rows per thread, at most ~1024 blocks: these reductions are latency-bound per thread, and the float atomics
  // they end with (NO * TX * CH per block) are cheap next to the streamed bytes
  // rows per thread: as many as keep >= ~512 blocks in flight (these kernels are latency-bound, not byte-bound)
  long rpt = (M * gy) / ((long)TY * target_blocks);
  if (rpt < 1) rpt = 1;
  if (rpt > rows_per_thread) rpt = rows_per_thread;
  constexpr int force_rpt = 0;
  if (force_rpt > 0 && NO <= 2) rpt = force_rpt;
  long rpb = (long)TY * rpt;
  long gx = (M + rpb - 1) / rpb;
  long cap = max_blocks / gy < 1 ? 1 : max_blocks / gy;
  if (gx > cap) {
    gx = cap;
    rpt = ((M + gx - 1) / gx + TY - 1) / TY;
    rpb = rpt * TY;
    gx = (M + rpb - 1) / rpb;
  }
  // the row loop is unrolled by the largest of 4 / 2 / 1 that DIVIDES the rows per thread (see colreduce_kernel)
  float* part = det_scratch(s, (size_t)gx * NO * C);
  if (NO <= 2 && rpt % 4 != 0 && rpt % 2 == 0)
    hipLaunchKernelGGL((colreduce_kernel<T, NO, F, (NO <= 2 ? 2 : 4)>), dim3((int)gx, gy), dim3(256), 0, s, f, M, C, (int)rpb, txl, o0, o1, nmain, part);
  else if (NO <= 2 && rpt % 2 != 0 && rpt < 4)
    hipLaunchKernelGGL((colreduce_kernel<T, NO, F, (NO <= 2 ? 1 : 4)>), dim3((int)gx, gy), dim3(256), 0, s, f, M, C, (int)rpb, txl, o0, o1, nmain, part);
  else
    hipLaunchKernelGGL((colreduce_kernel<T, NO, F, 4>), dim3((int)gx, gy), dim3(256), 0, s, f, M, C, (int)rpb, txl, o0, o1, nmain, part);
  if (part) {
    long n = (long)NO * C;
    // (row blocks in ranges of 32 first -- fold_par_kernel, in place -- then the ranges: one thread per output walked up to 1 024 partials)
    constexpr int FR = 32;
    if (gx > 2 * FR) {
      const int S = (int)((gx + FR - 1) / FR);
      hipLaunchKernelGGL(fold_par_kernel, dim3((int)((n + FOLD_CB - 1) / FOLD_CB), S), dim3(256), 0, s, part, (int)gx, 1L, n, n, FR, (float*)nullptr);
      hipLaunchKernelGGL(colreduce_fold_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, s, part, S, FR, NO, C, o0, o1, nmain);
    } else
    hipLaunchKernelGGL(colreduce_fold_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, s, part, (int)gx, 1, NO, C, o0, o1, nmain);
  }
}

// ---- BN batch statistics ------------------------------------------------------------------
template <typename T> struct StatsF {
  const T* y; int C;
  __device__ void prep(int) {}
  __device__ void operator()(long r, int c0, float (*acc)[TT<T>::CH]) const {
    float v[TT<T>::CH];
    unpack<T>(ld16(y + r * C + c0), v);
#pragma unroll
    for (int j = 0; j < TT<T>::CH; ++j) { acc[0][j] += v[j]; acc[1][j] += v[j] * v[j]; }
  }
};
void launch_colstats(int dt, const void* y, long M, int C, float* sums, hipStream_t s) {
  DISPATCH_T(dt, { StatsF<T> f{(const T*)y, C}; launch_colreduce<T, 2>(f, M, C, sums, sums + C, 1, s); });
}

__global__ void bn_finalize_kernel(const float* sums, float invM, float unbias, int C, const float* w, const float* b,
                                   float* rm, float* rv, int64_t* nbt, float eps, float mom, int train, float* ss,
                                   float* mr) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float mean, var;
  if (train) {
    mean = sums[c] * invM;
    var = fmaxf(sums[C + c] * invM - mean * mean, 0.f);
    rm[c] = (1.f - mom) * rm[c] + mom * mean;
    rv[c] = (1.f - mom) * rv[c] + mom * var * unbias;
    if (c == 0 && nbt) *nbt += 1;
  } else {
    mean = rm[c];
    var = rv[c];
  }
  float rstd = rsqrtf(var + eps);
  float sc = w[c] * rstd;
  ss[c] = sc;
  ss[C + c] = b[c] - mean * sc;
  mr[c] = mean;
  mr[C + c] = rstd;
}
void launch_bn_finalize(const float* sums, long M, int C, const float* w, const float* b, float* rm, float* rv,
                        int64_t* nbt, float eps, float mom, int train, float* ss, float* mr, hipStream_t s) {
  float unbias = M > 1 ? (float)((double)M / (double)(M - 1)) : 1.f;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, s, sums, 1.0f / (float)M, unbias, C, w, b,
                     rm, rv, nbt, eps, mom, train, ss, mr);
}

__global__ void bn_eval_prepare_kernel(const BnEvalDesc* descs) {
  const BnEvalDesc d = descs[blockIdx.x];
  for (int c = threadIdx.x; c < d.C; c += blockDim.x) {
    const float sc = d.w[c] * rsqrtf(d.rv[c] + d.eps);
    d.out[c] = sc;
    d.out[d.C + c] = d.b[c] - d.rm[c] * sc;
  }
}
void launch_bn_eval_prepare(const BnEvalDesc* descs_dev, int n, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(bn_eval_prepare_kernel, dim3(n), dim3(256), 0, s, descs_dev);
}

// grid for streaming kernels with per-channel parameters: total threads is a multiple of the chunk count CC, so a
// thread always meets the same channel chunk and keeps its parameters in registers
static inline int grid_chan(long nchunks, int CC) {
  int g = grid_for(nchunks);
  int a = CC, b = 256;
  while (b) { int t = a % b; a = b; b = t; }
  int q = CC / a;
  return ((g + q - 1) / q) * q;
}

#define BN_REP_MAXC 512  // widest output that uses replicated statistics is 256 channels
template <typename T>
__global__ void bn_act_kernel(const T* __restrict__ y, const float* sums, int sums_rep, const float* w, const float* b, float* rm, float* rv,
                              int64_t* nbt, float eps, float mom, float invM, float unbias, float* ss, float* mr,
                              const T* __restrict__ res, T* __restrict__ z, long nchunks, int C, int act) {
  constexpr int CH = TT<T>::CH;
  __shared__ float s_rep[2 * BN_REP_MAXC];
  const int CC = C / CH;
  const long tid = blockIdx.x * (long)blockDim.x + threadIdx.x, nth = (long)gridDim.x * blockDim.x;
  const int c0 = (int)(tid % CC) * CH;
  // the first chunk of the stream is requested BEFORE the statistics: in the late stages a thread has exactly one chunk, and
  // a request issued behind the finalize (whose stores it may not overtake) made the kernel two far round trips instead of one
  uint4 yv = zero16(), rv4 = zero16();
  if (tid < nchunks) {
    yv = ld16(y + tid * CH);
    if (res) rv4 = ld16(res + tid * CH);
  }
  // the publishing threads also fetch the running statistics now: the read-modify-write at the end is then stores only
  float rm0[CH], rv0[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) rm0[j] = rv0[j] = 0.f;
  if (tid < CC && sums) { ldv(rm + c0, rm0, CH); ldv(rv + c0, rv0, CH); }
  // BatchNorm finalize, redone by every thread for its own channel chunk (a handful of loads); the first CC threads
  // also publish scale/shift + mean/rstd for the backward and update the running statistics (after the stream)
  float sc[CH], sh[CH], mean[CH], var[CH];
  {
    float ww[CH], bb[CH];
    if (sums) {
      if (sums_rep > 1 && C <= BN_REP_MAXC) {
        // replicated statistics (tall narrow outputs): the BLOCK sums the replicas once into LDS.  Every thread doing it for
        // its own chunk is 4 * rep load instructions per wave -- with rep = 16 that preamble alone kept the load units busy
        // for ~20 us (the C = 24 / 48 residual passes ran at 0.8-1.3 TB/s in the step against 2.9-3.7 in isolation)
        for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
          float a = 0.f;
#pragma unroll 4
          for (int rp = 0; rp < sums_rep; ++rp) a += sums[(size_t)rp * 2 * C + i];
          s_rep[i] = a;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < CH; ++j) { mean[j] = s_rep[c0 + j]; var[j] = s_rep[C + c0 + j]; }
      } else {
        ldv(sums + c0, mean, CH); ldv(sums + C + c0, var, CH);
        for (int rp = 1; rp < sums_rep; ++rp) {
          float t0[CH], t1[CH];
          ldv(sums + (size_t)rp * 2 * C + c0, t0, CH); ldv(sums + (size_t)rp * 2 * C + C + c0, t1, CH);
#pragma unroll
          for (int j = 0; j < CH; ++j) { mean[j] += t0[j]; var[j] += t1[j]; }
        }
      }
#pragma unroll
      for (int j = 0; j < CH; ++j) { mean[j] *= invM; var[j] = fmaxf(var[j] * invM - mean[j] * mean[j], 0.f); }
    } else {
      ldv(rm + c0, mean, CH); ldv(rv + c0, var, CH);
    }
    ldv(w + c0, ww, CH); ldv(b + c0, bb, CH);
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      float rstd = rsqrtf(var[j] + eps);
      sc[j] = ww[j] * rstd;
      sh[j] = bb[j] - mean[j] * sc[j];
    }
  }
  for (long i = tid; i < nchunks; i += nth) {
    const long nx = i + nth;
    uint4 yn = zero16(), rn = zero16();
    if (nx < nchunks) {  // next chunk in flight while this one is computed
      yn = ld16(y + nx * CH);
      if (res) rn = ld16(res + nx * CH);
    }
    float v[CH], r[CH];
    unpack<T>(yv, v);
    if (res) unpack<T>(rv4, r);
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      float u = act_fwd(v[j] * sc[j] + sh[j], act);
      v[j] = res ? u + r[j] : u;
    }
    st16(z + i * CH, pack<T>(v));
    yv = yn; rv4 = rn;
  }
  if (tid < CC) {
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      ss[c0 + j] = sc[j]; ss[C + c0 + j] = sh[j]; mr[c0 + j] = mean[j]; mr[C + c0 + j] = rsqrtf(var[j] + eps);
      if (sums) {
        rm[c0 + j] = (1.f - mom) * rm0[j] + mom * mean[j];
        rv[c0 + j] = (1.f - mom) * rv0[j] + mom * var[j] * unbias;
      }
    }
  }
  if (tid == 0 && sums && nbt) *nbt += 1;
}
void launch_bn_act(int dt, const void* y, const float* sums, int sums_rep, const float* w, const float* b, float* rm,
                   float* rv, int64_t* nbt, float eps, float mom, float* ss, float* mr, const void* res, void* z, long M,
                   int C, int act, hipStream_t s) {
  float unbias = M > 1 ? (float)((double)M / (double)(M - 1)) : 1.f;
  DISPATCH_T(dt, {
    long n = M * C / TT<T>::CH;
    int CC = C / TT<T>::CH;
    constexpr int gdiv = 1;   // chunks per thread
    int g = grid_chan((n + gdiv - 1) / gdiv, CC);
    while ((long)g * 256 < CC) g *= 2;
    hipLaunchKernelGGL((bn_act_kernel<T>), dim3(g), dim3(256), 0, s, (const T*)y, sums, sums_rep < 1 ? 1 : sums_rep, w, b, rm, rv, nbt, eps, mom,
                       1.0f / (float)M, unbias, ss, mr, (const T*)res, (T*)z, n, C, act);
  });
}

// BatchNorm(batch statistics) + activation with the squeeze-and-excite average pool of the RESULT accumulated on the way:
// poolsum[b][c] += sum over the block's rows of z.  Tile form (32 channel chunks x 8 row lanes, 48 rows per workgroup; HW % 48
// == 0 so a workgroup never straddles two images).  The pool used to be the first phase of the per-image SE kernel, where ONE
// workgroup pulled a whole image (up to 368 KB) through one CU: 5-8 us of the 17 that kernel took on the dependent chain.
#define BNP_ROWS 48
template <typename T, int ROWS = BNP_ROWS>
__global__ __launch_bounds__(256) void bn_act_pool_kernel(const T* __restrict__ y, const float* sums, int sums_rep, const float* w, const float* b,
                                                          float* rm, float* rv, int64_t* nbt, float eps, float mom, float invM, float unbias,
                                                          float* ss, float* mr, T* __restrict__ z, float* poolsum, int C, int HW, int act) {
  constexpr int CH = TT<T>::CH;
  constexpr int RPT = ROWS / 8;
  __shared__ float sred[8][32 * CH];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int CC = C / CH, cc = blockIdx.y * 32 + tx, c0 = cc * CH;
  const long r0 = (long)blockIdx.x * ROWS;
  const int img = (int)(r0 / HW);
  const bool cok = cc < CC;
  uint4 yv[RPT];
#pragma unroll
  for (int k = 0; k < RPT; ++k) yv[k] = cok ? ld16(y + (r0 + ty + 8 * k) * C + c0) : zero16();
  const bool pub = blockIdx.x == 0 && ty == 0 && cok;
  float rm0[CH], rv0[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) rm0[j] = rv0[j] = 0.f;
  if (pub) { ldv(rm + c0, rm0, CH); ldv(rv + c0, rv0, CH); }
  float sc[CH], sh[CH], mean[CH], var[CH], acc[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) { sc[j] = sh[j] = mean[j] = var[j] = acc[j] = 0.f; }
  if (cok) {
    float ww[CH], bb[CH];
    ldv(sums + c0, mean, CH); ldv(sums + C + c0, var, CH);
    for (int rp = 1; rp < sums_rep; ++rp) {
      float t0[CH], t1[CH];
      ldv(sums + (size_t)rp * 2 * C + c0, t0, CH); ldv(sums + (size_t)rp * 2 * C + C + c0, t1, CH);
#pragma unroll
      for (int j = 0; j < CH; ++j) { mean[j] += t0[j]; var[j] += t1[j]; }
    }
    ldv(w + c0, ww, CH); ldv(b + c0, bb, CH);
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      mean[j] *= invM; var[j] = fmaxf(var[j] * invM - mean[j] * mean[j], 0.f);
      const float rstd = rsqrtf(var[j] + eps);
      sc[j] = ww[j] * rstd;
      sh[j] = bb[j] - mean[j] * sc[j];
    }
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      float v[CH];
      unpack<T>(yv[k], v);
#pragma unroll
      for (int j = 0; j < CH; ++j) v[j] = act_fwd(v[j] * sc[j] + sh[j], act);
      const uint4 o = pack<T>(v);
      st16(z + (r0 + ty + 8 * k) * C + c0, o);
      float r[CH];
      unpack<T>(o, r);  // the pool averages the STORED values (what the separate pooling pass read)
#pragma unroll
      for (int j = 0; j < CH; ++j) acc[j] += r[j];
    }
  }
#pragma unroll
  for (int j = 0; j < CH; ++j) sred[ty][tx * CH + j] = acc[j];
  __syncthreads();
  {
    const int c = threadIdx.x, col = blockIdx.y * 32 * CH + c;
    if (c < 32 * CH && col < C) {
      float sum = 0.f;
#pragma unroll
      for (int t = 0; t < 8; ++t) sum += sred[t][c];
      atomicAdd(poolsum + (long)img * C + col, sum);
    }
  }
  if (pub) {
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      ss[c0 + j] = sc[j]; ss[C + c0 + j] = sh[j]; mr[c0 + j] = mean[j]; mr[C + c0 + j] = rsqrtf(var[j] + eps);
      rm[c0 + j] = (1.f - mom) * rm0[j] + mom * mean[j];
      rv[c0 + j] = (1.f - mom) * rv0[j] + mom * var[j] * unbias;
    }
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
}
bool bn_act_pool_ok(long M, int C, int HW) {
  const bool off = sw_off("fused_pool");   // read per call: tests compare the fused and the plain forms in one process
  return !off && !g_det.on && HW > 0 && (HW % BNP_ROWS) == 0 && (M % HW) == 0;
}
// The same pass as an image kernel for the small maps of the late stages (bf16): workgroup = one image x 64 channels (full 128-byte
// lines), PPT pixels per thread all requested up front, coefficients derived once per workgroup -- and the pool of the image is
// complete inside the workgroup: plain stores, no atomics on poolsum.
template <int PPT>
__global__ __launch_bounds__(256) void bn_pool_img_kernel(const bf16_t* __restrict__ y, const float* sums, int sums_rep, const float* w, const float* b,
                                                          float* rm, float* rv, int64_t* nbt, float eps, float mom, float invM, float unbias,
                                                          float* ss, float* mr, bf16_t* __restrict__ z, float* poolsum, int C, int HW, int act) {
  typedef bf16_t T;
  constexpr int CH = 8, SC = 8;
  __shared__ __attribute__((aligned(16))) float cf[2][SC * CH];
  __shared__ float sred[4][SC * CH];
  const int tid = threadIdx.x, NT = blockDim.x, G = NT / SC;
  const int chunk = tid % SC, g = tid / SC;
  const int img = blockIdx.x, cb = blockIdx.y * SC * CH, c0 = cb + chunk * CH;
  const long base = (long)img * HW * C + c0;
  uint4 raw[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) raw[k] = ld16(y + base + (long)(g + k * G) * C);
  for (int c = tid; c < SC * CH; c += NT) {   // one thread per channel: BatchNorm finalize (bn_act_kernel's arithmetic)
    const int cg = cb + c;
    float mean = 0.f, var = 0.f;
    for (int rp = 0; rp < sums_rep; ++rp) { mean += sums[(size_t)rp * 2 * C + cg]; var += sums[(size_t)rp * 2 * C + C + cg]; }
    mean *= invM; var = fmaxf(var * invM - mean * mean, 0.f);
    const float rstd = rsqrtf(var + eps), sc = w[cg] * rstd, sh = b[cg] - mean * sc;
    cf[0][c] = sc; cf[1][c] = sh;
    if (img == 0) {
      ss[cg] = sc; ss[C + cg] = sh; mr[cg] = mean; mr[C + cg] = rstd;
      rm[cg] = (1.f - mom) * rm[cg] + mom * mean;
      rv[cg] = (1.f - mom) * rv[cg] + mom * var * unbias;
    }
  }
  __syncthreads();
  float sc[CH], sh[CH], acc[CH];
  lds8(cf[0] + chunk * CH, sc); lds8(cf[1] + chunk * CH, sh);
#pragma unroll
  for (int j = 0; j < CH; ++j) acc[j] = 0.f;
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    float v[CH];
    unpack<T>(raw[k], v);
#pragma unroll
    for (int j = 0; j < CH; ++j) v[j] = act_fwd(v[j] * sc[j] + sh[j], act);
    const uint4 o = pack<T>(v);
    st16(z + base + (long)(g + k * G) * C, o);
    float r[CH];
    unpack<T>(o, r);   // the pool averages the STORED values
#pragma unroll
    for (int j = 0; j < CH; ++j) acc[j] += r[j];
  }
#pragma unroll
  for (int o = SC; o < 64; o <<= 1)
#pragma unroll
    for (int j = 0; j < CH; ++j) acc[j] += __shfl_xor(acc[j], o, 64);
  const int lane = tid & 63, wave = tid >> 6;
  if (lane < SC) {
#pragma unroll
    for (int j = 0; j < CH; ++j) sred[wave][lane * CH + j] = acc[j];
  }
  __syncthreads();
  for (int c = tid; c < SC * CH; c += NT) {
    float sum = 0.f;
    for (int wv = 0; wv < NT / 64; ++wv) sum += sred[wv][c];
    poolsum[(long)img * C + cb + c] = sum;
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0 && nbt) *nbt += 1;
}

// ---- BatchNorm (batch statistics) + activation + the WHOLE squeeze-and-excite block in one launch, for the small maps of the late stages
// (bn_pool_img_kernel + se_mlp_scale_kernel: 6.6 + 13.2 us per MBConv block, two launch floors, the activated tensor written and read back).
// Workgroup = one image x 64 channels as in bn_pool_img_kernel; the activated tile stays in registers while the image's squeeze-and-
// excite MLP happens BETWEEN the workgroups of that image:
//   1. every workgroup computes ITS 64 channels' share of the hidden layer (S <= 64 partial dot products over its own pool: no wait) and
//      publishes it as {tag, f32} granules (8-byte relaxed agent-scope stores: the data is the flag, the hand-off form of the pipelined
//      decoder, cdna_hip_programming.md Guideline 16 R2; tag = a per-launch number, so the mailbox is never cleared);
//   2. every workgroup gathers the image's C/64 x S partial sums in a fixed order -> hidden layer -> the gates of its 64 channels -> tile * gate.
// (A first form with two hand-offs -- pool sums to a per-image workgroup that computed the hidden layer alone -- took 43 us: its row loop
// over the reduce matrix was one dependent round trip per row.)
// The image's workgroups wait for each other, so all of them must become resident: the launcher only takes grids that fit the chip
// several times over (other kernels holding CUs delay them but finish on their own).  Every wait is bounded by the wall clock: on a
// timeout the kernel flags g_satrn_errflag bit 2 and returns (wrong values, reported by the next read_loss) instead of hanging.
// Arithmetic and rounding are those of the two kernels it replaces (pool over the STORED bf16 values, gate rounded to bf16).
extern __device__ unsigned g_satrn_errflag;   // (defined with device_error_read_clear below)
struct BnSeP {
  const bf16_t* y; const float* sums; int sums_rep; const float* w; const float* b; float* rm; float* rv; int64_t* nbt;
  float eps, mom, invM, unbias; float* ss; float* mr; bf16_t* z /*null: the activated tensor is not kept*/;
  const bf16_t* W1; const float* b1; const bf16_t* W2; const float* b2;
  float* pooled; float* u1; float* s1; bf16_t* gate; bf16_t* out;
  unsigned long long* box;   // [B][C / 64][64] hidden-layer partial sums as granules
  unsigned tag; long long timeout_ticks;
  int B, C, HW, S, act;
};
template <int PPT>
__global__ __launch_bounds__(256) void bn_pool_se_img_kernel(BnSeP p) {
  typedef bf16_t T;
  constexpr int CH = 8, SC = 8;
  __shared__ __attribute__((aligned(16))) float cf[2][SC * CH];
  __shared__ float sred[4][SC * CH];
  __shared__ __attribute__((aligned(16))) float ps[SC * CH];   // this workgroup's pooled means
  __shared__ float hq[4][64];                                  // hidden-layer partial sums per group lane
  __shared__ __attribute__((aligned(16))) float hs[64];        // hidden layer
  __shared__ __attribute__((aligned(16))) float gl[SC * CH]; // this workgroup's gates (as stored: bf16-rounded)
  const int tid = threadIdx.x, NT = blockDim.x, G = NT / SC, lane = tid & 63, wave = tid >> 6;
  const int chunk = tid % SC, g = tid / SC;
  const int C = p.C, HW = p.HW, S = p.S;
  const int img = blockIdx.x, cb = blockIdx.y * SC * CH, c0 = cb + chunk * CH;
  const long base = (long)img * HW * C + c0;
  const long long t_end = (long long)wall_clock64() + p.timeout_ticks;
  uint4 raw[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) raw[k] = ld16(p.y + base + (long)(g + k * G) * C);
  // this thread's row of the expand matrix (threads 0..63: one gate each), requested now, used after two hand-offs
  uint4 w2r[8], w1r[8];
  float b2v = 0.f, b1v = 0.f;
  if (tid < SC * CH) {
#pragma unroll
    for (int u = 0; u < 8; ++u) w2r[u] = ld16(p.W2 + (long)(cb + tid) * S + (u * CH < S ? u * CH : 0));
    b2v = p.b2[cb + tid];
    // ... and, threads 0..S-1, the 64 columns of reduce-matrix row `tid` that belong to this workgroup's channels
#pragma unroll
    for (int u = 0; u < 8; ++u) w1r[u] = ld16(p.W1 + (long)(tid < S ? tid : 0) * C + cb + u * CH);
    if (tid < S) b1v = p.b1[tid];
  }
  for (int c = tid; c < SC * CH; c += NT) {   // one thread per channel: BatchNorm finalize (bn_act_kernel's arithmetic)
    const int cg = cb + c;
    float mean = 0.f, var = 0.f;
    for (int rp = 0; rp < p.sums_rep; ++rp) { mean += p.sums[(size_t)rp * 2 * C + cg]; var += p.sums[(size_t)rp * 2 * C + C + cg]; }
    mean *= p.invM; var = fmaxf(var * p.invM - mean * mean, 0.f);
    const float rstd = rsqrtf(var + p.eps), sc = p.w[cg] * rstd, sh = p.b[cg] - mean * sc;
    cf[0][c] = sc; cf[1][c] = sh;
    if (img == 0) {
      p.ss[cg] = sc; p.ss[C + cg] = sh; p.mr[cg] = mean; p.mr[C + cg] = rstd;
      p.rm[cg] = (1.f - p.mom) * p.rm[cg] + p.mom * mean;
      p.rv[cg] = (1.f - p.mom) * p.rv[cg] + p.mom * var * p.unbias;
    }
  }
  __syncthreads();
  uint4 zq[PPT];
  {
    float sc[CH], sh[CH], acc[CH];
    lds8(cf[0] + chunk * CH, sc); lds8(cf[1] + chunk * CH, sh);
#pragma unroll
    for (int j = 0; j < CH; ++j) acc[j] = 0.f;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      float v[CH];
      unpack<T>(raw[k], v);
#pragma unroll
      for (int j = 0; j < CH; ++j) v[j] = act_fwd(v[j] * sc[j] + sh[j], p.act);
      zq[k] = pack<T>(v);
      if (p.z) st16(p.z + base + (long)(g + k * G) * C, zq[k]);
      float r[CH];
      unpack<T>(zq[k], r);   // the pool averages the ROUNDED values
#pragma unroll
      for (int j = 0; j < CH; ++j) acc[j] += r[j];
    }
#pragma unroll
    for (int o = SC; o < 64; o <<= 1)
#pragma unroll
      for (int j = 0; j < CH; ++j) acc[j] += __shfl_xor(acc[j], o, 64);
    if (lane < SC) {
#pragma unroll
      for (int j = 0; j < CH; ++j) sred[wave][lane * CH + j] = acc[j];
    }
  }
  __syncthreads();
  // 1. this workgroup's share of the hidden layer: hp[j] = sum over ITS 64 channels of W1[j][c] * mean[c] (its own pool only: no wait),
  //    published as {tag, f32} granules; the pooled means themselves go to the backward's buffer
  if (tid < SC * CH) {
    float sum = 0.f;
    for (int wv = 0; wv < NT / 64; ++wv) sum += sred[wv][tid];
    const float m = sum * (1.0f / (float)HW);
    ps[tid] = m;
    p.pooled[(long)img * C + cb + tid] = m;
  }
  __syncthreads();
  {
    SeXchg xc;
    xc.NG = C / (SC * CH); xc.S = S; xc.tag = p.tag; xc.t_end = t_end; xc.err = &g_satrn_errflag;
    xc.ibox = (se_box_t*)p.box + (size_t)img * xc.NG * 64;
    float uu, sv;
    se_exchange_gates(xc, ps, w1r, w2r, b1v, b2v, hq, hs, gl, uu, sv);
    if (tid < S && blockIdx.y == 0) { p.u1[(long)img * S + tid] = uu; p.s1[(long)img * S + tid] = sv; }
    if (tid < SC * CH) p.gate[(long)img * C + cb + tid] = from_f<bf16_t>(gl[tid]);
  }
  {
    float gv[CH];
    lds8(gl + chunk * CH, gv);
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      float v[CH];
      unpack<T>(zq[k], v);
#pragma unroll
      for (int j = 0; j < CH; ++j) v[j] *= gv[j];
      st16(p.out + base + (long)(g + k * G) * C, pack<T>(v));
    }
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0 && p.nbt) *p.nbt += 1;
}
// false = shape / mode not taken (the caller launches launch_bn_act_pool and the squeeze-and-excite kernels)
bool launch_bn_pool_se(int dt, const void* y, const float* sums, int sums_rep, const float* w, const float* b, float* rm, float* rv, int64_t* nbt,
                       float eps, float mom, float* ss, float* mr, void* z /*may be null*/, const void* W1, const float* b1, const void* W2,
                       const float* b2, float* pooled, float* u1, float* s1, void* gate, void* out, unsigned long long* box, int box_images,
                       int B, int HW, int C, int S, int act, hipStream_t s) {
  const bool off = sw_off("fused_pool_se");   // read per call: tests compare the fused and the plain forms in one process
  if (off || g_det.on || dt != DT_BF16 || !sums || !box || B > box_images || (C % 64) != 0 || C > 1536 || S > 64 || (S % 8) != 0 || HW <= 0) return false;
  if (!se_box_usable(s)) return false;
  BnSeP p;
  p.y = (const bf16_t*)y; p.sums = sums; p.sums_rep = sums_rep < 1 ? 1 : sums_rep; p.w = w; p.b = b; p.rm = rm; p.rv = rv; p.nbt = nbt;
  const long M = (long)B * HW;
  p.eps = eps; p.mom = mom; p.invM = 1.0f / (float)M; p.unbias = M > 1 ? (float)((double)M / (double)(M - 1)) : 1.f;
  p.ss = ss; p.mr = mr; p.z = (bf16_t*)z; p.W1 = (const bf16_t*)W1; p.b1 = b1; p.W2 = (const bf16_t*)W2; p.b2 = b2;
  p.pooled = pooled; p.u1 = u1; p.s1 = s1; p.gate = (bf16_t*)gate; p.out = (bf16_t*)out; p.box = box;
  p.timeout_ticks = 200000000LL;   // 2 s of the 100 MHz wall clock
  p.B = B; p.C = C; p.HW = HW; p.S = S; p.act = act;
  for (int G = 32; G >= 8; G >>= 1) {
    if ((HW % G) != 0 || HW / G > 8 || (G * 8) % 64 != 0) continue;
    const int ppt = HW / G;
    p.tag = se_next_tag();
    const dim3 grid(B, C / 64), blk(G * 8);
    // the image's workgroups wait for each other: the whole grid must be resident at once -- asked of the very instantiation that would run
    // (the 7- and 8-pixel forms hold 176-186 VGPRs: two workgroups per CU, where the 3- to 6-pixel forms hold three)
#define BNSE_IMG(P) case P: if ((long)B * (C / 64) > resident_capacity((const void*)bn_pool_se_img_kernel<P>, G * 8, 0)) return false; \
                            hipLaunchKernelGGL((bn_pool_se_img_kernel<P>), grid, blk, 0, s, p); g_route[RT_BN_POOL_SE]++; return true;
    switch (ppt) { BNSE_IMG(1) BNSE_IMG(2) BNSE_IMG(3) BNSE_IMG(4) BNSE_IMG(5) BNSE_IMG(6) BNSE_IMG(7) BNSE_IMG(8) }
#undef BNSE_IMG
  }
  return false;
}

void launch_bn_act_pool(int dt, const void* y, const float* sums, int sums_rep, const float* w, const float* b, float* rm, float* rv,
                        int64_t* nbt, float eps, float mom, float* ss, float* mr, void* z, float* poolsum, long M, int C, int HW, int act,
                        hipStream_t s) {
  float unbias = M > 1 ? (float)((double)M / (double)(M - 1)) : 1.f;
  constexpr bool img_ok = true;
  if (img_ok && dt == DT_BF16 && (C % 64) == 0 && HW > 0 && (M % HW) == 0 && sums) {
    // pixel lanes: the largest of 32 / 16 / 8 that divides HW with at most 8 pixels per thread
    const int B = (int)(M / HW);
    for (int G = 32; G >= 8; G >>= 1) {
      if ((HW % G) != 0 || HW / G > 8 || (G * 8) % 64 != 0) continue;
      const int ppt = HW / G;
      const dim3 grid(B, C / 64), blk(G * 8);
#define BNP_IMG(P) case P: hipLaunchKernelGGL((bn_pool_img_kernel<P>), grid, blk, 0, s, (const bf16_t*)y, sums, sums_rep < 1 ? 1 : sums_rep, w, b, rm, rv, nbt, eps, mom, \
                                             1.0f / (float)M, unbias, ss, mr, (bf16_t*)z, poolsum, C, HW, act); return;
      switch (ppt) { BNP_IMG(1) BNP_IMG(2) BNP_IMG(3) BNP_IMG(4) BNP_IMG(5) BNP_IMG(6) BNP_IMG(7) BNP_IMG(8) }
#undef BNP_IMG
    }
  }
  DISPATCH_T(dt, {
    const int CC = C / TT<T>::CH;
    // 24 rows per workgroup where 48 would leave the grid under one workgroup per CU (the 4x12 stage: 192 -> 384 workgroups)
    constexpr bool half_ok = true;
    if (half_ok && (M / BNP_ROWS) * ((CC + 31) / 32) < 256)
      hipLaunchKernelGGL((bn_act_pool_kernel<T, BNP_ROWS / 2>), dim3((int)(M / (BNP_ROWS / 2)), (CC + 31) / 32), dim3(256), 0, s, (const T*)y, sums,
                         sums_rep < 1 ? 1 : sums_rep, w, b, rm, rv, nbt, eps, mom, 1.0f / (float)M, unbias, ss, mr, (T*)z, poolsum, C, HW, act);
    else
    hipLaunchKernelGGL((bn_act_pool_kernel<T>), dim3((int)(M / BNP_ROWS), (CC + 31) / 32), dim3(256), 0, s, (const T*)y, sums, sums_rep < 1 ? 1 : sums_rep,
                       w, b, rm, rv, nbt, eps, mom, 1.0f / (float)M, unbias, ss, mr, (T*)z, poolsum, C, HW, act);
  });
}

template <typename T> struct BnBwdRedF {
  const T* dz; const T* y; const float* ss; const float* mr; int C; int act;
  // optional squeeze-and-excite backward folded in: the gradient of the BN output is dz*gate[b] + dpool[b]*scale
  // (what se_bwd_x would have materialised), b = row / se_hw
  const T* se_gate; const T* se_dpool; int se_hw; float se_scale;
  float sc[TT<T>::CH], sh[TT<T>::CH], mu[TT<T>::CH], rs[TT<T>::CH];
  __device__ void prep(int c0) {
    ldv(ss + c0, sc, TT<T>::CH); ldv(ss + C + c0, sh, TT<T>::CH); ldv(mr + c0, mu, TT<T>::CH); ldv(mr + C + c0, rs, TT<T>::CH);
  }
  __device__ void operator()(long r, int c0, float (*acc)[TT<T>::CH]) const {
    constexpr int CH = TT<T>::CH;
    float d[CH], v[CH];
    unpack<T>(ld16(dz + r * C + c0), d);
    unpack<T>(ld16(y + r * C + c0), v);
    if (se_gate) {
      const long b = r / se_hw;
      float gt[CH], dp[CH];
      unpack<T>(ld16(se_gate + b * C + c0), gt);
      unpack<T>(ld16(se_dpool + b * C + c0), dp);
#pragma unroll
      for (int j = 0; j < CH; ++j) d[j] = d[j] * gt[j] + dp[j] * se_scale;
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      float u = v[j] * sc[j] + sh[j];
      float g = d[j] * act_bwd(u, act);
      float xh = (v[j] - mu[j]) * rs[j];
      acc[0][j] += g;
      acc[1][j] += g * xh;
    }
  }
};
void launch_bn_bwd_reduce(int dt, const void* dz, const void* y, const float* ss, const float* mr, long M, int C,
                          int act, float* red, hipStream_t s, const void* se_gate, const void* se_dpool, int se_hw) {
  DISPATCH_T(dt, {
    BnBwdRedF<T> f;
    f.dz = (const T*)dz; f.y = (const T*)y; f.ss = ss; f.mr = mr; f.C = C; f.act = act;
    f.se_gate = (const T*)se_gate; f.se_dpool = (const T*)se_dpool; f.se_hw = se_hw > 0 ? se_hw : 1; f.se_scale = se_hw > 0 ? 1.0f / (float)se_hw : 0.f;
    constexpr int tb = 512;
    launch_colreduce<T, 2>(f, M, C, red, red + C, 1, s, 8, tb, 1024);
  });
}

// dy = A*g + Bc + Cc*y with g = dz*act'(y*scale+shift): per-channel coefficients computed once per thread
template <typename T>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ dz, const T* __restrict__ y, const float* ss, const float* mr, const float* w,
                                    const float* red, int red_rep, float invM, long nchunks, int C, int act, T* __restrict__ dy,
                                    float* dw, float* db, const T* __restrict__ se_gate, const T* __restrict__ se_dpool, int se_hw, float se_scale) {
  constexpr int CH = TT<T>::CH;
  const int CC = C / CH;
  const long tid = blockIdx.x * (long)blockDim.x + threadIdx.x, nth = (long)gridDim.x * blockDim.x;
  const int c0 = (int)(tid % CC) * CH;
  // first chunk of the stream requested before the coefficients (see bn_act_kernel): one far round trip instead of two
  uint4 dzv = zero16(), yv = zero16(), gtv = zero16(), dpv = zero16();
  if (tid < nchunks) {
    dzv = ld16(dz + tid * CH);
    yv = ld16(y + tid * CH);
    if (se_gate) {
      const long b = (tid / CC) / se_hw;
      gtv = ld16(se_gate + b * C + c0);
      dpv = ld16(se_dpool + b * C + c0);
    }
  }
  float dw0 = 0.f, db0 = 0.f;  // parameter-gradient accumulators fetched now, stored after the stream
  if (tid < C && dw) { dw0 = dw[tid]; db0 = db[tid]; }
  __shared__ float s_rep[2 * BN_REP_MAXC];
  const bool lds_rep = red_rep > 1 && C <= BN_REP_MAXC;
  if (lds_rep) {  // the block sums the replicas once (see bn_act_kernel)
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
      float a = 0.f;
#pragma unroll 4
      for (int r = 0; r < red_rep; ++r) a += red[(size_t)r * 2 * C + i];
      s_rep[i] = a;
    }
    __syncthreads();
  }
  float sc[CH], sh[CH], A[CH], Bc[CH], Cc[CH];
  {
    float mu[CH], rs[CH], ww[CH], r0[CH], r1[CH];
    ldv(ss + c0, sc, CH); ldv(ss + C + c0, sh, CH); ldv(mr + c0, mu, CH); ldv(mr + C + c0, rs, CH);
    ldv(w + c0, ww, CH);
    if (lds_rep) {
#pragma unroll
      for (int j = 0; j < CH; ++j) { r0[j] = s_rep[c0 + j]; r1[j] = s_rep[C + c0 + j]; }
    } else {
      ldv(red + c0, r0, CH); ldv(red + C + c0, r1, CH);
      for (int r = 1; r < red_rep; ++r) {
        float t0[CH], t1[CH];
        ldv(red + (size_t)r * 2 * C + c0, t0, CH); ldv(red + (size_t)r * 2 * C + C + c0, t1, CH);
#pragma unroll
        for (int j = 0; j < CH; ++j) { r0[j] += t0[j]; r1[j] += t1[j]; }
      }
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      float a = ww[j] * rs[j], m1 = r0[j] * invM, m2 = r1[j] * invM;
      A[j] = a;
      Cc[j] = -a * rs[j] * m2;
      Bc[j] = -a * m1 + a * rs[j] * mu[j] * m2;
    }
  }
  for (long i = tid; i < nchunks; i += nth) {
    const long nx = i + nth;
    uint4 dzn = zero16(), yn = zero16(), gtn = zero16(), dpn = zero16();
    if (nx < nchunks) {  // next chunk in flight while this one is computed
      dzn = ld16(dz + nx * CH);
      yn = ld16(y + nx * CH);
      if (se_gate) {
        const long b = (nx / CC) / se_hw;
        gtn = ld16(se_gate + b * C + c0);
        dpn = ld16(se_dpool + b * C + c0);
      }
    }
    float d[CH], v[CH];
    unpack<T>(dzv, d);
    unpack<T>(yv, v);
    if (se_gate) {  // squeeze-and-excite backward folded in (see BnBwdRedF)
      float gt[CH], dp[CH];
      unpack<T>(gtv, gt);
      unpack<T>(dpv, dp);
#pragma unroll
      for (int j = 0; j < CH; ++j) d[j] = d[j] * gt[j] + dp[j] * se_scale;
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      float g = d[j] * act_bwd(v[j] * sc[j] + sh[j], act);
      d[j] = A[j] * g + Bc[j] + Cc[j] * v[j];
    }
    st16(dy + i * CH, pack<T>(d));
    dzv = dzn; yv = yn; gtv = gtn; dpv = dpn;
  }
  if (tid < C && dw) {  // parameter grads (grad buffers are zeroed per step: accumulate); after the stream: nothing waits for them
    float a = 0.f, b = 0.f;
    if (lds_rep) { a = s_rep[C + tid]; b = s_rep[tid]; }
    else for (int r = 0; r < red_rep; ++r) { a += red[(size_t)r * 2 * C + C + tid]; b += red[(size_t)r * 2 * C + tid]; }
    dw[tid] = dw0 + a;
    db[tid] = db0 + b;
  }
}
void launch_bn_bwd_apply(int dt, const void* dz, const void* y, const float* ss, const float* mr, const float* w,
                         const float* red, long M, int C, int act, void* dy, float* dw, float* db, hipStream_t s,
                         int red_rep, const void* se_gate, const void* se_dpool, int se_hw, int eval_stats) {
  DISPATCH_T(dt, {
    long n = M * C / TT<T>::CH;
    constexpr int gdiv = 1;   // chunks per thread
    int g = grid_chan((n + gdiv - 1) / gdiv, C / TT<T>::CH);
    while ((long)g * 256 < C) g *= 2;
    // eval statistics: invM = 0 removes the batch-mean / batch-variance terms (Bc = Cc = 0), dw / db stay sum(g*xhat) / sum(g)
    hipLaunchKernelGGL((bn_bwd_apply_kernel<T>), dim3(g), dim3(256), 0, s, (const T*)dz, (const T*)y, ss, mr, w, red,
                       red_rep < 1 ? 1 : red_rep, eval_stats ? 0.0f : 1.0f / (float)M, n, C, act, (T*)dy, dw, db, (const T*)se_gate, (const T*)se_dpool,
                       se_hw > 0 ? se_hw : 1, se_hw > 0 ? 1.0f / (float)se_hw : 0.f);
  });
}

// ---- stem conv (Cin = 1 or 3, 3x3): direct, fp32 image + fp32 master weights -------------------
// One input channel (grey-scale images: the benchmark case): the grid is a multiple of the chunk count, so a thread always meets the same
// 8 output channels and holds their 72 weights in registers.  The generic kernel below fetches every weight through the vector
// memory path for every pixel (81 memory instructions per thread and pixel: 65 us for the 32 x 128 x 384 batch, issue-bound).
template <typename T>
__global__ __launch_bounds__(256) void stem_conv1_kernel(const float* __restrict__ img, const float* __restrict__ w, T* __restrict__ y, int B, int H, int W,
                                                         int Co, int OH, int OW, int stride, int pad) {
  constexpr int CH = TT<T>::CH;
  const int CC = Co / CH;
  const long total = (long)B * OH * OW * CC;
  const long t0 = blockIdx.x * (long)blockDim.x + threadIdx.x, nth = (long)gridDim.x * blockDim.x;
  const int cc = (int)(t0 % CC);   // nth % CC == 0 (launcher)
  float wr[9][CH];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < CH; ++j) wr[t][j] = w[(cc * CH + j) * 9 + t];
  for (long i = t0; i < total; i += nth) {
    const long pix = i / CC;
    const int ox = (int)(pix % OW);
    const int oy = (int)((pix / OW) % OH);
    const int b = (int)(pix / ((long)OW * OH));
    const float* im = img + (long)b * H * W;
    float x[9];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int sy = oy * stride - pad + kh, sx = ox * stride - pad + kw;
        const bool ok = sy >= 0 && sy < H && sx >= 0 && sx < W;
        const float v = im[ok ? (long)sy * W + sx : 0];
        x[kh * 3 + kw] = ok ? v : 0.f;
      }
    float acc[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) acc[j] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t)   // same order as the generic kernel (kh, kw ascending): same sums
#pragma unroll
      for (int j = 0; j < CH; ++j) acc[j] += x[t] * wr[t][j];
    st16(y + pix * Co + cc * CH, pack<T>(acc));
  }
}
template <typename T>
__global__ void stem_conv_kernel(const float* img, const float* w, T* y, int B, int Cin, int H, int W, int Co, int OH,
                                 int OW, int stride, int pad) {
  constexpr int CH = TT<T>::CH;
  const int CC = Co / CH;
  long total = (long)B * OH * OW * CC;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int cc = (int)(i % CC);
    long pix = i / CC;
    int ox = (int)(pix % OW);
    int oy = (int)((pix / OW) % OH);
    int b = (int)(pix / ((long)OW * OH));
    float acc[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) acc[j] = 0.f;
    for (int ci = 0; ci < Cin; ++ci)
      for (int kh = 0; kh < 3; ++kh) {
        int sy = oy * stride - pad + kh;
        if (sy < 0 || sy >= H) continue;
        for (int kw = 0; kw < 3; ++kw) {
          int sx = ox * stride - pad + kw;
          if (sx < 0 || sx >= W) continue;
          float x = img[(((long)b * Cin + ci) * H + sy) * W + sx];
#pragma unroll
          for (int j = 0; j < CH; ++j) acc[j] += x * w[(((cc * CH + j) * Cin + ci) * 3 + kh) * 3 + kw];
        }
      }
    st16(y + pix * Co + cc * CH, pack<T>(acc));
  }
}
void launch_stem_conv(int dt, const float* img, const float* w, void* y, int B, int Cin, int H, int W, int Co, int OH,
                      int OW, int stride, int pad, hipStream_t s) {
  DISPATCH_T(dt, {
    long n = (long)B * OH * OW * (Co / TT<T>::CH);
    constexpr bool one_ok = true;
    if (Cin == 1 && one_ok) {
      // a few pixels per thread amortise the 72 weight loads; total threads a multiple of the chunk count
      const int g = grid_chan((n + 3) / 4, Co / TT<T>::CH);
      hipLaunchKernelGGL((stem_conv1_kernel<T>), dim3(g), dim3(256), 0, s, img, w, (T*)y, B, H, W, Co, OH, OW, stride, pad);
    } else
    hipLaunchKernelGGL((stem_conv_kernel<T>), dim3(grid_for(n)), dim3(256), 0, s, img, w, (T*)y, B, Cin, H, W, Co, OH,
                       OW, stride, pad);
  });
}

// dW[co][ci][kh][kw] += sum_pix dy[pix][co] * img[b][ci][sy][sx].  Per input channel every thread's partial goes to its own
// LDS slot and the pixel lanes are summed in a fixed order (deterministic inside the block); the block's result is added
// with one global atomic per weight, or -- deterministic mode -- stored as a per-block partial and folded afterwards.
template <typename T>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* img, const T* dy, float* dw, int B, int Cin,
                                                         int H, int W, int Co, int OH, int OW, int stride, int pad,
                                                         int pix_per_block, float* part) {
  constexpr int CH = TT<T>::CH;
  extern __shared__ float red[];  // [np][Co*9]
  const int nW = Co * Cin * 9, nC = Co * 9;
  const int CC = Co / CH;
  long p0 = (long)blockIdx.x * pix_per_block;
  long p1 = p0 + pix_per_block;
  long NP = (long)B * OH * OW;
  if (p1 > NP) p1 = NP;
  // thread -> channel chunk (fixed) and a strided set of pixels
  int cc = threadIdx.x % CC;
  int lane_p = threadIdx.x / CC, np = 256 / CC;
  for (int ci = 0; ci < Cin; ++ci) {
    if (lane_p < np) {
      float acc[9][CH];
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < CH; ++j) acc[t][j] = 0.f;
      for (long pix = p0 + lane_p; pix < p1; pix += np) {
        int ox = (int)(pix % OW);
        int oy = (int)((pix / OW) % OH);
        int b = (int)(pix / ((long)OW * OH));
        float d[CH];
        unpack<T>(ld16(dy + pix * Co + cc * CH), d);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          int sy = oy * stride - pad + kh;
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            int sx = ox * stride - pad + kw;
            float x = (sy >= 0 && sy < H && sx >= 0 && sx < W) ? img[(((long)b * Cin + ci) * H + sy) * W + sx] : 0.f;
#pragma unroll
            for (int j = 0; j < CH; ++j) acc[kh * 3 + kw][j] += d[j] * x;
          }
        }
      }
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < CH; ++j) red[(size_t)lane_p * nC + (cc * CH + j) * 9 + t] = acc[t][j];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nC; i += 256) {
      float a = 0.f;
      for (int q = 0; q < np; ++q) a += red[(size_t)q * nC + i];
      const int co = i / 9, t = i - co * 9;
      const long dst = ((long)co * Cin + ci) * 9 + t;
      if (part) part[(size_t)blockIdx.x * nW + dst] = a;
      else atomicAdd(dw + dst, a);
    }
    __syncthreads();
  }
}
void launch_stem_wgrad(int dt, const float* img, const void* dy, float* dw, int B, int Cin, int H, int W, int Co,
                       int OH, int OW, int stride, int pad, hipStream_t s) {
  DISPATCH_T(dt, {
    long NP = (long)B * OH * OW;
    int ppb = 2048;
    int g = (int)((NP + ppb - 1) / ppb);
    const int np = 256 / (Co / TT<T>::CH);
    const int nW = Co * Cin * 9;
    float* part = det_scratch(s, (size_t)g * nW);
    hipLaunchKernelGGL((stem_wgrad_kernel<T>), dim3(g), dim3(256), (size_t)np * Co * 9 * sizeof(float), s, img,
                       (const T*)dy, dw, B, Cin, H, W, Co, OH, OW, stride, pad, ppb, part);
    if (part) launch_fold(part, g, nW, nW, dw, s);
  });
}

// ---- depthwise 3x3 (forward gather / transposed gather for the data gradient) -----------------------
template <typename T, int MODE>
__global__ void dwconv_kernel(const T* x, const T* wp, const float* bias, T* y, int B, int H, int W, int C, int OH,
                              int OW, int stride, int pt, int pl, int beta, const float* esc, const float* esh, int eact) {
  // MODE 0: x = input [B,H,W,C], y = output [B,OH,OW,C].  MODE 1: x = dY [B,H,W,C] (H,W = conv OUTPUT dims),
  // y = dX [B,OH,OW,C] (OH,OW = conv INPUT dims).  One thread per output chunk: these tensors are small and
  // L2-resident, so occupancy (not instruction count) is what hides the nine gathers.
  constexpr int CH = TT<T>::CH;
  const int CC = C / CH;
  long total = (long)B * OH * OW * CC;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int cc = (int)(i % CC);
    long pix = i / CC;
    int ox = (int)(pix % OW);
    int oy = (int)((pix / OW) % OH);
    int b = (int)(pix / ((long)OW * OH));
    float acc[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) acc[j] = (bias && MODE == 0) ? bias[cc * CH + j] : 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        int sy, sx;
        bool ok;
        if (MODE == 0) {
          sy = oy * stride - pt + kh; sx = ox * stride - pl + kw;
          ok = sy >= 0 && sy < H && sx >= 0 && sx < W;
        } else {
          int ty = oy + pt - kh, tx = ox + pl - kw;
          sy = ty / stride; sx = tx / stride;
          ok = ty >= 0 && tx >= 0 && sy * stride == ty && sx * stride == tx && sy < H && sx < W;
        }
        if (ok) {
          float v[CH], wv[CH];
          unpack<T>(ld16(x + (((long)b * H + sy) * W + sx) * C + cc * CH), v);
          unpack<T>(ld16(wp + (kh * 3 + kw) * C + cc * CH), wv);
#pragma unroll
          for (int j = 0; j < CH; ++j) acc[j] += v[j] * wv[j];
        }
      }
    }
    T* o = y + pix * C + cc * CH;
    if (esc) {  // inference: eval-mode BatchNorm + activation folded in
      float sc[CH], sh[CH];
      ldv(esc + cc * CH, sc, CH); ldv(esh + cc * CH, sh, CH);
#pragma unroll
      for (int j = 0; j < CH; ++j) acc[j] = act_fwd(acc[j] * sc[j] + sh[j], eact);
    }
    if (beta) {
      float old[CH];
      unpack<T>(ld16(o), old);
#pragma unroll
      for (int j = 0; j < CH; ++j) acc[j] += old[j];
    }
    st16(o, pack<T>(acc));
  }
}
// stride-1 SAME specialisation (28 of the 32 depthwise convolutions of the backbone, both encoder ones): one thread
// produces TWO horizontally adjacent outputs of a channel chunk, so every loaded input chunk and every unpacked weight
// chunk is used twice (12 loads instead of 18 per pair, half the index arithmetic) -- these kernels are VALU-issue bound.
// FLIP = data gradient (correlation with the kernel rotated by 180 degrees; identical for stride 1, pad 1).
template <typename T, bool FLIP>
__global__ void dwconv_s1_kernel(const T* x, const T* wp, const float* bias, T* y, int B, int H, int W, int C, int beta,
                                 const float* esc, const float* esh, int eact) {
  constexpr int CH = TT<T>::CH;
  const int CC = C / CH, W2 = W >> 1;
  const long total = (long)B * H * W2 * CC;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % CC);
    long pix = i / CC;
    const int ox = (int)(pix % W2) * 2;
    const int oy = (int)((pix / W2) % H);
    const int b = (int)(pix / ((long)W2 * H));
    float a0[CH], a1[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) a0[j] = a1[j] = (bias && !FLIP) ? bias[cc * CH + j] : 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int iy = oy + kh - 1;
      if (iy < 0 || iy >= H) continue;
      const T* row = x + (((long)b * H + iy) * W) * C + cc * CH;
      float in[4][CH];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int ix = ox - 1 + t;
        if (ix >= 0 && ix < W) unpack<T>(ld16(row + (long)ix * C), in[t]);
        else {
#pragma unroll
          for (int j = 0; j < CH; ++j) in[t][j] = 0.f;
        }
      }
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        float wv[CH];
        const int tap = FLIP ? 8 - (kh * 3 + kw) : kh * 3 + kw;
        unpack<T>(ld16(wp + tap * C + cc * CH), wv);
#pragma unroll
        for (int j = 0; j < CH; ++j) { a0[j] += in[kw][j] * wv[j]; a1[j] += in[kw + 1][j] * wv[j]; }
      }
    }
    T* o = y + (((long)b * H + oy) * W + ox) * C + cc * CH;
    if (esc) {  // inference: eval-mode BatchNorm + activation folded in
      float sc[CH], sh[CH];
      ldv(esc + cc * CH, sc, CH); ldv(esh + cc * CH, sh, CH);
#pragma unroll
      for (int j = 0; j < CH; ++j) { a0[j] = act_fwd(a0[j] * sc[j] + sh[j], eact); a1[j] = act_fwd(a1[j] * sc[j] + sh[j], eact); }
    }
    if (beta) {
      float o0[CH], o1[CH];
      unpack<T>(ld16(o), o0); unpack<T>(ld16(o + C), o1);
#pragma unroll
      for (int j = 0; j < CH; ++j) { a0[j] += o0[j]; a1[j] += o1[j]; }
    }
    st16(o, pack<T>(a0));
    st16(o + C, pack<T>(a1));
  }
}

// The same stride-1 convolution as a (32 channel chunks x 8 pixel-pair lanes) tile with a column reduction behind it, so the
// pass that used to follow it on the dependent chain disappears (each such pass is >= 4.5 us of kernel boundary + one far
// round trip, 30 of each per training step):
//   stats[0..C) += sum y, stats[C..2C) += sum y*y            (was launch_colstats behind the convolution)
// (The same treatment of the data gradient -- the BatchNorm-backward sums of its input reduced in the epilogue -- was built and
// measured in round 2: 206 VGPRs, two waves per SIMD, 20 / 37 us against 19 / 30 for the two separate kernels.  Not kept.)
template <typename T>
__global__ __launch_bounds__(256) void dwconv_s1_red_kernel(const T* __restrict__ x, const T* __restrict__ wp, const float* bias, T* y, int B, int H,
                                                            int W, int C, int beta, int ppt, float* red) {
  constexpr bool FLIP = false;
  constexpr int CH = TT<T>::CH;
  __shared__ float sred[2][8][32 * CH];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int CC = C / CH, W2 = W >> 1;
  const int cc = blockIdx.y * 32 + tx;
  const long P = (long)B * H * W2;
  float s1[CH], s2[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) s1[j] = s2[j] = 0.f;
  if (cc < CC) {
    for (int k = 0; k < ppt; ++k) {
      const long pix = ((long)blockIdx.x * ppt + k) * 8 + ty;
      if (pix >= P) break;
      const int ox = (int)(pix % W2) * 2;
      const int oy = (int)((pix / W2) % H);
      const int b = (int)(pix / ((long)W2 * H));
      const long orow = ((long)b * H + oy) * W + ox;
      uint4 oq0 = zero16(), oq1 = zero16();
      T* o = y + orow * C + cc * CH;
      if (beta) { oq0 = ld16(o); oq1 = ld16(o + C); }
      float a0[CH], a1[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) a0[j] = a1[j] = (bias && !FLIP) ? bias[cc * CH + j] : 0.f;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int iy = oy + kh - 1;
        if (iy < 0 || iy >= H) continue;
        const T* row = x + (((long)b * H + iy) * W) * C + cc * CH;
        float in[4][CH];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int ix = ox - 1 + t;
          if (ix >= 0 && ix < W) unpack<T>(ld16(row + (long)ix * C), in[t]);
          else {
#pragma unroll
            for (int j = 0; j < CH; ++j) in[t][j] = 0.f;
          }
        }
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          float wv[CH];
          const int tap = FLIP ? 8 - (kh * 3 + kw) : kh * 3 + kw;
          unpack<T>(ld16(wp + tap * C + cc * CH), wv);
#pragma unroll
          for (int j = 0; j < CH; ++j) { a0[j] += in[kw][j] * wv[j]; a1[j] += in[kw + 1][j] * wv[j]; }
        }
      }
      if (beta) {
        float o0[CH], o1[CH];
        unpack<T>(oq0, o0); unpack<T>(oq1, o1);
#pragma unroll
        for (int j = 0; j < CH; ++j) { a0[j] += o0[j]; a1[j] += o1[j]; }
      }
      st16(o, pack<T>(a0));
      st16(o + C, pack<T>(a1));
#pragma unroll
      for (int j = 0; j < CH; ++j) { s1[j] += a0[j] + a1[j]; s2[j] += a0[j] * a0[j] + a1[j] * a1[j]; }
    }
  }
#pragma unroll
  for (int j = 0; j < CH; ++j) { sred[0][ty][tx * CH + j] = s1[j]; sred[1][ty][tx * CH + j] = s2[j]; }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * 32 * CH; i += 256) {
    const int k = i / (32 * CH), c = i - k * 32 * CH;
    const int col = blockIdx.y * 32 * CH + c;
    if (col >= C) continue;
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) sum += sred[k][t][c];
    atomicAdd(red + (long)k * C + col, sum);
  }
}

// ---- BatchNorm(batch statistics) + activation + stride-1 depthwise 3x3 + the statistics of ITS output in ONE launch --------
// The late MBConv stages (8x24 and 4x12 maps): a workgroup owns one image x one slab of 64 channels.  It pulls the slab of
// the expand convolution's raw output (24 KB at 8x24) with full 128-byte lines, applies the BatchNorm coefficients it derives
// from the column sums + SiLU ONCE per element, writes the activated slab back (the depthwise weight gradient reads it on the
// side stream) and into an LDS tile with a zero halo, and computes its 3x3 outputs from LDS -- the convolution's nine taps never
// go to memory.  Replaces bn_act_kernel + dwconv_s1_red_kernel (one kernel boundary and one pass over the expanded activation
// less per block, 28 blocks per step); accumulation order per output = dwconv_s1_red_kernel's, so the results are bitwise the same.
// column sums s1/s2 (8 channels of this thread's chunk) -> red[0..C) / red[C..2C): lanes of a wave with the same chunk, the waves
// through LDS, then one atomic per channel and workgroup
DEVI void bdw_colsums(float* s1, float* s2, float (*sred)[2][BDW_SC * 8], float* red, int C, int tid, int NT) {
  constexpr int CH = 8, SC = BDW_SC;
#pragma unroll
  for (int o = SC; o < 64; o <<= 1) {
#pragma unroll
    for (int j = 0; j < CH; ++j) { s1[j] += __shfl_xor(s1[j], o, 64); s2[j] += __shfl_xor(s2[j], o, 64); }
  }
  const int lane = tid & 63, wave = tid >> 6;
  if (lane < SC) {
#pragma unroll
    for (int j = 0; j < CH; ++j) { sred[wave][0][lane * CH + j] = s1[j]; sred[wave][1][lane * CH + j] = s2[j]; }
  }
  __syncthreads();
  for (int i = tid; i < 2 * SC * CH; i += NT) {   // NT may be a single wave
    const int k = i / (SC * CH), c = i - k * SC * CH;
    float sum = 0.f;
    for (int wv = 0; wv < NT / 64; ++wv) sum += sred[wv][k][c];
    atomicAdd(red + (long)k * C + blockIdx.y * SC * CH + c, sum);
  }
}
__global__ __launch_bounds__(512, 4) void bn_dw_img_kernel(const bf16_t* __restrict__ y, const float* sums, int sums_rep, const float* w,
                                                           const float* b, float* rm, float* rv, int64_t* nbt, float eps, float mom,
                                                           float invM, float unbias, float* ss, float* mr, bf16_t* __restrict__ z,
                                                           const bf16_t* __restrict__ wp, const float* dwbias, bf16_t* __restrict__ out,
                                                           float* red, int H, int W, int C, int rowpix, int act) {
  typedef bf16_t T;
  constexpr int CH = 8, SC = BDW_SC, RUN = BDW_RUN;
  extern __shared__ __attribute__((aligned(16))) unsigned char bdw_sm[];
  __shared__ float sred[8][2][SC * CH];
  __shared__ __attribute__((aligned(16))) float cf[3][SC * CH];   // scale, shift, depthwise bias
  __shared__ uint4 wl[9][SC];
  uint4* tile = reinterpret_cast<uint4*>(bdw_sm);   // [(H + 2)][rowpix][SC] chunks; rowpix odd: two rows apart = half the banks apart
  const int tid = threadIdx.x, NT = blockDim.x, G = NT / SC;
  const int chunk = tid % SC, g = tid / SC;
  const int img = blockIdx.x, cb = blockIdx.y * SC * CH, c0 = cb + chunk * CH;
  const int HW = H * W;
  const long base = (long)img * HW * C + c0;
  // the slab is requested before anything else
  uint4 raw[RUN];
#pragma unroll
  for (int k = 0; k < RUN; ++k) raw[k] = ld16(y + base + (long)(g + k * G) * C);
  for (int i = tid; i < 9 * SC; i += NT) wl[i / SC][i % SC] = ld16(wp + (long)(i / SC) * C + cb + (i % SC) * CH);
  for (int c = tid; c < SC * CH; c += NT) {   // one thread per channel: BatchNorm finalize (bn_act_kernel's arithmetic)
    const int cg = cb + c;
    float mean = 0.f, var = 0.f;
    for (int rp = 0; rp < sums_rep; ++rp) { mean += sums[(size_t)rp * 2 * C + cg]; var += sums[(size_t)rp * 2 * C + C + cg]; }
    mean *= invM; var = fmaxf(var * invM - mean * mean, 0.f);
    const float rstd = rsqrtf(var + eps), sc = w[cg] * rstd, sh = b[cg] - mean * sc;
    cf[0][c] = sc; cf[1][c] = sh; cf[2][c] = dwbias ? dwbias[cg] : 0.f;
    if (img == 0) {   // publish for the backward, update the running statistics
      ss[cg] = sc; ss[C + cg] = sh; mr[cg] = mean; mr[C + cg] = rstd;
      rm[cg] = (1.f - mom) * rm[cg] + mom * mean;
      rv[cg] = (1.f - mom) * rv[cg] + mom * var * unbias;
    }
  }
  bdw_zero_halo(tile, H, W, rowpix, tid, NT);
  __syncthreads();
  {
    float sc[CH], sh[CH];
    lds8(cf[0] + chunk * CH, sc); lds8(cf[1] + chunk * CH, sh);
#pragma unroll
    for (int k = 0; k < RUN; ++k) {
      const int pix = g + k * G, py = pix / W, px = pix - py * W;
      float v[CH];
      unpack<T>(raw[k], v);
#pragma unroll
      for (int j = 0; j < CH; ++j) v[j] = act_fwd(v[j] * sc[j] + sh[j], act);
      const uint4 q = pack<T>(v);
      st16(z + base + (long)pix * C, q);
      tile[((py + 1) * rowpix + px + 1) * SC + chunk] = q;
    }
  }
  __syncthreads();
  const int row = g % H, ox0 = (g / H) * RUN;
  float acc[RUN][CH];
  {
    float bb[CH];
    lds8(cf[2] + chunk * CH, bb);
#pragma unroll
    for (int p = 0; p < RUN; ++p)
#pragma unroll
      for (int j = 0; j < CH; ++j) acc[p][j] = bb[j];
  }
  bdw_taps<false>(tile, wl, row, ox0, rowpix, chunk, acc);
  float s1[CH], s2[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) s1[j] = s2[j] = 0.f;
#pragma unroll
  for (int p = 0; p < RUN; ++p) {
    st16(out + base + (long)(row * W + ox0 + p) * C, pack<T>(acc[p]));
#pragma unroll
    for (int j = 0; j < CH; ++j) { s1[j] += acc[p][j]; s2[j] += acc[p][j] * acc[p][j]; }
  }
  bdw_colsums(s1, s2, sred, red, C, tid, NT);
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0 && nbt) *nbt += 1;
}
// false = shape / mode not taken (the caller launches launch_bn_act + launch_dwconv)
bool launch_bn_dwconv(int dt, const void* y, const float* sums, int sums_rep, const float* w, const float* b, float* rm, float* rv, int64_t* nbt,
                      float eps, float mom, float* ss, float* mr, void* z, const void* wp, const float* dwbias, void* out, float* red, int B, int H,
                      int W, int C, int act, hipStream_t s) {
  const bool off = sw_off("fused_bn_dw");   // read per call: tests compare the fused and the plain forms in one process
  if (off || g_det.on || dt != DT_BF16 || !sums || !red || (C % (8 * BDW_SC)) != 0 || (W % BDW_RUN) != 0) return false;
  const int HW = H * W, NT = (HW / BDW_RUN) * BDW_SC;
  if (NT > 512 || (NT % 64) != 0) return false;
  const int rowpix = (W + 2) | 1;
  const size_t lds = (size_t)(H + 2) * rowpix * BDW_SC * 16;
  if (lds > 60 * 1024) return false;
  const long M = (long)B * HW;
  const float unbias = M > 1 ? (float)((double)M / (double)(M - 1)) : 1.f;
  hipLaunchKernelGGL(bn_dw_img_kernel, dim3(B, C / (8 * BDW_SC)), dim3(NT), lds, s, (const bf16_t*)y, sums, sums_rep < 1 ? 1 : sums_rep, w, b, rm, rv, nbt,
                     eps, mom, 1.0f / (float)M, unbias, ss, mr, (bf16_t*)z, (const bf16_t*)wp, dwbias, (bf16_t*)out, red, H, W, C, rowpix, act);
  return true;
}

// Inference form of the same tile: the input is already activated (the expand GEMM's epilogue applied the eval-mode BatchNorm), so the
// slab goes to LDS as loaded; the epilogue applies the NEXT BatchNorm's eval scale/shift + activation and -- the workgroup holds the
// whole image for its 64 channels -- leaves the squeeze-and-excite pool sums complete, no atomics:  pool[img][c] = sum_pix out.
// Replaces dwconv_s1_kernel (eval epilogue) + the pooling half of se_fwd_kernel on the greedy-decode encoder.
// se.box != null: the squeeze-and-excite block behind it as well (se_exchange_gates: the image's workgroups hand their shares of the hidden
// layer to each other) -- out = act(...) * gate, one launch for depthwise + BatchNorm + SiLU + SE on the inference encoder.
struct SeEvalP { unsigned long long* box; unsigned tag; long long timeout_ticks; const bf16_t* W1; const float* b1; const bf16_t* W2; const float* b2; int S; };
__global__ __launch_bounds__(512, 4) void dw_eval_img_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wp, const float* dwbias,
                                                             const float* esc, const float* esh, bf16_t* __restrict__ out, float* pool, int H,
                                                             int W, int C, int rowpix, int act, SeEvalP se) {
  typedef bf16_t T;
  constexpr int CH = 8, SC = BDW_SC, RUN = BDW_RUN;
  extern __shared__ __attribute__((aligned(16))) unsigned char bdw_sm[];
  __shared__ float sred[8][SC * CH];
  __shared__ __attribute__((aligned(16))) float cf[3][SC * CH];   // scale, shift, depthwise bias
  __shared__ uint4 wl[9][SC];
  __shared__ __attribute__((aligned(16))) float se_ps[SC * CH];
  __shared__ float se_hq[8][64];
  __shared__ __attribute__((aligned(16))) float se_hs[64];
  __shared__ __attribute__((aligned(16))) float se_gl[SC * CH];
  uint4* tile = reinterpret_cast<uint4*>(bdw_sm);
  const int tid = threadIdx.x, NT = blockDim.x, G = NT / SC;
  const int chunk = tid % SC, g = tid / SC;
  const int img = blockIdx.x, cb = blockIdx.y * SC * CH, c0 = cb + chunk * CH;
  const int HW = H * W;
  const long base = (long)img * HW * C + c0;
  uint4 raw[RUN];
#pragma unroll
  for (int k = 0; k < RUN; ++k) raw[k] = ld16(x + base + (long)(g + k * G) * C);
  for (int i = tid; i < 9 * SC; i += NT) wl[i / SC][i % SC] = ld16(wp + (long)(i / SC) * C + cb + (i % SC) * CH);
  for (int c = tid; c < SC * CH; c += NT) {
    cf[0][c] = esc ? esc[cb + c] : 1.f; cf[1][c] = esh ? esh[cb + c] : 0.f; cf[2][c] = dwbias ? dwbias[cb + c] : 0.f;
  }
  bdw_zero_halo(tile, H, W, rowpix, tid, NT);
#pragma unroll
  for (int k = 0; k < RUN; ++k) {
    const int pix = g + k * G, py = pix / W, px = pix - py * W;
    tile[((py + 1) * rowpix + px + 1) * SC + chunk] = raw[k];
  }
  __syncthreads();
  const int row = g % H, ox0 = (g / H) * RUN;
  float acc[RUN][CH];
  {
    float bb[CH];
    lds8(cf[2] + chunk * CH, bb);
#pragma unroll
    for (int p = 0; p < RUN; ++p)
#pragma unroll
      for (int j = 0; j < CH; ++j) acc[p][j] = bb[j];
  }
  bdw_taps<false>(tile, wl, row, ox0, rowpix, chunk, acc);
  // squeeze-and-excite operands of threads 0..63 (expand-matrix row, 64 columns of reduce-matrix row tid), requested before the pool
  const long long t_end = se.box ? (long long)wall_clock64() + se.timeout_ticks : 0;
  uint4 w2r[8], w1r[8];
  float b2v = 0.f, b1v = 0.f;
  if (se.box && tid < SC * CH) {
#pragma unroll
    for (int u = 0; u < 8; ++u) w2r[u] = ld16(se.W2 + (long)(cb + tid) * se.S + (u * CH < se.S ? u * CH : 0));
    b2v = se.b2[cb + tid];
#pragma unroll
    for (int u = 0; u < 8; ++u) w1r[u] = ld16(se.W1 + (long)(tid < se.S ? tid : 0) * C + cb + u * CH);
    if (tid < se.S) b1v = se.b1[tid];
  }
  float sc[CH], sh[CH], s1[CH];
  lds8(cf[0] + chunk * CH, sc); lds8(cf[1] + chunk * CH, sh);
#pragma unroll
  for (int j = 0; j < CH; ++j) s1[j] = 0.f;
  uint4 zq[RUN];
#pragma unroll
  for (int p = 0; p < RUN; ++p) {
#pragma unroll
    for (int j = 0; j < CH; ++j) acc[p][j] = act_fwd(acc[p][j] * sc[j] + sh[j], esc ? act : 0);
    zq[p] = pack<T>(acc[p]);
    if (!se.box) st16(out + base + (long)(row * W + ox0 + p) * C, zq[p]);
    float r[CH];
    unpack<T>(zq[p], r);   // the pool sums what the consumer reads (rounded), as se_fwd_kernel did
#pragma unroll
    for (int j = 0; j < CH; ++j) s1[j] += r[j];
  }
  if (!pool && !se.box) return;
#pragma unroll
  for (int o = SC; o < 64; o <<= 1) {
#pragma unroll
    for (int j = 0; j < CH; ++j) s1[j] += __shfl_xor(s1[j], o, 64);
  }
  const int lane = tid & 63, wave = tid >> 6;
  if (lane < SC) {
#pragma unroll
    for (int j = 0; j < CH; ++j) sred[wave][lane * CH + j] = s1[j];
  }
  __syncthreads();
  if (tid < SC * CH) {
    float sum = 0.f;
    for (int wv = 0; wv < NT / 64; ++wv) sum += sred[wv][tid];
    if (pool) pool[(long)img * C + cb + tid] = sum;
    se_ps[tid] = sum * (1.0f / (float)HW);
  }
  if (!se.box) return;
  __syncthreads();
  {
    SeXchg xc;
    xc.NG = C / (SC * CH); xc.S = se.S; xc.tag = se.tag; xc.t_end = t_end; xc.err = &g_satrn_errflag;
    xc.ibox = (se_box_t*)se.box + (size_t)img * xc.NG * 64;
    float uu, sv;
    se_exchange_gates<4>(xc, se_ps, w1r, w2r, b1v, b2v, se_hq, se_hs, se_gl, uu, sv);
    float gv[CH];
    lds8(se_gl + chunk * CH, gv);
#pragma unroll
    for (int p = 0; p < RUN; ++p) {
      float v[CH];
      unpack<T>(zq[p], v);
#pragma unroll
      for (int j = 0; j < CH; ++j) v[j] *= gv[j];
      st16(out + base + (long)(row * W + ox0 + p) * C, pack<T>(v));
    }
  }
}
// false = shape not taken (the caller launches launch_dwconv with the eval epilogue; the pool stays with the SE kernel)
bool launch_dwconv_eval_img(int dt, const void* x, const void* wp, const float* dwbias, const float* esc, const float* esh, int act, void* out,
                            float* pool, int B, int H, int W, int C, hipStream_t s, const SeEvalArgs* se) {
  const bool off = sw_off("dw_eval_img");   // read per call: tests compare the two forms in one process
  if (off || dt != DT_BF16 || (C % (8 * BDW_SC)) != 0 || (W % BDW_RUN) != 0) return false;
  const int HW = H * W, NT = (HW / BDW_RUN) * BDW_SC;
  if (NT > 512 || (NT % 64) != 0) return false;
  const int rowpix = (W + 2) | 1;
  const size_t lds = (size_t)(H + 2) * rowpix * BDW_SC * 16;
  if (lds > 60 * 1024) return false;
  SeEvalP sp;
  sp.box = nullptr; sp.tag = 0; sp.timeout_ticks = 200000000LL; sp.W1 = nullptr; sp.b1 = nullptr; sp.W2 = nullptr; sp.b2 = nullptr; sp.S = 0;
  if (se) {
    // with the squeeze-and-excite block: the image's workgroups wait for each other, so the WHOLE grid must be resident at once
    if (sw_off("dw_eval_se") || !se->box || B > se->box_images || se->S > 64 || (se->S % 8) != 0 || C > 1536 || !esc) return false;
    if (!se_box_usable(s)) return false;
    if ((long)B * (C / (8 * BDW_SC)) > resident_capacity((const void*)dw_eval_img_kernel, NT, lds)) return false;
    sp.box = se->box; sp.tag = se_next_tag(); sp.W1 = (const bf16_t*)se->W1; sp.b1 = se->b1; sp.W2 = (const bf16_t*)se->W2; sp.b2 = se->b2; sp.S = se->S;
  }
  hipLaunchKernelGGL(dw_eval_img_kernel, dim3(B, C / (8 * BDW_SC)), dim3(NT), lds, s, (const bf16_t*)x, (const bf16_t*)wp, dwbias, esc, esh, (bf16_t*)out,
                     pool, H, W, C, rowpix, act, sp);
  return true;
}

// per-image column sums (the squeeze-and-excite pool) for shapes the image tile does not take: one workgroup per image x 16-byte chunk
// column block, 256 threads = 8 chunks x 32 pixel lanes
template <typename T>
__global__ __launch_bounds__(256) void image_pool_kernel(const T* __restrict__ x, float* pool, int HW, int C) {
  constexpr int CH = TT<T>::CH;
  __shared__ float sm[32][8 * CH + 1];
  const int tid = threadIdx.x, ck = tid & 7, pl = tid >> 3;
  const int c0 = (blockIdx.y * 8 + ck) * CH, img = blockIdx.x;
  float a[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) a[j] = 0.f;
  if (c0 < C)
    for (int p = pl; p < HW; p += 32) {
      float v[CH];
      unpack<T>(ld16(x + ((long)img * HW + p) * C + c0), v);
#pragma unroll
      for (int j = 0; j < CH; ++j) a[j] += v[j];
    }
#pragma unroll
  for (int j = 0; j < CH; ++j) sm[pl][ck * CH + j] = a[j];
  __syncthreads();
  if (tid < 8 * CH && blockIdx.y * 8 * CH + tid < C) {
    float sum = 0.f;
    for (int r = 0; r < 32; ++r) sum += sm[r][tid];
    pool[(long)img * C + blockIdx.y * 8 * CH + tid] = sum;
  }
}
void launch_image_pool(int dt, const void* x, float* pool, int B, int HW, int C, hipStream_t s) {
  DISPATCH_T(dt, {
    constexpr int CH = TT<T>::CH;
    hipLaunchKernelGGL((image_pool_kernel<T>), dim3(B, (C + 8 * CH - 1) / (8 * CH)), dim3(256), 0, s, (const T*)x, pool, HW, C);
  });
}

// The backward twin of bn_dw_img_kernel: data gradient of the stride-1 depthwise 3x3 from an LDS tile of dy (whole image x 64
// channels per workgroup) AND the BatchNorm-backward column sums of the tensor it differentiates (z = act(bn(y))):
//   dz = conv^T(dy);  red[0..C) += sum dz*act'(u),  red[C..2C) += sum dz*act'(u)*xhat      (u = y*scale+shift)
// -- what launch_dwconv(mode 1) + launch_bn_bwd_reduce did in two launches (the thread-per-pixel-pair form of this fusion needed
// 206 VGPRs and was slower than the pair; from LDS a thread holds 3 outputs x 8 channels).  dz is rounded to bf16 before it
// enters the sums, as the separate reduction read it back from memory.
// APPLY: the tile is not dy as stored but the BatchNorm backward-apply result computed here from that BatchNorm's output gradient
// (ap.dz, with the squeeze-and-excite fold), its raw input (ap.y) and its column sums -- bn_bwd_apply_kernel's arithmetic, rounded to
// bf16 and written to ap.dy for the weight-gradient pass exactly as that kernel would have left it.
struct BnApplyP {
  const bf16_t* dz; const bf16_t* y; const float* ss; const float* mr; const float* w; const float* red; bf16_t* dy; float* dwp; float* dbp;
  const bf16_t* se_gate; const bf16_t* se_dpool; float se_scale; float invM; int act;
};
// TAIL: the backward-apply pass of the BatchNorm in FRONT as well (the tensor this kernel differentiates is z = act(bn(y))): its column sums
// are over the whole batch, so the B workgroups of a slab exchange their shares through tagged granules (tl.box[slab][img][128], the
// pattern of mbconv_front_kernel's statistics exchanges: everyone publishes its own share, everyone adds all B in image order -- the same
// bits in every workgroup), then  dyf = A*g + Bc + Cc*y  goes to tl.dy from the registers and dz is never written.  Replaces
// bn_bwd_apply_kernel behind this launch (one kernel boundary + one write and one read of the expanded gradient).  A slab's
// workgroups are consecutive in dispatch order (blockIdx.x = image) and wait for nobody else.
struct BnTailP {
  bf16_t* dy; const float* w; float* dwp; float* dbp; se_box_t* box; unsigned* err; float invM; unsigned tag; long long timeout_ticks;
};
template <bool APPLY, bool TAIL>
__global__ __launch_bounds__(512, 4) void dw_bwd_img_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ wp, bf16_t* __restrict__ dz,
                                                            const bf16_t* __restrict__ y, const float* ss, const float* mr, float* red, int H, int W,
                                                            int C, int rowpix, int act, int beta, BnApplyP ap, BnTailP tl) {
  typedef bf16_t T;
  constexpr int CH = 8, SC = BDW_SC, RUN = BDW_RUN;
  extern __shared__ __attribute__((aligned(16))) unsigned char bdw_sm[];
  __shared__ float sred[8][2][SC * CH];
  // per-channel coefficients of the slab, derived once: 0..3 = scale, shift, mean, rstd of the BatchNorm in FRONT (its sums are
  // reduced here); APPLY: 4..8 = scale, shift, A, Bc, Cc of the BatchNorm BEHIND, 9..10 = this image's SE gate and dpooled*scale
  __shared__ __attribute__((aligned(16))) float cf[APPLY ? 11 : 4][SC * CH];
  __shared__ uint4 wl[9][SC];
  uint4* tile = reinterpret_cast<uint4*>(bdw_sm);
  const int tid = threadIdx.x, NT = blockDim.x, G = NT / SC;
  const int chunk = tid % SC, g = tid / SC;
  const int img = blockIdx.x, cb = blockIdx.y * SC * CH, c0 = cb + chunk * CH;
  const int HW = H * W;
  const long base = (long)img * HW * C + c0;
  const int row = g % H, ox0 = (g / H) * RUN;
  uint4 raw[RUN], ay[RUN];
#pragma unroll
  for (int k = 0; k < RUN; ++k) {
    raw[k] = ld16((APPLY ? ap.dz : dy) + base + (long)(g + k * G) * C);
    ay[k] = APPLY ? ld16(ap.y + base + (long)(g + k * G) * C) : zero16();
  }
  for (int i = tid; i < 9 * SC; i += NT) wl[i / SC][i % SC] = ld16(wp + (long)(i / SC) * C + cb + (i % SC) * CH);
  for (int c = tid; c < SC * CH; c += NT) {   // one thread per channel
    const int cg = cb + c;
    if (y) { cf[0][c] = ss[cg]; cf[1][c] = ss[C + cg]; cf[2][c] = mr[cg]; cf[3][c] = mr[C + cg]; }
    if (APPLY) {
      // dy = A*g + Bc + Cc*y, g = dz*act'(y*scale+shift)   (bn_bwd_apply_kernel)
      const float amu = ap.mr[cg], ars = ap.mr[C + cg], r0 = ap.red[cg], r1 = ap.red[C + cg];
      const float a = ap.w[cg] * ars, m1 = r0 * ap.invM, m2 = r1 * ap.invM;
      cf[4][c] = ap.ss[cg]; cf[5][c] = ap.ss[C + cg];
      cf[6][c] = a; cf[7][c] = -a * m1 + a * ars * amu * m2; cf[8][c] = -a * ars * m2;
      cf[9][c] = ap.se_gate ? to_f(ap.se_gate[(long)img * C + cg]) : 1.f;
      cf[10][c] = ap.se_gate ? to_f(ap.se_dpool[(long)img * C + cg]) * ap.se_scale : 0.f;
      if (img == 0 && ap.dwp) { ap.dwp[cg] += r1; ap.dbp[cg] += r0; }   // parameter gradients (zeroed per step: accumulate)
    }
  }
  bdw_zero_halo(tile, H, W, rowpix, tid, NT);
  if (APPLY) {
    __syncthreads();
    float asc[CH], ash[CH], A[CH], Bc[CH], Cc[CH], gt[CH], dp[CH];
    lds8(cf[4] + chunk * CH, asc); lds8(cf[5] + chunk * CH, ash); lds8(cf[6] + chunk * CH, A); lds8(cf[7] + chunk * CH, Bc);
    lds8(cf[8] + chunk * CH, Cc); lds8(cf[9] + chunk * CH, gt); lds8(cf[10] + chunk * CH, dp);
#pragma unroll
    for (int k = 0; k < RUN; ++k) {
      float d[CH], v[CH];
      unpack<T>(raw[k], d);
      unpack<T>(ay[k], v);
      if (ap.se_gate) {
#pragma unroll
        for (int j = 0; j < CH; ++j) d[j] = d[j] * gt[j] + dp[j];
      }
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const float gg = d[j] * act_bwd(v[j] * asc[j] + ash[j], ap.act);
        d[j] = A[j] * gg + Bc[j] + Cc[j] * v[j];
      }
      raw[k] = pack<T>(d);
      st16(ap.dy + base + (long)(g + k * G) * C, raw[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < RUN; ++k) {
    const int pix = g + k * G, py = pix / W, px = pix - py * W;
    tile[((py + 1) * rowpix + px + 1) * SC + chunk] = raw[k];
  }
  __syncthreads();
  // operands of the epilogue requested now: they arrive while the taps are computed
  uint4 yq[RUN], oq[RUN];
#pragma unroll
  for (int p = 0; p < RUN; ++p) {
    yq[p] = y ? ld16(y + base + (long)(row * W + ox0 + p) * C) : zero16();
    oq[p] = (!TAIL && beta) ? ld16(dz + base + (long)(row * W + ox0 + p) * C) : zero16();   // (TAIL: beta == 0, launcher)
  }
  float acc[RUN][CH];
#pragma unroll
  for (int p = 0; p < RUN; ++p)
#pragma unroll
    for (int j = 0; j < CH; ++j) acc[p][j] = 0.f;
  bdw_taps<true>(tile, wl, row, ox0, rowpix, chunk, acc);   // transposed convolution: taps mirrored
  float s1[CH], s2[CH], sc[CH], sh[CH], mu[CH], rs[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) s1[j] = s2[j] = sc[j] = sh[j] = mu[j] = rs[j] = 0.f;
  if (y) { lds8(cf[0] + chunk * CH, sc); lds8(cf[1] + chunk * CH, sh); lds8(cf[2] + chunk * CH, mu); lds8(cf[3] + chunk * CH, rs); }
  uint4 dq[RUN];   // TAIL: dz as it would have been stored (bf16), kept for the apply pass
#pragma unroll
  for (int p = 0; p < RUN; ++p) {
    if (!TAIL && beta) {
      float o[CH];
      unpack<T>(oq[p], o);
#pragma unroll
      for (int j = 0; j < CH; ++j) acc[p][j] += o[j];
    }
    const uint4 q = pack<T>(acc[p]);
    if (TAIL) dq[p] = q;
    else st16(dz + base + (long)(row * W + ox0 + p) * C, q);
    float d[CH], v[CH];
    unpack<T>(q, d);
    unpack<T>(yq[p], v);
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const float u = v[j] * sc[j] + sh[j];
      const float gg = d[j] * act_bwd(u, act);
      s1[j] += gg; s2[j] += gg * ((v[j] - mu[j]) * rs[j]);
    }
  }
  if (!y) return;   // plain data gradient (uniform: no barrier is skipped by part of the workgroup)
  if (!TAIL) { bdw_colsums(s1, s2, sred, red, C, tid, NT); return; }
  // ---- this image's share of the 2 x 64 column sums -> mailbox; all B shares of the slab <- mailbox --------------------------------------
  const long long t_end = (long long)wall_clock64() + tl.timeout_ticks;
#pragma unroll
  for (int o = SC; o < 64; o <<= 1) {
#pragma unroll
    for (int j = 0; j < CH; ++j) { s1[j] += __shfl_xor(s1[j], o, 64); s2[j] += __shfl_xor(s2[j], o, 64); }
  }
  {
    const int lane = tid & 63, wave = tid >> 6;
    if (lane < SC) {
#pragma unroll
      for (int j = 0; j < CH; ++j) { sred[wave][0][lane * CH + j] = s1[j]; sred[wave][1][lane * CH + j] = s2[j]; }
    }
  }
  __syncthreads();
  const int B = gridDim.x;
  se_box_t* sbox = tl.box + (size_t)blockIdx.y * B * 128;
  if (tid < 128) {
    const int k = tid >> 6, c = tid & 63;
    float sum = 0.f;
    for (int wv = 0; wv < NT / 64; ++wv) sum += sred[wv][k][c];
    se_box_put(sbox + (size_t)img * 128 + tid, tl.tag, sum);
  }
  __syncthreads();   // sred is reused for the gathered totals
  {
    // thread = (column v of 128, image group grp of NT / 128): every workgroup adds the same shares in the same order
    const int v = tid & 127, grp = tid >> 7, NG = NT >> 7;
    constexpr int GB = 8;
    float a = 0.f;
    for (int i0 = grp; i0 < B; i0 += NG * GB) {
      float vals[GB];
      const int n = min(GB, (B - i0 + NG - 1) / NG);
      se_box_gather<GB>(sbox + (size_t)i0 * 128 + v, (size_t)NG * 128, n, tl.tag, t_end, vals, tl.err);
#pragma unroll
      for (int k = 0; k < GB; ++k) if (k < n) a += vals[k];
    }
    sred[grp][v >> 6][v & 63] = a;
  }
  __syncthreads();
  if (tid < SC * CH) {
    float r0 = 0.f, r1 = 0.f;
    for (int gq = 0; gq < (NT >> 7); ++gq) { r0 += sred[gq][0][tid]; r1 += sred[gq][1][tid]; }
    const int cg = cb + tid;
    const float amu = cf[2][tid], ars = cf[3][tid];
    const float a = tl.w[cg] * ars, m1 = r0 * tl.invM, m2 = r1 * tl.invM;
    if (img == 0 && tl.dwp) { tl.dwp[cg] += r1; tl.dbp[cg] += r0; }   // parameter gradients (zeroed per step: accumulate)
    cf[2][tid] = a; cf[3][tid] = -a * m1 + a * ars * amu * m2;
    sred[4][0][tid] = -a * ars * m2;
  }
  __syncthreads();
  {
    float A[CH], Bc[CH], Cc[CH];
    lds8(cf[0] + chunk * CH, sc); lds8(cf[1] + chunk * CH, sh);   // (again: not held in registers across the exchange)
    lds8(cf[2] + chunk * CH, A); lds8(cf[3] + chunk * CH, Bc); lds8(sred[4][0] + chunk * CH, Cc);
#pragma unroll
    for (int p = 0; p < RUN; ++p) {
      float d[CH], v[CH];
      unpack<T>(dq[p], d);
      unpack<T>(yq[p], v);
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const float gg = d[j] * act_bwd(v[j] * sc[j] + sh[j], act);
        d[j] = A[j] * gg + Bc[j] + Cc[j] * v[j];
      }
      st16(tl.dy + base + (long)(row * W + ox0 + p) * C, pack<T>(d));
    }
  }
}
// false = shape / mode not taken (the caller launches launch_dwconv(mode 1) and leaves the sums to launch_bn_bwd_reduce)
bool dwconv_img_ok(int dt, int H, int W, int C) {
  if (g_det.on || dt != DT_BF16 || (C % (8 * BDW_SC)) != 0 || (W % BDW_RUN) != 0) return false;
  const int NT = (H * W / BDW_RUN) * BDW_SC;
  return NT <= 512 && (NT % 64) == 0 && (size_t)(H + 2) * ((W + 2) | 1) * BDW_SC * 16 <= 60 * 1024;
}
bool launch_dwconv_bwd_bn(int dt, const void* dy, const void* wp, void* dz, int beta, const void* y, const float* ss, const float* mr, int act,
                          float* red, int B, int H, int W, int C, hipStream_t s, const BnBwdHold* hold, const BnBwdTail* tail) {
  const bool off = sw_off("fused_dw_bwd");   // read per call (tests)
  if (off || !dwconv_img_ok(dt, H, W, C) || (y && !red && !tail) || (!y && !hold)) return false;
  if (tail && (!hold || !y || beta || !tail->dy || !tail->w || sw_off("dw_bwd_tail"))) return false;
  const int HW = H * W, NT = (HW / BDW_RUN) * BDW_SC;
  const int rowpix = (W + 2) | 1;
  const size_t lds = (size_t)(H + 2) * rowpix * BDW_SC * 16;
  BnApplyP ap = {};
  if (hold) {
    if (sw_off("bn_apply_dw") || hold->C != C || hold->M != (long)B * HW) return false;
    ap.dz = (const bf16_t*)hold->dz; ap.y = (const bf16_t*)hold->y; ap.ss = hold->ss; ap.mr = hold->mr; ap.w = hold->w; ap.red = hold->red;
    ap.dy = (bf16_t*)hold->dy; ap.dwp = hold->dwp; ap.dbp = hold->dbp; ap.se_gate = (const bf16_t*)hold->se_gate; ap.se_dpool = (const bf16_t*)hold->se_dpool;
    ap.se_scale = hold->se_hw > 0 ? 1.0f / (float)hold->se_hw : 0.f; ap.invM = 1.0f / (float)hold->M; ap.act = hold->act;
    if (hold->se_gate && hold->se_hw != HW) return false;
    BnTailP tl = {};
    if (tail) {
      // workgroups wait for each other: the whole grid has to be resident, the mailbox has to hold B x slabs x 128 granules
      const long nwg = (long)B * (C / (8 * BDW_SC));
      if ((NT % 128) != 0) return false;   // 128 publishers, image groups of 128 threads in the gather
      if (!g_mbbox.box || B > g_mbbox.images || (size_t)nwg * 128 > g_mbbox.words) return false;
      if (nwg > resident_capacity((const void*)dw_bwd_img_kernel<true, true>, NT, lds)) return false;
      tl.dy = (bf16_t*)tail->dy; tl.w = tail->w; tl.dwp = tail->dwp; tl.dbp = tail->dbp; tl.box = (se_box_t*)g_mbbox.box;
      tl.err = device_error_word(); tl.invM = 1.0f / (float)((long)B * HW); tl.tag = se_next_tag(); tl.timeout_ticks = 200000000LL;   // 2 s at 100 MHz
      if (!tl.err) return false;
      hipLaunchKernelGGL((dw_bwd_img_kernel<true, true>), dim3(B, C / (8 * BDW_SC)), dim3(NT), lds, s, (const bf16_t*)dy, (const bf16_t*)wp, (bf16_t*)dz,
                         (const bf16_t*)y, ss, mr, red, H, W, C, rowpix, act, beta, ap, tl);
      return true;
    }
    hipLaunchKernelGGL((dw_bwd_img_kernel<true, false>), dim3(B, C / (8 * BDW_SC)), dim3(NT), lds, s, (const bf16_t*)dy, (const bf16_t*)wp, (bf16_t*)dz,
                       (const bf16_t*)y, ss, mr, red, H, W, C, rowpix, act, beta, ap, tl);
  } else {
    hipLaunchKernelGGL((dw_bwd_img_kernel<false, false>), dim3(B, C / (8 * BDW_SC)), dim3(NT), lds, s, (const bf16_t*)dy, (const bf16_t*)wp, (bf16_t*)dz,
                       (const bf16_t*)y, ss, mr, red, H, W, C, rowpix, act, beta, ap, BnTailP{});
  }
  return true;
}

static bool dwconv_fuses_stats(int H, int W, int OH, int OW, int stride, int pt, int pl) {
  const bool off = sw_off("dw_fused_red");   // read per call (tests)
  return !off && !g_det.on && stride == 1 && pt == 1 && pl == 1 && OH == H && OW == W && (W & 1) == 0;
}

void launch_dwconv(int dt, int mode, const void* x, const void* wp, const float* bias, void* y, int B, int H, int W,
                   int C, int OH, int OW, int stride, int pt, int pl, int beta, float* stats, hipStream_t s,
                   const float* esc, const float* esh, int eact) {
  constexpr bool no_s1 = false;
  if (!no_s1 && stride == 1 && pt == 1 && pl == 1 && OH == H && OW == W && (W & 1) == 0) {
    if (mode == 0 && stats && !esc && dwconv_fuses_stats(H, W, OH, OW, stride, pt, pl)) {
      DISPATCH_T(dt, {
        const int CC = C / TT<T>::CH, gy = (CC + 31) / 32;
        const long P = (long)B * H * (W / 2);
        // pixel pairs per thread: as many (<= 8) as still leave ~512 workgroups
        long ppt = (P * gy) / (8L * 512);
        ppt = ppt >= 8 ? 8 : (ppt >= 4 ? 4 : (ppt >= 2 ? 2 : 1));
        const int gx = (int)((P + 8 * ppt - 1) / (8 * ppt));
        hipLaunchKernelGGL((dwconv_s1_red_kernel<T>), dim3(gx, gy), dim3(256), 0, s, (const T*)x, (const T*)wp, bias, (T*)y, B, H, W, C, beta, (int)ppt,
                           stats);
      });
      return;
    }
    DISPATCH_T(dt, {
      long n2 = (long)B * H * (W / 2) * (C / TT<T>::CH);
      if (mode == 0)
        hipLaunchKernelGGL((dwconv_s1_kernel<T, false>), dim3(grid_for(n2)), dim3(256), 0, s, (const T*)x, (const T*)wp, bias,
                           (T*)y, B, H, W, C, beta, esc, esh, eact);
      else
        hipLaunchKernelGGL((dwconv_s1_kernel<T, true>), dim3(grid_for(n2)), dim3(256), 0, s, (const T*)x, (const T*)wp, bias,
                           (T*)y, B, H, W, C, beta, esc, esh, eact);
      if (stats && mode == 0) launch_colstats(dt, y, (long)B * OH * OW, C, stats, s);
    });
    return;
  }
  DISPATCH_T(dt, {
    long n = (long)B * OH * OW * (C / TT<T>::CH);
    if (mode == 0)
      hipLaunchKernelGGL((dwconv_kernel<T, 0>), dim3(grid_for(n)), dim3(256), 0, s, (const T*)x, (const T*)wp, bias,
                         (T*)y, B, H, W, C, OH, OW, stride, pt, pl, beta, esc, esh, eact);
    else
      hipLaunchKernelGGL((dwconv_kernel<T, 1>), dim3(grid_for(n)), dim3(256), 0, s, (const T*)x, (const T*)wp, bias,
                         (T*)y, B, H, W, C, OH, OW, stride, pt, pl, beta, esc, esh, eact);
    if (stats && mode == 0) launch_colstats(dt, y, (long)B * OH * OW, C, stats, s);
  });
}

// side-stream kernel (see launch_wgrad_tile): a small grid keeps it out of the data-gradient chain's way
static constexpr int DWW_BLK = 512, DWW_RPT = 8;   // target grid / rows per thread of the depthwise weight gradient (tools/elem_bench.cpp sweeps)
template <typename T> struct DwWgradF {
  const T* x; const T* dy; int H, W, C, OH, OW, stride, pt, pl;
  float rOW, rOHW; int small;   // reciprocals for the row decomposition while rows are exact in a float (small != 0)
  __device__ void prep(int) {}
  __device__ void operator()(long r, int c0, float (*acc)[TT<T>::CH]) const {
    constexpr int CH = TT<T>::CH;
    int ox, oy, b;
    if (small) {
      // three 64-bit divisions per row were several hundred instructions beside ten 16-byte loads and 80 multiply-adds
      const int ri = (int)r, ohw = OW * OH;
      b = (int)((float)ri * rOHW);
      { const int rr = ri - b * ohw; b += rr >= ohw ? 1 : (rr < 0 ? -1 : 0); }
      const int rem = ri - b * ohw;
      oy = (int)((float)rem * rOW);
      { const int rr = rem - oy * OW; oy += rr >= OW ? 1 : (rr < 0 ? -1 : 0); }
      ox = rem - oy * OW;
    } else {
      ox = (int)(r % OW);
      oy = (int)((r / OW) % OH);
      b = (int)(r / ((long)OW * OH));
    }
    float d[CH];
    unpack<T>(ld16(dy + r * C + c0), d);
#pragma unroll
    for (int j = 0; j < CH; ++j) acc[9][j] += d[j];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        int sy = oy * stride - pt + kh, sx = ox * stride - pl + kw;
        if (sy >= 0 && sy < H && sx >= 0 && sx < W) {
          float v[CH];
          unpack<T>(ld16(x + (((long)b * H + sy) * W + sx) * C + c0), v);
#pragma unroll
          for (int j = 0; j < CH; ++j) acc[kh * 3 + kw][j] += d[j] * v[j];
        }
      }
  }
};

// ---- max-pool 2x2 s2 -----------------------------------------------------------------------------
template <typename T, int BWD>
__global__ void maxpool_kernel(const T* x, const T* dy, T* out, int B, int H, int W, int C) {
  constexpr int CH = TT<T>::CH;
  const int CC = C / CH, OH = H / 2, OW = W / 2;
  long total = (long)B * OH * OW * CC;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int cc = (int)(i % CC);
    long pix = i / CC;
    int ox = (int)(pix % OW);
    int oy = (int)((pix / OW) % OH);
    int b = (int)(pix / ((long)OW * OH));
    float v[4][CH];
#pragma unroll
    for (int t = 0; t < 4; ++t)
      unpack<T>(ld16(x + (((long)b * H + oy * 2 + (t >> 1)) * W + ox * 2 + (t & 1)) * C + cc * CH), v[t]);
    if (!BWD) {
      float m[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) m[j] = fmaxf(fmaxf(v[0][j], v[1][j]), fmaxf(v[2][j], v[3][j]));
      st16(out + pix * C + cc * CH, pack<T>(m));
    } else {
      float d[CH], g[4][CH];
      unpack<T>(ld16(dy + pix * C + cc * CH), d);
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        int best = 0;
        float m = v[0][j];
#pragma unroll
        for (int t = 1; t < 4; ++t) if (v[t][j] > m) { m = v[t][j]; best = t; }
#pragma unroll
        for (int t = 0; t < 4; ++t) g[t][j] = (t == best) ? d[j] : 0.f;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
        st16(out + (((long)b * H + oy * 2 + (t >> 1)) * W + ox * 2 + (t & 1)) * C + cc * CH, pack<T>(g[t]));
    }
  }
}
void launch_maxpool(int dt, int bwd, const void* x, const void* dy, void* out, int B, int H, int W, int C,
                    hipStream_t s) {
  DISPATCH_T(dt, {
    long n = (long)B * (H / 2) * (W / 2) * (C / TT<T>::CH);
    if (bwd)
      hipLaunchKernelGGL((maxpool_kernel<T, 1>), dim3(grid_for(n)), dim3(256), 0, s, (const T*)x, (const T*)dy, (T*)out,
                         B, H, W, C);
    else
      hipLaunchKernelGGL((maxpool_kernel<T, 0>), dim3(grid_for(n)), dim3(256), 0, s, (const T*)x, (const T*)dy, (T*)out,
                         B, H, W, C);
  });
}

// ---- per-image reductions over HW: out[b][c (+C)] = sum_hw f(...)  ---------------------------------
// MODE 0: mean x ; 1: sum a*b ; 2: posenc dgate (sum dout*hpos[h], sum dout*wpos[w] -> out[b][2C])
// RG row groups of 64 chunk lanes each (block = 64 * RG threads): a thread sweeps HW / RG rows, so the number of
// dependent load batches -- what these latency-bound reductions cost -- falls with RG
template <typename T, int MODE, int RG>
__global__ __launch_bounds__(64 * RG) void hw_reduce_kernel(const T* a, const T* bb, const float* hpos, const float* wpos,
                                                            T* out, int HW, int Wd, int C, float scale) {
  constexpr int CH = TT<T>::CH;
  constexpr int NRED = MODE == 2 ? 2 : 1;
  __shared__ float red[NRED][RG][64 * CH];
  const int CC = C / CH;
  const int b = blockIdx.x;
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + cl;
  float s0[CH], s1[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) s0[j] = s1[j] = 0.f;
  if (c < CC) {
    // rows in batches of four whose loads are all in flight together; the last, partial batch re-reads a valid row with
    // weight 0 instead of branching (a partially unrolled loop would run its remainder one dependent round trip at a time:
    // HW = 48 with RG = 16 is three rows per thread, i.e. remainder only)
    for (int p0 = rg; p0 < HW; p0 += 4 * RG) {
      uint4 ra[4], rb[4];
      float m[4];
      int pp[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int p = p0 + u * RG;
        m[u] = p < HW ? 1.f : 0.f;
        pp[u] = p < HW ? p : p0;
        ra[u] = ld16(a + ((long)b * HW + pp[u]) * C + c * CH);
        if (MODE == 1) rb[u] = ld16(bb + ((long)b * HW + pp[u]) * C + c * CH);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float v[CH];
        unpack<T>(ra[u], v);
        if (MODE == 0) {
#pragma unroll
          for (int j = 0; j < CH; ++j) s0[j] += m[u] * v[j];
        } else if (MODE == 1) {
          float w[CH];
          unpack<T>(rb[u], w);
#pragma unroll
          for (int j = 0; j < CH; ++j) s0[j] += m[u] * v[j] * w[j];
        } else {
          int h = pp[u] / Wd, w_ = pp[u] - h * Wd;
#pragma unroll
          for (int j = 0; j < CH; ++j) {
            const float mv = m[u] * v[j];
            s0[j] += mv * hpos[h * C + c * CH + j];
            s1[j] += mv * wpos[w_ * C + c * CH + j];
          }
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < CH; ++j) { red[0][rg][cl * CH + j] = s0[j]; if (MODE == 2) red[NRED - 1][rg][cl * CH + j] = s1[j]; }
  __syncthreads();
  if (rg == 0 && c < CC) {
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      float t0 = 0.f, t1 = 0.f;
#pragma unroll
      for (int g = 0; g < RG; ++g) { t0 += red[0][g][cl * CH + j]; if (MODE == 2) t1 += red[NRED - 1][g][cl * CH + j]; }
      s0[j] = t0 * scale;
      s1[j] = t1 * scale;
    }
    if (MODE == 2) {
      st16(out + (long)b * 2 * C + c * CH, pack<T>(s0));
      st16(out + (long)b * 2 * C + C + c * CH, pack<T>(s1));
    } else {
      st16(out + (long)b * C + c * CH, pack<T>(s0));
    }
  }
}
void launch_pool_hw(int dt, const void* x, void* out, int B, int HW, int C, hipStream_t s) {
  DISPATCH_T(dt, {
    int gy = (C / TT<T>::CH + 63) / 64;
    hipLaunchKernelGGL((hw_reduce_kernel<T, 0, 16>), dim3(B, gy), dim3(1024), 0, s, (const T*)x, (const T*)nullptr, nullptr,
                       nullptr, (T*)out, HW, 1, C, 1.0f / (float)HW);
  });
}
void launch_se_bwd_gate(int dt, const void* dout, const void* x, void* dgate, int B, int HW, int C, hipStream_t s) {
  DISPATCH_T(dt, {
    int gy = (C / TT<T>::CH + 63) / 64;
    hipLaunchKernelGGL((hw_reduce_kernel<T, 1, 16>), dim3(B, gy), dim3(1024), 0, s, (const T*)dout, (const T*)x, nullptr,
                       nullptr, (T*)dgate, HW, 1, C, 1.0f);
  });
}
void launch_posenc2d_bwd(int dt, const void* dout, const float* hpos, const float* wpos, void* dgate, int B, int H,
                         int W, int C, hipStream_t s) {
  DISPATCH_T(dt, {
    int gy = (C / TT<T>::CH + 63) / 64;
    hipLaunchKernelGGL((hw_reduce_kernel<T, 2, 4>), dim3(B, gy), dim3(256), 0, s, (const T*)dout, (const T*)nullptr, hpos,
                       wpos, (T*)dgate, H * W, W, C, 1.0f);
  });
}

// ---- per-image broadcast elementwise ops ------------------------------------------------------------
// MODE 0: out = x*gate[b][c]          (SE scale)
// MODE 1: dx (+)= dout*gate[b][c] + dpool[b][c]*scale   (SE backward wrt x)
// MODE 2: dx += dpool[b][c]*scale     (mean-pool backward)
// MODE 3: out = x + g[b][c]*hpos[h][c] + g[b][C+c]*wpos[w][c]   (adaptive 2D positional encoding)
template <typename T, int MODE>
__global__ void bcast_kernel(const T* x, const T* gate, const T* dpool, const float* hpos, const float* wpos, T* out,
                             long nchunks, int HW, int Wd, int C, float scale, int beta) {
  constexpr int CH = TT<T>::CH;
  const int CC = C / CH;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nchunks; i += (long)gridDim.x * blockDim.x) {
    int c0 = (int)(i % CC) * CH;
    long pix = i / CC;
    int b = (int)(pix / HW);
    float v[CH], g[CH], r[CH];
    if (MODE != 2) unpack<T>(ld16(x + i * CH), v);
    if (MODE == 0) {
      unpack<T>(ld16(gate + (long)b * C + c0), g);
#pragma unroll
      for (int j = 0; j < CH; ++j) r[j] = v[j] * g[j];
    } else if (MODE == 1) {
      unpack<T>(ld16(gate + (long)b * C + c0), g);
#pragma unroll
      for (int j = 0; j < CH; ++j) r[j] = v[j] * g[j];
      if (dpool) {
        unpack<T>(ld16(dpool + (long)b * C + c0), g);
#pragma unroll
        for (int j = 0; j < CH; ++j) r[j] += g[j] * scale;
      }
      if (beta) {
        unpack<T>(ld16(out + i * CH), g);
#pragma unroll
        for (int j = 0; j < CH; ++j) r[j] += g[j];
      }
    } else if (MODE == 2) {
      unpack<T>(ld16(dpool + (long)b * C + c0), g);
      unpack<T>(ld16(out + i * CH), v);
#pragma unroll
      for (int j = 0; j < CH; ++j) r[j] = v[j] + g[j] * scale;
    } else {
      int p = (int)(pix % HW);
      int h = p / Wd, w_ = p - h * Wd;
      float g1[CH];
      unpack<T>(ld16(gate + (long)b * 2 * C + c0), g);
      unpack<T>(ld16(gate + (long)b * 2 * C + C + c0), g1);
#pragma unroll
      for (int j = 0; j < CH; ++j) r[j] = v[j] + g[j] * hpos[h * C + c0 + j] + g1[j] * wpos[w_ * C + c0 + j];
    }
    st16(out + i * CH, pack<T>(r));
  }
}
void launch_se_scale(int dt, const void* x, const void* gate, void* out, int B, int HW, int C, hipStream_t s) {
  DISPATCH_T(dt, {
    long n = (long)B * HW * C / TT<T>::CH;
    hipLaunchKernelGGL((bcast_kernel<T, 0>), dim3(grid_for(n)), dim3(256), 0, s, (const T*)x, (const T*)gate,
                       (const T*)nullptr, nullptr, nullptr, (T*)out, n, HW, 1, C, 1.f, 0);
  });
}
void launch_se_bwd_x(int dt, const void* dout, const void* gate, const void* dpool, void* dx, int B, int HW, int C,
                     int beta, hipStream_t s) {
  DISPATCH_T(dt, {
    long n = (long)B * HW * C / TT<T>::CH;
    hipLaunchKernelGGL((bcast_kernel<T, 1>), dim3(grid_for(n)), dim3(256), 0, s, (const T*)dout, (const T*)gate,
                       (const T*)dpool, nullptr, nullptr, (T*)dx, n, HW, 1, C, 1.0f / (float)HW, beta);
  });
}
void launch_bcast_add_hw(int dt, const void* dpool, void* dx, int B, int HW, int C, float scale, hipStream_t s) {
  DISPATCH_T(dt, {
    long n = (long)B * HW * C / TT<T>::CH;
    hipLaunchKernelGGL((bcast_kernel<T, 2>), dim3(grid_for(n)), dim3(256), 0, s, (const T*)nullptr, (const T*)nullptr,
                       (const T*)dpool, nullptr, nullptr, (T*)dx, n, HW, 1, C, scale, 1);
  });
}
void launch_posenc2d(int dt, const void* x, const void* gate, const float* hpos, const float* wpos, void* out, int B,
                     int H, int W, int C, hipStream_t s) {
  DISPATCH_T(dt, {
    long n = (long)B * H * W * C / TT<T>::CH;
    hipLaunchKernelGGL((bcast_kernel<T, 3>), dim3(grid_for(n)), dim3(256), 0, s, (const T*)x, (const T*)gate,
                       (const T*)nullptr, hpos, wpos, (T*)out, n, H * W, W, C, 1.f, 0);
  });
}

// ---- depthwise wgrad: ten column sums per channel (nine taps + bias) accumulated k-major into a zeroed scratch
// [10][C] with contiguous float atomics, then scattered into the torch layout dw[c*9+t] / dbias[c]
__global__ void dw_wgrad_scatter_kernel(const float* tmp, float* dw, float* dbias, int C) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C * 10) return;
  int t = i / C, c = i - t * C;
  if (t < 9) dw[c * 9 + t] += tmp[i];
  else if (dbias) dbias[c] += tmp[i];
}
void launch_dwconv_wgrad(int dt, const void* x, const void* dy, float* dw, float* dbias, float* scratch10C, int B, int H,
                         int W, int C, int OH, int OW, int stride, int pt, int pl, hipStream_t s) {
  DISPATCH_T(dt, {
    DwWgradF<T> f{(const T*)x, (const T*)dy, H, W, C, OH, OW, stride, pt, pl, 1.0f / (float)OW, 1.0f / ((float)OW * (float)OH), (long)B * OH * OW < (1L << 23) ? 1 : 0};
    if (scratch10C) {
      // (512 workgroups: 256 -> +0.23 ms, 128 -> +0.93 ms, 1024 -> +0.25 ms per step, round 4: the side stream's duration is on the critical path)
      launch_colreduce<T, 10>(f, (long)B * OH * OW, C, scratch10C, nullptr, -1, s, DWW_RPT, DWW_BLK, DWW_BLK);
      hipLaunchKernelGGL(dw_wgrad_scatter_kernel, dim3((C * 10 + 255) / 256), dim3(256), 0, s, scratch10C, dw, dbias, C);
    } else {
      launch_colreduce<T, 10>(f, (long)B * OH * OW, C, dw, dbias, 9, s, 16, DWW_BLK, DWW_BLK);
    }
  });
}

// ---- LayerNorm: one wave per row ----------------------------------------------------------------------
// RPW rows share a wave when a row has at most 64 / RPW chunks (SwinTRN's 96- and 192-channel stages would otherwise leave 52 / 40 of the
// 64 lanes idle): G = 64 / RPW lanes per row, reductions stay inside the G-lane group.
template <int G>
DEVI float group_sum(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <typename T, int NCH, int RPW>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* a, const T* b, const float* w, const float* bias,
                                                        T* out, float* mr, long R, int C, float eps, RowMap omap, LnAdd add) {
  constexpr int CH = TT<T>::CH;
  constexpr int G = 64 / RPW;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, sub = lane / G, l = lane % G;
  const int CC = C / CH;
  const uint32_t dseed = (add.on && add.drop_p > 0.f) ? *add.seed : 0u;
  for (long r = ((long)blockIdx.x * 4 + wv) * RPW + sub; r < R; r += (long)gridDim.x * 4 * RPW) {
    float v[NCH][CH];
    float sum = 0.f;
    const long rb = add.on ? rowmap_row(r, add.bmap) : r;
    const float bsc = (add.on && add.drop_p > 0.f) ? drop_scale(dseed, add.site, (uint32_t)(r / add.rows_per_sample), add.drop_p) : 1.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      int c = l + k * G;
      if (c < CC) {
        unpack<T>(ld16(a + r * C + c * CH), v[k]);
        if (b) {
          float t[CH];
          unpack<T>(ld16(b + rb * C + c * CH), t);
#pragma unroll
          for (int j = 0; j < CH; ++j) v[k][j] += t[j] * bsc;
          if (add.on) {   // the sum is kept (rounded to the compute dtype, and normalised from the rounded value: what a separate add would hand over)
            const uint4 pk = pack<T>(v[k]);
            st16((T*)add.sum_out + r * C + c * CH, pk);
            unpack<T>(pk, v[k]);
          }
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) sum += v[k][j];
      }
    }
    float mean = group_sum<G>(sum) / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      int c = l + k * G;
      if (c < CC) {
#pragma unroll
        for (int j = 0; j < CH; ++j) { float d = v[k][j] - mean; sq += d * d; }
      }
    }
    float rstd = rsqrtf(group_sum<G>(sq) / (float)C + eps);
    if (l == 0 && mr) { mr[r] = mean; mr[R + r] = rstd; }
    const long ro = rowmap_row(r, omap);
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      int c = l + k * G;
      if (c < CC) {
        float o[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) o[j] = (v[k][j] - mean) * rstd * w[c * CH + j] + bias[c * CH + j];
        st16(out + ro * C + c * CH, pack<T>(o));
      }
    }
  }
}
#define LN_GO(NCH, RPW) hipLaunchKernelGGL((layernorm_kernel<T, NCH, RPW>), dim3(grid_for(R, 4 * RPW, 2048)), dim3(256), 0, s, (const T*)a, \
                                           (const T*)b, w, bias, (T*)out, mr, R, C, eps, out_map, add)
void launch_layernorm(int dt, const void* a, const void* b, const float* w, const float* bias, void* out, float* mr,
                      long R, int C, float eps, float, const uint32_t*, uint32_t, hipStream_t s, RowMap out_map, LnAdd add) {
  DISPATCH_T(dt, {
    int cc = C / TT<T>::CH;
    if (cc <= 16) LN_GO(1, 4);
    else if (cc <= 32) LN_GO(1, 2);
    else if (cc <= 64) LN_GO(1, 1);
    else if (cc <= 128) LN_GO(2, 1);
    else if (cc <= 256) LN_GO(4, 1);
    else LN_GO(8, 1);  // up to 512 chunks per row (SwinTRN patch merging normalises 4*512 channels: 512 chunks in f32)
  });
}

template <typename T, int NCH, int RPW>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* dout, const T* a, const T* b, const float* w,
                                                            const float* mr, T* da, T* db, int beta_a, int beta_b,
                                                            float* dw, float* dbias, long R, int C, float* part, RowMap dmap, LnAdd add) {
  constexpr int CH = TT<T>::CH;
  constexpr int G = 64 / RPW;
  const T* gsum = add.on ? (const T*)add.gsum : nullptr;
  const uint32_t dseed = (add.on && add.drop_p > 0.f) ? *add.seed : 0u;
  extern __shared__ float red[];  // [4 waves][2][C]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, sub = lane / G, l = lane % G;
  const int CC = C / CH;
  float gw[NCH][CH], gb[NCH][CH];
#pragma unroll
  for (int k = 0; k < NCH; ++k)
#pragma unroll
    for (int j = 0; j < CH; ++j) gw[k][j] = gb[k][j] = 0.f;
  // a wave walks its rows with the NEXT row's operands (inputs, statistics and, when accumulating, the old gradient
  // values) already requested: every row would otherwise cost two dependent far round trips (inputs, then the
  // read-modify-write of da / db after the reduction)
  struct RowRaw { uint4 xa[NCH], xb[NCH], dd[NCH], oa[NCH], ob[NCH], gs[NCH]; float mean, rstd; };
  auto load_row = [&](long r, RowRaw& q) {
    q.mean = mr[r]; q.rstd = mr[R + r];
    const long rd = rowmap_row(r, dmap);
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int c = l + k * G;
      if (c < CC) {
        q.xa[k] = ld16(a + r * C + c * CH);
        if (b) q.xb[k] = ld16(b + r * C + c * CH);
        q.dd[k] = ld16(dout + rd * C + c * CH);
        if (beta_a) q.oa[k] = ld16(da + r * C + c * CH);
        if (db && beta_b) q.ob[k] = ld16(db + r * C + c * CH);
        if (gsum) q.gs[k] = ld16(gsum + r * C + c * CH);
      }
    }
  };
  const long rstep = (long)gridDim.x * 4 * RPW;
  long r = ((long)blockIdx.x * 4 + wv) * RPW + sub;
  RowRaw cur, nxt;
  if (r < R) load_row(r, cur);
  for (; r < R; r += rstep) {
    if (r + rstep < R) load_row(r + rstep, nxt);
    const float mean = cur.mean, rstd = cur.rstd;
    float xh[NCH][CH], g[NCH][CH];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      int c = l + k * G;
      if (c < CC) {
        float d[CH];
        unpack<T>(cur.xa[k], xh[k]);
        if (b) {
          float t[CH];
          unpack<T>(cur.xb[k], t);
#pragma unroll
          for (int j = 0; j < CH; ++j) xh[k][j] += t[j];
        }
        unpack<T>(cur.dd[k], d);
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          xh[k][j] = (xh[k][j] - mean) * rstd;
          g[k][j] = d[j] * w[c * CH + j];
          s1 += g[k][j];
          s2 += g[k][j] * xh[k][j];
          gw[k][j] += d[j] * xh[k][j];
          gb[k][j] += d[j];
        }
      }
    }
    s1 = group_sum<G>(s1) / (float)C;
    s2 = group_sum<G>(s2) / (float)C;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      int c = l + k * G;
      if (c < CC) {
        float o[CH], t[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) o[j] = rstd * (g[k][j] - s1 - xh[k][j] * s2);
        if (gsum) {
          unpack<T>(cur.gs[k], t);
#pragma unroll
          for (int j = 0; j < CH; ++j) o[j] += t[j];
        }
        if (beta_a) {
          unpack<T>(cur.oa[k], t);
#pragma unroll
          for (int j = 0; j < CH; ++j) t[j] += o[j];
          st16(da + r * C + c * CH, pack<T>(t));
        } else {
          st16(da + r * C + c * CH, pack<T>(o));
        }
        if (db) {
          if (add.on) {   // the branch's gradient: scaled by its sample's stochastic-depth factor, at its window row
            const long rb = rowmap_row(r, add.bmap);
            const float bsc = add.drop_p > 0.f ? drop_scale(dseed, add.site, (uint32_t)(r / add.rows_per_sample), add.drop_p) : 1.f;
#pragma unroll
            for (int j = 0; j < CH; ++j) t[j] = o[j] * bsc;
            st16(db + rb * C + c * CH, pack<T>(t));
          } else if (beta_b) {
            unpack<T>(cur.ob[k], t);
#pragma unroll
            for (int j = 0; j < CH; ++j) t[j] += o[j];
            st16(db + r * C + c * CH, pack<T>(t));
          } else {
            st16(db + r * C + c * CH, pack<T>(o));
          }
        }
      }
    }
    cur = nxt;
  }
  // the rows that shared a wave first (fixed order), then the four waves' partials through their own LDS slots
  if (RPW > 1) {
#pragma unroll
    for (int j = 0; j < CH; ++j) {
#pragma unroll
      for (int o = G; o < 64; o <<= 1) { gw[0][j] += __shfl_xor(gw[0][j], o, 64); gb[0][j] += __shfl_xor(gb[0][j], o, 64); }
    }
  }
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    int c = l + k * G;
    if (c < CC && sub == 0) {
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        red[wv * 2 * C + c * CH + j] = gw[k][j];
        red[wv * 2 * C + C + c * CH + j] = gb[k][j];
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    const float a = ((red[i] + red[2 * C + i]) + red[4 * C + i]) + red[6 * C + i];
    if (part) part[(size_t)blockIdx.x * 2 * C + i] = a;   // partial rows: folded afterwards in block order
    else atomicAdd((i < C ? dw : dbias - C) + i, a);
  }
}
// Weight / bias gradient partials folded off the critical path: with `part_ws` ([layernorm_bwd_blocks(R, C)][2][C] floats, the caller's)
// the kernel leaves one plain-stored partial row per block instead of 2*C contended atomics per block (512 blocks on 768 addresses cost
// half of the kernel's 30 us at [9216][384]), and launch_layernorm_fold adds them in block order -- on whichever stream the caller likes
// (the engine: the side stream, the sums feed only the optimizer).
int layernorm_bwd_blocks(long R) { return grid_for(R, 16, 1024); }

__global__ __launch_bounds__(256) void layernorm_fold_kernel(const float* part, int nrep_all, int n, int C, float* dw, float* dbias) {
  // gridDim.y > 1: the partial rows are split over blockIdx.y and each slice adds its sums with one atomic per column (a dozen blocks
  // walking 576 rows of 3 KB took 15 us; not in the deterministic mode, which keeps gridDim.y = 1 and the plain add)
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  const int per = (nrep_all + gridDim.y - 1) / gridDim.y;
  part += (size_t)blockIdx.y * per * n;
  const int nrep = max(0, min(per, nrep_all - (int)blockIdx.y * per));
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (i < n) {
    int r = wv;
    for (; r + 12 < nrep; r += 16) {
      a0 += part[(size_t)r * n + i]; a1 += part[(size_t)(r + 4) * n + i]; a2 += part[(size_t)(r + 8) * n + i]; a3 += part[(size_t)(r + 12) * n + i];
    }
    for (; r < nrep; r += 4) a0 += part[(size_t)r * n + i];
  }
  red[wv][lane] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (wv == 0 && i < n) {
    const float t = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    float* dst = i < C ? dw + i : dbias + (i - C);
    if (gridDim.y > 1) atomicAdd(dst, t); else *dst += t;
  }
}
void launch_layernorm_fold(const float* part, int nblocks, int C, float* dw, float* dbias, hipStream_t s) {
  const int ys = g_det.on ? 1 : (nblocks >= 256 ? 8 : (nblocks >= 64 ? 4 : 1));
  hipLaunchKernelGGL(layernorm_fold_kernel, dim3((2 * C + 63) / 64, ys), dim3(256), 0, s, part, nblocks, 2 * C, C, dw, dbias);
}

#define LNB_GO(NCH, RPW) hipLaunchKernelGGL((layernorm_bwd_kernel<T, NCH, RPW>), dim3(g), dim3(256), sh, s, (const T*)dout, (const T*)a, \
                                            (const T*)b, w, mr, (T*)da, (T*)db, beta_a, beta_b, dw, dbias, R, C, part, dout_map, add)
void launch_layernorm_bwd(int dt, const void* dout, const void* a, const void* b, const float* w, const float* mr,
                          void* da, void* db, int beta_a, int beta_b, float* dw, float* dbias, long R, int C, float,
                          const uint32_t*, uint32_t, hipStream_t s, float* part_ws, RowMap dout_map, LnAdd add) {
  DISPATCH_T(dt, {
    int cc = C / TT<T>::CH;
    int g = part_ws ? layernorm_bwd_blocks(R) : grid_for(R, 16, 512);
    size_t sh = (size_t)8 * C * sizeof(float);
    float* part = part_ws ? part_ws : det_scratch(s, (size_t)g * 2 * C);
    if (cc <= 16) LNB_GO(1, 4);
    else if (cc <= 32) LNB_GO(1, 2);
    else if (cc <= 64) LNB_GO(1, 1);
    else if (cc <= 128) LNB_GO(2, 1);
    else if (cc <= 256) LNB_GO(4, 1);
    else LNB_GO(8, 1);
    if (part && !part_ws) { launch_fold(part, g, 2L * C, C, dw, s); launch_fold(part + C, g, 2L * C, C, dbias, s); }
  });
}

// ---- EncoderLayer raw reshape: [b,hw,c] buffer reinterpreted as [b,c,h,w] (reference :269) -----------
// forward: Z[b][p][c] (NHWC of the reinterpreted tensor) = Y[b].flat[c*HW + p]; inverse scatters gradients back.
template <typename T>
__global__ void reshape_quirk_kernel(const T* in, T* out, int HW, int C, long total, int inverse, int beta) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long b = i / ((long)HW * C);
    int r = (int)(i - b * (long)HW * C);
    if (!inverse) {
      int p = r / C, c = r - p * C;  // output element Z[b][p][c]
      out[i] = in[b * (long)HW * C + (long)c * HW + p];
    } else {
      int c = r / HW, p = r - c * HW;  // output element dY[b].flat[c*HW+p] = dZ[b][p][c]
      float v = to_f(in[b * (long)HW * C + (long)p * C + c]);
      out[i] = from_f<T>(beta ? to_f(out[i]) + v : v);
    }
  }
}
void launch_reshape_quirk(int dt, int inverse, const void* in, void* out, int B, int HW, int C, int beta, hipStream_t s) {
  DISPATCH_T(dt, {
    long n = (long)B * HW * C;
    hipLaunchKernelGGL((reshape_quirk_kernel<T>), dim3(grid_for(n)), dim3(256), 0, s, (const T*)in, (T*)out, HW, C, n,
                       inverse, beta);
  });
}

// ---- device error word: set by kernels that meet an index they must not follow (a token id outside the embedding table /
// the vocabulary, e.g. the loader's -1 padding reaching the model before the -1 -> <PAD> rewrite of
// train_modules/train_single_opt.py:78); nn.Embedding / CrossEntropyLoss raise there, these kernels skip the element and
// flag it, and the next satrn_model_read_loss / satrn_device_error call reports it
__device__ unsigned g_satrn_errflag = 0;
unsigned* device_error_word() {
  static unsigned* p = nullptr;
  if (!p && hipGetSymbolAddress((void**)&p, HIP_SYMBOL(g_satrn_errflag)) != hipSuccess) p = nullptr;
  return p;
}
unsigned device_error_read_clear(hipStream_t s) {
  unsigned v = 0;
  if (hipMemcpyFromSymbolAsync(&v, HIP_SYMBOL(g_satrn_errflag), sizeof(v), 0, hipMemcpyDeviceToHost, s) != hipSuccess) return 0;
  (void)hipStreamSynchronize(s);
  if (v) { unsigned z = 0; (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_satrn_errflag), &z, sizeof(z), 0, hipMemcpyHostToDevice, s); (void)hipStreamSynchronize(s); }
  return v;
}

// ---- embedding * sqrt(D) + 1-D positional encoding (+dropout) ----------------------------------------
template <typename T>
__global__ void embed_kernel(const int64_t* ids, const float* table, const float* pe, T* out, int L, int ld_ids, int D,
                             int pos0, long total, float drop_p, const uint32_t* seedp, uint32_t site, int nrows) {
  const float sc = sqrtf((float)D);
  const uint32_t seed = drop_p > 0.f ? *seedp : 0u;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int d = (int)(i % D);
    long bt = i / D;
    int t = (int)(bt % L);
    long b = bt / L;
    int64_t id = ids[b * ld_ids + t];
    float tv = 0.f;
    if (nrows > 0 && (id < 0 || id >= nrows)) { if (d == 0) atomicOr(&g_satrn_errflag, 1u); }
    else tv = table[id * D + d];
    float v = tv * sc + pe[(long)(pos0 + t) * D + d];
    if (drop_p > 0.f) v *= drop_scale(seed, site, (uint32_t)i, drop_p);
    out[i] = from_f<T>(v);
  }
}
void launch_embed(int dt, const int64_t* ids, const float* table, const float* pe, void* out, int B, int L, int ld_ids,
                  int D, int pos0, float drop_p, const uint32_t* seed, uint32_t site, hipStream_t s, int nrows) {
  DISPATCH_T(dt, {
    long n = (long)B * L * D;
    hipLaunchKernelGGL((embed_kernel<T>), dim3(grid_for(n)), dim3(256), 0, s, ids, table, pe, (T*)out, L, ld_ids, D,
                       pos0, n, drop_p, seed, site, nrows);
  });
}
template <typename T>
__global__ void embed_bwd_kernel(const int64_t* ids, const T* dout, float* dtable, int L, int ld_ids, int D, long total,
                                 float drop_p, const uint32_t* seedp, uint32_t site, int nrows) {
  const float sc = sqrtf((float)D);
  const uint32_t seed = drop_p > 0.f ? *seedp : 0u;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int d = (int)(i % D);
    long bt = i / D;
    int t = (int)(bt % L);
    long b = bt / L;
    int64_t id = ids[b * ld_ids + t];
    if (nrows > 0 && (id < 0 || id >= nrows)) continue;  // flagged by the forward kernel
    float g = to_f(dout[i]) * sc;
    if (drop_p > 0.f) g *= drop_scale(seed, site, (uint32_t)i, drop_p);
    atomicAdd(dtable + id * D + d, g);
  }
}
// deterministic mode: one workgroup per table row walks all tokens in order (the table has 246 rows, a batch a few thousand tokens)
template <typename T>
__global__ __launch_bounds__(256) void embed_bwd_det_kernel(const int64_t* ids, const T* dout, float* dtable, int B, int L, int ld_ids,
                                                            int D, float drop_p, const uint32_t* seedp, uint32_t site) {
  const float sc = sqrtf((float)D);
  const uint32_t seed = drop_p > 0.f ? *seedp : 0u;
  const int64_t row = blockIdx.x;
  for (int d = threadIdx.x; d < D; d += 256) {
    float a = 0.f;
    bool any = false;
    for (long bt = 0; bt < (long)B * L; ++bt) {
      if (ids[(bt / L) * ld_ids + (bt % L)] != row) continue;
      const long i = bt * D + d;
      float g = to_f(dout[i]) * sc;
      if (drop_p > 0.f) g *= drop_scale(seed, site, (uint32_t)i, drop_p);
      a += g; any = true;
    }
    if (any) dtable[row * D + d] += a;
  }
}
void launch_embed_bwd(int dt, const int64_t* ids, const void* dout, float* dtable, int B, int L, int ld_ids, int D,
                      float drop_p, const uint32_t* seed, uint32_t site, hipStream_t s, int nrows) {
  if (g_det.on && nrows > 0) {
    DISPATCH_T(dt, { hipLaunchKernelGGL((embed_bwd_det_kernel<T>), dim3(nrows), dim3(256), 0, s, ids, (const T*)dout, dtable, B, L,
                                        ld_ids, D, drop_p, seed, site); });
    return;
  }
  DISPATCH_T(dt, {
    long n = (long)B * L * D;
    hipLaunchKernelGGL((embed_bwd_kernel<T>), dim3(grid_for(n)), dim3(256), 0, s, ids, (const T*)dout, dtable, L,
                       ld_ids, D, n, drop_p, seed, site, nrows);
  });
}

// ---- bias gradients: out[c] += sum_rows x[row*ld + c] --------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* x, long M, int C, int ld, float* out, int rows_per_block, float* part) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + cl;
  long r0 = (long)blockIdx.x * rows_per_block, r1 = r0 + rows_per_block;
  if (r1 > M) r1 = M;
  float sum = 0.f;
  if (c < C)
    for (long r = r0 + rg; r < r1; r += 4) sum += to_f(x[r * ld + c]);
  red[rg][cl] = sum;
  __syncthreads();
  if (rg == 0 && c < C) {
    const float a = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
    if (part) part[(size_t)blockIdx.x * C + c] = a;  // deterministic mode: folded afterwards
    else atomicAdd(out + c, a);
  }
}
// wide matrices (the raw-score gradients of SwinTRN's windows summed over the windows: 69 K .. 553 K columns, 16 .. 1 024 rows): a thread owns
// one 16-byte chunk of columns and walks a slice of the rows (colsum_kernel reads ONE element per thread and row: 127 us for 189 MB)
template <typename T>
__global__ __launch_bounds__(256) void colsum_wide_kernel(const T* x, long M, int CC, int ld, float* out, int rows_per_block) {
  constexpr int CH = TT<T>::CH;
  const int cc = blockIdx.y * 256 + threadIdx.x;
  if (cc >= CC) return;
  long r0 = (long)blockIdx.x * rows_per_block, r1 = r0 + rows_per_block;
  if (r1 > M) r1 = M;
  float acc[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) acc[j] = 0.f;
  const T* px = x + (long)cc * CH;
  long r = r0;
  for (; r + 3 < r1; r += 4) {
    const uint4 v0 = ld16(px + r * ld), v1 = ld16(px + (r + 1) * ld), v2 = ld16(px + (r + 2) * ld), v3 = ld16(px + (r + 3) * ld);
    float f0[CH], f1[CH], f2[CH], f3[CH];
    unpack<T>(v0, f0); unpack<T>(v1, f1); unpack<T>(v2, f2); unpack<T>(v3, f3);
#pragma unroll
    for (int j = 0; j < CH; ++j) acc[j] += (f0[j] + f1[j]) + (f2[j] + f3[j]);
  }
  for (; r < r1; ++r) {
    float f[CH];
    unpack<T>(ld16(px + r * ld), f);
#pragma unroll
    for (int j = 0; j < CH; ++j) acc[j] += f[j];
  }
  float* po = out + (long)cc * CH;
  if (gridDim.x == 1) {
#pragma unroll
    for (int j = 0; j < CH; ++j) po[j] += acc[j];
  } else {
#pragma unroll
    for (int j = 0; j < CH; ++j) atomicAdd(po + j, acc[j]);
  }
}
void launch_colsum(int dt, const void* x, long M, int C, int ld, float* out, hipStream_t s) {
  if (!g_det.on && C >= 8192 && (C % 8) == 0 && (ld % 8) == 0) {
    DISPATCH_T(dt, {
      const int CC = C / TT<T>::CH;
      const int gy = (CC + 255) / 256;
      // row slices only where the rows are many: a slice costs eight atomics per thread, which at 64 rows outweighed the reads (57 us against
      // 21 us for the one-element kernel); with one slice the sums are added with plain stores
      long gx = M >= 256 ? 1024 / gy : 1;
      if (gx > M / 32) gx = M / 32;
      if (gx < 1) gx = 1;
      const long rpb = (M + gx - 1) / gx;
      gx = (M + rpb - 1) / rpb;
      hipLaunchKernelGGL((colsum_wide_kernel<T>), dim3((int)gx, gy), dim3(256), 0, s, (const T*)x, M, CC, ld, out, (int)rpb);
    });
    return;
  }
  DISPATCH_T(dt, {
    int gy = (C + 63) / 64;
    long want = 512 / gy;
    if (want < 1) want = 1;
    long rpb = (M + want - 1) / want;
    if (rpb < 32) rpb = 32;
    int gx = (int)((M + rpb - 1) / rpb);
    float* part = det_scratch(s, (size_t)gx * C);
    hipLaunchKernelGGL((colsum_kernel<T>), dim3(gx, gy), dim3(256), 0, s, (const T*)x, M, C, ld, out, (int)rpb, part);
    if (part) launch_fold(part, gx, C, C, out, s);
  });
}

// ---- activation backward for GEMM-epilogue activations -------------------------------------------------
// RELU (+dropout): from the stored post-activation z.  SIGMOID: from post z.  SILU: from the PRE-activation u.
template <typename T>
__global__ void act_bwd_kernel(const T* dz, const T* zu, T* du, long n, int act, float keep_inv) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float d = to_f(dz[i]), z = to_f(zu[i]), r;
    if (act == ACT_RELU) r = z > 0.f ? d * keep_inv : 0.f;
    else if (act == ACT_SIGMOID) r = d * z * (1.f - z);
    else if (act == ACT_SILU || act == ACT_GELU || act == ACT_DFACTOR) r = d * act_bwd(z, act);   // zu holds the PRE-activation (DFACTOR: the derivative)
    else r = d;
    du[i] = from_f<T>(r);
  }
}
void launch_act_bwd(int dt, const void* dz, const void* zu, void* du, long n, int act, float drop_p, hipStream_t s) {
  DISPATCH_T(dt, {
    hipLaunchKernelGGL((act_bwd_kernel<T>), dim3(grid_for(n)), dim3(256), 0, s, (const T*)dz, (const T*)zu, (T*)du, n,
                       act, 1.0f / (1.0f - drop_p));
  });
}
template <typename T>
__global__ void act_fwd_kernel(const T* u, T* z, long n, int act) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    z[i] = from_f<T>(act_fwd(to_f(u[i]), act));
}
void launch_act_fwd(int dt, const void* u, void* z, long n, int act, hipStream_t s) {
  DISPATCH_T(dt, {
    hipLaunchKernelGGL((act_fwd_kernel<T>), dim3(grid_for(n)), dim3(256), 0, s, (const T*)u, (T*)z, n, act);
  });
}
template <typename T>
__global__ void dropout_bwd_kernel(const T* dz, T* du, long n, float p, const uint32_t* seedp, uint32_t site) {
  const uint32_t seed = *seedp;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    du[i] = from_f<T>(to_f(dz[i]) * drop_scale(seed, site, (uint32_t)i, p));
}
void launch_dropout_bwd(int dt, const void* dz, void* du, long M, int N, float p, const uint32_t* seed, uint32_t site,
                        hipStream_t s) {
  DISPATCH_T(dt, {
    long n = M * N;
    hipLaunchKernelGGL((dropout_bwd_kernel<T>), dim3(grid_for(n)), dim3(256), 0, s, (const T*)dz, (T*)du, n, p, seed,
                       site);
  });
}

// ---- cross-entropy (ignore_index) on fp32 logits [R][V]: one wave per row ---------------------------------
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* logits, const int64_t* tgt, int ld_tgt, int tgt_off,
                                                     int T_, int V, int pad_id, long R, float* out, float* lse_ws, float* part) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float lsum = 0.f, lcnt = 0.f;
  for (long r = (long)blockIdx.x * 4 + wv; r < R; r += (long)gridDim.x * 4) {
    const float* x = logits + r * V;
    float m = -INFINITY;
    for (int c = lane; c < V; c += 64) m = fmaxf(m, x[c]);
    m = wave_max(m);
    float se = 0.f;
    for (int c = lane; c < V; c += 64) se += __expf(x[c] - m);
    se = wave_sum(se);
    float lse = m + __logf(se);
    if (lane == 0) {
      lse_ws[r] = lse;
      int64_t t = tgt[(r / T_) * ld_tgt + tgt_off + (r % T_)];
      if (t != pad_id && (t < 0 || t >= V)) atomicOr(&g_satrn_errflag, 2u);  // torch raises here; flagged and ignored
      else if (t != pad_id) { lsum += lse - x[t]; lcnt += 1.f; }
    }
  }
  // one pair of atomics per BLOCK (thousands of same-address float atomics serialise at ~20 ns each)
  __shared__ float bsum[2][4];
  if (lane == 0) { bsum[0][wv] = lsum; bsum[1][wv] = lcnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = bsum[0][0] + bsum[0][1] + bsum[0][2] + bsum[0][3], b = bsum[1][0] + bsum[1][1] + bsum[1][2] + bsum[1][3];
    if (part) { part[blockIdx.x * 2] = a; part[blockIdx.x * 2 + 1] = b; }  // deterministic mode: folded afterwards
    else if (b > 0.f) { atomicAdd(out, a); atomicAdd(out + 1, b); }
  }
}
template <typename TO>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* logits, const int64_t* tgt, int ld_tgt, int tgt_off,
                                                     int T_, int V, int Vp, int pad_id, long R, float* out,
                                                     const float* lse_ws, TO* dlogits, const float* upstream) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float cnt = out[1];
  const float sc = (cnt > 0.f ? 1.0f / cnt : 0.f) * (upstream ? *upstream : 1.0f);
  if (blockIdx.x == 0 && threadIdx.x == 0) out[2] = cnt > 0.f ? out[0] / cnt : 0.f;
  for (long r = (long)blockIdx.x * 4 + wv; r < R; r += (long)gridDim.x * 4) {
    const float* x = logits + r * V;
    int64_t t = tgt[(r / T_) * ld_tgt + tgt_off + (r % T_)];
    const float lse = lse_ws[r];
    const bool valid = t != pad_id && t >= 0 && t < V;
    for (int c = lane; c < Vp; c += 64) {
      float g = 0.f;
      if (valid && c < V) g = (__expf(x[c] - lse) - (c == t ? 1.f : 0.f)) * sc;
      dlogits[r * Vp + c] = from_f<TO>(g);
    }
  }
}
// loss_out: [0]=sum, [1]=count, [2]=mean (written by the backward kernel), [4..4+R) = per-row lse scratch
void launch_ce(const float* logits, const int64_t* tgt, int ld_tgt, int tgt_off, int B, int T_, int V, int pad_id,
               float* loss_out, float* dlogits, hipStream_t s);

template <typename TO>
static void launch_ce_bwd_t(const float* logits, const int64_t* tgt, int ld_tgt, int tgt_off, int B, int T_, int V,
                            int Vp, int pad_id, float* out, const float* lse_ws, void* dl, const float* upstream,
                            hipStream_t s) {
  long R = (long)B * T_;
  hipLaunchKernelGGL((ce_bwd_kernel<TO>), dim3(grid_for(R, 4, 2048)), dim3(256), 0, s, logits, tgt, ld_tgt, tgt_off, T_,
                     V, Vp, pad_id, R, out, lse_ws, (TO*)dl, upstream);
}
void launch_ce_full(int dt_out, const float* logits, const int64_t* tgt, int ld_tgt, int tgt_off, int B, int T_, int V,
                    int Vp, int pad_id, float* loss_out /*[4]*/, float* lse_ws /*[R]*/, void* dlogits,
                    const float* upstream, hipStream_t s) {
  long R = (long)B * T_;
  launch_fill(loss_out, 0, 4 * sizeof(float), s);
  const int gce = grid_for(R, 16, 256);
  float* part = det_scratch(s, (size_t)gce * 2);
  hipLaunchKernelGGL(ce_fwd_kernel, dim3(gce), dim3(256), 0, s, logits, tgt, ld_tgt, tgt_off, T_, V,
                     pad_id, R, loss_out, lse_ws, part);
  if (part) launch_fold(part, gce, 2, 2, loss_out, s);
  if (dt_out == DT_BF16) launch_ce_bwd_t<bf16_t>(logits, tgt, ld_tgt, tgt_off, B, T_, V, Vp, pad_id, loss_out, lse_ws, dlogits, upstream, s);
  else launch_ce_bwd_t<float>(logits, tgt, ld_tgt, tgt_off, B, T_, V, Vp, pad_id, loss_out, lse_ws, dlogits, upstream, s);
}

// ---- knowledge-distillation loss (train_modules/train_distillation.py:49-55) on fp32 logits [R = B*T][V]: one wave per
// row.  loss = alpha*Tk^2/B * sum_rows KL(softmax(t/Tk) || softmax(s/Tk)) + (1-alpha)/R * sum_rows CE(s, label)  (the
// reference's CE here has no ignore_index).  dlogits = d loss / d s, written in the same pass.
__global__ __launch_bounds__(256) void kd_loss_kernel(const float* st, const float* te, const int64_t* lab, int ld_lab, int T_,
                                                      int V, long R, float inv_tk, float w_kd, float w_ce, float* out,
                                                      float* dl) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float lsum = 0.f;
  for (long r = (long)blockIdx.x * 4 + wv; r < R; r += (long)gridDim.x * 4) {
    const float* x = st + r * V;
    const float* y = te + r * V;
    float mx = -INFINITY, my = -INFINITY;
    for (int c = lane; c < V; c += 64) { mx = fmaxf(mx, x[c]); my = fmaxf(my, y[c]); }
    mx = wave_max(mx); my = wave_max(my);
    float s1 = 0.f, sk = 0.f, tk = 0.f;
    for (int c = lane; c < V; c += 64) {
      s1 += expf(x[c] - mx); sk += expf((x[c] - mx) * inv_tk); tk += expf((y[c] - my) * inv_tk);
    }
    s1 = wave_sum(s1); sk = wave_sum(sk); tk = wave_sum(tk);
    const float lse1 = mx + logf(s1), lsek = mx * inv_tk + logf(sk), lset = my * inv_tk + logf(tk);
    const int64_t t = lab[(r / T_) * ld_lab + (r % T_)];
    float kl = 0.f;
    for (int c = lane; c < V; c += 64) {
      const float lq = x[c] * inv_tk - lsek, lp = y[c] * inv_tk - lset, pp = expf(lp);
      kl += pp * (lp - lq);
      dl[r * V + c] = w_kd * inv_tk * (expf(lq) - pp) + w_ce * (expf(x[c] - lse1) - (c == t ? 1.f : 0.f));
    }
    kl = wave_sum(kl);
    if (lane == 0) lsum += w_kd * kl + w_ce * (lse1 - x[t]);
  }
  __shared__ float bsum[4];
  if (lane == 0) bsum[wv] = lsum;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, bsum[0] + bsum[1] + bsum[2] + bsum[3]);
}
void launch_kd_loss(const float* student, const float* teacher, const int64_t* labels, int ld_labels, int B, int T_, int V,
                    float temperature, float alpha, float* loss_out, float* dlogits, hipStream_t s) {
  const long R = (long)B * T_;
  launch_fill(loss_out, 0, sizeof(float), s);
  hipLaunchKernelGGL(kd_loss_kernel, dim3(grid_for(R, 4, 2048)), dim3(256), 0, s, student, teacher, labels, ld_labels, T_, V, R,
                     1.0f / temperature, alpha * temperature * temperature / (float)B, (1.0f - alpha) / (float)R, loss_out,
                     dlogits);
}

// ---- per-step training metrics on the device (train_modules/train_single_opt.py:101-109; utils/utils.py:134-164
// id_to_string(do_eval=1); utils/metrics.py:9-34): one wavefront per sample.  Token lists = ids with <PAD>/<SOS>/-1
// dropped, cut at <EOS>, plus one trailing "empty" element (the '' that .split(" ") leaves; the "" vocabulary entry is
// the same string).  WER term = Levenshtein(pred, truth) / max(len), sentence = lists equal, symbols = position-wise
// matches against expected[:, 1:] with <PAD> never matching.  acc (double [5]) += {sum_wer, sentences, correct
// sentences, correct symbols, total symbols}.
#define MET_MAXLEN 512
__global__ __launch_bounds__(64) void step_metrics_kernel(const int64_t* seq, int ld_seq, int T_, const int64_t* exp, int ld_exp,
                                                          int L, int pad_id, int sos_id, int eos_id, int empty_id, double* acc) {
  __shared__ int a[MET_MAXLEN + 1], g[MET_MAXLEN + 1];
  __shared__ int d0[MET_MAXLEN + 2], d1[MET_MAXLEN + 2], d2[MET_MAXLEN + 2];
  __shared__ int na_s, ng_s;
  const int b = blockIdx.x, lane = threadIdx.x;
  const int64_t* sr = seq + (long)b * ld_seq;
  const int64_t* er = exp + (long)b * ld_exp;
  if (lane == 0) {
    int n = 0;
    for (int t = 0; t < T_ && n < MET_MAXLEN; ++t) {
      const int v = (int)sr[t];
      if (v == eos_id) break;
      if (v == pad_id || v == sos_id || v == -1) continue;
      a[n++] = v == empty_id ? -2 : v;
    }
    a[n++] = -2;
    na_s = n;
    n = 0;
    for (int t = 0; t < L && n < MET_MAXLEN; ++t) {
      const int v = (int)er[t];
      if (v == eos_id) break;
      if (v == pad_id || v == sos_id || v == -1) continue;
      g[n++] = v == empty_id ? -2 : v;
    }
    g[n++] = -2;
    ng_s = n;
  }
  // position-wise symbol statistics
  int cs = 0, ts = 0;
  for (int t = lane; t < T_ && t + 1 < L; t += 64) {
    const int e = (int)er[t + 1];
    const bool valid = e != pad_id && e != -1;
    ts += valid;
    cs += valid && (int)sr[t] == e;
  }
  cs = (int)wave_sum((float)cs);
  ts = (int)wave_sum((float)ts);
  __syncthreads();
  const int na = na_s, ng = ng_s;
  // Levenshtein over anti-diagonals: cell (i, j), i in 0..na, j = k - i in 0..ng; dX[i] holds diagonal X
  int* pp = d0; int* p1 = d1; int* cur = d2;  // k-2, k-1, k
  if (lane == 0) { pp[0] = 0; p1[0] = 1; p1[1] = 1; }  // k = 0: (0,0); k = 1: (0,1) = 1, (1,0) = 1
  __syncthreads();
  for (int k = 2; k <= na + ng; ++k) {
    const int ilo = k - ng > 0 ? k - ng : 0, ihi = k < na ? k : na;
    for (int i = ilo + lane; i <= ihi; i += 64) {
      const int j = k - i;
      int v;
      if (i == 0) v = j;
      else if (j == 0) v = i;
      else {
        const int sub = pp[i - 1] + (a[i - 1] != g[j - 1]);
        const int del = p1[i - 1] + 1, ins = p1[i] + 1;
        v = sub < del ? sub : del;
        v = v < ins ? v : ins;
      }
      cur[i] = v;
    }
    __syncthreads();
    int* t = pp; pp = p1; p1 = cur; cur = t;
  }
  if (lane == 0) {
    const int dist = (na + ng >= 2) ? p1[na] : (na + ng == 1 ? 1 : 0);
    bool same = na == ng;
    for (int i = 0; same && i < na; ++i) same = a[i] == g[i];
    atomicAdd(acc + 0, (double)dist / (double)(na > ng ? na : ng));
    atomicAdd(acc + 1, 1.0);
    atomicAdd(acc + 2, same ? 1.0 : 0.0);
    atomicAdd(acc + 3, (double)cs);
    atomicAdd(acc + 4, (double)ts);
  }
}
void launch_step_metrics(const int64_t* seq, int ld_seq, int T_, const int64_t* exp, int ld_exp, int L, int B, int pad_id,
                         int sos_id, int eos_id, int empty_id, double* acc, hipStream_t s) {
  hipLaunchKernelGGL(step_metrics_kernel, dim3(B), dim3(64), 0, s, seq, ld_seq, T_, exp, ld_exp, L, pad_id, sos_id, eos_id, empty_id, acc);
}

// ---- misc -----------------------------------------------------------------------------------------------
template <typename TI, typename TO>
__global__ void cast_kernel(const TI* in, TO* out, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = from_f<TO>(to_f(in[i]));
}
void launch_cast(int dt_in, int dt_out, const void* in, void* out, long n, hipStream_t s) {
  int g = grid_for(n);
  if (dt_in == DT_F32 && dt_out == DT_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3(g), dim3(256), 0, s, (const float*)in, (bf16_t*)out, n);
  else if (dt_in == DT_BF16 && dt_out == DT_F32) hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3(g), dim3(256), 0, s, (const bf16_t*)in, (float*)out, n);
  else if (dt_in == DT_F32) (void)hipMemcpyAsync(out, in, n * 4, hipMemcpyDeviceToDevice, s);
  else (void)hipMemcpyAsync(out, in, n * 2, hipMemcpyDeviceToDevice, s);
}
// padded cast: in [R][C] fp32 -> out [R][Cp] as T with zero fill
template <typename TO>
__global__ void cast_pad_kernel(const float* in, TO* out, long R, int C, int Cp) {
  long n = R * Cp;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long r = i / Cp;
    int c = (int)(i - r * Cp);
    out[i] = from_f<TO>(c < C ? in[r * C + c] : 0.f);
  }
}
void launch_cast_pad(int dt_out, const float* in, void* out, long R, int C, int Cp, hipStream_t s) {
  int g = grid_for(R * Cp);
  if (dt_out == DT_BF16) hipLaunchKernelGGL((cast_pad_kernel<bf16_t>), dim3(g), dim3(256), 0, s, in, (bf16_t*)out, R, C, Cp);
  else hipLaunchKernelGGL((cast_pad_kernel<float>), dim3(g), dim3(256), 0, s, in, (float*)out, R, C, Cp);
}
// zero fill as an ordinary kernel (captured hipGraph memset nodes of tens of MB proved unreliable on replay)
__global__ void zero_kernel(uint4* p, size_t n16, unsigned char* tail, size_t ntail) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(0u, 0u, 0u, 0u);
  if (blockIdx.x == 0 && threadIdx.x < ntail) tail[threadIdx.x] = 0;
}
__global__ void zero_bytes_kernel(unsigned char* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0;
}
void launch_fill(void* p, int value_byte, size_t bytes, hipStream_t s) {
  if (value_byte != 0) { (void)hipMemsetAsync(p, value_byte, bytes, s); return; }
  if ((((size_t)p) & 15) != 0) {
    int g = (int)((bytes + 255) / 256);
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(zero_bytes_kernel, dim3(g < 1 ? 1 : g), dim3(256), 0, s, (unsigned char*)p, bytes);
    return;
  }
  size_t n16 = bytes / 16, ntail = bytes % 16;
  int g = (int)((n16 + 255) / 256);
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(zero_kernel, dim3(g), dim3(256), 0, s, (uint4*)p, n16, (unsigned char*)p + n16 * 16, ntail);
}

template <typename T>
__global__ void add_kernel(const T* a, const T* b, T* out, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = from_f<T>(to_f(a[i]) + to_f(b[i]));
}
void launch_add(int dt, const void* a, const void* b, void* out, long n, hipStream_t s) {
  DISPATCH_T(dt, { hipLaunchKernelGGL((add_kernel<T>), dim3(grid_for(n)), dim3(256), 0, s, (const T*)a, (const T*)b, (T*)out, n); });
}

__global__ __launch_bounds__(256) void argmax_kernel(const float* logits, int64_t* ids, int R, int V, int ld_in, int ld_out) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int r = blockIdx.x * 4 + wv;
  if (r >= R) return;
  const float* x = logits + (long)r * ld_in;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = lane; c < V; c += 64) {
    float v = x[c];
    if (v > best) { best = v; bi = c; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float ob = __shfl_xor(best, o, 64);
    int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if (lane == 0) ids[(long)r * ld_out] = bi;
}
void launch_argmax(const float* logits, int64_t* ids, int R, int V, int ld_in, int ld_out, hipStream_t s) {
  hipLaunchKernelGGL(argmax_kernel, dim3((R + 3) / 4), dim3(256), 0, s, logits, ids, R, V, ld_in, ld_out);
}
// dst[(b*dbs + doff + j)][:] (+)= src[(b*sbs + soff + j)][:] for b < B, j < n: row gather/scatter of the step decoder
template <typename T>
__global__ void copy_rows_kernel(const T* src, T* dst, int n, int C, long sbs, long soff, long dbs, long doff, long total,
                                 int beta) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    long r = i / C;
    int j = (int)(r % n);
    long b = r / n;
    const long so = (b * sbs + soff + j) * C + c, d_o = (b * dbs + doff + j) * C + c;
    dst[d_o] = beta ? from_f<T>(to_f(dst[d_o]) + to_f(src[so])) : src[so];
  }
}
void launch_copy_rows(int dt, const void* src, void* dst, int B, int n, int C, long sbs, long soff, long dbs, long doff,
                      int beta, hipStream_t s) {
  long total = (long)B * n * C;
  if (total <= 0) return;
  DISPATCH_T(dt, {
    hipLaunchKernelGGL((copy_rows_kernel<T>), dim3(grid_for(total)), dim3(256), 0, s, (const T*)src, (T*)dst, n, C, sbs, soff,
                       dbs, doff, total, beta);
  });
}
__global__ void fill_i64_kernel(int64_t* p, int64_t v, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}
void launch_fill_i64(int64_t* p, int64_t v, long n, hipStream_t s) { hipLaunchKernelGGL(fill_i64_kernel, dim3(grid_for(n)), dim3(256), 0, s, p, v, n); }
__global__ void seed_advance_kernel(uint32_t* seed) { *seed = mix32(*seed + 0x9E3779B9u); }
void launch_seed_advance(uint32_t* seed, hipStream_t s) { hipLaunchKernelGGL(seed_advance_kernel, dim3(1), dim3(1), 0, s, seed); }
// a handful of scalars into device memory as KERNEL ARGUMENTS: a host-to-device copy of 36 bytes runs as a blit on another queue and cost
// the chain ~50 us of idle time at every step boundary (tools/step_boundary.sh); a one-wave kernel stays in the stream's own queue
struct Scalars12 { float v[12]; };
__global__ void set_scalars_kernel(float* dst, Scalars12 h, int n) { if ((int)threadIdx.x < n) dst[threadIdx.x] = h.v[threadIdx.x]; }
void launch_set_scalars(float* dst, const float* src, int n, hipStream_t s) {
  Scalars12 h;
  for (int i = 0; i < 12; ++i) h.v[i] = i < n ? src[i] : 0.f;
  hipLaunchKernelGGL(set_scalars_kernel, dim3(1), dim3(64), 0, s, dst, h, n > 12 ? 12 : n);
}

// ---- weight packing (fp32 master -> compute dtype, contraction-major copies) --------------------------------
template <typename T>
__global__ void pack_dense_kernel(const float* w, T* fwd, T* bwd, int N, int K, int ldb) {
  long n = (long)N * K;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int r = (int)(i / K), k = (int)(i - (long)r * K);
    T v = from_f<T>(w[i]);
    if (fwd) fwd[i] = v;
    if (bwd) bwd[(long)k * ldb + r] = v;
  }
}
void launch_pack_dense(int dt, const float* w, void* fwd, void* bwd, int N, int K, hipStream_t s);
void launch_pack_dense_ld(int dt, const float* w, void* fwd, void* bwd, int N, int K, int ldb, hipStream_t s) {
  DISPATCH_T(dt, {
    hipLaunchKernelGGL((pack_dense_kernel<T>), dim3(grid_for((long)N * K)), dim3(256), 0, s, w, (T*)fwd, (T*)bwd, N, K, ldb);
  });
}
void launch_pack_dense(int dt, const float* w, void* fwd, void* bwd, int N, int K, hipStream_t s) {
  launch_pack_dense_ld(dt, w, fwd, bwd, N, K, N, s);
}
template <typename T>
__global__ void pack_conv_kernel(const float* w, T* fwd, T* bwd, int Co, int Ci, int taps) {
  long n = (long)Co * Ci * taps;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int t = (int)(i % taps);
    int ci = (int)((i / taps) % Ci);
    int co = (int)(i / ((long)taps * Ci));
    T v = from_f<T>(w[i]);
    fwd[((long)co * taps + t) * Ci + ci] = v;
    bwd[((long)ci * taps + t) * Co + co] = v;
  }
}
void launch_pack_conv(int dt, const float* w, void* fwd, void* bwd, int Co, int Ci, int taps, hipStream_t s) {
  DISPATCH_T(dt, {
    hipLaunchKernelGGL((pack_conv_kernel<T>), dim3(grid_for((long)Co * Ci * taps)), dim3(256), 0, s, w, (T*)fwd, (T*)bwd, Co, Ci, taps);
  });
}
template <typename T>
__global__ void pack_dw_kernel(const float* w, T* out, int C) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C * 9) return;
  int c = i / 9, t = i % 9;
  out[t * C + c] = from_f<T>(w[i]);
}
void launch_pack_dw(int dt, const float* w, void* out, int C, hipStream_t s) {
  DISPATCH_T(dt, { hipLaunchKernelGGL((pack_dw_kernel<T>), dim3((C * 9 + 255) / 256), dim3(256), 0, s, w, (T*)out, C); });
}

// one launch for every weight: blockIdx -> (descriptor, work item) through a host-built table.
// dense weights: work item = one 64x64 tile, transposed through LDS so both copies are written coalesced;
// conv3x3 / depthwise (small): work item = PACK_BLK consecutive elements.
template <typename T>
__global__ __launch_bounds__(256) void pack_all_kernel(const PackDesc* d, const int2* blk) {
  __shared__ float tile[64][65];
  const int2 bi = blk[blockIdx.x];
  const PackDesc e = d[bi.x];
  if (e.kind == 0) {
    const int tiles_k = (e.K + 63) / 64;
    const int n0 = (bi.y / tiles_k) * 64, k0 = (bi.y % tiles_k) * 64;
    const int c = threadIdx.x & 63, r0 = threadIdx.x >> 6;
#pragma unroll 4
    for (int r = r0; r < 64; r += 4) {
      float v = 0.f;
      if (n0 + r < e.N && k0 + c < e.K) {
        v = e.src[(long)(n0 + r) * e.K + k0 + c];
        ((T*)e.fwd)[(long)(n0 + r) * e.K + k0 + c] = from_f<T>(v);
      }
      tile[r][c] = v;
    }
    __syncthreads();
#pragma unroll 4
    for (int r = r0; r < 64; r += 4)
      if (k0 + r < e.K && n0 + c < e.N) ((T*)e.bwd)[(long)(k0 + r) * e.ldb + n0 + c] = from_f<T>(tile[c][r]);
    return;
  }
  const long n = e.kind == 1 ? (long)e.N * e.K * 9 : (long)e.N * 9;
  const long j0 = (long)bi.y * PACK_BLK;
  for (int t = threadIdx.x; t < PACK_BLK; t += 256) {
    const long j = j0 + t;
    if (j >= n) break;
    const T v = from_f<T>(e.src[j]);
    if (e.kind == 1) {  // conv3x3 [Co][Ci][9] -> fwd [Co][9][Ci], bwd [Ci][9][Co]
      int t9 = (int)(j % 9);
      int ci = (int)((j / 9) % e.K);
      int co = (int)(j / (9L * e.K));
      ((T*)e.fwd)[((long)co * 9 + t9) * e.K + ci] = v;
      ((T*)e.bwd)[((long)ci * 9 + t9) * e.N + co] = v;
    } else {  // depthwise [C][9] -> [9][C]
      int c = (int)(j / 9), t9 = (int)(j - (long)c * 9);
      ((T*)e.fwd)[(long)t9 * e.N + c] = v;
    }
  }
}
void launch_pack_all(int dt, const PackDesc* d, const void* blk, int nblk, hipStream_t s) {
  DISPATCH_T(dt, { hipLaunchKernelGGL((pack_all_kernel<T>), dim3(nblk), dim3(256), 0, s, d, (const int2*)blk); });
}

// ---- global grad-norm clip + AdamW over the flat parameter buffer ----------------------------------------------
// deterministic two-stage sum of squares (fixed grid, fixed reduction order): data-parallel replicas must derive
// bit-identical clip factors from their identical all-reduced gradients, or they drift apart
#define SUMSQ_BLOCKS 1024
__global__ __launch_bounds__(256) void sumsq_kernel(const float* g, long n, float* partial) {
  __shared__ float red[4];
  float s = 0.f;
  long i = (blockIdx.x * (long)blockDim.x + threadIdx.x) * 4;
  long stride = (long)gridDim.x * blockDim.x * 4;
  for (; i + 3 < n; i += stride) {
    float4 v = *reinterpret_cast<const float4*>(g + i);
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  for (; i < n; i += stride)
    for (long j = i; j < n && j < i + 4; ++j) s += g[j] * g[j];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* partial, int nb, float* out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) s += partial[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *out += (red[0] + red[1]) + (red[2] + red[3]);
}
void launch_sumsq(const float* g, long n, float* out, float* partial /*[SUMSQ_BLOCKS] scratch*/, hipStream_t s) {
  hipLaunchKernelGGL(sumsq_kernel, dim3(SUMSQ_BLOCKS), dim3(256), 0, s, g, n, partial);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, s, partial, SUMSQ_BLOCKS, out);
}
// hyper: [0] lr [1] beta1 [2] beta2 [3] eps [4] weight_decay [5] max_norm [6] 1-beta1^t [7] 1-beta2^t [8] grad_scale
__global__ void adamw_kernel(float* p, const float* g, float* m, float* v, long n, const float* gnorm_sq,
                             const float* hy) {
  const float lr = hy[0], b1 = hy[1], b2 = hy[2], eps = hy[3], wd = hy[4], maxn = hy[5], bc1 = hy[6], bc2 = hy[7],
              gs = hy[8];
  float coef = gs;
  if (maxn > 0.f) {
    float norm = sqrtf(*gnorm_sq) * gs;
    coef *= fminf(1.f, maxn / (norm + 1e-6f));
  }
  const float rs2 = rsqrtf(bc2);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float gi = g[i] * coef;
    float pi = p[i] * (1.f - lr * wd);
    float mi = b1 * m[i] + (1.f - b1) * gi;
    float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = pi - (lr / bc1) * mi / (sqrtf(vi) * rs2 + eps);
  }
}
void launch_adamw(float* p, const float* g, float* m, float* v, long n, const float* gnorm_sq, const float* hyper,
                  hipStream_t s) {
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0, s, p, g, m, v, n, gnorm_sq, hyper);
}
