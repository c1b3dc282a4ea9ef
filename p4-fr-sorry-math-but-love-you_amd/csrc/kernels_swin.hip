// Kernels specific to the SwinTRN path (BASELINE configs[3]; reference networks/SWIN.py): everything that is not a plain
// GEMM / LayerNorm / attention launch -- patch extraction for the 4x4 stride-4 patch embedding (:559-572), the absolute
// position embedding add (:693-694), cyclic shift + window partition / reverse as ONE row permutation (:49-80,338-371),
// the relative-position-bias gather and its gradient (:166-176), patch merging's 2x2 neighbourhood gather (:411-415),
// stochastic depth (timm DropPath, :373-374) and the sum over windows of the score gradient (bias gradient).
// Tokens are rows of [rows][C] tensors, moved in 16-byte chunks.
#include "common.h"
#include "kernels.h"

#define DISPATCH_T(dt, ...)                      \
  do {                                           \
    if ((dt) == DT_BF16) { typedef bf16_t T; __VA_ARGS__; } \
    else { typedef float T; __VA_ARGS__; }       \
  } while (0)

static inline int grid_for(long work, int per_block = 256, int cap = 8192) {
  long g = (work + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

// ---- patch extraction: img fp32 NCHW -> rows [B*(H/P)*(W/P)][Cin*P*P], k = (ci*P + ky)*P + kx (= the flattened conv weight)
template <typename T>
__global__ void patchify_kernel(const float* img, T* out, int B, int Cin, int H, int W, int P, long total) {
  const int K = Cin * P * P, OW = W / P, OH = H / P;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i % K);
    const long row = i / K;
    const int ox = (int)(row % OW), oy = (int)((row / OW) % OH), b = (int)(row / ((long)OW * OH));
    const int kx = k % P, ky = (k / P) % P, ci = k / (P * P);
    out[i] = from_f<T>(img[(((long)b * Cin + ci) * H + oy * P + ky) * W + ox * P + kx]);
  }
}
void launch_patchify(int dt, const float* img, void* out, int B, int Cin, int H, int W, int P, hipStream_t s) {
  const long n = (long)B * (H / P) * (W / P) * Cin * P * P;
  DISPATCH_T(dt, { hipLaunchKernelGGL((patchify_kernel<T>), dim3(grid_for(n)), dim3(256), 0, s, img, (T*)out, B, Cin, H, W, P, n); });
}

// ---- out[b][l][c] = x[b][l][c] + table[l][c]  (absolute position embedding; table fp32 parameter [L][C])
template <typename T>
__global__ void add_rows_table_kernel(const T* x, const float* table, T* out, long per_sample_chunks, long total_chunks) {
  constexpr int CH = TT<T>::CH;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total_chunks; i += (long)gridDim.x * blockDim.x) {
    float v[CH];
    unpack<T>(ld16(x + i * CH), v);
    const float* t = table + (i % per_sample_chunks) * CH;
#pragma unroll
    for (int j = 0; j < CH; ++j) v[j] += t[j];
    st16(out + i * CH, pack<T>(v));
  }
}
void launch_add_rows_table(int dt, const void* x, const float* table, void* out, int B, long LC, hipStream_t s) {
  DISPATCH_T(dt, {
    const long per = LC / TT<T>::CH, n = per * B;
    hipLaunchKernelGGL((add_rows_table_kernel<T>), dim3(grid_for(n)), dim3(256), 0, s, (const T*)x, table, (T*)out, per, n);
  });
}

// ---- cyclic shift + window partition as a row permutation.  Window row r = ((b*nWh + wy)*nWw + wx)*ws*ws + py*ws + px holds
// token (b, (wy*ws + py + shift) mod H, (wx*ws + px + shift) mod W): torch.roll(x, -shift) then window_partition (:338-351).
// reverse == 0: out[r] = in[token]; reverse == 1: out[token] (+)= in[r]  (window_reverse + roll back, :361-371)
template <typename T>
__global__ void window_perm_kernel(const T* in, T* out, int H, int W, int C, int ws, int shift, int reverse, int beta, long total_chunks) {
  constexpr int CH = TT<T>::CH;
  const int CC = C / CH, nWw = W / ws, nWh = H / ws, N = ws * ws;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total_chunks; i += (long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % CC);
    const long r = i / CC;
    const int p_ = (int)(r % N);
    const long w_ = r / N;
    const int wx = (int)(w_ % nWw), wy = (int)((w_ / nWw) % nWh);
    const long b = w_ / ((long)nWw * nWh);
    const int py = p_ / ws, px = p_ - py * ws;
    int y = wy * ws + py + shift, x = wx * ws + px + shift;
    if (y >= H) y -= H;
    if (x >= W) x -= W;
    const long tok = (b * H + y) * W + x;
    if (!reverse) {
      st16(out + (r * CC + cc) * CH, ld16(in + (tok * CC + cc) * CH));
    } else if (!beta) {
      st16(out + (tok * CC + cc) * CH, ld16(in + (r * CC + cc) * CH));
    } else {
      float a[CH], o[CH];
      unpack<T>(ld16(in + (r * CC + cc) * CH), a);
      unpack<T>(ld16(out + (tok * CC + cc) * CH), o);
#pragma unroll
      for (int j = 0; j < CH; ++j) o[j] += a[j];
      st16(out + (tok * CC + cc) * CH, pack<T>(o));
    }
  }
}
void launch_window_perm(int dt, const void* in, void* out, int B, int H, int W, int C, int ws, int shift, int reverse, int beta, hipStream_t s) {
  DISPATCH_T(dt, {
    const long n = (long)B * H * W * (C / TT<T>::CH);
    hipLaunchKernelGGL((window_perm_kernel<T>), dim3(grid_for(n)), dim3(256), 0, s, (const T*)in, (T*)out, H, W, C, ws, shift, reverse, beta, n);
  });
}

// ---- patch merging gather (:411-415): out[b][y2][x2][q*C + c] = x[b][2*y2 + (q & 1)][2*x2 + (q >> 1)][c], q = 0..3 in the order
// x0 (0,0), x1 (row+1), x2 (col+1), x3 (both).  reverse: the inverse copy (every input element appears exactly once).
template <typename T>
__global__ void patch_merge_kernel(const T* in, T* out, int H, int W, int C, int reverse, int beta, long total_chunks) {
  constexpr int CH = TT<T>::CH;
  const int CC = C / CH, H2 = H / 2, W2 = W / 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total_chunks; i += (long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % CC);
    long r = i / CC;
    const int q = (int)(r % 4); r /= 4;
    const int x2 = (int)(r % W2), y2 = (int)((r / W2) % H2);
    const long b = r / ((long)W2 * H2);
    const long tok = (b * H + 2 * y2 + (q & 1)) * W + 2 * x2 + (q >> 1);
    const long mrow = (b * H2 + y2) * W2 + x2;
    T* mo = (reverse ? (T*)in : out) + ((mrow * 4 + q) * CC + cc) * CH;   // merged-side address
    if (!reverse) st16(mo, ld16(in + (tok * CC + cc) * CH));
    else if (!beta) st16(out + (tok * CC + cc) * CH, ld16(mo));
    else {
      float a[CH], o[CH];
      unpack<T>(ld16(mo), a);
      unpack<T>(ld16(out + (tok * CC + cc) * CH), o);
#pragma unroll
      for (int j = 0; j < CH; ++j) o[j] += a[j];
      st16(out + (tok * CC + cc) * CH, pack<T>(o));
    }
  }
}
void launch_patch_merge(int dt, const void* in, void* out, int B, int H, int W, int C, int reverse, int beta, hipStream_t s) {
  DISPATCH_T(dt, {
    const long n = (long)B * H * W * (C / TT<T>::CH);
    hipLaunchKernelGGL((patch_merge_kernel<T>), dim3(grid_for(n)), dim3(256), 0, s, (const T*)in, (T*)out, H, W, C, reverse, beta, n);
  });
}

// ---- relative position bias: bias[h][i][j] = table[idx(i, j)][h], idx = (yi - yj + ws-1)*(2ws-1) + (xi - xj + ws-1) (:120-135,166-176)
__global__ void relpos_bias_kernel(const float* table, float* bias, int ws, int heads) {
  const int N = ws * ws;
  const long total = (long)heads * N * N;
  for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int j = (int)(t % N), i = (int)((t / N) % N), h = (int)(t / ((long)N * N));
    const int idx = (i / ws - j / ws + ws - 1) * (2 * ws - 1) + (i % ws - j % ws + ws - 1);
    bias[t] = table[(long)idx * heads + h];
  }
}
void launch_relpos_bias(const float* table, float* bias, int ws, int heads, hipStream_t s) {
  hipLaunchKernelGGL(relpos_bias_kernel, dim3(grid_for((long)heads * ws * ws * ws * ws)), dim3(256), 0, s, table, bias, ws, heads);
}
// dtable[e][h] += sum over the (i, j) pairs with idx(i, j) == e of dbias[h][i][ld*..]: one workgroup per table entry; wave w takes the heads
// w, w + 4, ..., its lanes the (at most ws*ws) query positions i whose partner j = i - (dy, dx) lies in the window, summed by a wave
// reduction in a fixed order -- no atomics, deterministic.  (The first form gave every HEAD a lane, 3 .. 24 active lanes each walking up to
// 144 strided loads: 25 us per launch for 276 K floats.)  dbias [heads][N][ld] (ld >= N).
__global__ __launch_bounds__(256) void relpos_bias_bwd_kernel(const float* dbias, float* dtable, int ws, int heads, int ld, float scale) {
  const int N = ws * ws, e = blockIdx.x;
  const int dy = e / (2 * ws - 1) - (ws - 1), dx = e % (2 * ws - 1) - (ws - 1);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int h = wv; h < heads; h += 4) {
    float a = 0.f;
    for (int i = lane; i < N; i += 64) {
      const int yi = i / ws, xi = i - yi * ws;
      const int yj = yi - dy, xj = xi - dx;
      if (yj >= 0 && yj < ws && xj >= 0 && xj < ws) a += dbias[((long)h * N + i) * ld + yj * ws + xj];
    }
    a = wave_sum(a);
    if (lane == 0) dtable[(long)e * heads + h] += a * scale;
  }
}
void launch_relpos_bias_bwd(const float* dbias, float* dtable, int ws, int heads, int ld, float scale, hipStream_t s) {
  hipLaunchKernelGGL(relpos_bias_bwd_kernel, dim3((2 * ws - 1) * (2 * ws - 1)), dim3(256), 0, s, dbias, dtable, ws, heads, ld, scale);
}

// ---- stochastic depth (timm DropPath): out = shortcut + branch * keep[b] / (1 - p), keep[b] drawn per SAMPLE.
// mode 0 forward (out = a + b * s), mode 1 backward of the branch (out = a * s; `b` unused)
template <typename T>
__global__ void droppath_kernel(const T* a, const T* b, T* out, long per_sample_chunks, long total_chunks, float p, const uint32_t* seedp,
                                uint32_t site, int mode, RowMap map, int CC) {
  constexpr int CH = TT<T>::CH;
  const uint32_t seed = p > 0.f ? *seedp : 0u;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total_chunks; i += (long)gridDim.x * blockDim.x) {
    const float sc = p > 0.f ? drop_scale(seed, site, (uint32_t)(i / per_sample_chunks), p) : 1.f;
    long iw = i;   // chunk index on the window-ordered (branch) side
    if (map.ws) { const long tok = i / CC; iw = rowmap_row(tok, map) * CC + (i - tok * CC); }
    float x[CH], y[CH];
    unpack<T>(ld16(a + i * CH), x);
    if (mode == 0) {
      unpack<T>(ld16(b + iw * CH), y);
#pragma unroll
      for (int j = 0; j < CH; ++j) x[j] += y[j] * sc;
      st16(out + i * CH, pack<T>(x));
    } else {
#pragma unroll
      for (int j = 0; j < CH; ++j) x[j] *= sc;
      st16(out + iw * CH, pack<T>(x));
    }
  }
}
void launch_droppath(int dt, int mode, const void* a, const void* b, void* out, int B, long per_sample, float p, const uint32_t* seed,
                     uint32_t site, hipStream_t s, RowMap map, int C) {
  DISPATCH_T(dt, {
    const long per = per_sample / TT<T>::CH, n = per * B;
    hipLaunchKernelGGL((droppath_kernel<T>), dim3(grid_for(n)), dim3(256), 0, s, (const T*)a, (const T*)b, (T*)out, per, n, p, seed, site, mode, map,
                       map.ws ? C / TT<T>::CH : 1);
  });
}
