// Row-streaming kernel for the TALL, THIN dense products of the fused-MBConv stages (bf16): C[M x N] = A[M x K] * W[N x K]^T with
// M = 24 576 .. 98 304 rows and N, K <= 256 -- the 1x1 projections of SURVEY Appendix B stages 1-2 (192 -> 48, 256 -> 64 channels at
// 32x96 / 16x48 maps; networks/EfficientSATRN.py:74,84 -> timm FusedMBConv conv_pwl) and their data gradients (48 -> 192, 64 -> 256).
// These launches move 40-85 MB each and did 1.1-1.2 TB/s on the tile kernel (gemm_kernel<64,64>: a workgroup's life was two dependent
// round trips around ~1 k-step of MFMA, the activation operand re-read once per 64-column tile, 2 N atomics per tile).  Here:
//   * the whole weight matrix (<= 36 KB) sits in LDS for the life of the workgroup, all N columns -- A is read once;
//   * every WAVE streams 16-row tiles on its own (persistent, tile = wave index + k * waves): its A fragments come straight from memory
//     into MFMA registers, the next tile's are requested while this tile's MFMAs run -- no workgroup barrier in the loop;
//   * the 16 x N tile leaves through a wave-private LDS image as whole 16-byte row chunks (full lines), where the epilogue operand of
//     the data-gradient form (the BatchNorm input y of the same chunk, requested one tile ahead) meets it;
//   * column sums (BatchNorm statistics of the output, or BatchNorm-backward sums of g = dx * act'(y scale + shift)) stay in registers
//     over all tiles of a wave and leave as ONE atomic per column and workgroup.
#include "common.h"
#include "kernels.h"
#include "tile_dev.h"

struct TallP {
  const bf16_t* A; const bf16_t* Bw; bf16_t* C;
  int M, N, K, lda, ldc;
  float* stats; int stats_rep;
  const bf16_t* bnb_y; const float* bnb_ss; const float* bnb_mr; int bnb_act;
};

// NCT = N / 16 column tiles, KS = ceil(K / 32) k-steps, BNB: 0 = output statistics (sum, sum of squares; optional), else BatchNorm-backward sums with
// the activation of that BatchNorm: 1 = ReLU, 2 = SiLU
template <int NCT, int KS, int BNB>
__global__ __launch_bounds__(256) void gemm_tall_kernel(TallP p) {
  typedef bf16_t T;
  constexpr int N = NCT * 16, KP = KS * 32 + 8, OP = N + 8;
  constexpr int CPR = N / 8, PPP = 64 / CPR, NP = (16 + PPP - 1) / PPP;   // 16-byte chunks per row, pixels per pass of a wave, passes per tile
  extern __shared__ __attribute__((aligned(16))) unsigned char tl_sm[];
  bf16_t* Bl = reinterpret_cast<bf16_t*>(tl_sm);                    // [N][KP]
  bf16_t* ow_all = Bl + N * KP;                                     // [4 waves][16][OP]
  float* cfl = reinterpret_cast<float*>(ow_all + 4 * 16 * OP);      // BNB: scale | shift | mean | rstd [4][N]
  float* wsum = reinterpret_cast<float*>(ow_all);                   // after the loop: the waves' column sums [4][2][N] over the output images
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
  // ---- the weights (zero beyond K), the BatchNorm coefficients
  // the first tile's operands are requested before the weights: one memory round trip covers the whole prologue
  bf16_t* ow = ow_all + wave * 16 * OP;
  const int ntile = p.M / 16, gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
  const int cp = lane / CPR, cc = lane - cp * CPR;
  const bool lane_on = lane < PPP * CPR;
  float s1[8], s2[8];          // BNB: this lane's chunk column, over all its tiles
#pragma unroll
  for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
  float t1[NCT], t2[NCT];      // output statistics: column fr of every column tile, from the f32 accumulators (as gemm_kernel's epilogue)
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) t1[ct] = t2[ct] = 0.f;
  Frag<T> af[KS];
  uint4 yq[NP];
  auto load_tile = [&](int t) {
    const bf16_t* ar = p.A + (size_t)(t * 16 + fr) * p.lda;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) af[ks].v = (ks * 32 + fq * 8 < p.K) ? ld16(ar + ks * 32 + fq * 8) : zero16();
  };
  auto load_y = [&](int t, int k) {
    const int pix = k * PPP + cp;
    yq[k] = (lane_on && pix < 16) ? ld16(p.bnb_y + (size_t)(t * 16 + pix) * N + cc * 8) : zero16();
  };
  int t = gw;
  if (t < ntile) {
    load_tile(t);
    if (BNB) {
#pragma unroll
      for (int k = 0; k < NP; ++k) load_y(t, k);
    }
  }
  // (all requests of a thread first, then the LDS stores: a load -> store loop is one dependent memory round trip per iteration)
  {
    constexpr int NCH = N * (KP / 8), WPT = (NCH + 255) / 256, CPT = (2 * N + 255) / 256;
    uint4 wr[WPT];
    float c0[CPT], c1[CPT];
#pragma unroll
    for (int k = 0; k < WPT; ++k) {
      const int i = tid + k * 256, n = i / (KP / 8), c = i - n * (KP / 8);
      wr[k] = (i < NCH && c * 8 < p.K) ? ld16(p.Bw + (size_t)n * p.K + c * 8) : zero16();
    }
    if (BNB) {
#pragma unroll
      for (int k = 0; k < CPT; ++k) { const int i = tid + k * 256; c0[k] = i < 2 * N ? p.bnb_ss[i] : 0.f; c1[k] = i < 2 * N ? p.bnb_mr[i] : 0.f; }
    }
#pragma unroll
    for (int k = 0; k < WPT; ++k) {
      const int i = tid + k * 256, n = i / (KP / 8), c = i - n * (KP / 8);
      if (i < NCH) st16(Bl + n * KP + c * 8, wr[k]);
    }
    if (BNB) {
#pragma unroll
      for (int k = 0; k < CPT; ++k) { const int i = tid + k * 256; if (i < 2 * N) { cfl[i] = c0[k]; cfl[2 * N + i] = c1[k]; } }
    }
  }
  __syncthreads();
  float sc[8], sh[8], mu[8], rs[8];
  if (BNB && lane_on) { lds8(cfl + cc * 8, sc); lds8(cfl + N + cc * 8, sh); lds8(cfl + 2 * N + cc * 8, mu); lds8(cfl + 3 * N + cc * 8, rs); }
  for (; t < ntile; t += nw) {
    f32x4 acc[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        Frag<T> b;
        b.v = ld16(Bl + (ct * 16 + fr) * KP + ks * 32 + fq * 8);
        mma(af[ks], b, acc[ct]);
      }
    }
    const int tn = t + nw;
    if (tn < ntile) load_tile(tn);   // in flight under the epilogue and the other waves' work
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ow[(fq * 4 + r) * OP + ct * 16 + fr] = from_f<T>(acc[ct][r]);
        if (!BNB) { t1[ct] += acc[ct][r]; t2[ct] += acc[ct][r] * acc[ct][r]; }
      }
    // the tile as 16-byte row chunks: pass k covers pixels k * PPP .. + PPP
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const int pix = k * PPP + cp;
      if (lane_on && pix < 16) {
        const uint4 q = ld16(ow + pix * OP + cc * 8);
        st16(p.C + (size_t)(t * 16 + pix) * p.ldc + cc * 8, q);
        if (BNB) {
          float v[8], y[8];
          unpack<T>(q, v);
          unpack<T>(yq[k], y);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float g = v[j] * act_bwd(y[j] * sc[j] + sh[j], BNB == 1 ? ACT_RELU : ACT_SILU);   // (compile-time kind: a run-time switch per element made this a 10 000-instruction kernel)
            s1[j] += g; s2[j] += g * ((y[j] - mu[j]) * rs[j]);
          }
        }
      }
      if (BNB && tn < ntile) load_y(tn, k);
    }
  }
  // ---- column sums: lanes with the same chunk column (cp = 0 .. PPP - 1) through the wave's LDS row, the waves through LDS, one atomic per column
  if (p.stats) {
    __syncthreads();   // every wave is done with its output image: the sums overlay it
    float* mine = wsum + wave * 2 * N;
    if (BNB) {
      for (int i = lane; i < 2 * N; i += 64) mine[i] = 0.f;
      // (wave-private: the LDS operations of one wave execute in order)
      for (int pp = 0; pp < PPP; ++pp) {
        if (lane_on && cp == pp) {
#pragma unroll
          for (int j = 0; j < 8; ++j) { mine[cc * 8 + j] += s1[j]; mine[N + cc * 8 + j] += s2[j]; }
        }
      }
    } else {
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        float a = t1[ct], b = t2[ct];
        a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
        b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
        if (fq == 0) { mine[ct * 16 + fr] = a; mine[N + ct * 16 + fr] = b; }
      }
    }
    __syncthreads();
    for (int i = tid; i < 2 * N; i += 256) {
      const float tot = (wsum[i] + wsum[2 * N + i]) + (wsum[4 * N + i] + wsum[6 * N + i]);
      atomicAdd(p.stats + (size_t)(blockIdx.x % p.stats_rep) * 2 * N + i, tot);
    }
  }
}

template <int NCT, int KS, int BNB>
static bool tall_go(const TallP& p, hipStream_t s) {
  constexpr int N = NCT * 16, KP = KS * 32 + 8, OP = N + 8;
  const size_t lds = (size_t)N * KP * 2 + (size_t)4 * 16 * OP * 2 + (size_t)(4 * N) * 4;   // (4 * 16 * OP * 2 >= 8 * N * 4: the sums fit the images)
  static bool attr = false;
  static int per_cu = 0;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm_tall_kernel<NCT, KS, BNB>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)gemm_tall_kernel<NCT, KS, BNB>, 256, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    attr = true;
  }
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  }
  const int ntile = p.M / 16;
  int grid = cus * per_cu;
  if (sw_knob("tall_grid", 0) > 0) grid = (int)sw_knob("tall_grid", 0);   // knob (tools/gemm_tall_bench.py)
  if (grid * 4 > ntile) grid = (ntile + 3) / 4;
  hipLaunchKernelGGL((gemm_tall_kernel<NCT, KS, BNB>), dim3(grid), dim3(256), lds, s, p);
  return true;
}

// true = launched.  Dense bf16 products with many rows and N, K <= 256: plain, with output statistics, or the data-gradient form with
// BatchNorm-backward sums.  Everything else (bias, activation, dropout, accumulate, inference epilogues, f32 output) stays on gemm_kernel.
bool gemm_tall_launch(const GemmP& g, hipStream_t s) {
  const char* mode_env = sw_knob_str("gemm_tall");   // read per call (tests, A/B): 0 = off, 2 = every shape that fits
  const int mode = mode_env ? atoi(mode_env) : 1;
  if (!mode) return false;
  if (g.bias || g.act || g.drop_p > 0.f || g.beta || g.out_f32 || g.escale || g.eres || g.pre_out || g.bact_u || g.stats_part) return false;
  if ((g.M & 15) || g.lda < g.K || (g.lda & 7) || (g.ldc & 7) || g.ldc < g.N || (g.K & 7)) return false;
  if (g.bnb_y && (!g.stats || g.ldc != g.N)) return false;
  if (g.stats && g.stats_rep < 1) return false;
  if (mode != 2 && g.M < 16384) return false;
  TallP p;
  p.A = (const bf16_t*)g.A; p.Bw = (const bf16_t*)g.Bw; p.C = (bf16_t*)g.C; p.M = g.M; p.N = g.N; p.K = g.K; p.lda = g.lda; p.ldc = g.ldc;
  p.stats = g.stats; p.stats_rep = g.stats ? g.stats_rep : 1; p.bnb_y = (const bf16_t*)g.bnb_y; p.bnb_ss = g.bnb_ss; p.bnb_mr = g.bnb_mr; p.bnb_act = g.bnb_act;
  const int ks = (g.K + 31) / 32;
  bool ok = false;
  if (!g.bnb_y) {
    // projections: N = 48 / 64 (K = 192 / 256), conv_last-like shapes are not tall
    if (g.N == 48 && ks == 6) ok = tall_go<3, 6, 0>(p, s);
    else if (g.N == 64 && ks == 8) ok = tall_go<4, 8, 0>(p, s);
    else if (g.N == 64 && ks == 6) ok = tall_go<4, 6, 0>(p, s);
    else if (g.N == 48 && ks == 8) ok = tall_go<3, 8, 0>(p, s);
  } else if (g.bnb_act == ACT_SILU) {
    if (g.N == 192 && ks == 2) ok = tall_go<12, 2, 2>(p, s);
    else if (g.N == 256 && ks == 2) ok = tall_go<16, 2, 2>(p, s);
    else if (g.N == 192 && ks == 1) ok = tall_go<12, 1, 2>(p, s);
  } else if (g.bnb_act == ACT_RELU) {
    if (g.N == 192 && ks == 2) ok = tall_go<12, 2, 1>(p, s);
    else if (g.N == 256 && ks == 2) ok = tall_go<16, 2, 1>(p, s);
  }
  if (ok) g_route[RT_GEMM_TALL]++;
  return ok;
}
