// Large dense bf16 GEMM for gfx950:  C[M,N] = epilogue(A[M,K] * W[N,K]^T)   (both operands K-contiguous)
//
// The 4-wave register-staged kernel of kernels_gemm.hip tops out below 20 % of the MFMA peak on the big products of SwinTRN
// (M = 9 216 .. 147 456, K = 96 .. 2 048) and the early backbone stages: every k-step is global load -> VGPR -> ds_write ->
// barrier -> ds_read -> MFMA by the same four waves, and every tile pays its own prologue and store tail.  This kernel is
// the chip-filling form:
//   * PERSISTENT: one 512-thread workgroup per compute unit walks a list of output tiles (BM x 128, BM = 128 / 192 / 256 chosen per
//     launch: fewest rounds x rows on the grid); the (tile, k-step) pairs of a workgroup form ONE stream, so the operand loads of the next
//     tile are in flight while the current tile is multiplied and stored (K = 96 .. 512 means only 2 .. 8 k-steps per tile).
//   * PRODUCER / CONSUMER WAVES: waves 4-7 only issue loads, waves 0-3 (2 along M x 2 along N, wave tile BM/2 x 64) only multiply and
//     store.  The loaders count their own DMA operations (`s_waitcnt vmcnt(N)` with N = the operations of the stages allowed to stay in
//     flight); the two groups meet at ONE raw s_barrier per k-step.
//   * DIRECT-TO-LDS: operands go global -> LDS with `buffer_load_dwordx4 ... offen lds` (no VGPRs, no ds_write); three 64-deep
//     stages of (BM + 128) rows x 128 B form a ring.  Out-of-range rows (M / N tails) and the K tail are zero-filled by the buffer
//     descriptor's range check.
//   * LDS image: row-major 128-byte rows, 16-byte chunk c of row r stored at chunk position c ^ ((r >> 1) & 7): the DMA writes
//     lane-linear (8 lanes = one row), so the swizzle is applied to each lane's SOURCE address and again on the fragment
//     reads -- ds_read_b128 of an MFMA fragment (16 rows x one chunk per 16-lane group) is then bank-conflict free.
//   * fragments are double-buffered in registers across the barrier: the ds_reads of half a k-step run under the MFMAs of the other half.
//   * the MFMA is issued with the operands swapped (D' = W_frag x A_frag^T), so a lane holds FOUR CONSECUTIVE output columns of
//     one row; the tile goes through a per-wave LDS scratch (2 x 16 rows x 128 B, XOR-swizzled) and leaves as whole 128-byte row
//     segments, 16 bytes per lane (the 4-wave kernel stores one 2-byte element per lane; 8-byte stores straight from the accumulator
//     layout were measured too: 10-25 % slower than the LDS hop).
// Epilogue: bias (the initial accumulator), activation, BatchNorm statistics / BatchNorm-backward sums (STATS), and by epilogue kind EK
// a second tensor: act'(u) stored beside the activation, a stored factor multiplied in, accumulate.  Dropout and f32 output
// stay with gemm_kernel; gemm_big_launch() returns false for those.
// Measured limits: the main loop runs at the LDS fill rate, 63 GB/s per CU alone and 48 GB/s per CU with all 256 CUs
// busy (tools/gemm_big_grid.sh), i.e. ~50 % of the MFMA rate of its tile; LDS reads (each operand row is read by two consumer waves)
// plus DMA writes are 120 KB per k-step against 128 B/clk.
// Reference shapes: networks/SWIN.py:84-209 (qkv / proj), :24-47 (Mlp fc1 / fc2), networks/EfficientSATRN.py:66-87 (1x1 convs).
#include <stdio.h>
#include <algorithm>
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

struct BigP {
  const bf16_t* A; const bf16_t* W; bf16_t* C;
  const float* bias;
  bf16_t* pre_out; const bf16_t* bact_u;
  const float* escale; const float* eshift; const bf16_t* eres;   // EK 4 (inference)
  int M, N, K, lda, ldc;
  int act, bact, beta, pre_grad;
  float bact_scale;
  int ntm, ntn, ntiles;
  unsigned a_bytes, w_bytes, c_bytes;   // buffer extents for the range check
  // STATS 1: stats[(tile_m % stats_rep)][2][N] += per-column sum / sum of squares of the stored output (the BatchNorm that follows);
  // STATS 2: the same slots receive the BatchNorm-backward sums [sum g, sum g * xhat], g = out * act'(y * scale + shift)  (GemmP::bnb_*)
  float* stats; int stats_rep;
  const bf16_t* bnb_y; const float* bnb_ss; const float* bnb_mr; int bnb_act; unsigned y_bytes;
  int dbg;   // timing experiments (SATRN_BIG_DBG; wrong results): 1 no epilogue, 2 no DMA waits, 4 no MFMA, 8 no fragment reads
  // 3x3 stride-1 'same' convolution over NHWC as a shifted GEMM (conv != 0): row r of A is output pixel r, k-step s = tap s / kpt (+ 64-channel
  // slice s % kpt): the loaders read pixel (y + dy, x + dx) of the same image -- or nothing (zero fill) outside the image / past Ci --
  // and the tap's Ci columns of the packed weights [N][9][Ci]; the consumers see ordinary 64-deep stages.  flip: the data gradient
  // (dy -> -dy, dx -> -dx; weights = the backward pack [Ci_out][9][Co]).
  int conv, cH, cW, cCi, flip, kpt;
  int cOH, cOW, cstride, cpt, cpl;   // rows run over [B][cOH][cOW]; the source tensor is [B][cH][cW][cCi]; stride 1 or 2, top / left padding
  int narrow;   // 64-column tiles (WN = 1)
};

#define BIG_THREADS 512
#define BIG_BK 64
#define BIG_ROWB 128            // bytes per LDS row (64 bf16)
#define BIG_NSTAGE 3

typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
DEVI uint4 lds_read16(unsigned addr) {
  const u32x4_t v = *reinterpret_cast<const __attribute__((address_space(3))) u32x4_t*>((lds_u8*)(size_t)addr);
  return make_uint4(v.x, v.y, v.z, v.w);
}
DEVI void lds_write8(unsigned addr, uint2 v) {
  u32x2_t w; w.x = v.x; w.y = v.y;
  *reinterpret_cast<__attribute__((address_space(3))) u32x2_t*>((lds_u8*)(size_t)addr) = w;
}

DEVI u32x4_t to_u32x4(uint4 v) { u32x4_t r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w; return r; }
DEVI uint4 from_u32x4(u32x4_t v) { return make_uint4(v.x, v.y, v.z, v.w); }

// One LDS-DMA piece: 64 lanes x 16 bytes from buffer `rsrc` (per-lane byte offset voff, wave-uniform offset soff) to the 1 KiB of LDS
// at `lds_addr` (wave-uniform), lane-linear.  Issued as inline asm on purpose: the compiler does not see a memory operation, so
// it neither counts it in its own s_waitcnt bookkeeping (it would wait vmcnt(0) before LDS reads it cannot prove independent)
// nor reorders it -- the kernel counts these operations itself (vm_wait_n).  M0 = LDS base, saved and restored (the compiler owns M0).
typedef int i32x4 __attribute__((ext_vector_type(4)));
DEVI void dma16(i32x4 rsrc, unsigned lds_addr, unsigned voff, int soff) {
  unsigned keep;
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
DEVI i32x4 make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long b = (unsigned long long)(size_t)base;
  i32x4 r;
  r.x = (int)(unsigned)b; r.y = (int)(unsigned)(b >> 32) & 0xffff; r.z = (int)bytes; r.w = 0x00020000;
  return r;
}

// MTW: 16-row MFMA tiles per CONSUMER wave along M.  Block tile = (32 * MTW) rows x 128 columns: consumer waves 0..3 form a 2 x 2 grid
// of (16 * MTW) x 64 wave tiles; waves 4..7 are LOADERS and only issue the LDS-DMA stream.
//
// Why two roles (measured on the first, symmetric version of this kernel, tools/gemm_big_dbg.py, 4096^3): the DMA stream alone ran
// in 88 us (70 GB/s per CU: the per-CU fill path), the MFMAs alone in ~55 us, and the kernel took their SUM, 155 us -- a wave whose
// next instruction is an LDS-DMA waits at the issue stage while the memory pipeline's queue is full, and the MFMAs behind it in
// program order wait with it.  A loader wave may sit there as long as it likes; the consumer waves never issue a DMA.
// Synchronisation: ONE s_barrier per k-step, joined by all eight waves.  A loader arrives when its pieces of stage t + 1 have
// landed (counted vmcnt), a consumer when its last fragment read of stage t has returned; behind the barrier the loaders refill
// the slot of stage t with stage t + 3 and the consumers start reading stage t + 1.
// EK, the epilogue kind, is a template parameter because the epilogue is unrolled over the wave tile's 2 * MTW row groups: every
// run-time branch in it is replicated 2 * MTW times, and the all-purpose form (EK 1: any activation, any derivative, accumulate) is an
// 34 K-instruction kernel whose epilogue ran 3x slower than its arithmetic (SwinTRN fc1 + GELU 73 us, fc2's data gradient 90 us
// against 22-25 us for the plain product).  The two forms the training step uses all the time get lean code of their own:
//   EK 0  no second tensor (activation by run-time switch, GELU through gelu_fast)
//   EK 1  all-purpose: pre_out / bact_u / beta in any combination
//   EK 2  GELU + its derivative stored beside it (pre_out with pre_grad; networks/SWIN.py:24-47 fc1)
//   EK 4  inference: per-column scale / shift (eval-mode BatchNorm), activation, residual -- the encoder of the greedy decode
//   EK 3  times a stored factor: bact_u with ACT_DFACTOR (the stored derivative) or ACT_RELU (sign of the stored output, x bact_scale);
//         the factor loads run four row groups ahead of their use (one dependent round trip per group otherwise)
// CONV (the shifted-GEMM convolution mode, BigP::conv) is a template parameter as well: as a run-time branch in the loaders it cost the
// dense products 5 % (the loaders' issue rate is what bounds the main loop).
// WN = consumer waves along N: 2 = the 128-column tile (consumers 2 x 2, BM = 32 MTW), 1 = a 64-column tile for narrow outputs (consumers 4 x 1,
// BM = 64 MTW: the data gradients of the 3 x 3 convolutions have 48 / 64 output channels, on the 128-column tile most of the MFMA and LDS
// work multiplied zeros).
template <int MTW, bool HAS_BIAS, int EK, int STATS = 0 /*1 BatchNorm statistics, 2 BatchNorm-backward sums*/, bool CONV = false, int WN = 2>
__global__ __launch_bounds__(BIG_THREADS, 2) void gemm_big_kernel(BigP p) {
  constexpr bool HAS_AUX = EK == 1;
  constexpr int NT = 4, HM = MTW / 2;
  constexpr int BM = 16 * MTW * (4 / WN), BN = 64 * WN;
  constexpr int TM = 16 * MTW;                                    // rows per consumer wave
  constexpr int STAGE = (BM + BN) * BIG_ROWB;                     // bytes per ring slot
  constexpr int EPI = BIG_NSTAGE * STAGE;                         // epilogue scratch: 4 consumer waves x 2 x 2 KB
  constexpr int NA = BM / 32, NB = BN / 32;                       // DMA instructions per LOADER wave and stage (8 rows each)
  extern __shared__ __attribute__((aligned(16))) unsigned char big_sm[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)big_sm;   // LDS byte address of the ring

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int KT = CONV ? 9 * p.kpt : (p.K + BIG_BK - 1) / BIG_BK;   // K % 8 == 0; a partial last k-step is zero-filled (both operands) by the loaders

  // ---- this workgroup's tiles: logical id L (XCD-contiguous) + i * gridDim; tiles ordered n fastest, so the 32 workgroups of an
  // XCD share two row blocks of A and all of W in their L2
  const int nwg = gridDim.x;
  const int L = xcd_remap(blockIdx.x, nwg);
  const int my_tiles = L < p.ntiles ? (p.ntiles - L + nwg - 1) / nwg : 0;
  const int T = my_tiles * KT;     // k-steps of this workgroup's stream
  if (T == 0) return;

  if (wave >= 4) {
    // =============================== loader waves ===============================
    const int lw = wave - 4;
    const i32x4 rA = make_rsrc(p.A, p.a_bytes), rW = make_rsrc(p.W, p.w_bytes);
    // lane -> (row within the 8-row piece, chunk slot); source chunk = slot ^ swizzle(row); piece i = lw + 4 j covers rows 8 i .. 8 i + 7
    const int drow = lane >> 3;
    const int dsw = ((lane >> 4) | ((lw & 1) << 2));                 // ((8 i + drow) >> 1) & 7
    const unsigned dchunk = (unsigned)(((lane & 7) ^ dsw) * 16);
    const unsigned lda2 = (unsigned)p.lda * 2u, ldw2 = (unsigned)p.K * 2u;
    int d_tile = 0, d_k = 0;                 // position of the NEXT stage to request
    unsigned voffA = 0, voffW = 0;
    // convolution mode: per piece of this lane the pixel's (y, x) and its byte offset (row r = pixel r of the NHWC map)
    int cy[CONV ? NA : 1], cx[CONV ? NA : 1];
    unsigned cbase[CONV ? NA : 1];
    const int cch = (int)(dchunk >> 4);      // this lane's 8-channel chunk inside a 64-channel slice
    auto dma_tile_setup = [&](int ti) {
      const int tile = L + ti * nwg;
      const int tm = tile / p.ntn, tn = tile - tm * p.ntn;
      voffA = (unsigned)(tm * BM + 8 * lw + drow) * lda2 + dchunk;
      voffW = (unsigned)(tn * BN + 8 * lw + drow) * ldw2 + dchunk;
      if constexpr (CONV) {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
          const int r = tm * BM + 8 * (lw + 4 * j) + drow;
          const int ohw = p.cOH * p.cOW, b = r / ohw, rem = r - b * ohw;
          cy[j] = r < p.M ? rem / p.cOW : -(1 << 20);  // rows past M: never inside the image
          cx[j] = rem % p.cOW;
          // stride 1: the row's own pixel (a tap then adds a constant); stride 2: the image's first source pixel
          cbase[j] = (p.cstride == 1 ? (unsigned)r : (unsigned)(b * p.cH * p.cW)) * (unsigned)(p.cCi * 2) + dchunk;
        }
      }
    };
    // K tail: in the last k-step of a tile the lanes whose 16-byte chunk lies past K get an out-of-range offset (zero fill)
    const bool tail_dead = (KT - 1) * BIG_BK + (int)(dchunk >> 1) >= p.K;
    auto dma_issue = [&](int slot) {
      const unsigned sbase = lds0 + (unsigned)slot * STAGE;
      if constexpr (CONV) {
        const int tap = d_k / p.kpt, kk = d_k - tap * p.kpt;
        const int kh = tap / 3, kw = tap - kh * 3;
        const bool cdead = kk * 64 + cch * 8 >= p.cCi;                                  // past the tap's channels (Ci not a multiple of 64)
        const int woff = (tap * p.cCi + kk * 64) * 2;
        const int sh1 = p.cstride - 1;   // stride 1 or 2
        if (p.cstride == 1) {
          // (the common case kept lean -- the loaders' issue rate bounds the main loop: the general form below cost the stride-1 data
          // gradients 20 %)
          const int dy = p.flip ? p.cpt - kh : kh - p.cpt, dx = p.flip ? p.cpl - kw : kw - p.cpl;
          const int aoff = ((dy * p.cW + dx) * p.cCi + kk * 64) * 2;
#pragma unroll
          for (int j = 0; j < NA; ++j) {
            const int yy = cy[j] + dy, xx = cx[j] + dx;
            const bool ok = !cdead && yy >= 0 && yy < p.cH && xx >= 0 && xx < p.cW;
            dma16(rA, sbase + (unsigned)(lw + 4 * j) * 1024u, ok ? cbase[j] + (unsigned)aoff : 0xfffffff0u, 0);
          }
        } else
#pragma unroll
        for (int j = 0; j < NA; ++j) {
          // source pixel of output pixel (cy, cx) under this tap: forward (oy * stride - pt + kh, ..); data gradient ((oy + pt - kh) / stride, ..)
          // where that division is exact
          int sy, sx;
          bool ok = !cdead;
          if (!p.flip) { sy = (cy[j] << sh1) - p.cpt + kh; sx = (cx[j] << sh1) - p.cpl + kw; }
          else {
            const int ty = cy[j] + p.cpt - kh, tx = cx[j] + p.cpl - kw;
            ok = ok && ((ty | tx) & sh1) == 0;
            sy = ty >> sh1; sx = tx >> sh1;
          }
          ok = ok && sy >= 0 && sy < p.cH && sx >= 0 && sx < p.cW;
          dma16(rA, sbase + (unsigned)(lw + 4 * j) * 1024u, ok ? cbase[j] + (unsigned)(((sy * p.cW + sx) * p.cCi + kk * 64) * 2) : 0xfffffff0u, 0);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j)
          dma16(rW, sbase + (unsigned)BM * BIG_ROWB + (unsigned)(lw + 4 * j) * 1024u, cdead ? 0xfffffff0u : voffW + (unsigned)j * 32u * ldw2 + (unsigned)woff, 0);
        if (++d_k == KT) { d_k = 0; ++d_tile; if (d_tile < my_tiles) dma_tile_setup(d_tile); }
        return;
      }
      const int soff = d_k * (BIG_BK * 2);
      const bool dead = tail_dead && d_k == KT - 1;
#pragma unroll
      for (int j = 0; j < NA; ++j) dma16(rA, sbase + (unsigned)(lw + 4 * j) * 1024u, dead ? 0xfffffff0u : voffA + (unsigned)j * 32u * lda2, soff);
#pragma unroll
      for (int j = 0; j < NB; ++j) dma16(rW, sbase + (unsigned)BM * BIG_ROWB + (unsigned)(lw + 4 * j) * 1024u, dead ? 0xfffffff0u : voffW + (unsigned)j * 32u * ldw2, soff);
      if (++d_k == KT) { d_k = 0; ++d_tile; if (d_tile < my_tiles) dma_tile_setup(d_tile); }
    };
    dma_tile_setup(0);
    dma_issue(0);
    if (1 < T) dma_issue(1);
    if (2 < T) dma_issue(2);
    // stage 0 landed: all but the stages requested after it
    if (T >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (NA + NB)) : "memory");
    else if (T == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA + NB) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int slot = 0;
    for (int t = 0; t < T; ++t) {
      // stage t + 1 landed (stage t + 2 may stay in flight)
      if (t + 2 < T) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA + NB) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (t + BIG_NSTAGE < T) dma_issue(slot);
      slot = slot == 2 ? 0 : slot + 1;
    }
    return;
  }

  // =============================== consumer waves ===============================
  const int wm = WN == 2 ? wave >> 1 : wave, wn = WN == 2 ? (wave & 1) : 0;
  const int fr = lane & 15, fq = lane >> 4;
  // outputs / epilogue operands through buffer descriptors: a lane outside the matrix gets an out-of-range offset and the access is
  // dropped by the range check (no divergent branch around the stores)
  const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc((void*)p.C, 0, (int)p.c_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc((void*)p.pre_out, 0, p.pre_out ? (int)p.c_bytes : 0, 0x00020000);
  const bf16_t* second = EK == 4 ? p.eres : p.bact_u;   // the tensor read beside the output: stored factor, or the inference residual
  const __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc((void*)second, 0, second ? (int)p.c_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc((void*)p.bnb_y, 0, STATS == 2 ? (int)p.y_bytes : 0, 0x00020000);

  const unsigned fsw0 = (unsigned)(((0 * 4 + fq) ^ ((fr >> 1) & 7)) * 16), fsw1 = (unsigned)(((1 * 4 + fq) ^ ((fr >> 1) & 7)) * 16);
  const unsigned fA = (unsigned)(wm * TM + fr) * BIG_ROWB;
  const unsigned fB = (unsigned)BM * BIG_ROWB + (unsigned)(wn * 64 + fr) * BIG_ROWB;
  // fragments: two ping-pong sets of HM A fragments (one MFMA group = HM x 4 tiles of one 32-deep half), B fragments of both halves
  uint4 aX[HM], aY[HM], bK0[NT], bK1[NT];
  auto read_a = [&](int slot, unsigned fsw, int half, uint4* af) {
    if (p.dbg & 8) return;
    const unsigned sbase = lds0 + (unsigned)slot * STAGE + fsw + fA + (unsigned)(half * HM) * 16u * BIG_ROWB;
#pragma unroll
    for (int i = 0; i < HM; ++i) af[i] = lds_read16(sbase + (unsigned)i * 16u * BIG_ROWB);
  };
  auto read_b = [&](int slot, unsigned fsw, uint4* bf) {
    if (p.dbg & 8) return;
    const unsigned sbase = lds0 + (unsigned)slot * STAGE + fsw + fB;
#pragma unroll
    for (int j = 0; j < NT; ++j) bf[j] = lds_read16(sbase + (unsigned)j * 16u * BIG_ROWB);
  };
  f32x4 acc[MTW][NT];
  auto mma_group = [&](int half, const uint4* af, const uint4* bf) {
    if (p.dbg & 4) return;
#pragma unroll
    for (int i = 0; i < HM; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        acc[half * HM + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bf[j]), __builtin_bit_cast(bf16x8, af[i]), acc[half * HM + i][j], 0, 0, 0);
  };

  // ---- bias of a tile in accumulator layout: lane holds columns nt * 16 + fq * 4 + r of its wave's 64
  f32x4 bias_r[NT];
  auto bias_load = [&](int ti) {
    const int tile = L + ti * nwg;
    const int tn = tile % p.ntn;
    const int nb = tn * BN + wn * 64 + fq * 4;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = nb + j * 16;
      bias_r[j] = *reinterpret_cast<const f32x4*>(p.bias + (n + 3 < p.N ? n : 0));   // (columns past N are never stored)
    }
  };
  auto acc_init = [&]() {
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = HAS_BIAS ? bias_r[j] : f32x4{0.f, 0.f, 0.f, 0.f};
  };

  // ---- epilogue of one finished tile: 16 rows x 64 columns per pass through this wave's 2 KB of LDS (XOR-swizzled 128-byte rows)
  // (two 2 KB buffers per consumer wave, alternated per pass: the writes of pass i + 1 do not wait for the reads of pass i)
  const unsigned eW0 = lds0 + EPI + (unsigned)wave * 4096u;
  const int e_row = lane >> 3, e_chunk = (lane & 7) ^ ((lane >> 3) & 7);
  auto epilogue = [&](int ti) {
    const int tile = L + ti * nwg;
    const int tm = tile / p.ntn, tn = tile - tm * p.ntn;
    const int col = tn * BN + wn * 64 + e_chunk * 8;
    // column sums of this lane's eight columns over the tile's rows (STATS): a lane keeps the same columns in every pass
    float s1[8], s2[8], bsc[8], bsh[8], bmu[8], brs[8];
    if (STATS) {
#pragma unroll
      for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
    }
    if (STATS == 2) {
      const int cc = col + 7 < p.N ? col : 0;   // (columns past N contribute nothing: masked below)
      const f32x4 t0 = *reinterpret_cast<const f32x4*>(p.bnb_ss + cc), t1 = *reinterpret_cast<const f32x4*>(p.bnb_ss + cc + 4);
      const f32x4 t2 = *reinterpret_cast<const f32x4*>(p.bnb_ss + p.N + cc), t3 = *reinterpret_cast<const f32x4*>(p.bnb_ss + p.N + cc + 4);
      const f32x4 t4 = *reinterpret_cast<const f32x4*>(p.bnb_mr + cc), t5 = *reinterpret_cast<const f32x4*>(p.bnb_mr + cc + 4);
      const f32x4 t6 = *reinterpret_cast<const f32x4*>(p.bnb_mr + p.N + cc), t7 = *reinterpret_cast<const f32x4*>(p.bnb_mr + p.N + cc + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { bsc[e] = t0[e]; bsc[4 + e] = t1[e]; bsh[e] = t2[e]; bsh[4 + e] = t3[e]; bmu[e] = t4[e]; bmu[4 + e] = t5[e]; brs[e] = t6[e]; brs[4 + e] = t7[e]; }
    }
    auto out_off = [&](int n) {   // byte offset of this lane's 16 bytes in row group n = 2 i + h (8 rows each); out of range: dropped
      const int row = tm * BM + wm * TM + n * 8 + e_row;
      return (row < p.M && col < p.N) ? (unsigned)(((long)row * p.ldc + col) * 2) : 0xfffffff0u;
    };
    constexpr int UQ = 4;
    u32x4_t uq[(EK == 3 || EK == 4) ? UQ : 1];
    if constexpr (EK == 3 || EK == 4) {   // (EK 4 without a residual: the resource has no extent, the loads return zeros)
#pragma unroll
      for (int n = 0; n < UQ; ++n) uq[n] = __builtin_amdgcn_raw_buffer_load_b128(rU, (int)out_off(n), 0, 0);
    }
    float esc8[8], esh8[8];
    if constexpr (EK == 4) {
      const int cc = col + 7 < p.N ? col : 0;   // (columns past N are never stored)
      const f32x4 t0 = *reinterpret_cast<const f32x4*>(p.escale + cc), t1 = *reinterpret_cast<const f32x4*>(p.escale + cc + 4);
      const f32x4 t2 = *reinterpret_cast<const f32x4*>(p.eshift + cc), t3 = *reinterpret_cast<const f32x4*>(p.eshift + cc + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { esc8[e] = t0[e]; esc8[4 + e] = t1[e]; esh8[e] = t2[e]; esh8[4 + e] = t3[e]; }
    }
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
      const unsigned eW = eW0 + (unsigned)(i & 1) * 2048u;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const uint2 w2 = make_uint2(pack2bf(acc[i][j][0], acc[i][j][1]), pack2bf(acc[i][j][2], acc[i][j][3]));
        lds_write8(eW + (unsigned)fr * 128u + (unsigned)(((j * 2 + (fq >> 1)) ^ (fr & 7)) * 16) + (unsigned)(fq & 1) * 8u, w2);
      }
      // (same wave writes and reads its own scratch: LDS operations of a wave execute in order, no barrier)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int rl = e_row + 8 * h;
        uint4 v = lds_read16(eW + (unsigned)rl * 128u + (unsigned)(lane & 7) * 16u);
        const int row = tm * BM + wm * TM + i * 16 + rl;
        const unsigned o = (row < p.M && col < p.N) ? (unsigned)(((long)row * p.ldc + col) * 2) : 0xfffffff0u;   // out of range: dropped
        if constexpr (EK == 2) {
          // GELU and its derivative from one erf / exp evaluation; the derivative is what the backward keeps
          float f[8], d[8];
          unpack<bf16_t>(v, f);
#pragma unroll
          for (int e = 0; e < 8; e += 2) {   // packed f32 math, two elements per issue slot (no measurable change: the form is not VALU-bound)
            f32x2 u2, g2, d2;
            u2.x = f[e]; u2.y = f[e + 1];
            gelu_pair(u2, &g2, &d2);
            f[e] = g2.x; f[e + 1] = g2.y; d[e] = d2.x; d[e + 1] = d2.y;
          }
          __builtin_amdgcn_raw_buffer_store_b128(to_u32x4(pack<bf16_t>(d)), rP, (int)o, 0, 0);
          v = pack<bf16_t>(f);
        } else if constexpr (EK == 3) {
          float f[8], u[8];   // (the row group index 2 * i + h is a compile-time constant under the unrolling)
          unpack<bf16_t>(v, f);
          unpack<bf16_t>(from_u32x4(uq[(2 * i + h) % UQ]), u);
          if (2 * i + h + UQ < 2 * MTW) uq[(2 * i + h) % UQ] = __builtin_amdgcn_raw_buffer_load_b128(rU, (int)out_off(2 * i + h + UQ), 0, 0);
          const float sc = p.bact_scale != 0.f ? p.bact_scale : 1.f;
          if (p.bact == ACT_DFACTOR) {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] *= u[e] * sc;
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = u[e] > 0.f ? f[e] * sc : 0.f;
          }
          v = pack<bf16_t>(f);
        } else if constexpr (EK == 4) {
          // inference: eval-mode BatchNorm as per-column scale / shift, activation, residual (gemm_kernel's escale / eshift / eres)
          float f[8], u[8];
          unpack<bf16_t>(v, f);
          unpack<bf16_t>(from_u32x4(uq[(2 * i + h) % UQ]), u);
          if (2 * i + h + UQ < 2 * MTW) uq[(2 * i + h) % UQ] = __builtin_amdgcn_raw_buffer_load_b128(rU, (int)out_off(2 * i + h + UQ), 0, 0);
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = __builtin_fmaf(f[e], esc8[e], esh8[e]);
          if (p.act == ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e], 0.f);
          } else if (p.act != ACT_NONE) {   // SiLU / sigmoid
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float sg = sigmoidf_(f[e]); f[e] = p.act == ACT_SILU ? f[e] * sg : sg; }
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] += u[e];
          v = pack<bf16_t>(f);
        } else if (HAS_AUX || STATS || p.act != ACT_NONE) {
          float f[8];
          unpack<bf16_t>(v, f);
          if (HAS_AUX && p.pre_out && !p.pre_grad) __builtin_amdgcn_raw_buffer_store_b128(to_u32x4(v), rP, (int)o, 0, 0);
          if (HAS_AUX && p.pre_out && p.pre_grad) {
            float d[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { d[e] = act_bwd(f[e], p.act); f[e] = act_fwd(f[e], p.act); }
            __builtin_amdgcn_raw_buffer_store_b128(to_u32x4(pack<bf16_t>(d)), rP, (int)o, 0, 0);
          } else if (p.act == ACT_GELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = gelu_fast(f[e]);
          } else if (p.act == ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e], 0.f);
          } else if (p.act != ACT_NONE) {   // SiLU / sigmoid (not through act_fwd: its erff branch would be unrolled in here too)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float sg = sigmoidf_(f[e]); f[e] = p.act == ACT_SILU ? f[e] * sg : sg; }
          }
          if (HAS_AUX && p.bact_u) {
            float u[8];
            unpack<bf16_t>(from_u32x4(__builtin_amdgcn_raw_buffer_load_b128(rU, (int)o, 0, 0)), u);
            const float sc = p.bact_scale != 0.f ? p.bact_scale : 1.f;
            if (p.bact == ACT_DFACTOR) {
#pragma unroll
              for (int e = 0; e < 8; ++e) f[e] *= u[e] * sc;
            } else if (p.bact == ACT_GELU) {
#pragma unroll
              for (int e = 0; e < 8; ++e) f[e] *= gelu_grad_fast(u[e]) * sc;
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) f[e] *= act_bwd(u[e], p.bact) * sc;
            }
          }
          if (STATS == 1 && o != 0xfffffff0u) {   // sums of the value this product contributes (before an accumulate), as gemm_kernel
#pragma unroll
            for (int e = 0; e < 8; ++e) { s1[e] += f[e]; s2[e] = __builtin_fmaf(f[e], f[e], s2[e]); }
          }
          if (HAS_AUX && p.beta) {
            float c0[8];
            unpack<bf16_t>(from_u32x4(__builtin_amdgcn_raw_buffer_load_b128(rC, (int)o, 0, 0)), c0);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += c0[e];
          }
          if (STATS == 2) {
            // y of the BatchNorm whose output gradient this is ([M][N], row stride N); out-of-range lanes read zeros and are masked
            const unsigned oy = (row < p.M && col < p.N) ? (unsigned)(((long)row * p.N + col) * 2) : 0xfffffff0u;
            float yv[8];
            unpack<bf16_t>(from_u32x4(__builtin_amdgcn_raw_buffer_load_b128(rY, (int)oy, 0, 0)), yv);
            if (oy != 0xfffffff0u) {
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const float g = f[e] * act_bwd(__builtin_fmaf(yv[e], bsc[e], bsh[e]), p.bnb_act);
                s1[e] += g; s2[e] = __builtin_fmaf(g, (yv[e] - bmu[e]) * brs[e], s2[e]);
              }
            }
          }
          v = pack<bf16_t>(f);
        }
        __builtin_amdgcn_raw_buffer_store_b128(to_u32x4(v), rC, (int)o, 0, 0);
      }
    }
    if (STATS) {
      // lanes r * 8 + (chunk ^ r), r = 0..7, hold the same eight columns: butterfly over r (lane ^ 9, ^ 18, ^ 36), then lanes 0..7
      // (r = 0: chunk = lane) lay the 64 column sums out through the scratch so that lane c adds column c: two 256-byte atomic
      // wave-instructions per tile and wave
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s1[e] += __shfl_xor(s1[e], 9, 64); s2[e] += __shfl_xor(s2[e], 9, 64);
        s1[e] += __shfl_xor(s1[e], 18, 64); s2[e] += __shfl_xor(s2[e], 18, 64);
        s1[e] += __shfl_xor(s1[e], 36, 64); s2[e] += __shfl_xor(s2[e], 36, 64);
      }
      __attribute__((address_space(3))) float* sc = (__attribute__((address_space(3))) float*)(lds_u8*)(size_t)eW0;   // [2][64]
      if (lane < 8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[lane * 8 + e] = s1[e]; sc[64 + lane * 8 + e] = s2[e]; }
      }
      const float a1 = sc[lane], a2 = sc[64 + lane];   // (same wave: LDS operations execute in order)
      const int c = tn * BN + wn * 64 + lane;
      if (c < p.N) {
        float* dst = p.stats + (size_t)(tm % p.stats_rep) * 2 * p.N + c;
        if (!(p.dbg & 32)) { atomicAdd(dst, a1); atomicAdd(dst + p.N, a2); }
      }
    }
  };

  if (HAS_BIAS) bias_load(0);
  acc_init();
  __builtin_amdgcn_s_barrier();      // stage 0 has landed
  read_a(0, fsw0, 0, aX);
  read_b(0, fsw0, bK0);
  int c_tile = 0, c_k = 0, slot = 0;
  for (int t = 0; t < T; ++t) {
    const int slot1 = slot == 2 ? 0 : slot + 1;
    const bool last_k = c_k == KT - 1;
    read_a(slot, fsw0, 1, aY);
    read_b(slot, fsw1, bK1);
    __builtin_amdgcn_s_setprio(1);
    mma_group(0, aX, bK0);
    __builtin_amdgcn_s_setprio(0);
    read_a(slot, fsw1, 0, aX);
    __builtin_amdgcn_s_setprio(1);
    mma_group(1, aY, bK0);
    __builtin_amdgcn_s_setprio(0);
    read_a(slot, fsw1, 1, aY);
    __builtin_amdgcn_s_setprio(1);
    mma_group(0, aX, bK1);
    __builtin_amdgcn_s_setprio(0);
    // every fragment of stage t is in registers: its slot may be refilled; stage t + 1 is complete once every loader has arrived
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + 1 < T) { read_a(slot1, fsw0, 0, aX); read_b(slot1, fsw0, bK0); }
    __builtin_amdgcn_s_setprio(1);
    mma_group(1, aY, bK1);
    __builtin_amdgcn_s_setprio(0);
    if (last_k) {
      if (HAS_BIAS && c_tile + 1 < my_tiles) bias_load(c_tile + 1);   // in flight during the epilogue
      if (!(p.dbg & 1)) epilogue(c_tile);
      if (c_tile + 1 < my_tiles) acc_init();
      c_k = 0; ++c_tile;
    } else {
      ++c_k;
    }
    slot = slot1;
  }
}

// =========================================================================================
// Weight gradient on the same skeleton:  dW[n][k] += sum_m dY[m][n] * X[m][k]   (fp32 atomics; + the bias gradient sum_m dY[m][n])
// The contraction index m is the ROW index of both operands, so the tiles are staged in their natural layout ([64 rows of m][n or k
// columns], whole 1 KiB DMA pieces = 2 rows of dY / 4 rows of X) and the MFMA fragments -- 8 consecutive m of one column -- come
// out of LDS with the transposing read ds_read_b64_tr_b16 (two per fragment).  A 16-lane group of that read touches 4 rows x 32
// bytes and a 32-lane half 8 rows (m = 0..3 and 8..11, or 4..7 and 12..15) at the SAME 32-byte column granule: with 512- / 256-byte rows
// all eight would sit on the same banks, so granule j of row m is stored at granule j ^ h(m), h(m) = (m & 3) | ((m >> 3) & 1) << 2
// (applied to each lane's DMA source address and to the fragment reads): eight distinct granules of one 256-byte bank row.
// Work item = (output tile BNo x 128, slice of M); items of one tile add their partial tiles with fp32 atomics (64-byte row segments
// straight from the accumulator layout).  The persistent grid, the loader / consumer roles and the one-barrier-per-k-step ring are
// the GEMM's above.  Reference: the weight / bias gradients autograd computes for every nn.Linear of networks/SWIN.py:24-47,84-209
// and the decoder (networks/EfficientSATRN.py:326-397).
// =========================================================================================
struct BigWP {
  const bf16_t* Y; const bf16_t* X; float* dW; float* dbias;
  float* part;   // null: fp32 atomics into dW; else the items' partial tiles go to part[slice][N][K] (plain stores; folded by the launcher)
  int M, N, K, ldy, lda, ldw;
  int ntn, ntk, nitems, splits, rows_per_split;   // rows_per_split: a multiple of 64
  unsigned y_bytes, x_bytes;
  int dbg;
};

typedef short s16x4_t __attribute__((ext_vector_type(4)));
DEVI uint2 lds_read_tr8(unsigned addr) {
  const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(lds_u8*)(size_t)addr);
  return __builtin_bit_cast(uint2, v);
}

// NTW: 16-row MFMA tiles per consumer wave along n (output tile = 32 * NTW rows of dW x 128 columns)
// NS: ring depth.  The bytes in flight per CU (NS - 1 stages) over the memory latency are the per-CU bandwidth: with three 32 KB stages the
// kernel ran at 34 GB/s per CU inside the training step (activations from the forward pass come from HBM, not from the cache-warm loop of
// a microbenchmark: 55 us against 32 us for the same shape), so the 128-row form takes a fourth stage (128 KB of LDS).
template <int NTW, bool DBIAS, int NS>
__global__ __launch_bounds__(BIG_THREADS, 2) void wgrad_big_kernel(BigWP p) {
  constexpr int KTW = 4, HN = NTW / 2;
  constexpr int BNo = 32 * NTW, BKo = 128;
  constexpr int YROW = BNo * 2, XROW = BKo * 2;                    // bytes per LDS row
  constexpr int RPY = 1024 / YROW, RPX = 1024 / XROW;              // rows per 1 KiB DMA piece
  constexpr int NYP = 64 / RPY / 4, NXP = 64 / RPX / 4;            // pieces per loader wave and stage
  constexpr int STAGE = 64 * (YROW + XROW);
  constexpr int XOFF = 64 * YROW;
  extern __shared__ __attribute__((aligned(16))) unsigned char big_sm[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)big_sm;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int KT = p.rows_per_split / 64;

  const int nwg = gridDim.x;
  const int L = xcd_remap(blockIdx.x, nwg);
  const int my_items = L < p.nitems ? (p.nitems - L + nwg - 1) / nwg : 0;
  const int T = my_items * KT;
  if (T == 0) return;
  // item -> (slice of M, n tile, k tile); k tile fastest: neighbouring items share their dY panel
  auto item_of = [&](int ii, int& sp, int& tn, int& tk) {
    const int it = L + ii * nwg;
    tk = it % p.ntk;
    const int r = it / p.ntk;
    tn = r % p.ntn;
    sp = r / p.ntn;
  };

  if (wave >= 4) {
    // =============================== loader waves ===============================
    const int lw = wave - 4;
    const i32x4 rY = make_rsrc(p.Y, p.y_bytes), rX = make_rsrc(p.X, p.x_bytes);
    // dY piece i = lw + 4 j: rows RPY * i + (lane / (YROW / 16)), chunk slot lane % (YROW / 16); granule of the source = (slot >> 1) ^ h(row)
    const int yr = lane / (YROW / 16), ys = lane % (YROW / 16);
    const int xr = lane / (XROW / 16), xs = lane % (XROW / 16);
    const unsigned ldy2 = (unsigned)p.ldy * 2u, lda2 = (unsigned)p.lda * 2u;
    int d_item = 0, d_k = 0;
    // per-lane source offsets of this wave's pieces, fixed for an item (rows of the stage x the lane's swizzled 16-byte column chunk); the
    // k-step only adds a wave-uniform row offset.  Rows past the end of M are zero-filled by the descriptor's range check -- slices
    // are whole multiples of 64 rows, so only the last slice can run past its end, and its end is M.  (The first form recomputed row,
    // swizzle, chunk, bounds and a 32-bit multiply per piece and k-step: ~40 VALU instructions x 8 pieces per loader wave and k-step.)
    unsigned offY[NYP], offX[NXP];
    int m_item0 = 0;
    auto dma_item_setup = [&](int ii) {
      int sp, tn, tk;
      item_of(ii, sp, tn, tk);
      m_item0 = sp * p.rows_per_split;
      const int n0 = tn * BNo, k0 = tk * BKo;
#pragma unroll
      for (int j = 0; j < NYP; ++j) {
        const int ml = (lw + 4 * j) * RPY + yr;                                  // row of the stage
        const int h = (ml & 3) | (((ml >> 3) & 1) << 2);
        const int c = ((((ys >> 1) ^ h) << 1) | (ys & 1));                        // source chunk (8 columns)
        offY[j] = n0 + c * 8 < p.N ? (unsigned)ml * ldy2 + (unsigned)(n0 + c * 8) * 2u : 0xfffffff0u;
      }
#pragma unroll
      for (int j = 0; j < NXP; ++j) {
        const int ml = (lw + 4 * j) * RPX + xr;
        const int h = (ml & 3) | (((ml >> 3) & 1) << 2);
        const int c = ((((xs >> 1) ^ h) << 1) | (xs & 1));
        offX[j] = k0 + c * 8 < p.K ? (unsigned)ml * lda2 + (unsigned)(k0 + c * 8) * 2u : 0xfffffff0u;
      }
    };
    auto dma_issue = [&](int slot) {
      const unsigned sbase = lds0 + (unsigned)slot * STAGE;
      const int mb = m_item0 + d_k * 64;
      // (the row offset goes into the VECTOR offset: the range check does not see the scalar offset)
      const unsigned rY0 = (unsigned)mb * ldy2, rX0 = (unsigned)mb * lda2;
#pragma unroll
      for (int j = 0; j < NYP; ++j) dma16(rY, sbase + (unsigned)(lw + 4 * j) * 1024u, offY[j] == 0xfffffff0u ? 0xfffffff0u : offY[j] + rY0, 0);
#pragma unroll
      for (int j = 0; j < NXP; ++j) dma16(rX, sbase + XOFF + (unsigned)(lw + 4 * j) * 1024u, offX[j] == 0xfffffff0u ? 0xfffffff0u : offX[j] + rX0, 0);
      if (++d_k == KT) { d_k = 0; ++d_item; if (d_item < my_items) dma_item_setup(d_item); }
    };
    constexpr int P = NYP + NXP;   // DMA instructions per loader wave and stage
    static_assert(NS == 3 || NS == 4, "ring depth");
    static_assert((NS - 1) * P <= 63, "vmcnt is a 6-bit counter");
    dma_item_setup(0);
    dma_issue(0);
    if (1 < T) dma_issue(1);
    if (2 < T) dma_issue(2);
    if (NS > 3 && 3 < T) dma_issue(3);
    // stage 0 landed: all but the stages requested after it
    if (NS > 3 && T >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * P) : "memory");
    else if (T >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * P) : "memory");
    else if (T == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int slot = 0;
    for (int t = 0; t < T; ++t) {
      // stage t + 1 landed (the stages behind it, at most NS - 2 of them, may stay in flight)
      if (NS > 3 && t + 3 < T) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * P) : "memory");
      else if (t + 2 < T) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (t + NS < T) dma_issue(slot);
      slot = slot == NS - 1 ? 0 : slot + 1;
    }
    return;
  }

  // =============================== consumer waves ===============================
  const int wm = wave >> 1, wn = wave & 1;          // halves of the tile along n / along k
  const int fr = lane & 15, fq = lane >> 4;
  const int tq = fr >> 2, tp = fr & 3;               // transposing read: lane 4 q + p of a 16-lane group addresses row q, columns 4 p .. 4 p + 3
  const int h = tq | ((fq & 1) << 2);                // h(m) of the rows this lane addresses (m = 32 ks + 8 fq + tq (+ 4))
  const unsigned rowoff = (unsigned)(8 * fq + tq);
  // per-lane byte offsets inside a stage (without the k-half and the tile index)
  const unsigned yA = rowoff * YROW + (unsigned)tp * 8u, xB = XOFF + rowoff * XROW + (unsigned)tp * 8u;
  uint4 aX[HN], aY[HN], bK0[KTW], bK1[KTW];
  auto read_y = [&](int slot, int ks, int half, uint4* af) {
    if (p.dbg & 8) return;
    const unsigned sbase = lds0 + (unsigned)slot * STAGE + yA + (unsigned)(ks * 32) * YROW;
#pragma unroll
    for (int i = 0; i < HN; ++i) {
      const unsigned a = sbase + (unsigned)(((wm * NTW + half * HN + i) ^ h) * 32);
      const uint2 lo = lds_read_tr8(a), hi = lds_read_tr8(a + 4u * YROW);
      af[i] = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
  };
  auto read_x = [&](int slot, int ks, uint4* bf) {
    if (p.dbg & 8) return;
    const unsigned sbase = lds0 + (unsigned)slot * STAGE + xB + (unsigned)(ks * 32) * XROW;
#pragma unroll
    for (int j = 0; j < KTW; ++j) {
      const unsigned a = sbase + (unsigned)(((wn * KTW + j) ^ h) * 32);
      const uint2 lo = lds_read_tr8(a), hi = lds_read_tr8(a + 4u * XROW);
      bf[j] = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
  };
  f32x4 acc[NTW][KTW];
  f32x4 accb[DBIAS ? NTW : 1];
  bool do_bias = false;                       // this wave sums the bias gradient of the current item (k tile 0, k half 0)
  const uint4 ones = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);   // eight bf16 1.0
  auto mma_group = [&](int half, const uint4* af, const uint4* bf) {
    if (p.dbg & 4) return;
#pragma unroll
    for (int i = 0; i < HN; ++i) {
#pragma unroll
      for (int j = 0; j < KTW; ++j)
        acc[half * HN + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[i]), __builtin_bit_cast(bf16x8, bf[j]), acc[half * HN + i][j], 0, 0, 0);
      if (DBIAS && do_bias)
        accb[half * HN + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[i]), __builtin_bit_cast(bf16x8, ones), accb[half * HN + i], 0, 0, 0);
    }
  };
  auto acc_init = [&](int ii) {
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
      for (int j = 0; j < KTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (DBIAS) {
#pragma unroll
      for (int i = 0; i < NTW; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      int sp, tn, tk;
      item_of(ii, sp, tn, tk);
      do_bias = p.dbias != nullptr && tk == 0 && wn == 0;
    }
  };
  auto epilogue = [&](int ii) {
    int sp, tn, tk;
    item_of(ii, sp, tn, tk);
    const int nb = tn * BNo + wm * 16 * NTW + fq * 4, kb = tk * BKo + wn * 64 + fr;
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
      for (int j = 0; j < KTW; ++j) {
        const int k = kb + j * 16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = nb + i * 16 + r;
          if (n < p.N && k < p.K) {
            if (p.part) p.part[((size_t)sp * p.N + n) * p.K + k] = acc[i][j][r];
            else atomicAdd(p.dW + (size_t)n * p.ldw + k, acc[i][j][r]);
          }
        }
      }
    if (DBIAS && do_bias && fr == 0) {
#pragma unroll
      for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = nb + i * 16 + r;
          if (n < p.N) atomicAdd(p.dbias + n, accb[i][r]);
        }
    }
  };

  acc_init(0);
  __builtin_amdgcn_s_barrier();      // stage 0 has landed
  read_y(0, 0, 0, aX);
  read_x(0, 0, bK0);
  int c_item = 0, c_k = 0, slot = 0;
  for (int t = 0; t < T; ++t) {
    const int slot1 = slot == NS - 1 ? 0 : slot + 1;
    const bool last_k = c_k == KT - 1;
    read_y(slot, 0, 1, aY);
    read_x(slot, 1, bK1);
    __builtin_amdgcn_s_setprio(1);
    mma_group(0, aX, bK0);
    __builtin_amdgcn_s_setprio(0);
    read_y(slot, 1, 0, aX);
    __builtin_amdgcn_s_setprio(1);
    mma_group(1, aY, bK0);
    __builtin_amdgcn_s_setprio(0);
    read_y(slot, 1, 1, aY);
    __builtin_amdgcn_s_setprio(1);
    mma_group(0, aX, bK1);
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + 1 < T) { read_y(slot1, 0, 0, aX); read_x(slot1, 0, bK0); }
    __builtin_amdgcn_s_setprio(1);
    mma_group(1, aY, bK1);
    __builtin_amdgcn_s_setprio(0);
    if (last_k) {
      if (!(p.dbg & 1)) epilogue(c_item);
      if (c_item + 1 < my_items) acc_init(c_item + 1);
      c_k = 0; ++c_item;
    } else {
      ++c_k;
    }
    slot = slot1;
  }
}

static int big_dbg() { static const int v = sw_timing("big_dbg"); return v; }
static int big_cu_count() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  }
  return cus;
}

template <int MT /*block tile = 64 * MT rows*/>
static void big_launch_t(const BigP& p, int grid, hipStream_t s) {
  constexpr int MTW = 2 * MT;
  constexpr size_t sh = (size_t)BIG_NSTAGE * (64 * MT + 128) * BIG_ROWB + 4 * 4096;
  const bool aux = p.pre_out || p.bact_u || p.beta;
  // epilogue kind (see the kernel): the two lean forms when nothing else is asked for, the all-purpose one otherwise
  const bool ek2 = p.pre_out && p.pre_grad && p.act == ACT_GELU && !p.bact_u && !p.beta && !p.stats;
  const bool ek3 = p.bact_u && (p.bact == ACT_DFACTOR || p.bact == ACT_RELU) && !p.pre_out && !p.beta && !p.stats && p.act == ACT_NONE;
#define BIG_GO(HB, EKIND, ST) do { \
    static bool attr = false; \
    if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_big_kernel<MTW, HB, EKIND, ST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); attr = true; } \
    hipLaunchKernelGGL((gemm_big_kernel<MTW, HB, EKIND, ST>), dim3(grid), dim3(BIG_THREADS), sh, s, p); } while (0)
  // the statistics forms keep 16 (sums) / 48 (+ BatchNorm coefficients) more registers: tiles of at most 192 / 128 rows
  const bool ek4 = p.escale != nullptr;   // inference epilogue: nothing else with it (gemm_big_go filters)
  if (p.conv && p.narrow) {
    // 64-column tiles, BM = 64 MT rows (MT = 2 / 4 only: the fragment ping-pong needs an even number of row groups per wave)
    constexpr int MTN = MT == 3 ? 2 : MT;
    constexpr size_t shn = (size_t)BIG_NSTAGE * (64 * MTN + 64) * BIG_ROWB + 4 * 4096;
#define BIG_GON(EKIND, ST) do { \
    static bool attr = false; \
    if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_big_kernel<MTN, false, EKIND, ST, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shn); attr = true; } \
    hipLaunchKernelGGL((gemm_big_kernel<MTN, false, EKIND, ST, true, 1>), dim3(grid), dim3(BIG_THREADS), shn, s, p); } while (0)
    if (ek4) { BIG_GON(4, 0); return; }
    if (p.stats && p.bnb_y) { if (aux) BIG_GON(1, 2); else BIG_GON(0, 2); return; }
    if (p.stats) { if (aux) BIG_GON(1, 1); else BIG_GON(0, 1); return; }
    if (aux) BIG_GON(1, 0); else BIG_GON(0, 0);
#undef BIG_GON
    return;
  }
  if (p.conv) {
    // convolution mode: no bias / activation / second tensor besides accumulate (gemm_big_conv_launch filters)
#define BIG_GOC(EKIND, ST) do { \
    static bool attr = false; \
    if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_big_kernel<MTW, false, EKIND, ST, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); attr = true; } \
    hipLaunchKernelGGL((gemm_big_kernel<MTW, false, EKIND, ST, true>), dim3(grid), dim3(BIG_THREADS), sh, s, p); } while (0)
    if constexpr (MT <= 3) { if (ek4) { BIG_GOC(4, 0); return; } }
    if constexpr (MT <= 2) { if (p.stats && p.bnb_y) { if (aux) BIG_GOC(1, 2); else BIG_GOC(0, 2); return; } }
    if constexpr (MT <= 3) {
      if (p.stats && !p.bnb_y) { if (aux) BIG_GOC(1, 1); else BIG_GOC(0, 1); return; }
      if (aux) { BIG_GOC(1, 0); return; }
    }
    BIG_GOC(0, 0);
#undef BIG_GOC
    return;
  }
  if constexpr (MT <= 2) { if (p.stats && p.bnb_y) { if (aux) BIG_GO(false, 1, 2); else BIG_GO(false, 0, 2); return; } }
  if constexpr (MT <= 3) { if (p.stats && !p.bnb_y) { if (aux) BIG_GO(false, 1, 1); else BIG_GO(false, 0, 1); return; } }
  // (the forms with a second tensor in the epilogue -- pre_out / bact_u / beta -- spill at 256-row tiles: at most 192 rows, as the statistics forms)
  if constexpr (MT <= 3) {
    if (ek4) { BIG_GO(false, 4, 0); return; }
    if (ek2) { if (p.bias) BIG_GO(true, 2, 0); else BIG_GO(false, 2, 0); return; }
    if (ek3) { if (p.bias) BIG_GO(true, 3, 0); else BIG_GO(false, 3, 0); return; }
    if (aux) { if (p.bias) BIG_GO(true, 1, 0); else BIG_GO(false, 1, 0); return; }
  }
  if (p.bias) BIG_GO(true, 0, 0); else BIG_GO(false, 0, 0);
#undef BIG_GO
}

// true = launched.  amode must be AM_DENSE, dtype bf16.
static bool gemm_big_go(const GemmP& g, hipStream_t s, bool conv, int flip);
bool gemm_big_launch(const GemmP& g, hipStream_t s) { return gemm_big_go(g, s, false, 0); }
// 3x3 stride-1 'same' convolution (forward: amode AM_CONV, weights [N][9][Ci]; data gradient: AM_DGRAD, weights = the backward pack) on the
// persistent kernel.  Same epilogue subset as the dense form (BatchNorm statistics / backward sums, accumulate).
bool gemm_big_conv_launch(int amode, const GemmP& g, hipStream_t s) {
  if (g.KW != 3 || (g.stride != 1 && g.stride != 2) || (g.Ci & 7) || g.ldc != g.N) return false;
  if (g.bias || (g.act && !g.escale) || g.pre_out || g.bact_u) return false;
  if ((long)(g.M / (g.OH * g.OW)) * g.OH * g.OW != g.M || g.K != 9 * g.Ci) return false;
  return gemm_big_go(g, s, true, amode == AM_DGRAD ? 1 : 0);
}
static bool gemm_big_go(const GemmP& g, hipStream_t s, bool conv, int flip) {
  const char* mode_env = sw_knob_str("gemm_big");   // read per call (tests and tools switch it): 0 = off, 2 = take every shape that fits
  const int mode = mode_env ? atoi(mode_env) : 1;
  if (!mode) return false;
  if (g.out_f32 || g.drop_p > 0.f || (g.eres && !g.escale)) return false;
  // inference epilogue (eval-mode BatchNorm scale / shift + activation + residual): on its own only
  if (g.escale && (!g.eshift || g.bias || g.stats || g.pre_out || g.bact_u || g.beta || g.bnb_y)) return false;
  if (g.stats && (g.stats_part || g.bias || g.stats_rep < 1)) return false;   // deterministic slabs / biased statistics: gemm_kernel
  if (g.bnb_y && !g.stats) return false;
  if ((g.K & 7) || (!conv && (g.lda & 7)) || (g.ldc & 7) || (g.N & 7) || g.M < 1) return false;
  if ((conv ? (size_t)(g.M / (g.OH * g.OW)) * g.H * g.W * g.Ci : (size_t)g.M * g.lda) * 2 >= (1ull << 31) || (size_t)g.N * g.K * 2 >= (1ull << 31) || (size_t)g.M * g.ldc * 2 >= (1ull << 31)) return false;   // 32-bit buffer offsets
  {
    // default mode: the large products only (SwinTRN's linears, M = 2 304 .. 147 456 with N >= 128).  Measured inside the EfficientSATRN
    // step (tools/shape_prof.py, SATRN_GEMM_BIG_MIN_GFLOP=0.5 against the default): its 1x1 convolutions -- 0.5 .. 2 GFLOP each, inputs
    // just written by another kernel, a weight-gradient kernel running beside them -- gain nothing from the persistent form (4.66 ->
    // 4.91 ms over the family): a 160 KB workgroup needs a whole drained CU to start and leaves no room for the side stream
    static const double min_gflop = sw_knobf("gemm_big_min_gflop", 2.0);
    const double flops = 2.0 * g.M * g.N * g.K;
    static const int min_n = (int)sw_knob("gemm_big_min_n", 128);
    // (inference products -- escale set -- run with no weight-gradient stream beside them: the smaller ones gain as well, 13 -> 7 us at 1.6 GFLOP)
    const double mg = g.escale ? std::min(min_gflop, 1.0) : min_gflop;
    if (mode != 2 && (flops < mg * 1e9 || g.N < (conv ? 32 : min_n) || g.M < 2048)) return false;
  }
  BigP p;
  p.A = (const bf16_t*)g.A; p.W = (const bf16_t*)g.Bw; p.C = (bf16_t*)g.C; p.bias = g.bias;
  p.pre_out = (bf16_t*)g.pre_out; p.bact_u = (const bf16_t*)g.bact_u;
  p.escale = g.escale; p.eshift = g.eshift; p.eres = (const bf16_t*)g.eres;
  p.M = g.M; p.N = g.N; p.K = g.K; p.lda = g.lda; p.ldc = g.ldc; p.act = g.act; p.bact = g.bact; p.beta = g.beta; p.bact_scale = g.bact_scale; p.pre_grad = g.pre_grad;
  p.a_bytes = conv ? (unsigned)((size_t)(g.M / (g.OH * g.OW)) * g.H * g.W * g.Ci * 2) : (unsigned)(((size_t)(g.M - 1) * g.lda + g.K) * 2);
  p.cOH = g.OH; p.cOW = g.OW; p.cstride = g.stride; p.cpt = g.pt; p.cpl = g.pl;
  p.conv = conv ? 1 : 0; p.cH = g.H; p.cW = g.W; p.cCi = g.Ci; p.flip = flip; p.kpt = conv ? (g.Ci + 63) / 64 : 0;
  p.w_bytes = (unsigned)((size_t)g.N * g.K * 2);
  p.c_bytes = (unsigned)(((size_t)(g.M - 1) * g.ldc + g.N) * 2);
  p.stats = g.stats; p.stats_rep = g.stats_rep; p.bnb_y = (const bf16_t*)g.bnb_y; p.bnb_ss = g.bnb_ss; p.bnb_mr = g.bnb_mr; p.bnb_act = g.bnb_act;
  p.y_bytes = (unsigned)((size_t)g.M * g.N * 2);
  p.narrow = (conv && g.N <= 64) ? 1 : 0;
  p.ntn = p.narrow ? (g.N + 63) / 64 : (g.N + 127) / 128;
  p.dbg = big_dbg();
  const int cus = big_cu_count();
  // tile height: the candidate whose tile count leaves the smallest idle share in the last round of the persistent grid
  const int force_mt = (int)sw_knob("gemm_big_mt", 0);
  int best_mt = 3;
  double best_cost = 1e30;
  const bool aux_form = g.pre_out || g.bact_u || g.beta || g.escale;
  const int mt_max = p.narrow ? 4 : (g.bnb_y ? 2 : ((g.stats || aux_form) ? 3 : 4));
  for (int mt = mt_max; mt >= 2; --mt) {
    if (p.narrow && mt == 3) continue;   // (64-column tiles: 128 or 256 rows)
    const long tiles = (long)((g.M + 64 * mt - 1) / (64 * mt)) * p.ntn;
    const long rounds = (tiles + cus - 1) / cus;
    // cost ~ rounds x (rows per tile + a fixed per-tile part worth ~48 rows: epilogue + the wider share of W traffic of flat tiles)
    const double cost = (double)rounds * (64.0 * mt + 48.0);
    if (cost < best_cost) { best_cost = cost; best_mt = mt; }
  }
  if (force_mt >= 2 && force_mt <= mt_max) best_mt = force_mt;
  p.ntm = (g.M + 64 * best_mt - 1) / (64 * best_mt);
  p.ntiles = p.ntm * p.ntn;
  int grid = p.ntiles < cus ? p.ntiles : cus;
  static const int grid_env = (int)sw_knob("big_grid", 0);   // experiment (tools/gemm_big_grid.sh): fewer CUs, same tiles
  if (grid_env > 0) grid = std::min(grid, grid_env);
  if (best_mt == 4) big_launch_t<4>(p, grid, s);
  else if (best_mt == 3) big_launch_t<3>(p, grid, s);
  else big_launch_t<2>(p, grid, s);
  g_route[conv ? RT_GEMM_BIG_CONV : RT_GEMM_BIG]++;
  return true;
}

template <int NTW, bool DB>
static void wgrad_big_go(const BigWP& p, int grid, hipStream_t s) {
  constexpr int NS = NTW <= 4 ? 4 : 3;   // 4 x 32 KB or 3 x 48 KB
  constexpr size_t sh = (size_t)NS * 64 * (32 * NTW * 2 + 256);
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)wgrad_big_kernel<NTW, DB, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); attr = true; }
  hipLaunchKernelGGL((wgrad_big_kernel<NTW, DB, NS>), dim3(grid), dim3(BIG_THREADS), sh, s, p);
}

// dense bf16 weight gradient (fp32 atomics into dW, optional bias gradient); true = launched
bool wgrad_big_launch(const WgradP& w, hipStream_t s) {
  const char* mode_env = sw_knob_str("wgrad_big");   // read per call: 0 = off, 2 = every shape that fits
  const int mode = mode_env ? atoi(mode_env) : 1;
  if (!mode || w.conv || w.out_t || w.det_part || (w.nbatch > 1)) return false;
  if ((w.N & 7) || (w.K & 7) || (w.ldy & 7) || (w.lda & 7) || w.M < 64) return false;
  if ((size_t)w.M * w.ldy * 2 >= (1ull << 31) || (size_t)w.M * w.lda * 2 >= (1ull << 31)) return false;
  const double flops = 2.0 * w.M * w.N * w.K;
  static const double wenv = sw_knobf("wgrad_big_min_gflop", 0.0);   // knob: overrides the per-network value
  const double wmin = wenv > 0.0 ? wenv : (double)g_wgrad_big_min_gflop;
  if (mode != 2 && (flops < wmin * 1e9 || w.M < 2048)) return false;
  BigWP p;
  p.Y = (const bf16_t*)w.dY; p.X = (const bf16_t*)w.A; p.dW = (float*)w.dW; p.dbias = w.dbias;
  p.M = w.M; p.N = w.N; p.K = w.K; p.ldy = w.ldy; p.lda = w.lda; p.ldw = w.K;
  p.y_bytes = (unsigned)(((size_t)(w.M - 1) * w.ldy + w.N) * 2);
  p.x_bytes = (unsigned)(((size_t)(w.M - 1) * w.lda + w.K) * 2);
  p.dbg = big_dbg();
  const int cus = big_cu_count();
  // 256-row tiles only where the output alone fills the grid: an item's partial tile leaves as fp32 atomics (1.3 TB/s over the chip), so
  // with slices of M the atomic bytes are items x tile -- the tall form doubles them (9216 x 1536 x 384: 40 us against 32 us with 128 rows)
  const bool tall = w.N > 128 && !w.dbias && (long)((w.N + 255) / 256) * ((w.K + 127) / 128) >= cus;
  const int bno = tall ? 256 : 128;
  p.ntn = (w.N + bno - 1) / bno; p.ntk = (w.K + 127) / 128;
  const int tiles = p.ntn * p.ntk;
  // slices of M: enough items to fill the persistent grid (at least 256 rows per slice)
  const int target = cus;   // (a smaller item count for the side stream was measured: no gain)
  // Slices of M: the persistent grid runs its items in rounds, so what counts is rounds x (k-steps per item + the item's fixed part, its
  // 16 K atomics ~ 3 k-steps) -- not merely "at least one item per CU": 36 tiles x 8 slices = 288 items on 256 CUs cost two rounds of 18
  // k-steps where 7 slices (252 items) cost one round of 21 (SwinTRN fc1 / fc2 at M = 9 216: 55 -> 3x us).
  const int max_splits = (w.M + 255) / 256;
  int splits = 1, rps = ((w.M + 63) / 64) * 64;
  {
    double best = 1e30;
    for (int sp = 1; sp <= max_splits; ++sp) {
      int r = (w.M + sp - 1) / sp;
      r = ((r + 63) / 64) * 64;
      const int sp2 = (w.M + r - 1) / r;
      const long items = (long)tiles * sp2;
      const long rounds = (items + target - 1) / target;
      const double cost = (double)rounds * (r / 64 + 3.0);
      if (cost < best - 1e-9) { best = cost; splits = sp2; rps = r; }
    }
  }
  // (measured: 2-4x more, shorter items on the side stream, one per workgroup, so that CUs are released more often for the chain's
  // persistent kernels -- SwinTRN 17.9 -> 19.4-20.3 ms per step: the extra partial-tile traffic and fold work cost more)
  p.splits = splits; p.rows_per_split = rps; p.nitems = tiles * splits;
  static const bool wgrad_atomics = sw_off("wgrad_big_partials");   // A/B: keep the float atomics
  // partial tiles instead of atomics when a slab is there (engine calls) and there is something to fold
  p.part = nullptr;
  const size_t need = (size_t)splits * w.N * w.K;
  if (splits > 1 && g_wgpart.cap >= need && p.ldw == w.K && (((size_t)w.N * w.K) & 3) == 0 && !wgrad_atomics)
    p.part = g_wgpart.scratch[(g_wgpart.side && s == g_wgpart.side) ? 1 : 0];
  const int grid = p.nitems < target ? p.nitems : target;
  if (tall) wgrad_big_go<8, false>(p, grid, s);
  else if (w.dbias) wgrad_big_go<4, true>(p, grid, s);
  else wgrad_big_go<4, false>(p, grid, s);
  if (p.part) launch_fold4(p.part, splits, (long)w.N * w.K, (long)w.N * w.K, p.dW, s);
  g_route[RT_WGRAD_BIG]++;
  return true;
}
