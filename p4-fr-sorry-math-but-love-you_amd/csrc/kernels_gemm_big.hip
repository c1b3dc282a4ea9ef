// Large dense bf16 GEMM for gfx950:  C[M,N] = epilogue(A[M,K] * W[N,K]^T)   (both operands K-contiguous)
//
// The 4-wave register-staged kernel of kernels_gemm.hip tops out below 20 % of the MFMA peak on the big products of SwinTRN
// (M = 9 216 .. 147 456, K = 128 .. 2 048) and the early backbone stages: every k-step is global load -> VGPR -> ds_write ->
// barrier -> ds_read -> MFMA by the same four waves, and every tile pays its own prologue and store tail.  This kernel is
// the chip-filling form:
//   * PERSISTENT: one 512-thread workgroup (8 waves, 4 along M x 2 along N) per compute unit walks a list of output tiles; the
//     (tile, k-step) pairs of a workgroup form ONE stream, so the operand loads of the next tile are in flight while the
//     current tile is multiplied and stored -- no per-tile prologue bubble (K = 128 .. 512 means only 2 .. 8 k-steps per tile).
//   * DIRECT-TO-LDS: operands go global -> LDS with `buffer_load_dwordx4 ... lds` (no VGPRs, no ds_write); three 64-deep
//     stages of (BM + 128) rows x 128 B form a ring, two stages in flight behind a COUNTED s_waitcnt vmcnt and ONE raw
//     s_barrier per k-step.  Out-of-range rows (M / N tails) are zero-filled by the buffer descriptor's range check.
//   * LDS image: row-major 128-byte rows, 16-byte chunk c of row r stored at chunk position c ^ ((r >> 1) & 7): the DMA writes
//     lane-linear (8 lanes = one row), so the swizzle is applied to each lane's SOURCE address and again on the fragment
//     reads -- ds_read_b128 of an MFMA fragment (16 rows x one chunk per 16-lane group) is then bank-conflict free.
//   * fragments are double-buffered in registers across the barrier: the ds_reads of half a k-step run under the 16 MFMAs
//     of the other half.
//   * the MFMA is issued with the operands swapped (D' = W_frag x A_frag^T), so a lane holds FOUR CONSECUTIVE output columns of
//     one row; the tile goes through a per-wave LDS scratch (16 rows x 128 B, XOR-swizzled) and leaves as whole 128-byte row
//     segments, 16 bytes per lane (the old epilogue stored one 2-byte element per lane).
// Epilogue subset: bias (added to the initial accumulator), activation, pre-activation copy (pre_out), act'(u) factor of a data
// gradient (bact_u), accumulate (beta).  Everything else (BatchNorm statistics, dropout, f32 output, inference scale/shift) stays
// with gemm_kernel; gemm_big_launch() returns false for those.
// Reference shapes: networks/SWIN.py:84-209 (qkv / proj), :24-47 (Mlp fc1 / fc2), networks/EfficientSATRN.py:66-87 (1x1 convs).
#include <stdio.h>
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

struct BigP {
  const bf16_t* A; const bf16_t* W; bf16_t* C;
  const float* bias;
  bf16_t* pre_out; const bf16_t* bact_u;
  int M, N, K, lda, ldc;
  int act, bact, beta;
  float bact_scale;
  int ntm, ntn, ntiles;
  unsigned a_bytes, w_bytes, c_bytes;   // buffer extents for the range check
};

#define BIG_THREADS 512
#define BIG_BK 64
#define BIG_ROWB 128            // bytes per LDS row (64 bf16)
#define BIG_NSTAGE 3

// counted wait on the vector-memory queue: N must be a compile-time immediate, the number of younger operations that may stay in flight
// is only known at run time (tile boundaries put stores and bias loads into the queue): dispatch over the even values
DEVI void vm_wait_n(int n) {
  n = n > 30 ? 30 : n;
  switch (n >> 1) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(26)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(30)" ::: "memory"); break;
  }
}

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

// MT: 16-row MFMA tiles per wave along M (block tile BM = 64 * MT rows); the block tile is 128 columns wide (2 waves x 4 MFMA tiles)
template <int MT, bool HAS_BIAS, bool HAS_AUX /*pre_out / bact_u / beta*/>
__global__ __launch_bounds__(BIG_THREADS, 2) void gemm_big_kernel(BigP p) {
  constexpr int NT = 4;
  constexpr int BM = 64 * MT, BN = 128;
  constexpr int TM = 16 * MT;                                     // rows per wave
  constexpr int STAGE = (BM + BN) * BIG_ROWB;                     // bytes per ring slot
  constexpr int EPI = BIG_NSTAGE * STAGE;                         // epilogue scratch: 8 waves x 2 KB
  constexpr int NA = MT, NB = 2;                                  // DMA instructions per wave and stage (8 rows each)
  constexpr int NDMA = NA + NB;
  extern __shared__ __attribute__((aligned(16))) unsigned char big_sm[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)big_sm;   // LDS byte address of the ring

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;
  const int KT = p.K / BIG_BK;

  // ---- this workgroup's tiles: logical id L (XCD-contiguous) + i * gridDim; tiles ordered n fastest, so the 32 workgroups of an
  // XCD share two row blocks of A and all of W in their L2
  const int nwg = gridDim.x;
  const int L = xcd_remap(blockIdx.x, nwg);
  const int my_tiles = L < p.ntiles ? (p.ntiles - L + nwg - 1) / nwg : 0;
  const int T = my_tiles * KT;     // k-steps of this workgroup's stream
  if (T == 0) return;

  const i32x4 rA = make_rsrc(p.A, p.a_bytes), rW = make_rsrc(p.W, p.w_bytes);
  // outputs / epilogue operands through buffer descriptors too: a lane outside the matrix gets an out-of-range offset and the
  // access is dropped by the range check, so EVERY lane issues EVERY epilogue access and the wave's operation count is exact
  const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc((void*)p.C, 0, (int)p.c_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc((void*)p.pre_out, 0, p.pre_out ? (int)p.c_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rU = __builtin_amdgcn_make_buffer_rsrc((void*)p.bact_u, 0, p.bact_u ? (int)p.c_bytes : 0, 0x00020000);

  // ---- DMA side: lane -> (row within the 8-row piece, chunk slot); source chunk = slot ^ swizzle(row)
  const int drow = lane >> 3;                                                        // 0..7
  const int dsw = ((lane >> 4) | ((wave & 1) << 2));                                 // ((8 * i + drow) >> 1) & 7 with i = wave + 8 j
  const unsigned dchunk = (unsigned)(((lane & 7) ^ dsw) * 16);
  const unsigned lda2 = (unsigned)p.lda * 2u, ldw2 = (unsigned)p.K * 2u;
  int d_tile = 0, d_k = 0;                 // position of the NEXT stage to request
  unsigned voffA = 0, voffW = 0;
  auto dma_tile_setup = [&](int ti) {
    const int tile = L + ti * nwg;
    const int tm = tile / p.ntn, tn = tile - tm * p.ntn;
    voffA = (unsigned)(tm * BM + 8 * wave + drow) * lda2 + dchunk;
    voffW = (unsigned)(tn * BN + 8 * wave + drow) * ldw2 + dchunk;
  };
  auto dma_issue = [&](int slot) {
    const unsigned sbase = lds0 + (unsigned)slot * STAGE;
    const int soff = d_k * (BIG_BK * 2);
#pragma unroll
    for (int j = 0; j < NA; ++j) dma16(rA, sbase + (unsigned)(wave + 8 * j) * 1024u, voffA + (unsigned)j * 64u * lda2, soff);
#pragma unroll
    for (int j = 0; j < NB; ++j) dma16(rW, sbase + (unsigned)BM * BIG_ROWB + (unsigned)(wave + 8 * j) * 1024u, voffW + (unsigned)j * 64u * ldw2, soff);
    if (++d_k == KT) { d_k = 0; ++d_tile; if (d_tile < my_tiles) dma_tile_setup(d_tile); }
  };

  // ---- fragment side
  const unsigned fsw0 = (unsigned)(((0 * 4 + fq) ^ ((fr >> 1) & 7)) * 16), fsw1 = (unsigned)(((1 * 4 + fq) ^ ((fr >> 1) & 7)) * 16);
  const unsigned fA = (unsigned)(wm * TM + fr) * BIG_ROWB;
  const unsigned fB = (unsigned)BM * BIG_ROWB + (unsigned)(wn * 64 + fr) * BIG_ROWB;
  uint4 a0[MT], b0[NT], a1[MT], b1[NT];
  auto read_half = [&](int slot, unsigned fsw, uint4* af, uint4* bf) {
    const unsigned sbase = lds0 + (unsigned)slot * STAGE + fsw;
#pragma unroll
    for (int i = 0; i < MT; ++i) af[i] = lds_read16(sbase + fA + (unsigned)i * 16u * BIG_ROWB);
#pragma unroll
    for (int j = 0; j < NT; ++j) bf[j] = lds_read16(sbase + fB + (unsigned)j * 16u * BIG_ROWB);
  };
  f32x4 acc[MT][NT];
  auto mma_half = [&](const uint4* af, const uint4* bf) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bf[j]), __builtin_bit_cast(bf16x8, af[i]), acc[i][j], 0, 0, 0);
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
      // (rows of bias past N: clamp, the columns are never stored)
      const float* src = p.bias + (n + 3 < p.N ? n : 0);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bias_r[j]) : "v"(src) : "memory");
    }
  };
  // after the counted wait: makes every later use of the asm-loaded registers depend on a statement behind that wait
  auto bias_tie = [&]() {
#pragma unroll
    for (int j = 0; j < NT; ++j) asm volatile("" : "+v"(bias_r[j]));
  };
  auto acc_init = [&]() {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = HAS_BIAS ? bias_r[j] : f32x4{0.f, 0.f, 0.f, 0.f};
  };

  // ---- epilogue of one finished tile
  const unsigned eW = lds0 + EPI + (unsigned)wave * 2048u;
  const int e_row = lane >> 3, e_chunk = (lane & 7) ^ ((lane >> 3) & 7);
  auto epilogue = [&](int ti) {
    const int tile = L + ti * nwg;
    const int tm = tile / p.ntn, tn = tile - tm * p.ntn;
    const int col = tn * BN + wn * 64 + e_chunk * 8;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
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
        if (HAS_AUX || p.act != ACT_NONE) {
          float f[8];
          unpack<bf16_t>(v, f);
          if (HAS_AUX && p.pre_out) __builtin_amdgcn_raw_buffer_store_b128(to_u32x4(v), rP, (int)o, 0, 0);
          if (p.act != ACT_NONE) {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = act_fwd(f[e], p.act);
          }
          if (HAS_AUX && p.bact_u) {
            float u[8];
            unpack<bf16_t>(from_u32x4(__builtin_amdgcn_raw_buffer_load_b128(rU, (int)o, 0, 0)), u);
            const float sc = p.bact_scale != 0.f ? p.bact_scale : 1.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] *= act_bwd(u[e], p.bact) * sc;
          }
          if (HAS_AUX && p.beta) {
            float c0[8];
            unpack<bf16_t>(from_u32x4(__builtin_amdgcn_raw_buffer_load_b128(rC, (int)o, 0, 0)), c0);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += c0[e];
          }
          v = pack<bf16_t>(f);
        }
        __builtin_amdgcn_raw_buffer_store_b128(to_u32x4(v), rC, (int)o, 0, 0);
      }
    }
  };
  // vector-memory operations a wave issues per epilogue (stores; the aux forms add loads the compiler waits for itself)
  const int n_epi_ops = 2 * MT * (1 + ((HAS_AUX && p.pre_out) ? 1 : 0) + ((HAS_AUX && p.bact_u) ? 1 : 0) + ((HAS_AUX && p.beta) ? 1 : 0));

  // ---- prologue: stages 0, 1, 2 requested; bias of tile 0
  int issued = 0;                    // vector-memory operations issued by this wave so far (the queue retires in order)
  // `issued` right after the DMA of the stage that is needed next (mk1: stage t + 1), the one after (mk2: t + 2), and of the
  // newest request (mk3); scalar variables, rotated (a runtime-indexed array would live in scratch memory)
  int mk0, mk1, mk2, mk3 = 0, mark_bias = 0;
  dma_tile_setup(0);
  if (HAS_BIAS) { bias_load(0); issued += NT; mark_bias = issued; }
  if (0 < T) { dma_issue(0); issued += NDMA; }
  mk0 = issued;
  if (1 < T) { dma_issue(1); issued += NDMA; }
  mk1 = issued;
  if (2 < T) { dma_issue(2); issued += NDMA; }
  mk2 = issued;
  if (HAS_BIAS) { vm_wait_n(issued - mark_bias); bias_tie(); }
  acc_init();
  vm_wait_n(issued - mk0);
  __builtin_amdgcn_s_barrier();
  read_half(0, fsw0, a0, b0);

  int c_tile = 0, c_k = 0;
  // stage index modulo 3 without a division
  int slot = 0;
  for (int t = 0; t < T; ++t) {
    const int slot1 = slot == 2 ? 0 : slot + 1;
    const bool last_k = c_k == KT - 1;
    // bias of the next tile, requested a whole k-step before it is needed
    if (HAS_BIAS && last_k && c_tile + 1 < my_tiles) { bias_load(c_tile + 1); issued += NT; mark_bias = issued; }
    read_half(slot, fsw1, a1, b1);
    __builtin_amdgcn_s_setprio(1);
    mma_half(a0, b0);
    __builtin_amdgcn_s_setprio(0);
    // stage t + 1 has landed (this wave's pieces), every wave is past its reads of stage t: the slot of stage t is free
    if (t + 1 < T) vm_wait_n(issued - mk1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + BIG_NSTAGE < T) { dma_issue(slot); issued += NDMA; }
    mk3 = issued;
    mk1 = mk2; mk2 = mk3;
    if (t + 1 < T) read_half(slot1, fsw0, a0, b0);
    __builtin_amdgcn_s_setprio(1);
    mma_half(a1, b1);
    __builtin_amdgcn_s_setprio(0);
    if (last_k) {
      epilogue(c_tile);
      issued += n_epi_ops;
      if (c_tile + 1 < my_tiles) {
        if (HAS_BIAS) { vm_wait_n(issued - mark_bias); bias_tie(); }
        acc_init();
      }
      c_k = 0; ++c_tile;
    } else {
      ++c_k;
    }
    slot = slot1;
  }
}

static int big_cu_count() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  }
  return cus;
}

template <int MT>
static void big_launch_t(const BigP& p, int grid, hipStream_t s) {
  constexpr size_t sh = (size_t)BIG_NSTAGE * (64 * MT + 128) * BIG_ROWB + 8 * 2048;
  const bool aux = p.pre_out || p.bact_u || p.beta;
#define BIG_GO(HB, HA) do { \
    static bool attr = false; \
    if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_big_kernel<MT, HB, HA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); attr = true; } \
    hipLaunchKernelGGL((gemm_big_kernel<MT, HB, HA>), dim3(grid), dim3(BIG_THREADS), sh, s, p); } while (0)
  if (p.bias) { if (aux) BIG_GO(true, true); else BIG_GO(true, false); }
  else { if (aux) BIG_GO(false, true); else BIG_GO(false, false); }
#undef BIG_GO
}

// true = launched.  amode must be AM_DENSE, dtype bf16.
bool gemm_big_launch(const GemmP& g, hipStream_t s) {
  const char* mode_env = getenv("SATRN_GEMM_BIG");   // read per call (tests and tools switch it): 0 = off, 2 = take every shape that fits
  const int mode = mode_env ? atoi(mode_env) : 1;
  if (!mode) return false;
  if (g.stats || g.bnb_y || g.escale || g.eres || g.out_f32 || g.drop_p > 0.f) return false;
  if ((g.K % BIG_BK) != 0 || (g.lda & 7) || (g.ldc & 7) || (g.N & 7) || g.M < 1) return false;
  if ((size_t)g.M * g.lda * 2 >= (1ull << 31) || (size_t)g.N * g.K * 2 >= (1ull << 31) || (size_t)g.M * g.ldc * 2 >= (1ull << 31)) return false;   // 32-bit buffer offsets
  const double flops = 2.0 * g.M * g.N * g.K;
  if (mode != 2 && (flops < 2.0e9 || g.N < 128 || g.M < 2048)) return false;   // small products: the 4-wave tiles fill the chip better
  BigP p;
  p.A = (const bf16_t*)g.A; p.W = (const bf16_t*)g.Bw; p.C = (bf16_t*)g.C; p.bias = g.bias;
  p.pre_out = (bf16_t*)g.pre_out; p.bact_u = (const bf16_t*)g.bact_u;
  p.M = g.M; p.N = g.N; p.K = g.K; p.lda = g.lda; p.ldc = g.ldc; p.act = g.act; p.bact = g.bact; p.beta = g.beta; p.bact_scale = g.bact_scale;
  p.a_bytes = (unsigned)(((size_t)(g.M - 1) * g.lda + g.K) * 2);
  p.w_bytes = (unsigned)((size_t)g.N * g.K * 2);
  p.c_bytes = (unsigned)(((size_t)(g.M - 1) * g.ldc + g.N) * 2);
  p.ntn = (g.N + 127) / 128;
  const int cus = big_cu_count();
  // tile height: the candidate whose tile count leaves the smallest idle share in the last round of the persistent grid
  const int force_mt = getenv("SATRN_GEMM_BIG_MT") ? atoi(getenv("SATRN_GEMM_BIG_MT")) : 0;
  int best_mt = 4;
  double best_cost = 1e30;
  for (int mt = 4; mt >= 2; --mt) {
    const long tiles = (long)((g.M + 64 * mt - 1) / (64 * mt)) * p.ntn;
    const long rounds = (tiles + cus - 1) / cus;
    // cost ~ rounds x (rows per tile + a fixed per-tile part worth ~48 rows: epilogue + the wider share of W traffic of flat tiles)
    const double cost = (double)rounds * (64.0 * mt + 48.0);
    if (cost < best_cost) { best_cost = cost; best_mt = mt; }
  }
  if (force_mt >= 2 && force_mt <= 4) best_mt = force_mt;
  p.ntm = (g.M + 64 * best_mt - 1) / (64 * best_mt);
  p.ntiles = p.ntm * p.ntn;
  const int grid = p.ntiles < cus ? p.ntiles : cus;
  if (best_mt == 4) big_launch_t<4>(p, grid, s);
  else if (best_mt == 3) big_launch_t<3>(p, grid, s);
  else big_launch_t<2>(p, grid, s);
  return true;
}
