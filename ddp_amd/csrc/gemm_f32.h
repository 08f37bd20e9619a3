// gemm_f32.h - token-major fp32 GEMM on the gfx950 f32 matrix cores, with fused row epilogues.
//
//   C[m][n] = sum_k A[m][k] * W[n][k]          A: (M,K) row-major tokens x channels
//                                              W: (N,K) row-major (torch nn.Linear / 1x1-conv layout)
//
// Every contraction of the DDP decoder is this shape with K in {256,512,1024} and small N
// (transform, value/offset/attention projections, output_proj, FFN, conv_seg; SURVEY.md §2.6
// K1,K6,K9,K11,K14), so weights are L2-resident and the kernel is bound by the f32 MFMA pipe
// (v_mfma_f32_32x32x2_f32, 64 cycles, 157 TF chip peak).
//
// CDNA4 mapping
//  * block = 256 threads = 4 waves, one per SIMD; block tile = 128 tokens x (NT*32) channels;
//    wave w owns tokens [32w, 32w+32) and ALL NT channel tiles, so a token's whole output row
//    lives in one wave.
//  * operands are swapped relative to the textbook orientation: the MFMA "A" operand is the
//    weight tile (rows = output channels), the "B" operand is the activation tile (cols = tokens).
//    D[i][j] then has j = lane&31 = token, i = channel: each lane pair (l, l^32) holds one
//    token's row -> LayerNorm / softmax / argmax epilogues are in-register reductions plus a
//    single cross-half exchange, and stores are float4 along channels.
//  * both tiles sit in LDS K-contiguous with a 36-float row stride; lane (row=l&31, kh=l>>5)
//    reads 4 consecutive k with one ds_read_b128 (conflict-free: 36*row mod 64 is distinct for
//    the 16 rows of a b128 lane group) and feeds 4 MFMAs from it.  The k -> (instruction, half)
//    assignment is a free permutation of the reduction, identical for both operands.
//  * global->register prefetch of tile k+1 is issued before the MFMAs of tile k; LDS is single
//    buffered (55 KB at NT=8) so two blocks are resident per CU and cover each other's
//    barrier / epilogue bubbles.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

namespace ddp {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int GEMM_BM = 128;
constexpr int GEMM_BK = 32;
constexpr int GEMM_LDS = GEMM_BK + 4;   // padded row stride (floats)
constexpr int GEMM_THREADS = 256;
constexpr int EPI_ROW4 = 4 * 32 + 4;    // floats per row of the epilogue staging patch at full pass width

template <int NT>
constexpr size_t gemm_lds_bytes() {
  const size_t ring = size_t(GEMM_BM + NT * 32) * GEMM_LDS * sizeof(float);
  const size_t epi = size_t(4) * 32 * (4 * 32 + 4) * sizeof(float);
  return ring > epi ? ring : epi;
}

// Per-lane view handed to an epilogue: acc[t][r] is the value of token `m`, channel
//   n0 + t*32 + 8*(r>>2) + 4*kh + (r&3);   i.e. for g=r>>2 a float4 at channel n0+t*32+8g+4kh.
struct LaneCtx {
  int m;       // global token row of this lane (may be >= M: not valid)
  bool valid;  // m < M
  int n0;      // first channel of the block tile
  int kh;      // lane >> 5
  int lane;    // 0..63
  int m_base;  // first token row of this wave
  int M;       // number of rows
  float* patch;  // wave-private LDS staging patch (32 x EPI_ROW4 floats), free once the main loop is done
};

__device__ __forceinline__ float half_sum(float v) { return v + __shfl_xor(v, 32, 64); }

// TAG only names the call site (value_proj, fc1, ...) so that rocprofv3 reports each separately.
template <int NT, class Epi, int TAG>
__global__ void __launch_bounds__(GEMM_THREADS, 2)
k_gemm_tok(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw, int M, int N, int K,
           int n_tiles_n, Epi epi, int stagger, unsigned long long* dbg) {
#define DDP_STAMP(slot)                                                                       \
  if (dbg && threadIdx.x == 0) dbg[size_t(blockIdx.x) * 4 + (slot)] = __builtin_readcyclecounter();
  DDP_STAMP(0)
  if (dbg && threadIdx.x == 0) {   // probe: where did this block land?  (HW_ID: cu/sh/se, XCC_ID)
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    dbg[size_t(gridDim.x) * 4 + blockIdx.x] = (static_cast<unsigned long long>(xcc) << 32) | hw;
  }
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Ws = smem + GEMM_BM * GEMM_LDS;
  if (stagger > 0 && blockIdx.x >= 256 && blockIdx.x < 512)
    for (int i = 0; i < (stagger & 0xffff); ++i) __builtin_amdgcn_s_sleep(127);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int j = lane & 31;
  const int kh = lane >> 5;

  // block -> (token tile, channel tile).  Blocks that share a token tile (n_tiles_n > 1) are made
  // consecutive on ONE XCD (dispatch is round-robin over the 8 XCDs) so the A tile is fetched
  // into that XCD's L2 once.
  int mt, nt;
  {
    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    const int idx = bid >> 3;
    nt = idx % n_tiles_n;
    mt = (idx / n_tiles_n) * 8 + xcd;
  }
  const int m0 = mt * GEMM_BM;
  if (m0 >= M) return;
  const int n0 = nt * NT * 32;

  constexpr int A_F4 = GEMM_BM * GEMM_BK / 4 / GEMM_THREADS;   // 4
  constexpr int W_F4 = NT * 32 * GEMM_BK / 4 / GEMM_THREADS;   // NT
  f32x4 ra[A_F4];
  f32x4 rw[W_F4];
  const int lrow = tid >> 3;
  const int lkq = (tid & 7) * 4;

  const float* a_src[A_F4];
  const float* w_src[W_F4];
#pragma unroll
  for (int p = 0; p < A_F4; ++p) {
    int gm = m0 + lrow + 32 * p;
    gm = gm < M ? gm : M - 1;
    a_src[p] = A + size_t(gm) * lda + lkq;
  }
#pragma unroll
  for (int p = 0; p < W_F4; ++p) {
    int gn = n0 + lrow + 32 * p;
    gn = gn < N ? gn : N - 1;
    w_src[p] = W + size_t(gn) * ldw + lkq;
  }

  auto gload = [&](int kt) {
#pragma unroll
    for (int p = 0; p < A_F4; ++p) ra[p] = *reinterpret_cast<const f32x4*>(a_src[p] + kt * GEMM_BK);
#pragma unroll
    for (int p = 0; p < W_F4; ++p) rw[p] = *reinterpret_cast<const f32x4*>(w_src[p] + kt * GEMM_BK);
  };
  auto sstore = [&]() {
#pragma unroll
    for (int p = 0; p < A_F4; ++p)
      *reinterpret_cast<f32x4*>(As + (lrow + 32 * p) * GEMM_LDS + lkq) = ra[p];
#pragma unroll
    for (int p = 0; p < W_F4; ++p)
      *reinterpret_cast<f32x4*>(Ws + (lrow + 32 * p) * GEMM_LDS + lkq) = rw[p];
  };

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  gload(0);
  sstore();
  __syncthreads();
  DDP_STAMP(1)

  const float* ap = As + (wave * 32 + j) * GEMM_LDS + 4 * kh;
  const float* wp = Ws + j * GEMM_LDS + 4 * kh;
  const int nk = K / GEMM_BK;
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = (kt + 1 < nk);
    if (more) gload(kt + 1);
#pragma unroll
    for (int c = 0; c < GEMM_BK / 8; ++c) {
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(ap + 8 * c);
      f32x4 w4[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) w4[t] = *reinterpret_cast<const f32x4*>(wp + t * 32 * GEMM_LDS + 8 * c);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < NT; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[t][s], a4[s], acc[t], 0, 0, 0);
    }
    __syncthreads();
    if (more) {
      sstore();
      __syncthreads();
    }
  }

  LaneCtx cx;
  cx.m = m0 + wave * 32 + j;
  cx.valid = cx.m < M;
  cx.n0 = n0;
  cx.kh = kh;
  cx.lane = lane;
  cx.m_base = m0 + wave * 32;
  cx.M = M;
  cx.patch = smem + wave * 32 * EPI_ROW4;      // all waves are past the last LDS read (trailing barrier)
  DDP_STAMP(2)
  epi.template run<NT>(acc, cx);
  DDP_STAMP(3)
}

// ------------------------------------------------------------------------------------------------
// v2 main loop: neither operand is staged through VGPRs.
//  * activation operand: every wave owns its 32 tokens exclusively, so lane (token j, half kh) loads
//    its own MFMA-B fragments A[m][kt*32 + 8c + 4kh .. +3] straight from global memory into registers
//    (4 x b128 per k-tile; the 32-B pieces of a 128-B line are requested back to back and merge in
//    L1), one tile ahead of use.  A never touches LDS and no barrier guards it.
//  * weight operand: LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction, no VGPR
//    destination, no ds_write pass), double buffered, issued one tile ahead so it lands in the shadow
//    of 128 MFMAs; ONE barrier per k-tile.  The LDS image is lane-linear (8 rows x 128 B per
//    instruction), so the bank-conflict fix is an XOR swizzle applied to the per-lane SOURCE address
//    and to the read address: 16-B chunk q of row r lives at position q ^ ((r>>1)&7); the 16 rows of a
//    ds_read_b128 lane group then hit 16 distinct 16-B slots of the 256-B bank row.
//  LDS: 2 x NT*32 x 128 B = 64 KB at NT = 8 -> two blocks per CU.
// ------------------------------------------------------------------------------------------------
template <int NT>
constexpr size_t gemm2_lds_bytes() {
  const size_t ring = size_t(2) * NT * 32 * GEMM_BK * sizeof(float);
  const size_t epi = size_t(4) * 32 * (4 * 32 + 4) * sizeof(float);
  return ring > epi ? ring : epi;
}

typedef __attribute__((address_space(3))) float lds_float_t;

// One LDS-DMA piece: 64 lanes x 16 B from global (uniform base + per-lane byte offset) to
// LDS[lds_dst + 16*lane].  Inline asm because hipcc drains vmcnt(0) in front of any ds_read that might
// alias a builtin LDS-DMA (seen in the ISA: the whole prefetch was waited for before the first MFMA of a
// tile).  hipcc does not count these loads; their completion is waited for with the s_waitcnt BUILTIN
// (which also resets hipcc's own scoreboard) before the barrier that publishes the stage.  M0 (the DMA's
// LDS base) is compiler-reserved: saved and restored inside the statement.
__device__ __forceinline__ void lds_dma16(const float* gbase, unsigned byte_off, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(byte_off), "s"(gbase), "s"(lds_dst)
      : "memory");
}
// s_waitcnt vmcnt(0) (expcnt/lgkmcnt untouched): gfx9 encoding vm[3:0]=0, exp=7, lgkm=15, vm[5:4]=0
__device__ __forceinline__ void wait_vm0() { __builtin_amdgcn_s_waitcnt(0x0F70); }

template <int NT, class Epi, int TAG>
__global__ void __launch_bounds__(GEMM_THREADS, 2)
k_gemm_tok2(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw, int M, int N, int K,
            int n_tiles_n, Epi epi, int stagger, unsigned long long* dbg) {
  DDP_STAMP(0)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int WTILE = NT * 32 * GEMM_BK;   // floats per stage (unpadded, swizzled)
  // Phase offset between the two blocks that share a CU.  All blocks of a launch start together and take
  // equally long, so without it every CU alternates chip-wide between an MFMA-only phase (main loops) and
  // a memory-only phase (epilogue stores + next prologue): measured 0.16 ms of a 0.40 ms K=256 launch.
  // The second block of each CU (blocks 256..511 under the observed dispatch order; a wrong guess only
  // costs speed) sleeps for about half a tile, after which the hardware keeps back-filling freed slots
  // out of phase: one block's epilogue traffic hides under its neighbour's MFMAs.
  if (stagger > 0 && blockIdx.x < 512) {
    const int mode = stagger >> 16, n = stagger & 0xffff;     // probe: which blocks of the first wave sleep
    const bool hit = mode == 0 ? blockIdx.x >= 256 : mode == 1 ? ((blockIdx.x >> 3) & 1) : ((blockIdx.x >> 8) ^ (blockIdx.x >> 3)) & 1;
    if (hit)
      for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
  }

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31;
  const int kh = lane >> 5;

  int mt, nt;
  {
    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    const int idx = bid >> 3;
    nt = idx % n_tiles_n;
    mt = (idx / n_tiles_n) * 8 + xcd;
  }
  const int m0 = mt * GEMM_BM;
  if (m0 >= M) return;
  const int n0 = nt * NT * 32;

  // weight DMA: wave-instruction p of wave w fills rows [(4p+w)*8, +8) of the stage; lane -> (row r,
  // slot lane&7) and fetches logical chunk q = slot ^ ((r>>1)&7)  (q is the same for every p).
  unsigned w_off[NT];
  {
    const int rl = wave * 8 + (lane >> 3);                 // row within a 32-row group
    const int q = (lane & 7) ^ ((rl >> 1) & 7);
#pragma unroll
    for (int p = 0; p < NT; ++p) {
      int gn = n0 + 32 * p + rl;
      gn = gn < N ? gn : N - 1;
      w_off[p] = unsigned(gn * ldw + 4 * q) * 4u;
    }
  }
  const unsigned lds0 = (unsigned)(size_t)(lds_float_t*)smem;   // LDS byte address of the carve
  auto dma_w = [&](int kt, int stage) {
    const float* wk = W + kt * GEMM_BK;                        // uniform
#pragma unroll
    for (int p = 0; p < NT; ++p)
      lds_dma16(wk, w_off[p], lds0 + unsigned(stage * WTILE + (4 * p + wave) * 8 * GEMM_BK) * 4u);
  };

  // activation fragments: this lane's token row
  int gm = m0 + wave * 32 + j;
  const bool valid = gm < M;
  gm = valid ? gm : M - 1;
  const float* a_src = A + size_t(gm) * lda + 4 * kh;
  f32x4 a_cur[GEMM_BK / 8], a_nxt[GEMM_BK / 8];
  auto gload_a = [&](f32x4 (&dst)[GEMM_BK / 8], int kt) {
#pragma unroll
    for (int c = 0; c < GEMM_BK / 8; ++c) dst[c] = *reinterpret_cast<const f32x4*>(a_src + kt * GEMM_BK + 8 * c);
  };

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int nk = K / GEMM_BK;      // even (host checks K % 64 == 0)
  dma_w(0, 0);
  gload_a(a_cur, 0);
  wait_vm0();
#pragma unroll
  for (int c = 0; c < GEMM_BK / 8; ++c) asm volatile("" : "+v"(a_cur[c]));
  __syncthreads();

  // read side of the swizzle: lane (row j of a 32-row channel tile, half kh), chunk q = 2c + kh
  const int swz = (j >> 1) & 7;
  const float* wrow = smem + j * GEMM_BK;
  auto frag = [&](int st, int c, int t) -> f32x4 {
    return *reinterpret_cast<const f32x4*>(wrow + st * WTILE + t * 32 * GEMM_BK + ((2 * c + kh) ^ swz) * 4);
  };
  f32x4 w4[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) w4[t] = frag(0, 0, t);
  DDP_STAMP(1)

  // One k-tile = 4 chunks x 4 k-pairs x NT MFMAs, software pipelined:
  //  * the weight fragment register w4[t] is refilled for the NEXT chunk right after its last use (k-pair
  //    s = 3), so its LDS latency hides under the remaining MFMAs of the pass and no second register set
  //    is needed;
  //  * the LDS-DMA of W(kt+1) and the register prefetch of A(kt+1) are issued in the shadow of chunk 0's
  //    MFMAs (two DMA pieces + one A load per 8-MFMA pass);
  //  * the ONE barrier of the tile sits before the last pass (chunk 3, s = 3): by then every wave has
  //    consumed stage `st` from LDS, and the refills of that last pass already read the NEXT stage, so the
  //    next tile starts without an LDS round trip.
  // Tile indices are clamped instead of branched so that every wave issues the same loads; the redundant
  // traffic of the last tile touches only dead data.
  auto step = [&](f32x4 (&ac)[GEMM_BK / 8], f32x4 (&an)[GEMM_BK / 8], int kt, int st) {
    const int k1 = kt + 1 < nk ? kt + 1 : nk - 1;
    const float* wk = W + k1 * GEMM_BK;
#pragma unroll
    for (int c = 0; c < GEMM_BK / 8; ++c) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (c == GEMM_BK / 8 - 1 && s == 3) {
          wait_vm0();                          // this wave's DMA pieces of stage st^1 and A(kt+1) have landed
          // launder the prefetched fragments: hipcc's scoreboard then considers them landed HERE (it cannot
          // see that wait_vm0 covered them and would emit a counted vmcnt in front of the next tile's first
          // MFMA, which - with the uncounted DMA pieces queued - stalls on the fresh prefetch), and it can
          // no longer rematerialise the loads at the loop head.
#pragma unroll
          for (int cc = 0; cc < GEMM_BK / 8; ++cc) asm volatile("" : "+v"(an[cc]));
          __syncthreads();                     // everyone's pieces have landed; stage st is free again
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[t][s], ac[c][s], acc[t], 0, 0, 0);
          if (s == 3) w4[t] = (c < GEMM_BK / 8 - 1) ? frag(st, c + 1, t) : frag(st ^ 1, 0, t);
        }
        if (c == 0) {
          // NT DMA pieces spread over the 4 passes of chunk 0
#pragma unroll
          for (int p = s * ((NT + 3) / 4); p < (s + 1) * ((NT + 3) / 4) && p < NT; ++p)
            lds_dma16(wk, w_off[p], lds0 + unsigned((st ^ 1) * WTILE + (4 * p + wave) * 8 * GEMM_BK) * 4u);
          an[s] = *reinterpret_cast<const f32x4*>(a_src + k1 * GEMM_BK + 8 * s);
        }
      }
    }
  };
  const int nk_run = (stagger == -2) ? 0 : nk;      // probe: -2 skips the main loop
  for (int kt = 0; kt < nk_run; kt += 2) {
    step(a_cur, a_nxt, kt, 0);
    step(a_nxt, a_cur, kt + 1, 1);
  }
  if (stagger == -1) {                               // probe: -1 skips the epilogue (keep acc alive)
#if defined(__HIP_DEVICE_COMPILE__)     // (the host pass cannot parse the "v" constraint of a 512-bit operand)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[t][r]));
#endif
    return;
  }

  LaneCtx cx;
  cx.m = m0 + wave * 32 + j;
  cx.valid = valid;
  cx.n0 = n0;
  cx.kh = kh;
  cx.lane = lane;
  cx.m_base = m0 + wave * 32;
  cx.M = M;
  // every wave passed the tile's last barrier with all of its real fragment reads done (the refills of
  // the last pass read dead prefetch data), so the ring can be reused as wave-private staging patches
  cx.patch = smem + wave * 32 * EPI_ROW4;
  DDP_STAMP(2)
  if (stagger & 0x10000) __builtin_amdgcn_s_setprio(3);   // probe: epilogue first
  epi.template run<NT>(acc, cx);
  DDP_STAMP(3)
}

// number of blocks for the XCD-grouped 1-D grid used by k_gemm_tok
inline int gemm_grid(int M, int n_tiles_n) {
  int mt = (M + GEMM_BM - 1) / GEMM_BM;
  int mt8 = (mt + 7) / 8 * 8;
  return mt8 * n_tiles_n;
}

// ------------------------------------------------------------------------------------------------
// Epilogue plumbing: tile <-> row-major transfers through a wave-private LDS patch.
//
// In the MFMA D layout a lane holds 4 consecutive channels of ONE token per (tile, g).  Two things
// follow.  (1) Storing that straight to global memory touches 32 rows x 32 B per instruction.
// (2) Every per-element operation written against the accumulator registers is fully unrolled (register
// arrays cannot be indexed at run time): the first version of these epilogues compiled to 21 000
// instructions (168 KB) per kernel - 2.6x the instruction cache - and the profile showed the epilogue
// of one block (11-46 us per 128 KB tile) slowing the MFMA loop of its CU neighbour as well.
//
// So the epilogues are split: register-only work that needs the D layout (LayerNorm statistics) stays
// unrolled but tiny; everything per-element (bias, addend rows, GELU, affine, FiLM, bounds) runs in a
// ROLLED loop over row-major float4 slots read back from LDS, where a lane's channel is loop-invariant
// and global accesses are 512-B contiguous row segments.  A pass covers up to 4 channel tiles
// (128 channels); the patch is private to the wave (32 rows x (32*W+4) floats: ds_write_b128 of the D
// layout and the row-major ds_read_b128 are both bank-conflict free), so only in-order LDS execution
// within the wave is relied on, no barrier.
// ------------------------------------------------------------------------------------------------

// D layout -> patch (Wt tiles starting at tile T0)
template <int NT, int T0, int Wt>
__device__ __forceinline__ void patch_put(const f32x16 (&acc)[NT], float* patch, int lane) {
  constexpr int ROW = Wt * 32 + 4;
  const int j = lane & 31, kh = lane >> 5;
#pragma unroll
  for (int t = 0; t < Wt; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v = {acc[T0 + t][4 * g], acc[T0 + t][4 * g + 1], acc[T0 + t][4 * g + 2], acc[T0 + t][4 * g + 3]};
      *reinterpret_cast<f32x4*>(patch + j * ROW + t * 32 + 8 * g + 4 * kh) = v;
    }
}
// patch -> D layout, accumulate
template <int NT, int T0, int Wt>
__device__ __forceinline__ void patch_add(f32x16 (&acc)[NT], const float* patch, int lane) {
  constexpr int ROW = Wt * 32 + 4;
  const int j = lane & 31, kh = lane >> 5;
#pragma unroll
  for (int t = 0; t < Wt; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(patch + j * ROW + t * 32 + 8 * g + 4 * kh);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[T0 + t][4 * g + e] += v[e];
    }
}

// Rolled loop over the row-major float4 slots of a pass, U rows per trip:
//   pre(row, col)              -> f32x4   global loads of the slot (issued for all U rows first, so U
//                                         independent requests are in flight: with one load -> use -> store
//                                         chain per trip the loop ran at one memory latency per row)
//   fin(row, col, slot, pre)   -> void    LDS read, math, global store
// row in [0,32), col = first of the slot's 4 channels within the pass; LPR lanes cover one row, so for
// LPR | 64 a lane's `col` is loop invariant (per-channel vectors can be hoisted by the caller).
template <int Wt, int U, class Pre, class Fin>
__device__ __forceinline__ void patch_rows(float* patch, int lane, Pre pre, Fin fin) {
  constexpr int ROW = Wt * 32 + 4;
  constexpr int LPR = Wt * 8;
  constexpr int ITERS = (32 * LPR) / 64;
  static_assert(ITERS % U == 0, "trip count must divide");
  if constexpr (64 % LPR == 0) {
    constexpr int RPI = 64 / LPR;                       // rows per iteration
    const int r0 = lane / LPR, col = (lane % LPR) * 4;
#pragma unroll 1
    for (int it = 0; it < ITERS; it += U) {
      f32x4 ld[U];
#pragma unroll
      for (int u = 0; u < U; ++u) ld[u] = pre((it + u) * RPI + r0, col);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int row = (it + u) * RPI + r0;
        fin(row, col, patch + row * ROW + col, ld[u]);
      }
    }
  } else {
#pragma unroll 1
    for (int it = 0; it < ITERS; it += U) {
      f32x4 ld[U];
      int rows[U], cols[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = (it + u) * 64 + lane;
        rows[u] = idx / LPR;
        cols[u] = (idx - rows[u] * LPR) * 4;
        ld[u] = pre(rows[u], cols[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) fin(rows[u], cols[u], patch + rows[u] * ROW + cols[u], ld[u]);
    }
  }
}

// run `body<T0,Wt>()` for every pass of an NT-tile row block
template <int NT, class Body>
__device__ __forceinline__ void for_each_pass(Body body) {
  if constexpr (NT >= 4) body(std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{});
  if constexpr (NT == 8) body(std::integral_constant<int, 4>{}, std::integral_constant<int, 4>{});
  if constexpr (NT == 5) body(std::integral_constant<int, 4>{}, std::integral_constant<int, 1>{});
  if constexpr (NT < 4) body(std::integral_constant<int, 0>{}, std::integral_constant<int, NT>{});
}

// ------------------------------------------------------------------------------------------------
// Epilogues
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_erf(float x) { return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f)); }

__device__ __forceinline__ f32x4 load4_guard(const float* p, int ch, int n_valid) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (ch + 4 <= n_valid) {
    v = *reinterpret_cast<const f32x4*>(p);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (ch + e < n_valid) v[e] = p[e];
  }
  return v;
}
__device__ __forceinline__ void store4_guard(float* p, const f32x4& v, int ch, int n_valid) {
  if (ch + 4 <= n_valid) {
    *reinterpret_cast<f32x4*>(p) = v;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (ch + e < n_valid) p[e] = v[e];
  }
}

// out[m][n] = acc + bias[n] (+ add[row map(m)][n]) ; optional exact GELU.
// Row map of the addend: the tokens of all r noise replicas of one image share the image's x
// projection: add_row = (m / (r*N)) * N + m % N  (rn = r*N; rn == 0 -> identity).
struct EpiBias {
  const float* bias;  // (N) or nullptr
  const float* add;   // (rows, ld_add) or nullptr
  int ld_add;
  int rn, n_tok;
  float* out;
  int ldo;
  int n_valid;        // channels >= n_valid are not stored
  int gelu;

  template <int NT>
  __device__ __forceinline__ void run(f32x16 (&acc)[NT], const LaneCtx& cx) const {
    for_each_pass<NT>([&](auto t0c, auto wtc) {
      constexpr int T0 = decltype(t0c)::value, Wt = decltype(wtc)::value;
      constexpr int U = (Wt == 3) ? 4 : 4;
      patch_put<NT, T0, Wt>(acc, cx.patch, cx.lane);
      const int cbase = cx.n0 + T0 * 32;
      patch_rows<Wt, U>(
          cx.patch, cx.lane,
          [&](int row, int col) -> f32x4 {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            int m = cx.m_base + row;
            m = m < cx.M ? m : cx.M - 1;
            const int ch = cbase + col;
            if (ch < n_valid) {
              if (bias) v = load4_guard(bias + ch, ch, n_valid);
              if (add) {
                const size_t ar = rn ? size_t(m / rn) * n_tok + m % n_tok : size_t(m);
                v += load4_guard(add + ar * ld_add + ch, ch, n_valid);
              }
            }
            return v;
          },
          [&](int row, int col, float* slot, const f32x4& pre) {
            const int m = cx.m_base + row, ch = cbase + col;
            if (m >= cx.M || ch >= n_valid) return;
            f32x4 v = *reinterpret_cast<const f32x4*>(slot) + pre;
            if (gelu) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
            }
            store4_guard(out + size_t(m) * ldo + ch, v, ch, n_valid);
          });
    });
  }
};

// y = acc + bias + res[m];  out = LayerNorm_256(y) * gamma + beta;  optional FiLM
// out = out * (scale + 1) + shift   (utils/transformer.py:390-392,413-417).  Needs NT == 8, n0 == 0.
struct EpiResLN {
  const float* bias;
  const float* res;
  int ldres;
  const float* gamma;
  const float* beta;
  const float* film;  // scale[256] | shift[256], or nullptr
  float* out;
  int ldo;

  template <int NT>
  __device__ __forceinline__ void run(f32x16 (&acc)[NT], const LaneCtx& cx) const {
    static_assert(NT == 8, "LayerNorm epilogue needs the full 256-channel row");
    // 1. y = acc + bias + residual: residual rows are read row-major (rolled), transposed via the patch
    for_each_pass<NT>([&](auto t0c, auto wtc) {
      constexpr int T0 = decltype(t0c)::value, Wt = decltype(wtc)::value;
      const f32x4 b = *reinterpret_cast<const f32x4*>(bias + T0 * 32 + (cx.lane % (Wt * 8)) * 4);
      patch_rows<Wt, 8>(
          cx.patch, cx.lane,
          [&](int row, int col) -> f32x4 {
            int m = cx.m_base + row;
            m = m < cx.M ? m : cx.M - 1;
            return *reinterpret_cast<const f32x4*>(res + size_t(m) * ldres + T0 * 32 + col);
          },
          [&](int row, int col, float* slot, const f32x4& r) { *reinterpret_cast<f32x4*>(slot) = r + b; });
      patch_add<NT, T0, Wt>(acc, cx.patch, cx.lane);
    });
    // 2. row statistics in the D layout: a token's 256 values live in the lane pair (l, l^32)
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[t][r];
    const float mean = half_sum(s) * (1.0f / 256.0f);
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d = acc[t][r] - mean;
        acc[t][r] = d;
        q += d * d;
      }
    const float rstd = 1.0f / sqrtf(half_sum(q) * (1.0f / 256.0f) + 1e-5f);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] *= rstd;
    // 3. affine (+FiLM) and store, row-major and rolled (channel of a lane is loop invariant)
    for_each_pass<NT>([&](auto t0c, auto wtc) {
      constexpr int T0 = decltype(t0c)::value, Wt = decltype(wtc)::value;
      patch_put<NT, T0, Wt>(acc, cx.patch, cx.lane);
      const int chl = T0 * 32 + (cx.lane % (Wt * 8)) * 4;
      const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + chl);
      const f32x4 be = *reinterpret_cast<const f32x4*>(beta + chl);
      f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = {0.f, 0.f, 0.f, 0.f};
      if (film) {
        sc = *reinterpret_cast<const f32x4*>(film + chl);
        sh = *reinterpret_cast<const f32x4*>(film + 256 + chl);
      }
      patch_rows<Wt, 4>(
          cx.patch, cx.lane, [](int, int) -> f32x4 { return f32x4{0.f, 0.f, 0.f, 0.f}; },
          [&](int row, int col, float* slot, const f32x4&) {
            const int m = cx.m_base + row;
            if (m >= cx.M) return;
            f32x4 v = *reinterpret_cast<const f32x4*>(slot);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[e] = v[e] * ga[e] + be[e];
              if (film) v[e] = v[e] * (sc[e] + 1.0f) + sh[e];
            }
            *reinterpret_cast<f32x4*>(out + size_t(m) * ldo + T0 * 32 + col) = v;
          });
    });
  }
};

// Sampling projection epilogue (multi_scale_deform_attn.py:319-334 for one level).
// Input columns: 0..63 offsets [head][point][x,y], 64..95 attention logits [head][point].
//   raw = acc + PY[i][col] + PX[j][col]        (positional term folded through the projection:
//                                               W(q+pos) = Wq + W_y pos_y(i) + W_x pos_x(j); bias in PY)
//   offsets -> pixel-unit sample coordinates  x = j + o_x, y = i + o_y
//   logits  -> softmax over the 4 points of a head (4 consecutive columns = one float4 slot)
struct EpiSamp {
  const float* py;  // (h, 96)
  const float* px;  // (w, 96)
  int n_tok, w;     // tokens per map, map width
  float* out;       // (M, 96)

  template <int NT>
  __device__ __forceinline__ void run(f32x16 (&acc)[NT], const LaneCtx& cx) const {
    static_assert(NT == 3, "sampling projection is 96 columns");
    patch_put<NT, 0, 3>(acc, cx.patch, cx.lane);
    patch_rows<3, 4>(
        cx.patch, cx.lane,
        [&](int row, int col) -> f32x4 {
          int m = cx.m_base + row;
          m = m < cx.M ? m : cx.M - 1;
          const int n = m % n_tok;
          const int i = n / w;
          const int jx = n - i * w;
          return *reinterpret_cast<const f32x4*>(py + i * 96 + col) + *reinterpret_cast<const f32x4*>(px + jx * 96 + col);
        },
        [&](int row, int col, float* slot, const f32x4& pos) {
      const int m = cx.m_base + row;
      if (m >= cx.M) return;
      const int n = m % n_tok;
      const int i = n / w;
      const int jx = n - i * w;
      f32x4 v = *reinterpret_cast<const f32x4*>(slot) + pos;
      if (col < 64) {
        const float fi = float(i), fj = float(jx);
        v[0] += fj; v[1] += fi; v[2] += fj; v[3] += fi;
      } else {
        const float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = expf(v[e] - mx);
        const float den = v[0] + v[1] + v[2] + v[3];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] /= den;
      }
      *reinterpret_cast<f32x4*>(out + size_t(m) * 96 + col) = v;
    });
  }
};

}  // namespace ddp
