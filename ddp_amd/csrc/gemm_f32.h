// gemm_f32.h - token GEMM on the gfx950 f32 matrix cores, with fused row epilogues.
//
//   C[m][n] = sum_k A[m][k] * W[n][k]          A: tokens x channels (row-major or fragment-major)
//                                              W: (N,K) row-major (torch nn.Linear / 1x1-conv layout)
//
// Every contraction of the DDP decoder is this shape with K in {256,512,1024} and small N (transform,
// value/offset/attention projections, output_proj, FFN, conv_seg; SURVEY.md §2.6 K1,K6,K9,K11,K14):
// weights are L2 resident and the kernel is bound by the f32 MFMA pipe (v_mfma_f32_32x32x2_f32, 64
// cycles, 157 TF chip peak).
//
// What the hardware dictates (measured on MI355X, scripts/ubench/mfma_valu.hip and the stamp probes):
//  * an f32-input MFMA runs on the SIMD's own FP32 datapath: while one wave streams MFMAs back to back,
//    the VALU instructions of every other wave on that SIMD make ZERO progress (a co-resident VALU wave
//    finished exactly one MFMA-stream later).  LDS, vector-memory and scalar instructions do issue.  So a
//    second block per CU hides memory latency but no arithmetic: every VALU instruction of a prologue or
//    epilogue is MFMA time lost, and the design rule is "no VALU that is not in the reference's math".
//  * hence: bias is folded into the accumulator initialisation; activations that only feed the next GEMM
//    live in HBM in FRAGMENT-MAJOR order (below) so that epilogue stores, residual reads and the next
//    kernel's operand loads are 1-KiB coalesced accesses at immediate offsets - no address arithmetic,
//    no LDS transposition, no bounds checks (buffers are padded to whole 128-token tiles);
//    row-major outputs (needed by the gather / sampler kernels) go through a wave-private LDS patch in
//    ROLLED loops (an earlier fully unrolled per-element epilogue was 21 000 instructions = 168 KB of
//    code, 2.6x the instruction cache).
//
// CDNA4 mapping
//  * block = 256 threads = 4 waves, one per SIMD; block tile = 128 tokens x (NT*32) channels; wave w owns
//    tokens [32w, 32w+32) and ALL NT channel tiles: a token's whole output row lives in one wave.
//  * operands are swapped relative to the textbook orientation: the MFMA "A" operand is the weight tile
//    (rows = output channels), the "B" operand is the activation tile (cols = tokens).  D[i][j] then has
//    j = lane&31 = token, i = channel: the lane pair (l, l^32) holds one token's row, LayerNorm is an
//    in-register reduction plus ONE cross-half exchange, and a lane's accumulator quad is 4 consecutive
//    channels.
//  * activation operand: every wave owns its tokens exclusively, so lane (token j, half kh) loads its
//    own MFMA-B fragments (4 consecutive k per b128, feeding 4 MFMAs; the k -> (instruction, half)
//    assignment is a free permutation of the reduction) straight from global memory into registers, one
//    k-tile ahead.  It never touches LDS and no barrier guards it.
//  * weight operand: LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction, no VGPR destination,
//    no ds_write pass), double buffered, issued one tile ahead in the shadow of 128 MFMAs; ONE barrier
//    per k-tile.  The LDS image is lane-linear, so the bank-conflict fix is an XOR swizzle applied to
//    the per-lane SOURCE address and to the read address: 16-B chunk q of row r lives at slot
//    q ^ ((r>>1)&7); the 16 rows of a ds_read_b128 lane group hit 16 distinct slots of the bank row.
//  * the k-loop is software pipelined by hand (see `step`): weight fragments are refilled right after
//    their last use, prefetches are issued under chunk 0's MFMAs, and the tile's barrier sits before its
//    last MFMA pass so the next tile starts without an LDS round trip.
//
// Fragment-major ("blocked") activation layout, C channels (C % 32 == 0), rows padded to 128:
//   element (m, ch) lives at float offset
//     (m/32)*32*C + (ch/32)*1024 + ((ch%32)/8)*256 + ((ch%8)/4)*128 + (m%32)*4 + ch%4
//   i.e. per 32-token group [tile t][quad g][half kh][token j][4 channels]: exactly the order in which a
//   wave holds a 32x32 D tile (reg quad g of lane kh*32+j), and exactly the order in which the next
//   GEMM's lane (j,kh) wants its B fragments for k-tile t, chunk g.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

namespace ddp {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int GEMM_BM = 128;
constexpr int GEMM_BK = 32;
constexpr int GEMM_THREADS = 256;
constexpr int EPI_ROW4 = 4 * 32 + 4;    // floats per row of the epilogue staging patch at full pass width

template <int NT>
constexpr size_t gemm_ring_bytes() {
  return size_t(2) * NT * 32 * GEMM_BK * sizeof(float);
}
constexpr size_t gemm_patch_bytes() { return size_t(4) * 32 * EPI_ROW4 * sizeof(float); }
template <int NT, class Epi>
constexpr size_t gemm_lds_bytes() {
  return (Epi::kNeedsPatch && gemm_patch_bytes() > gemm_ring_bytes<NT>()) ? gemm_patch_bytes() : gemm_ring_bytes<NT>();
}

// rows of a blocked buffer are padded to whole block tiles
__host__ __device__ inline size_t blk_rows(size_t m) { return (m + GEMM_BM - 1) / GEMM_BM * GEMM_BM; }

struct GemmArgs {
  const float* A;        // row-major (lda) or fragment-major (K channels)
  int lda;
  const float* W;        // (N, K) row-major, row stride ldw
  int ldw;
  int M, N, K;
  int n_tiles_n;         // number of NT*32-channel tiles
  const float* acc_bias; // (N) folded into the accumulator initialisation, or nullptr
};

// Per-lane view handed to an epilogue: acc[t][r] is the value of token `m`, channel
//   n0 + t*32 + 8*(r>>2) + 4*kh + (r&3);   i.e. for g=r>>2 a float4 at channel n0+t*32+8g+4kh.
struct LaneCtx {
  int m;         // global token row of this lane (may be >= M: not valid)
  bool valid;    // m < M
  int n0;        // first channel of the block tile
  int kh;        // lane >> 5
  int lane;      // 0..63
  int m_base;    // first token row of this wave
  int M;         // number of valid rows
  float* patch;  // wave-private LDS staging patch (32 x EPI_ROW4 floats), free once the main loop is done
};

__device__ __forceinline__ float half_sum(float v) { return v + __shfl_xor(v, 32, 64); }

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

// TAG only names the call site (value_proj, fc1, ...) so that rocprofv3 reports each separately.
template <int NT, bool A_BLK, class Epi, int TAG>
__global__ void __launch_bounds__(GEMM_THREADS, 2)
k_gemm_tok(GemmArgs ga, Epi epi, unsigned long long* dbg) {
#define DDP_STAMP(slot) \
  if (dbg && threadIdx.x == 0) dbg[size_t(blockIdx.x) * 4 + (slot)] = __builtin_readcyclecounter();
  DDP_STAMP(0)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int WTILE = NT * 32 * GEMM_BK;   // floats per stage (unpadded, swizzled)
  constexpr int NC = GEMM_BK / 8;            // 8-wide k chunks per k-tile

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31;
  const int kh = lane >> 5;
  const int M = ga.M, N = ga.N;

  // block -> (token tile, channel tile).  Blocks that share a token tile (n_tiles_n > 1) are made
  // consecutive on ONE XCD (dispatch is round-robin over the 8 XCDs) so the A tile is fetched into that
  // XCD's L2 once.
  int mt, nt;
  {
    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    const int idx = bid >> 3;
    nt = idx % ga.n_tiles_n;
    mt = (idx / ga.n_tiles_n) * 8 + xcd;
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
      w_off[p] = unsigned(gn * ga.ldw + 4 * q) * 4u;
    }
  }
  const unsigned lds0 = (unsigned)(size_t)(lds_float_t*)smem;   // LDS byte address of the carve

  // activation fragments of this lane: chunk c of k-tile kt
  const float* a_src;
  if constexpr (A_BLK) {
    a_src = ga.A + (size_t(m0 >> 5) + wave) * 32 * ga.K + lane * 4;        // + kt*1024 + c*256
  } else {
    int gm = m0 + wave * 32 + j;
    gm = gm < M ? gm : M - 1;
    a_src = ga.A + size_t(gm) * ga.lda + 4 * kh;                           // + kt*32 + c*8
  }
  auto a_frag = [&](int kt, int c) -> f32x4 {
    if constexpr (A_BLK) return *reinterpret_cast<const f32x4*>(a_src + kt * 1024 + c * 256);
    else return *reinterpret_cast<const f32x4*>(a_src + kt * GEMM_BK + c * 8);
  };

  // accumulators start at the bias (one less VALU pass in the epilogue)
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int ch = n0 + t * 32 + 8 * g + 4 * kh;
      f32x4 b = {0.f, 0.f, 0.f, 0.f};
      if (ga.acc_bias) {
        if (ch + 4 <= N) {
          b = *reinterpret_cast<const f32x4*>(ga.acc_bias + ch);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (ch + e < N) b[e] = ga.acc_bias[ch + e];
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[t][4 * g + e] = b[e];
    }

  const int nk = ga.K / GEMM_BK;
  f32x4 a_cur[NC], a_nxt[NC];
  {
    const float* wk = ga.W;
#pragma unroll
    for (int p = 0; p < NT; ++p) lds_dma16(wk, w_off[p], lds0 + unsigned((4 * p + wave) * 8 * GEMM_BK) * 4u);
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) a_cur[c] = a_frag(0, c);
  wait_vm0();
#pragma unroll
  for (int c = 0; c < NC; ++c) asm volatile("" : "+v"(a_cur[c]));
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
  //    MFMAs (NT/4 DMA pieces + one A load per 8-MFMA pass);
  //  * the ONE barrier of the tile sits before the last pass (chunk 3, s = 3): by then every wave has
  //    consumed stage `st` from LDS, and the refills of that last pass already read the NEXT stage, so the
  //    next tile starts without an LDS round trip.
  // Tile indices are clamped instead of branched so that every wave issues the same loads; the redundant
  // traffic of the last tile touches only dead data.
  auto step = [&](f32x4 (&ac)[NC], f32x4 (&an)[NC], int kt, int st) {
    const int k1 = kt + 1 < nk ? kt + 1 : nk - 1;
    const float* wk = ga.W + k1 * GEMM_BK;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (c == NC - 1 && s == 3) {
          wait_vm0();                          // this wave's DMA pieces of stage st^1 and A(kt+1) have landed
          // launder the prefetched fragments: hipcc's scoreboard then considers them landed HERE (it cannot
          // see that wait_vm0 covered them and would emit a counted vmcnt in front of the next tile's first
          // MFMA, which - with the uncounted DMA pieces queued - stalls on the fresh prefetch), and it can
          // no longer rematerialise the loads at the loop head.
#pragma unroll
          for (int cc = 0; cc < NC; ++cc) asm volatile("" : "+v"(an[cc]));
          __syncthreads();                     // everyone's pieces have landed; stage st is free again
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[t][s], ac[c][s], acc[t], 0, 0, 0);
          if (s == 3) {
            w4[t] = (c < NC - 1) ? frag(st, c + 1, t) : frag(st ^ 1, 0, t);
            // pin the refill right behind the MFMA that last used the register: left alone, hipcc
            // sinks all NT refills below the pass and the next chunk opens with an exposed LDS round trip
            // (~10 % of the loop in the PMC profile)
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (c == 0) {
#pragma unroll
          for (int p = s * ((NT + 3) / 4); p < (s + 1) * ((NT + 3) / 4) && p < NT; ++p)
            lds_dma16(wk, w_off[p], lds0 + unsigned((st ^ 1) * WTILE + (4 * p + wave) * 8 * GEMM_BK) * 4u);
          an[s] = a_frag(k1, s);
        }
      }
    }
  };
  int kt = 0;
  for (; kt + 1 < nk; kt += 2) {
    step(a_cur, a_nxt, kt, 0);
    step(a_nxt, a_cur, kt + 1, 1);
  }
  if (kt < nk) step(a_cur, a_nxt, kt, 0);

  LaneCtx cx;
  cx.m = m0 + wave * 32 + j;
  cx.valid = cx.m < M;
  cx.n0 = n0;
  cx.kh = kh;
  cx.lane = lane;
  cx.m_base = m0 + wave * 32;
  cx.M = M;
  // every wave passed the tile's last barrier with all of its real fragment reads done (the refills of
  // the last pass read dead prefetch data), so the ring can be reused as wave-private staging patches
  cx.patch = smem + wave * 32 * EPI_ROW4;
  DDP_STAMP(2)
  epi.template run<NT>(acc, cx);
  DDP_STAMP(3)
#undef DDP_STAMP
}

// number of blocks for the XCD-grouped 1-D grid used by k_gemm_tok
inline int gemm_grid(int M, int n_tiles_n) {
  int mt = (M + GEMM_BM - 1) / GEMM_BM;
  int mt8 = (mt + 7) / 8 * 8;
  return mt8 * n_tiles_n;
}

// ------------------------------------------------------------------------------------------------
// Row-major epilogue plumbing: tile <-> row-major transfers through a wave-private LDS patch.
// A pass covers up to 4 channel tiles (128 channels): ds_write_b128 of the D layout (row stride 32*W+4
// floats: the 8 lanes of a write group hit 8 distinct bank quads) and row-major ds_read_b128 are both
// conflict free; global accesses are 512-B contiguous row segments.  The patch is private to the wave,
// so only in-order LDS execution within the wave is relied on, no barrier.  Per-element work runs in
// ROLLED loops (a lane's channel is loop invariant there).
// ------------------------------------------------------------------------------------------------
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
//                                         independent requests are in flight)
//   fin(row, col, slot, pre)   -> void    LDS read, math, global store
// row in [0,32), col = first of the slot's 4 channels within the pass; LPR lanes cover one row.
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

// run `body(T0, Wt)` for every pass of an NT-tile row block
template <int NT, class Body>
__device__ __forceinline__ void for_each_pass(Body body) {
  if constexpr (NT >= 4) body(std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{});
  if constexpr (NT == 8) body(std::integral_constant<int, 4>{}, std::integral_constant<int, 4>{});
  if constexpr (NT == 5) body(std::integral_constant<int, 4>{}, std::integral_constant<int, 1>{});
  if constexpr (NT < 4) body(std::integral_constant<int, 0>{}, std::integral_constant<int, NT>{});
}

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

// ------------------------------------------------------------------------------------------------
// Exact-GELU at VALU cost.  GELU(x) = 0.5 x (1 + erf(x/sqrt2)).  libm's erff is ~45 instructions with
// branches; under the no-free-VALU rule above the FFN's 1024 GELUs per token were 4 % of a layer.
// Abramowitz-Stegun 7.1.26:  erf(z) = 1 - (a1 t + .. + a5 t^5) exp(-z^2), t = 1/(1 + p z), z >= 0,
// |error| <= 1.5e-7 absolute (fp32 evaluation ~3e-7), i.e. |GELU error| <= 1.5e-7 |x|: below the fp32
// summation-order noise of the contraction feeding it.  13 VALU incl. v_rcp_f32 and v_exp_f32.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  p *= t;
  const float e = __builtin_amdgcn_exp2f(-1.44269504088896340736f * z * z);
  const float erf_abs = fmaf(-p, e, 1.0f);              // erf(|x|/sqrt2)
  const float h = 0.5f * x;
  return fmaf(copysignf(erf_abs, x), h, h);              // 0.5 x (1 + erf)
}

// ------------------------------------------------------------------------------------------------
// Epilogues.  (bias is already inside the accumulators.)
// ------------------------------------------------------------------------------------------------

// Row-major output: out[m][n] = acc (+ add[row map(m)][n]) ; optional GELU.
// Row map of the addend: the tokens of all r noise replicas of one image share the image's x
// projection: add_row = (m / (r*N)) * N + m % N  (rn = r*N; rn == 0 -> identity).
struct EpiRow {
  static constexpr bool kNeedsPatch = true;
  const float* add;   // (rows, ld_add) or nullptr
  int ld_add;
  int rn, n_tok;
  float* out;
  int ldo;
  int n_valid;        // channels >= n_valid are not stored
  int gelu;           // 0 none, 1 GELU, 2 ReLU

  template <int NT>
  __device__ __forceinline__ void run(f32x16 (&acc)[NT], const LaneCtx& cx) const {
    for_each_pass<NT>([&](auto t0c, auto wtc) {
      constexpr int T0 = decltype(t0c)::value, Wt = decltype(wtc)::value;
      patch_put<NT, T0, Wt>(acc, cx.patch, cx.lane);
      const int cbase = cx.n0 + T0 * 32;
      patch_rows<Wt, 4>(
          cx.patch, cx.lane,
          [&](int row, int col) -> f32x4 {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            const int ch = cbase + col;
            if (add && ch < n_valid) {
              int m = cx.m_base + row;
              m = m < cx.M ? m : cx.M - 1;
              const size_t ar = rn ? size_t(m / rn) * n_tok + m % n_tok : size_t(m);
              v = load4_guard(add + ar * ld_add + ch, ch, n_valid);
            }
            return v;
          },
          [&](int row, int col, float* slot, const f32x4& pre) {
            const int m = cx.m_base + row, ch = cbase + col;
            if (m >= cx.M || ch >= n_valid) return;
            f32x4 v = *reinterpret_cast<const f32x4*>(slot) + pre;
            if (gelu == 1) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = gelu_fast(v[e]);
            } else if (gelu == 2) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            store4_guard(out + size_t(m) * ldo + ch, v, ch, n_valid);
          });
    });
  }
};

// Fragment-major output with c_out channels: out <- acc (+ row-major addend) ; optional GELU.
// The store is the accumulator dump itself: per (tile, quad) one 1-KiB coalesced wave store at an
// immediate offset; rows beyond M land in the buffer's padding.
struct EpiBlk {
  static constexpr bool kNeedsPatch = true;   // only used when `add` is set
  const float* add;   // row-major (rows, ld_add) or nullptr
  int ld_add;
  int rn, n_tok;
  float* out;         // fragment-major, c_out channels
  int c_out;
  int gelu;

  template <int NT>
  __device__ __forceinline__ void run(f32x16 (&acc)[NT], const LaneCtx& cx) const {
    if (add) {
      for_each_pass<NT>([&](auto t0c, auto wtc) {
        constexpr int T0 = decltype(t0c)::value, Wt = decltype(wtc)::value;
        const int cbase = cx.n0 + T0 * 32;
        patch_rows<Wt, 8>(
            cx.patch, cx.lane,
            [&](int row, int col) -> f32x4 {
              int m = cx.m_base + row;
              m = m < cx.M ? m : cx.M - 1;
              const size_t ar = rn ? size_t(m / rn) * n_tok + m % n_tok : size_t(m);
              return *reinterpret_cast<const f32x4*>(add + ar * ld_add + cbase + col);
            },
            [&](int, int, float* slot, const f32x4& r) { *reinterpret_cast<f32x4*>(slot) = r; });
        patch_add<NT, T0, Wt>(acc, cx.patch, cx.lane);
      });
    }
    float* dst = out + size_t(cx.m_base >> 5) * 32 * c_out + size_t(cx.n0 >> 5) * 1024 + cx.lane * 4;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v = {acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
        if (gelu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_fast(v[e]);
        }
        *reinterpret_cast<f32x4*>(dst + t * 1024 + g * 256) = v;
      }
  }
};

// y = acc + res;  out = (y - mean) * rstd * ga + be, with ga/be the LayerNorm affine pre-multiplied by
// the layer's FiLM (ga = gamma*(scale+1), be = beta*(scale+1)+shift; utils/transformer.py:390-392,413-417).
// res and out are fragment-major 256-channel buffers: the residual read and the store are accumulator-
// shaped 1-KiB wave accesses.  Needs NT == 8, n0 == 0.
struct EpiResLNBlk {
  static constexpr bool kNeedsPatch = false;
  const float* res;   // fragment-major (rows, 256)
  const float* ga;    // (256)
  const float* be;    // (256)
  float* out;         // fragment-major (rows, 256)

  template <int NT>
  __device__ __forceinline__ void run(f32x16 (&acc)[NT], const LaneCtx& cx) const {
    static_assert(NT == 8, "LayerNorm epilogue needs the full 256-channel row");
    const size_t goff = size_t(cx.m_base >> 5) * 32 * 256 + cx.lane * 4;
    const float* rsrc = res + goff;
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 r = *reinterpret_cast<const f32x4*>(rsrc + t * 1024 + g * 256);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = acc[t][4 * g + e] + r[e];
          acc[t][4 * g + e] = v;
          s += v;
        }
      }
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
    float* dst = out + goff;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ch = t * 32 + 8 * g + 4 * cx.kh;
        const f32x4 a = *reinterpret_cast<const f32x4*>(ga + ch);
        const f32x4 b = *reinterpret_cast<const f32x4*>(be + ch);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[t][4 * g + e] * (rstd * a[e]) + b[e];
        *reinterpret_cast<f32x4*>(dst + t * 1024 + g * 256) = v;
      }
  }
};

// Sampling projection epilogue (multi_scale_deform_attn.py:319-334 for one level), row-major (M,96) out.
// Input columns: 0..63 offsets [head][point][x,y], 64..95 attention logits [head][point].
//   raw = acc + PY[i][col] + PX[j][col]        (positional term folded through the projection:
//                                               W(q+pos) = Wq + W_y pos_y(i) + W_x pos_x(j); bias in PY)
//   offsets -> pixel-unit sample coordinates  x = j + o_x, y = i + o_y
//   logits  -> softmax over the 4 points of a head (4 consecutive columns = one float4 slot)
struct EpiSamp {
  static constexpr bool kNeedsPatch = true;
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
