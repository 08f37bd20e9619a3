// ffn_bf16x3.h - the whole FFN block of a decoder layer in ONE kernel (bf16x3-split arithmetic):
//
//     out = FiLM( LayerNorm( x + W2 . GELU(W1 . x + b1) + b2 ) )            (mmcv FFN, transformer.py:269-280,
//                                                                             utils/transformer.py:390-392,413-417)
//
// Why fuse.  With the 2.67x faster bf16x3 contractions the layer became HBM bound: the 1024-wide hidden
// activation alone cost 1.57 GB written + 1.57 GB read per layer (6 B per element in split form, M = 262144
// tokens), 35 % of all traffic.  Here it never leaves the register file:
//   * a wave owns 32 tokens and keeps their 256-channel input x as B-operand fragments in registers for the
//     whole kernel (16 K16-blocks x 3 bf16 pieces x 4 VGPRs = 192 VGPRs; one wave per SIMD = 512 VGPRs);
//   * the hidden layer is produced 64 channels at a time: acc1 (2 tiles) <- b1 + W1[chunk] . x; then GELU;
//   * because an accumulator lane (token j, half h) holds, for tile t, the channels 8g + 4h + e, and the
//     k-slots of an MFMA B operand may be ANY fixed permutation of K shared with the (pre-permuted) weights,
//     the split GELU output IS the B fragment of the second contraction (k-block = (tile, quad pair)): no LDS
//     round trip, no shuffle;
//   * acc2 (8 tiles, 128 VGPRs) accumulates W2[:, chunk] . h over the 16 chunks, then the usual
//     residual + LayerNorm + FiLM epilogue writes fp32 fragment-major + split fragment-major outputs.
// Weights stream through a 2 x 48 KB LDS ring by LDS-DMA, one barrier per stage of 96 MFMAs per wave:
// per chunk two W1 stages [64 hidden rows x 128 k] and two W2 stages [256 out rows x 32 hidden]; all CUs
// walk the same 3 MB of split weights in the same order, so they are served from each XCD's L2.
#pragma once
#include "gemm_bf16x3.h"

namespace ddp {
namespace b3 {

constexpr int FFN_BM = 128;
constexpr int FFN_THREADS = 256;
constexpr int FFN_STAGE_B = 48 * 1024;
constexpr int FFN_RING = 3;                                   // stages in flight: compute q, landed q+1, landing q+2
constexpr int FFN_B1_OFF = FFN_RING * FFN_STAGE_B;            // fc1 bias (4 KB) behind the ring
constexpr size_t FFN_LDS_B = size_t(FFN_B1_OFF) + 4096;

struct FfnArgs {
  const unsigned short* X;     // SB input (256 ch)
  const unsigned short* W1p;   // split fc1 weights [3][1024][256]
  const unsigned short* W2p;   // split fc2 weights [3][256][1024]
  const float* b1;             // (1024)
  const float* b2;             // (256)
  int M;
  // OUTPROJ variant: X is unused; x = LayerNorm(Q + Wo . S + bo) is computed in the kernel's prologue
  const unsigned short* S;     // SB attention output (256 ch): A operand of output_proj
  const unsigned short* Q;     // SB layer input (residual of output_proj)
  const unsigned short* Wop;   // split output_proj weights [3][256][256]
  const float* bo;             // (256)
  const float* ga0;            // norms.0 weight (256)
  const float* be0;            // norms.0 bias (256)
};

__device__ __forceinline__ void wait_vm12() { __builtin_amdgcn_s_waitcnt(0x0F7C); }   // vmcnt(12): one DMA stage may stay in flight

// element u (0..7) of a packed bf16x8 fragment as fp32
__device__ __forceinline__ float bf_elem(const u32x4& v, int u) {
  const unsigned w = v[u >> 1];
  return __uint_as_float((u & 1) ? (w & 0xFFFF0000u) : (w << 16));
}

template <class Epi, int TAG, bool OUTPROJ>
__global__ void __launch_bounds__(FFN_THREADS, 1)
k_ffn(FfnArgs fa, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31;
  const int h = lane >> 5;
  const int M = fa.M;
  const int m0 = blockIdx.x * FFN_BM;
  if (m0 >= M) return;
  const unsigned lds0 = (unsigned)(size_t)(lds_float_t*)smem;
  constexpr size_t W1_COMP = size_t(1024) * 256;   // elements
  constexpr size_t W2_COMP = size_t(256) * 1024;

  // ---- LDS-DMA of one 48 KB stage = 48 pieces, 12 per wave: piece p = wave*12 + i -> (comp = p/16, rb = p%16)
  // W1 stage s1 (0/1) of chunk hc: rows [hc*64, +64) x k [s1*128, +128): piece = 4 rows x 256 B; lane -> (row, slot)
  unsigned w1_lane[4];   // per (rb & 3): byte offset of this lane inside a piece's source
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) {
    const int rowp = lane >> 4;                       // row within the piece
    const int row15 = (4 * r4 + rowp) & 15;           // (row & 15) for rb = r4 (mod 4)
    const int chunk = (lane & 15) ^ row15;
    w1_lane[r4] = unsigned(rowp * 256 * 2 + chunk * 16);
  }
  // W2 stage s2 (0/1) of chunk hc: rows [0,256) x hidden [hc*64 + s2*32, +32): piece = 16 rows x 64 B
  const unsigned w2_lane = unsigned((lane >> 2) * 1024 * 2 + (((lane & 3) ^ ((lane >> 4) & 3)) * 16));
  auto dma_w1 = [&](int hc, int s1, int slot, int i0, int i1) {
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      if (i < i0 || i >= i1) continue;
      const int p = wave * 12 + i;
      const int comp = p >> 4, rb = p & 15;
      const char* src = reinterpret_cast<const char*>(fa.W1p) + (size_t(comp) * W1_COMP + size_t(hc * 64 + rb * 4) * 256) * 2 + s1 * 256;
      lds_dma16(reinterpret_cast<const float*>(src), w1_lane[i & 3], lds0 + unsigned(slot * FFN_STAGE_B + comp * 16384 + rb * 1024));
    }
  };
  auto dma_w2 = [&](int hc, int s2, int slot, int i0, int i1) {
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      if (i < i0 || i >= i1) continue;
      const int p = wave * 12 + i;
      const int comp = p >> 4, rb = p & 15;
      const char* src = reinterpret_cast<const char*>(fa.W2p) + (size_t(comp) * W2_COMP + size_t(rb * 16) * 1024) * 2 + (hc * 2 + s2) * 64;
      lds_dma16(reinterpret_cast<const float*>(src), w2_lane, lds0 + unsigned(slot * FFN_STAGE_B + comp * 16384 + rb * 1024));
    }
  };
  // fragment reads (per-lane base + slot base + compile-time offsets)
  const char* lbase = reinterpret_cast<const char*>(smem);
  const int f1_lane = j * 256;                        // W1 stage: row t*32+j, 16 slots of 16 B, swizzle (j & 15)
  const int f2_lane = j * 64;                         // W2 stage: row t*32+j, 4 slots, swizzle ((j>>2)&3)
  auto frag1 = [&](int slot, int comp, int t, int b) -> u32x4 {      // K16 step b (0..7)
    return *reinterpret_cast<const u32x4*>(lbase + slot * FFN_STAGE_B + comp * 16384 + t * 8192 + f1_lane + (((2 * b + h) ^ (j & 15)) << 4));
  };
  auto frag2 = [&](int slot, int comp, int t, int ks) -> u32x4 {     // K16 step ks (0..1)
    return *reinterpret_cast<const u32x4*>(lbase + slot * FFN_STAGE_B + comp * 16384 + t * 2048 + f2_lane + (((2 * ks + h) ^ ((j >> 2) & 3)) << 4));
  };
  auto nxt = [](int s) { return s == FFN_RING - 1 ? 0 : s + 1; };

  // output_proj stage st (0..7): rows [0,256) x k [st*32, +32) of Wo (row = 256 k): same piece shape as a W2 stage
  const unsigned wo_lane = unsigned((lane >> 2) * 256 * 2 + (((lane & 3) ^ ((lane >> 4) & 3)) * 16));
  constexpr size_t WO_COMP = size_t(256) * 256;
  auto dma_wo = [&](int st, int slot, int i0, int i1) {
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      if (i < i0 || i >= i1) continue;
      const int p = wave * 12 + i;
      const int comp = p >> 4, rb = p & 15;
      const char* src = reinterpret_cast<const char*>(fa.Wop) + (size_t(comp) * WO_COMP + size_t(rb * 16) * 256) * 2 + st * 64;
      lds_dma16(reinterpret_cast<const float*>(src), wo_lane, lds0 + unsigned(slot * FFN_STAGE_B + comp * 16384 + rb * 1024));
    }
  };

#define DDP_FFN_BLOCK(A0, A1, X0, X1, X2, RELOAD2, RELOAD1, RELOAD0, FILL1, FILL0)                                   \
  A0 = mma(w[0][2], X0, A0);                                                                                        \
  A1 = mma(w[1][2], X0, A1);                                                                                        \
  RELOAD2;                                                                                                          \
  __builtin_amdgcn_sched_barrier(0);                                                                                \
  A0 = mma(w[0][1], X1, A0);                                                                                        \
  A1 = mma(w[1][1], X1, A1);                                                                                        \
  A0 = mma(w[0][1], X0, A0);                                                                                        \
  A1 = mma(w[1][1], X0, A1);                                                                                        \
  FILL1;                                                                                                            \
  RELOAD1;                                                                                                          \
  __builtin_amdgcn_sched_barrier(0);                                                                                \
  A0 = mma(w[0][0], X2, A0);                                                                                        \
  A1 = mma(w[1][0], X2, A1);                                                                                        \
  A0 = mma(w[0][0], X1, A0);                                                                                        \
  A1 = mma(w[1][0], X1, A1);                                                                                        \
  A0 = mma(w[0][0], X0, A0);                                                                                        \
  A1 = mma(w[1][0], X0, A1);                                                                                        \
  FILL0;                                                                                                            \
  RELOAD0;                                                                                                          \
  __builtin_amdgcn_sched_barrier(0);

  const size_t grp = size_t(m0 >> 5) + wave;          // this wave's 32-token group
  u32x4 xa[16][3];
  f32x16 acc2[8];
  int slot = 0;
  *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(smem) + FFN_B1_OFF + tid * 16) = *reinterpret_cast<const f32x4*>(fa.b1 + tid * 4);
  if constexpr (OUTPROJ) {
    // ---- P0: acc2 = bo + Wo . s  (8 stages of the ring; the last two prefetch fc1's first two stages)
    dma_wo(0, 0, 0, 12);
    dma_wo(1, 1, 0, 12);
    const char* ss = reinterpret_cast<const char*>(fa.S) + grp * 256 * 192 + lane * 16;
    u32x4 sc[2][3], sn[2][3];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int c = 0; c < 3; ++c) sc[ks][c] = *reinterpret_cast<const u32x4*>(ss + (ks * 3 + c) * 1024);
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(fa.bo + t * 32 + 8 * g + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc2[t][4 * g + e] = b[e];
      }
    wait_vm0();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int c = 0; c < 3; ++c) asm volatile("" : "+v"(sc[ks][c]));
    __syncthreads();
    auto p0_stage = [&](int st, auto tailc) {
      constexpr bool TAIL = decltype(tailc)::value;      // stages 6,7: the ring's look-ahead is fc1 chunk 0
      const int dslot = nxt(nxt(slot));
      const int stn = st + 1 < 8 ? st + 1 : 7;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int c = 0; c < 3; ++c) sn[ks][c] = *reinterpret_cast<const u32x4*>(ss + ((2 * stn + ks) * 3 + c) * 1024);
      u32x4 w[2][3];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c = 0; c < 3; ++c) w[t][c] = frag2(slot, c, t, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
          const int blk = ks * 4 + tp;
          const bool nb = blk + 1 < 8;
          const int tp2 = tp + 1 < 4 ? tp + 1 : 0, ks2 = tp + 1 < 4 ? ks : ks + 1;
          DDP_FFN_BLOCK(acc2[2 * tp], acc2[2 * tp + 1], sc[ks][0], sc[ks][1], sc[ks][2],
                        if (nb) { w[0][2] = frag2(slot, 2, 2 * tp2, ks2); w[1][2] = frag2(slot, 2, 2 * tp2 + 1, ks2); },
                        if (nb) { w[0][1] = frag2(slot, 1, 2 * tp2, ks2); w[1][1] = frag2(slot, 1, 2 * tp2 + 1, ks2); },
                        if (nb) { w[0][0] = frag2(slot, 0, 2 * tp2, ks2); w[1][0] = frag2(slot, 0, 2 * tp2 + 1, ks2); },
                        {
                          if (TAIL) dma_w1(0, st - 6, dslot, blk, blk + 1); else dma_wo(st + 2, dslot, blk, blk + 1);
                        },
                        {
                          if (blk < 4) {
                            if (TAIL) dma_w1(0, st - 6, dslot, 8 + blk, 9 + blk); else dma_wo(st + 2, dslot, 8 + blk, 9 + blk);
                          }
                        })
        }
      wait_vm12();
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          asm volatile("" : "+v"(sn[ks][c]));
          sc[ks][c] = sn[ks][c];
        }
      __syncthreads();
      slot = nxt(slot);
    };
    for (int st = 0; st < 6; ++st) p0_stage(st, std::false_type{});
    // residual fragments: fetched under the last two stages
    u32x4 qa[16][3];
    {
      const char* qs = reinterpret_cast<const char*>(fa.Q) + grp * 256 * 192 + lane * 16;
#pragma unroll
      for (int b = 0; b < 16; ++b)
#pragma unroll
        for (int c = 0; c < 3; ++c) qa[b][c] = *reinterpret_cast<const u32x4*>(qs + (b * 3 + c) * 1024);
    }
    p0_stage(6, std::true_type{});
    p0_stage(7, std::true_type{});
    // ---- P1: y = acc2 + q; x = LayerNorm(y) * ga0 + be0 -> input fragments; acc2 <- b2 + x (fc2 bias + FFN residual)
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int b = 2 * t + (r >> 3), u = r & 7;
        const float v = acc2[t][r] + ((bf_elem(qa[b][0], u) + bf_elem(qa[b][1], u)) + bf_elem(qa[b][2], u));
        acc2[t][r] = v;
        sum += v;
      }
    const float mean = half_sum(sum) * (1.0f / 256.0f);
    float var = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d = acc2[t][r] - mean;
        acc2[t][r] = d;
        var += d * d;
      }
    const float rstd = 1.0f / sqrtf(half_sum(var) * (1.0f / 256.0f) + 1e-5f);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ch = t * 32 + 8 * g + 4 * h;
        const f32x4 ga = *reinterpret_cast<const f32x4*>(fa.ga0 + ch);
        const f32x4 be = *reinterpret_cast<const f32x4*>(fa.be0 + ch);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc2[t][4 * g + e] = acc2[t][4 * g + e] * (rstd * ga[e]) + be[e];
      }
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        float xv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[e] = acc2[t][8 * gp + e];
        split8(xv, xa[2 * t + gp][0], xa[2 * t + gp][1], xa[2 * t + gp][2]);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(fa.b2 + t * 32 + 8 * g + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc2[t][4 * g + e] += b[e];
      }
    }
  } else {
    // ---- prologue: two weight stages, this wave's input fragments; acc2 <- b2 + x (fc2 bias + FFN residual)
    dma_w1(0, 0, 0, 0, 12);
    dma_w1(0, 1, 1, 0, 12);
    {
      const char* xs = reinterpret_cast<const char*>(fa.X) + grp * 256 * 192 + lane * 16;
#pragma unroll
      for (int b = 0; b < 16; ++b)
#pragma unroll
        for (int c = 0; c < 3; ++c) xa[b][c] = *reinterpret_cast<const u32x4*>(xs + (b * 3 + c) * 1024);
    }
    wait_vm0();
#pragma unroll
    for (int b = 0; b < 16; ++b)
#pragma unroll
      for (int c = 0; c < 3; ++c) asm volatile("" : "+v"(xa[b][c]));
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(fa.b2 + t * 32 + 8 * g + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int bb = 2 * t + (g >> 1), u = 4 * (g & 1) + e;
          acc2[t][4 * g + e] = b[e] + ((bf_elem(xa[bb][0], u) + bf_elem(xa[bb][1], u)) + bf_elem(xa[bb][2], u));
        }
      }
    __syncthreads();
  }

  // Stage sequence per chunk: W1(k 0..127) W1(k 128..255) W2(hidden 0..31) W2(hidden 32..63); stage q lives in
  // ring slot q % 3 and its DMA is issued at the start of stage q-2.
  const float* b1s = reinterpret_cast<const float*>(reinterpret_cast<const char*>(smem) + FFN_B1_OFF);
  for (int hc = 0; hc < 16; ++hc) {
    f32x16 acc1[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(b1s + hc * 64 + t * 32 + 8 * g + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc1[t][4 * g + e] = b[e];
      }
    // Scheduling rules for one wave per SIMD (MI355X_MICROARCH.md, latency table):
    //  * a dependent MFMA right behind its producer is free only when NOTHING is issued between them, and the
    //    loads / GELU / DMA must sit somewhere -> two accumulator chains (a tile pair) are always interleaved;
    //  * a fragment is re-read into its own registers as soon as its last MFMA of the block has issued (w2 after
    //    2 MFMAs, w1 after 6, w0 after 12 - terms ordered for that), >= 6 MFMAs (192 cycles) before its next use;
    //  * the next-next stage's 12 DMA pieces are spread over the 8 blocks of a stage.

    // ---- phase 1: acc1 += W1[chunk] . x over K = 256 (two 128-k stages)
#pragma unroll
    for (int s1 = 0; s1 < 2; ++s1) {
      const int dslot = nxt(nxt(slot));
      u32x4 w[2][3];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c = 0; c < 3; ++c) w[t][c] = frag1(slot, c, t, 0);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const int kb = s1 * 8 + b;
        const bool nb = b + 1 < 8;
        DDP_FFN_BLOCK(acc1[0], acc1[1], xa[kb][0], xa[kb][1], xa[kb][2],
                      if (nb) { w[0][2] = frag1(slot, 2, 0, b + 1); w[1][2] = frag1(slot, 2, 1, b + 1); },
                      if (nb) { w[0][1] = frag1(slot, 1, 0, b + 1); w[1][1] = frag1(slot, 1, 1, b + 1); },
                      if (nb) { w[0][0] = frag1(slot, 0, 0, b + 1); w[1][0] = frag1(slot, 0, 1, b + 1); },
                      dma_w2(hc, s1, dslot, b, b + 1),
                      if (b < 4) dma_w2(hc, s1, dslot, 8 + b, 9 + b))
      }
      wait_vm12();
      __syncthreads();
      slot = nxt(slot);
    }
    // ---- GELU + exact split: the result IS the B operand of phase 2 (k-block kb = (tile kb/2, quad pair kb%2));
    //      block 0 here, block kb+1 under block kb's MFMAs
    u32x4 hcur[3];
    float xg[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) xg[e] = gelu_fast(acc1[0][e]);
    split8(xg, hcur[0], hcur[1], hcur[2]);
    // ---- phase 2: acc2 += W2[:, chunk] . h (two 32-hidden stages, each = one tile of the chunk)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int hn = hc + 1 < 16 ? hc + 1 : 15;      // last chunk: a harmless re-fetch keeps the loop branch-free
      const int dslot = nxt(nxt(slot));
      u32x4 w[2][3];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c = 0; c < 3; ++c) w[t][c] = frag2(slot, c, t, 0);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int kb = s2 * 2 + ks;
        const int kn = kb + 1;
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
          const int blk = ks * 4 + tp;
          const bool nb = blk + 1 < 8;
          const int tp2 = tp + 1 < 4 ? tp + 1 : 0, ks2 = tp + 1 < 4 ? ks : ks + 1;
          DDP_FFN_BLOCK(acc2[2 * tp], acc2[2 * tp + 1], hcur[0], hcur[1], hcur[2],
                        if (nb) { w[0][2] = frag2(slot, 2, 2 * tp2, ks2); w[1][2] = frag2(slot, 2, 2 * tp2 + 1, ks2); },
                        if (nb) { w[0][1] = frag2(slot, 1, 2 * tp2, ks2); w[1][1] = frag2(slot, 1, 2 * tp2 + 1, ks2); },
                        if (nb) { w[0][0] = frag2(slot, 0, 2 * tp2, ks2); w[1][0] = frag2(slot, 0, 2 * tp2 + 1, ks2); },
                        {
                          dma_w1(hn, s2, dslot, blk, blk + 1);
                          if (kb < 3) xg[2 * tp] = gelu_fast(acc1[kn >> 1][8 * (kn & 1) + 2 * tp]);
                        },
                        {
                          if (blk < 4) dma_w1(hn, s2, dslot, 8 + blk, 9 + blk);
                          if (kb < 3) xg[2 * tp + 1] = gelu_fast(acc1[kn >> 1][8 * (kn & 1) + 2 * tp + 1]);
                        })
        }
        if (kb < 3) split8(xg, hcur[0], hcur[1], hcur[2]);
      }
      wait_vm12();
      __syncthreads();
      slot = nxt(slot);
    }
  }
#undef DDP_FFN_BLOCK
  wait_vm0();

  LaneCtx cx;
  cx.m = m0 + wave * 32 + j;
  cx.valid = cx.m < M;
  cx.n0 = 0;
  cx.kh = h;
  cx.lane = lane;
  cx.m_base = m0 + wave * 32;
  cx.M = M;
  cx.patch = smem;
  epi.template run<8>(acc2, cx);
}

}  // namespace b3
}  // namespace ddp
