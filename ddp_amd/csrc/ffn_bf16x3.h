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
constexpr size_t FFN_LDS_B = 2 * FFN_STAGE_B;

struct FfnArgs {
  const unsigned short* X;     // SB input (256 ch)
  const unsigned short* W1p;   // split fc1 weights [3][1024][256]
  const unsigned short* W2p;   // split fc2 weights [3][256][1024]
  const float* b1;             // (1024)
  const float* b2;             // (256)
  int M;
};

template <class Epi, int TAG>
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
  auto dma_w1 = [&](int hc, int s1, int stage) {
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int p = wave * 12 + i;
      const int comp = p >> 4, rb = p & 15;
      const char* src = reinterpret_cast<const char*>(fa.W1p) + (size_t(comp) * W1_COMP + size_t(hc * 64 + rb * 4) * 256) * 2 + s1 * 256;
      lds_dma16(reinterpret_cast<const float*>(src), w1_lane[i & 3], lds0 + unsigned(stage * FFN_STAGE_B + comp * 16384 + rb * 1024));
    }
  };
  auto dma_w2 = [&](int hc, int s2, int stage) {
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int p = wave * 12 + i;
      const int comp = p >> 4, rb = p & 15;
      const char* src = reinterpret_cast<const char*>(fa.W2p) + (size_t(comp) * W2_COMP + size_t(rb * 16) * 1024) * 2 + (hc * 2 + s2) * 64;
      lds_dma16(reinterpret_cast<const float*>(src), w2_lane, lds0 + unsigned(stage * FFN_STAGE_B + comp * 16384 + rb * 1024));
    }
  };
  // fragment reads
  const char* lbase = reinterpret_cast<const char*>(smem);
  auto frag1 = [&](int stage, int comp, int t, int b) -> u32x4 {      // W1 stage: row t*32+j, K16 step b (0..7)
    const int row = t * 32 + j;
    return *reinterpret_cast<const u32x4*>(lbase + stage * FFN_STAGE_B + comp * 16384 + row * 256 + (((2 * b + h) ^ (j & 15)) << 4));
  };
  auto frag2 = [&](int stage, int comp, int t, int ks) -> u32x4 {     // W2 stage: row t*32+j, K16 step ks (0..1)
    const int row = t * 32 + j;
    return *reinterpret_cast<const u32x4*>(lbase + stage * FFN_STAGE_B + comp * 16384 + row * 64 + (((2 * ks + h) ^ ((j >> 2) & 3)) << 4));
  };

  // ---- prologue: first weight stage + this wave's input fragments (kept in registers for the whole kernel)
  dma_w1(0, 0, 0);
  u32x4 xa[16][3];
  {
    const char* xs = reinterpret_cast<const char*>(fa.X) + (size_t(m0 >> 5) + wave) * 256 * 192 + lane * 16;
#pragma unroll
    for (int b = 0; b < 16; ++b)
#pragma unroll
      for (int c = 0; c < 3; ++c) xa[b][c] = *reinterpret_cast<const u32x4*>(xs + (b * 3 + c) * 1024);
  }
  f32x16 acc2[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(fa.b2 + t * 32 + 8 * g + 4 * h);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc2[t][4 * g + e] = b[e];
    }
  wait_vm0();
#pragma unroll
  for (int b = 0; b < 16; ++b)
#pragma unroll
    for (int c = 0; c < 3; ++c) asm volatile("" : "+v"(xa[b][c]));
  __syncthreads();

  // stage sequence per chunk: W1(s1=0) W1(s1=1) W2(s2=0) W2(s2=1); ring slot alternates 0,1,0,1
  for (int hc = 0; hc < 16; ++hc) {
    f32x16 acc1[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(fa.b1 + hc * 64 + t * 32 + 8 * g + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc1[t][4 * g + e] = b[e];
      }
    // ---- phase 1: acc1 += W1[chunk] . x over K = 256 (two 128-k stages)
#pragma unroll
    for (int s1 = 0; s1 < 2; ++s1) {
      if (s1 == 0) dma_w1(hc, 1, 1); else dma_w2(hc, 0, 0);          // prefetch the next stage into the other slot
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          const u32x4 w0 = frag1(s1, 0, t, b), w1 = frag1(s1, 1, t, b), w2 = frag1(s1, 2, t, b);
          const int kb = s1 * 8 + b;
          acc1[t] = mma(w2, xa[kb][0], acc1[t]);
          acc1[t] = mma(w0, xa[kb][2], acc1[t]);
          acc1[t] = mma(w1, xa[kb][1], acc1[t]);
          acc1[t] = mma(w1, xa[kb][0], acc1[t]);
          acc1[t] = mma(w0, xa[kb][1], acc1[t]);
          acc1[t] = mma(w0, xa[kb][0], acc1[t]);
        }
      wait_vm0();
      __syncthreads();
    }
    // ---- GELU + exact split: the result is the B operand of phase 2 (k-block = (tile, quad pair))
    u32x4 hp[4][3];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = gelu_fast(acc1[t][8 * gp + e]);
        split8(x, hp[t * 2 + gp][0], hp[t * 2 + gp][1], hp[t * 2 + gp][2]);
      }
    // ---- phase 2: acc2 += W2[:, chunk] . h (two 32-hidden stages, each = one tile of the chunk)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      if (s2 == 0) dma_w2(hc, 1, 1);
      else if (hc + 1 < 16) dma_w1(hc + 1, 0, 0);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const u32x4 w0 = frag2(s2, 0, t, ks), w1 = frag2(s2, 1, t, ks), w2 = frag2(s2, 2, t, ks);
          const int kb = s2 * 2 + ks;
          acc2[t] = mma(w2, hp[kb][0], acc2[t]);
          acc2[t] = mma(w0, hp[kb][2], acc2[t]);
          acc2[t] = mma(w1, hp[kb][1], acc2[t]);
          acc2[t] = mma(w1, hp[kb][0], acc2[t]);
          acc2[t] = mma(w0, hp[kb][1], acc2[t]);
          acc2[t] = mma(w0, hp[kb][0], acc2[t]);
        }
      wait_vm0();
      __syncthreads();
    }
  }

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
