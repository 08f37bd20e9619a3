// gemm_bf16x3.h - fp32-accurate token GEMM on the gfx950 bf16 matrix cores ("3-way split, 6 products").
//
//   C[m][n] = sum_k A[m][k] * W[n][k]
//
// Why.  The exact fp32 MFMA (gemm_f32.h) is limited to 157 TFLOP/s and runs on the SIMD's own FP32
// datapath, so while it streams, co-resident waves get no VALU at all (scripts/ubench/mfma_valu.hip).
// The bf16 matrix core is 16x faster (2.5 PFLOP/s) and separate from the VALU.  An fp32 number is exactly
// the sum of three bf16 numbers (its 24-bit significand cut 8+8+8: x = x1 + x2 + x3, truncation split),
// and a bf16 x bf16 product is exact in fp32, so with fp32 accumulation inside the MFMA
//     a*b = a1b1 + (a1b2 + a2b1) + (a1b3 + a3b1 + a2b2) + O(2^-24 |a||b|)
// Six bf16 MFMAs replace eight fp32 MFMAs of half the K each: 6 x 32 cycles per 32x32x16 block instead of
// 8 x 64, a 2.67x higher ceiling (417 TFLOP/s fp32-equivalent) at fp32-class accuracy (the dropped terms a2b3, a3b2,
// a3b3 are below one fp32 ulp of the product).  The claim is a test (ddp_linear_b3 + tests/test_b3_arithmetic.py, against
// fp64): worst error 5.7 .. 8.8 units of 2^-24 sum|a||w| at K = 256 .. 1024 where the exact-product fp32 MFMA engine
// measures 6.4 .. 9.0 on the same operands, wide-dynamic-range, cancelling and near-subnormal rows included.  This is
// NOT a reduced-precision mode: parity tests run against the same fp32 oracle with the same tolerances.
//
// Layouts
//  * weights are split + K-permuted once per ddp_prepare into Wp[comp][n][K] (bf16, comp = 0..2):
//        Wp[c][n][16b + 8h + u] = piece_c( W[n][16b + 8*(u/4) + 4h + (u%4)] )
//    i.e. inside every 16-wide K block the order is the one in which an accumulator lane pair holds it
//    (below), so that producers need no cross-lane exchange.
//  * activations travel between GEMMs as split fragment-major ("SB") buffers, C channels, rows padded to
//    256: per 32-token group [K16 block b = C/16][comp 3][lane 64][8 bf16]; lane = h*32 + (m%32),
//    element u of lane h in block b = channel 16b + 8*(u/4) + 4h + (u%4).  For the MFMA this is exactly the
//    B fragment of lane (j, h) for K16 step b; for the producer it is exactly "my own accumulator quads
//    g = 2gp and 2gp+1 of tile t" (b = 2t + gp) - a 1-KiB coalesced wave store per (b, comp).
//
// Kernel shape: block = 512 threads = 8 waves (2 per SIMD), tile = 256 tokens x NT*32 channels, one block
// per CU (LDS ring 2 x 3 x NT*32 x 64 B = 96 KB at NT = 8).  Wave w owns tokens [32w, 32w+32) and all
// channel tiles (token = MFMA column = lane&31 as in gemm_f32.h, so the row epilogues carry over).
// Weights: LDS-DMA, double buffered, XOR-swizzled 16-B slots (slot = chunk ^ ((row>>2)&3)); activations:
// direct 1-KiB fragment loads one k-tile ahead.
#pragma once
#include <hip/hip_runtime.h>

#include "gemm_f32.h"

namespace ddp {
namespace b3 {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;   // 8 packed bf16 (HIP's u32x4 is a struct: no "+v")

constexpr int BM = 256;
constexpr int THREADS = 512;
constexpr int BK = 32;                 // k-tile: two K16 MFMA steps
constexpr int ROW_B = BK * 2;          // bytes per (row, k-tile, comp) = 64

template <int NT>
constexpr size_t ring_bytes() {
  return size_t(2) * 3 * NT * 32 * ROW_B;
}
constexpr size_t patch_bytes() { return size_t(8) * 32 * EPI_ROW4 * sizeof(float); }
template <int NT, class Epi>
constexpr size_t lds_bytes() {
  return (Epi::kNeedsPatch && patch_bytes() > ring_bytes<NT>()) ? patch_bytes() : ring_bytes<NT>();
}

struct Args {
  const unsigned short* A;   // SB activations, K channels
  const unsigned short* Wp;  // split weights, comp stride = w_comp_stride elements
  size_t w_comp_stride;
  int M, N, K;
  int n_tiles_n;
  const float* acc_bias;
};

__device__ __forceinline__ f32x16 mma(u32x4 w, u32x4 a, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), c, 0, 0, 0);
}

// exact 3-way truncation split of 8 fp32 (two accumulator quads) into three packed bf16x8
__device__ __forceinline__ void split8(const float (&x)[8], u32x4& p1, u32x4& p2, u32x4& p3) {
  unsigned h[8], m[8], l[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const unsigned xb = __float_as_uint(x[i]);
    h[i] = xb & 0xFFFF0000u;
    const float r = x[i] - __uint_as_float(h[i]);
    const unsigned rb = __float_as_uint(r);
    m[i] = rb & 0xFFFF0000u;
    const float r2 = r - __uint_as_float(m[i]);
    l[i] = __float_as_uint(r2);
  }
  // pack the high halves of element pairs: (hi16(e1) << 16) | hi16(e0)
  p1[0] = __builtin_amdgcn_perm(h[1], h[0], 0x07060302); p1[1] = __builtin_amdgcn_perm(h[3], h[2], 0x07060302);
  p1[2] = __builtin_amdgcn_perm(h[5], h[4], 0x07060302); p1[3] = __builtin_amdgcn_perm(h[7], h[6], 0x07060302);
  p2[0] = __builtin_amdgcn_perm(m[1], m[0], 0x07060302); p2[1] = __builtin_amdgcn_perm(m[3], m[2], 0x07060302);
  p2[2] = __builtin_amdgcn_perm(m[5], m[4], 0x07060302); p2[3] = __builtin_amdgcn_perm(m[7], m[6], 0x07060302);
  p3[0] = __builtin_amdgcn_perm(l[1], l[0], 0x07060302); p3[1] = __builtin_amdgcn_perm(l[3], l[2], 0x07060302);
  p3[2] = __builtin_amdgcn_perm(l[5], l[4], 0x07060302); p3[3] = __builtin_amdgcn_perm(l[7], l[6], 0x07060302);
}

// store one 32x32 accumulator tile (tile index tg of a c_out-channel SB buffer) as split fragments
__device__ __forceinline__ void store_tile_sb(const f32x16& a, unsigned short* sb, int c_out, int m_base, int tg, int lane) {
  char* base = reinterpret_cast<char*>(sb) + size_t(m_base >> 5) * c_out * 192 + size_t(tg) * 2 * 3 * 1024 + lane * 16;
#pragma unroll
  for (int gp = 0; gp < 2; ++gp) {
    const float x[8] = {a[8 * gp], a[8 * gp + 1], a[8 * gp + 2], a[8 * gp + 3],
                        a[8 * gp + 4], a[8 * gp + 5], a[8 * gp + 6], a[8 * gp + 7]};
    u32x4 p1, p2, p3;
    split8(x, p1, p2, p3);
    *reinterpret_cast<u32x4*>(base + (gp * 3 + 0) * 1024) = p1;
    *reinterpret_cast<u32x4*>(base + (gp * 3 + 1) * 1024) = p2;
    *reinterpret_cast<u32x4*>(base + (gp * 3 + 2) * 1024) = p3;
  }
}

// (3x3 convolutions run as implicit GEMMs on the persistent stream kernel: layer_bf16x3.h MODE 5)
template <int NT, class Epi, int TAG>
__global__ void __launch_bounds__(THREADS, 2)
k_gemm(Args ga, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int COMP_B = NT * 32 * ROW_B;        // bytes per component per stage
  constexpr int STAGE_B = 3 * COMP_B;
  constexpr int PIECES = 3 * NT * 2;             // 1-KiB DMA pieces per stage (16 rows each)
  constexpr int PPW = (PIECES + 7) / 8;          // pieces per wave

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31;
  const int h = lane >> 5;
  const int M = ga.M, N = ga.N;

  int mt, nt;
  {
    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    const int idx = bid >> 3;
    nt = idx % ga.n_tiles_n;
    mt = (idx / ga.n_tiles_n) * 8 + xcd;
  }
  const int m0 = mt * BM;
  if (m0 >= M) return;
  const int n0 = nt * NT * 32;

  // weight DMA pieces of this wave: piece pi = wave + 8*i -> (comp, 16-row block); lane -> (row, slot)
  unsigned w_off[PPW];
  unsigned w_dst[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int pi = wave + 8 * i;
    const int comp = pi / (NT * 2), rb = pi % (NT * 2);
    const int row = rb * 16 + (lane >> 2);
    int gn = n0 + row;
    gn = gn < N ? gn : N - 1;
    const int chunk = (lane & 3) ^ ((row >> 2) & 3);
    w_off[i] = unsigned((size_t(comp) * ga.w_comp_stride + size_t(gn) * ga.K) * 2 + chunk * 16);
    w_dst[i] = unsigned(comp * COMP_B + rb * 1024);
  }
  const unsigned lds0 = (unsigned)(size_t)(lds_float_t*)smem;
  auto dma_w = [&](int kt, int stage) {
    const float* wk = reinterpret_cast<const float*>(reinterpret_cast<const char*>(ga.Wp) + size_t(kt) * ROW_B);
#pragma unroll
    for (int i = 0; i < PPW; ++i)
      if (wave + 8 * i < PIECES) lds_dma16(wk, w_off[i], lds0 + unsigned(stage * STAGE_B) + w_dst[i]);
  };

  // activation fragments: SB buffer, this wave's 32-token group
  const char* a_src = reinterpret_cast<const char*>(ga.A) + (size_t(m0 >> 5) + wave) * ga.K * 192 + lane * 16;
  struct ASrc {
    const char* p;     // address of (K16 block 0, piece 0) of this lane's slot
  };
  auto a_source = [&](int kt) -> ASrc {
    ASrc r;
    r.p = a_src + size_t(kt) * 2 * 3 * 1024;
    return r;
  };
  auto a_frag = [&](const ASrc& sp, int ks, int comp) -> u32x4 {
    return *reinterpret_cast<const u32x4*>(sp.p + (ks * 3 + comp) * 1024);
  };

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int ch = n0 + t * 32 + 8 * g + 4 * h;
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

  const int nk = ga.K / BK;
  u32x4 a_cur[2][3], a_nxt[2][3];
  dma_w(0, 0);
  {
    const ASrc s0 = a_source(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int c = 0; c < 3; ++c) a_cur[ks][c] = a_frag(s0, ks, c);
  }
  wait_vm0();
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int c = 0; c < 3; ++c) asm volatile("" : "+v"(a_cur[ks][c]));
  __syncthreads();

  // fragment read: row i = lane&31 of tile t, 16-B chunk (2 ks + h), swizzled slot
  const char* wbase = reinterpret_cast<const char*>(smem) + j * ROW_B;
  const int sw = (j >> 2) & 3;
  auto frag = [&](int st, int comp, int t, int ks) -> u32x4 {
    return *reinterpret_cast<const u32x4*>(wbase + st * STAGE_B + comp * COMP_B + t * 32 * ROW_B + (((2 * ks + h) ^ sw) << 4));
  };

  auto step = [&](u32x4 (&ac)[2][3], u32x4 (&an)[2][3], int kt, int st) {
    const int k1 = kt + 1 < nk ? kt + 1 : nk - 1;
    dma_w(k1, st ^ 1);
    const ASrc s1 = a_source(k1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int c = 0; c < 3; ++c) an[ks][c] = a_frag(s1, ks, c);
    u32x4 w[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) w[c] = frag(st, c, 0, 0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        u32x4 wn[3];
        const bool last = (ks == 1 && t == NT - 1);
        if (!last) {
          const int t2 = (t + 1 < NT) ? t + 1 : 0, ks2 = (t + 1 < NT) ? ks : ks + 1;
#pragma unroll
          for (int c = 0; c < 3; ++c) wn[c] = frag(st, c, t2, ks2);
        }
        // smallest terms first
        acc[t] = mma(w[2], ac[ks][0], acc[t]);
        acc[t] = mma(w[0], ac[ks][2], acc[t]);
        acc[t] = mma(w[1], ac[ks][1], acc[t]);
        acc[t] = mma(w[1], ac[ks][0], acc[t]);
        acc[t] = mma(w[0], ac[ks][1], acc[t]);
        acc[t] = mma(w[0], ac[ks][0], acc[t]);
        if (!last) {
#pragma unroll
          for (int c = 0; c < 3; ++c) w[c] = wn[c];
        }
      }
    }
    wait_vm0();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int c = 0; c < 3; ++c) asm volatile("" : "+v"(an[ks][c]));
    __syncthreads();
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
  cx.kh = h;
  cx.lane = lane;
  cx.m_base = m0 + wave * 32;
  cx.M = M;
  cx.patch = smem + wave * 32 * EPI_ROW4;   // ring is dead after the trailing barrier
  epi.template run<NT>(acc, cx);
}

inline int grid(int M, int n_tiles_n) {
  int mt = (M + BM - 1) / BM;
  int mt8 = (mt + 7) / 8 * 8;
  return mt8 * n_tiles_n;
}

// ---------------------------------------------------------------------------------------------------
// Epilogues (bias already inside the accumulators).  The row-major ones are gemm_f32.h's EpiRow / EpiSamp.
// ---------------------------------------------------------------------------------------------------

// SB (split fragment-major) output with c_out channels (+ optional fp32 fragment-major copy for a later
// residual read, + optional row-major addend, + optional GELU)
struct EpiSB {
  static constexpr bool kNeedsPatch = true;   // used when `add` is set
  const float* add;      // row-major (rows, ld_add) or nullptr
  int ld_add;
  int rn, n_tok;
  unsigned short* out_sb;
  float* out_f32;        // fragment-major fp32 (256 ch) or nullptr
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
    float* dstf = out_f32 ? out_f32 + size_t(cx.m_base >> 5) * 32 * c_out + size_t(cx.n0 >> 5) * 1024 + cx.lane * 4 : nullptr;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (gelu) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = gelu_fast(acc[t][r]);
      }
      if (dstf) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<f32x4*>(dstf + t * 1024 + g * 256) =
              f32x4{acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
      }
      store_tile_sb(acc[t], out_sb, c_out, cx.m_base, (cx.n0 >> 5) + t, cx.lane);
    }
  }
};

// y = acc + res; LayerNorm; affine x FiLM; out as SB (+ optional fp32 fragment-major copy).
// The residual comes as SB (res_sb: the three bf16 pieces sum to the fp32 value exactly), as fp32 fragment-major
// (res), or not at all (both null: the caller added it).
struct EpiResLNSB {
  static constexpr bool kNeedsPatch = false;
  const float* res;
  const unsigned short* res_sb;
  const float* ga;
  const float* be;
  float* out_f32;
  unsigned short* out_sb;

  template <int NT>
  __device__ __forceinline__ void run(f32x16 (&acc)[NT], const LaneCtx& cx) const {
    static_assert(NT == 8, "LayerNorm epilogue needs the full 256-channel row");
    const size_t goff = size_t(cx.m_base >> 5) * 32 * 256 + cx.lane * 4;
    float s = 0.f;
    if (res_sb) {
      const char* rs = reinterpret_cast<const char*>(res_sb) + size_t(cx.m_base >> 5) * 256 * 192 + cx.lane * 16;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          const char* p = rs + size_t(2 * t + gp) * 3 * 1024;
          const u32x4 p1 = *reinterpret_cast<const u32x4*>(p);
          const u32x4 p2 = *reinterpret_cast<const u32x4*>(p + 1024);
          const u32x4 p3 = *reinterpret_cast<const u32x4*>(p + 2048);
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const unsigned w1 = p1[u >> 1], w2 = p2[u >> 1], w3 = p3[u >> 1];
            const float r = (__uint_as_float((u & 1) ? (w1 & 0xFFFF0000u) : (w1 << 16)) +
                             __uint_as_float((u & 1) ? (w2 & 0xFFFF0000u) : (w2 << 16))) +
                            __uint_as_float((u & 1) ? (w3 & 0xFFFF0000u) : (w3 << 16));
            acc[t][8 * gp + u] += r;
          }
        }
    } else if (res) {
      const float* rsrc = res + goff;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 r = *reinterpret_cast<const f32x4*>(rsrc + t * 1024 + g * 256);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[t][4 * g + e] += r[e];
        }
    }
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
    float* dst = out_f32 ? out_f32 + goff : nullptr;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ch = t * 32 + 8 * g + 4 * cx.kh;
        const f32x4 a = *reinterpret_cast<const f32x4*>(ga + ch);
        const f32x4 b = *reinterpret_cast<const f32x4*>(be + ch);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[t][4 * g + e] * (rstd * a[e]) + b[e];
          acc[t][4 * g + e] = v[e];
        }
        if (dst) *reinterpret_cast<f32x4*>(dst + t * 1024 + g * 256) = v;
      }
      store_tile_sb(acc[t], out_sb, 256, cx.m_base, t, cx.lane);
    }
  }
};

}  // namespace b3
}  // namespace ddp
