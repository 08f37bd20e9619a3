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

namespace ddp {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int GEMM_BM = 128;
constexpr int GEMM_BK = 32;
constexpr int GEMM_LDS = GEMM_BK + 4;   // padded row stride (floats)
constexpr int GEMM_THREADS = 256;

template <int NT>
constexpr size_t gemm_lds_bytes() {
  return size_t(GEMM_BM + NT * 32) * GEMM_LDS * sizeof(float);
}

// Per-lane view handed to an epilogue: acc[t][r] is the value of token `m`, channel
//   n0 + t*32 + 8*(r>>2) + 4*kh + (r&3);   i.e. for g=r>>2 a float4 at channel n0+t*32+8g+4kh.
struct LaneCtx {
  int m;       // global token row of this lane (may be >= M: not valid)
  bool valid;  // m < M
  int n0;      // first channel of the block tile
  int kh;      // lane >> 5
};

__device__ __forceinline__ float half_sum(float v) { return v + __shfl_xor(v, 32, 64); }

// TAG only names the call site (value_proj, fc1, ...) so that rocprofv3 reports each separately.
template <int NT, class Epi, int TAG>
__global__ void __launch_bounds__(GEMM_THREADS, 2)
k_gemm_tok(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw, int M, int N, int K,
           int n_tiles_n, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Ws = smem + GEMM_BM * GEMM_LDS;

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
  epi.template run<NT>(acc, cx);
}

// number of blocks for the XCD-grouped 1-D grid used by k_gemm_tok
inline int gemm_grid(int M, int n_tiles_n) {
  int mt = (M + GEMM_BM - 1) / GEMM_BM;
  int mt8 = (mt + 7) / 8 * 8;
  return mt8 * n_tiles_n;
}

// ------------------------------------------------------------------------------------------------
// Epilogues
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_erf(float x) { return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f)); }

// out[m][n] = acc + bias[n] (+ add[(m / add_div) ... row map][n]) ; optional exact GELU.
// Row map for the addend: tokens of all r noise replicas of one image share the image's x
// projection: add_row = (m / (r*N)) * N + m % N  (rn = r*N).
struct EpiBias {
  const float* bias;  // (N) or nullptr
  const float* add;   // (rows, ld_add) or nullptr
  int ld_add;
  int rn, n_tok;      // r*N and N for the addend row map (rn == n_tok == 0 -> identity)
  float* out;
  int ldo;
  int n_valid;        // channels >= n_valid are not stored
  int gelu;

  template <int NT>
  __device__ __forceinline__ void run(f32x16 (&acc)[NT], const LaneCtx& cx) const {
    if (!cx.valid) return;
    size_t arow = 0;
    if (add) arow = rn ? (size_t(cx.m / rn) * n_tok + cx.m % n_tok) : size_t(cx.m);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ch = cx.n0 + t * 32 + 8 * g + 4 * cx.kh;
        if (ch >= n_valid) continue;
        f32x4 v = {acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
        if (ch + 4 <= n_valid) {
          if (bias) v += *reinterpret_cast<const f32x4*>(bias + ch);
          if (add) v += *reinterpret_cast<const f32x4*>(add + arow * ld_add + ch);
          if (gelu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
          }
          *reinterpret_cast<f32x4*>(out + size_t(cx.m) * ldo + ch) = v;
        } else {  // ragged last float4 of the row (e.g. 150 classes)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (ch + e < n_valid) {
              float x = v[e];
              if (bias) x += bias[ch + e];
              if (add) x += add[arow * ld_add + ch + e];
              if (gelu) x = gelu_erf(x);
              out[size_t(cx.m) * ldo + ch + e] = x;
            }
          }
        }
      }
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
    const size_t row = cx.valid ? size_t(cx.m) : 0;
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ch = t * 32 + 8 * g + 4 * cx.kh;
        const f32x4 b = *reinterpret_cast<const f32x4*>(bias + ch);
        const f32x4 r = *reinterpret_cast<const f32x4*>(res + row * ldres + ch);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = acc[t][4 * g + e] + b[e] + r[e];
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
        q += d * d;
      }
    const float rstd = 1.0f / sqrtf(half_sum(q) * (1.0f / 256.0f) + 1e-5f);
    if (!cx.valid) return;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ch = t * 32 + 8 * g + 4 * cx.kh;
        const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + ch);
        const f32x4 be = *reinterpret_cast<const f32x4*>(beta + ch);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (acc[t][4 * g + e] - mean) * rstd * ga[e] + be[e];
        if (film) {
          const f32x4 sc = *reinterpret_cast<const f32x4*>(film + ch);
          const f32x4 sh = *reinterpret_cast<const f32x4*>(film + 256 + ch);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] * (sc[e] + 1.0f) + sh[e];
        }
        *reinterpret_cast<f32x4*>(out + size_t(cx.m) * ldo + ch) = v;
      }
  }
};

// Sampling projection epilogue (multi_scale_deform_attn.py:319-334 for one level).
// Input columns: 0..63 offsets [head][point][x,y], 64..95 attention logits [head][point].
//   raw = acc + PY[i][col] + PX[j][col]        (positional term folded through the projection:
//                                               W(q+pos) = Wq + W_y pos_y(i) + W_x pos_x(j); bias in PY)
//   offsets -> pixel-unit sample coordinates  x = j + o_x, y = i + o_y
//   logits  -> softmax over the 4 points of a head (one lane holds them as a float4)
struct EpiSamp {
  const float* py;  // (h, 96)
  const float* px;  // (w, 96)
  int n_tok, w;     // tokens per map, map width
  float* out;       // (M, 96)

  template <int NT>
  __device__ __forceinline__ void run(f32x16 (&acc)[NT], const LaneCtx& cx) const {
    static_assert(NT == 3, "sampling projection is 96 columns");
    if (!cx.valid) return;
    const int n = cx.m % n_tok;
    const int i = n / w;
    const int jx = n - i * w;
    const float fi = float(i), fj = float(jx);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ch = t * 32 + 8 * g + 4 * cx.kh;
        const f32x4 a = *reinterpret_cast<const f32x4*>(py + i * 96 + ch);
        const f32x4 b = *reinterpret_cast<const f32x4*>(px + jx * 96 + ch);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[t][4 * g + e] + (a[e] + b[e]);
        if (t < 2) {
          v[0] += fj; v[1] += fi; v[2] += fj; v[3] += fi;
        } else {
          const float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = expf(v[e] - mx);
          const float den = v[0] + v[1] + v[2] + v[3];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] /= den;
        }
        *reinterpret_cast<f32x4*>(out + size_t(cx.m) * 96 + ch) = v;
      }
  }
};

}  // namespace ddp
