// ddp_kernels.hip - the memory-bound kernels of the DDP sampling loop (gfx950, wave64).
//
// All activations are token-major fp32 (rows = tokens, 256 contiguous channels = 1 KiB per row), so
// one wave64 owns one token row as 64 x float4: every load / store of a row is a single fully
// coalesced 1 KiB access and per-token reductions are wave reductions.
#include <math.h>
#include <type_traits>
#include "ddp_internal.h"

namespace ddp {

using f32x4 = __attribute__((ext_vector_type(4))) float;

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ------------------------------------------------------------------------------------------------
// NCHW (R, C, N) -> token-major (R, N, C): 64x64 tiles through LDS (stride 65: conflict-free both ways)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_nchw_to_tok(const float* __restrict__ in, float* __restrict__ out, int C,
                                                      int N) {
  __shared__ float tile[64][65];
  const int r = blockIdx.z;
  const int n0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const float* src = in + size_t(r) * C * N;
  float* dst = out + size_t(r) * N * C;
  for (int cc = ty; cc < 64; cc += 4) {
    const int c = c0 + cc, n = n0 + tx;
    tile[cc][tx] = (c < C && n < N) ? src[size_t(c) * N + n] : 0.f;
  }
  __syncthreads();
  for (int nn = ty; nn < 64; nn += 4) {
    const int n = n0 + nn, c = c0 + tx;
    if (n < N && c < C) dst[size_t(n) * C + c] = tile[tx][nn];
  }
}

// fragment-major offset of (row m, channel ch) in a 256-channel buffer (gemm_f32.h)
__device__ __forceinline__ size_t blk_off256(int m, int ch) {
  return size_t(m >> 5) * 8192 + (ch >> 5) * 1024 + ((ch & 31) >> 3) * 256 + ((ch & 7) >> 2) * 128 + (m & 31) * 4 + (ch & 3);
}

// row-major (rows,256) -> fragment-major: one wave per token, lane = 4 channels
__global__ void __launch_bounds__(256) k_row_to_blk(const float* __restrict__ in, float* __restrict__ out, int rows) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= rows) return;
  const f32x4 v = *reinterpret_cast<const f32x4*>(in + size_t(m) * 256 + lane * 4);
  *reinterpret_cast<f32x4*>(out + blk_off256(m, lane * 4)) = v;
}

// exact 3-way truncation split of an fp32 into bf16 pieces (see gemm_bf16x3.h)
__device__ __forceinline__ void split3(float x, unsigned short& p1, unsigned short& p2, unsigned short& p3) {
  const unsigned xb = __float_as_uint(x);
  const unsigned hb = xb & 0xFFFF0000u;
  const float r = x - __uint_as_float(hb);
  const unsigned mb = __float_as_uint(r) & 0xFFFF0000u;
  const float r2 = r - __uint_as_float(mb);
  p1 = (unsigned short)(hb >> 16);
  p2 = (unsigned short)(mb >> 16);
  p3 = (unsigned short)(__float_as_uint(r2) >> 16);
}

// W fp32 (rows, ld) -> Wp[c][row][16b + 8h + u] = piece_c(W[row][16b + 8*(u/4) + 4h + (u%4)])
__global__ void k_split_weights(const float* __restrict__ W, int ld, int rows, int K, unsigned short* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (row, 8-slot group)
  const int groups = K / 8;
  if (idx >= rows * groups) return;
  const int row = idx / groups, gq = idx - row * groups;
  const int b = gq >> 1, h = gq & 1;
  const size_t comp = size_t(rows) * K;
  unsigned short* dst = out + size_t(row) * K + 16 * b + 8 * h;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const float x = W[size_t(row) * ld + 16 * b + 8 * (u >> 2) + 4 * h + (u & 3)];
    unsigned short p1, p2, p3;
    split3(x, p1, p2, p3);
    dst[u] = p1;
    dst[comp + u] = p2;
    dst[2 * comp + u] = p3;
  }
}

// Stage images of the layer kernel's weight stream (layer_bf16x3.h): one 48 KiB image per (row block, k block) of a
// split weight matrix, byte-for-byte what the LDS ring slot holds ([piece 3][row][16-B slot], slots XOR-swizzled).
//   tall = 1: 64 rows x 128 k, 16 slots per row, slot = chunk ^ (row & 15)
//   tall = 0: 256 rows x 32 k,  4 slots per row, slot = chunk ^ ((row >> 2) & 3)
//   tall = 2: "split-K" tall image of 32 outputs x 256 k: image row r holds output a + (r & 31), k half r >> 5
// image index = base + rowblk*c + (kblk>>1)*a + (kblk&1)*b (tall = 2: base); rows >= rows_valid are zero.
__global__ void __launch_bounds__(256) k_build_stages(const unsigned short* __restrict__ Wp, size_t comp_stride, int K,
                                                       int rows_valid, int tall, int n_kblk, int base, int a, int b, int c,
                                                       unsigned char* __restrict__ stream) {
  const int st = blockIdx.y;                     // (rowblk, kblk)
  const int rowblk = st / n_kblk, kblk = st - rowblk * n_kblk;
  const int idx = blockIdx.x * 256 + threadIdx.x;   // 16-B chunk of the image: 3072 per stage
  if (idx >= 3072) return;
  const int comp = idx >> 10, rem = idx & 1023;
  int row, chunk, slot, grow, kcol;
  if (tall) {
    row = rem >> 4; chunk = rem & 15; slot = chunk ^ (row & 15);
    grow = tall == 2 ? a + (row & 31) : rowblk * 64 + row;
    kcol = (tall == 2 ? (row >> 5) : kblk) * 128 + 8 * chunk;
  } else {
    row = rem >> 2; chunk = rem & 3; slot = chunk ^ ((row >> 2) & 3);
    grow = rowblk * 256 + row;
    kcol = kblk * 32 + 8 * chunk;
  }
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if (grow < rows_valid) v = *reinterpret_cast<const uint4*>(Wp + size_t(comp) * comp_stride + size_t(grow) * K + kcol);
  const int image = tall == 2 ? base : base + rowblk * c + (kblk >> 1) * a + (kblk & 1) * b;
  const int per_row = tall ? 16 : 4;
  *reinterpret_cast<uint4*>(stream + size_t(image) * 49152 + comp * 16384 + (row * per_row + slot) * 16) = v;
}

// fp32 row-major (rows, C) -> SB: one half-wave per token, lane (b, h) handles the 8 channels of its slot
__global__ void __launch_bounds__(256) k_row_to_sb(const float* __restrict__ in, int ld, unsigned short* __restrict__ out,
                                                    int rows, int C) {
  const int slots = C / 8;                                  // (b, h) slots per token
  const long idx = long(blockIdx.x) * blockDim.x + threadIdx.x;
  if (idx >= long(rows) * slots) return;
  const int m = int(idx / slots), sl = int(idx - long(m) * slots);
  const int b = sl >> 1, h = sl & 1;
  const float* src = in + size_t(m) * ld + 16 * b + 4 * h;
  const f32x4 lo4 = *reinterpret_cast<const f32x4*>(src);
  const f32x4 hi4 = *reinterpret_cast<const f32x4*>(src + 8);
  unsigned short p[3][8];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    split3(lo4[u], p[0][u], p[1][u], p[2][u]);
    split3(hi4[u], p[0][4 + u], p[1][4 + u], p[2][4 + u]);
  }
  char* base = reinterpret_cast<char*>(out) + size_t(m >> 5) * C * 192 + size_t(b) * 3 * 1024 + (h * 32 + (m & 31)) * 16;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    uint4 v;
    v.x = p[c][0] | (unsigned(p[c][1]) << 16);
    v.y = p[c][2] | (unsigned(p[c][3]) << 16);
    v.z = p[c][4] | (unsigned(p[c][5]) << 16);
    v.w = p[c][6] | (unsigned(p[c][7]) << 16);
    *reinterpret_cast<uint4*>(base + c * 1024) = v;
  }
}

// NCHW (R, C, N) fp32 -> SB directly (k_nchw_to_tok + k_row_to_sb in one pass: the token-major fp32 copy - 1 KiB per
// token written and read back - is never made).  A block owns 64 consecutive tokens of the flattened (R*N) row space
// (two SB groups; a group may straddle two maps) x 64 channels (four K16 blocks): coalesced 256-B reads along n into
// an LDS tile, then 16-B SB slots out.
__global__ void __launch_bounds__(256) k_nchw_to_sb(const float* __restrict__ in, unsigned short* __restrict__ out, int C, int N,
                                                     int rows) {
  __shared__ float tile[64][65];
  const int m0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  {
    const int m = m0 + tx;
    const int r = m / N, n = m - r * N;
    const float* src = in + (size_t(r) * C + c0) * N + n;
    for (int cc = ty; cc < 64; cc += 4) tile[cc][tx] = (m < rows && c0 + cc < C) ? src[size_t(cc) * N] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int item = it * 256 + threadIdx.x;          // (group gi, K16 block bl, lane l2 = (hh, j))
    const int l2 = item & 63, bl = (item >> 6) & 3, gi = item >> 8;
    const int j = l2 & 31, hh = l2 >> 5;
    if (c0 + 16 * bl >= C) continue;
    unsigned short p[3][8];
#pragma unroll
    for (int u = 0; u < 8; ++u) split3(tile[16 * bl + 8 * (u >> 2) + 4 * hh + (u & 3)][gi * 32 + j], p[0][u], p[1][u], p[2][u]);
    char* base = reinterpret_cast<char*>(out) + size_t((m0 >> 5) + gi) * C * 192 + size_t((c0 >> 4) + bl) * 3 * 1024 + l2 * 16;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      uint4 v;
      v.x = p[c][0] | (unsigned(p[c][1]) << 16);
      v.y = p[c][2] | (unsigned(p[c][3]) << 16);
      v.z = p[c][4] | (unsigned(p[c][5]) << 16);
      v.w = p[c][6] | (unsigned(p[c][7]) << 16);
      *reinterpret_cast<uint4*>(base + c * 1024) = v;
    }
  }
}

// LayerNorm affine pre-multiplied by FiLM: out[i] = {gamma*(scale+1) | beta*(scale+1)+shift}, i over (S*L)
__global__ void k_fold_affine(const float* __restrict__ gamma, const float* __restrict__ beta,
                              const float* __restrict__ film, float* __restrict__ out) {
  const int c = threadIdx.x;          // 256 channels
  const float* f = film + size_t(blockIdx.x) * 512;
  const float sc = f[c] + 1.0f, sh = f[256 + c];
  out[size_t(blockIdx.x) * 512 + c] = gamma[c] * sc;
  out[size_t(blockIdx.x) * 512 + 256 + c] = beta[c] * sc + sh;
}

// ------------------------------------------------------------------------------------------------
// Deformable attention core (one level): one wave per token; lane = (head = lane>>3, 4 channels).
// 8 lanes of a head read one 128-B line per bilinear tap; the wave writes its token's 1 KiB row.
// Sample coordinates are already in pixel units (x = j + o_x, y = i + o_y): with one level the
// (j+0.5)/w reference point, the /w normaliser and grid_sample's align_corners=False mapping cancel
// (multi_scale_deform_attn.py:329-334,123-128).  Taps outside the map contribute 0.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_msda_gather(const float* __restrict__ value, const float* __restrict__ samp,
                                                      float* __restrict__ out, int rows, int n_tok, int h, int w) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= rows) return;
  const int hd = lane >> 3;
  const int cq = (lane & 7) * 4;
  const float* vbase = value + size_t(m / n_tok) * n_tok * 256 + hd * 32 + cq;
  const float* sp = samp + size_t(m) * DDP_SAMP_STRIDE;
  const f32x4 c01 = *reinterpret_cast<const f32x4*>(sp + hd * 8);
  const f32x4 c23 = *reinterpret_cast<const f32x4*>(sp + hd * 8 + 4);
  const f32x4 aw = *reinterpret_cast<const f32x4*>(sp + 64 + hd * 4);
  const float xs[4] = {c01[0], c01[2], c23[0], c23[2]};
  const float ys[4] = {c01[1], c01[3], c23[1], c23[3]};
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float x = xs[p], y = ys[p];
    const float xf = floorf(x), yf = floorf(y);
    const float fx = x - xf, fy = y - yf;
    const int x0 = int(xf), y0 = int(yf);
    const bool vx0 = (x0 >= 0) & (x0 < w), vx1 = (x0 + 1 >= 0) & (x0 + 1 < w);
    const bool vy0 = (y0 >= 0) & (y0 < h), vy1 = (y0 + 1 >= 0) & (y0 + 1 < h);
    const int xc0 = min(max(x0, 0), w - 1), xc1 = min(max(x0 + 1, 0), w - 1);
    const int yc0 = min(max(y0, 0), h - 1), yc1 = min(max(y0 + 1, 0), h - 1);
    const float w00 = (vx0 & vy0) ? (1.f - fy) * (1.f - fx) : 0.f;
    const float w01 = (vx1 & vy0) ? (1.f - fy) * fx : 0.f;
    const float w10 = (vx0 & vy1) ? fy * (1.f - fx) : 0.f;
    const float w11 = (vx1 & vy1) ? fy * fx : 0.f;
    const f32x4 v00 = *reinterpret_cast<const f32x4*>(vbase + size_t(yc0 * w + xc0) * 256);
    const f32x4 v01 = *reinterpret_cast<const f32x4*>(vbase + size_t(yc0 * w + xc1) * 256);
    const f32x4 v10 = *reinterpret_cast<const f32x4*>(vbase + size_t(yc1 * w + xc0) * 256);
    const f32x4 v11 = *reinterpret_cast<const f32x4*>(vbase + size_t(yc1 * w + xc1) * 256);
    const f32x4 s = v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11;
    acc += s * aw[p];
  }
  *reinterpret_cast<f32x4*>(out + size_t(m) * 256 + hd * 32 + cq) = acc;
}

// Same gather, output written as SB (the split fragment-major A operand of output_proj): a block owns one
// 32-token group, each wave gathers 8 of its tokens into a padded fp32 LDS tile (row stride 260 dwords: a
// ds_read_b128 lane group of 16 tokens hits 16 distinct 4-bank sets), then the block emits the group's
// 16 K16-blocks x 3 pieces x 64 lanes as fully coalesced 16-B stores.  Replaces gather + k_row_to_sb
// (saves a 1 KiB write + 1 KiB read per token and a launch).
constexpr int GSB_LD = 260;
constexpr int GSB_WAVES = 8;                     // 32 tokens per block, 32 / GSB_WAVES per wave
__global__ void __launch_bounds__(64 * GSB_WAVES) k_msda_gather_sb(const float* __restrict__ value, const float* __restrict__ samp,
                                                         unsigned short* __restrict__ out_sb, int rows, int n_tok, int h,
                                                         int w) {
  __shared__ __attribute__((aligned(16))) float tile[32 * GSB_LD];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int hd = lane >> 3;
  const int cq = (lane & 7) * 4;
  const int m_base = blockIdx.x * 32;
#pragma unroll
  for (int it = 0; it < 32 / GSB_WAVES; ++it) {
    const int jj = wave * (32 / GSB_WAVES) + it;
    const int m = m_base + jj;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (m < rows) {
      const float* vbase = value + size_t(m / n_tok) * n_tok * 256 + hd * 32 + cq;
      const float* sp = samp + size_t(m) * DDP_SAMP_STRIDE;
      const f32x4 c01 = *reinterpret_cast<const f32x4*>(sp + hd * 8);
      const f32x4 c23 = *reinterpret_cast<const f32x4*>(sp + hd * 8 + 4);
      const f32x4 aw = *reinterpret_cast<const f32x4*>(sp + 64 + hd * 4);
      const float xs[4] = {c01[0], c01[2], c23[0], c23[2]};
      const float ys[4] = {c01[1], c01[3], c23[1], c23[3]};
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const float x = xs[p], y = ys[p];
        const float xf = floorf(x), yf = floorf(y);
        const float fx = x - xf, fy = y - yf;
        const int x0 = int(xf), y0 = int(yf);
        const bool vx0 = (x0 >= 0) & (x0 < w), vx1 = (x0 + 1 >= 0) & (x0 + 1 < w);
        const bool vy0 = (y0 >= 0) & (y0 < h), vy1 = (y0 + 1 >= 0) & (y0 + 1 < h);
        const int xc0 = min(max(x0, 0), w - 1), xc1 = min(max(x0 + 1, 0), w - 1);
        const int yc0 = min(max(y0, 0), h - 1), yc1 = min(max(y0 + 1, 0), h - 1);
        const float w00 = (vx0 & vy0) ? (1.f - fy) * (1.f - fx) : 0.f;
        const float w01 = (vx1 & vy0) ? (1.f - fy) * fx : 0.f;
        const float w10 = (vx0 & vy1) ? fy * (1.f - fx) : 0.f;
        const float w11 = (vx1 & vy1) ? fy * fx : 0.f;
        const f32x4 v00 = *reinterpret_cast<const f32x4*>(vbase + size_t(yc0 * w + xc0) * 256);
        const f32x4 v01 = *reinterpret_cast<const f32x4*>(vbase + size_t(yc0 * w + xc1) * 256);
        const f32x4 v10 = *reinterpret_cast<const f32x4*>(vbase + size_t(yc1 * w + xc0) * 256);
        const f32x4 v11 = *reinterpret_cast<const f32x4*>(vbase + size_t(yc1 * w + xc1) * 256);
        const f32x4 sv = v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11;
        acc += sv * aw[p];
      }
    }
    *reinterpret_cast<f32x4*>(tile + jj * GSB_LD + hd * 32 + cq) = acc;
  }
  __syncthreads();
  char* gbase = reinterpret_cast<char*>(out_sb) + size_t(blockIdx.x) * 256 * 192;
#pragma unroll
  for (int r = 0; r < 16 / GSB_WAVES; ++r) {
    const int item = r * 64 * GSB_WAVES + threadIdx.x;        // (b, lane') with lane' = (h', j)
    const int b = item >> 6, l2 = item & 63;
    const int j = l2 & 31, hh = l2 >> 5;
    const float* src = tile + j * GSB_LD + 16 * b + 4 * hh;
    const f32x4 lo4 = *reinterpret_cast<const f32x4*>(src);
    const f32x4 hi4 = *reinterpret_cast<const f32x4*>(src + 8);
    unsigned short p[3][8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      split3(lo4[u], p[0][u], p[1][u], p[2][u]);
      split3(hi4[u], p[0][4 + u], p[1][4 + u], p[2][4 + u]);
    }
    char* base = gbase + size_t(b) * 3 * 1024 + l2 * 16;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      uint4 v;
      v.x = p[c][0] | (unsigned(p[c][1]) << 16);
      v.y = p[c][2] | (unsigned(p[c][3]) << 16);
      v.z = p[c][4] | (unsigned(p[c][5]) << 16);
      v.w = p[c][6] | (unsigned(p[c][7]) << 16);
      *reinterpret_cast<uint4*>(base + c * 1024) = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// LDS-staged gather ("deformable-offset gathers staged through LDS", north_star).  The tap traffic of the wave-per-
// token kernels above is 16 KiB of L2 -> L1 requests per token for 1 KiB of unique value data; here a block owns a
// GL_TH x GL_TW tile of tokens of one map and ONE head: it stages that head's 128-B slice of the value map for the
// tile plus a halo into LDS by LDS-DMA (one 1-KiB instruction = 8 pixels x 128 B, coalesced), then serves the 16 bilinear
// taps of every token from LDS (ds_read_b128, the four corners as immediate offsets of one address).  Fill traffic is
// (GL_WW x GL_WH) / (GL_TH x GL_TW) = 2.4 x 128 B per (token, head) instead of 2 KiB, and - heads pinned to XCDs, see the
// kernel - the 1.4 x that is halo comes out of L2, not HBM.
//   * The window of a head is centred on the tile's mean sampling offset of that head: the reference initialises the
//     offsets as a ring of radius 1..4 px per head (multi_scale_deform_attn.py:233-244), so the points of a head spread
//     +-1.5 px around their mean and GL_HALO = 3 leaves 1.5 px for the learned, content-dependent part.  The fill starts
//     from the data-independent part of that mean (bias + positional term at the tile centre, read from the separable
//     tables) and is repeated only if the tile's actual mean - a deterministic block reduction over its 4 x 128 sample
//     points - is 2 px or more away (r03b: 0.196 -> 0.185 ms at C2, 0.387 -> 0.357 at Cityscapes size, the same with
//     spread-out offsets).
//   * A (token, point) whose four corners are not all inside the window is served from the zero-padded map in global
//     memory (an 8-token group with such a point takes a lane-divergent mixed path): any offset is handled, only slower.
//     Both paths load the same values and combine them in the same order, so the result does not depend on which one ran.
//   * wave = 8 x-adjacent tokens x 8 channel quads of ONE head; the 32 channels of
//     (token, head) are K16 blocks 2hd, 2hd+1 of the SB operand: lane pairs (q, q^2) are joined by one DPP quad
//     permute per register and written straight to HBM as 16-B slots (128-B runs of 8 tokens) - no LDS output tile.
// Addresses: per-image scalar base + 32-bit offsets (an image's padded map is < 4 GiB).
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) unsigned char lds_byte_t;
typedef __attribute__((address_space(3))) f32x4 lds_f32x4_t;
constexpr int GL_THREADS = 512;
// tile GL_TH x GL_TW tokens (128 per block: 8 waves x 2 runs of 8 x-adjacent tokens), halo GL_HALO:
//   window width  GL_WW = TW + 2 HALO + 1 - GL_TRIM.  TRIM = 0: floor(x) in [x0 + mean - HALO, x0 + TW - 1 + mean + HALO] and its +1
//   corner.  TRIM = 1 (the product): the +1 corner comes out of the HIGH-side halo - floor(x) - (x0 + mean) in [-HALO, TW - 2 + HALO].
//   floor() already biases the corner index downward (a point 1.5 px above the mean has its top-left corner 1 px above it), so the
//   lost column / row is the one used least, and 22 x 14 px = 39 936 B + 64 B is exactly a QUARTER of the CU's 160 KiB: four blocks per
//   CU instead of three, at 47 registers (the compiler budgets for the occupancy the LDS size allows).  Same box, ms per launch, init /
//   trained_like profile / Cityscapes size: TRIM 0 0.154 / 0.237 / 0.310, **TRIM 1 0.143 / 0.219 / 0.289**, TRIM 2 0.149 / 0.226,
//   TRIM 3 0.163 / 0.232 (profiles/r06g_*, r06h_*; bit-identical outputs: which memory serves a tap never changes its value).
#ifndef DDP_GL_TRIM
#define DDP_GL_TRIM 1
#endif
template <int TH, int TW, int HALO, int NT = GL_THREADS>
struct GlGeom {
  static constexpr int TOK = TH * TW, NW = NT / 64, NG = TOK / (8 * NW);   // NG runs of 8 x-adjacent tokens per wave
  static_assert(TW % 8 == 0 && NG * 8 * NW == TOK && TOK % (NT / 4) == 0, "rows of whole 8-token runs, whole runs per wave");
  static constexpr int WW = TW + 2 * HALO + 1 - DDP_GL_TRIM, WH = TH + 2 * HALO + 1 - DDP_GL_TRIM;
  static constexpr int PIX = WW * WH;               // window pixels x 128 B
  static constexpr int DMA = (PIX + 7) / 8;         // LDS-DMA instructions of 8 pixels (1 KiB)
  static constexpr int WIN_B = DMA * 1024;
};

__device__ __forceinline__ void gl_dma(const float* gbase, unsigned byte_off, unsigned lds_dst) {
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

// DDP_GL_V: build-time level of the block's latency chain (same-box A/B: scripts/variant_build.sh x -DDDP_GL_V=0):
//   0  round-3 baseline: origin guess (scalar table loads) -> sample-table loads -> fill by inline-asm LDS-DMA -> vmcnt(0) ->
//      butterfly reduction of the mean by ds_bpermute -> barrier
//   1  sample-table loads issued FIRST (they depend on nothing but the kernel arguments: in flight under the scalar loads of
//      the guess), mean reduced by DPP row operations (no trip through the LDS crossbar, no lgkmcnt waits)
//   2  1 + a point's corner offset and weights reach the token's 8 lanes by DPP quad_perm broadcasts instead of ds_swizzle
// (The fill stays inline asm: __builtin_amdgcn_global_load_lds is counted by the compiler's s_waitcnt insertion as a FLAT
// operation that touches both address spaces - "pending flat": every later wait becomes vmcnt(0) lgkmcnt(0) - so it buys no
// finer wait than the asm form, which that pass cannot see at all.)
#ifndef DDP_GL_V
#define DDP_GL_V 2
#endif
// window halo of the product launch (launch_msda_gather_sb_pad): 3, trimmed -> 22 x 14 px = 39 KiB, four blocks per CU (until round 5:
// 23 x 15 px = 44 KiB, three blocks - the figures below were measured in that form).  Round 5, same
// box (profiles/r05c_ab_gather_*.txt; ms per launch at C2, init / trained_like weight profile): halo 3 0.151 / 0.230; halo 5
// (64 KiB, two blocks per CU) 0.200 / 0.274; halo 6 (77 KiB) 0.209 / 0.254 - with content-dependent offsets of +- 2.4 px
// nearly every 8-token group has SOME corner outside even a +-6 px window, so a larger window only costs occupancy.  The mixed
// path itself with the four points unrolled on DPP broadcasts: hipcc spills in the divergent region at the 85 registers three
// blocks per CU allow; the same through flat loads (per-lane address = shared aperture or global, one instruction stream, 80
// registers, no spill) 0.149 / 0.254 - slower than the rolled ds_bpermute loop below on the profile it was meant for; per-point fma
// chains with an LDS pass for every point and a batched global pass (two points' loads in flight) for the points outside the window:
// 0.151 / 0.227 against 0.150 / 0.229 (profiles/r05i_*) - the out-of-window taps cost what they cost in the texture path, not in
// the loop around them.  Tile / halo pairs at 256 threads (profiles/r05l_*): 8x8 + halo 5 0.203 / 0.284, 8x8 + halo 4 0.182 / 0.260,
// 8x16 + halo 4 0.189 / 0.257 against 0.152 / 0.232: the shipped shape is the fastest on BOTH profiles, so there is nothing for a
// spread-adaptive window choice to pick from.  Tile ORDER in super-columns of 8 / 4 / 2 tiles (the tile below 8 blocks away instead of
// tiles_x): 0.322 against 0.310 ms at Cityscapes size, 0.155-0.161 against 0.153 at C2 (profiles/r05n_*) - raster order stays.
#ifndef DDP_GL_HALO
#define DDP_GL_HALO 3
#endif

// sum over the 64 lanes by DPP: quad permutes, row_half_mirror, row_mirror (every lane of a row of 16 holds the row's sum),
// row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and 3 - lanes 48..63 hold the total
__device__ __forceinline__ float gl_wave_sum_hi(float v) {
  auto dpp = [](float x, auto ctrl, auto rows) __attribute__((always_inline)) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, decltype(rows)::value, 0xF, false));
  };
  using std::integral_constant;
  v += dpp(v, integral_constant<int, 0xB1>{}, integral_constant<int, 0xF>{});      // quad_perm [1,0,3,2]
  v += dpp(v, integral_constant<int, 0x4E>{}, integral_constant<int, 0xF>{});      // quad_perm [2,3,0,1]
  v += dpp(v, integral_constant<int, 0x141>{}, integral_constant<int, 0xF>{});     // row_half_mirror
  v += dpp(v, integral_constant<int, 0x140>{}, integral_constant<int, 0xF>{});     // row_mirror
  v += dpp(v, integral_constant<int, 0x142>{}, integral_constant<int, 0xA>{});     // row_bcast15 -> rows 1, 3
  v += dpp(v, integral_constant<int, 0x143>{}, integral_constant<int, 0xC>{});     // row_bcast31 -> rows 2, 3
  return v;
}

// Each of a token's 8 lanes does the coordinate arithmetic of ONE sample point (p = q & 3) and the results travel to the
// other lanes by ds_swizzle (a cross-lane move through the LDS crossbar, no storage) instead of every lane redoing all four
// points (0.214 -> 0.206 ms, r02d); the 3-way split of the result happens BEFORE the quad exchange (each lane splits its
// own four values).  Measured shapes (r02c/r02d, C2, ms per launch): 8x16 halo 3 at two blocks per CU 0.206; the same at
// three blocks per CU (80 registers: spills) 0.232; halo 4 0.215; halo 2 (more fallbacks) 0.251; 4x32 tiles 0.224; a
// double-buffered 16x16 tile with one 1024-thread block per CU 0.231; the wave-per-token kernels before it 0.278 - 0.314.
// r02m-r02o (same-box A/B, shipped shape 0.211): 256-token tiles 16x16 / 8x32 (NG = 4, two blocks per CU) 0.216 / 0.217;
// 8x8 tiles with 256 threads (five blocks per CU) 0.220; one dword per 128-B line of the NEXT head's window requested under
// this head's taps (L2 prefetch, asynchronous through an LDS-DMA sink) 0.240 - the extra requests cost more than they hide.
// Counters (profiles/r02o_gather_counters.txt): 65 % of the kernel's L2 requests miss (window lines are re-fetched by the
// neighbouring tiles: 4 MB of L2 per XCD turn over every ~5 us at this rate), FETCH x 2 + WRITE = 1.37 GB per launch
// = 6.5 TB/s through the fabric: the kernel is bound by HBM-side traffic at 1.78 x its algorithmic bytes (0.77 GB), not by
// LDS (IDX_ACTIVE 34 % of a CU's cycles, a quarter of those bank conflicts) or VALU (8 %).
// r02p: one head per block, head = XCD (below): 0.210 -> 0.197.  r02t, in that form: 8x8 tiles / 256 threads 0.197, 4x16 tiles /
// 256 threads 0.199 against 0.183 - 0.195 for 8x16 on the same box.  r02u: persistent blocks (3 or 6 per CU and head, tile loop,
// the next tile's table entries fetched under the current tile's taps) 0.213 - 0.217 against 0.197: one short block per
// (tile, head), scheduled by the hardware as LDS frees up, overlaps better than a block-level software pipeline.
// F32OUT: the result leaves as fp32 in the accumulator layout of the layer kernel ([32-token group][tile t = head][quad g]
// [lane = half * 32 + token][4]: this lane's four channels ARE one 16-B slot of it - no split, no quad exchange; 1 KiB per
// token instead of the 1.5 KiB of SB); the layer kernel's P0 splits it in its own filler slots (layer_bf16x3.h, DDP_S_F32).
template <int GL_TH, int GL_TW, int GL_HALO, int MINW, int NT = GL_THREADS, bool F32OUT = false>
__global__ void __launch_bounds__(NT, MINW) k_msda_gather_lds(const float* __restrict__ vpad, const float* __restrict__ samp,
                                                                    unsigned short* __restrict__ out_sb, int n_tok, int h, int w,
                                                                    int tiles_x, int tiles_y, int n_tiles, int m_total,
                                                                    const float* __restrict__ tab_y, const float* __restrict__ tab_x,
                                                                    int zero_guess) {
  using G = GlGeom<GL_TH, GL_TW, GL_HALO, NT>;
  constexpr int GL_WW = G::WW, GL_WH = G::WH, GL_PIX = G::PIX, GL_DMA = G::DMA, GL_WIN_B = G::WIN_B, NG = G::NG;
  __shared__ __attribute__((aligned(16))) unsigned char win[GL_WIN_B];
  __shared__ float msum[G::NW][2];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // ONE head per block, head = block id mod 8 = the XCD the block lands on (workgroups go round-robin over the 8 XCDs): an
  // XCD's L2 only ever holds its head's 128-B slices, and its ~96 resident blocks work on neighbouring tiles of the same head
  // at the same time, so the halo lines neighbouring tiles share are L2 hits instead of HBM re-reads (r02p: FETCH_SIZE x 2
  // 972 -> 375 MB per launch = the algorithmic 268 MB map + 100 MB table; 0.210 -> 0.197 ms at C2, 0.445 -> 0.374 at C3)
  const int tile = int(blockIdx.x) >> 3;
  if (tile >= n_tiles) return;
  const int hd = int(blockIdx.x) & 7;
  const int tpi = tiles_x * tiles_y;
  const int img = tile / tpi;
  const int t2 = tile - img * tpi;
  const int tyi = t2 / tiles_x;
  const int y0 = tyi * GL_TH, x0 = (t2 - tyi * tiles_x) * GL_TW;
  const int wp = w + 2;
  const float* vimg = vpad + size_t(img) * size_t(h + 2) * wp * 256;
  // head-major sample table [head][token][8 pixel coordinates | 4 attention weights] (written by the layer kernel's P3)
  const float* simg = samp + (size_t(hd) * m_total + size_t(img) * n_tok) * 12;
  const unsigned lds_win = (unsigned)(size_t)(lds_byte_t*)win;

  const int tk = lane >> 3, q = lane & 7;
  const float xmax = float(w), ymax = float(h);
  // this lane's NG tokens (groups g: runs of 8 x-adjacent tokens of the wave's share of the tile)
  int mtok[NG];
  bool tval[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int tl = wave * (NG * 8) + g * 8 + tk;
    const int gy = y0 + tl / GL_TW, gx = x0 + tl % GL_TW;
    tval[g] = gy < h && gx < w;
    mtok[g] = tval[g] ? gy * w + gx : 0;
  }
  const size_t img_tok = size_t(img) * n_tok;

  {
    // The window origin starts from the DATA-INDEPENDENT part of the head's mean sampling offset at the tile centre - the
    // offset bias plus the positional term, i.e. the separable tables the layer kernel's P3 adds (k_pos_tables: tab_y (h,96),
    // tab_x (w,96), bias folded in) - so the fill is issued at once, next to the table loads, instead of behind "sample table ->
    // block reduction -> origin".  The tile's actual mean is then formed from the coordinates the taps need anyway, and the
    // window is refilled only when it sits 2 px or more away from the guess (any origin is CORRECT - taps outside the window
    // take the mixed path - the mean only decides how many do).
    // this lane's sample point p = q & 3 of its two tokens (x, y, attention weight): in flight under the window fill
    float px_[NG], py_[NG], pw_[NG];
    auto table_loads = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const float* sp = simg + size_t(mtok[g]) * 12;
        px_[g] = sp[2 * (q & 3)];
        py_[g] = sp[2 * (q & 3) + 1];
        pw_[g] = sp[8 + (q & 3)];
      }
    };
    if constexpr (DDP_GL_V >= 1) table_loads();
    int ox, oy;
    {
      float gx = 0.f, gy = 0.f;
      if (!zero_guess) {
        const float* ty = tab_y + min(y0 + GL_TH / 2, h - 1) * 96 + hd * 8;
        const float* tx = tab_x + min(x0 + GL_TW / 2, w - 1) * 96 + hd * 8;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          gx += ty[2 * p] + tx[2 * p];
          gy += ty[2 * p + 1] + tx[2 * p + 1];
        }
        gx = fminf(fmaxf(gx * 0.25f, -32768.0f), 32768.0f);
        gy = fminf(fmaxf(gy * 0.25f, -32768.0f), 32768.0f);
      }
      ox = __builtin_amdgcn_readfirstlane(x0 + int(rintf(gx)) - GL_HALO);
      oy = __builtin_amdgcn_readfirstlane(y0 + int(rintf(gy)) - GL_HALO);
    }
    {
    if constexpr (DDP_GL_V == 0) table_loads();
    // ---- fill: window pixel idx = py * GL_WW + px <- padded map pixel (oy + 1 + py, ox + 1 + px), clamped into the map
    auto fill = [&]() __attribute__((always_inline)) {
      for (int k = wave; k < GL_DMA; k += G::NW) {
        int idx = k * 8 + tk;
        idx = idx < GL_PIX ? idx : GL_PIX - 1;
        const int py = idx / GL_WW, px = idx - py * GL_WW;
        const int gyp = min(max(oy + 1 + py, 0), h + 1), gxp = min(max(ox + 1 + px, 0), wp - 1);
        const unsigned off = unsigned(gyp * wp + gxp) * 1024u + unsigned(hd * 128 + q * 16);
        gl_dma(vimg, off, __builtin_amdgcn_readfirstlane(lds_win + unsigned(k) * 1024u));
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): this wave's DMA pieces (and its coordinate loads) landed
    };
    fill();
    {
      // the tile's mean offset from the coordinates in registers: lanes q < 4 hold one (token, point) each
      float sx = 0.f, sy = 0.f;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int tl = wave * (NG * 8) + g * 8 + tk;
        if (tval[g] && q < 4) {
          sx += px_[g] - float(x0 + tl % GL_TW);
          sy += py_[g] - float(y0 + tl / GL_TW);
        }
      }
      if constexpr (DDP_GL_V >= 1) {
        sx = gl_wave_sum_hi(sx);
        sy = gl_wave_sum_hi(sy);
        if (lane == 63) {
          msum[wave][0] = sx;
          msum[wave][1] = sy;
        }
      } else {
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          sx += __shfl_xor(sx, o, 64);
          sy += __shfl_xor(sy, o, 64);
        }
        if (lane == 0) {
          msum[wave][0] = sx;
          msum[wave][1] = sy;
        }
      }
      __syncthreads();                                       // the window (guess) is complete, the partial sums are visible
      float mx = 0.f, my = 0.f;
#pragma unroll
      for (int k = 0; k < G::NW; ++k) {
        mx += msum[k][0];
        my += msum[k][1];
      }
      const float inv = 1.0f / float(min(GL_TW, w - x0) * min(GL_TH, h - y0) * 4);
      mx = fminf(fmaxf(mx * inv, -32768.0f), 32768.0f);      // NaN / inf coordinates: any finite origin is fine (mixed path)
      my = fminf(fmaxf(my * inv, -32768.0f), 32768.0f);
      const int oxi = __builtin_amdgcn_readfirstlane(x0 + int(rintf(mx)) - GL_HALO);
      const int oyi = __builtin_amdgcn_readfirstlane(y0 + int(rintf(my)) - GL_HALO);
      if (abs(oxi - ox) >= 2 || abs(oyi - oy) >= 2) {        // block-uniform: every thread summed the same eight partials
        ox = oxi;
        oy = oyi;
        fill();                                              // (no wave reads the window before the barrier below)
        __syncthreads();
      }
    }
    // ---- taps
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      // tokens outside the map (ragged tile) sample the window centre with a result that is never stored
      const float xv = tval[g] ? px_[g] : float(ox + GL_HALO), yv = tval[g] ? py_[g] : float(oy + GL_HALO);
      const float x = __builtin_amdgcn_fmed3f(xv, -1.0f, xmax), y = __builtin_amdgcn_fmed3f(yv, -1.0f, ymax);
      const float xf = floorf(x), yf = floorf(y);
      const float fx = x - xf, fy = y - yf;
      const int ix = int(xf), iy = int(yf);
      const int rx = ix - ox, ry = iy - oy;
      const bool inwin = unsigned(rx) < unsigned(GL_WW - 1) && unsigned(ry) < unsigned(GL_WH - 1);
      // corner weights with the attention weight folded in: (aw gy) gx, (aw gy) fx, (aw fy) gx, (aw fy) fx
      const float wa = pw_[g] * (1.f - fy), wb = pw_[g] * fy;
      const float w00 = wa * (1.f - fx), w01 = wa * fx, w10 = wb * (1.f - fx), w11 = wb * fx;
      const bool staged = __builtin_amdgcn_ballot_w64(!inwin) == 0;       // wave-uniform: every corner of the 8 tokens is in LDS
      // byte offset of the point's top-left corner: in the window when all four corners are staged, else in the padded map of
      // this image
      const int aoff = inwin ? (ry * GL_WW + rx) * 128 : int(unsigned((iy + 1) * wp + ix + 1) * 1024u + unsigned(hd * 128));
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      // one sample point for the token's 8 lanes: four corner loads of 16 B, one explicit fma chain per channel (the same
      // arithmetic whichever memory the corners come from)
      auto corners_lds = [&](int o, f32x4& v00, f32x4& v01, f32x4& v10, f32x4& v11) __attribute__((always_inline)) {
        // explicit LDS address space: a generic pointer would let the two paths be merged into flat loads
        const lds_byte_t* a = (const lds_byte_t*)win + (o + q * 16);
        v00 = *reinterpret_cast<const lds_f32x4_t*>(a);
        v01 = *reinterpret_cast<const lds_f32x4_t*>(a + 128);
        v10 = *reinterpret_cast<const lds_f32x4_t*>(a + GL_WW * 128);
        v11 = *reinterpret_cast<const lds_f32x4_t*>(a + GL_WW * 128 + 128);
      };
      auto accumulate = [&](const f32x4& v00, const f32x4& v01, const f32x4& v10, const f32x4& v11, float a00, float a01, float a10,
                            float a11) __attribute__((always_inline)) {
        acc = __builtin_elementwise_fma(v00, f32x4{a00, a00, a00, a00}, acc);
        acc = __builtin_elementwise_fma(v01, f32x4{a01, a01, a01, a01}, acc);
        acc = __builtin_elementwise_fma(v10, f32x4{a10, a10, a10, a10}, acc);
        acc = __builtin_elementwise_fma(v11, f32x4{a11, a11, a11, a11}, acc);
      };
      if (staged) {
        auto point = [&](auto ppc) __attribute__((always_inline)) {
          // fetch point pp from lane (token, pp) of this token's 8 lanes: lane' = (lane & 0x18) | pp within each 32-lane half
          // (ds_swizzle bit mode: and_mask [4:0], or_mask [9:5], xor_mask [14:10])
          int o;
          float a00, a01, a10, a11;
          if constexpr (DDP_GL_V >= 2) {
            // lanes q and q + 4 of a token hold the SAME point (both loaded entry q & 3), so "lane (token, pp)" exists in
            // either quad of the token's 8 lanes: a DPP quad_perm broadcast (a VALU move) instead of a trip through the
            // LDS crossbar and its lgkmcnt wait
            constexpr int qp = decltype(ppc)::value * 0x55;
            o = __builtin_amdgcn_mov_dpp(aoff, qp, 0xF, 0xF, true);
            a00 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(w00), qp, 0xF, 0xF, true));
            a01 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(w01), qp, 0xF, 0xF, true));
            a10 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(w10), qp, 0xF, 0xF, true));
            a11 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(w11), qp, 0xF, 0xF, true));
          } else {
            constexpr int pat = (decltype(ppc)::value << 5) | 0x18;
            o = __builtin_amdgcn_ds_swizzle(aoff, pat);
            a00 = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(w00), pat));
            a01 = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(w01), pat));
            a10 = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(w10), pat));
            a11 = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(w11), pat));
          }
          f32x4 v00, v01, v10, v11;
          corners_lds(o, v00, v01, v10, v11);
          accumulate(v00, v01, v10, v11, a00, a01, a10, a11);
        };
        point(std::integral_constant<int, 0>{});
        point(std::integral_constant<int, 1>{});
        point(std::integral_constant<int, 2>{});
        point(std::integral_constant<int, 3>{});
      } else {
        // some corner of the group is outside the window.  One point at a time (rolled: the registers of one point); the
        // (token, point) pairs whose corners ARE staged still read LDS, only the others go to global memory - lane-divergent,
        // both sides run, but the L2 -> L1 tap traffic that makes the global path slow shrinks with the staged fraction
#pragma unroll 1
        for (int pp = 0; pp < 4; ++pp) {
          const int src = ((lane & 0x38) | pp) << 2;           // lane (token, pp) of this token's 8 lanes
          const int o = __builtin_amdgcn_ds_bpermute(src, aoff);
          const bool from_lds = __builtin_amdgcn_ds_bpermute(src, int(inwin)) != 0;
          const float a00 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(w00)));
          const float a01 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(w01)));
          const float a10 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(w10)));
          const float a11 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(w11)));
          f32x4 v00, v01, v10, v11;
          if (from_lds) {
            corners_lds(o, v00, v01, v10, v11);
          } else {
            const char* vb = reinterpret_cast<const char*>(vimg) + q * 16;
            const unsigned o00 = unsigned(o);
            v00 = *reinterpret_cast<const f32x4*>(vb + o00);
            v01 = *reinterpret_cast<const f32x4*>(vb + o00 + 1024u);
            v10 = *reinterpret_cast<const f32x4*>(vb + o00 + unsigned(wp) * 1024u);
            v11 = *reinterpret_cast<const f32x4*>(vb + o00 + unsigned(wp) * 1024u + 1024u);
          }
          accumulate(v00, v01, v10, v11, a00, a01, a10, a11);
        }
      }
      if constexpr (F32OUT) {
        if (tval[g]) {
          const int m = int(img_tok) + mtok[g];
          float* dst = reinterpret_cast<float*>(out_sb) + size_t(m >> 5) * 8192 + hd * 1024 + (q >> 1) * 256 + ((q & 1) * 32 + (m & 31)) * 4;
          *reinterpret_cast<f32x4*>(dst) = acc;
        }
        continue;
      }
      // exact 3-way split of this lane's four channels, THEN the quad exchange with lane q ^ 2 (the other half of the
      // 16-B slot): 6 packed dwords travel instead of 4 fp32 + a second split
      unsigned hh[4], mm[4], ll[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned xb = __float_as_uint(acc[u]);
        hh[u] = xb & 0xFFFF0000u;
        const float r = acc[u] - __uint_as_float(hh[u]);
        mm[u] = __float_as_uint(r) & 0xFFFF0000u;
        ll[u] = __float_as_uint(r - __uint_as_float(mm[u]));
      }
      unsigned own[3][2];
      own[0][0] = __builtin_amdgcn_perm(hh[1], hh[0], 0x07060302); own[0][1] = __builtin_amdgcn_perm(hh[3], hh[2], 0x07060302);
      own[1][0] = __builtin_amdgcn_perm(mm[1], mm[0], 0x07060302); own[1][1] = __builtin_amdgcn_perm(mm[3], mm[2], 0x07060302);
      own[2][0] = __builtin_amdgcn_perm(ll[1], ll[0], 0x07060302); own[2][1] = __builtin_amdgcn_perm(ll[3], ll[2], 0x07060302);
      unsigned oth[3][2];
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int e = 0; e < 2; ++e) oth[c][e] = unsigned(__builtin_amdgcn_mov_dpp(int(own[c][e]), 0x4E, 0xF, 0xF, true));
      if (tval[g] && (q & 2) == 0) {
        const int m = int(img_tok) + mtok[g];
        char* base = reinterpret_cast<char*>(out_sb) + size_t(m >> 5) * 256 * 192 + size_t(2 * hd + (q >> 2)) * 3 * 1024 +
                     ((q & 1) * 32 + (m & 31)) * 16;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          uint4 v;
          v.x = own[c][0];
          v.y = own[c][1];
          v.z = oth[c][0];
          v.w = oth[c][1];
          *reinterpret_cast<uint4*>(base + c * 1024) = v;
        }
      }
    }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Adapters that put the PRODUCT gather (k_msda_gather_lds) behind the plain interface of ddp_msda_forward (row-major value
// map, token-major sample table, row-major output): ddp_msda_forward_lds.  In the loop these conversions do not exist -
// the layer kernel's P3 writes the padded map and the head-major table, the next layer kernel's P0 reads SB.
// ------------------------------------------------------------------------------------------------
// value (R, h*w, 256) row-major -> interior of the zero-padded maps (R, h+2, w+2, 256); one wave per token
__global__ void __launch_bounds__(256) k_pad_value(const float* __restrict__ value, float* __restrict__ vpad, int rows, int n_tok,
                                                    int w, int h) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= rows) return;
  const int img = m / n_tok, t = m - img * n_tok;
  const int i = t / w, j = t - i * w;
  const size_t row = size_t(img) * (h + 2) * (w + 2) + size_t(i + 1) * (w + 2) + (j + 1);
  *reinterpret_cast<f32x4*>(vpad + row * 256 + lane * 4) = *reinterpret_cast<const f32x4*>(value + size_t(m) * 256 + lane * 4);
}
// token-major table rows of DDP_SAMP_STRIDE floats [64 coords (head, point, xy) | 32 weights (head, point)] -> head-major
// [head][token][8 coords | 4 weights]
__global__ void __launch_bounds__(256) k_samp_head_major(const float* __restrict__ samp, float* __restrict__ out, int rows) {
  const long idx = long(blockIdx.x) * 256 + threadIdx.x;        // (token, head)
  if (idx >= long(rows) * 8) return;
  const int m = int(idx >> 3), hd = int(idx & 7);
  const float* sp = samp + size_t(m) * DDP_SAMP_STRIDE;
  float* dst = out + (size_t(hd) * rows + m) * 12;
  *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(sp + hd * 8);
  *reinterpret_cast<f32x4*>(dst + 4) = *reinterpret_cast<const f32x4*>(sp + hd * 8 + 4);
  *reinterpret_cast<f32x4*>(dst + 8) = *reinterpret_cast<const f32x4*>(sp + 64 + hd * 4);
}
// SB (rows padded to 32, C channels) -> fp32 row-major: the three bf16 pieces of an element sum to its fp32 value exactly
__global__ void __launch_bounds__(256) k_sb_to_row(const unsigned short* __restrict__ in_sb, float* __restrict__ out, int rows, int C) {
  const int slots = C / 8;
  const long idx = long(blockIdx.x) * 256 + threadIdx.x;
  if (idx >= long(rows) * slots) return;
  const int m = int(idx / slots), sl = int(idx - long(m) * slots);
  const int b = sl >> 1, h = sl & 1;
  const char* base = reinterpret_cast<const char*>(in_sb) + size_t(m >> 5) * C * 192 + size_t(b) * 3 * 1024 + (h * 32 + (m & 31)) * 16;
  float v[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) v[u] = 0.f;
#pragma unroll
  for (int c = 2; c >= 0; --c) {                       // smallest piece first: every partial sum is exact
    const uint4 q = *reinterpret_cast<const uint4*>(base + c * 1024);
    const unsigned d[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] += __uint_as_float((u & 1) ? (d[u >> 1] & 0xFFFF0000u) : (d[u >> 1] << 16));
  }
  float* dst = out + size_t(m) * C + 16 * b + 4 * h;
  *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f32x4*>(dst + 8) = f32x4{v[4], v[5], v[6], v[7]};
}
// separable guess tables of the LDS gather for a per-head constant guess (gx, gy): tab_y[i][hd*8 + 2p (+1)] = g (the kernel
// averages the four points of tab_y + tab_x), tab_x = 0
__global__ void k_fill_guess_tables(const float* __restrict__ guess /* (8,2) or nullptr */, float* __restrict__ tab_y,
                                    float* __restrict__ tab_x, int h, int w) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx < h * 96) {
    const int c = idx % 96;
    tab_y[idx] = (guess && c < 64) ? guess[(c >> 3) * 2 + (c & 1)] : 0.f;
  }
  if (idx < w * 96) tab_x[idx] = 0.f;
}

// ------------------------------------------------------------------------------------------------
// Post-loop epilogue of the segmentor, fused: bilinear resize of the low-resolution class
// scores to the (padded) image size (segmentors/ddp.py:124-128), crop to img_shape + bilinear resize to ori_shape
// (encoder_decoder.py:236-248), softmax (:277, monotone: skipped), flip (:278-285), argmax (:296) -> uint8 map.
// The reference materialises (1,K,H,W) fp32 twice (160 MB per 1024x2048x19 image, 315 MB at 512x1024x150); here
// the scores are read once per tap from L2 and only the class map is written.
// Index / weight arithmetic follows at::native::compute_source_index_and_lambda (UpSample.h): identity when
// sizes match, src = scale*(dst+0.5)-0.5 clamped at 0 (align_corners=False) or scale*dst, scale in fp32.
// One thread per output pixel, classes in ascending order (first maximum wins, as torch.argmax on CPU).
// ------------------------------------------------------------------------------------------------
struct UpIdx {
  int i0, i1;
  float l0, l1;
};
__device__ __forceinline__ UpIdx up_index(int dst, int in_size, int out_size, int align) {
  UpIdx u;
  if (in_size == out_size) {
    u.i0 = u.i1 = dst;
    u.l0 = 1.f;
    u.l1 = 0.f;
    return u;
  }
  float real;
  if (align) {
    const float scale = out_size > 1 ? float(in_size - 1) / float(out_size - 1) : 0.f;
    real = scale * float(dst);
  } else {
    const float scale = float(in_size) / float(out_size);
    real = fmaxf(__fsub_rn(__fmul_rn(scale, float(dst) + 0.5f), 0.5f), 0.f);
  }
  u.i0 = min(int(floorf(real)), in_size - 1);
  u.l1 = fminf(fmaxf(real - float(u.i0), 0.f), 1.f);
  u.i1 = u.i0 + (u.i0 < in_size - 1 ? 1 : 0);
  u.l0 = 1.f - u.l1;
  return u;
}
// (v00*wx0 + v01*wx1)*wy0 + (v10*wx0 + v11*wx1)*wy1, no contraction (the CPU kernel's nesting)
__device__ __forceinline__ float bilerp(float v00, float v01, float v10, float v11, const UpIdx& y, const UpIdx& x) {
  const float top = __fadd_rn(__fmul_rn(v00, x.l0), __fmul_rn(v01, x.l1));
  const float bot = __fadd_rn(__fmul_rn(v10, x.l0), __fmul_rn(v11, x.l1));
  return __fadd_rn(__fmul_rn(top, y.l0), __fmul_rn(bot, y.l1));
}
__global__ void __launch_bounds__(256) k_seg_postprocess(SegPostArgs a) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.z;
  if (x >= a.ow || y >= a.oh) return;
  // flip acts on the final probabilities: out[y][x] = p[y][ow-1-x] (horizontal) / p[oh-1-y][x] (vertical)
  const int sx = a.flip == 1 ? a.ow - 1 - x : x;
  const int sy = a.flip == 2 ? a.oh - 1 - y : y;
  const bool two_stage = !(a.oh == a.ch && a.ow == a.cw);
  // stage 2 (ori <- crop of the H x W image); identity when sizes match
  const UpIdx Y = up_index(sy, a.ch, a.oh, a.align);
  const UpIdx X = up_index(sx, a.cw, a.ow, a.align);
  // stage 1 (H x W <- h x w) at the (up to) 2 x 2 stage-2 taps
  const UpIdx ya = up_index(Y.i0, a.h, a.H, a.align), yb = up_index(Y.i1, a.h, a.H, a.align);
  const UpIdx xa = up_index(X.i0, a.w, a.W, a.align), xb = up_index(X.i1, a.w, a.W, a.align);
  const float* plane = a.logits + size_t(b) * a.K * a.h * a.w;
  const size_t ps = size_t(a.h) * a.w;
  float best = -INFINITY;
  int arg = 0;
  for (int c = 0; c < a.K; ++c, plane += ps) {
    const float* r0 = plane + size_t(ya.i0) * a.w;
    const float* r1 = plane + size_t(ya.i1) * a.w;
    float v = bilerp(r0[xa.i0], r0[xa.i1], r1[xa.i0], r1[xa.i1], ya, xa);
    if (two_stage) {
      const float* q0 = plane + size_t(yb.i0) * a.w;
      const float* q1 = plane + size_t(yb.i1) * a.w;
      const float v01 = bilerp(r0[xb.i0], r0[xb.i1], r1[xb.i0], r1[xb.i1], ya, xb);
      const float v10 = bilerp(q0[xa.i0], q0[xa.i1], q1[xa.i0], q1[xa.i1], yb, xa);
      const float v11 = bilerp(q0[xb.i0], q0[xb.i1], q1[xb.i0], q1[xb.i1], yb, xb);
      v = bilerp(v, v01, v10, v11, Y, X);
    }
    if (v > best) {
      best = v;
      arg = c;
    }
  }
  a.seg[(size_t(b) * a.oh + y) * a.ow + x] = (unsigned char)arg;
}

// The common case (network input = 4 x map, no crop / second resize, align_corners=False): one thread per map cell
// produces its 4 x 4 output pixels from the cell's 3 x 3 neighbourhood - 9 loads per class instead of 64 - with the
// interpolation done separably (4 horizontal lerps per neighbourhood row, then 16 vertical ones): the same
// nesting, hence the same roundings, as the generic kernel and the reference's CPU kernel.
__global__ void __launch_bounds__(256) k_seg_postprocess_x4(SegPostArgs a) {
  const int j = blockIdx.x * 64 + (threadIdx.x & 63);
  const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.z;
  if (j >= a.w || i >= a.h) return;
  UpIdx Y[4], X[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    Y[r] = up_index(4 * i + r, a.h, a.H, 0);
    X[r] = up_index(4 * j + r, a.w, a.W, 0);
  }
  // rows / columns of the neighbourhood: outputs 0,1 read (r0, r1) = (i-1, i), outputs 2,3 read (i, i+1) = (r1, r2);
  // at the top / left border the clamped source makes outputs 2,3 read (r0, r1) instead -> low tap chosen per thread
  const int r0 = Y[0].i0, r1 = Y[0].i1, r2 = Y[3].i1;
  const int c0 = X[0].i0, c1 = X[0].i1, c2 = X[3].i1;
  const bool row_in = Y[2].i0 == r1, col_in = X[2].i0 == c1;
  const float* plane = a.logits + size_t(b) * a.K * a.h * a.w;
  const size_t ps = size_t(a.h) * a.w;
  float best[16];
  int arg[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    best[q] = -INFINITY;
    arg[q] = 0;
  }
  for (int c = 0; c < a.K; ++c, plane += ps) {
    float v[3][3];
    const float* p0 = plane + size_t(r0) * a.w;
    const float* p1 = plane + size_t(r1) * a.w;
    const float* p2 = plane + size_t(r2) * a.w;
    v[0][0] = p0[c0]; v[0][1] = p0[c1]; v[0][2] = p0[c2];
    v[1][0] = p1[c0]; v[1][1] = p1[c1]; v[1][2] = p1[c2];
    v[2][0] = p2[c0]; v[2][1] = p2[c1]; v[2][2] = p2[c2];
    float hz[3][4];       // horizontally interpolated, per neighbourhood row
#pragma unroll
    for (int rr = 0; rr < 3; ++rr)
#pragma unroll
      for (int xq = 0; xq < 4; ++xq) {
        const float lo = xq < 2 ? v[rr][0] : (col_in ? v[rr][1] : v[rr][0]);
        const float hi = xq < 2 ? v[rr][1] : (col_in ? v[rr][2] : v[rr][1]);
        hz[rr][xq] = __fadd_rn(__fmul_rn(lo, X[xq].l0), __fmul_rn(hi, X[xq].l1));
      }
#pragma unroll
    for (int yq = 0; yq < 4; ++yq)
#pragma unroll
      for (int xq = 0; xq < 4; ++xq) {
        const float top = yq < 2 ? hz[0][xq] : (row_in ? hz[1][xq] : hz[0][xq]);
        const float bot = yq < 2 ? hz[1][xq] : (row_in ? hz[2][xq] : hz[1][xq]);
        const float val = __fadd_rn(__fmul_rn(top, Y[yq].l0), __fmul_rn(bot, Y[yq].l1));
        if (val > best[yq * 4 + xq]) {
          best[yq * 4 + xq] = val;
          arg[yq * 4 + xq] = c;
        }
      }
  }
#pragma unroll
  for (int yq = 0; yq < 4; ++yq) {
    const int oy = a.flip == 2 ? a.oh - 1 - (4 * i + yq) : 4 * i + yq;
    unsigned packed;
    if (a.flip == 1) {
      packed = unsigned(arg[yq * 4 + 3]) | (unsigned(arg[yq * 4 + 2]) << 8) | (unsigned(arg[yq * 4 + 1]) << 16) | (unsigned(arg[yq * 4]) << 24);
      *reinterpret_cast<unsigned*>(a.seg + (size_t(b) * a.oh + oy) * a.ow + (a.ow - 4 - 4 * j)) = packed;
    } else {
      packed = unsigned(arg[yq * 4]) | (unsigned(arg[yq * 4 + 1]) << 8) | (unsigned(arg[yq * 4 + 2]) << 16) | (unsigned(arg[yq * 4 + 3]) << 24);
      *reinterpret_cast<unsigned*>(a.seg + (size_t(b) * a.oh + oy) * a.ow + 4 * j) = packed;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Test-time-augmentation epilogue (encoder_decoder.py:306-331 aug_test), fused.  A block = 64 output pixels of one row x 4
// waves; every wave walks the SAME 64 pixels and a quarter of the classes (the class-independent index arithmetic is repeated
// per wave: ~7 % of the work at 150 classes).  Per pixel and augmentation: the two-stage bilinear resize of k_seg_postprocess
// for the wave's classes (values parked in LDS), softmax over ALL classes (per-wave maximum / sum combined through LDS in a
// fixed order), accumulation into the pixel's running sum (LDS), flip undone by reading the mirrored source pixel; after the
// last augmentation: / n_aug, argmax (first maximum wins), optional store of the mean probabilities.
// LDS: 2 x K x 64 floats (tmp | acc, column = thread -> conflict-free) + 2 x 4 x 64 for the reductions.  (Until round 4 a block
// was ONE wave walking all classes: 77 KB of LDS per wave at 150 classes = 2 waves per CU, latency bound - 5.5 ms for the ADE
// multi-scale + flip case of scripts/epilogue_times.py.)
// ------------------------------------------------------------------------------------------------
struct SegAugArgs {
  ddp_seg_aug aug[DDP_MAX_AUGS];
  int n_aug, B, K, oh, ow, align;
  unsigned char* seg;
  float* prob;      // optional (B,K,oh,ow)
};
__global__ void __launch_bounds__(256) k_seg_aug_postprocess(SegAugArgs a) {
  extern __shared__ float aug_lds[];
  float* tmp = aug_lds;
  float* acc = aug_lds + size_t(a.K) * 64;
  float* red = aug_lds + size_t(a.K) * 128;                 // [2][4][64]: per-wave maxima | per-wave sums (then: best value | class)
  const int tid = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int x = blockIdx.x * 64 + tid;
  const int y = blockIdx.y;
  const int b = blockIdx.z;
  const bool live = x < a.ow;
  const int xs = live ? x : a.ow - 1;                        // (threads past the row's end compute a copy: they take part in the barriers)
  const int kq = (a.K + 3) >> 2;
  const int c0 = wv * kq, c1 = min(a.K, c0 + kq);            // this wave's classes
  for (int i = 0; i < a.n_aug; ++i) {
    const ddp_seg_aug& g = a.aug[i];
    const int sx = g.flip == 1 ? a.ow - 1 - xs : xs;
    const int sy = g.flip == 2 ? a.oh - 1 - y : y;
    const bool two_stage = !(a.oh == g.crop_h && a.ow == g.crop_w);
    const UpIdx Y = up_index(sy, g.crop_h, a.oh, a.align);
    const UpIdx X = up_index(sx, g.crop_w, a.ow, a.align);
    const UpIdx ya = up_index(Y.i0, g.h, g.img_h, a.align), yb = up_index(Y.i1, g.h, g.img_h, a.align);
    const UpIdx xa = up_index(X.i0, g.w, g.img_w, a.align), xb = up_index(X.i1, g.w, g.img_w, a.align);
    const size_t ps = size_t(g.h) * g.w;
    const float* plane = g.d_scores + (size_t(b) * a.K + c0) * ps;
    float mx = -INFINITY;
    for (int c = c0; c < c1; ++c, plane += ps) {
      const float* r0 = plane + size_t(ya.i0) * g.w;
      const float* r1 = plane + size_t(ya.i1) * g.w;
      float v = bilerp(r0[xa.i0], r0[xa.i1], r1[xa.i0], r1[xa.i1], ya, xa);
      if (two_stage) {
        const float* q0 = plane + size_t(yb.i0) * g.w;
        const float* q1 = plane + size_t(yb.i1) * g.w;
        const float v01 = bilerp(r0[xb.i0], r0[xb.i1], r1[xb.i0], r1[xb.i1], ya, xb);
        const float v10 = bilerp(q0[xa.i0], q0[xa.i1], q1[xa.i0], q1[xa.i1], yb, xa);
        const float v11 = bilerp(q0[xb.i0], q0[xb.i1], q1[xb.i0], q1[xb.i1], yb, xb);
        v = bilerp(v, v01, v10, v11, Y, X);
      }
      tmp[c * 64 + tid] = v;
      mx = fmaxf(mx, v);
    }
    red[wv * 64 + tid] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[tid], red[64 + tid]), fmaxf(red[128 + tid], red[192 + tid]));
    float sum = 0.f;
    for (int c = c0; c < c1; ++c) {
      const float e = expf(tmp[c * 64 + tid] - mx);
      tmp[c * 64 + tid] = e;
      sum += e;
    }
    red[256 + wv * 64 + tid] = sum;
    __syncthreads();                                         // (the next write of red[0..255] is behind this barrier: all maxima are read)
    sum = ((red[256 + tid] + red[320 + tid]) + red[384 + tid]) + red[448 + tid];
    for (int c = c0; c < c1; ++c) {
      const float p = tmp[c * 64 + tid] / sum;
      acc[c * 64 + tid] = i == 0 ? p : acc[c * 64 + tid] + p;
    }
    // (the next write of red[256..511] is behind the NEXT augmentation's first barrier: all sums are read by then)
  }
  const float nf = float(a.n_aug);
  float best = -INFINITY;
  int arg = 0x7fffffff;
  float* pout = (a.prob && live) ? a.prob + (size_t(b) * a.K * a.oh + y) * a.ow + x : nullptr;
  for (int c = c0; c < c1; ++c) {
    const float p = acc[c * 64 + tid] / nf;
    if (pout) pout[size_t(c) * a.oh * a.ow] = p;
    if (p > best) {
      best = p;
      arg = c;
    }
  }
  __syncthreads();                                           // the sums of the last augmentation are read: red is free
  red[wv * 64 + tid] = best;
  red[256 + wv * 64 + tid] = __int_as_float(arg);
  __syncthreads();
  if (wv == 0 && live) {
    // first maximum wins: the waves hold ascending class ranges, a strict '>' keeps the lower class on ties
    for (int q = 1; q < 4; ++q) {
      const float ob = red[q * 64 + tid];
      if (ob > best) {
        best = ob;
        arg = __float_as_int(red[256 + q * 64 + tid]);
      }
    }
    a.seg[(size_t(b) * a.oh + y) * a.ow + x] = (unsigned char)(arg == 0x7fffffff ? 0 : arg);
  }
}

// ------------------------------------------------------------------------------------------------
// Sliding-window inference epilogue (encoder_decoder.py:180-227 slide_inference + :266-296 inference / simple_test), fused: see
// ddp_seg_slide_postprocess in the header.  One thread per output pixel.  Window coverage is separable (a grid of window rows
// x columns), so per stage-2 tap the thread keeps the <= 4 covering rows and <= 4 covering columns with their stage-1
// interpolation indices in registers and then walks the classes: value(tap, c) = sum over covering (row, col), row-major as
// the reference adds them, of bilerp(window scores) / count.  LDS (prob output only): K x 64 floats, column = thread.
// ------------------------------------------------------------------------------------------------
struct SlideArgs {
  const float* scores[DDP_MAX_WINDOWS];
  int y1[DDP_MAX_WINDOWS], x1[DDP_MAX_WINDOWS];
  int n_rows, n_cols, B, K, h, w, ch, cw, H, W, kh, kw, oh, ow, align, flip, prob_mode;
  unsigned char* seg;
  float* prob;
};
// (4, not 3: clamping the last window back into the image puts a fourth window over pixels that stride >= crop / 3 alone
// would cover three times - crop 9, stride 3, H = 16 gives origins 0, 3, 6, 7 - and the reference accepts such grids)
constexpr int SLIDE_MAX_COVER = 4;
struct SlideCover {
  int n;
  int idx[SLIDE_MAX_COVER];
  UpIdx u[SLIDE_MAX_COVER];
};
__device__ __forceinline__ SlideCover slide_cover(int p, const int* start, int n_win, int crop, int lowres, int align) {
  SlideCover c;
  c.n = 0;
#pragma unroll 1
  for (int i = 0; i < n_win; ++i)
    if (p >= start[i] && p < start[i] + crop && c.n < SLIDE_MAX_COVER) {
      c.idx[c.n] = i;
      c.u[c.n] = up_index(p - start[i], lowres, crop, align);
      ++c.n;
    }
  return c;
}
__global__ void __launch_bounds__(64) k_seg_slide_postprocess(SlideArgs a) {
  extern __shared__ float slide_lds[];
  const int tid = threadIdx.x;
  const int x = blockIdx.x * 64 + tid;
  const int y = blockIdx.y;
  const int b = blockIdx.z;
  if (x >= a.ow) return;
  const int sx = a.flip == 1 ? a.ow - 1 - x : x;
  const int sy = a.flip == 2 ? a.oh - 1 - y : y;
  const bool two_stage = !(a.oh == a.kh && a.ow == a.kw);
  const UpIdx Y = up_index(sy, a.kh, a.oh, a.align);
  const UpIdx X = up_index(sx, a.kw, a.ow, a.align);
  // stage-2 taps are image pixels (the crop to img_shape starts at the origin)
  SlideCover cy[2], cx[2];
  cy[0] = slide_cover(Y.i0, a.y1, a.n_rows, a.ch, a.h, a.align);
  cx[0] = slide_cover(X.i0, a.x1, a.n_cols, a.cw, a.w, a.align);
  cy[1] = two_stage ? slide_cover(Y.i1, a.y1, a.n_rows, a.ch, a.h, a.align) : cy[0];
  cx[1] = two_stage ? slide_cover(X.i1, a.x1, a.n_cols, a.cw, a.w, a.align) : cx[0];
  const size_t ps = size_t(a.h) * a.w;
  const size_t boff = size_t(b) * a.K * ps;
  auto tap = [&](const SlideCover& ry, const SlideCover& rx, int c) {
    float s = 0.f;
    for (int i = 0; i < ry.n; ++i)
      for (int jx = 0; jx < rx.n; ++jx) {
        const float* plane = a.scores[ry.idx[i] * a.n_cols + rx.idx[jx]] + boff + size_t(c) * ps;
        const float* r0 = plane + size_t(ry.u[i].i0) * a.w;
        const float* r1 = plane + size_t(ry.u[i].i1) * a.w;
        const float v = bilerp(r0[rx.u[jx].i0], r0[rx.u[jx].i1], r1[rx.u[jx].i0], r1[rx.u[jx].i1], ry.u[i], rx.u[jx]);
        s = (i == 0 && jx == 0) ? v : __fadd_rn(s, v);      // preds starts at zero: 0 + v == v
      }
    return s / float(ry.n * rx.n);                          // preds / count_mat
  };
  float best = -INFINITY, mx = -INFINITY;
  int arg = 0;
  for (int c = 0; c < a.K; ++c) {
    float v = tap(cy[0], cx[0], c);
    if (two_stage) v = bilerp(v, tap(cy[0], cx[1], c), tap(cy[1], cx[0], c), tap(cy[1], cx[1], c), Y, X);
    if (a.prob_mode) slide_lds[c * 64 + tid] = v;
    mx = fmaxf(mx, v);
    if (v > best) {
      best = v;
      arg = c;
    }
  }
  if (a.seg) a.seg[(size_t(b) * a.oh + y) * a.ow + x] = (unsigned char)arg;
  if (a.prob_mode) {
    float* pout = a.prob + (size_t(b) * a.K * a.oh + y) * a.ow + x;
    const size_t cs = size_t(a.oh) * a.ow;
    if (a.prob_mode == 2) {
      for (int c = 0; c < a.K; ++c) pout[c * cs] = slide_lds[c * 64 + tid];
    } else {
      float sum = 0.f;
      for (int c = 0; c < a.K; ++c) {
        const float e = expf(slide_lds[c * 64 + tid] - mx);
        slide_lds[c * 64 + tid] = e;
        sum += e;
      }
      for (int c = 0; c < a.K; ++c) pout[c * cs] = slide_lds[c * 64 + tid] / sum;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Post-loop epilogue of the depth toolbox, fused (depth/depth/models/depther/ddp.py:95-109 encode_decode: clamp + resize;
// encoder_decoder.py:187-194 inference: flip; :210-229 aug_test: running sum in list order, / n).  HBM-bound: the low-resolution
// maps (<= 107 KB per image at KITTI size) stay in L2, the only real traffic is the (B,1,H,W) store, so a thread owns 4
// x-adjacent output pixels and writes them as one 16-B store when the row allows it.  torch.clamp keeps NaN (comparisons,
// not fmin / fmax); the clamp comes BEFORE the interpolation, as in the reference.
// ------------------------------------------------------------------------------------------------
struct DepthAugArgs {
  ddp_depth_aug aug[DDP_MAX_AUGS];
  int n_aug, B, oh, ow, align;
  float lo, hi;
  float* out;
};
__device__ __forceinline__ float clamp_keep_nan(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
__global__ void __launch_bounds__(256) k_depth_aug_postprocess(DepthAugArgs a) {
  const int x0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.z;
  if (x0 >= a.ow || y >= a.oh) return;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < a.n_aug; ++i) {
    const ddp_depth_aug& g = a.aug[i];
    const float* plane = g.d_depth + size_t(b) * g.h * g.w;
    const int sy = g.flip == 2 ? a.oh - 1 - y : y;
    const UpIdx Y = up_index(sy, g.h, a.oh, a.align);
    const float* r0 = plane + size_t(Y.i0) * g.w;
    const float* r1 = plane + size_t(Y.i1) * g.w;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int x = min(x0 + q, a.ow - 1);
      const int sx = g.flip == 1 ? a.ow - 1 - x : x;
      const UpIdx X = up_index(sx, g.w, a.ow, a.align);
      const float v = bilerp(clamp_keep_nan(r0[X.i0], a.lo, a.hi), clamp_keep_nan(r0[X.i1], a.lo, a.hi),
                             clamp_keep_nan(r1[X.i0], a.lo, a.hi), clamp_keep_nan(r1[X.i1], a.lo, a.hi), Y, X);
      acc[q] = i == 0 ? v : __fadd_rn(acc[q], v);
    }
  }
  if (a.n_aug > 1) {
    const float nf = float(a.n_aug);
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = acc[q] / nf;
  }
  float* o = a.out + (size_t(b) * a.oh + y) * a.ow + x0;
  if ((a.ow & 3) == 0 && (reinterpret_cast<size_t>(a.out) & 15) == 0) {       // (a caller's view may start at any float)
    *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (x0 + q < a.ow) o[q] = acc[q];
  }
}

// ------------------------------------------------------------------------------------------------
// GroupNorm of the necks: deterministic two-stage statistics (fp64, fixed order) + normalise.  The stream GEMM
// writes the partial sums itself when it can (k_gn_final32); these kernels cover the other cases and the merged map.
// ------------------------------------------------------------------------------------------------
// GroupNorm(32 groups of 8 channels) statistics over a token-major (B, N, 256) tensor, stage 1: block (chunk, b)
// sums its 256-token chunk; thread = (token row 0..3, channel quad); partial[b][chunk][g] = {sum, sum of squares}
template <bool BLK>
__global__ void __launch_bounds__(256) k_gn_partial(const float* __restrict__ y, double* __restrict__ partial, int N,
                                                     int chunks) {
  __shared__ double red[4][32][2];
  const int lane = threadIdx.x & 63, row = threadIdx.x >> 6;
  const int b = blockIdx.y, ck = blockIdx.x;
  const int n0 = ck * 256, n1 = min(n0 + 256, N);
  double s = 0.0, q = 0.0;
  for (int n = n0 + row; n < n1; n += 4) {
    const size_t mrow = size_t(b) * N + n;
    const f32x4 v = *reinterpret_cast<const f32x4*>(BLK ? y + blk_off256(int(mrow), lane * 4) : y + mrow * 256 + lane * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s += double(v[e]);
      q += double(v[e]) * double(v[e]);
    }
  }
  // a group = 8 channels = the lane pair (2g, 2g+1)
  s += __shfl_xor(s, 1, 64);
  q += __shfl_xor(q, 1, 64);
  if ((lane & 1) == 0) {
    red[row][lane >> 1][0] = s;
    red[row][lane >> 1][1] = q;
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int g = threadIdx.x >> 1, k = threadIdx.x & 1;
    const double t = (red[0][g][k] + red[1][g][k]) + (red[2][g][k] + red[3][g][k]);
    partial[((size_t(b) * chunks + ck) * 32 + g) * 2 + k] = t;
  }
}
// stage 2: fixed-order sum over the chunks -> {mean, rstd} per (b, g)
__global__ void k_gn_final(const double* __restrict__ partial, float* __restrict__ stats, int N, int chunks, float eps) {
  const int b = blockIdx.x, g = threadIdx.x;      // 32 threads
  double s = 0.0, q = 0.0;
  for (int ck = 0; ck < chunks; ++ck) {
    s += partial[((size_t(b) * chunks + ck) * 32 + g) * 2];
    q += partial[((size_t(b) * chunks + ck) * 32 + g) * 2 + 1];
  }
  const double cnt = double(N) * 8.0;
  const double mean = s / cnt;
  const double var = fmax(q / cnt - mean * mean, 0.0);      // biased variance, as torch.nn.GroupNorm
  stats[(b * 32 + g) * 2] = float(mean);
  stats[(b * 32 + g) * 2 + 1] = float(1.0 / sqrt(var + double(eps)));
}
// the same from partial sums over chunks of 32 tokens (written by the stream GEMM's epilogue): block = image, thread =
// (group, one of 32 strided parts), fixed-order sums -> deterministic
__global__ void __launch_bounds__(1024) k_gn_final32(const double* __restrict__ partial, float* __restrict__ stats, int N, int chunks,
                                                      float eps) {
  __shared__ double red[32][32][2];
  const int b = blockIdx.x, g = threadIdx.x & 31, part = threadIdx.x >> 5;
  double s = 0.0, q = 0.0;
  for (int ck = part; ck < chunks; ck += 32) {
    const double* p = partial + ((size_t(b) * chunks + ck) * 32 + g) * 2;
    s += p[0];
    q += p[1];
  }
  red[part][g][0] = s;
  red[part][g][1] = q;
  __syncthreads();
  if (part == 0) {
    double ts = 0.0, tq = 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      ts += red[k][g][0];
      tq += red[k][g][1];
    }
    const double cnt = double(N) * 8.0;
    const double mean = ts / cnt;
    const double var = fmax(tq / cnt - mean * mean, 0.0);
    stats[(b * 32 + g) * 2] = float(mean);
    stats[(b * 32 + g) * 2 + 1] = float(1.0 / sqrt(var + double(eps)));
  }
}
// normalise + per-channel affine, token-major (B,N,256) -> NCHW (B,256,N), 64x64 tiles through LDS
template <bool BLK>
__global__ void __launch_bounds__(256) k_gn_apply_nchw(const float* __restrict__ y, const float* __restrict__ stats,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ out, int N) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z;
  const int n0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  if constexpr (BLK) {
    // fragment-major y: thread = (token, channel quad): 16-B reads, 32 consecutive tokens of a quad are 512 B contiguous
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int nn = tx, cq = it * 4 + ty;             // 64 tokens x 16 quads
      const int n = n0 + nn, c = c0 + cq * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (n < N) {
        v = *reinterpret_cast<const f32x4*>(y + blk_off256(int(size_t(b) * N + n), c));
        const float mean = stats[(b * 32 + (c >> 3)) * 2], rstd = stats[(b * 32 + (c >> 3)) * 2 + 1];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (v[e] - mean) * rstd * gamma[c + e] + beta[c + e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) tile[nn][cq * 4 + e] = v[e];
    }
  } else {
    const int c = c0 + tx;
    const float mean = stats[(b * 32 + (c >> 3)) * 2], rstd = stats[(b * 32 + (c >> 3)) * 2 + 1];
    const float ga = gamma[c], be = beta[c];
    for (int nn = ty; nn < 64; nn += 4) {
      const int n = n0 + nn;
      float v = 0.f;
      if (n < N) v = (y[(size_t(b) * N + n) * 256 + c] - mean) * rstd * ga + be;
      tile[nn][tx] = v;
    }
  }
  __syncthreads();
  for (int cc = ty; cc < 64; cc += 4) {
    const int n = n0 + tx;
    if (n < N) out[(size_t(b) * 256 + c0 + cc) * N + n] = tile[tx][cc];
  }
}

// ------------------------------------------------------------------------------------------------
// Neck kernels on fp32 FRAGMENT-MAJOR activations ("blk": per 32-token group [channel tile t][quad g][half][token][4], the
// layout the stream GEMM (k_layer MODE 5) reads its A operand from and writes its result in; C channels: group stride 32 C
// floats).  A "piece" p = 4 consecutive channels 4p .. 4p+3 of one token = one 16-B slot; the 32 tokens of a group hold a
// piece contiguously (512 B), so thread = (group, piece, token) makes every access coalesced.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ size_t blk_piece_off(int m, int p, int C) {        // floats; p = channel / 4
  return size_t(m >> 5) * 32 * C + size_t(p) * 128 + (m & 31) * 4;
}
// NCHW (R, C, N) -> blk (rows R*N, C channels, C % 32 == 0): 64 tokens x 64 channels per block through LDS
__global__ void __launch_bounds__(256) k_nchw_to_blk(const float* __restrict__ in, float* __restrict__ out, int C, int N, int rows) {
  __shared__ float tile[64][65];
  const int m0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  {
    const int m = m0 + tx;
    const int r = m / N, n = m - r * N;
    const float* src = in + (size_t(r) * C + c0) * N + n;
    for (int cc = ty; cc < 64; cc += 4) tile[cc][tx] = (m < rows && c0 + cc < C) ? src[size_t(cc) * N] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int nn = tx, cq = it * 4 + ty;                 // 64 tokens x 16 pieces
    const int c = c0 + cq * 4;
    if (c >= C) continue;
    const f32x4 v = {tile[cq * 4][nn], tile[cq * 4 + 1][nn], tile[cq * 4 + 2][nn], tile[cq * 4 + 3][nn]};
    *reinterpret_cast<f32x4*>(out + blk_piece_off(m0 + nn, c >> 2, C)) = v;      // (rows up to the next multiple of 64: zeros)
  }
}
// FPN top-down step on blk maps (necks/fpn.py:167-185): lat = GroupNorm(y) [+ nearest-upsampled coarser lateral]
//   out[m][c] = (y[m][c] - mean) * rstd * gamma[c] + beta[c] + coarse[nearest(m)][c]
__global__ void __launch_bounds__(256) k_gn_apply_add_blk(const float* __restrict__ y, const float* __restrict__ stats,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ coarse, float* __restrict__ out, int hf, int wf,
                                                           int hc, int wc, int rows) {
  const long idx = long(blockIdx.x) * 256 + threadIdx.x;        // (group, piece, token)
  const int ml = int(idx & 31), p = int((idx >> 5) & 63), grp = int(idx >> 11);
  const int m = grp * 32 + ml;
  if (m >= rows) return;
  const int Nf = hf * wf;
  const int b = m / Nf, n = m - b * Nf;
  const size_t off = size_t(grp) * 8192 + p * 128 + ml * 4;
  const f32x4 v = *reinterpret_cast<const f32x4*>(y + off);
  const float mean = stats[(b * 32 + (p >> 1)) * 2], rstd = stats[(b * 32 + (p >> 1)) * 2 + 1];
  const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + p * 4);
  const f32x4 be = *reinterpret_cast<const f32x4*>(beta + p * 4);
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = (v[e] - mean) * rstd * ga[e] + be[e];
  if (coarse) {
    // F.interpolate(mode='nearest'): src = min(floor(dst * scale), in - 1), scale = in / out in fp32
    const int i = n / wf, j = n - i * wf;
    const float sy = float(hc) / float(hf), sx = float(wc) / float(wf);
    const int ic = min(int(floorf(float(i) * sy)), hc - 1), jc = min(int(floorf(float(j) * sx)), wc - 1);
    const int mc = b * hc * wc + ic * wc + jc;
    o = o + *reinterpret_cast<const f32x4*>(coarse + blk_piece_off(mc, p, 256));
  }
  *reinterpret_cast<f32x4*>(out + off) = o;
}
// MultiStageMerging by linearity: the 1x1 conv over the concatenated, resized levels (necks/multi_stage_merging.py:40-52)
// = sum over the levels of the bilinear resize of (that level's 256 x 256 block of the conv applied at the level's own
// resolution): y0 += sum_l resize(y_l), all blk.  (The reference resizes first; both orders are linear maps with the same
// weights, the results differ by fp32 rounding only.)
struct MsmSumArgs {
  float* y0;                // (rows, 256) blk: level 0's conv output in, the merged map out
  const float* yl[3];       // levels 1..3 conv outputs at their own resolution, blk
  int lh[3], lw[3];
  int h, w, rows, align;
  double* gn_partial;       // optional (h * w % 32 == 0): GroupNorm partial sums of the merged map, [image][token / 32][group] {sum, sum of
                            // squares} - the format k_gn_final32 reduces; a wave = 32 tokens x the 8 channels of ONE group
};
__global__ void __launch_bounds__(256) k_msm_sum_blk(MsmSumArgs a) {
  const long idx = long(blockIdx.x) * 256 + threadIdx.x;
  const int ml = int(idx & 31), p = int((idx >> 5) & 63), grp = int(idx >> 11);
  const int m = grp * 32 + ml;
  if (m >= a.rows) return;
  const int N = a.h * a.w;
  const int b = m / N, n = m - b * N;
  const int i = n / a.w, j = n - i * a.w;
  const size_t off = size_t(grp) * 8192 + p * 128 + ml * 4;
  f32x4 acc = *reinterpret_cast<const f32x4*>(a.y0 + off);
#pragma unroll
  for (int l = 0; l < 3; ++l) {
    const int hl = a.lh[l], wl = a.lw[l];
    const UpIdx y = up_index(i, hl, a.h, a.align), x = up_index(j, wl, a.w, a.align);
    const int mb = b * hl * wl;
    const f32x4 v00 = *reinterpret_cast<const f32x4*>(a.yl[l] + blk_piece_off(mb + y.i0 * wl + x.i0, p, 256));
    const f32x4 v01 = *reinterpret_cast<const f32x4*>(a.yl[l] + blk_piece_off(mb + y.i0 * wl + x.i1, p, 256));
    const f32x4 v10 = *reinterpret_cast<const f32x4*>(a.yl[l] + blk_piece_off(mb + y.i1 * wl + x.i0, p, 256));
    const f32x4 v11 = *reinterpret_cast<const f32x4*>(a.yl[l] + blk_piece_off(mb + y.i1 * wl + x.i1, p, 256));
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] += bilerp(v00[e], v01[e], v10[e], v11[e], y, x);
  }
  *reinterpret_cast<f32x4*>(a.y0 + off) = acc;
  if (a.gn_partial) {
    // fused statistics (round 6: the separate pass over the merged map - k_gn_partial - is gone): this wave holds pieces 2k, 2k + 1
    // = channels 8k .. 8k + 7 = group k of its 32 tokens (rows % 32 == 0: no lane returned above); fp64, fixed butterfly order
    double sv = 0.0, qv = 0.0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sv += double(acc[e]);
      qv += double(acc[e]) * double(acc[e]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      sv += __shfl_xor(sv, o, 64);
      qv += __shfl_xor(qv, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
      double* dst = a.gn_partial + ((size_t(b) * (N >> 5) + (n >> 5)) * 32 + (p >> 1)) * 2;
      dst[0] = sv;
      dst[1] = qv;
    }
  }
}

// k_gn_final32 for up to four maps in ONE launch (grid.y = map): the FPN finalises the statistics of its four levels together
struct GnFinalMulti {
  const double* partial[4];
  float* stats[4];
  int N[4];
};
__global__ void __launch_bounds__(1024) k_gn_final32_multi(GnFinalMulti a, float eps) {
  __shared__ double red[32][32][2];
  const int lv = blockIdx.y;
  const double* partial = a.partial[lv];
  float* stats = a.stats[lv];
  const int N = a.N[lv], chunks = N / 32;
  const int b = blockIdx.x, g = threadIdx.x & 31, part = threadIdx.x >> 5;
  double s = 0.0, q = 0.0;
  for (int ck = part; ck < chunks; ck += 32) {
    const double* p = partial + ((size_t(b) * chunks + ck) * 32 + g) * 2;
    s += p[0];
    q += p[1];
  }
  red[part][g][0] = s;
  red[part][g][1] = q;
  __syncthreads();
  if (part == 0) {
    double ts = 0.0, tq = 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      ts += red[k][g][0];
      tq += red[k][g][1];
    }
    const double cnt = double(N) * 8.0;
    const double mean = ts / cnt;
    const double var = fmax(tq / cnt - mean * mean, 0.0);
    stats[(b * 32 + g) * 2] = float(mean);
    stats[(b * 32 + g) * 2 + 1] = float(1.0 / sqrt(var + double(eps)));
  }
}

// ------------------------------------------------------------------------------------------------
// FCNHeadWithTime (decode_heads/fcn_head_with_time.py:205-225,285-305): each ConvWithTimeModule is
// conv3x3 -> norm -> x*(scale+1)+shift (FiLM from the time embedding) -> ReLU.  In eval mode the norm and the FiLM are
// one per-channel affine: the scale goes into the (re-packed) weights, the shift into the accumulator bias, ReLU into
// the GEMM epilogue - the convolution itself is ONE bf16x3 implicit GEMM with K = 9*256 (b3::k_gemm<..., CONV>: every
// lane fetches the SB slot of its tap-shifted source token; no im2col buffer).
// ------------------------------------------------------------------------------------------------
// conv.weight (cout, cin, 3, 3) -> (cout, 9*cin) tap-major rows, each row scaled by scale[cout] (nullptr: 1)
__global__ void k_pack_conv3x3_scaled(const float* __restrict__ w, const float* __restrict__ scale, float* __restrict__ out,
                                      int cout, int cin) {
  const long i = long(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= long(cout) * 9 * cin) return;
  const int o = int(i / (9 * cin)), r = int(i - long(o) * 9 * cin);
  const int tap = r / cin, c = r - tap * cin;
  const float sc = scale ? scale[o] : 1.0f;
  out[i] = w[(size_t(o) * cin + c) * 9 + tap] * sc;
}
// per-channel affine of one ConvWithTimeModule in eval mode: y = conv * scale + shift with
//   n = gamma / sqrt(var + eps) (1 without a norm), f = film_scale + 1, scale = n * f,
//   shift = ((conv_bias - mean) * n + beta) * f + film_shift          (film == nullptr: f = 1, film_shift = 0)
__global__ void k_fcn_fold(const float* __restrict__ bn_w, const float* __restrict__ bn_b, const float* __restrict__ bn_mean,
                           const float* __restrict__ bn_var, float bn_eps, const float* __restrict__ conv_bias,
                           const float* __restrict__ film, float* __restrict__ scale, float* __restrict__ shift) {
  const int c = threadIdx.x;      // 256
  float n = 1.f, b = conv_bias ? conv_bias[c] : 0.f;
  if (bn_w) {
    n = bn_w[c] / sqrtf(bn_var[c] + bn_eps);
    b = (b - bn_mean[c]) * n + bn_b[c];
  }
  const float f = film ? film[c] + 1.0f : 1.0f, fs = film ? film[256 + c] : 0.f;
  scale[c] = n * f;
  shift[c] = b * f + fs;
}

// ------------------------------------------------------------------------------------------------
// time embedding pieces (segmentors/ddp.py:41-46,107-112; utils/transformer.py:275-278)
// ------------------------------------------------------------------------------------------------
__global__ void k_sinusoid(const float* __restrict__ freq, const float* __restrict__ t_in, int S, float* __restrict__ u) {
  const int s = blockIdx.x;
  const int k = threadIdx.x;  // 0..16
  if (s >= S || k >= DDP_SINU_FEATS) return;
  const float x = t_in[s];
  float v;
  if (k == 0) {
    v = x;
  } else {
    const int f = (k - 1) & 7;
    const float fr = ((x * freq[f]) * 2.0f) * 3.14159265358979323846f;  // x * w * 2 * pi, left to right in fp32
    v = (k <= 8) ? sinf(fr) : cosf(fr);
  }
  u[s * DDP_SINU_FEATS + k] = v;
}

// y[s][o] = out_act(W[o][:] . in_act(x[s][:]) + b[o]); one wave per (s, o)
__global__ void __launch_bounds__(256) k_matvec(const float* __restrict__ W, const float* __restrict__ b,
                                                 const float* __restrict__ x, float* __restrict__ y, int in_dim,
                                                 int out_dim, int S, int ldx, int ldy, int in_act, int out_act) {
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (gw >= out_dim * S) return;
  const int s = gw / out_dim, o = gw - s * out_dim;
  const float* wr = W + size_t(o) * in_dim;
  const float* xr = x + size_t(s) * ldx;
  float acc = 0.f;
  for (int k = lane; k < in_dim; k += 64) {
    float xv = xr[k];
    if (in_act == 2) xv = xv / (1.0f + expf(-xv));  // SiLU
    acc += wr[k] * xv;
  }
  acc = wave_sum(acc);
  if (lane == 0) {
    float v = acc + (b ? b[o] : 0.f);
    if (out_act == 1) v = v * 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
    y[size_t(s) * ldy + o] = v;
  }
}

// x0 look-up table: lut[k][c] = (sigmoid(E[k][c]) * 2 - 1) * bit_scale   (ddp.py:236-237)
__global__ void k_build_lut(const float* __restrict__ emb, float* __restrict__ lut, int n, float bit_scale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) lut[i] = (sigmoidf_(emb[i]) * 2.0f - 1.0f) * bit_scale;
}

// Separable positional tables (utils/transformer.py:78-113 folded through sampling_offsets /
// attention_weights):  PY[i][o] = b[o] + sum_{c<128} W[o][c] * pe_y(i,c);  PX[j][o] = sum_c W[o][128+c] * pe_x(j,c)
// pe(p, c) = sin|cos( ((p + 1 - 0.5) / (len + 1e-6) * 2pi) / 10000^(2*(c/2)/128) ), sin for even c.
// one wave per (row, o): rows 0..h-1 are PY, h..h+w-1 are PX.
__global__ void __launch_bounds__(256) k_pos_tables(const float* __restrict__ wcat, const float* __restrict__ bcat,
                                                     float* __restrict__ py, float* __restrict__ px, int h, int w) {
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (gw >= (h + w) * 96) return;
  const int row = gw / 96, o = gw - row * 96;
  const bool is_y = row < h;
  const int p = is_y ? row : row - h;
  const float len = float(is_y ? h : w);
  const float embed = (float(p + 1) + (-0.5f)) / (len + 1e-6f) * 6.283185307179586f;
  const float* wr = wcat + o * 256 + (is_y ? 0 : 128);
  float acc = 0.f;
#pragma unroll
  for (int c = lane; c < 128; c += 64) {
    const float dim_t = powf(10000.0f, float(2 * (c / 2)) / 128.0f);
    const float a = embed / dim_t;
    const float pe = (c & 1) ? cosf(a) : sinf(a);
    acc += wr[c] * pe;
  }
  acc = wave_sum(acc);
  if (lane == 0) {
    if (is_y) py[p * 96 + o] = acc + bcat[o];
    else px[p * 96 + o] = acc;
  }
}

__global__ void k_pack_rows(const float* __restrict__ a, int na, const float* __restrict__ b, int nb,
                            float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < na) out[i] = a[i];
  else if (i < na + nb) out[i] = b[i - na];
}

// out[r][c] = in[r*ld_in + off + c]  (split the concat-conv weight into its x / noisy-map column blocks)
__global__ void k_pack_cols(const float* __restrict__ in, int ld_in, int off, int rows, int cols, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  const int r = i / cols, c = i - r * cols;
  out[i] = in[size_t(r) * ld_in + off + c];
}

// conv_depth.weight (1,256,3,3) -> tap-major (9,256): out[dy*3+dx][c] = w[c][dy][dx]
__global__ void k_pack_conv3x3(const float* __restrict__ w, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 9 * 256) return;
  const int t = i / 256, c = i - t * 256;
  out[i] = w[c * 9 + t];
}

struct FloatList { float v[DDP_MAX_STEPS]; };
__global__ void k_write_floats(FloatList f, int n, float* __restrict__ out) {
  const int i = threadIdx.x;
  if (i < n) out[i] = f.v[i];
}

// ------------------------------------------------------------------------------------------------
// seg: argmax -> LUT -> DDIM / DDPM update of the noisy map, + softmax accumulation.
// One wave per token.  (segmentors/ddp.py:235-245 ddim, :276-287 ddpm)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_seg_update(SegUpdateArgs a) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= a.rows) return;
  const float* lg = a.logits + size_t(m) * a.ldl;
  const int K = a.num_classes;
  // up to 256 classes: 4 per lane (k = lane + 64*q)
  float v[4];
  float best = -INFINITY;
  int bi = 0x7fffffff;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int k = lane + 64 * q;
    v[q] = (k < K) ? lg[k] : -INFINITY;
    if (k < K && (v[q] > best)) {   // strict >: keeps the first (smallest k) maximum within the lane
      best = v[q];
      bi = k;
    }
  }
  // wave arg-max, ties -> smallest index (torch.argmax returns the first maximal element)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) {
      best = ob;
      bi = oi;
    }
  }
  // a row without a maximum (every score NaN or -inf: a NaN in x spreads over the row through LayerNorm) must still index
  // the LUT in range - the NaN then propagates through the update arithmetic instead of faulting (the tail kernel does the same)
  if (bi >= K) bi = 0;
  if (a.x0_idx && lane == 0) a.x0_idx[m] = (unsigned char)bi;
  if (a.x0_force) {               // teacher forcing (DDP_FLAG_FORCE_X0): the caller's class goes into the update, not the argmax
    bi = a.x0_force[m];
    if (bi >= K) bi = 0;
  }
  if (a.prob && a.prob_mode) {
    float* pr = a.prob + size_t(m) * a.ldl;
    if (a.prob_mode == 3) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = lane + 64 * q;
        if (k < K) pr[k] = v[q];
      }
    } else {
      float e[4], s = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = lane + 64 * q;
        e[q] = (k < K) ? expf(v[q] - best) : 0.f;
        s += e[q];
      }
      s = wave_sum(s);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = lane + 64 * q;
        if (k < K) {
          const float pv = e[q] / s;
          pr[k] = (a.prob_mode == 2) ? pr[k] + pv : pv;
        }
      }
    }
  }
  // x0 row and update of this token's 256 channels
  const f32x4 x0 = *reinterpret_cast<const f32x4*>(a.lut + size_t(bi) * 256 + lane * 4);
  float* mp = a.mask + size_t(m) * 256 + lane * 4;
  const f32x4 mt = *reinterpret_cast<const f32x4*>(mp);
  f32x4 mn;
  if (a.sampler == DDP_SAMPLER_DDIM) {
    const float sig = fmaxf(a.st.sigma, 1e-8f);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float pn = (mt[e] - a.st.alpha * x0[e]) / sig;
      mn[e] = x0[e] * a.st.alpha_next + pn * a.st.sigma_next;
    }
  } else {
    f32x4 nz = {0.f, 0.f, 0.f, 0.f};
    if (a.st.ddpm_add_noise && a.step_noise) nz = *reinterpret_cast<const f32x4*>(a.step_noise + size_t(m) * 256 + lane * 4);
    const float c = a.st.ddpm_c;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float mean = a.st.alpha_next * (mt[e] * (1.0f - c) / a.st.alpha + c * x0[e]);
      mn[e] = mean + a.st.ddpm_std * nz[e];
    }
  }
  *reinterpret_cast<f32x4*>(mp) = mn;
}

// x0 projection alone, NCHW in / out (segmentors/ddp.py:235-237; the self-aligned pre-pass of
// segmentors/self_aligned_ddp.py:160-164): idx = argmax_k scores[b][k][n] (first maximum), out[b][c][n] =
// (sigmoid(E[idx][c]) * 2 - 1) * bit_scale.  One thread per pixel: score reads and map writes are coalesced over n.
__global__ void __launch_bounds__(256) k_seg_x0_nchw(const float* __restrict__ scores, const float* __restrict__ emb,
                                                      float* __restrict__ out, int K, int N, float bit_scale) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (n >= N) return;
  const float* sp = scores + size_t(b) * K * N + n;
  float best = -INFINITY;
  int bi = 0;                                         // a row without a maximum (all NaN) keeps class 0: in range
  for (int k = 0; k < K; ++k) {
    const float v = sp[size_t(k) * N];
    if (v > best) {
      best = v;
      bi = k;
    }
  }
  const float* e = emb + size_t(bi) * 256;
  float* op = out + size_t(b) * 256 * N + n;
  for (int c = 0; c < 256; ++c) op[size_t(c) * N] = (sigmoidf_(e[c]) * 2.0f - 1.0f) * bit_scale;
}

// where class k of token m lives: token-major rows (m, ldl), or - nch > 0, what the layer kernel's seg tails write - fragment-major:
// per 32-token group [chunk k / 64][t][g][lane 64][4] = nch * 2048 floats, class 64c + 32t + 8g + 4h + e of token j at lane h * 32 + j
__device__ __forceinline__ size_t prob_off(size_t m, int k, int ldl, int nch) {
  if (nch == 0) return m * size_t(ldl) + k;
  return (m >> 5) * size_t(nch * 2048) + size_t(k >> 3) * 256 + (((k >> 2) & 1) * 32 + int(m & 31)) * 4 + (k & 3);
}
// out[b][k][n] = (sum_ri prob[(b*r+ri)*N + n][k]) / div ; block = 64 tokens, LDS transpose
__global__ void __launch_bounds__(256) k_finalize_nchw(const float* __restrict__ prob, int ldl, float* __restrict__ out,
                                                        int r, int N, int K, float div, int nch) {
  extern __shared__ float tile[];  // [64][K+1]
  const int b = blockIdx.y;
  const int n0 = blockIdx.x * 64;
  const int ldt = K + 1;
  for (int idx = threadIdx.x; idx < 64 * K; idx += 256) {
    const int nn = idx / K, k = idx - nn * K;
    const int n = n0 + nn;
    float s = 0.f;
    if (n < N)
      for (int ri = 0; ri < r; ++ri) s += prob[prob_off(size_t(b * r + ri) * N + n, k, ldl, nch)];
    tile[nn * ldt + k] = s / div;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int k = wv; k < K; k += 4) {
    const int n = n0 + lane;
    if (n < N) out[(size_t(b) * K + k) * N + n] = tile[lane * ldt + k];
  }
}

// The same reduction + transpose for maps whose token count is a multiple of 4 (every BASELINE configuration): a block moves a
// tile of 256 tokens x 32 classes.  Reads: 8 lanes per token row fetch its 128-B class segment (one cache line: ldl % 32 == 0),
// 32 rows per pass; writes: per class a wave stores 256 consecutive tokens as ONE 1-KiB instruction.  (k_finalize_nchw above
// writes 256-B pieces to K different planes per block and divides by a run-time K per element: 2.6 TB/s at C2.)
constexpr int FIN_T = 256, FIN_K = 32, FIN_LD = FIN_T + 4;
__global__ void __launch_bounds__(256) k_finalize_nchw_t(const float* __restrict__ prob, int ldl, float* __restrict__ out, int r, int N,
                                                          int K, float div, int nch) {
  __shared__ float tile[FIN_K][FIN_LD];
  const int b = blockIdx.z;
  const int k0 = blockIdx.y * FIN_K;
  const int n0 = blockIdx.x * FIN_T;
  const int tt = threadIdx.x >> 3, q = threadIdx.x & 7;        // row of the pass, 4-class group of the segment
#pragma unroll
  for (int pass = 0; pass < FIN_T / 32; ++pass) {
    const int n = n0 + pass * 32 + tt;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (n < N) {
      for (int ri = 0; ri < r; ++ri) {
        // (fragment-major: the 8 lanes of a token read its (g, h) pieces, 32 tokens of a pass 512-B runs)
        const f32x4 v = *reinterpret_cast<const f32x4*>(prob + prob_off(size_t(b * r + ri) * N + n, k0 + 4 * q, ldl, nch));
        s = ri == 0 ? v : s + v;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[4 * q + e][pass * 32 + tt] = s[e] / div;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int n = n0 + 4 * lane;
  if (n >= N) return;
  for (int kk = wv; kk < FIN_K; kk += 4) {
    const int k = k0 + kk;
    if (k >= K) break;
    *reinterpret_cast<f32x4*>(out + (size_t(b) * K + k) * N + n) = *reinterpret_cast<const f32x4*>(&tile[kk][4 * lane]);
  }
}

// ------------------------------------------------------------------------------------------------
// depth (depth/depth/models/depther/ddp.py:220-247; decode_heads/decode_head.py:264-269)
// ------------------------------------------------------------------------------------------------
// q[(b*r+ri)*N+n][c] = xproj[b*N+n][c] + wm[c] * d[(b*r+ri)*N+n]   (down conv over cat[x, depth_t], Cm = 1)
__global__ void __launch_bounds__(256) k_feat_depth(const float* __restrict__ xproj, const float* __restrict__ wm,
                                                     const float* __restrict__ d, float* __restrict__ q, int r, int N,
                                                     int rows) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= rows) return;
  const int b = m / (r * N), n = m % N;
  const f32x4 xp = *reinterpret_cast<const f32x4*>(xproj + (size_t(b) * N + n) * 256 + lane * 4);
  const f32x4 wv = *reinterpret_cast<const f32x4*>(wm + lane * 4);
  const float dv = d[m];
  *reinterpret_cast<f32x4*>(q + size_t(m) * 256 + lane * 4) = xp + wv * dv;   // row-major (published by publish_q)
}

// taps (M,32): column t = dy*3+dx holds w[:,dy,dx] . q[m]; depth[i][j] = relu(sum_t taps[(i+dy-1, j+dx-1)][t] + b) + eps
// (scale_up: sigmoid(.) * eps; depth/depth/models/decode_heads/decode_head.py:252-262)
__global__ void __launch_bounds__(256) k_depth_update(DepthUpdateArgs a) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  const int N = a.h * a.w;
  if (m >= a.B_r * N) return;
  const int img = m / N, n = m - img * N;
  const int i = n / a.w, j = n - i * a.w;
  // (branch-free: nine loads in flight, taps outside the map selected to zero - a conditional load each costs a serial round trip)
  float tv[9];
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int ii = i + dy - 1, jj = j + dx - 1;
      const bool ok = ii >= 0 && ii < a.h && jj >= 0 && jj < a.w;
      const float v = a.taps[(size_t(img) * N + min(max(ii, 0), a.h - 1) * a.w + min(max(jj, 0), a.w - 1)) * 32 + dy * 3 + dx];
      tv[dy * 3 + dx] = ok ? v : 0.f;
    }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t) s += tv[t];
  s += a.bias_ptr[0];
  const float pred = a.scale_up ? a.eps_depth / (1.0f + expf(-s)) : fmaxf(s, 0.f) + a.eps_depth;
  a.pred[m] = pred;
  if (!a.depth_t) return;   // head-only call: no sampler update
  float x0 = (pred - a.min_depth) / (a.max_depth - a.min_depth);
  x0 = (x0 * 2.0f - 1.0f) * a.bit_scale;
  x0 = fminf(fmaxf(x0, -a.bit_scale), a.bit_scale);
  const float xt = a.depth_t[m];
  const float eps = a.st.sigma * (xt - a.st.alpha * x0);   // sigma field = 1/sqrt(1-gamma_now)
  a.depth_t[m] = a.st.alpha_next * x0 + a.st.sigma_next * eps;
}

// The depth step head on the chain path, WITHOUT a GEMM.  The concat-conv over cat[x, depth_t] has ONE noisy-map channel
// (depth/depth/models/depther/ddp.py:236-237), so q = (W_x x + b) + w_m d and - value_proj and the sampling projections being linear -
//     v_0 = [W_v (W_x x + b) + b_v] + (W_v w_m) d,        s_0 = [W_cat (W_x x + b)] + (W_cat w_m) d + pos
// with the bracketed terms loop invariant (rvpad, rs: one projection pass per sample) and the rest a rank-1 update per token: every
// step's head is elementwise.  One wave per token: lane = 4 channels of the value row; lanes 0..15 the 64 sampling offsets (-> pixel
// coordinates), lanes 16..23 the 32 attention logits (softmax over a head's 4 points: the hardware exp2 / rcp of k_layer's P3 epilogue);
// the previous step's DDIM update (k_depth_update's arithmetic, scalar per wave) in front when a.upd is set.  q itself is never
// formed: layer 0 builds its residual from xproj and d (k_layer MODE 10).
struct DepthHeadDev {
  const float *rvpad, *rs, *wv, *ws, *py, *px, *taps, *dbias;
  float *dvec, *v_out, *samp_out;
  int R, h, w, has_upd, scale_up;
  float d_min, d_max, d_bit, d_eps, d_sig, d_alpha, d_alpha_next, d_sigma_next;
};
__global__ void __launch_bounds__(256) k_depth_head(DepthHeadDev a) {
  const int lane = threadIdx.x & 63;
  const int m = __builtin_amdgcn_readfirstlane(int(blockIdx.x) * 4 + int(threadIdx.x >> 6));      // wave-uniform: scalar loads below
  const int N = a.h * a.w, M = a.R * N;
  if (m >= M) return;
  const int img = m / N, n = m - img * N;
  const int i = n / a.w, j = n - i * a.w;
  float dv = a.dvec[m];
  if (a.has_upd) {
    const float* tb = a.taps + size_t(img) * N * 32;
    float tv[9];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int ii = i + dy - 1, jj = j + dx - 1;
        const bool ok = ii >= 0 && ii < a.h && jj >= 0 && jj < a.w;
        const float v = tb[size_t(min(max(ii, 0), a.h - 1) * a.w + min(max(jj, 0), a.w - 1)) * 32 + dy * 3 + dx];
        tv[dy * 3 + dx] = ok ? v : 0.f;
      }
    float sacc = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) sacc += tv[t];
    sacc += a.dbias[0];
    const float pred = a.scale_up ? a.d_eps / (1.0f + expf(-sacc)) : fmaxf(sacc, 0.f) + a.d_eps;
    float x0 = (pred - a.d_min) / (a.d_max - a.d_min);
    x0 = (x0 * 2.0f - 1.0f) * a.d_bit;
    x0 = fminf(fmaxf(x0, -a.d_bit), a.d_bit);
    const float epsn = a.d_sig * (dv - a.d_alpha * x0);
    dv = a.d_alpha_next * x0 + a.d_sigma_next * epsn;
    if (lane == 0) a.dvec[m] = dv;
  }
  const size_t vrow = size_t(img) * (a.h + 2) * (a.w + 2) + size_t(i + 1) * (a.w + 2) + (j + 1);
  const f32x4 rv = *reinterpret_cast<const f32x4*>(a.rvpad + vrow * 256 + lane * 4);
  const f32x4 wv = *reinterpret_cast<const f32x4*>(a.wv + lane * 4);
  *reinterpret_cast<f32x4*>(a.v_out + vrow * 256 + lane * 4) = rv + wv * dv;
  if (lane < 24) {
    const int col = lane * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(a.rs + size_t(m) * 96 + col) + *reinterpret_cast<const f32x4*>(a.ws + col) * dv;
    v += *reinterpret_cast<const f32x4*>(a.py + i * 96 + col) + *reinterpret_cast<const f32x4*>(a.px + j * 96 + col);
    if (lane < 16) {                       // offsets of head lane / 2, points 2 (lane & 1), + 1 -> pixel coordinates (x, y, x, y)
      const float fi = float(i), fj = float(j);
      v[0] += fj; v[1] += fi; v[2] += fj; v[3] += fi;
      *reinterpret_cast<f32x4*>(a.samp_out + (size_t(lane >> 1) * M + m) * 12 + (lane & 1) * 4) = v;
    } else {                               // attention weights of head lane - 16: softmax over its 4 points
      const float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_exp2f((v[e] - mx) * 1.44269504088896340736f);
      const float inv = __builtin_amdgcn_rcpf((v[0] + v[1]) + (v[2] + v[3]));
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= inv;
      *reinterpret_cast<f32x4*>(a.samp_out + (size_t(lane - 16) * M + m) * 12 + 8) = v;
    }
  }
}

__global__ void k_mean_r(const float* __restrict__ pred, float* __restrict__ out, int r, int N, int total) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int b = idx / N, n = idx - b * N;
  float s = 0.f;
  for (int ri = 0; ri < r; ++ri) s += pred[size_t(b * r + ri) * N + n];
  out[idx] = s / float(r);
}

// ------------------------------------------------------------------------------------------------
// bev (heads/segm/deformable_head_with_time.py:70-97; fusion_models/ddp.py:290-300)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float bev_src_coord(const BevGeom& g, int axis, int k, int in_len) {
  // c_k = first + k*step; normalised g = (c - imin)/(imax - imin)*2 - 1; pixel = ((g + 1)*len - 1)/2
  const float c = g.out_first[axis] + float(k) * g.out_step[axis];
  const float gn = (c - g.in_min[axis]) / (g.in_max[axis] - g.in_min[axis]) * 2.0f - 1.0f;
  return ((gn + 1.0f) * float(in_len) - 1.0f) * 0.5f;
}

// The four corners of a bilinear tap on a row-major (h*w, 256) map, zeros padding: BRANCH-FREE - clamped addresses, out-of-range corners
// selected to zero after the load - so that the four 1-KiB row loads of a token (and those of the next tokens) are all in flight
// together.  With `if (in range) acc += load * w` hipcc puts an s_waitcnt vmcnt(0) behind every conditional load: five serial memory
// round trips per token (r06c: k_bev_q 0.298 ms = 2.7 TB/s; the same sums in the same order, adding exact zeros where a corner is
// outside).
__device__ __forceinline__ f32x4 bev_bilinear_row(const float* __restrict__ base /* + lane * 4 */, float x, float y, int w, int h) {
  const float xf = floorf(x), yf = floorf(y);
  const float fx = x - xf, fy = y - yf;
  const int x0 = int(xf), y0 = int(yf);
  f32x4 v[2][2];
  bool ok[2][2];
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const int xx = x0 + dx, yy = y0 + dy;
      ok[dy][dx] = xx >= 0 && xx < w && yy >= 0 && yy < h;
      const int xc = min(max(xx, 0), w - 1), yc = min(max(yy, 0), h - 1);
      v[dy][dx] = *reinterpret_cast<const f32x4*>(base + size_t(yc * w + xc) * 256);
    }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const float wgt = (dy ? fy : 1.f - fy) * (dx ? fx : 1.f - fx);
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      acc += (ok[dy][dx] ? v[dy][dx] : z) * wgt;
    }
  return acc;
}

// bilinear (zeros padding, align_corners=False) resample of token-major feat (R, h*w, 256) onto (R, hh*wh, 256)
__global__ void __launch_bounds__(256) k_bev_resample(const float* __restrict__ feat, float* __restrict__ out, int R,
                                                       BevGeom g) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int Nh = g.hh * g.wh, N = g.h * g.w;
  if (m >= R * Nh) return;
  const int img = m / Nh, n = m - img * Nh;
  const int oi = n / g.wh, oj = n - oi * g.wh;
  const float y = bev_src_coord(g, 0, oi, g.h);
  const float x = bev_src_coord(g, 1, oj, g.w);
  const f32x4 acc = bev_bilinear_row(feat + size_t(img) * N * 256 + lane * 4, x, y, g.w, g.h);
  *reinterpret_cast<f32x4*>(out + size_t(m) * 256 + lane * 4) = acc;   // row-major (published by publish_q)
}

// The BEV sampler's u chain (ddp_api.hip, bev/mmdet3d/models/fusion_models/ddp.py:268-301 with the linear operators regrouped): the grid
// transform (bilinear grid_sample, zeros padding) and the 1x1 concat-conv are both linear, so
//     resample(W_x x + b + W_m m_t) = resample(W_x x + b) + resample(u_t),   u_t = W_m m_t   (at the map size h x w)
// - the first term is loop invariant, and u follows the DDIM update affinely because x0 is one of 2^K rows of a table (the
// thresholded maps select a mean embedding, :290-295): u' = ua u + uc T[code], T = LUT . W_m^T.
// q[(b r + ri) Nh + n] (fragment-major) = rx[b Nh + n] (row-major, shared by the r replicas) + resample(u[(b r + ri)])[n]
// One block = one 32-token group of the fragment-major q: each of its 4 waves forms 8 tokens' rows (a row = 64 lanes x 16 B: the four
// corner rows of u and the row of rx are coalesced 1-KiB loads), the rows meet in LDS and leave in fragment order - 8 coalesced 1-KiB
// stores per wave instead of 64 scattered 16-B pieces per token.  (What made this kernel slow was neither the store pattern nor the
// block order - 0.285 / 0.298 ms with either - but the conditional corner loads: bev_bilinear_row above, 0.18 ms; profiles/r06c, r06d.)
__global__ void __launch_bounds__(256) k_bev_q(const float* __restrict__ u, const float* __restrict__ rx, float* __restrict__ q_blk, int R,
                                                int r, BevGeom g) {
  __shared__ __attribute__((aligned(16))) float tile[32][260];        // +4 floats per row: the fragment-order reads spread over the banks
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int Nh = g.hh * g.wh, N = g.h * g.w;
  const int M = R * Nh;
  // XCD-aware order: workgroups go round-robin over the 8 XCDs (each with its own 4-MiB L2), so block b lands on XCD b % 8.  Group
  // gid = (b % 8) * ceil(groups / 8) + b / 8 gives every XCD ONE contiguous eighth of the token range (one map at R = 8): the u rows
  // neighbouring tokens share (each is a corner of ~2.4 x 4 tokens) are hits in that XCD's L2 instead of eight XCDs each pulling every
  // row through the fabric.  (Kept as the placement that bounds the fabric traffic; by itself it did not move the kernel's time - r06c.)
  const int groups = (M + 31) / 32, per = (groups + 7) / 8;
  const int gid = int(blockIdx.x & 7) * per + int(blockIdx.x >> 3);
  if (gid >= groups || int(blockIdx.x >> 3) >= per) return;
  const int m0 = gid * 32;
#pragma unroll 4
  for (int k = 0; k < 8; ++k) {
    const int jt = wave * 8 + k;
    const int m = min(m0 + jt, M - 1);                          // (rows past M: computed from the last token, never read)
    const int img = m / Nh, n = m - img * Nh;
    const int oi = n / g.wh, oj = n - oi * g.wh;
    const float y = bev_src_coord(g, 0, oi, g.h);
    const float x = bev_src_coord(g, 1, oj, g.w);
    const f32x4 rxv = *reinterpret_cast<const f32x4*>(rx + (size_t(img / r) * Nh + n) * 256 + lane * 4);
    f32x4 acc = bev_bilinear_row(u + size_t(img) * N * 256 + lane * 4, x, y, g.w, g.h);
    acc += rxv;
    *reinterpret_cast<f32x4*>(&tile[jt][lane * 4]) = acc;
  }
  __syncthreads();
  // fragment f = (t, g) * 64 + lane64, lane64 = h * 32 + j: channels 32 t + 8 g + 4 h .. + 3 of token j (gemm_f32.h)
  float* dst = q_blk + size_t(gid) * 8192;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int f = k * 256 + threadIdx.x;
    const int tg = f >> 6, l64 = f & 63;
    const int j = l64 & 31, h = l64 >> 5;
    const int ch = (tg >> 2) * 32 + (tg & 3) * 8 + 4 * h;
    *reinterpret_cast<f32x4*>(dst + size_t(f) * 4) = *reinterpret_cast<const f32x4*>(&tile[j][ch]);
  }
}

// LUT64[code][c] = (sigmoid(mean_k E[bit k of code ? k + 1 : 0][c]) * 2 - 1) * bit_scale: the 2^K values x0 can take at a pixel
// (fusion_models/ddp.py:291-295; the sum in class order, as k_bev_update forms it)
__global__ void k_build_bev_lut(const float* __restrict__ emb, float* __restrict__ lut, int K, float bit_scale) {
  const int code = blockIdx.x, c = threadIdx.x;
  float e = 0.f;
  for (int k = 0; k < K; ++k) e += emb[size_t(((code >> k) & 1) ? k + 1 : 0) * 256 + c];
  lut[size_t(code) * 256 + c] = (sigmoidf_(e / float(K)) * 2.0f - 1.0f) * bit_scale;
}

// u[(img, i, j)] = ua u + uc T[code[(img, nearest source of (i, j) in the head grid)]]   (F.interpolate(mode='nearest') of the
// thresholded maps, :293: src = min(floor(dst * (in / out)), in - 1))
__global__ void __launch_bounds__(256) k_bev_u_update(float* __restrict__ u, const unsigned char* __restrict__ code,
                                                       const float* __restrict__ tlut, int R, BevGeom g, float ua, float uc) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int Nh = g.hh * g.wh, N = g.h * g.w;
  if (m >= R * N) return;
  const int img = m / N, n = m - img * N;
  const int i = n / g.w, j = n - i * g.w;
  const int si = min(int(floorf(float(i) * (float(g.hh) / float(g.h)))), g.hh - 1);
  const int sj = min(int(floorf(float(j) * (float(g.wh) / float(g.w)))), g.wh - 1);
  const int cd = code[size_t(img) * Nh + si * g.wh + sj];
  float* up = u + size_t(m) * 256 + lane * 4;
  const f32x4 t = *reinterpret_cast<const f32x4*>(tlut + size_t(cd) * 256 + lane * 4);
  *reinterpret_cast<f32x4*>(up) = *reinterpret_cast<const f32x4*>(up) * ua + t * uc;
}

// (a) per head token: prob = sigmoid(logit), accumulate;  (b) per map token (h,w): nearest source in the
// head grid, threshold -> class ids -> mean embedding -> x0 -> DDIM update.  Two launches of this kernel
// body are avoided by doing (a) for all head tokens in blocks [0, nbA) and (b) in the rest.
__global__ void __launch_bounds__(256) k_bev_update(BevUpdateArgs a, int nbA) {
  const int lane = threadIdx.x & 63;
  const int Nh = a.g.hh * a.g.wh, N = a.g.h * a.g.w;
  const int K = a.num_classes;
  if (int(blockIdx.x) < nbA) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.R * Nh * K) return;
    const int m = idx / K, k = idx - m * K;
    const float p = sigmoidf_(a.logits[size_t(m) * 32 + k]);
    float* pr = a.prob + size_t(m) * 32 + k;
    *pr = a.first ? p : (*pr + p);
    return;
  }
  const int m = (blockIdx.x - nbA) * 4 + (threadIdx.x >> 6);
  if (m >= a.R * N || !a.mask) return;
  const int img = m / N, n = m - img * N;
  const int i = n / a.g.w, j = n - i * a.g.w;
  // F.interpolate(mode='nearest') (hh,wh) -> (h,w): src = min(floor(dst * (in/out)), in-1)
  const int si = min(int(floorf(float(i) * (float(a.g.hh) / float(a.g.h)))), a.g.hh - 1);
  const int sj = min(int(floorf(float(j) * (float(a.g.wh) / float(a.g.w)))), a.g.wh - 1);
  const float* lg = a.logits + (size_t(img) * Nh + si * a.g.wh + sj) * 32;
  f32x4 e = {0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < K; ++k) {
    const int id = (sigmoidf_(lg[k]) > a.threshold) ? (k + 1) : 0;
    e += *reinterpret_cast<const f32x4*>(a.emb + size_t(id) * 256 + lane * 4);
  }
  float* mp = a.mask + size_t(m) * 256 + lane * 4;
  const f32x4 mt = *reinterpret_cast<const f32x4*>(mp);
  const float sig = fmaxf(a.st.sigma, 1e-8f);
  f32x4 mn;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float x0 = (sigmoidf_(e[q] / float(K)) * 2.0f - 1.0f) * a.bit_scale;
    const float pn = (mt[q] - a.st.alpha * x0) / sig;
    mn[q] = x0 * a.st.alpha_next + pn * a.st.sigma_next;
  }
  *reinterpret_cast<f32x4*>(mp) = mn;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static inline int cdiv(long a, long b) { return int((a + b - 1) / b); }

int launch_nchw_to_tok(const float* in, float* out, int R, int C, int N, hipStream_t st) {
  hipLaunchKernelGGL(k_nchw_to_tok, dim3(cdiv(N, 64), cdiv(C, 64), R), dim3(256), 0, st, in, out, C, N);
  return check_launch("k_nchw_to_tok");
}
int launch_split_weights(const float* W, int ld, int rows, int K, unsigned short* out, hipStream_t st) {
  const long n = long(rows) * (K / 8);
  hipLaunchKernelGGL(k_split_weights, dim3(cdiv(n, 256)), dim3(256), 0, st, W, ld, rows, K, out);
  return check_launch("k_split_weights");
}
int launch_build_stages(const unsigned short* Wp, size_t comp_stride, int K, int rows_valid, int tall, int n_rowblk, int n_kblk,
                        int base, int a, int b, int c, unsigned char* stream, hipStream_t st) {
  hipLaunchKernelGGL(k_build_stages, dim3(12, n_rowblk * n_kblk), dim3(256), 0, st, Wp, comp_stride, K, rows_valid, tall,
                     n_kblk, base, a, b, c, stream);
  return check_launch("k_build_stages");
}
int launch_row_to_sb(const float* in, int ld, unsigned short* out_sb, int rows, int C, hipStream_t st) {
  const long n = long(rows) * (C / 8);
  hipLaunchKernelGGL(k_row_to_sb, dim3(cdiv(n, 256)), dim3(256), 0, st, in, ld, out_sb, rows, C);
  return check_launch("k_row_to_sb");
}
int launch_nchw_to_sb(const float* in, unsigned short* out_sb, int R, int C, int N, hipStream_t st) {
  const long rows = long(R) * N;
  hipLaunchKernelGGL(k_nchw_to_sb, dim3(cdiv(rows, 64), cdiv(C, 64)), dim3(256), 0, st, in, out_sb, C, N, int(rows));
  return check_launch("k_nchw_to_sb");
}
int launch_row_to_blk(const float* in, float* out_blk, int rows, hipStream_t st) {
  hipLaunchKernelGGL(k_row_to_blk, dim3(cdiv(rows, 4)), dim3(256), 0, st, in, out_blk, rows);
  return check_launch("k_row_to_blk");
}
int launch_fold_affine(const float* gamma, const float* beta, const float* film, float* out, int count, hipStream_t st) {
  if (count <= 0) return DDP_OK;
  hipLaunchKernelGGL(k_fold_affine, dim3(count), dim3(256), 0, st, gamma, beta, film, out);
  return check_launch("k_fold_affine");
}
int launch_msda_gather(const float* value, const float* samp, float* out, int rows, int n_tok, int h, int w,
                       hipStream_t st) {
  hipLaunchKernelGGL(k_msda_gather, dim3(cdiv(rows, 4)), dim3(256), 0, st, value, samp, out, rows, n_tok, h, w);
  return check_launch("k_msda_gather");
}
int launch_msda_gather_sb(const float* value, const float* samp, unsigned short* out_sb, int rows, int n_tok, int h, int w,
                          hipStream_t st) {
  hipLaunchKernelGGL(k_msda_gather_sb, dim3(cdiv(rows, 32)), dim3(64 * GSB_WAVES), 0, st, value, samp, out_sb, rows, n_tok, h, w);
  return check_launch("k_msda_gather_sb");
}
int launch_seg_postprocess(const SegPostArgs& a, hipStream_t st) {
  if (a.H == 4 * a.h && a.W == 4 * a.w && a.oh == a.ch && a.ow == a.cw && a.ch == a.H && a.cw == a.W && !a.align) {
    hipLaunchKernelGGL(k_seg_postprocess_x4, dim3(cdiv(a.w, 64), cdiv(a.h, 4), a.B), dim3(256), 0, st, a);
    return check_launch("k_seg_postprocess_x4");
  }
  hipLaunchKernelGGL(k_seg_postprocess, dim3(cdiv(a.ow, 64), cdiv(a.oh, 4), a.B), dim3(256), 0, st, a);
  return check_launch("k_seg_postprocess");
}
int launch_seg_aug_postprocess(const ddp_seg_aug* augs, int n_aug, int B, int K, int oh, int ow, int align, unsigned char* seg,
                               float* prob, hipStream_t st) {
  SegAugArgs a;
  for (int i = 0; i < n_aug; ++i) a.aug[i] = augs[i];
  a.n_aug = n_aug;
  a.B = B;
  a.K = K;
  a.oh = oh;
  a.ow = ow;
  a.align = align;
  a.seg = seg;
  a.prob = prob;
  const int lds = (2 * K * 64 + 512) * int(sizeof(float));
  // the attribute is set once per device: ask for the largest size the API accepts (K = 256: 130 KiB), not this call's
  static LdsAttrOnce attr;
  attr.ensure(reinterpret_cast<const void*>(k_seg_aug_postprocess), (2 * 256 * 64 + 512) * int(sizeof(float)));
  hipLaunchKernelGGL(k_seg_aug_postprocess, dim3(cdiv(ow, 64), oh, B), dim3(256), lds, st, a);
  return check_launch("k_seg_aug_postprocess");
}
int launch_seg_slide_postprocess(const float* const* scores, const int* y1, const int* x1, int n_rows, int n_cols, int B, int K, int h,
                                 int w, int ch, int cw, int H, int W, int kh, int kw, int oh, int ow, int align, int flip, int prob_mode,
                                 unsigned char* seg, float* prob, hipStream_t st) {
  SlideArgs a;
  for (int i = 0; i < n_rows * n_cols; ++i) a.scores[i] = scores[i];
  for (int i = 0; i < n_rows; ++i) a.y1[i] = y1[i];
  for (int i = 0; i < n_cols; ++i) a.x1[i] = x1[i];
  a.n_rows = n_rows; a.n_cols = n_cols; a.B = B; a.K = K; a.h = h; a.w = w; a.ch = ch; a.cw = cw; a.H = H; a.W = W;
  a.kh = kh; a.kw = kw; a.oh = oh; a.ow = ow; a.align = align; a.flip = flip; a.prob_mode = prob_mode;
  a.seg = seg;
  a.prob = prob;
  const int lds = prob_mode ? K * 64 * int(sizeof(float)) : 0;
  static LdsAttrOnce attr;
  attr.ensure(reinterpret_cast<const void*>(k_seg_slide_postprocess), 256 * 64 * int(sizeof(float)));
  hipLaunchKernelGGL(k_seg_slide_postprocess, dim3(cdiv(ow, 64), oh, B), dim3(64), lds, st, a);
  return check_launch("k_seg_slide_postprocess");
}
int launch_depth_aug_postprocess(const ddp_depth_aug* augs, int n_aug, int B, int oh, int ow, int align, float lo, float hi,
                                 float* out, hipStream_t st) {
  DepthAugArgs a;
  for (int i = 0; i < n_aug; ++i) a.aug[i] = augs[i];
  a.n_aug = n_aug;
  a.B = B;
  a.oh = oh;
  a.ow = ow;
  a.align = align;
  a.lo = lo;
  a.hi = hi;
  a.out = out;
  hipLaunchKernelGGL(k_depth_aug_postprocess, dim3(cdiv(ow, 256), cdiv(oh, 4), B), dim3(256), 0, st, a);
  return check_launch("k_depth_aug_postprocess");
}
int launch_msda_gather_sb_pad(const float* vpad, const float* samp, unsigned short* out_sb, float* out_f32_blk, int rows, int n_tok,
                              int h, int w, const float* tab_y, const float* tab_x, int zero_guess, hipStream_t st) {
#ifndef DDP_GL_TH
#define DDP_GL_TH 8
#define DDP_GL_TW 16
#define DDP_GL_NT GL_THREADS
#define DDP_GL_MINW (DDP_GL_HALO <= 3 ? (DDP_GL_TRIM >= 1 ? 8 : 6) : 4)
#endif
  constexpr int TH = DDP_GL_TH, TW = DDP_GL_TW, NT = DDP_GL_NT;
  const int tiles_x = cdiv(w, TW), tiles_y = cdiv(h, TH);
  const int n_tiles = (rows / n_tok) * tiles_x * tiles_y;
  prof_begin(TAG_GATHER, st);
  if (out_f32_blk)
    hipLaunchKernelGGL((k_msda_gather_lds<TH, TW, DDP_GL_HALO, DDP_GL_MINW, NT, true>), dim3(n_tiles * 8), dim3(NT), 0, st, vpad, samp,
                       reinterpret_cast<unsigned short*>(out_f32_blk), n_tok, h, w, tiles_x, tiles_y, n_tiles, rows, tab_y, tab_x,
                       zero_guess);
  else
    hipLaunchKernelGGL((k_msda_gather_lds<TH, TW, DDP_GL_HALO, DDP_GL_MINW, NT, false>), dim3(n_tiles * 8), dim3(NT), 0, st, vpad, samp, out_sb,
                       n_tok, h, w, tiles_x, tiles_y, n_tiles, rows, tab_y, tab_x, zero_guess);
  prof_end(TAG_GATHER, st);
  return check_launch("k_msda_gather_lds");
}
// fragment-major fp32 (accumulator layout, 256 channels) -> row-major: one wave per token
__global__ void __launch_bounds__(256) k_blk_to_row(const float* __restrict__ in, float* __restrict__ out, int rows, int ld, int cols) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= rows || lane * 4 >= cols) return;
  *reinterpret_cast<f32x4*>(out + size_t(m) * ld + lane * 4) = *reinterpret_cast<const f32x4*>(in + blk_off256(m, lane * 4));
}
int launch_blk_to_row(const float* in_blk, float* out, int rows, hipStream_t st, int ld, int cols) {
  hipLaunchKernelGGL(k_blk_to_row, dim3(cdiv(rows, 4)), dim3(256), 0, st, in_blk, out, rows, ld, cols);
  return check_launch("k_blk_to_row");
}
int launch_msda_lds_adapters_in(const float* value, const float* samp, const float* guess, float* vpad, size_t vpad_floats,
                                float* samp_hm, float* tab_y, float* tab_x, int rows, int n_tok, int h, int w, hipStream_t st) {
  if (hipMemsetAsync(vpad, 0, vpad_floats * sizeof(float), st) != hipSuccess) {
    set_error("hipMemsetAsync(vpad) failed");
    return DDP_E_LAUNCH;
  }
  hipLaunchKernelGGL(k_pad_value, dim3(cdiv(rows, 4)), dim3(256), 0, st, value, vpad, rows, n_tok, w, h);
  hipLaunchKernelGGL(k_samp_head_major, dim3(cdiv(long(rows) * 8, 256)), dim3(256), 0, st, samp, samp_hm, rows);
  hipLaunchKernelGGL(k_fill_guess_tables, dim3(cdiv(long(h > w ? h : w) * 96, 256)), dim3(256), 0, st, guess, tab_y, tab_x, h, w);
  return check_launch("msda_lds adapters (in)");
}
int launch_sb_to_row(const unsigned short* in_sb, float* out, int rows, int C, hipStream_t st) {
  hipLaunchKernelGGL(k_sb_to_row, dim3(cdiv(long(rows) * (C / 8), 256)), dim3(256), 0, st, in_sb, out, rows, C);
  return check_launch("k_sb_to_row");
}
int launch_nchw_to_blk(const float* in, float* out_blk, int R, int C, int N, hipStream_t st) {
  const long rows = long(R) * N;
  hipLaunchKernelGGL(k_nchw_to_blk, dim3(cdiv(rows, 64), cdiv(C, 64)), dim3(256), 0, st, in, out_blk, C, N, int(rows));
  return check_launch("k_nchw_to_blk");
}
int launch_gn_stats_blk(const float* y_blk, double* partial, float* stats, int B, int N, float eps, hipStream_t st) {
  const int chunks = cdiv(N, 256);
  hipLaunchKernelGGL(k_gn_partial<true>, dim3(chunks, B), dim3(256), 0, st, y_blk, partial, N, chunks);
  hipLaunchKernelGGL(k_gn_final, dim3(B), dim3(32), 0, st, partial, stats, N, chunks, eps);
  return check_launch("gn_stats_blk");
}
int launch_gn_final32(const double* partial, float* stats, int B, int N, float eps, hipStream_t st) {
  hipLaunchKernelGGL(k_gn_final32, dim3(B), dim3(1024), 0, st, partial, stats, N, N / 32, eps);
  return check_launch("k_gn_final32");
}
int launch_gn_apply_add_blk(const float* y_blk, const float* stats, const float* gamma, const float* beta, const float* coarse_blk,
                            float* out_blk, int B, int hf, int wf, int hc, int wc, hipStream_t st) {
  const int rows = B * hf * wf;
  hipLaunchKernelGGL(k_gn_apply_add_blk, dim3(cdiv(long(cdiv(rows, 32)) * 2048, 256)), dim3(256), 0, st, y_blk, stats, gamma, beta,
                     coarse_blk, out_blk, hf, wf, hc, wc, rows);
  return check_launch("k_gn_apply_add_blk");
}
int launch_gn_apply_nchw_blk(const float* y_blk, const float* stats, const float* gamma, const float* beta, float* out, int B, int N,
                             hipStream_t st) {
  hipLaunchKernelGGL(k_gn_apply_nchw<true>, dim3(cdiv(N, 64), 4, B), dim3(256), 0, st, y_blk, stats, gamma, beta, out, N);
  return check_launch("k_gn_apply_nchw<blk>");
}
int launch_gn_final32_multi(const double* const* partial, float* const* stats, const int* N, int n_maps, int B, float eps, hipStream_t st) {
  GnFinalMulti a;
  for (int l = 0; l < 4; ++l) {
    a.partial[l] = l < n_maps ? partial[l] : nullptr;
    a.stats[l] = l < n_maps ? stats[l] : nullptr;
    a.N[l] = l < n_maps ? N[l] : 32;
  }
  hipLaunchKernelGGL(k_gn_final32_multi, dim3(B, n_maps), dim3(1024), 0, st, a, eps);
  return check_launch("k_gn_final32_multi");
}
int launch_msm_sum_blk(float* y0, const float* const* yl, const int* lh, const int* lw, int B, int h, int w, int align, hipStream_t st,
                       double* gn_partial) {
  MsmSumArgs a;
  a.gn_partial = (h * w) % 32 == 0 ? gn_partial : nullptr;
  a.y0 = y0;
  for (int l = 0; l < 3; ++l) {
    a.yl[l] = yl[l];
    a.lh[l] = lh[l];
    a.lw[l] = lw[l];
  }
  a.h = h;
  a.w = w;
  a.rows = B * h * w;
  a.align = align;
  hipLaunchKernelGGL(k_msm_sum_blk, dim3(cdiv(long(cdiv(a.rows, 32)) * 2048, 256)), dim3(256), 0, st, a);
  return check_launch("k_msm_sum_blk");
}
int launch_pack_conv3x3_scaled(const float* w, const float* scale, float* out, int cout, int cin, hipStream_t st) {
  hipLaunchKernelGGL(k_pack_conv3x3_scaled, dim3(cdiv(long(cout) * 9 * cin, 256)), dim3(256), 0, st, w, scale, out, cout, cin);
  return check_launch("k_pack_conv3x3_scaled");
}
int launch_fcn_fold(const float* bn_w, const float* bn_b, const float* bn_mean, const float* bn_var, float bn_eps,
                    const float* conv_bias, const float* film, float* scale, float* shift, hipStream_t st) {
  hipLaunchKernelGGL(k_fcn_fold, dim3(1), dim3(256), 0, st, bn_w, bn_b, bn_mean, bn_var, bn_eps, conv_bias, film, scale, shift);
  return check_launch("k_fcn_fold");
}
int launch_sinusoid(const float* freq, const float* t_in, int S, float* u, hipStream_t st) {
  hipLaunchKernelGGL(k_sinusoid, dim3(S), dim3(64), 0, st, freq, t_in, S, u);
  return check_launch("k_sinusoid");
}
int launch_matvec(const float* W, const float* b, const float* x, float* y, int in_dim, int out_dim, int S,
                  int ldx, int ldy, int in_act, int out_act, hipStream_t st) {
  hipLaunchKernelGGL(k_matvec, dim3(cdiv(long(out_dim) * S, 4)), dim3(256), 0, st, W, b, x, y, in_dim, out_dim, S,
                     ldx, ldy, in_act, out_act);
  return check_launch("k_matvec");
}
int launch_build_lut(const float* emb, float* lut, int rows, float bit_scale, hipStream_t st) {
  const int n = rows * 256;
  hipLaunchKernelGGL(k_build_lut, dim3(cdiv(n, 256)), dim3(256), 0, st, emb, lut, n, bit_scale);
  return check_launch("k_build_lut");
}
int launch_pos_tables(const float* wcat, const float* bcat, float* py, float* px, int h, int w, hipStream_t st) {
  hipLaunchKernelGGL(k_pos_tables, dim3(cdiv(long(h + w) * 96, 4)), dim3(256), 0, st, wcat, bcat, py, px, h, w);
  return check_launch("k_pos_tables");
}
int launch_pack_rows(const float* a, int rows_a, const float* b, int rows_b, float* out, int cols, hipStream_t st) {
  const int na = rows_a * cols, nb = rows_b * cols;
  hipLaunchKernelGGL(k_pack_rows, dim3(cdiv(na + nb, 256)), dim3(256), 0, st, a, na, b, nb, out);
  return check_launch("k_pack_rows");
}
int launch_pack_cols(const float* in, int ld_in, int off, int rows, int cols, float* out, hipStream_t st) {
  hipLaunchKernelGGL(k_pack_cols, dim3(cdiv(long(rows) * cols, 256)), dim3(256), 0, st, in, ld_in, off, rows, cols, out);
  return check_launch("k_pack_cols");
}
int launch_pack_conv3x3(const float* w, float* out, hipStream_t st) {
  hipLaunchKernelGGL(k_pack_conv3x3, dim3(9), dim3(256), 0, st, w, out);
  return check_launch("k_pack_conv3x3");
}
int launch_write_floats(const float* host_vals, int n, float* out, hipStream_t st) {
  FloatList f;
  for (int i = 0; i < DDP_MAX_STEPS; ++i) f.v[i] = i < n ? host_vals[i] : 0.f;
  hipLaunchKernelGGL(k_write_floats, dim3(1), dim3(64), 0, st, f, n, out);
  return check_launch("k_write_floats");
}
int launch_seg_update(const SegUpdateArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(k_seg_update, dim3(cdiv(a.rows, 4)), dim3(256), 0, st, a);
  return check_launch("k_seg_update");
}
int launch_seg_x0_nchw(const float* scores, const float* emb, float* out, int B, int K, int N, float bit_scale, hipStream_t st) {
  hipLaunchKernelGGL(k_seg_x0_nchw, dim3(cdiv(N, 256), B), dim3(256), 0, st, scores, emb, out, K, N, bit_scale);
  return check_launch("k_seg_x0_nchw");
}
int launch_finalize_nchw(const float* prob, int ldl, float* out, int B, int r, int N, int K, float div,
                         hipStream_t st, int frag_nch) {
  if (N % 4 == 0 && (ldl % 32 == 0 || frag_nch) && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
    hipLaunchKernelGGL(k_finalize_nchw_t, dim3(cdiv(N, FIN_T), cdiv(K, FIN_K), B), dim3(256), 0, st, prob, ldl, out, r, N, K, div, frag_nch);
    return check_launch("k_finalize_nchw_t");
  }
  const size_t lds = size_t(64) * (K + 1) * sizeof(float);
  static LdsAttrOnce attr;
  attr.ensure(reinterpret_cast<const void*>(&k_finalize_nchw), 64 * 257 * int(sizeof(float)));
  hipLaunchKernelGGL(k_finalize_nchw, dim3(cdiv(N, 64), B), dim3(256), lds, st, prob, ldl, out, r, N, K, div, frag_nch);
  return check_launch("k_finalize_nchw");
}
int launch_feat_depth(const float* xproj, const float* wm, const float* d, float* q, int B, int r, int N,
                      hipStream_t st) {
  const int rows = B * r * N;
  hipLaunchKernelGGL(k_feat_depth, dim3(cdiv(rows, 4)), dim3(256), 0, st, xproj, wm, d, q, r, N, rows);
  return check_launch("k_feat_depth");
}
int launch_depth_update(const DepthUpdateArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(k_depth_update, dim3(cdiv(long(a.B_r) * a.h * a.w, 256)), dim3(256), 0, st, a);
  return check_launch("k_depth_update");
}
int launch_depth_head(const DepthHeadArgs& a, hipStream_t st) {
  DepthHeadDev d;
  d.rvpad = a.rvpad; d.rs = a.rs; d.wv = a.wv; d.ws = a.ws; d.py = a.py; d.px = a.px;
  d.dvec = a.dvec; d.v_out = a.v_out; d.samp_out = a.samp_out;
  d.R = a.R; d.h = a.h; d.w = a.w;
  d.has_upd = a.upd ? 1 : 0;
  d.taps = nullptr; d.dbias = nullptr; d.scale_up = 0;
  d.d_min = d.d_max = d.d_bit = d.d_eps = d.d_sig = d.d_alpha = d.d_alpha_next = d.d_sigma_next = 0.f;
  if (a.upd) {
    const DepthUpdateArgs& u = *a.upd;
    if (u.depth_t != a.dvec || u.B_r != a.R || u.h != a.h || u.w != a.w) {
      set_error("depth head: the fused update must work on the head's own map");
      return DDP_E_BADCFG;
    }
    d.taps = u.taps; d.dbias = u.bias_ptr; d.scale_up = u.scale_up;
    d.d_min = u.min_depth; d.d_max = u.max_depth; d.d_bit = u.bit_scale; d.d_eps = u.eps_depth;
    d.d_sig = u.st.sigma; d.d_alpha = u.st.alpha; d.d_alpha_next = u.st.alpha_next; d.d_sigma_next = u.st.sigma_next;
  }
  hipLaunchKernelGGL(k_depth_head, dim3(cdiv(long(a.R) * a.h * a.w, 4)), dim3(256), 0, st, d);
  return check_launch("k_depth_head");
}
int launch_mean_r(const float* pred, float* out, int B, int r, int N, hipStream_t st) {
  hipLaunchKernelGGL(k_mean_r, dim3(cdiv(long(B) * N, 256)), dim3(256), 0, st, pred, out, r, N, B * N);
  return check_launch("k_mean_r");
}
int launch_bev_resample(const float* feat, float* out, int R, const BevGeom& g, hipStream_t st) {
  hipLaunchKernelGGL(k_bev_resample, dim3(cdiv(long(R) * g.hh * g.wh, 4)), dim3(256), 0, st, feat, out, R, g);
  return check_launch("k_bev_resample");
}
int launch_bev_q(const float* u, const float* rx, float* q_blk, int R, int r, const BevGeom& g, hipStream_t st) {
  const int groups = cdiv(long(R) * g.hh * g.wh, 32);                      // (q holds whole 256-token tiles)
  hipLaunchKernelGGL(k_bev_q, dim3(8 * cdiv(groups, 8)), dim3(256), 0, st, u, rx, q_blk, R, r, g);
  return check_launch("k_bev_q");
}
int launch_build_bev_lut(const float* emb, float* lut, int K, float bit_scale, hipStream_t st) {
  hipLaunchKernelGGL(k_build_bev_lut, dim3(1 << K), dim3(256), 0, st, emb, lut, K, bit_scale);
  return check_launch("k_build_bev_lut");
}
int launch_bev_u_update(float* u, const unsigned char* code, const float* tlut, int R, const BevGeom& g, float ua, float uc,
                        hipStream_t st) {
  hipLaunchKernelGGL(k_bev_u_update, dim3(cdiv(long(R) * g.h * g.w, 4)), dim3(256), 0, st, u, code, tlut, R, g, ua, uc);
  return check_launch("k_bev_u_update");
}
int launch_bev_update(const BevUpdateArgs& a, hipStream_t st) {
  const int nbA = cdiv(long(a.R) * a.g.hh * a.g.wh * a.num_classes, 256);
  const int nbB = cdiv(long(a.R) * a.g.h * a.g.w, 4);
  hipLaunchKernelGGL(k_bev_update, dim3(nbA + nbB), dim3(256), 0, st, a, nbA);
  return check_launch("k_bev_update");
}

}  // namespace ddp
