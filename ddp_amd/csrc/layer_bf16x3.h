// layer_bf16x3.h - everything of a decoder layer that follows the deformable gather, in ONE persistent kernel
// (bf16x3-split arithmetic, see gemm_bf16x3.h):
//
//     x   = LayerNorm0( q + Wo . s + bo )                                  output_proj + identity + norms.0
//     q'  = FiLM( LayerNorm1( x + W2 . GELU(W1 . x + b1) + b2 ) )          FFN + identity + norms.1 + time FiLM
//     v'  = Wv' . q' + bv' ;  samp' = softmax4/pixel-coords( Wcat' . q' + pos' )   NEXT layer's projections
//
// (utils/transformer.py:352-358,390-392,413-417; mmcv FFN transformer.py:269-280; multi_scale_deform_attn.py:313-334)
//
// Why one kernel.  With the 2.65x faster bf16x3 contractions the layer was HBM bound; every tensor between these
// GEMMs (q1, the 1024-wide hidden activation, and q' as the operand of the next projections) now stays in the
// register file.  Per token the kernel reads s (1.5 KiB, SB) + q (1 KiB, fp32) and writes q' (1 KiB) + v' (1 KiB) +
// samp' (0.4 KiB).
//
//   * a wave owns 32 tokens; the block's 4 waves (one per SIMD, 512 registers each) share the weight stream;
//   * an accumulator lane (token j, half h) holds channels 32t + 8g + 4h + e, and the k-slots of an MFMA B operand
//     may be ANY fixed permutation of K shared with the (pre-permuted) weights, so "split the accumulator into
//     three bf16 pieces" IS "build the B fragment of the next contraction": LayerNorm0's output feeds fc1,
//     GELU(fc1) feeds fc2, LayerNorm1's output feeds the next layer's projections - no LDS round trip, no shuffle;
//   * ALL weights of the kernel are one linear STREAM of 48 KiB stage images (already in the swizzled LDS layout,
//     built once by k_build_stages) in the order they are consumed: 8 output_proj stages, 16 x (2 fc1 + 2 fc2)
//     stages, 8 + 3 stages of the next layer's projections (the last one split-K: 32 outputs x 256 k).  A 3-slot LDS ring runs two stages ahead by LDS-DMA;
//     every piece is "1 KiB from stream + off to ring + off", so four pieces share one M0 / one base (the
//     instruction's immediate offset moves BOTH sides - scripts/ubench/lds_dma_off.hip) and the stream simply wraps
//     to the next token tile: the ring never drains between tiles (persistent blocks).
// One-wave-per-SIMD scheduling rules (MI355X_MICROARCH.md latency table; each one measured here):
//   * two accumulator chains are always interleaved - a dependent MFMA right behind its producer is free only when
//     NOTHING is issued between them, and loads / GELU / DMA have to sit somewhere;
//   * a weight fragment is re-read into its own registers as soon as its last MFMA of the block has issued
//     (w2 after 2 MFMAs, w1 after 6, w0 after 12 - terms ordered for that), >= 6 MFMAs before its next use;
//   * the 12 DMA pieces of a stage are spread over its 8 MFMA blocks;
//   * behind one MFMA a single wave hides 5 independent instructions but only 3 of one dependency chain
//     (scripts/ubench/mfma_fill.hip): every MFMA has its own filler slot, and GELU + split of k-block n+1 is handed
//     out to the slots of k-block n as single instructions of four values at a time (GELU_SCHED).
// The same machinery runs the other per-step pieces (template parameter MODE):
//   1  tail of a segmentation step: conv_seg, argmax, softmax accumulation (+ the legacy update of an SB noisy map);
//   2  head of the FIRST step: q = W_m . noise + (W_x x + b), u_0 = W_m . noise kept in fp32, layer 0's projections;
//   3  layer 0's projections alone (bev, ddp_head_forward) / the depth step head (one-channel concat-conv, no GEMM);
//   4  tail of step s fused with the head of step s + 1: the DDIM update runs on u = W_m . m (affine in u and a row of
//      the table W_m . LUT) - no concat-conv GEMM and no 256-channel noisy map after step 0.
//   7  the head of the FIRST step straight from the caller's NCHW tensors: u_0 = W_m . noise (8 wide stages), xproj = W_x x + b
//      (8 more), q = xproj + u_0, layer 0's projections.  The B fragments are read from the NCHW planes with 16 coalesced dword
//      loads per stage (a lane's token is a column of the MFMA, 32 x-adjacent tokens = one 128-B run of a channel plane) and
//      split in the filler slots like the attention output of MODE 0: replaces two NCHW -> SB conversions, the x-projection
//      GEMM and MODE 2 (ddp.py:223-224 with the x columns hoisted out of the loop).
//   6  the LAST decoder layer of a step (MODE 0 without a next layer) + that step's tail (MODE 4, or MODE 1 after the last
//      step: has_next = 0) in one kernel: LayerNorm1's split output IS conv_seg's operand, exactly as it feeds P3 in MODE 0,
//      so the layer output never travels to HBM (1 KiB written + 1 KiB read per token and step) and a launch with its
//      ring prologue / drain goes away.  One concatenated weight stream: 72 layer stages + 2 NCH conv_seg stages + 11
//      projection stages of layer 0.
//   8  the LAST decoder layer of a BEV step + that step's tail: conv_seg (<= 32 classes), sigmoid, accumulation of the
//      probabilities (token-major rows of 32) and - for <= 8 classes - the step's x0 CODE per token: bit k = (sigmoid_k > threshold)
//      (bev/mmdet3d/models/fusion_models/ddp.py:290-293: the thresholded maps select one of 2^K mean embeddings).
//   9  the LAST decoder layer of a depth step + the nine per-tap dot products of the 3x3 conv_depth (token-major rows of 32: column
//      dy*3+dx; k_depth_update sums the neighbours' taps - depth/depth/models/decode_heads/decode_head.py:264-269).
//   10 layer 0 of a DEPTH step on the chain path (ddp_api.hip): MODE 0 whose residual is FORMED here - q = xproj + w_m d, the depth
//      concat-conv with its x half hoisted (depth/depth/models/depther/ddp.py:236-237) - from the fragment-major xproj (la.res) and
//      the noisy depth (la.dvec); the step head (k_depth_head) wrote only layer 0's value map and sample table, q never exists.
// Where the cycles of MODE 0 go is measured, not estimated: -DDDP_LYR_STAMP builds + scripts/stamp_layer.py
// (profiles/r02_layer_cycle_stamps*.json).
#pragma once
#include <type_traits>

#include "gemm_bf16x3.h"

namespace ddp {
namespace b3 {

constexpr int LYR_BM = 128;
constexpr int LYR_THREADS = 256;
constexpr int LYR_STAGE_B = 48 * 1024;
constexpr int LYR_RING = 3;                                   // stages in flight: compute q, landed q+1, landing q+2
constexpr int LYR_BIAS_OFF = LYR_RING * LYR_STAGE_B;
constexpr int LYR_BIAS_N = 1024 + 256 + 128;                  // fc1 bias | next value_proj bias | zeros (sampling proj)
// per-channel vectors kept in LDS behind the bias table (their pointers die after the kernel prologue)
constexpr int LYR_T_BO = LYR_BIAS_N, LYR_T_GA0 = LYR_T_BO + 256, LYR_T_BE0 = LYR_T_GA0 + 256, LYR_T_B2 = LYR_T_BE0 + 256,
              LYR_T_GA1 = LYR_T_B2 + 256, LYR_T_BE1 = LYR_T_GA1 + 256,
              LYR_T_SEG = LYR_T_BE1 + 256,                    // MODE 6: conv_seg's bias (tab[0..) belongs to fc1's there)
              LYR_TABLE_N = LYR_T_SEG + 256;
static_assert(LYR_T_SEG % 64 == 0, "bias_init addresses the table in chunks of 64 floats");
constexpr size_t LYR_LDS_B = size_t(LYR_BIAS_OFF) + LYR_TABLE_N * 4;
constexpr int LYR_ST_OUT = 8, LYR_ST_FFN = 64, LYR_ST_NEXT = 11;   // next: 8 value_proj + 2 + 1 (split-K) sampling stages
constexpr int LYR_STAGES = LYR_ST_OUT + LYR_ST_FFN + LYR_ST_NEXT;

struct LayerArgs {
  const unsigned short* S;       // SB A operand of P0 (MODE 2: the start noise; MODE 0 without DDP_S_F32: the attention output)
  const float* Sf;               // MODE 0 (DDP_S_F32): the attention output as fp32 in the accumulator layout (same as Q): 1 KiB per
                                 // token instead of the 1.5 KiB of SB, written by the gather without a split; P0 splits it in the
                                 // filler slots of its own MFMAs
  float* Q;                      // layer input (residual) -> layer output, in place (a block touches only its tokens): fp32 in the
                                 // accumulator layout (fragment-major, gemm_f32.h): [32-token group][tile t][quad g][lane][4] - 4 B per
                                 // element instead of the 6 B of SB; whoever needs it as an MFMA operand splits it after the load
  unsigned short* Q_sb;          // optional: the layer output ALSO as SB (a tile GEMM follows: head conv of depth / bev / ddpm)
  const unsigned char* stream;   // LYR_STAGES stage images
  const float* bias_ext;         // (LYR_BIAS_N)
  const float* bo;               // output_proj bias (256)
  const float* ga0;              // norms.0 weight / bias
  const float* be0;
  const float* b2;               // fc2 bias
  const float* ga1;              // norms.1 affine x FiLM (folded)
  const float* be1;
  int M;
  int has_next;                  // also emit the next layer's value / sampling projections
  float* v_out;                  // fp32 rows of 256, ZERO-PADDED map: token (b,i,j) -> row b*(h+2)*(w+2) + (i+1)*(w+2) + (j+1)
  float* samp_out;               // (8 heads, M, 12): per head and token 4 x (x, y) pixel coordinates + 4 attention weights
  const float* py;               // next layer's positional tables (h,96) / (w,96), bias folded in
  const float* px;
  int n_tok, w;
  // MODE 1 (seg tail): conv_seg + x0 projection + DDIM update + softmax accumulation on the tokens of Q
  const float* lut;              // (num_classes + 1, 256): (sigmoid(embedding) * 2 - 1) * bit_scale
  float* prob;                   // accumulated softmax / last-step scores, FRAGMENT-major: per 32-token group [chunk c][t][g][lane 64][4]
                                 // = NCH * 2048 floats (class 64c + 32t + 8g + 4h + e of token j at lane h * 32 + j): every access of the
                                 // tails is one coalesced 1-KiB instruction (token-major rows: 32 rows x 32 B per instruction)
  unsigned short* mask_sb;       // SB noisy map m_t (256 ch): read, replaced by m_{t_next}
  unsigned char* x0_idx;         // optional (DDP_FLAG_RECORD_X0): the step's argmax class per token
  const unsigned char* x0_force; // FORCE instantiations (DDP_FLAG_FORCE_X0): the class fed back INSTEAD of the argmax
  int num_classes, ldl, prob_mode;   // prob_mode: 0 none, 1 prob = softmax, 2 prob += softmax, 3 prob = scores
  float threshold;               // MODE 8: the bev sampler's threshold on the sigmoid maps (x0_idx = the per-token code, prob = rows of 32)
  float alpha, sigma, alpha_next, sigma_next;
  // MODE 2 (step prologue): Q = S . Wm^T + res[row(m)], then layer 0's value / sampling projections of Q
  const float* res;              // fp32 rows of 256 (the loop-invariant half of the concat-conv, bias included)
  int res_rn;                    // row(m) = res_rn ? (m / res_rn) * n_tok + m % n_tok : m   (r noisy maps share one x)
  int res_frag;                  // MODE 4 / 6 / 7, res_rn == 0: res is fp32 FRAGMENT-major like Q (one coalesced 1-KiB load per (t, g) instead of
                                 // 32 rows x 32 B per instruction); written that way by MODE 7
  // MODE 4 (seg tail of step s fused with the head of step s + 1) / MODE 2 (u out): the noisy map only ever enters the
  // loop through u_t = W_m . m_t, and the DDIM update is affine in (m_t, x0) with x0 one of K + 1 table rows, so
  //   u_{t+1} = ua . u_t + uc . (W_m . LUT)[argmax],   ua = sigma' / max(sigma, 1e-8),  uc = alpha' - alpha . ua
  // replaces both the update of the 256-channel map and the W_m GEMM of every step but the first (ddp.py:223-239)
  float* ubuf;                   // u, fp32 fragment-major rows of 256 (gemm_f32.h layout): written by MODE 2, read + written by MODE 4
  const float* tlut;             // (num_classes + 1, 256): W_m . LUT^T
  float ua, uc;
  // MODE 3 (layer 0's projections for the tasks without a fused step head): q comes as SB (Q, res == nullptr: bev after
  // its grid resampling, ddp_head_forward) or is formed here as the depth concat-conv, whose noisy-map half has ONE input
  // channel: q = res[row(m)] + wm * dvec[m]  (depth/depth/models/depther/ddp.py:236-237; wm travels as `bo`)
  const float* dvec;             // (M) noisy depth map
  // MODE 3, depth, steps >= 1: the DDIM update of the PREVIOUS step runs here, in front of the head (k_depth_update's arithmetic,
  // depth/depth/models/depther/ddp.py:238-246): d' from the nine taps the previous step's last layer left (MODE 9) - no launch
  // between the steps.  dtaps == nullptr: dvec is read as it is.
  const float* dtaps;            // (M, 32) per-tap dot products of conv_depth (column dy*3+dx)
  const float* dbias;            // conv_depth's bias (1)
  float* dvec_rw;                // = dvec, written
  float d_min, d_max, d_bit, d_eps, d_sig, d_alpha, d_alpha_next, d_sigma_next;
  int d_scale_up;
  // MODE 7 (first step's head from NCHW): S = nullptr; noise / x planes (maps of n_tok tokens, 256 channels each; one noisy map
  // per image), `bo` = the concat-conv's bias, res = xproj OUT (fp32 rows of 256), ubuf = u_0 out
  const float* nchw_noise;
  const float* nchw_x;
  // MODE 6 (last layer + seg tail): conv_seg's bias, zero padded to 256 floats (bias_ext holds fc1's bias and, at [1024, 1280),
  // layer 0's value_proj bias); has_next = "a next STEP follows" (MODE 4's part: u update, next q, layer 0's projections)
  const float* seg_bias;
  // MODE 5 (stream GEMM of the necks): up to four problems in ONE persistent launch, tiles of 128 tokens numbered through all
  // of them (gp[i].tile0 = first tile of problem i; unused problems: tile0 = INT_MAX): out (fp32 fragment-major, 256 outputs)
  // = act(A . W^T), A fp32 fragment-major with 32 * ns channels (split into bf16 pieces in the filler slots, as P0 does with
  // the attention output), the weights as ns "wide" stage images of the problem's own stream.  conv_h > 0: A is a 256-channel
  // map of images of conv_h x conv_w tokens and stage st = (tap st / 8, channel block st % 8) of a 3x3 convolution with zero
  // padding and dilation conv_dil (ns = 72).  The ring's look-ahead follows the block's tile order across problems.
  struct GemmProblem {
    const float* A;
    float* out;
    const unsigned char* stream;
    const float* bias;       // optional (256): added before the activation
    double* gn_partial;      // optional: GroupNorm(32 groups of 8 channels) partial sums of the result, [image][token / 32][group]
                             // {sum, sum of squares} - one entry per wave; needs gn_N % 32 == 0 (a wave's 32 tokens in one image)
    int M, ns, tile0, conv_h, conv_w, gn_N;
    int nchw_N;              // > 0: A is the caller's NCHW tensor (images of nchw_N tokens, 32 * ns channel planes): the fragments
                             // are read from the planes as MODE 7 reads x (16 coalesced dword loads per stage), no layout conversion
  } gp[4];
  int g_tiles, g_act, conv_dil;
  unsigned long long* stamps;    // -DDDP_LYR_STAMP builds only (scripts/stamp_layer.py): per (block, wave) cycle sums of the phases
};

// vmcnt(12): everything but the 12 newest vector-memory ops (= the DMA pieces of the stage just issued) is done
__device__ __forceinline__ void wait_vm12() { __builtin_amdgcn_s_waitcnt(0x0F7C); }

// element u (0..7) of a packed bf16x8 fragment as fp32
__device__ __forceinline__ float bf_elem(const u32x4& v, int u) {
  const unsigned w = v[u >> 1];
  return __uint_as_float((u & 1) ? (w & 0xFFFF0000u) : (w << 16));
}

// gelu_fast (gemm_f32.h: erf by Abramowitz-Stegun 7.1.26) + the exact 3-way bf16 split of its result as a list of
// 18 single instructions, so that a caller can hand out "ops [LO, HI)" to the filler slot behind each MFMA: with one
// wave per SIMD ~5 instructions hide behind an MFMA, the 6th costs ~4 cycles, the 8th ~25
// (scripts/ubench/mfma_fill.hip).  Constants are folded so that exp2 and the final sign need no extra instruction:
// z' = |x| sqrt(log2 e / 2);  t = 1 / (1 + p z), p z = (p / sqrt(log2 e)) z';  exp(-z^2) = exp2(-z'^2);
// gelu = hf + |hf| erf(|x|/sqrt 2), hf = x/2.
struct GeluState {
  float x, a, b, c;      // after op 17 the three bf16 pieces live in (b, c, a): gelu_ph / gelu_pm / gelu_pl
};
constexpr int GELU_OPS = 18;
template <int OP>
__device__ __forceinline__ void gelu_op(GeluState& g, float v) {
  if (OP == 0) g.x = v;
  else if (OP == 1) g.a = fabsf(g.x) * 0.84932180028801904272f;                  // z' = |x| sqrt(log2(e)/2)
  else if (OP == 2) g.b = fmaf(0.27274550239055780f, g.a, 1.0f);                 // 1 + 0.3275911 z, z = z' / sqrt(log2 e)
  else if (OP == 3) g.c = g.a * g.a;
  else if (OP == 4) g.b = __builtin_amdgcn_rcpf(g.b);                            // t
  else if (OP == 5) g.a = fmaf(1.061405429f, g.b, -1.453152027f);                // p
  else if (OP == 6) g.c = __builtin_amdgcn_exp2f(-g.c);                          // exp(-z^2)
  else if (OP == 7) g.a = fmaf(g.a, g.b, 1.421413741f);
  else if (OP == 8) g.a = fmaf(g.a, g.b, -0.284496736f);
  else if (OP == 9) g.a = fmaf(g.a, g.b, 0.254829592f);
  else if (OP == 10) g.a = g.a * g.b;
  else if (OP == 11) g.a = fmaf(-g.a, g.c, 1.0f);                                // erf(|x| / sqrt 2)
  else if (OP == 12) g.b = 0.5f * g.x;
  else if (OP == 13) g.a = fmaf(g.a, fabsf(g.b), g.b);                           // gelu(x)
  else if (OP == 14) g.b = __uint_as_float(__float_as_uint(g.a) & 0xFFFF0000u);  // piece 1
  else if (OP == 15) g.a = g.a - g.b;
  else if (OP == 16) g.c = __uint_as_float(__float_as_uint(g.a) & 0xFFFF0000u);  // piece 2
  else if (OP == 17) g.a = g.a - g.c;                                            // piece 3 (its high half)
}
__device__ __forceinline__ unsigned gelu_ph(const GeluState& g) { return __float_as_uint(g.b); }
__device__ __forceinline__ unsigned gelu_pm(const GeluState& g) { return __float_as_uint(g.c); }
__device__ __forceinline__ unsigned gelu_pl(const GeluState& g) { return __float_as_uint(g.a); }
template <int OP>
__device__ __forceinline__ void gelu_maybe(GeluState& g, float v) {
  if constexpr (OP >= 0) gelu_op<OP>(g, v);
}
// Schedule of FOUR values over the 24 filler slots of two consecutive MFMA blocks: GELU_SCHED[slot][value] = the
// instruction of that value issued in that slot (-1: none).  At most one instruction per value per slot (four
// independent dependency chains: a dependent VALU instruction behind an MFMA costs ~8 cycles, an independent one 4),
// at most 5 issue units per slot counting the slot's ds_read (1), its DMA piece (2) and v_rcp / v_exp double;
// slots 21..23 carry the six packs.  Built by a greedy list scheduler.
__device__ constexpr int GELU_SCHED[24][4] = {
    { 0,  0,  0,  0},
    { 1,  1,  1,  1},
    { 2,  2,  2,  2},
    {-1,  3,  3,  3},
    { 3,  4, -1, -1},
    { 4, -1, -1,  4},
    { 5,  5,  4,  5},
    {-1,  6,  5,  6},
    { 6,  7, -1, -1},
    { 7,  8,  6,  7},
    { 8,  9,  7,  8},
    { 9, 10,  8,  9},
    {10, 11,  9, 10},
    {11, 12, 10, 11},
    {12, 13, 11, 12},
    {13, -1, 12, 13},
    {14, 14, 13, 14},
    {15, 15, 14, 15},
    {16, 16, 15, 16},
    {17, 17, 16, 17},
    {-1, -1, 17, -1},
    {-1, -1, -1, -1},
    {-1, -1, -1, -1},
    {-1, -1, -1, -1}};

// The same GELU + split for the eight values of a k-block that has NOTHING to hide behind (k-block 0 of every FFN chunk
// waits for fc1's last MFMA): two values per instruction on the packed-fp32 path.  Packed fp32 shares the matrix pipe
// (scripts/ubench/mfma_fill.hip: +19 cycles behind an MFMA), which is idle exactly here.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_split8_packed(const float (&x)[8], u32x4& p1, u32x4& p2, u32x4& p3) {
  unsigned hh[8], mm[8], ll[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x2 xx = {x[2 * i], x[2 * i + 1]};
    const f32x2 ax = {fabsf(xx[0]), fabsf(xx[1])};
    const f32x2 a = ax * 0.84932180028801904272f;
    f32x2 b = __builtin_elementwise_fma(a, f32x2{0.27274550239055780f, 0.27274550239055780f}, f32x2{1.0f, 1.0f});
    f32x2 c = a * a;
    b = f32x2{__builtin_amdgcn_rcpf(b[0]), __builtin_amdgcn_rcpf(b[1])};
    f32x2 p = __builtin_elementwise_fma(b, f32x2{1.061405429f, 1.061405429f}, f32x2{-1.453152027f, -1.453152027f});
    c = f32x2{__builtin_amdgcn_exp2f(-c[0]), __builtin_amdgcn_exp2f(-c[1])};
    p = __builtin_elementwise_fma(p, b, f32x2{1.421413741f, 1.421413741f});
    p = __builtin_elementwise_fma(p, b, f32x2{-0.284496736f, -0.284496736f});
    p = __builtin_elementwise_fma(p, b, f32x2{0.254829592f, 0.254829592f});
    p = p * b;
    p = __builtin_elementwise_fma(-p, c, f32x2{1.0f, 1.0f});              // erf(|x| / sqrt 2)
    const f32x2 hf = xx * 0.5f, ahf = ax * 0.5f;
    const f32x2 g = __builtin_elementwise_fma(p, ahf, hf);                // gelu(x)
    const f32x2 gh = {__uint_as_float(__float_as_uint(g[0]) & 0xFFFF0000u), __uint_as_float(__float_as_uint(g[1]) & 0xFFFF0000u)};
    const f32x2 r = g - gh;
    const f32x2 rm = {__uint_as_float(__float_as_uint(r[0]) & 0xFFFF0000u), __uint_as_float(__float_as_uint(r[1]) & 0xFFFF0000u)};
    const f32x2 r2 = r - rm;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      hh[2 * i + e] = __float_as_uint(gh[e]);
      mm[2 * i + e] = __float_as_uint(rm[e]);
      ll[2 * i + e] = __float_as_uint(r2[e]);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    p1[i] = __builtin_amdgcn_perm(hh[2 * i + 1], hh[2 * i], 0x07060302);
    p2[i] = __builtin_amdgcn_perm(mm[2 * i + 1], mm[2 * i], 0x07060302);
    p3[i] = __builtin_amdgcn_perm(ll[2 * i + 1], ll[2 * i], 0x07060302);
  }
}

// split8 (gemm_bf16x3.h) with the two subtractions of each element pair as one packed instruction - for the phases of
// the kernel that run without MFMAs
__device__ __forceinline__ void split8_packed(const float (&x)[8], u32x4& p1, u32x4& p2, u32x4& p3) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x2 g = {x[2 * i], x[2 * i + 1]};
    const f32x2 gh = {__uint_as_float(__float_as_uint(g[0]) & 0xFFFF0000u), __uint_as_float(__float_as_uint(g[1]) & 0xFFFF0000u)};
    const f32x2 r = g - gh;
    const f32x2 rm = {__uint_as_float(__float_as_uint(r[0]) & 0xFFFF0000u), __uint_as_float(__float_as_uint(r[1]) & 0xFFFF0000u)};
    const f32x2 r2 = r - rm;
    p1[i] = __builtin_amdgcn_perm(__float_as_uint(gh[1]), __float_as_uint(gh[0]), 0x07060302);
    p2[i] = __builtin_amdgcn_perm(__float_as_uint(rm[1]), __float_as_uint(rm[0]), 0x07060302);
    p3[i] = __builtin_amdgcn_perm(__float_as_uint(r2[1]), __float_as_uint(r2[0]), 0x07060302);
  }
}

// one LDS-DMA piece of the stream: 64 lanes x 16 B, global (base + voff + IMM) -> LDS (m0base + M0ADD + 16*lane + IMM).
// Three instructions: everything else (stage base, ring-slot base) is computed once per stage - an MFMA hides at most
// ~5 other instructions behind it when the SIMD runs a single wave (scripts/ubench/mfma_fill.hip), so a piece must
// not bring its own address arithmetic.
template <int IMM, int M0ADD>
__device__ __forceinline__ void stream_piece(unsigned long long base, unsigned voff, unsigned m0base) {
  asm volatile(
      "s_add_u32 m0, %2, %4\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %1 offset:%3"
      :
      : "v"(voff), "s"(base), "s"(m0base), "n"(IMM), "n"(M0ADD)
      : "memory", "scc");
}

// The exact 3-way bf16 split of the eight values of one K16 block (split8) as a list of 44 single non-packed instructions
// (packed fp32 shares the matrix pipe and does not hide behind an MFMA), so that P0 can hand them out to its filler slots:
// pair i = ops 11 i .. 11 i + 10:  h0 h1 | r0 r1 | m0 m1 | l0 l1 | pack p1[i] p2[i] p3[i].
struct SplitState {
  float h[8], r[8], m[8];
};
__device__ __forceinline__ void split_op(int op, SplitState& s, const f32x4& a, const f32x4& b, u32x4 (&p)[3]) {
  const int i = op / 11, o = op - 11 * i, e0 = 2 * i, e1 = 2 * i + 1;
  auto x = [&](int e) { return e < 4 ? a[e] : b[e - 4]; };
  switch (o) {
    case 0: s.h[e0] = __uint_as_float(__float_as_uint(x(e0)) & 0xFFFF0000u); break;
    case 1: s.h[e1] = __uint_as_float(__float_as_uint(x(e1)) & 0xFFFF0000u); break;
    case 2: s.r[e0] = x(e0) - s.h[e0]; break;
    case 3: s.r[e1] = x(e1) - s.h[e1]; break;
    case 4: s.m[e0] = __uint_as_float(__float_as_uint(s.r[e0]) & 0xFFFF0000u); break;
    case 5: s.m[e1] = __uint_as_float(__float_as_uint(s.r[e1]) & 0xFFFF0000u); break;
    case 6: s.r[e0] = s.r[e0] - s.m[e0]; break;
    case 7: s.r[e1] = s.r[e1] - s.m[e1]; break;
    case 8: p[0][i] = __builtin_amdgcn_perm(__float_as_uint(s.h[e1]), __float_as_uint(s.h[e0]), 0x07060302); break;
    case 9: p[1][i] = __builtin_amdgcn_perm(__float_as_uint(s.m[e1]), __float_as_uint(s.m[e0]), 0x07060302); break;
    default: p[2][i] = __builtin_amdgcn_perm(__float_as_uint(s.r[e1]), __float_as_uint(s.r[e0]), 0x07060302); break;
  }
}
// ops [SPLIT_HAND_LO[k], SPLIT_HAND_LO[k + 1]) of a K16 block go to filler slot k of ONE MFMA block (the hand-over block of a
// P0 stage): 3 where the slot already carries a ds_read, 4 - 5 elsewhere
__device__ constexpr int SPLIT_HAND_LO[13] = {0, 3, 6, 10, 14, 17, 20, 24, 29, 33, 38, 41, 44};

// MODE 0: the decoder layer (output_proj + LN0, FFN + LN1 + FiLM, next layer's projections).
// MODE 1: the tail of a segmentation step on the same machinery: scores = conv_seg(q) as NCH chunks of 64 classes,
//         then - per token, in registers - argmax, softmax accumulation, x0 = LUT[argmax], DDIM update of the noisy
//         map (read and written as SB: the next step's concat-conv operand).  Replaces conv_seg GEMM + k_seg_update +
//         k_row_to_sb (segmentors/ddp.py:235-245; decode_head.py:133).
// MODE 2: the head of a step: q = W_m . m_t + (W_x x + b) (the noisy-map half of the concat-conv, ddp.py:223-224;
//         8 wide stages + the loop-invariant fp32 rows), q -> SB, then layer 0's value / sampling projections from the
//         q fragments still in registers (P3).  Replaces the FEAT / VALUE / SAMP GEMM launches of every step.
// Cycle accounting (debug builds, -DDDP_LYR_STAMP): s_memtime at the phase boundaries of the MODE 0 tile loop, deltas summed
// per phase in SGPRs (uniform values, constant indices) and added to a buffer once per wave at kernel exit.  Not compiled
// into the product library.
#ifdef DDP_LYR_STAMP
constexpr int LYR_NSTAMP = 16;
#define DDP_LYR_STAMP_DECL unsigned long long st_prev = __builtin_readcyclecounter(), st_acc[LYR_NSTAMP] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; \
  const unsigned long long st_t0 = __builtin_amdgcn_s_memrealtime();
#define DDP_LYR_STAMP_AT(i)                                   \
  {                                                           \
    const unsigned long long st_now = __builtin_readcyclecounter(); \
    st_acc[i] += st_now - st_prev;                            \
    st_prev = st_now;                                         \
  }
#else
#define DDP_LYR_STAMP_DECL
#define DDP_LYR_STAMP_AT(i)
#endif

// NT: the activation streams of the kernel that are touched once per launch (attention output, residual rows, q' / v' /
// sample-table stores) carry the non-temporal hint.  Same-box A/B at C2 (profiles/r03n_ab_nontemporal_streams.txt): the
// layer kernel itself does not care (1.490 vs 1.495 ms), but the gather that follows gets 3 % faster (0.1579 -> 0.1533 ms: v'
// and the sample table no longer sit in L2 as dirty lines to be written back while the gather streams them in again), +0.7 %
// on the batch.  At B = 1 (32 768 tokens) it is the other way round - v' (33 MB) and the table stay L2 / MALL resident between
// the two kernels, the hint throws that away (gather 0.0208 -> 0.0252 ms, -3 % on the image) - hence a template parameter
// chosen per launch by the token count (launch_b3_layer), not a build switch.  (The same hint on the GATHER's own table loads
// and result stores costs it 1 %: not used.)
// (round 5, same-box A/B profiles/r05g_ab_samp_hint_m7_pointer.txt: the sample table's stores WITHOUT the hint while v' keeps it -
// could the 100 MB table stay in the 256 MB memory-side cache until the gather's first loads ask for it? - changes nothing: gather
// 0.1538 vs 0.1535 ms)
constexpr int LYR_NT_MIN_TOKENS = 131072;                    // >= 128 MiB per fp32 activation tensor: beyond L2 + MALL reuse
template <bool NT>
__device__ __forceinline__ f32x4 ld_stream(const float* p) {
  if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  else return *reinterpret_cast<const f32x4*>(p);
}
template <bool NT>
__device__ __forceinline__ void st_stream(float* p, const f32x4& v) {
  if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
  else *reinterpret_cast<f32x4*>(p) = v;
}

template <int TAG, int MODE = 0, int NCH = 0, bool NT = false, bool FORCE = false>
__global__ void __launch_bounds__(LYR_THREADS, 1)
k_layer(LayerArgs la) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  // Everything derived from the lane index is RE-DERIVED at the start of every phase (refresh()) from a fresh, opaque lane
  // id instead of living in registers across the whole kernel.  The values are identical - functionally a no-op - but
  // the register allocator no longer spills them around the phases that do not need them (0 B of scratch instead of 76):
  // a scratch reload is vector memory, completes in order behind the weight-stream DMA and every earlier load, and so
  // costs a full memory round trip of idle matrix pipe at one wave per SIMD.
  int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int j = lane & 31;
  int h = lane >> 5;
  const int M = la.M;
  constexpr bool LT = MODE == 6 || MODE == 8 || MODE == 9;      // a whole decoder layer followed by a step tail
  constexpr bool LYR = MODE == 0 || LT || MODE == 10;            // the decoder layer's phases P0 .. LayerNorm1
  static_assert(!(MODE == 8 || MODE == 9) || NCH == 1, "the bev / depth tails are one 64-row chunk");
  const int ntiles = MODE == 5 ? la.g_tiles : (M + LYR_BM - 1) / LYR_BM;
  if (int(blockIdx.x) >= ntiles) return;
  const unsigned lds0 = (unsigned)(size_t)(lds_float_t*)smem;
  unsigned voff0 = unsigned(lane * 16), voff1 = voff0 + 4096, voff2 = voff0 + 8192;   // piece groups of 4 KiB
  const unsigned wave_off = unsigned(wave * 12 * 1024);        // this wave's 12 pieces of every stage
  auto gp_of = [&](int tile) __attribute__((always_inline)) {          // MODE 5: problem of a tile (uniform)
    return (tile >= la.gp[1].tile0 ? 1 : 0) + (tile >= la.gp[2].tile0 ? 1 : 0) + (tile >= la.gp[3].tile0 ? 1 : 0);
  };
  int fs_tile = blockIdx.x;                                    // MODE 5: the tile whose stages are being fetched
  int n_stages = MODE == 5 ? la.gp[gp_of(int(blockIdx.x))].ns
                       : MODE == 1 ? 2 * NCH : MODE == 4 ? 2 * NCH + LYR_ST_NEXT : MODE == 3 ? LYR_ST_NEXT : MODE == 2 ? LYR_ST_OUT + LYR_ST_NEXT
                       : LT ? LYR_ST_OUT + LYR_ST_FFN + 2 * NCH + (la.has_next ? LYR_ST_NEXT : 0)
                       : MODE == 7 ? 2 * LYR_ST_OUT + LYR_ST_NEXT
                                                                                    : (la.has_next ? LYR_STAGES : LYR_ST_OUT + LYR_ST_FFN);
  int sidx = 0;                                                // stage image to fetch next
  unsigned long long nb = 0;                                   // its base (+ this wave's share), set by stage_begin
  unsigned mb = 0;                                             // ring-slot LDS base (+ this wave's share)
  auto nxt = [](int s) { return s == LYR_RING - 1 ? 0 : s + 1; };

  // once per stage: where the look-ahead comes from / goes to (kept in SGPRs), then advance the stream index
  const unsigned char* fs_stream = MODE == 5 ? la.gp[gp_of(int(blockIdx.x))].stream : la.stream;
  auto stage_begin = [&](int dslot) __attribute__((always_inline)) {
    const unsigned long long p = reinterpret_cast<unsigned long long>(fs_stream) + size_t(sidx) * LYR_STAGE_B + wave_off;
    const unsigned lo = __builtin_amdgcn_readfirstlane(unsigned(p)), hi = __builtin_amdgcn_readfirstlane(unsigned(p >> 32));
    nb = (static_cast<unsigned long long>(hi) << 32) | lo;
    mb = __builtin_amdgcn_readfirstlane(lds0 + unsigned(dslot * LYR_STAGE_B) + wave_off);
    if (sidx + 1 == n_stages) {
      sidx = 0;
      if constexpr (MODE == 5) {         // the stream continues with the block's NEXT tile, possibly of another problem
        fs_tile += int(gridDim.x);
        const int pi = gp_of(fs_tile);   // (past the last tile: some problem's stream, fetched and never used)
        fs_stream = la.gp[pi].stream;
        n_stages = la.gp[pi].ns;
      }
    } else {
      sidx = sidx + 1;
    }
  };
  // pieces [i0, i1) of that stage image
  auto dma = [&](int i0, int i1) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      if (i < i0 || i >= i1) continue;
      switch (i) {
        case 0: stream_piece<0, 0>(nb, voff0, mb); break;
        case 1: stream_piece<1024, 0>(nb, voff0, mb); break;
        case 2: stream_piece<2048, 0>(nb, voff0, mb); break;
        case 3: stream_piece<3072, 0>(nb, voff0, mb); break;
        case 4: stream_piece<0, 4096>(nb, voff1, mb); break;
        case 5: stream_piece<1024, 4096>(nb, voff1, mb); break;
        case 6: stream_piece<2048, 4096>(nb, voff1, mb); break;
        case 7: stream_piece<3072, 4096>(nb, voff1, mb); break;
        case 8: stream_piece<0, 8192>(nb, voff2, mb); break;
        case 9: stream_piece<1024, 8192>(nb, voff2, mb); break;
        case 10: stream_piece<2048, 8192>(nb, voff2, mb); break;
        default: stream_piece<3072, 8192>(nb, voff2, mb); break;
      }
    }
  };

  // The look-ahead stage's 12 DMA pieces, handed to filler slot k of MFMA block b (0..7) of the running stage:
  //   spread  one piece in every block and a second one in blocks 0..3 - the FFN, whose fillers are full of GELU;
  //   front   all twelve in blocks 0..3 - every stage of a phase that has vector-memory operations of its own (P0: S and
  //           residual fragments; P3: value / table stores, positional loads; conv_seg of the tails).  hipcc's s_waitcnt
  //           insertion does not see the inline-asm DMA: for a load it does see it counts only its own younger operations and
  //           emits vmcnt(0..3) in front of the first use - in hardware terms "drain every DMA piece issued since".  With the
  //           pieces spread over the stage that drained the piece requested ~300 cycles earlier: a forced round trip at the end
  //           of every P0 stage, in front of P1 and behind every sampling chunk (r03: the disassembly shows vmcnt(3)/(2)/(1)/(0)
  //           next to the hand-placed vmcnt(11)).  Front-loaded, the youngest piece is >= 4 MFMA blocks (~1600 cycles) old at
  //           any such wait.  (Issuing these loads by inline asm instead - invisible like the DMA - is NOT an option: the
  //           register allocator treats the destination as written at the asm statement and may copy or reuse it before the
  //           data lands; r03b faulted exactly that way.)
  auto dma_slot = [&](bool front, int b, int k) __attribute__((always_inline)) {
    if (front) {
      if (b < 4) {
        if (k == 3) dma(3 * b, 3 * b + 1);
        if (k == 6) dma(3 * b + 1, 3 * b + 2);
        if (k == 8) dma(3 * b + 2, 3 * b + 3);
      }
    } else {
      if (k == 3) dma(b, b + 1);
      if (k == 8 && b < 4) dma(8 + b, 9 + b);
    }
  };
  // hand-over wait in front of a stage's last block: everything but the pieces of the stage being requested has landed
  // (spread: 11 of them issued by then; front: all 12)
  auto wait_hand = [&](bool front) __attribute__((always_inline)) {
    if (front) __builtin_amdgcn_s_waitcnt(0x0F7C);             // vmcnt(12)
    else __builtin_amdgcn_s_waitcnt(0x0F7B);                   // vmcnt(11)
  };

  // fragment reads (per-lane base + slot base + compile-time offsets)
  const char* lbase = reinterpret_cast<const char*>(smem);
  int f1_lane = j * 256;            // "tall" stage [64 rows x 128 k]: row t*32+j, 16 slots of 16 B, swizzle (row & 15)
  int f2_lane = j * 64;             // "wide" stage [256 rows x 32 k]: row t*32+j, 4 slots, swizzle ((row>>2) & 3)
  auto frag1 = [&](int slot, int comp, int t, int b) -> u32x4 {      // K16 step b (0..7)
    return *reinterpret_cast<const u32x4*>(lbase + slot * LYR_STAGE_B + comp * 16384 + t * 8192 + f1_lane + (((2 * b + h) ^ (j & 15)) << 4));
  };
  auto frag2 = [&](int slot, int comp, int t, int ks) -> u32x4 {     // K16 step ks (0..1)
    return *reinterpret_cast<const u32x4*>(lbase + slot * LYR_STAGE_B + comp * 16384 + t * 2048 + f2_lane + (((2 * ks + h) ^ ((j >> 2) & 3)) << 4));
  };

  const float* bias_s = reinterpret_cast<const float*>(reinterpret_cast<const char*>(smem) + LYR_BIAS_OFF);
  int ht = h;
  // (the lane id is produced by a volatile asm: the builtin pair is pure, so hipcc computes it once at kernel entry,
  // spills THAT and reloads it here - 16 B of scratch and four in-order reloads per tile; the table base and the lane's
  // table index are re-derived with it for the same reason)
  auto refresh = [&]() __attribute__((always_inline)) {
    unsigned l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    lane = int(l);
    j = lane & 31;
    h = lane >> 5;
    ht = h;
    voff0 = unsigned(lane * 16);
    voff1 = voff0 + 4096;
    voff2 = voff0 + 8192;
    f1_lane = j * 256;
    f2_lane = j * 64;
    unsigned tab_off = LYR_BIAS_OFF;
    asm volatile("" : "+v"(tab_off));
    bias_s = reinterpret_cast<const float*>(reinterpret_cast<const char*>(smem) + tab_off);
  };

  // One block = 12 MFMAs on a tile pair (two independent accumulator chains, 6 split products each) with a filler
  // slot behind EVERY MFMA: F(k) is whatever may issue in the shadow of MFMA k (k = 0..11), kept to <= 4-5
  // instructions by construction (the scheduler cannot move code across the barriers).  Slots 0,1 / 4,5 / 10,11
  // re-read w2 / w1 / w0 of the next block right after their last use.
#define DDP_LYR_SB __builtin_amdgcn_sched_barrier(0);
#define DDP_LYR_K(k) std::integral_constant<int, k>{}
#define DDP_LYR_BLOCK2(A0, A1, X0, X1, X2, Y0, Y1, Y2, F)                                                            \
  A0 = mma(w[0][2], X0, A0); F(DDP_LYR_K(0));  DDP_LYR_SB                                                            \
  A1 = mma(w[1][2], Y0, A1); F(DDP_LYR_K(1));  DDP_LYR_SB                                                            \
  A0 = mma(w[0][1], X1, A0); F(DDP_LYR_K(2));  DDP_LYR_SB                                                            \
  A1 = mma(w[1][1], Y1, A1); F(DDP_LYR_K(3));  DDP_LYR_SB                                                            \
  A0 = mma(w[0][1], X0, A0); F(DDP_LYR_K(4));  DDP_LYR_SB                                                            \
  A1 = mma(w[1][1], Y0, A1); F(DDP_LYR_K(5));  DDP_LYR_SB                                                            \
  A0 = mma(w[0][0], X2, A0); F(DDP_LYR_K(6));  DDP_LYR_SB                                                            \
  A1 = mma(w[1][0], Y2, A1); F(DDP_LYR_K(7));  DDP_LYR_SB                                                            \
  A0 = mma(w[0][0], X1, A0); F(DDP_LYR_K(8));  DDP_LYR_SB                                                            \
  A1 = mma(w[1][0], Y1, A1); F(DDP_LYR_K(9));  DDP_LYR_SB                                                            \
  A0 = mma(w[0][0], X0, A0); F(DDP_LYR_K(10)); DDP_LYR_SB                                                            \
  A1 = mma(w[1][0], Y0, A1); F(DDP_LYR_K(11)); DDP_LYR_SB
#define DDP_LYR_BLOCK(A0, A1, X0, X1, X2, F) DDP_LYR_BLOCK2(A0, A1, X0, X1, X2, X0, X1, X2, F)

  // ---- kernel prologue: first two stages of the stream, bias table -> LDS
  int slot = 0;
  stage_begin(0);
  dma(0, 12);
  stage_begin(1);
  dma(0, 12);
  {
    float* tab = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + LYR_BIAS_OFF);
    if constexpr (MODE != 5)
      for (int i = tid; i < LYR_BIAS_N; i += LYR_THREADS) tab[i] = la.bias_ext[i];
    if constexpr (MODE == 3) tab[LYR_T_BO + tid] = la.bo ? la.bo[tid] : 0.f;      // depth: the concat-conv's depth column
    if constexpr (MODE == 7) tab[LYR_T_BO + tid] = la.bo ? la.bo[tid] : 0.f;      // the concat-conv's bias
    if constexpr (LT || MODE == 10) tab[LYR_T_SEG + tid] = la.seg_bias[tid];      // (MODE 10: w_m, the concat-conv's depth column)
    if constexpr (LYR) {
      tab[LYR_T_BO + tid] = la.bo[tid];
      tab[LYR_T_GA0 + tid] = la.ga0[tid];
      tab[LYR_T_BE0 + tid] = la.be0[tid];
      tab[LYR_T_B2 + tid] = la.b2[tid];
      tab[LYR_T_GA1 + tid] = la.ga1[tid];
      tab[LYR_T_BE1 + tid] = la.be1[tid];
    }
  }
  wait_vm0();
  __syncthreads();

  u32x4 xa[16][3];
  f32x16 acc2[8];

  // acc (+)= W[64 rows x 256 k] . xa over two "tall" stages; the ring's look-ahead is fetched on the way.
  // s1 = 2: ONE "split-K" stage of 32 output rows - image rows 0..31 hold k 0..127, rows 32..63 the SAME outputs' k 128..255 -
  // so chain a0 runs k-blocks 0..7 and chain a1 k-blocks 8..15 (the caller adds them): no zero rows are multiplied
  auto tall_stage = [&](f32x16& a0, f32x16& a1, auto s1c) __attribute__((always_inline)) {
    constexpr int s1 = decltype(s1c)::value;
    stage_begin(nxt(nxt(slot)));
    u32x4 w[2][3];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int c = 0; c < 3; ++c) w[t][c] = frag1(slot, c, t, 0);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool nb = b + 1 < 8;
      const int kb = s1 == 2 ? b : s1 * 8 + b;
      const int kb1 = s1 == 2 ? 8 + b : kb;
      auto fill = [&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        if (k == 0 && nb) w[0][2] = frag1(slot, 2, 0, b + 1);
        if (k == 1 && nb) w[1][2] = frag1(slot, 2, 1, b + 1);
        if (k == 4 && nb) w[0][1] = frag1(slot, 1, 0, b + 1);
        if (k == 5 && nb) w[1][1] = frag1(slot, 1, 1, b + 1);
        if (k == 10 && nb) w[0][0] = frag1(slot, 0, 0, b + 1);
        if (k == 11 && nb) w[1][0] = frag1(slot, 0, 1, b + 1);
        dma_slot(true, b, k);                                  // (only P3's split-K sampling stage runs here)
      };
      DDP_LYR_BLOCK2(a0, a1, xa[kb][0], xa[kb][1], xa[kb][2], xa[kb1][0], xa[kb1][1], xa[kb1][2], fill)
    }
    wait_vm12();
    __syncthreads();
    slot = nxt(slot);
  };
  const std::integral_constant<int, 0> I0{};
  const std::integral_constant<int, 1> I1{};
  const std::integral_constant<int, 2> I2{};
  // Both tall stages of a 64-row chunk with an EARLY hand-over between them (+0.8 % end to end, same-box A/B r02a).  The
  // classic stage ends with "wait for stage q+1's DMA; barrier" and the next one starts with six exposed ds_reads.  Here
  // the wait + barrier sit in front of the LAST block of stage q (11 of stage q+2's 12 pieces are in flight then:
  // vmcnt(11)); nothing reads stage q's ring slot after that barrier, because the last block's fragment re-reads fetch
  // block 0 of stage q+1 instead - so the slot may be overwritten by stage q+3's DMA as before, and stage q+1 starts with
  // its operands already in registers.
  auto tall_pair = [&](f32x16& a0, f32x16& a1, auto frontc) __attribute__((always_inline)) {
    constexpr bool front = decltype(frontc)::value != 0;
    u32x4 w[2][3];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int c = 0; c < 3; ++c) w[t][c] = frag1(slot, c, t, 0);
#pragma unroll
    for (int s1 = 0; s1 < 2; ++s1) {
      stage_begin(nxt(nxt(slot)));
      const int nslot = nxt(slot);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const bool nb = b + 1 < 8;
        const bool hand = s1 == 0 && b == 7;                    // fetch block 0 of the second stage
        const int rs = hand ? nslot : slot, rb = hand ? 0 : b + 1;
        const int kb = s1 * 8 + b;
        if (hand) {
          wait_hand(front);
          __syncthreads();
        }
        auto fill = [&](auto kc) __attribute__((always_inline)) {
          constexpr int k = decltype(kc)::value;
          if (k == 0 && (nb || hand)) w[0][2] = frag1(rs, 2, 0, rb);
          if (k == 1 && (nb || hand)) w[1][2] = frag1(rs, 2, 1, rb);
          if (k == 4 && (nb || hand)) w[0][1] = frag1(rs, 1, 0, rb);
          if (k == 5 && (nb || hand)) w[1][1] = frag1(rs, 1, 1, rb);
          if (k == 10 && (nb || hand)) w[0][0] = frag1(rs, 0, 0, rb);
          if (k == 11 && (nb || hand)) w[1][0] = frag1(rs, 0, 1, rb);
          dma_slot(front, b, k);
        };
        DDP_LYR_BLOCK(a0, a1, xa[kb][0], xa[kb][1], xa[kb][2], fill)
      }
      if (s1 == 1) {
        wait_vm12();
        __syncthreads();
      }
      slot = nxt(slot);
    }
  };
  auto bias_init = [&](f32x16 (&a)[2], int chunk) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(bias_s + chunk * 64 + t * 32 + 8 * g + 4 * ht);
#pragma unroll
        for (int e = 0; e < 4; ++e) a[t][4 * g + e] = b[e];
      }
  };

  // q (fp32 accumulator layout) -> the 16 x 3 B fragments of a contraction over its 256 channels: K16 block b = 2t + gp is
  // the quad pair (2gp, 2gp + 1) of tile t
  auto load_q_fragments = [&](const float* qp) __attribute__((always_inline)) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      f32x4 qv[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) qv[i][g] = ld_stream<NT>(qp + (half * 4 + i) * 1024 + g * 256);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          const int b = 2 * (half * 4 + i) + gp;
          float xv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) xv[e] = qv[i][2 * gp + (e >> 2)][e & 3];
          split8_packed(xv, xa[b][0], xa[b][1], xa[b][2]);
        }
    }
  };
  // eight values of K16 block b (as produced for split8) -> q in HBM
  auto store_q_block = [&](float* qp, int b, const float (&xv)[8]) __attribute__((always_inline)) {
    st_stream<NT>(qp + (b >> 1) * 1024 + (2 * (b & 1)) * 256, f32x4{xv[0], xv[1], xv[2], xv[3]});
    st_stream<NT>(qp + (b >> 1) * 1024 + (2 * (b & 1) + 1) * 256, f32x4{xv[4], xv[5], xv[6], xv[7]});
  };

  DDP_LYR_STAMP_DECL
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    DDP_LYR_STAMP_AT(9)                                        // tile turnaround (and the kernel prologue)
    const size_t grp = size_t(tile) * (LYR_BM / 32) + wave;    // this wave's 32-token group
    // opaque per tile: otherwise the (tile-invariant) reads of the per-channel table are hoisted out of the tile
    // loop - ~800 values per lane - and spilled to scratch
    refresh();
    const int m_base = tile * LYR_BM + wave * 32;
    const char* ss = reinterpret_cast<const char*>(la.S) + grp * 256 * 192 + lane * 16;
    float* qf = la.Q + grp * 8192 + lane * 4;                   // + t * 1024 + g * 256: this lane's accumulator quad (t, g)

    if constexpr (MODE == 5) {
      // ---- stream GEMM (FPN laterals / 3x3 output convolutions, MultiStageMerging's per-level 1x1 convs): P0's machinery with
      // a run-time number of wide stages.  The A fragments of a stage are four fp32 quads of this lane's token (for the 3x3
      // convolution: of the token shifted by the stage's tap; lanes whose source is outside the map take zeros), split into
      // the three bf16 pieces in the filler slots exactly as P0 splits the attention output.
      const int pi = gp_of(tile);
      const int ns = la.gp[pi].ns;
      const int lt = tile - la.gp[pi].tile0;                   // tile within its problem
      const int Mp = la.gp[pi].M;
      int ch = la.gp[pi].conv_h, cw = la.gp[pi].conv_w;
      asm volatile("" : "+s"(ch), "+s"(cw));                  // (opaque per tile: see P3's index arithmetic)
      const size_t grp5 = size_t(lt) * (LYR_BM / 32) + wave;
      const int m = lt * LYR_BM + wave * 32 + j;
      const bool mvalid = m < Mp;
      int ci = 0, cj = 0;
      if (ch > 0) {
        const int mm = mvalid ? m : Mp - 1;
        const int n = mm - (mm / (ch * cw)) * (ch * cw);
        ci = n / cw;
        cj = n - ci * cw;
      }
      const float* abase = la.gp[pi].A;
      const float* arow = abase + grp5 * size_t(ns) * 1024 + lane * 4;      // plain: this wave's group, 32 * ns channels
      // 3x3 view: the shifted token's address and its border verdict change with the TAP, i.e. every eighth stage - they are
      // carried across the eight channel blocks of a tap (round 4: re-deriving tap (a division by 3), shifted token, four
      // border comparisons and a 64-bit address in front of EVERY stage put ~60 scalar + 25 vector instructions before the
      // stage's first MFMA; at one wave per SIMD all of them exposed)
      const float* tap_ptr = abase;
      bool tap_ok = false;
      // NCHW source (FPN laterals straight from the backbone's levels): this lane's token in the planes, as in MODE 7
      int nN = la.gp[pi].nchw_N;
      asm volatile("" : "+s"(nN));
      unsigned nchw_off5 = 0;
      if (nN > 0) {
        const int mm = mvalid ? m : Mp - 1;
        const int img5 = mm / nN;
        nchw_off5 = (unsigned(img5) * unsigned(32 * ns) + unsigned(4 * h)) * unsigned(nN) + unsigned(mm - img5 * nN);   // (< 2^32: host guard)
      }
      auto a_load = [&](int st, f32x4 (&dst)[4]) __attribute__((always_inline)) {
        if (nN > 0) {                                          // (uniform)
          const float* p = abase + size_t(32 * st) * nN + nchw_off5;
          const size_t nstr = size_t(nN);
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              asm volatile("" : "+v"(p));
              dst[g][e] = *p;
              p += e == 3 ? 5 * nstr : nstr;
            }
          return;
        }
        const float* ap = arow + size_t(st) * 1024;
        bool ok = true;
        if (ch > 0) {
          if ((st & 7) == 0) {                                     // (uniform: a scalar branch)
            const int tap = st >> 3;
            const int dy = (tap / 3 - 1) * la.conv_dil, dx = (tap - (tap / 3) * 3 - 1) * la.conv_dil;
            tap_ok = mvalid && ci + dy >= 0 && ci + dy < ch && cj + dx >= 0 && cj + dx < cw;
            const int ms = tap_ok ? m + dy * cw + dx : 0;
            tap_ptr = abase + size_t(ms >> 5) * 8192 + (h * 32 + (ms & 31)) * 4;
          }
          ok = tap_ok;
          ap = tap_ptr + size_t(st & 7) * 1024;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (ok) v = *reinterpret_cast<const f32x4*>(ap + g * 256);
          dst[g] = v;
        }
      };
      u32x4 sc[2][3];
      f32x4 sf[4], sh[2];
      a_load(0, sf);
      {
        const float* bp = la.gp[pi].bias;
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4 b = {0.f, 0.f, 0.f, 0.f};
            if (bp) b = *reinterpret_cast<const f32x4*>(bp + t * 32 + 8 * g + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc2[t][4 * g + e] = b[e];
          }
      }
      u32x4 w[2][3];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c = 0; c < 3; ++c) w[t][c] = frag2(slot, c, t, 0);
      {
        float xv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[e] = sf[e >> 2][e & 3];
        split8_packed(xv, sc[0][0], sc[0][1], sc[0][2]);
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[e] = sf[2 + (e >> 2)][e & 3];
        split8_packed(xv, sc[1][0], sc[1][1], sc[1][2]);
      }
      auto g_stage = [&](int st, auto lastc, auto firstc) __attribute__((always_inline)) {
        constexpr bool last = decltype(lastc)::value != 0;
        constexpr bool first = decltype(firstc)::value != 0;
        stage_begin(nxt(nxt(slot)));
        const int nslot = nxt(slot);
        SplitState spa, spb;
        if constexpr (!last) a_load(st + 1, sf);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int tp = 0; tp < 4; ++tp) {
            const int blk = ks * 4 + tp;
            const bool hand = !last && blk == 7;
            const bool nb = blk + 1 < 8 || hand;
            const int rs = hand ? nslot : slot;
            const int tp2 = hand ? 0 : (tp + 1 < 4 ? tp + 1 : 0), ks2 = hand ? 0 : (tp + 1 < 4 ? ks : ks + 1);
            if (hand) {
              wait_hand(true);
              __syncthreads();
            }
            auto fill = [&](auto kc) __attribute__((always_inline)) {
              constexpr int k = decltype(kc)::value;
              if (k == 0 && nb) w[0][2] = frag2(rs, 2, 2 * tp2, ks2);
              if (k == 1 && nb) w[1][2] = frag2(rs, 2, 2 * tp2 + 1, ks2);
              if (k == 4 && nb) w[0][1] = frag2(rs, 1, 2 * tp2, ks2);
              if (k == 5 && nb) w[1][1] = frag2(rs, 1, 2 * tp2 + 1, ks2);
              if (k == 10 && nb) w[0][0] = frag2(rs, 0, 2 * tp2, ks2);
              if (k == 11 && nb) w[1][0] = frag2(rs, 0, 2 * tp2 + 1, ks2);
              dma_slot(true, blk, k);
              if (hand) {
#pragma unroll
                for (int op = SPLIT_HAND_LO[k]; op < SPLIT_HAND_LO[k + 1]; ++op) split_op(op, spa, sf[0], sf[1], sc[0]);
              }
              if (!first && blk < 4 && k < 11) split_op(blk * 11 + k, spb, sh[0], sh[1], sc[1]);
            };
            DDP_LYR_BLOCK(acc2[2 * tp], acc2[2 * tp + 1], sc[ks][0], sc[ks][1], sc[ks][2], fill)
          }
        if constexpr (last) {
          wait_vm12();
          __syncthreads();
        } else {
          sh[0] = sf[2];
          sh[1] = sf[3];
        }
        slot = nxt(slot);
      };
      g_stage(0, I0, I1);
      for (int st = 1; st + 1 < ns; ++st) g_stage(st, I0, I0);
      g_stage(ns - 1, I1, I0);
      // epilogue: activation (0 none, 2 ReLU), fp32 fragment-major rows out (1-KiB coalesced stores; padding rows of a
      // problem's last tile are written too - its buffer is padded to whole tiles - and never read)
      float* qo = la.gp[pi].out + grp5 * 8192 + lane * 4;
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v = {acc2[t][4 * g], acc2[t][4 * g + 1], acc2[t][4 * g + 2], acc2[t][4 * g + 3]};
          if (la.g_act == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          *reinterpret_cast<f32x4*>(qo + t * 1024 + g * 256) = v;
        }
      if (la.gp[pi].gn_partial) {
        // GroupNorm statistics of the tile, fused: group G = 4 t + g is this lane's quad (t, g) and its partner half's (lane ^ 32);
        // fp64 sums over the wave's 32 tokens by a butterfly, lane G keeps group G: one 16-B store per lane of the low half.
        const int m0w = lt * LYR_BM + wave * 32;               // the wave's first token (all 32 in one image: gn_N % 32 == 0)
        // value index k = 2 * group + {0: sum, 1: sum of squares}; a transposing butterfly over the 64 lanes: at distance d a
        // lane keeps the half of its values whose index has bit d set like its own lane id, sends the other half to lane ^ d
        // and adds what it receives - after six steps lane k holds the total of value k (63 exchanges instead of 64 x 6)
        double v[64];
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            double sv = 0.0, qv = 0.0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const double x = double(la.g_act == 2 ? fmaxf(acc2[t][4 * g + e], 0.f) : acc2[t][4 * g + e]);
              sv += x;
              qv += x * x;
            }
            v[2 * (4 * t + g)] = sv;
            v[2 * (4 * t + g) + 1] = qv;
          }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
          const bool up = (lane & d) != 0;
#pragma unroll
          for (int i = 0; i < d; ++i) {
            const double keep = up ? v[i + d] : v[i];
            const double send = up ? v[i] : v[i + d];
            v[i] = keep + __shfl_xor(send, d, 64);
          }
        }
        if (m0w < Mp) {
          const int nimg = la.gp[pi].gn_N;
          const int b = m0w / nimg, ck = (m0w - b * nimg) >> 5;
          la.gp[pi].gn_partial[(size_t(b) * (nimg >> 5) + ck) * 64 + lane] = v[0];
        }
      }
      continue;
    }
    if constexpr (MODE == 3) {
      if (la.res) {
        // depth step head: q = (W_x x + b)[row] + w_d * d  -> SB + fragments (no GEMM: the noisy map has one channel)
        int mr = m_base + j;
        mr = mr < M ? mr : M - 1;
        const size_t row = la.res_rn ? size_t(mr / la.res_rn) * la.n_tok + mr % la.n_tok : size_t(mr);
        const float* rp = la.res + row * 256 + 4 * h;
        float dv = la.dvec[mr];
        if (la.dtaps) {                                       // (uniform) the previous step's depth update, then this step's head
          int ntk3 = la.n_tok, wm3 = la.w;
          asm volatile("" : "+s"(ntk3), "+s"(wm3));           // (opaque per tile: see P3's index arithmetic)
          const int img3 = mr / ntk3, n3 = mr - img3 * ntk3;
          const int i3 = n3 / wm3, j3 = n3 - i3 * wm3, hm3 = ntk3 / wm3;
          const float* tb = la.dtaps + size_t(img3) * ntk3 * 32;
          // (branch-free: nine loads in flight, taps outside the map selected to zero; k_depth_update's sum in its order)
          float tv[9];
#pragma unroll
          for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
              const int ii = i3 + dy - 1, jj = j3 + dx - 1;
              const bool ok = ii >= 0 && ii < hm3 && jj >= 0 && jj < wm3;
              const float v = tb[size_t(min(max(ii, 0), hm3 - 1) * wm3 + min(max(jj, 0), wm3 - 1)) * 32 + dy * 3 + dx];
              tv[dy * 3 + dx] = ok ? v : 0.f;
            }
          float sacc = 0.f;
#pragma unroll
          for (int t = 0; t < 9; ++t) sacc += tv[t];
          sacc += la.dbias[0];
          const float pred = la.d_scale_up ? la.d_eps / (1.0f + expf(-sacc)) : fmaxf(sacc, 0.f) + la.d_eps;
          float x0 = (pred - la.d_min) / (la.d_max - la.d_min);
          x0 = (x0 * 2.0f - 1.0f) * la.d_bit;
          x0 = fminf(fmaxf(x0, -la.d_bit), la.d_bit);
          const float epsn = la.d_sig * (dv - la.d_alpha * x0);
          dv = la.d_alpha_next * x0 + la.d_sigma_next * epsn;
          if (h == 0 && m_base + j < M) la.dvec_rw[mr] = dv;
        }
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
          f32x4 xx[2][4];
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) xx[i][g] = *reinterpret_cast<const f32x4*>(rp + (qt * 2 + i) * 32 + 8 * g);
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int t = qt * 2 + i;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const f32x4 wd = *reinterpret_cast<const f32x4*>(bias_s + LYR_T_BO + t * 32 + 8 * g + 4 * ht);
#pragma unroll
              for (int e = 0; e < 4; ++e) xx[i][g][e] = xx[i][g][e] + wd[e] * dv;
            }
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
              const int b = 2 * t + gp;
              float xv[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) xv[e] = xx[i][2 * gp + (e >> 2)][e & 3];
              split8_packed(xv, xa[b][0], xa[b][1], xa[b][2]);
              store_q_block(qf, b, xv);
            }
          }
        }
      } else {
        load_q_fragments(qf);
      }
    }
    if constexpr (LYR || MODE == 2 || MODE == 7) {
    {
    // ---- P0: acc2 = bo + Wo . s  (8 "wide" stages; s fragments fetched one stage ahead); MODE 2: acc2 = Wm . mask
    u32x4 sc[2][3], sn[2][3];
    // MODE 0: the attention output arrives as fp32 fragments (la.Sf): per stage (32 channels = tile t) four quads
    constexpr bool SF = (LYR && (DDP_S_F32 != 0)) || MODE == 7;
    f32x4 sf[4], sh[2];
    const float* sfp = la.Sf + grp * 8192 + lane * 4;
    // MODE 7: this lane's token in the NCHW planes: element offset of (map, channel 4h, token n); stage st (0..7 noise, 8..15 x)
    // reads channels 32 (st & 7) + 8g + 4h + e - 16 dword loads, each a 128-B run per half wave
    unsigned nchw_off = 0;
    if constexpr (MODE == 7) {
      int ntk7 = la.n_tok;
      asm volatile("" : "+s"(ntk7));                           // (opaque per tile: see P3's index arithmetic)
      int m7 = m_base + j;
      m7 = m7 < M ? m7 : M - 1;
      const int img7 = m7 / ntk7;
      nchw_off = (unsigned(img7) * 256u + unsigned(4 * h)) * unsigned(ntk7) + unsigned(m7 - img7 * ntk7);   // (< 2^32: host guard)
    }
    auto sf_fetch = [&](int stn) __attribute__((always_inline)) {          // the fp32 operand fragments of stage stn -> sf[0..3]
      if constexpr (MODE == 7) {
        const float* plane = (stn < 8 ? la.nchw_noise : la.nchw_x) + size_t(32 * (stn & 7)) * la.n_tok;
        // ONE per-lane pointer walking the 16 channel planes (strides N, N, N, 5 N: channels 8g + e, the lane's half adds 4):
        // with 16 independent uniform plane addresses the compiler keeps 32 SGPRs of offsets alive across the kernel and spills
        // them to VGPR lanes (32 v_readlane in front of every stage's loads; no measurable difference in time, 173 fewer lane moves)
        const float* p = plane + nchw_off;
        const size_t nstr = size_t(la.n_tok);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            asm volatile("" : "+v"(p));
            if constexpr (NT) sf[g][e] = __builtin_nontemporal_load(p);
            else sf[g][e] = *p;
            p += e == 3 ? 5 * nstr : nstr;
          }
        }
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) sf[g] = ld_stream<NT>(sfp + stn * 1024 + g * 256);
      }
    };
    if constexpr (SF) {
      sf_fetch(0);
    } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int c = 0; c < 3; ++c) sc[ks][c] = *reinterpret_cast<const u32x4*>(ss + (ks * 3 + c) * 1024);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 b = {0.f, 0.f, 0.f, 0.f};
        if constexpr (LYR) b = *reinterpret_cast<const f32x4*>(bias_s + LYR_T_BO + t * 32 + 8 * g + 4 * ht);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc2[t][4 * g + e] = b[e];
      }
    // (every stage but the last hands over early, see tall_pair)
    u32x4 w[2][3];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int c = 0; c < 3; ++c) w[t][c] = frag2(slot, c, t, 0);
    DDP_LYR_STAMP_AT(10)                                       // tile start: first fragments issued, bias in, weight fragments read
    if constexpr (SF) {       // stage 0's two K16 blocks: split with nothing to hide behind, once per tile (packed path)
      float xv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) xv[e] = sf[e >> 2][e & 3];
      split8_packed(xv, sc[0][0], sc[0][1], sc[0][2]);
#pragma unroll
      for (int e = 0; e < 8; ++e) xv[e] = sf[2 + (e >> 2)][e & 3];
      split8_packed(xv, sc[1][0], sc[1][1], sc[1][2]);
    }
    auto p0_stage = [&](int st, auto lastc, auto firstc) __attribute__((always_inline)) {
      constexpr bool last = decltype(lastc)::value != 0;
      constexpr bool first = decltype(firstc)::value != 0;
      stage_begin(nxt(nxt(slot)));
      const int nslot = nxt(slot);
      SplitState spa, spb;      // SF: split of the next stage's K16 block 0 (hand-over block) / of this stage's K16 block 1 (blocks 0..3)
      // the LAST stage has no successor: no fragment fetch (it used to re-fetch stage 7's own operands - six dead
      // loads and 24 registers held under the 48 residual loads in flight, which is what tipped the allocator into
      // spilling one of those residual fragments behind a vmcnt(0))
      if constexpr (!last) {
        const int stn = st + 1;
        if constexpr (SF) {
          sf_fetch(stn);
        } else {
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int c = 0; c < 3; ++c) sn[ks][c] = *reinterpret_cast<const u32x4*>(ss + ((2 * stn + ks) * 3 + c) * 1024);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
          const int blk = ks * 4 + tp;
          const bool hand = !last && blk == 7;
          const bool nb = blk + 1 < 8 || hand;
          const int rs = hand ? nslot : slot;
          const int tp2 = hand ? 0 : (tp + 1 < 4 ? tp + 1 : 0), ks2 = hand ? 0 : (tp + 1 < 4 ? ks : ks + 1);
          if (hand) {
            wait_hand(true);                                      // the next stage's image and S fragments have landed
            __syncthreads();
          }
          auto fill = [&](auto kc) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value;
            if (k == 0 && nb) w[0][2] = frag2(rs, 2, 2 * tp2, ks2);
            if (k == 1 && nb) w[1][2] = frag2(rs, 2, 2 * tp2 + 1, ks2);
            if (k == 4 && nb) w[0][1] = frag2(rs, 1, 2 * tp2, ks2);
            if (k == 5 && nb) w[1][1] = frag2(rs, 1, 2 * tp2 + 1, ks2);
            if (k == 10 && nb) w[0][0] = frag2(rs, 0, 2 * tp2, ks2);
            if (k == 11 && nb) w[1][0] = frag2(rs, 0, 2 * tp2 + 1, ks2);
            dma_slot(true, blk, k);
            if constexpr (SF) {
              // the hand-over block (ks = 1: reads sc[1]) splits the NEXT stage's first K16 block into sc[0]; blocks 0..3
              // (ks = 0: read sc[0]) split THIS stage's second K16 block into sc[1], one instruction per slot, from the
              // copies kept at the end of the previous stage
              if (hand) {
#pragma unroll
                for (int op = SPLIT_HAND_LO[k]; op < SPLIT_HAND_LO[k + 1]; ++op) split_op(op, spa, sf[0], sf[1], sc[0]);
              }
              if (!first && blk < 4 && k < 11) split_op(blk * 11 + k, spb, sh[0], sh[1], sc[1]);
            }
          };
          DDP_LYR_BLOCK(acc2[2 * tp], acc2[2 * tp + 1], sc[ks][0], sc[ks][1], sc[ks][2], fill)
        }
      if constexpr (last) {
        wait_vm12();
        __syncthreads();
      } else if constexpr (SF) {
        sh[0] = sf[2];
        sh[1] = sf[3];
      } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            asm volatile("" : "+v"(sn[ks][c]));
            sc[ks][c] = sn[ks][c];
          }
      }
      slot = nxt(slot);
    };
    p0_stage(0, I0, I1);
    DDP_LYR_STAMP_AT(11)                                       // stage 0 (waits for the tile's first fragments)
    for (int st = 1; st < 6; ++st) p0_stage(st, I0, I0);
    DDP_LYR_STAMP_AT(12)                                       // stages 1..5
    if constexpr (MODE == 7) {
      // stages 6, 7 finish u_0 = W_m . noise; its accumulators are copied aside (and stored for the u recursion of the following
      // steps), the SAME accumulators restart from the concat-conv's bias and take the eight W_x stages: xproj.  (Continuing the
      // accumulation instead would give q but not u_0 - the roundings of "0 + products" and "xproj + products" differ.)
      p0_stage(6, I0, I0);
      p0_stage(7, I0, I0);
      f32x16 accu[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) accu[t] = acc2[t];
      if (la.ubuf) {
        float* ub = la.ubuf + grp * 8192 + lane * 4;
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(ub + t * 1024 + g * 256) = f32x4{acc2[t][4 * g], acc2[t][4 * g + 1], acc2[t][4 * g + 2], acc2[t][4 * g + 3]};
      }
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(bias_s + LYR_T_BO + t * 32 + 8 * g + 4 * ht);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc2[t][4 * g + e] = b[e];
        }
      for (int st = 8; st < 15; ++st) p0_stage(st, I0, I0);
      p0_stage(15, I1, I0);
      // xproj rows out (the loop-invariant half of the concat-conv: every later step's tail reads them), q = xproj + u_0
      {
        const int m = m_base + j;
        if (la.res_frag) {                                      // fragment-major (whole 32-token groups: the buffer is padded)
          float* rf = const_cast<float*>(la.res) + grp * 8192 + lane * 4;
#pragma unroll
          for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g)
              *reinterpret_cast<f32x4*>(rf + t * 1024 + g * 256) = f32x4{acc2[t][4 * g], acc2[t][4 * g + 1], acc2[t][4 * g + 2], acc2[t][4 * g + 3]};
        } else if (m < M) {
          float* rp = const_cast<float*>(la.res) + size_t(m) * 256 + 4 * h;
#pragma unroll
          for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g)
              *reinterpret_cast<f32x4*>(rp + t * 32 + 8 * g) = f32x4{acc2[t][4 * g], acc2[t][4 * g + 1], acc2[t][4 * g + 2], acc2[t][4 * g + 3]};
        }
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) {
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          const int b = 2 * t + gp;
          float xv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) xv[e] = accu[t][8 * gp + e] + acc2[t][8 * gp + e];
          split8_packed(xv, xa[b][0], xa[b][1], xa[b][2]);
          store_q_block(qf, b, xv);
        }
      }
    }
    if constexpr (MODE == 2) {
      // residual rows (fp32, W_x x + b): fetched under the last two stages, then q = acc2 + res -> SB + fragments
      f32x4 xr[8][4];
      {
        int m = m_base + j;
        m = m < M ? m : M - 1;
        const size_t row = la.res_rn ? size_t(m / la.res_rn) * la.n_tok + m % la.n_tok : size_t(m);
        const float* rp = la.res + row * 256 + 4 * h;
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
          for (int g = 0; g < 4; ++g) xr[t][g] = *reinterpret_cast<const f32x4*>(rp + t * 32 + 8 * g);
      }
      p0_stage(6, I0, I0);
      p0_stage(7, I1, I0);
      if (la.ubuf) {       // u_0 = W_m . m_0 for the u recursion of the following steps (MODE 4)
        float* ub = la.ubuf + grp * 8192 + lane * 4;
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(ub + t * 1024 + g * 256) = f32x4{acc2[t][4 * g], acc2[t][4 * g + 1], acc2[t][4 * g + 2], acc2[t][4 * g + 3]};
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) {
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          const int b = 2 * t + gp;
          float xv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) xv[e] = acc2[t][8 * gp + e] + xr[t][2 * gp + (e >> 2)][e & 3];
          split8_packed(xv, xa[b][0], xa[b][1], xa[b][2]);
          store_q_block(qf, b, xv);
        }
      }
    }
    if constexpr (LYR) {
    // residual fragments: fetched under the last two stages
    // (fp32: 32 x 16 B per lane instead of 48 - the r02i stamps put most of P0's idle time on these two fetches: every CU
    // asks for its residual rows in the same microseconds)
    f32x4 qr[8][4];
    // MODE 10: the residual is xproj (fragment-major, la.res) + w_m d - formed behind stage 7, where the plain residual is consumed
    const float* qres = MODE == 10 ? la.res + grp * 8192 + lane * 4 : qf;
    float dv10 = 0.f;
    if constexpr (MODE == 10) {
      int m10 = m_base + j;
      m10 = m10 < M ? m10 : M - 1;
      dv10 = la.dvec[m10];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) qr[t][g] = ld_stream<NT>(qres + t * 1024 + g * 256);
    p0_stage(6, I0, I0);
    DDP_LYR_STAMP_AT(13)                                       // residual fetch (first half) + stage 6
#pragma unroll
    for (int t = 4; t < 8; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) qr[t][g] = ld_stream<NT>(qres + t * 1024 + g * 256);
    p0_stage(7, I1, I0);
    if constexpr (MODE == 10) {
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 wd = *reinterpret_cast<const f32x4*>(bias_s + LYR_T_SEG + t * 32 + 8 * g + 4 * ht);
#pragma unroll
          for (int e = 0; e < 4; ++e) qr[t][g][e] = qr[t][g][e] + wd[e] * dv10;       // (the arithmetic of MODE 3's depth head)
        }
    }

    DDP_LYR_STAMP_AT(0)                                        // P0: output_proj stages
    // ---- P1: y = acc2 + q; x = LayerNorm0(y) -> fc1's B fragments (registers); acc2 <- b2 + x (fc2 bias + residual)
    {
      float sum = 0.f;
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc2[t][r] + qr[t][r >> 2][r & 3];
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
          const int ch = t * 32 + 8 * g + 4 * ht;
          const f32x4 ga = *reinterpret_cast<const f32x4*>(bias_s + LYR_T_GA0 + ch);
          const f32x4 be = *reinterpret_cast<const f32x4*>(bias_s + LYR_T_BE0 + ch);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc2[t][4 * g + e] = acc2[t][4 * g + e] * (rstd * ga[e]) + be[e];
        }
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          float xv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) xv[e] = acc2[t][8 * gp + e];
          split8_packed(xv, xa[2 * t + gp][0], xa[2 * t + gp][1], xa[2 * t + gp][2]);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(bias_s + LYR_T_B2 + t * 32 + 8 * g + 4 * ht);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc2[t][4 * g + e] += b[e];
        }
      }
    }

    DDP_LYR_STAMP_AT(1)                                        // P1: residual + LayerNorm0 + split
    refresh();
    // ---- P2: FFN, 64 hidden channels at a time: acc1 = b1 + W1[chunk] . x; h = GELU(acc1); acc2 += W2[:, chunk] . h
    for (int hc = 0; hc < 16; ++hc) {
      f32x16 acc1[2];
      bias_init(acc1, hc);
      tall_pair(acc1[0], acc1[1], I0);
      DDP_LYR_STAMP_AT(2)                                      // fc1 stage pairs
      // GELU + exact split: the result IS the B operand of fc2 (k-block kb = (tile kb/2, quad pair kb%2)).
      // k-block 0 here; k-block kb+1 in the filler slots of k-block kb's four MFMA blocks: four elements per pair of
      // blocks, 4 x 18 single instructions + 6 packs handed out slot by slot (GELU_SCHED).
      u32x4 hcur[3], hn[3];
      {
        float xg[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) xg[e] = acc1[0][e];
        gelu_split8_packed(xg, hcur[0], hcur[1], hcur[2]);
      }
      DDP_LYR_STAMP_AT(3)                                      // exposed GELU of k-block 0
      // (early hand-over between the two wide stages, see tall_pair)
      u32x4 w[2][3];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c = 0; c < 3; ++c) w[t][c] = frag2(slot, c, t, 0);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        stage_begin(nxt(nxt(slot)));
        const int nslot = nxt(slot);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int kb = s2 * 2 + ks;
          const int kn = kb + 1;
#pragma unroll
          for (int tpp = 0; tpp < 2; ++tpp) {
            // two blocks share four GELU values: elements 4*tpp .. 4*tpp+3 of the NEXT k-block
            GeluState g0, g1, g2, g3;
            const float v0 = kb < 3 ? acc1[kn >> 1][8 * (kn & 1) + 4 * tpp] : 0.f;
            const float v1 = kb < 3 ? acc1[kn >> 1][8 * (kn & 1) + 4 * tpp + 1] : 0.f;
            const float v2 = kb < 3 ? acc1[kn >> 1][8 * (kn & 1) + 4 * tpp + 2] : 0.f;
            const float v3 = kb < 3 ? acc1[kn >> 1][8 * (kn & 1) + 4 * tpp + 3] : 0.f;
            auto block = [&](auto halfc) __attribute__((always_inline)) {
              constexpr int half = decltype(halfc)::value;
              const int tp = 2 * tpp + half;
              const int blk = ks * 4 + tp;
              const bool hand = s2 == 0 && blk == 7;            // fetch block 0 of the second wide stage
              const bool nb = blk + 1 < 8 || hand;
              const int rs = hand ? nslot : slot;
              const int tp2 = hand ? 0 : (tp + 1 < 4 ? tp + 1 : 0), ks2 = hand ? 0 : (tp + 1 < 4 ? ks : ks + 1);
              if (hand) {
                __builtin_amdgcn_s_waitcnt(0x0F7B);               // vmcnt(11)
                __syncthreads();
              }
              auto fill = [&](auto kc) __attribute__((always_inline)) {
                constexpr int k = decltype(kc)::value;
                if (k == 0 && nb) w[0][2] = frag2(rs, 2, 2 * tp2, ks2);
                if (k == 1 && nb) w[1][2] = frag2(rs, 2, 2 * tp2 + 1, ks2);
                if (k == 4 && nb) w[0][1] = frag2(rs, 1, 2 * tp2, ks2);
                if (k == 5 && nb) w[1][1] = frag2(rs, 1, 2 * tp2 + 1, ks2);
                if (k == 10 && nb) w[0][0] = frag2(rs, 0, 2 * tp2, ks2);
                if (k == 11 && nb) w[1][0] = frag2(rs, 0, 2 * tp2 + 1, ks2);
                if (k == 3) dma(blk, blk + 1);
                if (k == 8 && blk < 4) dma(8 + blk, 9 + blk);
                if (kb < 3) {
                  constexpr int sl = half * 12 + k;
                  gelu_maybe<GELU_SCHED[sl][0]>(g0, v0);
                  gelu_maybe<GELU_SCHED[sl][1]>(g1, v1);
                  gelu_maybe<GELU_SCHED[sl][2]>(g2, v2);
                  gelu_maybe<GELU_SCHED[sl][3]>(g3, v3);
                  if (sl == 21) {
                    hn[0][2 * tpp] = __builtin_amdgcn_perm(gelu_ph(g1), gelu_ph(g0), 0x07060302);
                    hn[0][2 * tpp + 1] = __builtin_amdgcn_perm(gelu_ph(g3), gelu_ph(g2), 0x07060302);
                  }
                  if (sl == 22) {
                    hn[1][2 * tpp] = __builtin_amdgcn_perm(gelu_pm(g1), gelu_pm(g0), 0x07060302);
                    hn[1][2 * tpp + 1] = __builtin_amdgcn_perm(gelu_pm(g3), gelu_pm(g2), 0x07060302);
                  }
                  if (sl == 23) {
                    hn[2][2 * tpp] = __builtin_amdgcn_perm(gelu_pl(g1), gelu_pl(g0), 0x07060302);
                    hn[2][2 * tpp + 1] = __builtin_amdgcn_perm(gelu_pl(g3), gelu_pl(g2), 0x07060302);
                  }
                }
              };
              DDP_LYR_BLOCK(acc2[2 * tp], acc2[2 * tp + 1], hcur[0], hcur[1], hcur[2], fill)
            };
            block(I0);
            block(I1);
          }
          if (kb < 3) {
#pragma unroll
            for (int c = 0; c < 3; ++c) hcur[c] = hn[c];
          }
        }
        if (s2 == 1) {
          wait_vm12();
          __syncthreads();
        }
        slot = nxt(slot);
      }
      DDP_LYR_STAMP_AT(4)                                      // fc2 stage pairs (GELU of k-blocks 1..3 in the filler slots)
    }

    refresh();
    // ---- LayerNorm1 x FiLM: q' -> HBM as SB, and -> registers as the B fragments of the next projections
    {
      // fresh base (re-derived, not copied: qs would have to live - spilled - across the FFN): otherwise the 48 64-bit
      // addresses of the residual loads are kept for these stores
      float* qst = la.Q + grp * 8192 + lane * 4;
      asm volatile("" : "+v"(qst));
      char* qsb = reinterpret_cast<char*>(la.Q_sb) + grp * 256 * 192 + lane * 16;      // only dereferenced when la.Q_sb is set
      // nothing to hide behind here: two channels per instruction on the packed-fp32 path
      f32x2 sum2 = {0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; r += 2) sum2 += f32x2{acc2[t][r], acc2[t][r + 1]};
      const float mean = half_sum(sum2[0] + sum2[1]) * (1.0f / 256.0f);
      const f32x2 mean2 = {mean, mean};
      f32x2 var2 = {0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2 d = f32x2{acc2[t][r], acc2[t][r + 1]} - mean2;
          acc2[t][r] = d[0];
          acc2[t][r + 1] = d[1];
          var2 = __builtin_elementwise_fma(d, d, var2);
        }
      const float rstd = 1.0f / sqrtf(half_sum(var2[0] + var2[1]) * (1.0f / 256.0f) + 1e-5f);
      const f32x2 rstd2 = {rstd, rstd};
#pragma unroll
      for (int t = 0; t < 8; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int ch = t * 32 + 8 * g + 4 * ht;
          const f32x4 ga = *reinterpret_cast<const f32x4*>(bias_s + LYR_T_GA1 + ch);
          const f32x4 be = *reinterpret_cast<const f32x4*>(bias_s + LYR_T_BE1 + ch);
#pragma unroll
          for (int e = 0; e < 4; e += 2) {
            const f32x2 y = __builtin_elementwise_fma(f32x2{acc2[t][4 * g + e], acc2[t][4 * g + e + 1]},
                                                     rstd2 * f32x2{ga[e], ga[e + 1]}, f32x2{be[e], be[e + 1]});
            acc2[t][4 * g + e] = y[0];
            acc2[t][4 * g + e + 1] = y[1];
          }
        }
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          const int b = 2 * t + gp;
          float xv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) xv[e] = acc2[t][8 * gp + e];
          split8_packed(xv, xa[b][0], xa[b][1], xa[b][2]);
          if constexpr (!LT) {                        // (MODE 6 / 8 / 9: the layer output stays in xa - the tail's operand, below)
            store_q_block(qst, b, xv);
            if (la.Q_sb) {
#pragma unroll
              for (int c = 0; c < 3; ++c) *reinterpret_cast<u32x4*>(qsb + (b * 3 + c) * 1024) = xa[b][c];
            }
          }
        }
      }
    }

    DDP_LYR_STAMP_AT(5)                                        // LayerNorm1 + FiLM + split + q' stores
    }   // MODE 0 / 6
    }   // MODE 0 / 2 / 6 only
    }
    if constexpr (MODE == 8 || MODE == 9) {
      // ---- bev / depth tail: ONE 64-row chunk of the head convolution on the layer output LayerNorm1 left in xa.  This lane holds
      // rows 8g + 4h + e of tile t = 0 (the head has <= 32 rows: tile 1 is zero weights)
      refresh();
      f32x16 lg[2];
      bias_init(lg, LYR_T_SEG / 64);
      tall_pair(lg[0], lg[1], I1);
      const int m = m_base + j;
      const bool valid = m < M;
      float* pr = la.prob + size_t(valid ? m : 0) * 32 + 4 * h;            // token-major rows of 32 floats
      if constexpr (MODE == 8) {
        const int K = la.num_classes;
        unsigned code = 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (8 * g >= K) break;                                            // (uniform)
          f32x4 p;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            p[e] = 1.0f / (1.0f + expf(-lg[0][4 * g + e]));                  // torch.sigmoid (heads/segm/deformable_head_with_time.py:235)
            if (g == 0 && 4 * h + e < K && p[e] > la.threshold) code |= 1u << (4 * h + e);
          }
          if (la.prob_mode == 2) p += *reinterpret_cast<const f32x4*>(pr + 8 * g);
          if (valid) *reinterpret_cast<f32x4*>(pr + 8 * g) = p;             // (columns >= K: never read)
        }
        code |= unsigned(__shfl_xor(int(code), 32, 64));
        if (la.x0_idx && valid && h == 0) la.x0_idx[m] = (unsigned char)code;
      } else {
        if (valid) {
          *reinterpret_cast<f32x4*>(pr) = f32x4{lg[0][0], lg[0][1], lg[0][2], lg[0][3]};                      // taps 4h .. 4h + 3
          if (h == 0) *reinterpret_cast<f32x4*>(pr + 8) = f32x4{lg[0][4], lg[0][5], lg[0][6], lg[0][7]};      // taps 8 .. 11 (8 is the last)
        }
      }
      continue;
    }
    if constexpr (MODE == 1 || MODE == 4 || MODE == 6) {
      // ---- seg tail: q fragments of this tile (the layer output; MODE 6: LayerNorm1 just left them in xa), scores = conv_seg(q),
      // per-token update
      if constexpr (MODE == 6) refresh();
      // fresh base (as LayerNorm1's stores above): derived from the tile-top qf, the addresses of the next q's stores would be
      // computed at the top of the tile and kept - spilled - across the whole FFN in MODE 6
      float* qft = la.Q + grp * 8192 + lane * 4;
      asm volatile("" : "+v"(qft));
      if constexpr (MODE != 6) load_q_fragments(qft);
      f32x16 lg[NCH > 0 ? NCH : 1][2];
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        bias_init(lg[c], MODE == 6 ? LYR_T_SEG / 64 + c : c);
        tall_pair(lg[c][0], lg[c][1], I1);
      }
      const int m = m_base + j;
      const bool valid = m < M;
      const int K = la.num_classes;
      // argmax over the token's classes: this lane holds classes 64c + 32t + 8g + 4h + e, its partner (lane ^ 32) the
      // other half; first maximum wins (torch.argmax)
      float best = -INFINITY;
      int bi = 0x7fffffff;
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int cls = c * 64 + t * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
            const float v = lg[c][t][r];
            if (cls < K && v > best) {
              best = v;
              bi = cls;
            }
          }
      {
        const float ob = __shfl_xor(best, 32, 64);
        const int oi = __shfl_xor(bi, 32, 64);
        if (ob > best || (ob == best && oi < bi)) {
          best = ob;
          bi = oi;
        }
      }
      // padding tokens of the last group carry garbage (possibly NaN scores: no maximum found): keep the LUT row in range
      if (!valid || bi >= K) bi = 0;
      if (la.x0_idx && valid && h == 0) la.x0_idx[m] = (unsigned char)bi;
      if constexpr (FORCE) {      // teacher forcing: a test instrument in its own instantiations, the product's tails do not have it
        if (valid) {
          bi = la.x0_force[m];
          if (bi >= K) bi = 0;
        }
      }
      // Every global load of the epilogue is issued in a batch ahead of its consumers.  The straightforward loops (load 5,
      // wait, compute, store 3 - sixteen times; load, wait, add, store - 24 times on the accumulated probabilities) pay a
      // full memory round trip per iteration at one wave per SIMD: vector memory completes in order, vmcnt counts stores
      // too, and the LUT loads may not move above the map stores they could alias.
      float* pr = la.prob + grp * size_t(NCH * 2048) + lane * 4;           // + ((2 c + t) * 4 + g) * 256
      char* ms = reinterpret_cast<char*>(la.mask_sb) + grp * 256 * 192 + lane * 16;
      if (la.prob_mode == 1 || la.prob_mode == 2) {          // softmax over the classes, accumulated over the steps
        float ssum = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int cls = c * 64 + t * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
              const float e = cls < K ? __builtin_amdgcn_exp2f((lg[c][t][r] - best) * 1.44269504088896340736f) : 0.f;
              lg[c][t][r] = e;
              ssum += e;
            }
        ssum += __shfl_xor(ssum, 32, 64);
        const float inv = 1.0f / ssum;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) lg[c][t][r] = lg[c][t][r] * inv;
      }
      const bool accumulate = la.prob_mode == 2;
      f32x4 old[NCH > 0 ? NCH : 1][2][4];
      if (accumulate) {
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g)
              old[c][t][g] = *reinterpret_cast<const f32x4*>(pr + ((2 * c + t) * 4 + g) * 256);        // (whole chunks are allocated)
      }
      if (accumulate) {
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) lg[c][t][r] += old[c][t][r >> 2][r & 3];
      }
      if (valid && la.prob_mode != 0) {
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int cls0 = c * 64 + t * 32 + 8 * g;
              if (cls0 < K)                                     // (uniform: whole 1-KiB pieces; classes past K are never read back)
                *reinterpret_cast<f32x4*>(pr + ((2 * c + t) * 4 + g) * 256) = f32x4{lg[c][t][4 * g], lg[c][t][4 * g + 1], lg[c][t][4 * g + 2], lg[c][t][4 * g + 3]};
            }
      }
      // MODE 6 after the LAST step: nothing to update, no next head.  One uniform exit instead of two conditions (this one and
      // P3's): with two, the allocator keeps the old xa alive - spilled - along the path that skips the update
      if constexpr (MODE == 6) {
        if (!la.has_next) continue;
      }
      if constexpr (MODE == 4 || MODE == 6) {
        // ---- the update in terms of u = W_m . m (see LayerArgs), then the NEXT step's q = (W_x x + b) + u' -> SB + fragments:
        // per quarter of the row 8 + 8 + 8 loads (u, table row of the argmax class, x projection row) in flight
        const float* trow = la.tlut + size_t(bi) * 256 + 4 * h;
        float* ub = la.ubuf + grp * 8192 + lane * 4;
        int mr = m < M ? m : M - 1;
        const size_t row = la.res_rn ? size_t(mr / la.res_rn) * la.n_tok + mr % la.n_tok : size_t(mr);
        const float* rp = la.res + row * 256 + 4 * h;
        const float* rpf = la.res + grp * 8192 + lane * 4;                    // (res_frag: the accumulator layout, as Q)
        const bool rfrag = la.res_frag != 0;
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
          f32x4 uu[2][4], tt[2][4], xx[2][4];
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int t = qt * 2 + i;
              uu[i][g] = *reinterpret_cast<const f32x4*>(ub + t * 1024 + g * 256);
              tt[i][g] = *reinterpret_cast<const f32x4*>(trow + t * 32 + 8 * g);
            }
          if (rfrag) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int g = 0; g < 4; ++g) xx[i][g] = *reinterpret_cast<const f32x4*>(rpf + (qt * 2 + i) * 1024 + g * 256);
          } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int g = 0; g < 4; ++g) xx[i][g] = *reinterpret_cast<const f32x4*>(rp + (qt * 2 + i) * 32 + 8 * g);
          }
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int t = qt * 2 + i;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
              for (int e = 0; e < 4; ++e) uu[i][g][e] = la.ua * uu[i][g][e] + la.uc * tt[i][g][e];
              *reinterpret_cast<f32x4*>(ub + t * 1024 + g * 256) = uu[i][g];
            }
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
              const int b = 2 * t + gp;
              float xv[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) xv[e] = xx[i][2 * gp + (e >> 2)][e & 3] + uu[i][2 * gp + (e >> 2)][e & 3];
              split8_packed(xv, xa[b][0], xa[b][1], xa[b][2]);
              store_q_block(qft, b, xv);
            }
          }
        }
      } else if (la.mask_sb) {
      // x0 = LUT[argmax]; DDIM step of the noisy map (ddp.py:235-239), SB in, SB out - two batches of eight K16 blocks:
      // 24 + 16 loads in flight, then eight compute / store rounds (all 16 blocks at once need 320 architectural VGPRs)
      {
        const float* x0row = la.lut + size_t(bi) * 256 + 4 * h;
        const float inv_sig = 1.0f / fmaxf(la.sigma, 1e-8f);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          u32x4 mk[8][3];
          f32x4 xl[8][2];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int b = half * 8 + i;
#pragma unroll
            for (int c = 0; c < 3; ++c) mk[i][c] = *reinterpret_cast<const u32x4*>(ms + (b * 3 + c) * 1024);
            xl[i][0] = *reinterpret_cast<const f32x4*>(x0row + 16 * b);
            xl[i][1] = *reinterpret_cast<const f32x4*>(x0row + 16 * b + 8);
          }
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int b = half * 8 + i;
            float mn[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const float mt = (bf_elem(mk[i][0], u) + bf_elem(mk[i][1], u)) + bf_elem(mk[i][2], u);
              const float x0 = u < 4 ? xl[i][0][u] : xl[i][1][u - 4];
              const float pn = (mt - la.alpha * x0) * inv_sig;
              mn[u] = x0 * la.alpha_next + pn * la.sigma_next;
            }
            u32x4 q1, q2, q3;
            split8_packed(mn, q1, q2, q3);
            *reinterpret_cast<u32x4*>(ms + (b * 3 + 0) * 1024) = q1;
            *reinterpret_cast<u32x4*>(ms + (b * 3 + 1) * 1024) = q2;
            *reinterpret_cast<u32x4*>(ms + (b * 3 + 2) * 1024) = q3;
          }
        }
      }
      }   // la.mask_sb
    }
    if constexpr (MODE != 1 && MODE != 8 && MODE != 9) {
    refresh();
    // ---- P3: the next layer's value_proj (4 chunks of 64 channels) and sampling projection (2 chunks)
    if (MODE == 2 || MODE == 3 || MODE == 4 || MODE == 6 || MODE == 7 || la.has_next) {       // (MODE 6 without a next step left the tile above)
      const int m = m_base + j;
      const bool valid = m < M;
      const int mm = valid ? m : M - 1;
      // opaque per tile: otherwise the reciprocals behind these divisions are hoisted out of the tile loop and held
      // (spilled) across all of it
      int ntk = la.n_tok, wmap = la.w;
      asm volatile("" : "+s"(ntk), "+s"(wmap));
      const int bimg = mm / ntk;
      const int n = mm - bimg * ntk;
      const int pi = n / wmap;
      const int pj = n - pi * wmap;
      const float fi = float(pi), fj = float(pj);
      const int hmap = ntk / wmap;
      const size_t vrow = size_t(bimg) * (hmap + 2) * (wmap + 2) + size_t(pi + 1) * (wmap + 2) + (pj + 1);
      for (int vc = 0; vc < 4; ++vc) {
        f32x16 a[2];
        bias_init(a, 16 + vc);
        tall_pair(a[0], a[1], I1);
        if (valid) {
          float* dst = la.v_out + vrow * 256 + vc * 64 + 4 * h;
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g)
              st_stream<NT>(dst + t * 32 + 8 * g, f32x4{a[t][4 * g], a[t][4 * g + 1], a[t][4 * g + 2], a[t][4 * g + 3]});
        }
      }
      DDP_LYR_STAMP_AT(6)                                      // next value_proj (8 stages + stores)
#pragma unroll
      for (int sc2 = 0; sc2 < 2; ++sc2) {
        f32x16 a[2];
        bias_init(a, 20 + sc2);
        // the positional term py[i] + px[j] of this chunk's columns is fetched HERE, in front of the chunk's stages, not in
        // its epilogue: there, "2 loads, wait, compute, store" per column group made every group wait for the store before
        // it and for the weight-stream DMA in flight (vector memory completes in order) - a memory round trip each, at
        // one wave per SIMD.  (Rows of clamped tokens are valid addresses; chunk 1 has one 32-column tile.)
        f32x4 pos[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int col = sc2 * 64 + (sc2 == 1 ? 0 : t) * 32 + 8 * g + 4 * h;
            pos[t][g] = *reinterpret_cast<const f32x4*>(la.py + pi * 96 + col) + *reinterpret_cast<const f32x4*>(la.px + pj * 96 + col);
          }
        if (sc2 == 0) {
          tall_pair(a[0], a[1], I1);
        } else {                                               // columns 64..95 as one split-K stage: a[0] + a[1]
#pragma unroll
          for (int r = 0; r < 16; ++r) a[1][r] = 0.f;
          tall_stage(a[0], a[1], I2);
#pragma unroll
          for (int r = 0; r < 16; ++r) a[0][r] += a[1][r];
        }
        if (valid) {
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            if (sc2 == 1 && t == 1) continue;                 // one 32-column tile only
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              f32x4 v = f32x4{a[t][4 * g], a[t][4 * g + 1], a[t][4 * g + 2], a[t][4 * g + 3]} + pos[t][g];
              if (sc2 == 0) {                                  // sampling offsets -> pixel coordinates (x, y, x, y)
                v[0] += fj; v[1] += fi; v[2] += fj; v[3] += fi;
              } else {                                         // attention weights: softmax over the head's 4 points
                // (hardware exp2 / rcp: 1 ulp each - the weights stay within 2e-7 of the IEEE softmax; this runs
                // with nothing to hide behind, and expf + four IEEE divisions were 2/3 of the epilogue's instructions)
                const float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_exp2f((v[e] - mx) * 1.44269504088896340736f);
                const float inv = __builtin_amdgcn_rcpf((v[0] + v[1]) + (v[2] + v[3]));
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= inv;
              }
              // head-major table [head][token][8 coordinates | 4 weights]: the gather runs one head per block.  This lane's four
              // columns are sc2 * 64 + t * 32 + 8g + 4h ..: coordinates of head 4t + g (points 2h, 2h + 1) / weights of head 2g + h
              const int shd = sc2 == 0 ? (t * 4 + g) : 2 * g + h;
              st_stream<NT>(la.samp_out + (size_t(shd) * M + m) * 12 + (sc2 == 0 ? 4 * h : 8), v);
            }
          }
        }
      }
      DDP_LYR_STAMP_AT(7)                                      // next sampling projection (3 stages + epilogues)
    }
    }
  }
#ifdef DDP_LYR_STAMP
  if (la.stamps && (threadIdx.x & 63) == 0) {
    for (int i = 0; i < LYR_NSTAMP; ++i) la.stamps[(size_t(blockIdx.x) * 4 + wave) * LYR_NSTAMP + i] += st_acc[i];   // summed over launches
    // per-XCD completion skew (scripts/xcd_skew.py): the 100 MHz chip-wide clock at the start and the end of this wave's share of
    // the LAST launch, and the XCD it ran on (slots 8, 14, 15 carry no phase)
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long* row = la.stamps + (size_t(blockIdx.x) * 4 + wave) * LYR_NSTAMP;
    row[8] = st_t0;
    row[14] = __builtin_amdgcn_s_memrealtime();
    row[15] = xcc & 0xF;
  }
#endif
#undef DDP_LYR_BLOCK
#undef DDP_LYR_BLOCK2
#undef DDP_LYR_K
#undef DDP_LYR_SB
  wait_vm0();
}

}  // namespace b3
}  // namespace ddp
