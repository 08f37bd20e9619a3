// ddp_internal.h - internal launch interface between the translation units of libddp_mi355x.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <atomic>
#include "../../include/ddp_mi355x.h"

// The sample table the layer kernel's P3 epilogue hands to the LDS-staged gather is head-major: [head][token][8 pixel
// coordinates | 4 attention weights] (the gather runs one head per block); the standalone GEMM epilogue / wave-per-token
// gathers keep token-major rows of DDP_SAMP_STRIDE floats.  Same size, same workspace slot.

// Build-time variant of the layer kernel / gather pair (same-box A/B builds: scripts/variant_build.sh x -DDDP_S_F32=0):
//   DDP_S_F32          the LDS gather hands the attention output to the layer kernel as fp32 fragments (1 KiB per token) and
//                      P0 splits it in its filler slots; 0: as SB (1.5 KiB per token, split in the gather)
#ifndef DDP_S_F32
#define DDP_S_F32 1
#endif

namespace ddp {

void set_error(const char* fmt, ...);
int check_launch(const char* what);

// Per-(kernel, device) one-time setup.  A process may drive several GPUs (one engine per device, possibly from several host
// threads): the dynamic-LDS attribute has to be set on each of them, and the CU count is a property of the device the
// launch goes to - neither may be cached process-wide.  Atomics: concurrent launches can at worst repeat the (idempotent) call.
struct LdsAttrOnce {
  std::atomic<unsigned long long> done{0};           // bit per device id < 64
  void ensure(const void* fn, int bytes) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && ((done.load(std::memory_order_acquire) >> dev) & 1ull)) return;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
      (void)hipGetLastError();                       // the launch that follows fails and reports through check_launch
      set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d) failed on device %d", bytes, dev);
      return;
    }
    if (dev >= 0 && dev < 64) done.fetch_or(1ull << dev, std::memory_order_release);
  }
};
inline int cu_count() {                              // compute units of the CURRENT device (persistent-kernel grids)
  static std::atomic<int> cache[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64) {
    const int c = cache[dev].load(std::memory_order_relaxed);
    if (c > 0) return c;
  }
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  if (dev >= 0 && dev < 64) cache[dev].store(n, std::memory_order_relaxed);
  return n;
}

// call-site tags of the GEMM launches (distinct kernel symbols for rocprofv3; profiler hook ids)
enum GemmTag {
  TAG_GENERIC = 0, TAG_XPROJ = 1, TAG_FEAT = 2, TAG_VALUE = 3, TAG_SAMP = 4, TAG_OUTPROJ_LN = 5, TAG_FC1 = 6,
  TAG_FC2_LN = 7, TAG_HEAD = 8, TAG_GATHER = 9, TAG_LAYER_TAIL = 10, TAG_COUNT = 11
};
// optional per-call-site event timing (ddp_profile_* in the C ABI); no-ops unless armed
void prof_begin(int tag, hipStream_t st);
void prof_end(int tag, hipStream_t st);

// ---- ddp_gemm.hip -------------------------------------------------------------------------------
// "blk" = fragment-major activation layout (gemm_f32.h); rows padded to 128.
// row-major out = A W^T + bias (+ add[row map]) (+GELU);  A row-major (lda) or fragment-major (a_blk, K ch)
int launch_linear(const float* A, int lda, bool a_blk, const float* W, int ldw, const float* bias, const float* add,
                  int ld_add, int rn, int n_tok, float* out, int ldo, int M, int N, int K, int gelu, hipStream_t st,
                  int tag = TAG_GENERIC);
// fragment-major out (N % 256 == 0) = A W^T + bias (+ row-major add[row map]) (+GELU)
int launch_linear_blk(const float* A, int lda, bool a_blk, const float* W, int ldw, const float* bias,
                      const float* add, int ld_add, int rn, int n_tok, float* out_blk, int M, int N, int K, int gelu,
                      hipStream_t st);
// fragment-major out = LN(A W^T + bias + res_blk) * ga + be   (N == 256; ga/be = affine x FiLM)
int launch_linear_res_ln_blk(const float* A, int lda, bool a_blk, const float* W, int ldw, const float* bias,
                             const float* res_blk, const float* ga_aff, const float* be_aff, float* out_blk, int M,
                             int K, hipStream_t st);
// samp (M,96) row-major = sampling epilogue(A_blk Wcat^T) with positional tables; Wcat (96,256)
int launch_linear_samp(const float* A_blk, const float* Wcat, const float* py, const float* px, int n_tok, int w,
                       float* out, int M, hipStream_t st);

// ---- ddp_gemm_bf16.hip --------------------------------------------------------------------------
// bf16x3-split GEMM (gemm_bf16x3.h).  "sb" = split fragment-major bf16 triplets; rows padded to 256.
struct SplitW {
  const unsigned short* p;   // [3][rows][K] bf16, K-permuted (k_split_weights)
  size_t comp_stride;        // elements between components
};
int launch_b3_linear(const unsigned short* A_sb, const SplitW& w, const float* bias, const float* add, int ld_add,
                     int rn, int n_tok, float* out, int ldo, int M, int N, int K, hipStream_t st, int tag);
int launch_b3_linear_sb(const unsigned short* A_sb, const SplitW& w, const float* bias, const float* add, int ld_add,
                        int rn, int n_tok, unsigned short* out_sb, float* out_f32_blk, int M, int N, int K, int gelu,
                        hipStream_t st, int tag);
int launch_b3_linear_res_ln(const unsigned short* A_sb, const SplitW& w, const float* bias, const float* res_blk,
                            const unsigned short* res_sb, const float* ga_aff, const float* be_aff, float* out_f32_blk,
                            unsigned short* out_sb, int M, int K, hipStream_t st, int tag);
int launch_b3_linear_samp(const unsigned short* A_sb, const SplitW& wcat, const float* py, const float* px, int n_tok,
                          int w, float* out, int M, hipStream_t st);
// layer kernel (layer_bf16x3.h): weight stream builder + launcher
int launch_build_stages(const unsigned short* Wp, size_t comp_stride, int K, int rows_valid, int tall, int n_rowblk, int n_kblk,
                        int base, int a, int b, int c, unsigned char* stream, hipStream_t st);
struct LayerLaunch {
  const unsigned short* S;   // attention output as SB, or
  const float* Sf;           // as fp32 fragment-major (what the LDS gather writes on the product path)
  float* Q;                 // fp32 fragment-major rows of 256 (residual in, layer output out, in place)
  unsigned short* Q_sb;     // optional: the output also as SB (a tile GEMM consumes it)
  const unsigned char* stream;
  const float* bias_ext;
  const float *bo, *ga0, *be0, *b2, *ga1, *be1;
  int M, has_next;
  float* v_out;             // zero-padded value map: row of token (b,i,j) = b*(h+2)*(w+2) + (i+1)*(w+2) + (j+1)
  float* samp_out;
  const float *py, *px;
  int n_tok, w;
  // layer 0 of a depth step on the chain path (k_layer MODE 10): the residual q = res_f + wm * dvec[m] is formed in the kernel from the
  // fragment-major xproj (res_f), the concat-conv's depth column (wm, 256) and the noisy depth (dvec); Q is only written.  nullptr: plain.
  const float* res_f;
  const float* wm;
  const float* dvec;
};
int launch_b3_layer(const LayerLaunch& a, hipStream_t st);
// seg tail on the layer kernel's machinery: conv_seg + argmax + softmax accumulation + x0 LUT + DDIM update (SB in / out)
struct TailLaunch {
  float* Q;                     // fp32 fragment-major decoder output (fuse_next: replaced by the next step's q)
  const unsigned char* stream;  // 2 * chunks stage images of conv_seg (64 classes per chunk)
  const float* bias_ext;        // conv_seg bias, zero padded to b3_layer_bias_floats()
  const float* lut;
  float* prob;
  unsigned short* mask_sb;
  unsigned char* x0_idx;        // optional: argmax class per token (diagnostic trace)
  const unsigned char* x0_force;  // optional (DDP_FLAG_FORCE_X0): the class to feed back instead of the argmax
  int M, num_classes, ldl, prob_mode;
  float alpha, sigma, alpha_next, sigma_next;
  // fuse_next: this launch is also the head of the NEXT step (k_layer MODE 4): mask_sb = nullptr, the update runs on
  // u = W_m . m (ubuf, in place) with the table tlut = W_m . LUT^T, then q_next = res[row] + u' -> Q (SB, in place) and
  // layer 0's value / sampling projections.  `stream` = 2 * chunks conv_seg images + 11 projection images,
  // `bias_ext` = conv_seg bias | zeros | layer 0's value_proj bias at [1024, 1280) | zeros.
  int fuse_next;
  float* ubuf;
  const float* tlut;
  const float* res;
  int res_rn;
  int res_frag;                 // res is fragment-major (written by launch_b3_head_nchw with res_frag)
  float* v_out;
  float* samp_out;
  const float *py, *px;
  int n_tok, w;
  // launch_b3_layer_tail only - kind 1 (bev, k_layer MODE 8): prob = accumulated sigmoid maps, token-major rows of 32 (prob_mode 1 / 2),
  // x0_idx = the step's code per token (bit k = sigmoid_k > threshold; num_classes <= 8) or nullptr; kind 2 (depth, MODE 9): prob = the
  // nine per-tap dot products of conv_depth, token-major rows of 32.  No next-step head in either (fuse_next = 0).
  int kind;
  float threshold;
};
int launch_b3_tail(const TailLaunch& a, hipStream_t st);
// the LAST decoder layer of a step and that step's seg tail as ONE kernel (k_layer MODE 6, ddp_layer_tail.hip): `t.Q`, the layer
// output, never travels to HBM.  t.fuse_next as in launch_b3_tail (a next step follows) or the last step's plain tail; t.mask_sb
// must be nullptr (u chain).  `stream` = 72 layer stages + t's stream; `bias_ext` = fc1 bias | layer 0's value_proj bias at
// [1024, 1280); `seg_bias` = conv_seg's bias zero padded to 256 floats.  Built for 1..256 classes (b3_layer_tail_supported()).
// t.kind 1 / 2: the bev / depth tails (k_layer MODE 8 / 9): `stream` = 72 layer stages + the head's 2 tall stages.
bool b3_layer_tail_supported(int num_classes);
int launch_b3_layer_tail(const LayerLaunch& l, const TailLaunch& t, const unsigned char* stream, const float* bias_ext,
                         const float* seg_bias, hipStream_t st);
// head of a step on the same machinery: q = W_m . m_t + xproj -> SB, then layer 0's value / sampling projections
struct PrologueLaunch {
  const unsigned short* mask_sb;   // SB noisy map (A operand)
  float* Q;                        // fp32 fragment-major q out
  const unsigned char* stream;     // 8 wide stages of W_m + 8 + 3 tall stages of layer 0's value_proj / sampling projection
  const float* bias_ext;           // zeros | value_proj bias at [1024, 1280) | zeros
  const float* res;                // xproj rows (W_x x + b)
  int res_rn;                      // r * N when r noisy maps share one x row block, else 0
  int res_frag;                    // launch_b3_head_nchw: write xproj fragment-major (the tails that follow read it that way)
  float* ubuf;                     // optional out: u_0 = W_m . m_0 (fp32 fragment-major) for the fused tails that follow
  int M;
  float* v_out;                    // zero-padded value map
  float* samp_out;
  const float *py, *px;
  int n_tok, w;
};
int launch_b3_prologue(const PrologueLaunch& a, hipStream_t st);
// the head of the FIRST step straight from the caller's NCHW tensors (k_layer MODE 7, ddp_layer_tail.hip): u_0 = W_m . noise,
// xproj = W_x x + b (written as fp32 rows: the loop-invariant half of the concat-conv), q = xproj + u_0, layer 0's projections.
// One noisy map per image, 256 feature channels.  `a.mask_sb` unused, `a.res` = xproj OUT, stream = 8 wide stages of W_m + 8 of
// W_x + the 11 projection images.
int launch_b3_head_nchw(const PrologueLaunch& a, const float* nchw_noise, const float* nchw_x, const float* bias, hipStream_t st);
// layer 0's value / sampling projections alone (k_layer MODE 3): q given as SB (res == nullptr), or formed as the depth
// concat-conv q = res[row] + wm * dvec[m] and written to Q.  stream = the 11 projection images, bias_ext as PrologueLaunch.
struct DepthUpdateArgs;
struct L0ProjLaunch {   // (plain aggregate: every field is set by the caller)
  float* Q;                        // fp32 fragment-major q (in; out when formed here)
  const unsigned char* stream;
  const float* bias_ext;
  const float* res;
  int res_rn;
  const float* wm;                 // (256) depth column of the concat-conv, with res
  const float* dvec;               // (M)
  int M;
  float* v_out;
  float* samp_out;
  const float *py, *px;
  int n_tok, w;
  // depth, steps >= 1: the previous step's DDIM update (k_depth_update's arithmetic on the taps the previous step's last layer left)
  // runs inside this launch, in front of the head; upd == nullptr: dvec is used as it is
  const struct DepthUpdateArgs* upd;
};
int launch_b3_l0proj(const L0ProjLaunch& a, hipStream_t st);
// stream GEMM of the necks (k_layer MODE 5): up to four problems in one persistent launch, tiles in the order given
struct SgemmProblem {
  const float* A;               // fp32 fragment-major, 32 * ns channels (conv: the 256-channel map)
  float* out;                   // fp32 fragment-major, 256 channels, rows padded to 128
  const unsigned char* stream;  // ns wide stage images
  int M, ns, conv_h, conv_w;    // conv_h > 0: 3x3 convolution view (ns = 72)
  const float* bias;            // optional (256 floats)
  double* gn_partial;           // optional: fused GroupNorm partial sums [image][token / 32][32 groups][2] (gn_N % 32 == 0)
  int gn_N;                     // tokens per image
  int nchw_N;                   // > 0: A is an NCHW tensor (images of nchw_N tokens, 32 * ns channels), read in place (no conv view)
};
int launch_b3_sgemm(const SgemmProblem* pr, int n, int act, int conv_dil, hipStream_t st);
size_t b3_stage_bytes();
size_t b3_prologue_stream_bytes();
size_t b3_layer_stream_bytes();
int b3_layer_bias_floats();
// W fp32 (rows, ld) -> Wp[3][rows][K]
int launch_split_weights(const float* W, int ld, int rows, int K, unsigned short* out, hipStream_t st);
// fp32 row-major (rows, C) ld -> SB
int launch_row_to_sb(const float* in, int ld, unsigned short* out_sb, int rows, int C, hipStream_t st);

// ---- ddp_kernels.hip ----------------------------------------------------------------------------
int launch_nchw_to_tok(const float* in, float* out, int R, int C, int N, hipStream_t st);
// NCHW (R,C,N) fp32 -> SB (rows R*N padded to 256, C channels, C % 16 == 0)
int launch_nchw_to_sb(const float* in, unsigned short* out_sb, int R, int C, int N, hipStream_t st);
// row-major (rows,256) -> fragment-major
int launch_row_to_blk(const float* in, float* out_blk, int rows, hipStream_t st);
// ga = gamma*(scale+1), be = beta*(scale+1)+shift for S x L (film (S,L,512) = scale|shift); out (S,L,512) = ga|be
int launch_fold_affine(const float* gamma, const float* beta, const float* film, float* out, int count, hipStream_t st);
int launch_msda_gather(const float* value, const float* samp, float* out, int rows, int n_tok, int h, int w,
                       hipStream_t st);
int launch_msda_gather_sb(const float* value, const float* samp, unsigned short* out_sb, int rows, int n_tok, int h, int w,
                          hipStream_t st);
// the LDS-staged gather of the sampling loop; out_f32_blk != nullptr: fp32 fragment-major output (the layer kernel's S operand),
// else SB
int launch_msda_gather_sb_pad(const float* vpad, const float* samp, unsigned short* out_sb, float* out_f32_blk, int rows, int n_tok,
                              int h, int w, const float* tab_y, const float* tab_x, int zero_guess, hipStream_t st);
// fragment-major (256 channels) -> row-major rows of `ld` floats, the first `cols` channels (cols % 4 == 0)
int launch_blk_to_row(const float* in_blk, float* out, int rows, hipStream_t st, int ld = 256, int cols = 256);
// adapters of ddp_msda_forward_lds: plain layouts -> padded map / head-major table / guess tables, SB -> row-major
int launch_msda_lds_adapters_in(const float* value, const float* samp, const float* guess, float* vpad, size_t vpad_floats,
                                float* samp_hm, float* tab_y, float* tab_x, int rows, int n_tok, int h, int w, hipStream_t st);
int launch_sb_to_row(const unsigned short* in_sb, float* out, int rows, int C, hipStream_t st);
int launch_sinusoid(const float* freq, const float* time_in_dev, int S, float* u, hipStream_t st);
// y[s][o] = out_act( W[o][:] . in_act(x[s][:]) + b[o] )   act: 0 none, 1 gelu(out), 2 silu(in)
int launch_matvec(const float* W, const float* b, const float* x, float* y, int in_dim, int out_dim, int S,
                  int ldx, int ldy, int in_act, int out_act, hipStream_t st);
int launch_build_lut(const float* emb, float* lut, int rows, float bit_scale, hipStream_t st);
int launch_pos_tables(const float* wcat /* (96,256) */, const float* bcat /* (96) */, float* py, float* px,
                      int h, int w, hipStream_t st);
int launch_pack_cols(const float* in, int ld_in, int off, int rows, int cols, float* out, hipStream_t st);
int launch_pack_conv3x3(const float* w, float* out, hipStream_t st);
int launch_write_floats(const float* host_vals, int n, float* out, hipStream_t st);
int launch_pack_rows(const float* a, int rows_a, const float* b, int rows_b, float* out, int cols, hipStream_t st);
// seg x0 projection + ddim/ddpm update + accumulation (segmentors/ddp.py:235-245,276-287)
struct SegUpdateArgs {
  const float* logits;  // (M, ldl)
  int ldl, num_classes;
  const float* lut;     // (K+1, 256)
  float* mask;          // (M, 256) in/out
  float* prob;          // (M, ldl) accumulated softmax / last logits; may be nullptr
  int prob_mode;        // 0 none, 1 prob = softmax, 2 prob += softmax, 3 prob = logits
  const float* step_noise;  // ddpm (M,256) token-major or nullptr
  unsigned char* x0_idx;    // optional: argmax class per token (diagnostic trace)
  const unsigned char* x0_force;  // optional (DDP_FLAG_FORCE_X0): the class to feed back instead of the argmax
  int sampler;
  ddp_step st;
  int rows;
};
int launch_seg_update(const SegUpdateArgs& a, hipStream_t st);
// x0 = (sigmoid(E[argmax_k scores]) * 2 - 1) * bit_scale, NCHW (B,K,N) -> (B,256,N)
int launch_seg_x0_nchw(const float* scores, const float* emb, float* out, int B, int K, int N, float bit_scale, hipStream_t st);
struct SegPostArgs {
  const float* logits;      // (B,K,h,w)
  int B, K, h, w;
  int H, W;                 // stage-1 size (padded image)
  int ch, cw;               // crop = img_shape
  int oh, ow;               // output = ori_shape
  int align, flip;          // flip: 0 none, 1 horizontal, 2 vertical
  unsigned char* seg;       // (B,oh,ow)
};
int launch_seg_postprocess(const SegPostArgs& a, hipStream_t st);
int launch_seg_aug_postprocess(const ddp_seg_aug* augs, int n_aug, int B, int K, int oh, int ow, int align, unsigned char* seg,
                               float* prob, hipStream_t st);
int launch_seg_slide_postprocess(const float* const* scores, const int* y1, const int* x1, int n_rows, int n_cols, int B, int K, int h,
                                 int w, int ch, int cw, int H, int W, int kh, int kw, int oh, int ow, int align, int flip, int prob_mode,
                                 unsigned char* seg, float* prob, hipStream_t st);
int launch_depth_aug_postprocess(const ddp_depth_aug* augs, int n_aug, int B, int oh, int ow, int align, float lo, float hi,
                                 float* out, hipStream_t st);
// necks on fp32 fragment-major ("blk") activations - the stream GEMM's operand / result layout
int launch_nchw_to_blk(const float* in, float* out_blk, int R, int C, int N, hipStream_t st);
int launch_gn_stats_blk(const float* y_blk, double* partial, float* stats, int B, int N, float eps, hipStream_t st);
// {mean, rstd} from the per-wave partial sums the stream GEMM wrote (chunks of 32 tokens)
int launch_gn_final32(const double* partial, float* stats, int B, int N, float eps, hipStream_t st);
int launch_gn_apply_add_blk(const float* y_blk, const float* stats, const float* gamma, const float* beta, const float* coarse_blk,
                            float* out_blk, int B, int hf, int wf, int hc, int wc, hipStream_t st);
int launch_gn_apply_nchw_blk(const float* y_blk, const float* stats, const float* gamma, const float* beta, float* out, int B, int N,
                             hipStream_t st);
// gn_partial (optional, used when h * w % 32 == 0): the merged map's GroupNorm partial sums in k_gn_final32's format, fused
int launch_msm_sum_blk(float* y0, const float* const* yl, const int* lh, const int* lw, int B, int h, int w, int align, hipStream_t st,
                       double* gn_partial = nullptr);
// launch_gn_final32 for up to four maps (N[l] % 32 == 0 each) in one launch
int launch_gn_final32_multi(const double* const* partial, float* const* stats, const int* N, int n_maps, int B, float eps, hipStream_t st);
// 3x3 convolution as an implicit GEMM (FCNHeadWithTime, FPN: launch_b3_sgemm with conv_h > 0): weights packed tap-major
int launch_pack_conv3x3_scaled(const float* w, const float* scale, float* out, int cout, int cin, hipStream_t st);
int launch_fcn_fold(const float* bn_w, const float* bn_b, const float* bn_mean, const float* bn_var, float bn_eps,
                    const float* conv_bias, const float* film, float* scale, float* shift, hipStream_t st);
// out[b][k][n] = (1/div) * sum_ri prob[(b*r+ri)*N + n][k]
// frag_nch > 0: prob is fragment-major as the layer kernel's seg tails write it (per 32-token group frag_nch * 2048 floats)
int launch_finalize_nchw(const float* prob, int ldl, float* out, int B, int r, int N, int K, float div,
                         hipStream_t st, int frag_nch = 0);
// depth
int launch_feat_depth(const float* xproj, const float* wm, const float* d, float* q, int B, int r, int N,
                      hipStream_t st);
struct DepthUpdateArgs {
  const float* taps;   // (M, 32) 9 per-tap partial dots of conv_depth
  float bias;
  const float* bias_ptr;
  float* depth_t;      // (M) in/out noisy depth
  float* pred;         // (M) metric depth prediction out
  int B_r, h, w;
  float min_depth, max_depth, bit_scale, eps_depth;
  int scale_up;             // depth = sigmoid(s) * eps_depth instead of relu(s) + eps_depth
  ddp_step st;
};
int launch_depth_update(const DepthUpdateArgs& a, hipStream_t st);
int launch_mean_r(const float* pred, float* out, int B, int r, int N, hipStream_t st);
// the depth step head of the chain path WITHOUT a GEMM (ddp_kernels.hip: k_depth_head): layer 0's value map and sample table from their
// loop-invariant parts + a rank-1 term in the noisy depth, and - upd != nullptr - the previous step's DDIM update in front
struct DepthHeadArgs {
  const float* rvpad;       // zero-padded map of W_v (W_x x + b) + b_v, loop invariant (the layout of v_out)
  const float* rs;          // (M, 96) W_cat (W_x x + b), raw
  const float* wv;          // (256) W_v w_m
  const float* ws;          // (96)  W_cat w_m
  const float *py, *px;     // layer 0's positional tables (bias folded in)
  float* dvec;              // (M) noisy depth: read; with upd also written
  float* v_out;
  float* samp_out;          // head-major [head][M][12]
  int R, h, w;
  const DepthUpdateArgs* upd;
};
int launch_depth_head(const DepthHeadArgs& a, hipStream_t st);
// bev
struct BevGeom {
  int h, w, hh, wh;
  float in_min[2], in_max[2], out_first[2], out_step[2];
};
int launch_bev_resample(const float* feat, float* out, int R, const BevGeom& g, hipStream_t st);
struct BevUpdateArgs {
  const float* logits;  // (R*Nh, 32) raw conv_seg
  int num_classes;
  const float* emb;     // (K+1, 256) raw embedding table
  float* mask;          // (R*N, 256)
  float* prob;          // (R*Nh, 32) accumulated sigmoid
  int first;            // prob = (first) else +=
  int R;
  BevGeom g;
  float threshold, bit_scale;
  ddp_step st;
};
int launch_bev_update(const BevUpdateArgs& a, hipStream_t st);
// the bev sampler's u chain (ddp_kernels.hip: k_bev_q): q (fragment-major, R maps of hh x wh) = rx[map / r] + resample(u[map]);
// the 2^K x0 vectors of a pixel; u <- ua u + uc T[code at the pixel's nearest head-grid source]
int launch_bev_q(const float* u, const float* rx, float* q_blk, int R, int r, const BevGeom& g, hipStream_t st);
int launch_build_bev_lut(const float* emb, float* lut, int K, float bit_scale, hipStream_t st);
int launch_bev_u_update(float* u, const unsigned char* code, const float* tlut, int R, const BevGeom& g, float ua, float uc,
                        hipStream_t st);

}  // namespace ddp
