// ddp_layer_tail.hip - the last decoder layer of a step fused with that step's segmentation tail: k_layer MODE 6
// (layer_bf16x3.h).  Its own translation unit: each instantiation is the whole layer kernel (FFN included) plus a tail, and the
// four files of the library compile in parallel.
//
// Replaces, per step: b3::k_layer<7,0> for layer L-1 (utils/transformer.py:317-419) + b3::k_layer<8,4,NCH> (conv_seg
// decode_head.py:133, argmax / x0 projection / DDIM update segmentors/ddp.py:235-239, softmax accumulation :241-242, the next
// step's concat-conv :223-224 and layer 0's projections) - or + b3::k_layer<8,1,NCH> after the last step.
#include <cstring>

#include "ddp_internal.h"

#include "layer_bf16x3.h"

namespace ddp {

namespace {
template <int NCH, bool NT, bool FORCE>
int launch_lt(const b3::LayerArgs& la, int grid, hipStream_t st) {
  static LdsAttrOnce attr;
  attr.ensure(reinterpret_cast<const void*>(&b3::k_layer<TAG_LAYER_TAIL, 6, NCH, NT, FORCE>), int(b3::LYR_LDS_B));
  prof_begin(TAG_LAYER_TAIL, st);
  hipLaunchKernelGGL((b3::k_layer<TAG_LAYER_TAIL, 6, NCH, NT, FORCE>), dim3(grid), dim3(b3::LYR_THREADS), b3::LYR_LDS_B, st, la);
  prof_end(TAG_LAYER_TAIL, st);
  return check_launch("b3::k_layer (last layer + seg tail)");
}
// bev (MODE 8) / depth (MODE 9) tails: depth/depth/models/decode_heads/decode_head.py:264-269 (conv_depth);
// bev/mmdet3d/models/heads/segm/deformable_head_with_time.py:213,235 (conv_seg + sigmoid), fusion_models/ddp.py:290-293 (threshold)
template <int MODE, bool NT>
int launch_lt_other(const b3::LayerArgs& la, int grid, hipStream_t st) {
  static LdsAttrOnce attr;
  attr.ensure(reinterpret_cast<const void*>(&b3::k_layer<TAG_LAYER_TAIL, MODE, 1, NT, false>), int(b3::LYR_LDS_B));
  prof_begin(TAG_LAYER_TAIL, st);
  hipLaunchKernelGGL((b3::k_layer<TAG_LAYER_TAIL, MODE, 1, NT, false>), dim3(grid), dim3(b3::LYR_THREADS), b3::LYR_LDS_B, st, la);
  prof_end(TAG_LAYER_TAIL, st);
  return check_launch(MODE == 8 ? "b3::k_layer (last layer + bev tail)" : "b3::k_layer (last layer + depth tail)");
}
// teacher forcing (DDP_FLAG_FORCE_X0) is a test instrument: its instantiations exist without the non-temporal variant
template <int NCH>
int launch_lt_n(const b3::LayerArgs& la, int grid, hipStream_t st) {
  if (la.x0_force) return launch_lt<NCH, false, true>(la, grid, st);
  if (la.M >= b3::LYR_NT_MIN_TOKENS) return launch_lt<NCH, true, false>(la, grid, st);
  return launch_lt<NCH, false, false>(la, grid, st);
}
}  // namespace

bool b3_layer_tail_supported(int num_classes) { return num_classes >= 1 && num_classes <= 256; }

int launch_b3_layer_tail(const LayerLaunch& l, const TailLaunch& t, const unsigned char* stream, const float* bias_ext,
                         const float* seg_bias, hipStream_t st) {
  if (l.M <= 0) return DDP_OK;
  if (!b3_layer_tail_supported(t.num_classes) || t.mask_sb || l.M != t.M || l.Q != t.Q || t.kind < 0 || t.kind > 2 ||
      (t.kind != 0 && (t.fuse_next || t.num_classes > 32 || t.x0_force || (t.x0_idx && t.num_classes > 8)))) {
    set_error("layer + tail kernel: unsupported call (kind %d, classes %d, legacy noisy map %d)", t.kind, t.num_classes, t.mask_sb ? 1 : 0);
    return DDP_E_BADCFG;
  }
#if DDP_S_F32
  if (!l.Sf) {
    set_error("b3::k_layer: this build takes the attention output as fp32 fragments (Sf)");
    return DDP_E_NULL;
  }
#endif
  b3::LayerArgs la;
  memset(&la, 0, sizeof(la));
  // ---- the layer (MODE 0's arguments)
  la.S = l.S;
  la.Sf = l.Sf;
  la.Q = l.Q;
  la.stream = stream;
  la.bias_ext = bias_ext;
  la.seg_bias = seg_bias;
  la.bo = l.bo;
  la.ga0 = l.ga0;
  la.be0 = l.be0;
  la.b2 = l.b2;
  la.ga1 = l.ga1;
  la.be1 = l.be1;
  la.M = l.M;
  // ---- the tail (MODE 1 / MODE 4's arguments)
  la.lut = t.lut;
  la.prob = t.prob;
  la.x0_idx = t.x0_idx;
  la.x0_force = t.x0_force;
  la.num_classes = t.num_classes;
  la.ldl = t.ldl;
  la.prob_mode = t.prob_mode;
  la.threshold = t.threshold;
  la.alpha = t.alpha;
  la.sigma = t.sigma;
  la.alpha_next = t.alpha_next;
  la.sigma_next = t.sigma_next;
  la.has_next = t.fuse_next ? 1 : 0;
  if (t.fuse_next) {
    // u' = ua u + uc T[argmax]:  m' = alpha' x0 + sigma' (m - alpha x0) / max(sigma, 1e-8)  (ddp.py:238-239) under W_m
    la.ua = t.sigma_next / (t.sigma > 1e-8f ? t.sigma : 1e-8f);
    la.uc = t.alpha_next - t.alpha * la.ua;
    la.ubuf = t.ubuf;
    la.tlut = t.tlut;
    la.res = t.res;
    la.res_rn = t.res_rn;
    la.res_frag = t.res_frag;
    la.v_out = t.v_out;
    la.samp_out = t.samp_out;
    la.py = t.py;
    la.px = t.px;
    la.n_tok = t.n_tok;
    la.w = t.w;
  }
  const int n_cu = cu_count();
  const int tiles = (l.M + b3::LYR_BM - 1) / b3::LYR_BM;
  const int grid = tiles < n_cu ? tiles : n_cu;
  const bool nt = l.M >= b3::LYR_NT_MIN_TOKENS;
  if (t.kind == 1) return nt ? launch_lt_other<8, true>(la, grid, st) : launch_lt_other<8, false>(la, grid, st);
  if (t.kind == 2) return nt ? launch_lt_other<9, true>(la, grid, st) : launch_lt_other<9, false>(la, grid, st);
  switch ((t.num_classes + 63) / 64) {
    case 1: return launch_lt_n<1>(la, grid, st);
    case 2: return launch_lt_n<2>(la, grid, st);
    case 3: return launch_lt_n<3>(la, grid, st);
    default: return launch_lt_n<4>(la, grid, st);
  }
}

int launch_b3_head_nchw(const PrologueLaunch& a, const float* nchw_noise, const float* nchw_x, const float* bias, hipStream_t st) {
  if (a.M <= 0) return DDP_OK;
  if (a.res_rn != 0 || !a.res || (size_t(a.M) * 256) >> 32) {
    set_error("first-step head from NCHW: one noisy map per image and < 2^32 elements per tensor");
    return DDP_E_BADCFG;
  }
  b3::LayerArgs la;
  memset(&la, 0, sizeof(la));
  la.nchw_noise = nchw_noise;
  la.nchw_x = nchw_x;
  la.bo = bias;
  la.Q = a.Q;
  la.stream = a.stream;
  la.bias_ext = a.bias_ext;
  la.res = a.res;
  la.res_frag = a.res_frag;
  la.ubuf = a.ubuf;
  la.M = a.M;
  la.has_next = 1;
  la.v_out = a.v_out;
  la.samp_out = a.samp_out;
  la.py = a.py;
  la.px = a.px;
  la.n_tok = a.n_tok;
  la.w = a.w;
  const int n_cu = cu_count();
  const int tiles = (a.M + b3::LYR_BM - 1) / b3::LYR_BM;
  const int grid = tiles < n_cu ? tiles : n_cu;
  static LdsAttrOnce attr, attr_nt;
  prof_begin(TAG_FEAT, st);
  if (a.M >= b3::LYR_NT_MIN_TOKENS) {
    attr_nt.ensure(reinterpret_cast<const void*>(&b3::k_layer<TAG_FEAT, 7, 0, true>), int(b3::LYR_LDS_B));
    hipLaunchKernelGGL((b3::k_layer<TAG_FEAT, 7, 0, true>), dim3(grid), dim3(b3::LYR_THREADS), b3::LYR_LDS_B, st, la);
  } else {
    attr.ensure(reinterpret_cast<const void*>(&b3::k_layer<TAG_FEAT, 7>), int(b3::LYR_LDS_B));
    hipLaunchKernelGGL((b3::k_layer<TAG_FEAT, 7>), dim3(grid), dim3(b3::LYR_THREADS), b3::LYR_LDS_B, st, la);
  }
  prof_end(TAG_FEAT, st);
  return check_launch("b3::k_layer (first step's head from NCHW)");
}

}  // namespace ddp
