// ddp_gemm_bf16.hip - launchers of the bf16x3-split ("fp32-equivalent") token GEMM (gemm_bf16x3.h).
#include <cstring>

#include "ddp_internal.h"
#include <stdlib.h>

#include "layer_bf16x3.h"
#include "gemm_bf16x3.h"

namespace ddp {

namespace {

template <int NT, int TAG, class Epi>
int launch_b3(const b3::Args& a_in, const Epi& epi, hipStream_t st) {
  b3::Args ga = a_in;
  if (ga.M <= 0) return DDP_OK;
  if (ga.K % b3::BK != 0) {
    set_error("gemm_bf16x3: K=%d must be a multiple of %d", ga.K, b3::BK);
    return DDP_E_BADCFG;
  }
  ga.n_tiles_n = (ga.N + NT * 32 - 1) / (NT * 32);
  constexpr size_t lds = b3::lds_bytes<NT, Epi>();
  static LdsAttrOnce attr;
  attr.ensure(reinterpret_cast<const void*>(&b3::k_gemm<NT, Epi, TAG>), int(lds));
  prof_begin(TAG, st);
  hipLaunchKernelGGL((b3::k_gemm<NT, Epi, TAG>), dim3(b3::grid(ga.M, ga.n_tiles_n)), dim3(b3::THREADS), lds, st, ga, epi);
  prof_end(TAG, st);
  return check_launch("b3::k_gemm");
}

b3::Args make_b3(const unsigned short* A_sb, const SplitW& w, const float* bias, int M, int N, int K) {
  b3::Args a;
  a.A = A_sb;
  a.Wp = w.p;
  a.w_comp_stride = w.comp_stride;
  a.M = M;
  a.N = N;
  a.K = K;
  a.n_tiles_n = 1;
  a.acc_bias = bias;
  return a;
}

}  // namespace

int launch_b3_linear(const unsigned short* A_sb, const SplitW& w, const float* bias, const float* add, int ld_add,
                     int rn, int n_tok, float* out, int ldo, int M, int N, int K, hipStream_t st, int tag) {
  EpiRow e;
  e.add = add;
  e.ld_add = ld_add;
  e.rn = rn;
  e.n_tok = n_tok;
  e.out = out;
  e.ldo = ldo;
  e.n_valid = N;
  e.gelu = 0;
  if (ldo & 3) {
    set_error("b3 linear: ldo=%d must be a multiple of 4", ldo);
    return DDP_E_BADCFG;
  }
  const b3::Args ga = make_b3(A_sb, w, bias, M, N, K);
  if (N <= 32) return launch_b3<1, TAG_HEAD>(ga, e, st);
  if (N <= 96) return launch_b3<3, TAG_HEAD>(ga, e, st);
  if (N <= 160) return launch_b3<5, TAG_HEAD>(ga, e, st);
  if (tag == TAG_VALUE) return launch_b3<8, TAG_VALUE>(ga, e, st);
  if (tag == TAG_XPROJ) return launch_b3<8, TAG_XPROJ>(ga, e, st);
  return launch_b3<8, TAG_HEAD>(ga, e, st);
}



int launch_b3_linear_sb(const unsigned short* A_sb, const SplitW& w, const float* bias, const float* add, int ld_add,
                        int rn, int n_tok, unsigned short* out_sb, float* out_f32_blk, int M, int N, int K, int gelu,
                        hipStream_t st, int tag) {
  if (N % 256) {
    set_error("b3 linear_sb: N=%d must be a multiple of 256", N);
    return DDP_E_BADCFG;
  }
  b3::EpiSB e;
  e.add = add;
  e.ld_add = ld_add;
  e.rn = rn;
  e.n_tok = n_tok;
  e.out_sb = out_sb;
  e.out_f32 = out_f32_blk;
  e.c_out = N;
  e.gelu = gelu;
  const b3::Args ga = make_b3(A_sb, w, bias, M, N, K);
  if (tag == TAG_FC1) return launch_b3<8, TAG_FC1>(ga, e, st);
  return launch_b3<8, TAG_FEAT>(ga, e, st);
}

int launch_b3_linear_res_ln(const unsigned short* A_sb, const SplitW& w, const float* bias, const float* res_blk,
                            const unsigned short* res_sb, const float* ga_aff, const float* be_aff, float* out_f32_blk,
                            unsigned short* out_sb, int M, int K, hipStream_t st, int tag) {
  b3::EpiResLNSB e;
  e.res = res_blk;
  e.res_sb = res_sb;
  e.ga = ga_aff;
  e.be = be_aff;
  e.out_f32 = out_f32_blk;
  e.out_sb = out_sb;
  const b3::Args ga = make_b3(A_sb, w, bias, M, 256, K);
  if (tag == TAG_FC2_LN) return launch_b3<8, TAG_FC2_LN>(ga, e, st);
  return launch_b3<8, TAG_OUTPROJ_LN>(ga, e, st);
}

int launch_b3_linear_samp(const unsigned short* A_sb, const SplitW& wcat, const float* py, const float* px, int n_tok,
                          int w, float* out, int M, hipStream_t st) {
  EpiSamp e;
  e.py = py;
  e.px = px;
  e.n_tok = n_tok;
  e.w = w;
  e.out = out;
  const b3::Args ga = make_b3(A_sb, wcat, nullptr, M, 96, 256);
  return launch_b3<3, TAG_SAMP>(ga, e, st);
}

#ifdef DDP_LYR_STAMP
// debug builds only (not declared in include/ddp_mi355x.h, not part of the product library): where the layer kernel's
// cycle stamps go - scripts/stamp_layer.py
static unsigned long long* g_lyr_stamps = nullptr;
extern "C" __attribute__((visibility("default"))) void ddp_debug_set_layer_stamps(void* d_buf) {
  g_lyr_stamps = static_cast<unsigned long long*>(d_buf);
}
#endif

int launch_b3_layer(const LayerLaunch& a, hipStream_t st) {
  if (a.M <= 0) return DDP_OK;
  b3::LayerArgs la;
  memset(&la, 0, sizeof(la));
  la.S = a.S;
  la.Sf = a.Sf;
#if DDP_S_F32
  if (!a.Sf) {
    set_error("b3::k_layer: this build takes the attention output as fp32 fragments (Sf)");
    return DDP_E_NULL;
  }
#endif
  la.Q = a.Q;
  la.Q_sb = a.Q_sb;
  la.stream = a.stream;
  la.bias_ext = a.bias_ext;
  la.bo = a.bo;
  la.ga0 = a.ga0;
  la.be0 = a.be0;
  la.b2 = a.b2;
  la.ga1 = a.ga1;
  la.be1 = a.be1;
  la.M = a.M;
  la.has_next = a.has_next;
  la.v_out = a.v_out;
  la.samp_out = a.samp_out;
  la.py = a.py;
  la.px = a.px;
  la.n_tok = a.n_tok;
  la.w = a.w;
#ifdef DDP_LYR_STAMP
  la.stamps = g_lyr_stamps;
#endif
  static LdsAttrOnce attr, attr_nt;
  attr.ensure(reinterpret_cast<const void*>(&b3::k_layer<TAG_FC2_LN>), int(b3::LYR_LDS_B));
  attr_nt.ensure(reinterpret_cast<const void*>(&b3::k_layer<TAG_FC2_LN, 0, 0, true>), int(b3::LYR_LDS_B));
  const int n_cu = cu_count();
  const int tiles = (a.M + b3::LYR_BM - 1) / b3::LYR_BM;
  const int grid = tiles < n_cu ? tiles : n_cu;       // persistent: one block per CU walks tiles blockIdx, +grid, ...
  if (a.res_f) {
    // layer 0 of a depth step on the chain path: the residual q = res_f + wm * dvec is formed in the kernel (k_layer MODE 10)
    if (!a.wm || !a.dvec) {
      set_error("b3::k_layer (depth layer 0): wm / dvec missing");
      return DDP_E_NULL;
    }
    la.res = a.res_f;
    la.seg_bias = a.wm;
    la.dvec = a.dvec;
    static LdsAttrOnce attr10, attr10_nt;
    attr10.ensure(reinterpret_cast<const void*>(&b3::k_layer<TAG_FC2_LN, 10>), int(b3::LYR_LDS_B));
    attr10_nt.ensure(reinterpret_cast<const void*>(&b3::k_layer<TAG_FC2_LN, 10, 0, true>), int(b3::LYR_LDS_B));
    prof_begin(TAG_FC2_LN, st);
    if (a.M >= b3::LYR_NT_MIN_TOKENS)
      hipLaunchKernelGGL((b3::k_layer<TAG_FC2_LN, 10, 0, true>), dim3(grid), dim3(b3::LYR_THREADS), b3::LYR_LDS_B, st, la);
    else
      hipLaunchKernelGGL((b3::k_layer<TAG_FC2_LN, 10>), dim3(grid), dim3(b3::LYR_THREADS), b3::LYR_LDS_B, st, la);
    prof_end(TAG_FC2_LN, st);
    return check_launch("b3::k_layer (depth layer 0)");
  }
  prof_begin(TAG_FC2_LN, st);
  // activation tensors too large to survive in L2 / MALL until the next kernel reads them: non-temporal streams (layer_bf16x3.h)
  if (a.M >= b3::LYR_NT_MIN_TOKENS)
    hipLaunchKernelGGL((b3::k_layer<TAG_FC2_LN, 0, 0, true>), dim3(grid), dim3(b3::LYR_THREADS), b3::LYR_LDS_B, st, la);
  else
    hipLaunchKernelGGL((b3::k_layer<TAG_FC2_LN>), dim3(grid), dim3(b3::LYR_THREADS), b3::LYR_LDS_B, st, la);
  prof_end(TAG_FC2_LN, st);
  return check_launch("b3::k_layer");
}
namespace {
template <int MODE, int NCH, bool FORCE>
int launch_tail_tf(const b3::LayerArgs& la, int grid, hipStream_t st) {
  static LdsAttrOnce attr;
  attr.ensure(reinterpret_cast<const void*>(&b3::k_layer<TAG_HEAD, MODE, NCH, false, FORCE>), int(b3::LYR_LDS_B));
  prof_begin(TAG_HEAD, st);
  hipLaunchKernelGGL((b3::k_layer<TAG_HEAD, MODE, NCH, false, FORCE>), dim3(grid), dim3(b3::LYR_THREADS), b3::LYR_LDS_B, st, la);
  prof_end(TAG_HEAD, st);
  return check_launch(MODE == 4 ? "b3::k_layer (seg tail + next step head)" : "b3::k_layer (seg tail)");
}
// la.x0_force (DDP_FLAG_FORCE_X0, a test instrument) selects the teacher-forcing instantiation; the product path runs <.., false>
template <int MODE, int NCH>
int launch_tail_t(const b3::LayerArgs& la, int grid, hipStream_t st) {
  return la.x0_force ? launch_tail_tf<MODE, NCH, true>(la, grid, st) : launch_tail_tf<MODE, NCH, false>(la, grid, st);
}
}  // namespace

int launch_b3_tail(const TailLaunch& a, hipStream_t st) {
  if (a.M <= 0) return DDP_OK;
  b3::LayerArgs la;
  memset(&la, 0, sizeof(la));
  la.Q = a.Q;
  la.stream = a.stream;
  la.bias_ext = a.bias_ext;
  la.M = a.M;
  la.lut = a.lut;
  la.prob = a.prob;
  la.mask_sb = a.mask_sb;
  la.x0_idx = a.x0_idx;
  la.x0_force = a.x0_force;
  la.num_classes = a.num_classes;
  la.ldl = a.ldl;
  la.prob_mode = a.prob_mode;
  la.alpha = a.alpha;
  la.sigma = a.sigma;
  la.alpha_next = a.alpha_next;
  la.sigma_next = a.sigma_next;
  const int n_cu = cu_count();
  const int tiles = (a.M + b3::LYR_BM - 1) / b3::LYR_BM;
  const int grid = tiles < n_cu ? tiles : n_cu;
  const int nch = (a.num_classes + 63) / 64;
  if (a.fuse_next) {
    // u' = ua u + uc T[argmax]:  m' = alpha' x0 + sigma' (m - alpha x0) / max(sigma, 1e-8)  (ddp.py:238-239) under W_m
    la.ua = a.sigma_next / (a.sigma > 1e-8f ? a.sigma : 1e-8f);
    la.uc = a.alpha_next - a.alpha * la.ua;
    la.ubuf = a.ubuf;
    la.tlut = a.tlut;
    la.res = a.res;
    la.res_rn = a.res_rn;
    la.res_frag = a.res_frag;
    la.has_next = 1;
    la.v_out = a.v_out;
    la.samp_out = a.samp_out;
    la.py = a.py;
    la.px = a.px;
    la.n_tok = a.n_tok;
    la.w = a.w;
    la.mask_sb = nullptr;
    switch (nch) {
      case 1: return launch_tail_t<4, 1>(la, grid, st);
      case 2: return launch_tail_t<4, 2>(la, grid, st);
      case 3: return launch_tail_t<4, 3>(la, grid, st);
      case 4: return launch_tail_t<4, 4>(la, grid, st);
      default: set_error("seg tail: %d classes (1..256)", a.num_classes); return DDP_E_BADCFG;
    }
  }
  switch (nch) {
    case 1: return launch_tail_t<1, 1>(la, grid, st);
    case 2: return launch_tail_t<1, 2>(la, grid, st);
    case 3: return launch_tail_t<1, 3>(la, grid, st);
    case 4: return launch_tail_t<1, 4>(la, grid, st);
    default: set_error("seg tail: %d classes (1..256)", a.num_classes); return DDP_E_BADCFG;
  }
}

int launch_b3_prologue(const PrologueLaunch& a, hipStream_t st) {
  if (a.M <= 0) return DDP_OK;
  b3::LayerArgs la;
  memset(&la, 0, sizeof(la));
  la.S = a.mask_sb;
  la.Q = a.Q;
  la.stream = a.stream;
  la.bias_ext = a.bias_ext;
  la.res = a.res;
  la.res_rn = a.res_rn;
  la.ubuf = a.ubuf;
  la.M = a.M;
  la.has_next = 1;
  la.v_out = a.v_out;
  la.samp_out = a.samp_out;
  la.py = a.py;
  la.px = a.px;
  la.n_tok = a.n_tok;
  la.w = a.w;
  static LdsAttrOnce attr;
  attr.ensure(reinterpret_cast<const void*>(&b3::k_layer<TAG_FEAT, 2>), int(b3::LYR_LDS_B));
  const int n_cu = cu_count();
  const int tiles = (a.M + b3::LYR_BM - 1) / b3::LYR_BM;
  const int grid = tiles < n_cu ? tiles : n_cu;
  prof_begin(TAG_FEAT, st);
  hipLaunchKernelGGL((b3::k_layer<TAG_FEAT, 2>), dim3(grid), dim3(b3::LYR_THREADS), b3::LYR_LDS_B, st, la);
  prof_end(TAG_FEAT, st);
  return check_launch("b3::k_layer (step prologue)");
}

int launch_b3_l0proj(const L0ProjLaunch& a, hipStream_t st) {
  if (a.M <= 0) return DDP_OK;
  b3::LayerArgs la;
  memset(&la, 0, sizeof(la));
  la.Q = a.Q;
  la.stream = a.stream;
  la.bias_ext = a.bias_ext;
  la.res = a.res;
  la.res_rn = a.res_rn;
  la.bo = a.wm;
  la.dvec = a.dvec;
  la.M = a.M;
  la.has_next = 1;
  la.v_out = a.v_out;
  la.samp_out = a.samp_out;
  la.py = a.py;
  la.px = a.px;
  la.n_tok = a.n_tok;
  la.w = a.w;
  if (a.upd) {
    const DepthUpdateArgs& u = *a.upd;
    if (!a.res || !a.dvec || u.depth_t != a.dvec || u.B_r * u.h * u.w != a.M || u.h * u.w != a.n_tok || u.w != a.w) {
      set_error("layer 0 projections: the fused depth update needs the depth head's own map");
      return DDP_E_BADCFG;
    }
    la.dtaps = u.taps;
    la.dbias = u.bias_ptr;
    la.dvec_rw = u.depth_t;
    la.d_min = u.min_depth;
    la.d_max = u.max_depth;
    la.d_bit = u.bit_scale;
    la.d_eps = u.eps_depth;
    la.d_scale_up = u.scale_up;
    la.d_sig = u.st.sigma;
    la.d_alpha = u.st.alpha;
    la.d_alpha_next = u.st.alpha_next;
    la.d_sigma_next = u.st.sigma_next;
  }
  static LdsAttrOnce attr;
  attr.ensure(reinterpret_cast<const void*>(&b3::k_layer<TAG_VALUE, 3>), int(b3::LYR_LDS_B));
  const int n_cu = cu_count();
  const int tiles = (a.M + b3::LYR_BM - 1) / b3::LYR_BM;
  const int grid = tiles < n_cu ? tiles : n_cu;
  prof_begin(TAG_VALUE, st);
  hipLaunchKernelGGL((b3::k_layer<TAG_VALUE, 3>), dim3(grid), dim3(b3::LYR_THREADS), b3::LYR_LDS_B, st, la);
  prof_end(TAG_VALUE, st);
  return check_launch("b3::k_layer (layer 0 projections)");
}

// stream GEMM of the necks (k_layer MODE 5): up to four problems (out = act(A . W^T), 256 outputs each) in ONE persistent launch;
// the tiles are numbered through the problems in the order given (long tiles first fills the tail best)
int launch_b3_sgemm(const SgemmProblem* pr, int n, int act, int conv_dil, hipStream_t st) {
  if (n < 1 || n > 4) {
    set_error("b3 stream GEMM: %d problems (1..4)", n);
    return DDP_E_BADCFG;
  }
  b3::LayerArgs la;
  memset(&la, 0, sizeof(la));
  int tiles = 0;
  for (int i = 0; i < 4; ++i) {
    b3::LayerArgs::GemmProblem& g = la.gp[i];
    if (i >= n) {
      g.tile0 = 0x7fffffff;
      g.ns = 2;
      continue;
    }
    if (pr[i].M <= 0) {
      set_error("b3 stream GEMM: problem %d is empty", i);
      return DDP_E_BADCFG;
    }
    if (pr[i].ns < 2 || (pr[i].conv_h > 0 && pr[i].ns != 72)) {
      set_error("b3 stream GEMM: %d stages (K = %d) unsupported", pr[i].ns, pr[i].ns * 32);
      return DDP_E_BADCFG;
    }
    g.A = pr[i].A;
    g.out = pr[i].out;
    g.stream = pr[i].stream;
    g.M = pr[i].M;
    g.ns = pr[i].ns;
    g.tile0 = tiles;
    g.conv_h = pr[i].conv_h;
    g.conv_w = pr[i].conv_w;
    g.bias = pr[i].bias;
    g.gn_partial = pr[i].gn_partial;
    g.gn_N = pr[i].gn_N;
    g.nchw_N = pr[i].nchw_N;
    if (g.nchw_N > 0 && (pr[i].conv_h > 0 || pr[i].M % g.nchw_N || ((size_t(pr[i].M) * 32 * pr[i].ns) >> 32))) {
      set_error("b3 stream GEMM: NCHW operand needs a plain (1x1) problem of whole images below 2^32 elements");
      return DDP_E_BADCFG;
    }
    if (g.gn_partial && (g.gn_N < 32 || g.gn_N % 32 || pr[i].M % g.gn_N)) {
      set_error("b3 stream GEMM: fused GroupNorm statistics need tokens per image (%d) to be a multiple of 32", g.gn_N);
      return DDP_E_BADCFG;
    }
    tiles += (pr[i].M + b3::LYR_BM - 1) / b3::LYR_BM;
  }
  la.g_tiles = tiles;
  la.M = tiles * b3::LYR_BM;
  la.g_act = act;
  la.conv_dil = conv_dil;
  static LdsAttrOnce attr;
  attr.ensure(reinterpret_cast<const void*>(&b3::k_layer<TAG_GENERIC, 5>), int(b3::LYR_LDS_B));
  const int n_cu = cu_count();
  const int grid = tiles < n_cu ? tiles : n_cu;
  hipLaunchKernelGGL((b3::k_layer<TAG_GENERIC, 5>), dim3(grid), dim3(b3::LYR_THREADS), b3::LYR_LDS_B, st, la);
  return check_launch("b3::k_layer (stream GEMM)");
}
size_t b3_stage_bytes() { return size_t(b3::LYR_STAGE_B); }

size_t b3_prologue_stream_bytes() { return size_t(b3::LYR_ST_OUT + b3::LYR_ST_NEXT) * b3::LYR_STAGE_B; }
size_t b3_layer_stream_bytes() { return size_t(b3::LYR_STAGES) * b3::LYR_STAGE_B; }
int b3_layer_bias_floats() { return b3::LYR_BIAS_N; }

}  // namespace ddp
