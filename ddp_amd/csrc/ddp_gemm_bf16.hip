// ddp_gemm_bf16.hip - launchers of the bf16x3-split ("fp32-equivalent") token GEMM (gemm_bf16x3.h).
#include <cstring>

#include "ddp_internal.h"
#include <stdlib.h>

#include "ffn_bf16x3.h"
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
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&b3::k_gemm<NT, Epi, TAG>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    attr_done = true;
  }
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

namespace {
template <bool OUTPROJ>
int launch_ffn_t(const b3::FfnArgs& fa, const b3::EpiResLNSB& e, hipStream_t st) {
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&b3::k_ffn<b3::EpiResLNSB, TAG_FC2_LN, OUTPROJ>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, int(b3::FFN_LDS_B));
    attr_done = true;
  }
  const int grid = (fa.M + b3::FFN_BM - 1) / b3::FFN_BM;
  prof_begin(TAG_FC2_LN, st);
  hipLaunchKernelGGL((b3::k_ffn<b3::EpiResLNSB, TAG_FC2_LN, OUTPROJ>), dim3(grid), dim3(b3::FFN_THREADS), b3::FFN_LDS_B, st,
                     fa, e);
  prof_end(TAG_FC2_LN, st);
  return check_launch("b3::k_ffn");
}
}  // namespace

// out = FiLM(LN1(x + FFN(x))); x = X_sb, or - when S_sb is given - x = LN0(Q_sb + Wo . S_sb + bo) computed in-kernel
int launch_b3_ffn(const unsigned short* X_sb, const SplitW& w1, const SplitW& w2, const float* b1, const float* b2,
                  const float* ga_aff, const float* be_aff, unsigned short* out_sb, int M, hipStream_t st,
                  const unsigned short* S_sb, const unsigned short* Q_sb, const SplitW* wo, const float* bo,
                  const float* ga0, const float* be0) {
  if (M <= 0) return DDP_OK;
  b3::FfnArgs fa;
  memset(&fa, 0, sizeof(fa));
  fa.X = X_sb;
  fa.W1p = w1.p;
  fa.W2p = w2.p;
  fa.b1 = b1;
  fa.b2 = b2;
  fa.M = M;
  b3::EpiResLNSB e;
  e.res = nullptr;      // the residual is the kernel's own input
  e.res_sb = nullptr;
  e.ga = ga_aff;
  e.be = be_aff;
  e.out_f32 = nullptr;
  e.out_sb = out_sb;
  if (S_sb) {
    fa.S = S_sb;
    fa.Q = Q_sb;
    fa.Wop = wo->p;
    fa.bo = bo;
    fa.ga0 = ga0;
    fa.be0 = be0;
    return launch_ffn_t<true>(fa, e, st);
  }
  return launch_ffn_t<false>(fa, e, st);
}

bool b3_ffn_fused_enabled() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("DDP_FFN_FUSED");
    v = e ? atoi(e) : 1;
  }
  return v != 0;
}

}  // namespace ddp
