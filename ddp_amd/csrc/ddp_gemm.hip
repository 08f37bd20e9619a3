// ddp_gemm.hip - instantiations / launchers of the fp32-MFMA token GEMM (gemm_f32.h).
#include <stdlib.h>

#include "ddp_internal.h"
#include "gemm_f32.h"

namespace ddp {

unsigned long long* const g_gemm_dbg = nullptr;   // per-block cycle stamps of k_gemm_tok (DDP_STAMP): off

namespace {

template <int NT, bool A_BLK, int TAG, class Epi>
int launch_gemm(const GemmArgs& ga_in, const Epi& epi, hipStream_t st) {
  GemmArgs ga = ga_in;
  if (ga.M <= 0) return DDP_OK;
  if (ga.K % GEMM_BK != 0 || (!A_BLK && (ga.lda & 3)) || (ga.ldw & 3)) {
    set_error("gemm: K=%d must be a multiple of %d and lda/ldw multiples of 4", ga.K, GEMM_BK);
    return DDP_E_BADCFG;
  }
  ga.n_tiles_n = (ga.N + NT * 32 - 1) / (NT * 32);
  constexpr size_t lds = gemm_lds_bytes<NT, Epi>();
  static LdsAttrOnce attr;        // per instantiation and device
  attr.ensure(reinterpret_cast<const void*>(&k_gemm_tok<NT, A_BLK, Epi, TAG>), int(lds));
  prof_begin(TAG, st);
  hipLaunchKernelGGL((k_gemm_tok<NT, A_BLK, Epi, TAG>), dim3(gemm_grid(ga.M, ga.n_tiles_n)), dim3(GEMM_THREADS), lds, st,
                     ga, epi, g_gemm_dbg);
  prof_end(TAG, st);
  return check_launch("k_gemm_tok");
}

GemmArgs make_args(const float* A, int lda, const float* W, int ldw, const float* bias, int M, int N, int K) {
  GemmArgs ga;
  ga.A = A;
  ga.lda = lda;
  ga.W = W;
  ga.ldw = ldw;
  ga.M = M;
  ga.N = N;
  ga.K = K;
  ga.n_tiles_n = 1;
  ga.acc_bias = bias;
  return ga;
}

}  // namespace

int launch_linear(const float* A, int lda, bool a_blk, const float* W, int ldw, const float* bias, const float* add,
                  int ld_add, int rn, int n_tok, float* out, int ldo, int M, int N, int K, int gelu, hipStream_t st,
                  int tag) {
  EpiRow e;
  e.add = add;
  e.ld_add = ld_add;
  e.rn = rn;
  e.n_tok = n_tok;
  e.out = out;
  e.ldo = ldo;
  e.n_valid = N;
  e.gelu = gelu;
  if (ldo & 3) {
    set_error("linear: N=%d, ldo=%d must be a multiple of 4", N, ldo);
    return DDP_E_BADCFG;
  }
  const GemmArgs ga = make_args(A, lda, W, ldw, bias, M, N, K);
  if (tag == TAG_XPROJ && N > 160 && !a_blk) return launch_gemm<8, false, TAG_XPROJ>(ga, e, st);
  if (tag == TAG_VALUE && N > 160 && a_blk) return launch_gemm<8, true, TAG_VALUE>(ga, e, st);
  if (tag == TAG_HEAD && a_blk) {
    if (N <= 32) return launch_gemm<1, true, TAG_HEAD>(ga, e, st);
    if (N <= 96) return launch_gemm<3, true, TAG_HEAD>(ga, e, st);
    if (N <= 160) return launch_gemm<5, true, TAG_HEAD>(ga, e, st);
    return launch_gemm<8, true, TAG_HEAD>(ga, e, st);
  }
  if (a_blk) {
    set_error("linear: fragment-major input is only instantiated for the value / head call sites");
    return DDP_E_BADCFG;
  }
  if (N <= 32) return launch_gemm<1, false, TAG_GENERIC>(ga, e, st);
  if (N <= 96) return launch_gemm<3, false, TAG_GENERIC>(ga, e, st);
  if (N <= 160) return launch_gemm<5, false, TAG_GENERIC>(ga, e, st);
  return launch_gemm<8, false, TAG_GENERIC>(ga, e, st);
}

int launch_linear_blk(const float* A, int lda, bool a_blk, const float* W, int ldw, const float* bias,
                      const float* add, int ld_add, int rn, int n_tok, float* out_blk, int M, int N, int K, int gelu,
                      hipStream_t st) {
  if (N % 256) {
    set_error("linear_blk: N=%d must be a multiple of 256", N);
    return DDP_E_BADCFG;
  }
  EpiBlk e;
  e.add = add;
  e.ld_add = ld_add;
  e.rn = rn;
  e.n_tok = n_tok;
  e.out = out_blk;
  e.c_out = N;
  e.gelu = gelu;
  const GemmArgs ga = make_args(A, lda, W, ldw, bias, M, N, K);
  if (a_blk) return launch_gemm<8, true, TAG_FC1>(ga, e, st);
  return launch_gemm<8, false, TAG_FEAT>(ga, e, st);
}

int launch_linear_res_ln_blk(const float* A, int lda, bool a_blk, const float* W, int ldw, const float* bias,
                             const float* res_blk, const float* ga_aff, const float* be_aff, float* out_blk, int M,
                             int K, hipStream_t st) {
  EpiResLNBlk e;
  e.res = res_blk;
  e.ga = ga_aff;
  e.be = be_aff;
  e.out = out_blk;
  const GemmArgs ga = make_args(A, lda, W, ldw, bias, M, 256, K);
  if (a_blk) return launch_gemm<8, true, TAG_FC2_LN>(ga, e, st);
  return launch_gemm<8, false, TAG_OUTPROJ_LN>(ga, e, st);
}

int launch_linear_samp(const float* A_blk, const float* Wcat, const float* py, const float* px, int n_tok, int w,
                       float* out, int M, hipStream_t st) {
  EpiSamp e;
  e.py = py;
  e.px = px;
  e.n_tok = n_tok;
  e.w = w;
  e.out = out;
  const GemmArgs ga = make_args(A_blk, 256, Wcat, 256, nullptr, M, 96, 256);
  return launch_gemm<3, true, TAG_SAMP>(ga, e, st);
}

}  // namespace ddp

