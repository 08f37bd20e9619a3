// ddp_gemm.hip - instantiations / launchers of the fp32-MFMA token GEMM (gemm_f32.h).
#include <stdlib.h>

#include "ddp_internal.h"
#include "gemm_f32.h"

namespace ddp {

unsigned long long* g_gemm_dbg = nullptr;   // probe: per-block cycle stamps (ddp_debug_set_stamps)

namespace {

template <int NT, int TAG, class Epi>
int launch_gemm(const float* A, int lda, const float* W, int ldw, int M, int N, int K, const Epi& epi,
                hipStream_t st) {
  if (M <= 0) return DDP_OK;
  if (K % GEMM_BK != 0 || (lda & 3) || (ldw & 3)) {
    set_error("gemm: K=%d must be a multiple of %d and lda/ldw multiples of 4", K, GEMM_BK);
    return DDP_E_BADCFG;
  }
  const int n_tiles_n = (N + NT * 32 - 1) / (NT * 32);
  static int variant = -1;        // DDP_GEMM_V=1 selects the LDS-staged-A main loop (A/B experiments)
  if (variant < 0) {
    const char* e = getenv("DDP_GEMM_V");
    variant = e ? atoi(e) : 2;
  }
  static bool attr_done = false;  // per instantiation
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_tok<NT, Epi, TAG>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, int(gemm_lds_bytes<NT>()));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_tok2<NT, Epi, TAG>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, int(gemm2_lds_bytes<NT>()));
    attr_done = true;
  }
  prof_begin(TAG, st);
  {
    static int stagger_mul = -1;    // DDP_GEMM_STAGGER = sleeps (of 8128 cycles) per k-tile, x16; default 16 = 1 per k-tile
    if (stagger_mul < 0) {
      const char* e = getenv("DDP_GEMM_STAGGER");
      stagger_mul = e ? atoi(e) : 16;
    }
    static int stagger_mode = -1;
    if (stagger_mode < 0) {
      const char* e = getenv("DDP_GEMM_STAGGER_MODE");
      stagger_mode = e ? atoi(e) : 0;
    }
    const int stagger = stagger_mul < 0 ? stagger_mul : ((K / GEMM_BK) * stagger_mul / 16) | (stagger_mode << 16);
    if (variant == 1 || (K % (2 * GEMM_BK)) != 0)
      hipLaunchKernelGGL((k_gemm_tok<NT, Epi, TAG>), dim3(gemm_grid(M, n_tiles_n)), dim3(GEMM_THREADS),
                         gemm_lds_bytes<NT>(), st, A, lda, W, ldw, M, N, K, n_tiles_n, epi, stagger, g_gemm_dbg);
    else
      hipLaunchKernelGGL((k_gemm_tok2<NT, Epi, TAG>), dim3(gemm_grid(M, n_tiles_n)), dim3(GEMM_THREADS),
                         gemm2_lds_bytes<NT>(), st, A, lda, W, ldw, M, N, K, n_tiles_n, epi, stagger, g_gemm_dbg);
  }
  prof_end(TAG, st);
  return check_launch("k_gemm_tok");
}

}  // namespace

int launch_linear(const float* A, int lda, const float* W, int ldw, const float* bias, const float* add,
                  int ld_add, int rn, int n_tok, float* out, int ldo, int M, int N, int K, int gelu,
                  hipStream_t st, int tag) {
  EpiBias e;
  e.bias = bias;
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
  if (N > 160) {
    switch (tag) {
      case TAG_XPROJ: return launch_gemm<8, TAG_XPROJ>(A, lda, W, ldw, M, N, K, e, st);
      case TAG_FEAT: return launch_gemm<8, TAG_FEAT>(A, lda, W, ldw, M, N, K, e, st);
      case TAG_VALUE: return launch_gemm<8, TAG_VALUE>(A, lda, W, ldw, M, N, K, e, st);
      case TAG_FC1: return launch_gemm<8, TAG_FC1>(A, lda, W, ldw, M, N, K, e, st);
      case TAG_HEAD: return launch_gemm<8, TAG_HEAD>(A, lda, W, ldw, M, N, K, e, st);
      default: return launch_gemm<8, TAG_GENERIC>(A, lda, W, ldw, M, N, K, e, st);
    }
  }
  if (tag == TAG_HEAD) {
    if (N <= 32) return launch_gemm<1, TAG_HEAD>(A, lda, W, ldw, M, N, K, e, st);
    if (N <= 96) return launch_gemm<3, TAG_HEAD>(A, lda, W, ldw, M, N, K, e, st);
    return launch_gemm<5, TAG_HEAD>(A, lda, W, ldw, M, N, K, e, st);
  }
  if (N <= 32) return launch_gemm<1, TAG_GENERIC>(A, lda, W, ldw, M, N, K, e, st);
  if (N <= 96) return launch_gemm<3, TAG_GENERIC>(A, lda, W, ldw, M, N, K, e, st);
  return launch_gemm<5, TAG_GENERIC>(A, lda, W, ldw, M, N, K, e, st);
}

int launch_linear_res_ln(const float* A, int lda, const float* W, int ldw, const float* bias,
                         const float* res, int ldres, const float* gamma, const float* beta,
                         const float* film, float* out, int ldo, int M, int K, hipStream_t st, int tag) {
  EpiResLN e;
  e.bias = bias;
  e.res = res;
  e.ldres = ldres;
  e.gamma = gamma;
  e.beta = beta;
  e.film = film;
  e.out = out;
  e.ldo = ldo;
  if (tag == TAG_FC2_LN) return launch_gemm<8, TAG_FC2_LN>(A, lda, W, ldw, M, 256, K, e, st);
  return launch_gemm<8, TAG_OUTPROJ_LN>(A, lda, W, ldw, M, 256, K, e, st);
}

int launch_linear_samp(const float* A, int lda, const float* Wcat, const float* py, const float* px,
                       int n_tok, int w, float* out, int M, hipStream_t st) {
  EpiSamp e;
  e.py = py;
  e.px = px;
  e.n_tok = n_tok;
  e.w = w;
  e.out = out;
  return launch_gemm<3, TAG_SAMP>(A, lda, Wcat, 256, M, 96, 256, e, st);
}

}  // namespace ddp

extern "C" void ddp_debug_set_stamps(void* buf) { ddp::g_gemm_dbg = static_cast<unsigned long long*>(buf); }

