// ddp_api.hip - C-ABI entry points of libddp_mi355x.so and the K-step loop orchestration.
//
// One stream, no host synchronisation, no allocation: the caller hands over a workspace which is
// carved deterministically from `ddp_cfg` (constants first, then activations).  Layout in HBM:
// every activation is token-major fp32 (rows = tokens of all B*r maps back to back, 256 channels).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "ddp_internal.h"

namespace ddp {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return DDP_E_LAUNCH;
  }
  return DDP_OK;
}

// ---- optional event profiler of one GEMM call site (bench.py roofline leg) ----------------------
namespace {
constexpr int PROF_MAX = 2048;
constexpr int PROF_ALL = 255;        // ddp_profile_begin(255): every tagged call site at once (ddp_profile_read per tag)
struct Prof {
  int tag = -1;
  int n = 0;
  hipEvent_t ev[2 * PROF_MAX];
  unsigned char rec_tag[PROF_MAX];
  float ms[PROF_MAX];
  bool created = false;
  bool resolved = false;             // ddp_profile_end ran for the records in ms[] (cleared by ddp_profile_begin)
} g_prof;
}  // namespace

void prof_begin(int tag, hipStream_t st) {
  if ((tag != g_prof.tag && g_prof.tag != PROF_ALL) || g_prof.n >= PROF_MAX) return;
  (void)hipEventRecord(g_prof.ev[2 * g_prof.n], st);
}
void prof_end(int tag, hipStream_t st) {
  if ((tag != g_prof.tag && g_prof.tag != PROF_ALL) || g_prof.n >= PROF_MAX) return;
  (void)hipEventRecord(g_prof.ev[2 * g_prof.n + 1], st);
  g_prof.rec_tag[g_prof.n] = (unsigned char)tag;
  ++g_prof.n;
}

#define DDP_TRY(expr)            \
  do {                           \
    int _rc = (expr);            \
    if (_rc != DDP_OK) return _rc; \
  } while (0)

namespace {

inline size_t align64(size_t floats) { return (floats + 63) & ~size_t(63); }  // 256-byte granules

struct Carve {
  float* base;
  size_t off = 0;
  float* take(size_t floats) {
    float* p = base ? base + off : nullptr;
    off += align64(floats);
    return p;
  }
};

struct Layout {
  // dims
  int B, r, K, L, Kc, Cx, Cm, h, w, hh, wh, N, Nh, R;
  size_t M0, M;   // tokens at (h,w) and at the head grid, over all B*r maps
  int ldl;        // row stride of logits / prob buffers
  // constants
  float *tin, *u, *hid, *temb, *film, *aff, *lut, *wx, *wm, *wtap;
  float *wcat[DDP_MAX_LAYERS], *bcat[DDP_MAX_LAYERS], *py[DDP_MAX_LAYERS], *px[DDP_MAX_LAYERS];
  // activations
  float *xproj, *mask, *pred, *feat0, *q, *q1, *v, *s, *samp, *hbuf, *logits, *prob, *snoise, *xtok;
  // bf16x3 mode: split weights and split fragment-major activations
  int b3;
  unsigned char* x0_trace;       // DDP_FLAG_RECORD_X0: (K, M) argmax class of every step
  bool fused_layer, fused_pro;   // bf16x3: persistent layer kernel / step-prologue kernel in use (cfg->flags)
  bool guess_zero;               // DDP_FLAG_GATHER_GUESS_ZERO
  SplitW wp_x, wp_m, wp_head, wp_v[DDP_MAX_LAYERS], wp_cat[DDP_MAX_LAYERS], wp_o[DDP_MAX_LAYERS], wp_f0[DDP_MAX_LAYERS],
      wp_f1[DDP_MAX_LAYERS];
  unsigned short *q_sb, *q1_sb, *s_sb, *h_sb, *in_sb;   // in_sb: mask / x / feat staging (row-major producers)
  float* vpad;                                          // zero-padded value maps (R, hh+2, wh+2, 256) written by the layer kernel
  size_t vpad_floats;
  unsigned char* wstream[DDP_MAX_LAYERS];               // layer kernel: weight stream (stage images) per layer
  float* bias_ext[DDP_MAX_LAYERS];                      //               fc1 bias | next value_proj bias | zeros
  unsigned char* pro_stream;                            // step prologue: W_m (8 wide) + layer 0's value / sampling proj (11 tall)
  float* pro_bias;                                      //                layer 0's value_proj bias at [1024, 1280)
  float* ubuf;                                          // u = W_m . m_t, fp32 fragment-major (fused seg tails)
  float* tlut;                                          // seg: (Kc + 1, 256) = LUT . W_m^T; bev (<= 8 classes): (2^Kc, 256) = LUT64 . W_m^T
  float* lut64;                                         // bev (<= 8 classes): the 2^Kc x0 vectors of a pixel
  float* wvs;                                           // depth: W_v0 w_m (256) | W_cat0 w_m (96): the rank-1 terms of the GEMM-free step head
  float *xproj_f, *rs0, *rvpad;                         // depth chain (inside hbuf, unused by the bf16x3 engine otherwise): xproj fragment-major,
                                                        // W_cat0 xproj (M, 96), zero-padded map of W_v0 xproj + b_v0; nullptr: does not fit
  unsigned char* tail4_stream;                          // fused tail: conv_seg images + layer 0's 11 projection images
  unsigned char* head7_stream;                          // first step's head from NCHW (k_layer MODE 7): W_m 8 wide + W_x 8 wide + 11
  unsigned char* lt_stream;                             // last layer + tail (k_layer MODE 6): 72 stages of layer L-1 + tail4_stream
  float* lt_bias;                                       //   fc1 bias of layer L-1 | layer 0's value_proj bias at [1024, 1280)
  float* tail4_bias;                                    //             conv_seg bias | layer 0's value_proj bias at [1024, 1280)
  unsigned char* tail_stream;                           // seg tail: conv_seg stage images (2 per 64 classes)
  float* tail_bias;                                     //           conv_seg bias, zero padded
  size_t const_bytes;   // region A (model constants): a prefix of the workspace that does not depend on the geometry
  size_t total;
};

int validate(const ddp_cfg* c) {
  if (!c) {
    set_error("cfg is NULL");
    return DDP_E_NULL;
  }
  if (c->abi_version != DDP_ABI_VERSION) {
    set_error("abi_version %d != %d", c->abi_version, DDP_ABI_VERSION);
    return DDP_E_BADCFG;
  }
  if (c->task < 0 || c->task > 2) {
    set_error("unknown task %d", c->task);
    return DDP_E_BADCFG;
  }
  if (c->batch < 1 || c->randsteps < 1 || c->timesteps < 1 || c->timesteps > DDP_MAX_STEPS) {
    set_error("batch/randsteps/timesteps out of range (%d,%d,%d)", c->batch, c->randsteps, c->timesteps);
    return DDP_E_BADCFG;
  }
  if (c->num_layers < 1 || c->num_layers > DDP_MAX_LAYERS) {
    set_error("num_layers %d out of range", c->num_layers);
    return DDP_E_BADCFG;
  }
  if (c->h < 1 || c->w < 1 || c->head_h < 1 || c->head_w < 1) {
    set_error("bad spatial size");
    return DDP_E_BADCFG;
  }
  if (c->task != DDP_TASK_BEV && (c->head_h != c->h || c->head_w != c->w)) {
    set_error("head grid must equal (h,w) except for bev");
    return DDP_E_BADCFG;
  }
  if (c->feat_channels < 32 || c->feat_channels % 32) {
    set_error("feat_channels %d must be a positive multiple of 32", c->feat_channels);
    return DDP_E_BADCFG;
  }
  if (c->task != DDP_TASK_DEPTH && (c->num_classes < 1 || c->num_classes > 256)) {
    set_error("num_classes %d out of range [1,256]", c->num_classes);
    return DDP_E_BADCFG;
  }
  if (c->task == DDP_TASK_BEV && c->num_classes > 32) {
    set_error("bev supports at most 32 classes");
    return DDP_E_BADCFG;
  }
  if (c->flags & ~(DDP_FLAG_UNFUSED_LAYER | DDP_FLAG_UNFUSED_PROLOGUE | DDP_FLAG_RECORD_X0 | DDP_FLAG_GATHER_GUESS_ZERO |
                   DDP_FLAG_FORCE_X0 | DDP_FLAG_UNFUSED_TAIL | DDP_FLAG_SB_HEAD | DDP_FLAG_DEPTH_SCALE_UP | DDP_FLAG_DEPTH_NO_EPS)) {
    set_error("unknown flags 0x%x", c->flags);
    return DDP_E_BADCFG;
  }
  if ((c->flags & (DDP_FLAG_FORCE_X0 | DDP_FLAG_RECORD_X0)) && c->task != DDP_TASK_SEG) {
    set_error("DDP_FLAG_RECORD_X0 / DDP_FLAG_FORCE_X0 exist for the segmentation sampler only (task %d)", c->task);
    return DDP_E_BADCFG;
  }
  if ((c->flags & (DDP_FLAG_DEPTH_SCALE_UP | DDP_FLAG_DEPTH_NO_EPS)) && c->task != DDP_TASK_DEPTH) {
    set_error("DDP_FLAG_DEPTH_* configure the depth head only (task %d)", c->task);
    return DDP_E_BADCFG;
  }
  if (c->gemm_mode != DDP_GEMM_F32_MFMA && c->gemm_mode != DDP_GEMM_BF16X3) {
    set_error("unknown gemm_mode %d", c->gemm_mode);
    return DDP_E_BADCFG;
  }
  if (c->sampler != DDP_SAMPLER_DDIM && !(c->sampler == DDP_SAMPLER_DDPM && c->task == DDP_TASK_SEG)) {
    set_error("sampler %d unsupported for task %d (the reference defines ddpm for seg only)", c->sampler, c->task);
    return DDP_E_BADCFG;
  }
  const double tokens = double(c->batch) * c->randsteps * double(c->head_h) * c->head_w;
  if (tokens > 1.5e9 / 1024) {  // keep int row*1024 offsets and 32-bit row counters safe
    set_error("problem too large for one call: %.0f tokens", tokens);
    return DDP_E_BADCFG;
  }
  return DDP_OK;
}

void carve(const ddp_cfg* c, float* base, Layout* o) {
  Carve cv{base};
  o->B = c->batch;
  o->r = c->randsteps;
  o->K = c->timesteps;
  o->L = c->num_layers;
  o->Kc = c->task == DDP_TASK_DEPTH ? 1 : c->num_classes;
  o->Cx = c->feat_channels;
  o->Cm = c->task == DDP_TASK_DEPTH ? 1 : 256;
  o->h = c->h;
  o->w = c->w;
  o->hh = c->head_h;
  o->wh = c->head_w;
  o->N = c->h * c->w;
  o->Nh = c->head_h * c->head_w;
  o->R = c->batch * c->randsteps;
  o->M0 = size_t(o->R) * o->N;
  o->M = size_t(o->R) * o->Nh;
  o->ldl = c->task == DDP_TASK_SEG ? ((o->Kc + 31) / 32) * 32 : 32;
  // ---- region A: MODEL constants.  Sizes depend on (task, K, L, K_cls, Cx, gemm_mode) only - not on batch, r or the map
  // size - so this prefix of the workspace stays valid when only the geometry changes (ddp_prepare_geometry): the
  // reference's own test protocol is one image per call with a new (h, w) almost every call (tools/test.py:214-219)
  o->tin = cv.take(DDP_MAX_STEPS);
  o->u = cv.take(size_t(o->K) * DDP_SINU_FEATS);
  o->hid = cv.take(size_t(o->K) * DDP_TIME_DIM);
  o->temb = cv.take(size_t(o->K) * DDP_TIME_DIM);
  o->film = cv.take(size_t(o->K) * o->L * 512);
  o->aff = cv.take(size_t(o->K) * o->L * 512);   // norms.1 affine x FiLM per (step, layer)
  o->lut = cv.take(size_t(o->Kc + 1) * 256);
  o->wx = cv.take(size_t(256) * o->Cx);
  o->wm = cv.take(size_t(256) * o->Cm);
  o->wtap = cv.take(size_t(9) * 256);
  for (int l = 0; l < DDP_MAX_LAYERS; ++l) {
    const bool on = l < o->L;
    o->wcat[l] = on ? cv.take(96 * 256) : nullptr;
    o->bcat[l] = on ? cv.take(96) : nullptr;
  }
  o->b3 = c->gemm_mode == DDP_GEMM_BF16X3;
  o->guess_zero = (c->flags & DDP_FLAG_GATHER_GUESS_ZERO) != 0;
  o->fused_layer = o->b3 && !(c->flags & DDP_FLAG_UNFUSED_LAYER);
  o->fused_pro = o->fused_layer && !(c->flags & DDP_FLAG_UNFUSED_PROLOGUE);
  const bool segp = c->task == DDP_TASK_SEG;
  if (o->b3) {
    // split weights: 3 bf16 per fp32 = 1.5 floats per element
    auto takew = [&](size_t rows, size_t K) {
      SplitW w;
      w.p = reinterpret_cast<unsigned short*>(cv.take((rows * K * 3 + 1) / 2));
      w.comp_stride = rows * K;
      return w;
    };
    const int head_rows = c->task == DDP_TASK_DEPTH ? 9 : o->Kc;
    o->wp_x = takew(256, o->Cx);
    o->wp_m = takew(256, 256);
    o->wp_head = takew(head_rows, 256);
    for (int l = 0; l < DDP_MAX_LAYERS; ++l) {
      if (l >= o->L) continue;
      o->wp_v[l] = takew(256, 256);
      o->wp_cat[l] = takew(96, 256);
      o->wp_o[l] = takew(256, 256);
      o->wp_f0[l] = takew(DDP_FFN, 256);
      o->wp_f1[l] = takew(256, DDP_FFN);
      o->wstream[l] = reinterpret_cast<unsigned char*>(cv.take(b3_layer_stream_bytes() / sizeof(float)));
      o->bias_ext[l] = cv.take(size_t(b3_layer_bias_floats()));
    }
    o->tail_stream = reinterpret_cast<unsigned char*>(cv.take(size_t(8) * 48 * 1024 / sizeof(float)));
    o->tail_bias = cv.take(size_t(b3_layer_bias_floats()));
    o->pro_stream = reinterpret_cast<unsigned char*>(cv.take(b3_prologue_stream_bytes() / sizeof(float)));
    o->pro_bias = cv.take(size_t(b3_layer_bias_floats()));
    o->tail4_stream = reinterpret_cast<unsigned char*>(cv.take(segp ? size_t(8 + 11) * 48 * 1024 / sizeof(float) : 0));
    o->tail4_bias = cv.take(segp ? size_t(b3_layer_bias_floats()) : 0);
    o->head7_stream = (segp && o->Cx == 256) ? reinterpret_cast<unsigned char*>(cv.take(size_t(8 + 8 + 11) * 48 * 1024 / sizeof(float))) : nullptr;
    // last layer + tail (k_layer MODE 6 / 8 / 9): seg 72 + 2 * chunks + 11 images, bev (<= 32 classes) and depth (9 taps) 72 + 2
    const bool lt = o->L >= 1 && (segp ? b3_layer_tail_supported(o->Kc) : o->Kc <= 32);
    o->lt_stream = lt ? reinterpret_cast<unsigned char*>(cv.take(size_t(segp ? 72 + 8 + 11 : 72 + 2) * 48 * 1024 / sizeof(float))) : nullptr;
    o->lt_bias = lt ? cv.take(size_t(b3_layer_bias_floats())) : nullptr;
    const bool bev_tab = c->task == DDP_TASK_BEV && o->Kc <= 8;
    o->tlut = cv.take(segp ? size_t(o->Kc + 1) * 256 : bev_tab ? (size_t(1) << o->Kc) * 256 : 0);
    o->lut64 = bev_tab ? cv.take((size_t(1) << o->Kc) * 256) : nullptr;
    o->wvs = c->task == DDP_TASK_DEPTH ? cv.take(512) : nullptr;
  } else {
    o->tail_stream = nullptr;
    o->tail_bias = nullptr;
    o->pro_stream = nullptr;
    o->pro_bias = nullptr;
    o->tail4_stream = nullptr;
    o->tail4_bias = nullptr;
    o->head7_stream = nullptr;
    o->lt_stream = nullptr;
    o->lt_bias = nullptr;
    o->tlut = nullptr;
    o->lut64 = nullptr;
    o->wvs = nullptr;
  }
  o->const_bytes = cv.off * sizeof(float);
  // ---- region B: everything that depends on the geometry (batch, r, map size): positional tables, activations
  for (int l = 0; l < DDP_MAX_LAYERS; ++l) {
    const bool on = l < o->L;
    o->py[l] = on ? cv.take(size_t(o->hh) * 96) : nullptr;
    o->px[l] = on ? cv.take(size_t(o->wh) * 96) : nullptr;
  }
  o->xproj = cv.take((size_t(o->B) * o->N + 127) / 128 * 128 * 256);      // (whole tiles: the first step's head may write it fragment-major)
  o->mask = cv.take(o->M0 * (c->task == DDP_TASK_DEPTH ? 1 : 256));
  o->pred = cv.take(c->task == DDP_TASK_DEPTH ? o->M0 : 0);
  o->feat0 = cv.take(c->task == DDP_TASK_BEV ? o->M0 * 256 : 0);
  const size_t Mp = (o->M + 255) / 256 * 256;        // fragment-major buffers hold whole block tiles (128 / 256 tokens)
  o->q = cv.take(Mp * 256);                          // fragment-major
  o->q1 = cv.take(Mp * 256);                         // fragment-major
  o->v = cv.take(o->M * 256);                        // row-major (gather taps want a head's 128 B contiguous)
  o->s = cv.take(o->M * 256);                        // row-major
  o->samp = cv.take(o->M * DDP_SAMP_STRIDE);
  size_t hb = Mp * DDP_FFN;                          // fragment-major
  const size_t xt = size_t(o->B) * o->N * o->Cx;
  if (xt > hb) hb = xt;
  o->hbuf = cv.take(hb);
  o->xtok = o->hbuf;  // x in token-major form is dead once xproj exists
  // (seg: the layer kernel's tails keep scores / probabilities fragment-major: whole 128-token TILES x whole 64-class chunks -
  // the wholly invalid waves of the last tile pre-load their groups' accumulators like every other wave)
  const size_t pfl = c->task == DDP_TASK_SEG ? (o->M + 127) / 128 * 128 * size_t((o->Kc + 63) / 64 * 64) : 0;
  o->logits = cv.take(o->M * o->ldl > pfl ? o->M * o->ldl : pfl);
  o->prob = cv.take(o->M * o->ldl > pfl ? o->M * o->ldl : pfl);
  o->snoise = cv.take(c->sampler == DDP_SAMPLER_DDPM ? o->M0 * 256 : 0);
  o->x0_trace = reinterpret_cast<unsigned char*>(
      cv.take((c->flags & (DDP_FLAG_RECORD_X0 | DDP_FLAG_FORCE_X0)) && c->task == DDP_TASK_SEG
                  ? ((c->flags & DDP_FLAG_FORCE_X0 ? 2 : 1) * size_t(o->K) * o->M + 3) / 4
                  : 0));
  if (o->b3) {
    auto takesb = [&](size_t rows, size_t C) {       // SB: 6 bytes per element, rows padded to 256
      const size_t rp = (rows + 255) / 256 * 256;
      return reinterpret_cast<unsigned short*>(cv.take((rp * C * 3 + 1) / 2));
    };
    // + one more zero row: a sample clamped to y = h reads (with weight 0) the row below the bottom border, which for the
    // last map would otherwise be whatever follows in the workspace (0 x NaN = NaN)
    o->vpad_floats = (size_t(o->R) * (o->hh + 2) + 1) * (o->wh + 2) * 256 + 256;
    o->vpad = cv.take(o->vpad_floats);
    o->ubuf = cv.take(segp ? (o->M + 255) / 256 * 256 * 256 : 0);
    o->q_sb = takesb(o->M, 256);
    o->q1_sb = takesb(o->M, 256);
    o->s_sb = takesb(o->M, 256);
    o->h_sb = takesb(o->M, DDP_FFN);
    size_t in_rows = o->M0 > o->M ? o->M0 : o->M, in_c = 256;
    if (size_t(o->B) * o->N * o->Cx > in_rows * in_c) {
      in_rows = size_t(o->B) * o->N;
      in_c = o->Cx;
    }
    o->in_sb = takesb(in_rows, in_c);
  } else {
    o->q_sb = o->q1_sb = o->s_sb = o->h_sb = o->in_sb = nullptr;
    o->vpad = nullptr;
    o->vpad_floats = 0;
    o->ubuf = nullptr;
  }
  // depth chain: three loop-invariant tensors in the FFN scratch of the fp32 engine (Mp x 1024 floats, idle on the bf16x3 engine)
  o->xproj_f = o->rs0 = o->rvpad = nullptr;
  if (o->b3 && c->task == DDP_TASK_DEPTH && o->r == 1) {
    const size_t need = Mp * 256 + align64(o->M * 96) + o->vpad_floats;
    if (need <= Mp * DDP_FFN && base) {                 // (a one-row map does not fit: its padded map is 3x the map - the GEMM head stays)
      o->xproj_f = o->hbuf;
      o->rs0 = o->hbuf + Mp * 256;
      o->rvpad = o->rs0 + align64(o->M * 96);
    }
  }
  o->total = cv.off * sizeof(float);
}

BevGeom bev_geom(const ddp_cfg* c) {
  BevGeom g;
  g.h = c->h;
  g.w = c->w;
  g.hh = c->head_h;
  g.wh = c->head_w;
  for (int a = 0; a < 2; ++a) {
    g.in_min[a] = c->bev_in_min[a];
    g.in_max[a] = c->bev_in_max[a];
    g.out_first[a] = c->bev_out_first[a];
    g.out_step[a] = c->bev_out_step[a];
  }
  return g;
}

int check_ptr(const void* p, const char* name) {
  if (!p) {
    set_error("%s is NULL", name);
    return DDP_E_NULL;
  }
  if (reinterpret_cast<uintptr_t>(p) & 15) {
    set_error("%s is not 16-byte aligned", name);
    return DDP_E_ALIGN;
  }
  return DDP_OK;
}

int check_weights(const ddp_cfg* c, const ddp_weights* w) {
  if (!w) {
    set_error("weights is NULL");
    return DDP_E_NULL;
  }
  DDP_TRY(check_ptr(w->transform_w, "transform_w"));
  DDP_TRY(check_ptr(w->transform_b, "transform_b"));
  DDP_TRY(check_ptr(w->head_w, "head_w"));
  DDP_TRY(check_ptr(w->head_b, "head_b"));
  if (c->task != DDP_TASK_DEPTH) DDP_TRY(check_ptr(w->embedding, "embedding"));
  for (int l = 0; l < c->num_layers; ++l) {
    const ddp_layer_weights& lw = w->layers[l];
    const void* ps[] = {lw.sampling_offsets_w, lw.sampling_offsets_b, lw.attention_weights_w, lw.attention_weights_b,
                        lw.value_proj_w, lw.value_proj_b, lw.output_proj_w, lw.output_proj_b, lw.ffn0_w, lw.ffn0_b,
                        lw.ffn1_w, lw.ffn1_b, lw.norm0_w, lw.norm0_b, lw.norm1_w, lw.norm1_b};
    for (const void* p : ps) DDP_TRY(check_ptr(p, "layer weight"));
  }
  return DDP_OK;
}

// time embedding + FiLM for S time inputs already on device at tin (segmentors/ddp.py:107-112,
// utils/transformer.py:275-278)
int time_embed_dev(const ddp_weights* w, int L, const float* tin, int S, float* u, float* hid, float* temb,
                   float* film, hipStream_t st) {
  DDP_TRY(launch_sinusoid(w->time_freq, tin, S, u, st));
  DDP_TRY(launch_matvec(w->time1_w, w->time1_b, u, hid, DDP_SINU_FEATS, DDP_TIME_DIM, S, DDP_SINU_FEATS,
                        DDP_TIME_DIM, 0, 1, st));
  DDP_TRY(launch_matvec(w->time3_w, w->time3_b, hid, temb, DDP_TIME_DIM, DDP_TIME_DIM, S, DDP_TIME_DIM,
                        DDP_TIME_DIM, 0, 0, st));
  for (int l = 0; l < L; ++l) {
    if (!w->layers[l].time_w) continue;
    DDP_TRY(launch_matvec(w->layers[l].time_w, w->layers[l].time_b, temb, film + size_t(l) * 512, DDP_TIME_DIM, 512,
                          S, DDP_TIME_DIM, L * 512, 2, 0, st));
  }
  return DDP_OK;
}

// norms.1 affine folded with the per-(step,layer) FiLM vectors
int fold_affine_dev(const ddp_weights* w, int L, int S, const float* film, float* aff, hipStream_t st) {
  for (int l = 0; l < L; ++l) {
    const ddp_layer_weights& lw = w->layers[l];
    for (int s = 0; s < S; ++s) {
      float* dst = aff + (size_t(s) * L + l) * 512;
      if (lw.time_w && film) {
        DDP_TRY(launch_fold_affine(lw.norm1_w, lw.norm1_b, film + (size_t(s) * L + l) * 512, dst, 1, st));
      } else {
        DDP_TRY(launch_pack_rows(lw.norm1_w, 1, lw.norm1_b, 1, dst, 256, st));
      }
    }
  }
  return DDP_OK;
}

// geometry-dependent constants (region B): separable positional tables of every layer (from the packed offset / attention
// projections of region A) and the zero border of the padded value maps
int prepare_geometry(const Layout& o, hipStream_t st) {
  for (int l = 0; l < o.L; ++l) DDP_TRY(launch_pos_tables(o.wcat[l], o.bcat[l], o.py[l], o.px[l], o.hh, o.wh, st));
  // border rows of the padded value maps: zero once, the layer kernel only ever writes the interior
  if (o.b3 && hipMemsetAsync(o.vpad, 0, o.vpad_floats * sizeof(float), st) != hipSuccess) {
    set_error("hipMemsetAsync(vpad) failed");
    return DDP_E_LAUNCH;
  }
  if (o.rvpad && hipMemsetAsync(o.rvpad, 0, o.vpad_floats * sizeof(float), st) != hipSuccess) {
    set_error("hipMemsetAsync(rvpad) failed");
    return DDP_E_LAUNCH;
  }
  return DDP_OK;
}

// model constants that do not depend on the schedule (region A)
int prepare_model(const ddp_cfg* c, const ddp_weights* w, const Layout& o, hipStream_t st) {
  DDP_TRY(launch_pack_cols(w->transform_w, o.Cx + o.Cm, 0, 256, o.Cx, o.wx, st));
  DDP_TRY(launch_pack_cols(w->transform_w, o.Cx + o.Cm, o.Cx, 256, o.Cm, o.wm, st));
  if (c->task == DDP_TASK_SEG) DDP_TRY(launch_build_lut(w->embedding, o.lut, o.Kc + 1, c->bit_scale, st));
  if (c->task == DDP_TASK_DEPTH) DDP_TRY(launch_pack_conv3x3(w->head_w, o.wtap, st));
  for (int l = 0; l < o.L; ++l) {
    const ddp_layer_weights& lw = w->layers[l];
    DDP_TRY(launch_pack_rows(lw.sampling_offsets_w, 64, lw.attention_weights_w, 32, o.wcat[l], 256, st));
    DDP_TRY(launch_pack_rows(lw.sampling_offsets_b, 64, lw.attention_weights_b, 32, o.bcat[l], 1, st));
  }
  if (o.b3) {
    auto wr = [](const SplitW& w) { return const_cast<unsigned short*>(w.p); };
    DDP_TRY(launch_split_weights(o.wx, o.Cx, 256, o.Cx, wr(o.wp_x), st));
    if (c->task != DDP_TASK_DEPTH) DDP_TRY(launch_split_weights(o.wm, 256, 256, 256, wr(o.wp_m), st));
    if (c->task == DDP_TASK_DEPTH) DDP_TRY(launch_split_weights(o.wtap, 256, 9, 256, wr(o.wp_head), st));
    else DDP_TRY(launch_split_weights(w->head_w, 256, o.Kc, 256, wr(o.wp_head), st));
    for (int l = 0; l < o.L; ++l) {
      const ddp_layer_weights& lw = w->layers[l];
      DDP_TRY(launch_split_weights(lw.value_proj_w, 256, 256, 256, wr(o.wp_v[l]), st));
      DDP_TRY(launch_split_weights(o.wcat[l], 256, 96, 256, wr(o.wp_cat[l]), st));
      DDP_TRY(launch_split_weights(lw.output_proj_w, 256, 256, 256, wr(o.wp_o[l]), st));
      DDP_TRY(launch_split_weights(lw.ffn0_w, 256, DDP_FFN, 256, wr(o.wp_f0[l]), st));
      DDP_TRY(launch_split_weights(lw.ffn1_w, DDP_FFN, 256, DDP_FFN, wr(o.wp_f1[l]), st));
    }
    if (c->task == DDP_TASK_SEG) {      // seg tail: conv_seg as tall stages of 64 classes, bias table
      const int nch = (o.Kc + 63) / 64;
      DDP_TRY(launch_build_stages(o.wp_head.p, o.wp_head.comp_stride, 256, o.Kc, 1, nch, 2, 0, 0, 1, 2, o.tail_stream, st));
      if (hipMemsetAsync(o.tail_bias, 0, size_t(b3_layer_bias_floats()) * sizeof(float), st) != hipSuccess ||
          (w->head_b && hipMemcpyAsync(o.tail_bias, w->head_b, size_t(o.Kc) * sizeof(float), hipMemcpyDeviceToDevice, st) !=
                            hipSuccess)) {
        set_error("tail bias copy failed");
        return DDP_E_LAUNCH;
      }
    }
    if (c->task == DDP_TASK_SEG) {      // fused tail (k_layer MODE 4): [conv_seg: 2 tall per 64 classes][layer 0's Wv: 8 tall][Wcat: 2 tall + 1 split-K]
      const int nch = (o.Kc + 63) / 64;
      DDP_TRY(launch_build_stages(o.wp_head.p, o.wp_head.comp_stride, 256, o.Kc, 1, nch, 2, 0, 0, 1, 2, o.tail4_stream, st));
      DDP_TRY(launch_build_stages(o.wp_v[0].p, o.wp_v[0].comp_stride, 256, 256, 1, 4, 2, 2 * nch, 0, 1, 2, o.tail4_stream, st));
      DDP_TRY(launch_build_stages(o.wp_cat[0].p, o.wp_cat[0].comp_stride, 256, 96, 1, 1, 2, 2 * nch + 8, 0, 1, 2, o.tail4_stream, st));
      DDP_TRY(launch_build_stages(o.wp_cat[0].p, o.wp_cat[0].comp_stride, 256, 96, 2, 1, 1, 2 * nch + 10, 64, 0, 0, o.tail4_stream, st));
      if (hipMemsetAsync(o.tail4_bias, 0, size_t(b3_layer_bias_floats()) * sizeof(float), st) != hipSuccess ||
          (w->head_b && hipMemcpyAsync(o.tail4_bias, w->head_b, size_t(o.Kc) * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) ||
          hipMemcpyAsync(o.tail4_bias + DDP_FFN, w->layers[0].value_proj_b, 256 * sizeof(float), hipMemcpyDeviceToDevice, st) !=
              hipSuccess) {
        set_error("fused tail bias copy failed");
        return DDP_E_LAUNCH;
      }
      // T = LUT . W_m^T: the noisy-map half of the concat-conv applied to each of the K + 1 possible x0 vectors (exact
      // fp32 products: the f32-input MFMA GEMM)
      DDP_TRY(launch_linear(o.lut, 256, false, o.wm, 256, nullptr, nullptr, 0, 0, 0, o.tlut, 256, o.Kc + 1, 256, 256, 0, st));
    }
    {                                   // step prologue: [W_m: 8 wide stages][layer 0's Wv: 8 tall][layer 0's Wcat: 2 tall + 1 split-K]
      // (depth has no W_m GEMM - one input channel - and uses images 8..18 only: k_layer MODE 3)
      if (c->task != DDP_TASK_DEPTH)
        DDP_TRY(launch_build_stages(o.wp_m.p, o.wp_m.comp_stride, 256, 256, 0, 1, 8, 0, 2, 1, 0, o.pro_stream, st));
      DDP_TRY(launch_build_stages(o.wp_v[0].p, o.wp_v[0].comp_stride, 256, 256, 1, 4, 2, 8, 0, 1, 2, o.pro_stream, st));
      DDP_TRY(launch_build_stages(o.wp_cat[0].p, o.wp_cat[0].comp_stride, 256, 96, 1, 1, 2, 16, 0, 1, 2, o.pro_stream, st));
      DDP_TRY(launch_build_stages(o.wp_cat[0].p, o.wp_cat[0].comp_stride, 256, 96, 2, 1, 1, 18, 64, 0, 0, o.pro_stream, st));
      if (hipMemsetAsync(o.pro_bias, 0, size_t(b3_layer_bias_floats()) * sizeof(float), st) != hipSuccess ||
          hipMemcpyAsync(o.pro_bias + DDP_FFN, w->layers[0].value_proj_b, 256 * sizeof(float), hipMemcpyDeviceToDevice, st) !=
              hipSuccess) {
        set_error("prologue bias copy failed");
        return DDP_E_LAUNCH;
      }
    }
    if (o.head7_stream) {               // first step's head from NCHW (k_layer MODE 7): [W_m: 8 wide][W_x: 8 wide][layer 0's Wv, Wcat as above]
      DDP_TRY(launch_build_stages(o.wp_m.p, o.wp_m.comp_stride, 256, 256, 0, 1, 8, 0, 2, 1, 0, o.head7_stream, st));
      DDP_TRY(launch_build_stages(o.wp_x.p, o.wp_x.comp_stride, 256, 256, 0, 1, 8, 8, 2, 1, 0, o.head7_stream, st));
      if (hipMemcpyAsync(o.head7_stream + size_t(16) * 48 * 1024, o.pro_stream + size_t(8) * 48 * 1024, size_t(11) * 48 * 1024,
                         hipMemcpyDeviceToDevice, st) != hipSuccess) {
        set_error("first-step head stream copy failed");
        return DDP_E_LAUNCH;
      }
    }
    // layer-kernel weight streams: [Wo: 8 wide stages][16 x (fc1 tall, tall, fc2 wide, wide)][next Wv: 8 tall][next Wcat: 2 tall + 1 split-K]
    for (int l = 0; l < o.L; ++l) {
      const ddp_layer_weights& lw = w->layers[l];
      unsigned char* sp = o.wstream[l];
      DDP_TRY(launch_build_stages(o.wp_o[l].p, o.wp_o[l].comp_stride, 256, 256, 0, 1, 8, 0, 2, 1, 0, sp, st));
      DDP_TRY(launch_build_stages(o.wp_f0[l].p, o.wp_f0[l].comp_stride, 256, DDP_FFN, 1, 16, 2, 8, 0, 1, 4, sp, st));
      DDP_TRY(launch_build_stages(o.wp_f1[l].p, o.wp_f1[l].comp_stride, DDP_FFN, 256, 0, 1, 32, 10, 4, 1, 0, sp, st));
      if (hipMemsetAsync(o.bias_ext[l], 0, size_t(b3_layer_bias_floats()) * sizeof(float), st) != hipSuccess ||
          hipMemcpyAsync(o.bias_ext[l], lw.ffn0_b, DDP_FFN * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) {
        set_error("bias table copy failed");
        return DDP_E_LAUNCH;
      }
      if (l + 1 < o.L) {
        const ddp_layer_weights& nw = w->layers[l + 1];
        DDP_TRY(launch_build_stages(o.wp_v[l + 1].p, o.wp_v[l + 1].comp_stride, 256, 256, 1, 4, 2, 72, 0, 1, 2, sp, st));
        DDP_TRY(launch_build_stages(o.wp_cat[l + 1].p, o.wp_cat[l + 1].comp_stride, 256, 96, 1, 1, 2, 80, 0, 1, 2, sp, st));
        DDP_TRY(launch_build_stages(o.wp_cat[l + 1].p, o.wp_cat[l + 1].comp_stride, 256, 96, 2, 1, 1, 82, 64, 0, 0, sp, st));
        if (hipMemcpyAsync(o.bias_ext[l] + DDP_FFN, nw.value_proj_b, 256 * sizeof(float), hipMemcpyDeviceToDevice, st) !=
            hipSuccess) {
          set_error("bias table copy failed");
          return DDP_E_LAUNCH;
        }
      }
    }
    if (o.lt_stream && c->task != DDP_TASK_SEG) {
      // last layer + bev / depth tail (k_layer MODE 8 / 9): [layer L-1: 72 stages][the head convolution: 2 tall stages of <= 32 rows]
      // (conv_seg of the bev head; the nine taps of the 3x3 conv_depth as nine output rows).  Bias table: fc1's; tail_bias: conv_seg's
      // (depth: zeros - conv_depth's bias is added once per pixel by k_depth_update).
      const size_t sb = size_t(48) * 1024;
      const int rows = c->task == DDP_TASK_DEPTH ? 9 : o.Kc;
      DDP_TRY(launch_build_stages(o.wp_head.p, o.wp_head.comp_stride, 256, rows, 1, 1, 2, 0, 0, 1, 2, o.tail_stream, st));
      if (hipMemsetAsync(o.tail_bias, 0, size_t(b3_layer_bias_floats()) * sizeof(float), st) != hipSuccess ||
          (c->task == DDP_TASK_BEV && w->head_b &&
           hipMemcpyAsync(o.tail_bias, w->head_b, size_t(o.Kc) * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) ||
          hipMemcpyAsync(o.lt_stream, o.wstream[o.L - 1], 72 * sb, hipMemcpyDeviceToDevice, st) != hipSuccess ||
          hipMemcpyAsync(o.lt_stream + 72 * sb, o.tail_stream, 2 * sb, hipMemcpyDeviceToDevice, st) != hipSuccess ||
          hipMemsetAsync(o.lt_bias, 0, size_t(b3_layer_bias_floats()) * sizeof(float), st) != hipSuccess ||
          hipMemcpyAsync(o.lt_bias, w->layers[o.L - 1].ffn0_b, DDP_FFN * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) {
        set_error("layer + tail stream copy failed");
        return DDP_E_LAUNCH;
      }
    }
    if (o.wvs) {
      // depth chain: the rank-1 terms of the GEMM-free step head, W_v0 w_m and W_cat0 w_m (fp32 dot products)
      DDP_TRY(launch_matvec(w->layers[0].value_proj_w, nullptr, o.wm, o.wvs, 256, 256, 1, 256, 256, 0, 0, st));
      DDP_TRY(launch_matvec(o.wcat[0], nullptr, o.wm, o.wvs + 256, 256, 96, 1, 256, 96, 0, 0, st));
    }
    if (o.lut64) {
      // bev u chain: the 2^K x0 vectors of a pixel and their images under W_m (exact fp32 products, as the seg table)
      DDP_TRY(launch_build_bev_lut(w->embedding, o.lut64, o.Kc, c->bit_scale, st));
      DDP_TRY(launch_linear(o.lut64, 256, false, o.wm, 256, nullptr, nullptr, 0, 0, 0, o.tlut, 256, 1 << o.Kc, 256, 256, 0, st));
    }
    if (o.lt_stream && c->task == DDP_TASK_SEG) {
      // last layer + tail (k_layer MODE 6): the images do not depend on the step, so ONE concatenated stream serves every step -
      // [layer L-1: Wo 8 wide, 16 x (fc1 2 tall, fc2 2 wide)][conv_seg 2 tall per 64 classes][layer 0's Wv 8 tall][Wcat 2 tall +
      // 1 split-K]; after the last step the kernel wraps behind conv_seg.  Bias table: fc1's | layer 0's value_proj bias.
      const int nch = (o.Kc + 63) / 64;
      const size_t sb = size_t(48) * 1024;
      if (hipMemcpyAsync(o.lt_stream, o.wstream[o.L - 1], 72 * sb, hipMemcpyDeviceToDevice, st) != hipSuccess ||
          hipMemcpyAsync(o.lt_stream + 72 * sb, o.tail4_stream, size_t(2 * nch + 11) * sb, hipMemcpyDeviceToDevice, st) != hipSuccess ||
          hipMemsetAsync(o.lt_bias, 0, size_t(b3_layer_bias_floats()) * sizeof(float), st) != hipSuccess ||
          hipMemcpyAsync(o.lt_bias, w->layers[o.L - 1].ffn0_b, DDP_FFN * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess ||
          hipMemcpyAsync(o.lt_bias + DDP_FFN, w->layers[0].value_proj_b, 256 * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) {
        set_error("layer + tail stream copy failed");
        return DDP_E_LAUNCH;
      }
    }
  }
  return DDP_OK;
}

// publish a row-major (M,256) activation as the encoder input: fp32 fragment-major q, or SB in bf16x3 mode
int publish_q(const Layout& o, const float* row_major, hipStream_t st) {
  if (o.b3 && !o.fused_layer) return launch_row_to_sb(row_major, 256, o.q_sb, int(o.M), 256, st);
  return launch_row_to_blk(row_major, o.q, int(o.M), st);     // the layer kernels (and the fp32 engine) take q as fp32 fragments
}

// DetrTransformerEncoder over the fragment-major q (in/out); aff (L,512) = norms.1 affine x FiLM
// `tail` (seg, u chain): the step's tail is fused into the LAST layer's kernel (k_layer MODE 6) - the caller launches no tail
int encoder_forward(const ddp_weights* w, const Layout& o, const float* aff, hipStream_t st, bool l0_projected = false,
                    bool sb_out = true, const TailLaunch* tail = nullptr, bool depth_chain = false) {
  const int M = int(o.M);
  if (o.b3) {
    // same dataflow on the bf16 matrix cores: q / q1 travel only as SB (operands AND residuals: the three pieces
    // of an element sum to its fp32 value exactly)
    const bool fused = o.fused_layer;
    for (int l = 0; l < o.L; ++l) {
      const ddp_layer_weights& lw = w->layers[l];
      // later layers: emitted by the previous layer kernel; layer 0: by the step prologue kernel when that ran
      bool own_proj = !fused || (l == 0 && !l0_projected);
      if (own_proj && fused) {
        // layer 0 of a path without a fused step head (bev after its grid resampling, ddp_head_forward): the value /
        // sampling projections of the SB q as ONE launch of the layer kernel's P3 (k_layer MODE 3), padded value map out
        L0ProjLaunch pj;
        pj.Q = o.q;
        pj.stream = o.pro_stream + size_t(8) * 48 * 1024;
        pj.bias_ext = o.pro_bias;
        pj.res = nullptr;
        pj.res_rn = 0;
        pj.wm = nullptr;
        pj.dvec = nullptr;
        pj.upd = nullptr;
        pj.M = M;
        pj.v_out = o.vpad;
        pj.samp_out = o.samp;
        pj.py = o.py[0];
        pj.px = o.px[0];
        pj.n_tok = o.Nh;
        pj.w = o.wh;
        DDP_TRY(launch_b3_l0proj(pj, st));
        own_proj = false;
      }
      if (own_proj) {
        DDP_TRY(launch_b3_linear(o.q_sb, o.wp_v[l], lw.value_proj_b, nullptr, 0, 0, 0, o.v, 256, M, 256, 256, st, TAG_VALUE));
        DDP_TRY(launch_b3_linear_samp(o.q_sb, o.wp_cat[l], o.py[l], o.px[l], o.Nh, o.wh, o.samp, M, st));
      }
      // a row-major GEMM leaves a plain value map; the layer / prologue kernels write it zero-padded
      // (the LDS gather feeds the layer kernel only: its output is fp32 in the accumulator layout - o.q1 is free on this path)
      if (own_proj) DDP_TRY(launch_msda_gather_sb(o.v, o.samp, o.s_sb, M, o.Nh, o.hh, o.wh, st));
      else DDP_TRY(launch_msda_gather_sb_pad(o.vpad, o.samp, o.s_sb, DDP_S_F32 ? o.q1 : nullptr, M, o.Nh, o.hh, o.wh, o.py[l], o.px[l],
                                             o.guess_zero, st));
      const float* a = aff + size_t(l) * 512;
      if (fused) {
        // output_proj + LN0 + FFN + LN1 + FiLM + the next layer's value / sampling projections: one persistent kernel
        LayerLaunch ll;
        ll.S = o.s_sb;
        ll.Sf = DDP_S_F32 ? o.q1 : nullptr;
        ll.Q = o.q;
        ll.Q_sb = (sb_out && l + 1 == o.L) ? o.q_sb : nullptr;      // a head GEMM after the encoder reads SB
        ll.stream = o.wstream[l];
        ll.bias_ext = o.bias_ext[l];
        ll.bo = lw.output_proj_b;
        ll.ga0 = lw.norm0_w;
        ll.be0 = lw.norm0_b;
        ll.b2 = lw.ffn1_b;
        ll.ga1 = a;
        ll.be1 = a + 256;
        ll.M = M;
        ll.has_next = l + 1 < o.L;
        ll.v_out = o.vpad;
        ll.samp_out = o.samp;
        ll.py = ll.has_next ? o.py[l + 1] : nullptr;
        ll.px = ll.has_next ? o.px[l + 1] : nullptr;
        ll.n_tok = o.Nh;
        ll.w = o.wh;
        // depth chain: q of the step never exists - layer 0 forms its residual from xproj and the noisy depth (k_layer MODE 10)
        ll.res_f = (depth_chain && l == 0) ? o.xproj_f : nullptr;
        ll.wm = o.wm;
        ll.dvec = o.mask;
        if (tail && l + 1 == o.L) DDP_TRY(launch_b3_layer_tail(ll, *tail, o.lt_stream, o.lt_bias, o.tail_bias, st));
        else DDP_TRY(launch_b3_layer(ll, st));
      } else {
        DDP_TRY(launch_b3_linear_res_ln(o.s_sb, o.wp_o[l], lw.output_proj_b, nullptr, o.q_sb, lw.norm0_w, lw.norm0_b, nullptr,
                                        o.q1_sb, M, 256, st, TAG_OUTPROJ_LN));
        DDP_TRY(launch_b3_linear_sb(o.q1_sb, o.wp_f0[l], lw.ffn0_b, nullptr, 0, 0, 0, o.h_sb, nullptr, M, DDP_FFN, 256, 1, st,
                                    TAG_FC1));
        DDP_TRY(launch_b3_linear_res_ln(o.h_sb, o.wp_f1[l], lw.ffn1_b, nullptr, o.q1_sb, a, a + 256, nullptr, o.q_sb, M,
                                        DDP_FFN, st, TAG_FC2_LN));
      }
    }
    return DDP_OK;
  }
  for (int l = 0; l < o.L; ++l) {
    const ddp_layer_weights& lw = w->layers[l];
    // value / sampling projections (multi_scale_deform_attn.py:313-328)
    DDP_TRY(launch_linear(o.q, 256, true, lw.value_proj_w, 256, lw.value_proj_b, nullptr, 0, 0, 0, o.v, 256, M, 256, 256, 0,
                          st, TAG_VALUE));
    DDP_TRY(launch_linear_samp(o.q, o.wcat[l], o.py[l], o.px[l], o.Nh, o.wh, o.samp, M, st));
    // bilinear gather + weighted sum (:94-151)
    DDP_TRY(launch_msda_gather(o.v, o.samp, o.s, M, o.Nh, o.hh, o.wh, st));
    // output_proj + identity, LayerNorm (:352-358; utils/transformer.py:390-392)
    DDP_TRY(launch_linear_res_ln_blk(o.s, 256, false, lw.output_proj_w, 256, lw.output_proj_b, o.q, lw.norm0_w, lw.norm0_b,
                                     o.q1, M, 256, st));
    // FFN + identity, LayerNorm, FiLM (mmcv FFN :269-280; utils/transformer.py:413-417)
    DDP_TRY(launch_linear_blk(o.q1, 256, true, lw.ffn0_w, 256, lw.ffn0_b, nullptr, 0, 0, 0, o.hbuf, M, DDP_FFN, 256, 1, st));
    const float* a = aff + size_t(l) * 512;
    DDP_TRY(launch_linear_res_ln_blk(o.hbuf, DDP_FFN, true, lw.ffn1_w, DDP_FFN, lw.ffn1_b, o.q1, a, a + 256, o.q, M, DDP_FFN,
                                     st));
  }
  return DDP_OK;
}

}  // namespace
}  // namespace ddp

using namespace ddp;

extern "C" {

const char* ddp_last_error(void) { return g_err; }
int ddp_abi_version(void) { return DDP_ABI_VERSION; }

int ddp_profile_begin(int tag) {
  if ((tag < 0 || tag >= TAG_COUNT) && tag != PROF_ALL) {
    set_error("profile: unknown tag %d", tag);
    return DDP_E_BADCFG;
  }
  if (!g_prof.created) {
    for (int i = 0; i < 2 * PROF_MAX; ++i)
      if (hipEventCreate(&g_prof.ev[i]) != hipSuccess) {
        set_error("hipEventCreate failed");
        return DDP_E_LAUNCH;
      }
    g_prof.created = true;
  }
  g_prof.n = 0;
  g_prof.tag = tag;
  g_prof.resolved = false;
  return DDP_OK;
}

int ddp_profile_end(float* total_ms, int* launches) {
  const int n = g_prof.n;
  g_prof.tag = -1;
  float tot = 0.f;
  for (int i = 0; i < n; ++i) {
    if (hipEventSynchronize(g_prof.ev[2 * i + 1]) != hipSuccess) {
      set_error("hipEventSynchronize failed");
      return DDP_E_LAUNCH;
    }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]);
    g_prof.ms[i] = ms;
    tot += ms;
  }
  g_prof.resolved = true;
  if (total_ms) *total_ms = tot;
  if (launches) *launches = n;
  return DDP_OK;
}

int ddp_profile_read(int tag, float* total_ms, int* launches) {
  // after ddp_profile_end: the share of one call site in the records of the last session
  if (tag < 0 || tag >= TAG_COUNT) {
    set_error("profile: unknown tag %d", tag);
    return DDP_E_BADCFG;
  }
  if (!g_prof.resolved) {
    set_error("profile: no finished session (call ddp_profile_end first)");
    return DDP_E_BADCFG;
  }
  float tot = 0.f;
  int cnt = 0;
  for (int i = 0; i < g_prof.n; ++i)
    if (g_prof.rec_tag[i] == tag) {
      tot += g_prof.ms[i];
      ++cnt;
    }
  if (total_ms) *total_ms = tot;
  if (launches) *launches = cnt;
  return DDP_OK;
}

int ddp_query_workspace(const ddp_cfg* cfg, size_t* bytes) {
  DDP_TRY(validate(cfg));
  if (!bytes) {
    set_error("bytes is NULL");
    return DDP_E_NULL;
  }
  Layout o;
  carve(cfg, nullptr, &o);
  *bytes = o.total;
  return DDP_OK;
}

int ddp_prepare(const ddp_cfg* cfg, const ddp_weights* weights, const ddp_step* steps, void* d_workspace,
                void* stream) {
  DDP_TRY(validate(cfg));
  DDP_TRY(check_weights(cfg, weights));
  DDP_TRY(check_ptr(d_workspace, "workspace"));
  if (!steps) {
    set_error("steps is NULL");
    return DDP_E_NULL;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  Layout o;
  carve(cfg, static_cast<float*>(d_workspace), &o);
  DDP_TRY(prepare_model(cfg, weights, o, st));
  DDP_TRY(prepare_geometry(o, st));
  float tin[DDP_MAX_STEPS];
  for (int s = 0; s < o.K; ++s) tin[s] = steps[s].time_in;
  DDP_TRY(launch_write_floats(tin, o.K, o.tin, st));
  DDP_TRY(time_embed_dev(weights, o.L, o.tin, o.K, o.u, o.hid, o.temb, o.film, st));
  DDP_TRY(fold_affine_dev(weights, o.L, o.K, o.film, o.aff, st));
  return DDP_OK;
}

int ddp_query_const_workspace(const ddp_cfg* cfg, size_t* bytes) {
  DDP_TRY(validate(cfg));
  if (!bytes) {
    set_error("bytes is NULL");
    return DDP_E_NULL;
  }
  Layout o;
  carve(cfg, nullptr, &o);
  *bytes = o.const_bytes;
  return DDP_OK;
}

int ddp_prepare_geometry(const ddp_cfg* cfg, void* d_workspace, void* stream) {
  DDP_TRY(validate(cfg));
  DDP_TRY(check_ptr(d_workspace, "workspace"));
  Layout o;
  carve(cfg, static_cast<float*>(d_workspace), &o);
  return prepare_geometry(o, static_cast<hipStream_t>(stream));
}

int ddp_sample(const ddp_cfg* cfg, const ddp_weights* weights, const ddp_step* steps, const float* d_x,
               const float* d_noise, const float* d_step_noise, float* d_out, void* d_workspace, void* stream) {
  DDP_TRY(validate(cfg));
  DDP_TRY(check_weights(cfg, weights));
  DDP_TRY(check_ptr(d_workspace, "workspace"));
  DDP_TRY(check_ptr(d_x, "x"));
  DDP_TRY(check_ptr(d_noise, "noise"));
  DDP_TRY(check_ptr(d_out, "out"));
  if (!steps) {
    set_error("steps is NULL");
    return DDP_E_NULL;
  }
  if (cfg->sampler == DDP_SAMPLER_DDPM) DDP_TRY(check_ptr(d_step_noise, "step_noise"));
  hipStream_t st = static_cast<hipStream_t>(stream);
  Layout o;
  carve(cfg, static_cast<float*>(d_workspace), &o);
  const int M = int(o.M), M0 = int(o.M0);
  const BevGeom geom = bev_geom(cfg);

  // seg + DDIM u chain with one noisy map per image: the first step's head reads the caller's NCHW x and start noise directly
  // (k_layer MODE 7) and writes xproj itself - no NCHW -> SB conversions, no x-projection GEMM
  const bool head7 = o.b3 && o.fused_layer && o.fused_pro && cfg->task == DDP_TASK_SEG && cfg->sampler == DDP_SAMPLER_DDIM &&
                     o.h == o.hh && o.w == o.wh && o.r == 1 && o.head7_stream && !(cfg->flags & DDP_FLAG_SB_HEAD) &&
                     !((size_t(o.M) * 256) >> 32);
  // loop-invariant half of the concat-conv: xproj = W_x x + b  (ddp.py:223-224 with the x columns hoisted)
  if (head7) {
    // (inside the first step's head)
  } else if (o.b3) {
    DDP_TRY(launch_nchw_to_sb(d_x, o.in_sb, o.B, o.Cx, o.N, st));       // NCHW -> split fragments in one pass
    DDP_TRY(launch_b3_linear(o.in_sb, o.wp_x, weights->transform_b, nullptr, 0, 0, 0, o.xproj, 256, o.B * o.N, 256, o.Cx, st,
                             TAG_XPROJ));
  } else {
    DDP_TRY(launch_nchw_to_tok(d_x, o.xtok, o.B, o.Cx, o.N, st));
    DDP_TRY(launch_linear(o.xtok, o.Cx, false, o.wx, o.Cx, weights->transform_b, nullptr, 0, 0, 0, o.xproj, 256, o.B * o.N,
                          256, o.Cx, 0, st, TAG_XPROJ));
  }
  // seg + DDIM on the bf16x3 engine: conv_seg, argmax, softmax accumulation, x0 LUT and the DDIM update run as the
  // "tail" mode of the layer kernel, which leaves m_{t_next} as the SB operand of the next step's concat-conv
  const bool seg_tail = o.fused_layer && cfg->task == DDP_TASK_SEG && cfg->sampler == DDP_SAMPLER_DDIM &&
                        o.h == o.hh && o.w == o.wh;
  if (cfg->task == DDP_TASK_DEPTH) {
    if (hipMemcpyAsync(o.mask, d_noise, size_t(M0) * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) {
      set_error("noise copy failed");
      return DDP_E_LAUNCH;
    }
  } else if (head7) {
    // (the first step's head reads d_noise itself)
  } else if (seg_tail) {
    DDP_TRY(launch_nchw_to_sb(d_noise, o.in_sb, o.R, 256, o.N, st));    // the noisy map only ever exists as SB on this path
  } else if (cfg->task == DDP_TASK_BEV && o.b3 && o.fused_layer && o.fused_pro && o.lt_stream && o.lut64 &&
             !(cfg->flags & DDP_FLAG_UNFUSED_TAIL)) {
    // (bev u chain - the condition `bev_chain` below: the start noise goes straight into u_0 = W_m . noise)
  } else {
    DDP_TRY(launch_nchw_to_tok(d_noise, o.mask, o.R, 256, o.N, st));
  }

  // tasks whose concat-conv feeds the encoder directly (seg): the head of the step is one kernel with layer 0's projections
  const bool pro_fused = o.fused_pro && cfg->task == DDP_TASK_SEG && o.h == o.hh &&
                         o.w == o.wh;
  // seg + DDIM with both fusions on: the noisy map enters the loop only through u = W_m . m_t and its update is affine in
  // (m_t, x0) with x0 one of K + 1 table rows, so steps 1 .. K-1 run NO concat-conv GEMM and keep no 256-channel map:
  // the tail of step s updates u (u' = ua u + uc (W_m . LUT)[argmax]) and is at the same time the head of step s + 1
  // (q = W_x x + b + u', layer 0's projections): k_layer MODE 4.  Step 0 starts from the noise with the step-prologue
  // kernel, the last step's tail has nothing to update.
  const bool u_chain = seg_tail && pro_fused;
  // bev / depth on the fused path: the step's LAST layer runs with the head convolution as its tail (k_layer MODE 8 / 9) - no head GEMM,
  // no SB copy of the layer output.  bev with <= 8 classes additionally runs the u chain: the grid transform and the concat-conv are
  // linear, so resample(W_x x + b) is hoisted out of the loop (rx, row-major in o.s: the fused layers never touch that buffer), the
  // noisy map is carried as u = W_m m (o.feat0, row-major at the map size) and follows the DDIM update through the 2^K-row table
  // T = LUT64 . W_m^T - per step: u update, q = rx + resample(u), layer 0's projections; no GEMM at the map size after u_0.
  const bool lt_other = o.b3 && o.fused_layer && o.fused_pro && o.lt_stream && !(cfg->flags & DDP_FLAG_UNFUSED_TAIL);
  const bool depth_lt = cfg->task == DDP_TASK_DEPTH && lt_other;
  // depth chain (one noisy map per image, >= 2 layers): the step head without a GEMM.  Loop invariant, once per sample: xproj
  // fragment-major (layer 0's residual operand), rvpad = W_v0 xproj + b_v0 as a padded map (layer 0's projection kernel run on xproj
  // itself), rs0 = W_cat0 xproj; per step k_depth_head adds the rank-1 terms in the noisy depth (and runs the previous step's update)
  const bool depth_chain = depth_lt && o.rvpad && o.L >= 2;
  if (depth_chain) {
    DDP_TRY(launch_row_to_blk(o.xproj, o.xproj_f, M0, st));
    L0ProjLaunch pj;
    pj.Q = o.xproj_f;
    pj.stream = o.pro_stream + size_t(8) * 48 * 1024;
    pj.bias_ext = o.pro_bias;
    pj.res = nullptr;
    pj.res_rn = 0;
    pj.wm = nullptr;
    pj.dvec = nullptr;
    pj.upd = nullptr;
    pj.M = M0;
    pj.v_out = o.rvpad;
    pj.samp_out = o.samp;                                  // (the table of a zero depth: overwritten by the first step's head)
    pj.py = o.py[0];
    pj.px = o.px[0];
    pj.n_tok = o.Nh;
    pj.w = o.wh;
    DDP_TRY(launch_b3_l0proj(pj, st));
    DDP_TRY(launch_row_to_sb(o.xproj, 256, o.in_sb, M0, 256, st));
    DDP_TRY(launch_b3_linear(o.in_sb, o.wp_cat[0], nullptr, nullptr, 0, 0, 0, o.rs0, 96, M0, 96, 256, st, TAG_SAMP));
  }
  const bool bev_chain = cfg->task == DDP_TASK_BEV && lt_other && o.lut64;
  if (bev_chain) {
    DDP_TRY(launch_bev_resample(o.xproj, o.s, o.B, geom, st));                      // rx = resample(W_x x + b), B maps
    DDP_TRY(launch_nchw_to_sb(d_noise, o.in_sb, o.R, 256, o.N, st));
    DDP_TRY(launch_b3_linear(o.in_sb, o.wp_m, nullptr, nullptr, 0, 0, 0, o.feat0, 256, M0, 256, 256, st, TAG_FEAT));   // u_0 = W_m . noise
  }
  unsigned char* bev_code = reinterpret_cast<unsigned char*>(o.logits);            // (M) the step's x0 code per head-grid token
  auto depth_update_args = [&](const ddp_step& stp) {
    DepthUpdateArgs a;
    a.taps = o.logits;
    a.bias = 0.f;
    a.bias_ptr = weights->head_b;
    a.depth_t = o.mask;
    a.pred = o.pred;
    a.B_r = o.R;
    a.h = o.h;
    a.w = o.w;
    a.min_depth = cfg->min_depth;
    a.max_depth = cfg->max_depth;
    a.bit_scale = cfg->bit_scale;
    a.scale_up = (cfg->flags & DDP_FLAG_DEPTH_SCALE_UP) ? 1 : 0;      // decode_head.py:252-262
    a.eps_depth = (cfg->flags & DDP_FLAG_DEPTH_NO_EPS) ? (a.scale_up ? 1.0f : 0.0f) : (a.scale_up ? cfg->max_depth : cfg->min_depth);
    a.st = stp;
    return a;
  };
  for (int s = 0; s < o.K; ++s) {
    const ddp_step& sp = steps[s];
    const float* aff = o.aff + size_t(s) * o.L * 512;
    // feat = transform(cat[x, mask_t])
    bool depth_head = false;
    if (depth_chain) {
      DepthHeadArgs ha;
      ha.rvpad = o.rvpad;
      ha.rs = o.rs0;
      ha.wv = o.wvs;
      ha.ws = o.wvs + 256;
      ha.py = o.py[0];
      ha.px = o.px[0];
      ha.dvec = o.mask;
      ha.v_out = o.vpad;
      ha.samp_out = o.samp;
      ha.R = o.R;
      ha.h = o.h;
      ha.w = o.w;
      DepthUpdateArgs prev;
      ha.upd = nullptr;
      if (s > 0) {
        prev = depth_update_args(steps[s - 1]);
        ha.upd = &prev;
      }
      DDP_TRY(launch_depth_head(ha, st));
      depth_head = true;
    } else if (cfg->task == DDP_TASK_DEPTH && o.fused_pro) {
      // down conv over cat[x, depth_t] (depther/ddp.py:236-237) = hoisted x half + ONE depth column: q is formed inside
      // the layer-0 projection kernel (k_layer MODE 3): no feat / SB-conversion / VALUE / SAMP launches
      L0ProjLaunch pj;
      pj.Q = o.q;
      pj.stream = o.pro_stream + size_t(8) * 48 * 1024;
      pj.bias_ext = o.pro_bias;
      pj.res = o.xproj;
      pj.res_rn = o.r > 1 ? o.r * o.N : 0;
      pj.wm = o.wm;
      pj.dvec = o.mask;
      pj.M = M0;
      pj.v_out = o.vpad;
      pj.samp_out = o.samp;
      pj.py = o.py[0];
      pj.px = o.px[0];
      pj.n_tok = o.Nh;
      pj.w = o.wh;
      // fused step boundary: the PREVIOUS step's DDIM update (from the taps its last layer's tail left) runs in front of this head
      DepthUpdateArgs prev;
      pj.upd = nullptr;
      if (depth_lt && s > 0) {
        prev = depth_update_args(steps[s - 1]);
        pj.upd = &prev;
      }
      DDP_TRY(launch_b3_l0proj(pj, st));
      depth_head = true;
    } else if (cfg->task == DDP_TASK_DEPTH) {
      DDP_TRY(launch_feat_depth(o.xproj, o.wm, o.mask, o.s, o.B, o.r, o.N, st));
      DDP_TRY(publish_q(o, o.s, st));
    } else {
      if (o.b3 && !seg_tail && !bev_chain) DDP_TRY(launch_row_to_sb(o.mask, 256, o.in_sb, M0, 256, st));
      if (bev_chain) {
        if (s > 0) {
          // m' = alpha' x0 + sigma' (m - alpha x0) / max(sigma, 1e-8) (fusion_models/ddp.py:296-297) under W_m, with the PREVIOUS step's scalars
          const ddp_step& pp = steps[s - 1];
          const float ua = pp.sigma_next / (pp.sigma > 1e-8f ? pp.sigma : 1e-8f);
          DDP_TRY(launch_bev_u_update(o.feat0, bev_code, o.tlut, o.R, geom, ua, pp.alpha_next - pp.alpha * ua, st));
        }
        DDP_TRY(launch_bev_q(o.feat0, o.s, o.q, o.R, o.r, geom, st));
      } else if (cfg->task == DDP_TASK_BEV) {
        if (o.b3)
          DDP_TRY(launch_b3_linear(o.in_sb, o.wp_m, nullptr, o.xproj, 256, o.r * o.N, o.N, o.feat0, 256, M0, 256, 256, st,
                                   TAG_XPROJ));
        else
          DDP_TRY(launch_linear(o.mask, 256, false, o.wm, 256, nullptr, o.xproj, 256, o.r * o.N, o.N, o.feat0, 256, M0, 256,
                                256, 0, st));
        DDP_TRY(launch_bev_resample(o.feat0, o.s, o.R, geom, st));
        DDP_TRY(publish_q(o, o.s, st));
      } else if (u_chain && s > 0) {
        // the previous step's fused tail already wrote q (SB) and layer 0's value map / sampling table
      } else if (o.b3 && pro_fused) {
        // q = W_m m_t + xproj -> SB, layer 0's value / sampling projections: one persistent kernel
        PrologueLaunch pl;
        pl.mask_sb = o.in_sb;
        pl.Q = o.q;
        pl.stream = o.pro_stream;
        pl.bias_ext = o.pro_bias;
        pl.res = o.xproj;
        pl.res_rn = o.r > 1 ? o.r * o.N : 0;
        pl.ubuf = u_chain && o.K > 1 ? o.ubuf : nullptr;
        pl.M = M0;
        pl.v_out = o.vpad;
        pl.samp_out = o.samp;
        pl.py = o.py[0];
        pl.px = o.px[0];
        pl.n_tok = o.Nh;
        pl.w = o.wh;
        pl.res_frag = 0;
        if (head7) {
          pl.mask_sb = nullptr;
          pl.res_frag = 1;
          pl.stream = o.head7_stream;
          DDP_TRY(launch_b3_head_nchw(pl, d_noise, d_x, weights->transform_b, st));
        } else {
          DDP_TRY(launch_b3_prologue(pl, st));
        }
      } else if (o.b3) {
        // separate concat-conv GEMM: SB for the tile-GEMM layers, and fp32 fragments for the layer kernels
        DDP_TRY(launch_b3_linear_sb(o.in_sb, o.wp_m, nullptr, o.xproj, 256, o.r * o.N, o.N, o.q_sb, o.fused_layer ? o.q : nullptr, M0,
                                    256, 256, 0, st, TAG_FEAT));
      } else {
        DDP_TRY(launch_linear_blk(o.mask, 256, false, o.wm, 256, nullptr, o.xproj, 256, o.r * o.N, o.N, o.q, M0, 256, 256, 0,
                                  st));
      }
    }
    TailLaunch tl;
    memset(&tl, 0, sizeof(tl));
    if (seg_tail) {
      tl.Q = o.q;
      tl.stream = o.tail_stream;
      tl.bias_ext = o.tail_bias;
      tl.lut = o.lut;
      // accumulation: softmax summed over the steps; otherwise the last step's scores are the output
      tl.prob = cfg->accumulation ? o.prob : o.logits;
      tl.prob_mode = cfg->accumulation ? (s == 0 ? 1 : 2) : (s == o.K - 1 ? 3 : 0);
      tl.mask_sb = u_chain ? nullptr : o.in_sb;
      tl.fuse_next = u_chain && s + 1 < o.K;
      if (tl.fuse_next) {
        tl.stream = o.tail4_stream;
        tl.bias_ext = o.tail4_bias;
        tl.ubuf = o.ubuf;
        tl.tlut = o.tlut;
        tl.res = o.xproj;
        tl.res_rn = o.r > 1 ? o.r * o.N : 0;
        // written fragment-major by the first step's head (k_layer MODE 7): one coalesced 1-KiB load per (t, g) in the tail instead of
        // 32 rows x 32 B (same box: head 0.62 -> 0.59 ms, tail -0.006 ms, +0.15 % on the batch; profiles/r05h_ab_xproj_frag.txt)
        tl.res_frag = head7 ? 1 : 0;
        tl.v_out = o.vpad;
        tl.samp_out = o.samp;
        tl.py = o.py[0];
        tl.px = o.px[0];
        tl.n_tok = o.Nh;
        tl.w = o.wh;
      }
      // FORCE_X0: [0] = the caller's decisions (read), [1] = the step's own argmax (written); RECORD_X0 alone: [0] written
      tl.x0_force = (cfg->flags & DDP_FLAG_FORCE_X0) ? o.x0_trace + size_t(s) * o.M : nullptr;
      tl.x0_idx = (cfg->flags & DDP_FLAG_FORCE_X0)   ? o.x0_trace + size_t(o.K + s) * o.M
                  : (cfg->flags & DDP_FLAG_RECORD_X0) ? o.x0_trace + size_t(s) * o.M
                                                      : nullptr;
      tl.M = M;
      tl.num_classes = o.Kc;
      tl.ldl = o.ldl;
      tl.alpha = sp.alpha;
      tl.sigma = sp.sigma;
      tl.alpha_next = sp.alpha_next;
      tl.sigma_next = sp.sigma_next;
    }
    // u chain: the step's tail runs inside the LAST layer's kernel (k_layer MODE 6) - the layer output never leaves the registers
    const bool lt_fused = (seg_tail && u_chain && o.lt_stream && !(cfg->flags & DDP_FLAG_UNFUSED_TAIL)) || depth_lt || bev_chain;
    if (lt_fused && !seg_tail) {
      tl.Q = o.q;
      tl.M = M;
      tl.num_classes = cfg->task == DDP_TASK_DEPTH ? 9 : o.Kc;
      tl.kind = cfg->task == DDP_TASK_DEPTH ? 2 : 1;
      tl.prob = cfg->task == DDP_TASK_DEPTH ? o.logits : o.prob;      // depth: the nine taps (rows of 32) k_depth_update sums
      tl.prob_mode = cfg->task == DDP_TASK_DEPTH ? 3 : (s == 0 ? 1 : 2);
      tl.threshold = cfg->threshold;
      tl.x0_idx = bev_chain ? bev_code : nullptr;
      tl.ldl = 32;
    }
    DDP_TRY(encoder_forward(weights, o, aff, st, pro_fused || depth_head, !(seg_tail || lt_fused), lt_fused ? &tl : nullptr, depth_chain));
    if (lt_fused && cfg->task == DDP_TASK_BEV) {
      // (probabilities accumulated and the step's x0 codes written by the fused tail; the next step's head updates u from them)
    } else if (seg_tail) {
      if (!lt_fused) DDP_TRY(launch_b3_tail(tl, st));
    } else if (cfg->task == DDP_TASK_SEG) {
      if (o.b3)
        DDP_TRY(launch_b3_linear(o.q_sb, o.wp_head, weights->head_b, nullptr, 0, 0, 0, o.logits, o.ldl, M, o.Kc, 256, st,
                                 TAG_HEAD));
      else
        DDP_TRY(launch_linear(o.q, 256, true, weights->head_w, 256, weights->head_b, nullptr, 0, 0, 0, o.logits, o.ldl, M,
                              o.Kc, 256, 0, st, TAG_HEAD));
      SegUpdateArgs a;
      a.logits = o.logits;
      a.ldl = o.ldl;
      a.num_classes = o.Kc;
      a.lut = o.lut;
      a.mask = o.mask;
      a.prob = o.prob;
      a.prob_mode = cfg->accumulation ? (s == 0 ? 1 : 2) : 0;
      a.step_noise = nullptr;
      // FORCE_X0: [0] = the caller's decisions (read), [1] = the step's own argmax (written); RECORD_X0 alone: [0] written
      a.x0_force = (cfg->flags & DDP_FLAG_FORCE_X0) ? o.x0_trace + size_t(s) * o.M : nullptr;
      a.x0_idx = (cfg->flags & DDP_FLAG_FORCE_X0)   ? o.x0_trace + size_t(o.K + s) * o.M
                  : (cfg->flags & DDP_FLAG_RECORD_X0) ? o.x0_trace + size_t(s) * o.M
                                                      : nullptr;
      a.sampler = cfg->sampler;
      a.st = sp;
      a.rows = M;
      if (cfg->sampler == DDP_SAMPLER_DDPM && sp.ddpm_add_noise) {
        DDP_TRY(launch_nchw_to_tok(d_step_noise + size_t(s) * M0 * 256, o.snoise, o.R, 256, o.N, st));
        a.step_noise = o.snoise;
      }
      DDP_TRY(launch_seg_update(a, st));
    } else if (cfg->task == DDP_TASK_DEPTH) {
      if (depth_lt) {
        // (the nine taps were written by the last layer's tail: k_layer MODE 9)
      } else if (o.b3) {
        DDP_TRY(launch_b3_linear(o.q_sb, o.wp_head, nullptr, nullptr, 0, 0, 0, o.logits, 32, M, 9, 256, st, TAG_HEAD));
      } else {
        DDP_TRY(launch_linear(o.q, 256, true, o.wtap, 256, nullptr, nullptr, 0, 0, 0, o.logits, 32, M, 9, 256, 0, st, TAG_HEAD));
      }
      // the update of every step but the last runs inside the next step's head (k_layer MODE 3) on the fused path; the last step's
      // (and every step's on the other paths) here: it also leaves the metric depth prediction the output is made of
      if (!(depth_lt && s + 1 < o.K)) DDP_TRY(launch_depth_update(depth_update_args(sp), st));
    } else {
      if (o.b3)
        DDP_TRY(launch_b3_linear(o.q_sb, o.wp_head, weights->head_b, nullptr, 0, 0, 0, o.logits, 32, M, o.Kc, 256, st, TAG_HEAD));
      else
        DDP_TRY(launch_linear(o.q, 256, true, weights->head_w, 256, weights->head_b, nullptr, 0, 0, 0, o.logits, 32, M, o.Kc,
                              256, 0, st, TAG_HEAD));
      BevUpdateArgs a;
      a.logits = o.logits;
      a.num_classes = o.Kc;
      a.emb = weights->embedding;
      a.mask = o.mask;
      a.prob = o.prob;
      a.first = (s == 0);
      a.R = o.R;
      a.g = geom;
      a.threshold = cfg->threshold;
      a.bit_scale = cfg->bit_scale;
      a.st = sp;
      DDP_TRY(launch_bev_update(a, st));
    }
  }
  // reduction over (steps x r) and token-major -> NCHW (ddp.py:243-245)
  if (cfg->task == DDP_TASK_SEG) {
    if (cfg->accumulation)
      DDP_TRY(launch_finalize_nchw(o.prob, o.ldl, d_out, o.B, o.r, o.Nh, o.Kc, float(o.r * o.K), st, seg_tail ? (o.Kc + 63) / 64 : 0));
    else
      DDP_TRY(launch_finalize_nchw(o.logits, o.ldl, d_out, o.B, o.r, o.Nh, o.Kc, float(o.r), st, seg_tail ? (o.Kc + 63) / 64 : 0));
  } else if (cfg->task == DDP_TASK_DEPTH) {
    DDP_TRY(launch_mean_r(o.pred, d_out, o.B, o.r, o.N, st));
  } else {
    DDP_TRY(launch_finalize_nchw(o.prob, 32, d_out, o.B, o.r, o.Nh, o.Kc, float(o.r * o.K), st));
  }
  return DDP_OK;
}

int ddp_x0_trace(const ddp_cfg* cfg, void* d_workspace, const unsigned char** d_idx) {
  DDP_TRY(validate(cfg));
  DDP_TRY(check_ptr(d_workspace, "workspace"));
  if (!d_idx || !(cfg->flags & (DDP_FLAG_RECORD_X0 | DDP_FLAG_FORCE_X0)) || cfg->task != DDP_TASK_SEG) {
    set_error("x0_trace: needs a segmentation cfg with DDP_FLAG_RECORD_X0 or DDP_FLAG_FORCE_X0");
    return DDP_E_BADCFG;
  }
  Layout o;
  carve(cfg, static_cast<float*>(d_workspace), &o);
  *d_idx = o.x0_trace;
  return DDP_OK;
}

int ddp_head_forward(const ddp_cfg* cfg, const ddp_weights* weights, const float* d_feat, const float* d_temb,
                     float* d_out, void* d_workspace, void* stream) {
  DDP_TRY(validate(cfg));
  DDP_TRY(check_weights(cfg, weights));
  DDP_TRY(check_ptr(d_workspace, "workspace"));
  DDP_TRY(check_ptr(d_feat, "feat"));
  DDP_TRY(check_ptr(d_out, "out"));
  hipStream_t st = static_cast<hipStream_t>(stream);
  Layout o;
  carve(cfg, static_cast<float*>(d_workspace), &o);
  const int M = int(o.M);
  DDP_TRY(prepare_model(cfg, weights, o, st));
  DDP_TRY(prepare_geometry(o, st));
  const float* film = nullptr;
  if (d_temb) {
    for (int l = 0; l < o.L; ++l) {
      if (!weights->layers[l].time_w) continue;
      DDP_TRY(launch_matvec(weights->layers[l].time_w, weights->layers[l].time_b, d_temb, o.film + size_t(l) * 512,
                            DDP_TIME_DIM, 512, 1, DDP_TIME_DIM, o.L * 512, 2, 0, st));
    }
    film = o.film;
  }
  DDP_TRY(fold_affine_dev(weights, o.L, 1, film, o.aff, st));
  if (cfg->task == DDP_TASK_BEV) {
    DDP_TRY(launch_nchw_to_tok(d_feat, o.feat0, o.R, 256, o.N, st));
    DDP_TRY(launch_bev_resample(o.feat0, o.s, o.R, bev_geom(cfg), st));
  } else {
    DDP_TRY(launch_nchw_to_tok(d_feat, o.s, o.R, 256, o.N, st));      // row-major staging in `s`
  }
  DDP_TRY(publish_q(o, o.s, st));
  DDP_TRY(encoder_forward(weights, o, o.aff, st));
  if (cfg->task == DDP_TASK_SEG) {
    if (o.b3)
      DDP_TRY(launch_b3_linear(o.q_sb, o.wp_head, weights->head_b, nullptr, 0, 0, 0, o.logits, o.ldl, M, o.Kc, 256, st, TAG_HEAD));
    else
      DDP_TRY(launch_linear(o.q, 256, true, weights->head_w, 256, weights->head_b, nullptr, 0, 0, 0, o.logits, o.ldl, M, o.Kc,
                            256, 0, st, TAG_HEAD));
    DDP_TRY(launch_finalize_nchw(o.logits, o.ldl, d_out, o.R, 1, o.Nh, o.Kc, 1.0f, st));
  } else if (cfg->task == DDP_TASK_DEPTH) {
    if (o.b3) DDP_TRY(launch_b3_linear(o.q_sb, o.wp_head, nullptr, nullptr, 0, 0, 0, o.logits, 32, M, 9, 256, st, TAG_HEAD));
    else DDP_TRY(launch_linear(o.q, 256, true, o.wtap, 256, nullptr, nullptr, 0, 0, 0, o.logits, 32, M, 9, 256, 0, st, TAG_HEAD));
    DepthUpdateArgs a;
    memset(&a, 0, sizeof(a));
    a.taps = o.logits;
    a.bias_ptr = weights->head_b;
    a.depth_t = nullptr;
    a.pred = d_out;  // (R,1,h,w) == (R*N)
    a.B_r = o.R;
    a.h = o.h;
    a.w = o.w;
    a.min_depth = cfg->min_depth;
    a.max_depth = cfg->max_depth;
    a.bit_scale = cfg->bit_scale;
    a.scale_up = (cfg->flags & DDP_FLAG_DEPTH_SCALE_UP) ? 1 : 0;      // decode_head.py:252-262
      a.eps_depth = (cfg->flags & DDP_FLAG_DEPTH_NO_EPS) ? (a.scale_up ? 1.0f : 0.0f) : (a.scale_up ? cfg->max_depth : cfg->min_depth);
    DDP_TRY(launch_depth_update(a, st));
  } else {
    if (o.b3)
      DDP_TRY(launch_b3_linear(o.q_sb, o.wp_head, weights->head_b, nullptr, 0, 0, 0, o.logits, 32, M, o.Kc, 256, st, TAG_HEAD));
    else
      DDP_TRY(launch_linear(o.q, 256, true, weights->head_w, 256, weights->head_b, nullptr, 0, 0, 0, o.logits, 32, M, o.Kc, 256,
                            0, st, TAG_HEAD));
    BevUpdateArgs a;
    memset(&a, 0, sizeof(a));
    a.logits = o.logits;
    a.num_classes = o.Kc;
    a.emb = weights->embedding;
    a.mask = nullptr;
    a.prob = o.prob;
    a.first = 1;
    a.R = o.R;
    a.g = bev_geom(cfg);
    a.threshold = cfg->threshold;
    a.bit_scale = cfg->bit_scale;
    DDP_TRY(launch_bev_update(a, st));
    DDP_TRY(launch_finalize_nchw(o.prob, 32, d_out, o.R, 1, o.Nh, o.Kc, 1.0f, st));
  }
  return DDP_OK;
}

int ddp_msda_forward(const float* d_value, const float* d_samp, float* d_out, int rows, int h, int w, void* stream) {
  DDP_TRY(check_ptr(d_value, "value"));
  DDP_TRY(check_ptr(d_samp, "samp"));
  DDP_TRY(check_ptr(d_out, "out"));
  if (rows < 0 || h < 1 || w < 1 || rows % (h * w)) {
    set_error("msda: rows=%d must be a multiple of h*w=%d", rows, h * w);
    return DDP_E_BADCFG;
  }
  return launch_msda_gather(d_value, d_samp, d_out, rows, h * w, h, w, static_cast<hipStream_t>(stream));
}

namespace {
struct MsdaLdsLayout {
  float *vpad, *samp_hm, *tab_y, *tab_x;
  unsigned short* out_sb;
  size_t vpad_floats, bytes;
};
int msda_lds_layout(int rows, int h, int w, char* base, MsdaLdsLayout* o) {
  if (rows < 1 || h < 1 || w < 1 || rows % (h * w)) {
    set_error("msda_lds: rows=%d must be a positive multiple of h*w=%d", rows, h * w);
    return DDP_E_BADCFG;
  }
  size_t off = 0;
  auto take = [&](size_t nbytes) {
    char* p = base ? base + off : nullptr;
    off += (nbytes + 255) / 256 * 256;
    return p;
  };
  const size_t R = size_t(rows) / (size_t(h) * w);
  o->vpad_floats = (R * (h + 2) + 1) * (w + 2) * 256 + 256;      // + one zero row below the last map, as in the sampler's workspace
  o->vpad = reinterpret_cast<float*>(take(o->vpad_floats * 4));
  o->samp_hm = reinterpret_cast<float*>(take(size_t(rows) * DDP_SAMP_STRIDE * 4));
  o->tab_y = reinterpret_cast<float*>(take(size_t(h) * 96 * 4));
  o->tab_x = reinterpret_cast<float*>(take(size_t(w) * 96 * 4));
  o->out_sb = reinterpret_cast<unsigned short*>(take((size_t(rows) + 255) / 256 * 256 * 256 * 6));
  o->bytes = off;
  return DDP_OK;
}
}  // namespace

int ddp_msda_forward_lds_workspace(int rows, int h, int w, size_t* bytes) {
  if (!bytes) {
    set_error("bytes is NULL");
    return DDP_E_NULL;
  }
  MsdaLdsLayout o;
  DDP_TRY(msda_lds_layout(rows, h, w, nullptr, &o));
  *bytes = o.bytes;
  return DDP_OK;
}

int ddp_msda_forward_lds(const float* d_value, const float* d_samp, const float* d_guess, float* d_out, int rows, int h, int w,
                         void* d_workspace, void* stream) {
  DDP_TRY(check_ptr(d_value, "value"));
  DDP_TRY(check_ptr(d_samp, "samp"));
  DDP_TRY(check_ptr(d_out, "out"));
  DDP_TRY(check_ptr(d_workspace, "workspace"));
  MsdaLdsLayout o;
  DDP_TRY(msda_lds_layout(rows, h, w, static_cast<char*>(d_workspace), &o));
  hipStream_t st = static_cast<hipStream_t>(stream);
  DDP_TRY(launch_msda_lds_adapters_in(d_value, d_samp, d_guess, o.vpad, o.vpad_floats, o.samp_hm, o.tab_y, o.tab_x, rows, h * w, h, w,
                                      st));
  // the kernel of the sampling loop, launched exactly as encoder_forward launches it
#if DDP_S_F32
  float* out_blk = reinterpret_cast<float*>(o.out_sb);            // (the SB slot is 1.5x the size of the fp32 fragments)
  DDP_TRY(launch_msda_gather_sb_pad(o.vpad, o.samp_hm, nullptr, out_blk, rows, h * w, h, w, o.tab_y, o.tab_x, d_guess ? 0 : 1, st));
  return launch_blk_to_row(out_blk, d_out, rows, st);
#else
  DDP_TRY(launch_msda_gather_sb_pad(o.vpad, o.samp_hm, o.out_sb, nullptr, rows, h * w, h, w, o.tab_y, o.tab_x, d_guess ? 0 : 1, st));
  return launch_sb_to_row(o.out_sb, d_out, rows, 256, st);
#endif
}

int ddp_linear(const float* d_a, const float* d_w, const float* d_bias, float* d_out, int m, int n, int k, int gelu,
               void* stream) {
  DDP_TRY(check_ptr(d_a, "a"));
  DDP_TRY(check_ptr(d_w, "w"));
  DDP_TRY(check_ptr(d_out, "out"));
  if (n % 4) {
    set_error("linear: n=%d must be a multiple of 4 (row stride of out)", n);
    return DDP_E_BADCFG;
  }
  return launch_linear(d_a, k, false, d_w, k, d_bias, nullptr, 0, 0, 0, d_out, n, m, n, k, gelu,
                       static_cast<hipStream_t>(stream));
}

namespace {
struct LinB3Layout {
  unsigned short *a_sb, *wsplit;
  size_t bytes;
};
int linb3_layout(int m, int n, int k, char* base, LinB3Layout* o) {
  if (m < 1 || n < 1 || k < 32 || k % 32 || n % 4) {
    set_error("linear_b3: m=%d n=%d k=%d (k must be a positive multiple of 32, n of 4)", m, n, k);
    return DDP_E_BADCFG;
  }
  const size_t mp = (size_t(m) + 255) / 256 * 256;
  size_t off = 0;
  auto take = [&](size_t nbytes) {
    char* p = base ? base + off : nullptr;
    off += (nbytes + 255) / 256 * 256;
    return p;
  };
  o->a_sb = reinterpret_cast<unsigned short*>(take(mp * k * 6));
  o->wsplit = reinterpret_cast<unsigned short*>(take(size_t(3) * n * k * 2));
  o->bytes = off;
  return DDP_OK;
}
}  // namespace

int ddp_linear_b3_workspace(int m, int n, int k, size_t* bytes) {
  if (!bytes) {
    set_error("bytes is NULL");
    return DDP_E_NULL;
  }
  LinB3Layout o;
  DDP_TRY(linb3_layout(m, n, k, nullptr, &o));
  *bytes = o.bytes;
  return DDP_OK;
}

int ddp_linear_b3(const float* d_a, const float* d_w, const float* d_bias, float* d_out, int m, int n, int k,
                  void* d_workspace, void* stream) {
  DDP_TRY(check_ptr(d_a, "a"));
  DDP_TRY(check_ptr(d_w, "w"));
  DDP_TRY(check_ptr(d_out, "out"));
  DDP_TRY(check_ptr(d_workspace, "workspace"));
  LinB3Layout o;
  DDP_TRY(linb3_layout(m, n, k, static_cast<char*>(d_workspace), &o));
  hipStream_t st = static_cast<hipStream_t>(stream);
  // operands -> exact 3-way bf16 splits (activations as SB fragments, weights K-permuted), then the tile GEMM
  DDP_TRY(launch_row_to_sb(d_a, k, o.a_sb, m, k, st));
  DDP_TRY(launch_split_weights(d_w, k, n, k, o.wsplit, st));
  SplitW w;
  w.p = o.wsplit;
  w.comp_stride = size_t(n) * k;
  return launch_b3_linear(o.a_sb, w, d_bias, nullptr, 0, 0, 0, d_out, n, m, n, k, st, TAG_GENERIC);
}

int ddp_time_embed(const ddp_weights* weights, int num_layers, const float* time_in_host, int s, float* d_temb,
                   float* d_film, float* d_scratch, void* stream) {
  if (!weights || !time_in_host || s < 1 || s > DDP_MAX_STEPS || num_layers < 0 || num_layers > DDP_MAX_LAYERS) {
    set_error("time_embed: bad arguments");
    return DDP_E_BADCFG;
  }
  DDP_TRY(check_ptr(d_temb, "temb"));
  DDP_TRY(check_ptr(d_scratch, "scratch"));
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* tin = d_scratch;
  float* u = d_scratch + 64;
  float* hid = u + align64(size_t(s) * DDP_SINU_FEATS);
  DDP_TRY(launch_write_floats(time_in_host, s, tin, st));
  return time_embed_dev(weights, d_film ? num_layers : 0, tin, s, u, hid, d_temb, d_film, st);
}

int ddp_ddim_update_seg(const float* d_logits, int ld_logits, int num_classes, const float* d_lut, float* d_mask,
                        int rows, const ddp_step* step, void* stream) {
  DDP_TRY(check_ptr(d_logits, "logits"));
  DDP_TRY(check_ptr(d_lut, "lut"));
  DDP_TRY(check_ptr(d_mask, "mask"));
  if (!step || num_classes < 1 || num_classes > 256) {
    set_error("ddim_update_seg: bad arguments");
    return DDP_E_BADCFG;
  }
  SegUpdateArgs a;
  a.logits = d_logits;
  a.ldl = ld_logits;
  a.num_classes = num_classes;
  a.lut = d_lut;
  a.mask = d_mask;
  a.prob = nullptr;
  a.prob_mode = 0;
  a.step_noise = nullptr;
  a.x0_idx = nullptr;
    a.x0_force = nullptr;
  a.sampler = DDP_SAMPLER_DDIM;
  a.st = *step;
  a.rows = rows;
  return launch_seg_update(a, static_cast<hipStream_t>(stream));
}

int ddp_seg_x0_project(const float* d_scores, int batch, int num_classes, int n_pix, const float* d_embedding, float bit_scale,
                       float* d_x0, void* stream) {
  DDP_TRY(check_ptr(d_scores, "scores"));
  DDP_TRY(check_ptr(d_embedding, "embedding"));
  DDP_TRY(check_ptr(d_x0, "x0"));
  if (batch < 1 || num_classes < 1 || n_pix < 1) {
    set_error("seg_x0_project: bad sizes (batch %d classes %d pixels %d)", batch, num_classes, n_pix);
    return DDP_E_BADCFG;
  }
  return launch_seg_x0_nchw(d_scores, d_embedding, d_x0, batch, num_classes, n_pix, bit_scale, static_cast<hipStream_t>(stream));
}

int ddp_seg_postprocess(const float* d_scores, int batch, int num_classes, int h, int w, int img_h, int img_w, int crop_h,
                        int crop_w, int out_h, int out_w, int align_corners, int flip, unsigned char* d_seg, void* stream) {
  DDP_TRY(check_ptr(d_scores, "scores"));
  if (!d_seg) {
    set_error("seg is NULL");
    return DDP_E_NULL;
  }
  if (batch < 1 || num_classes < 1 || num_classes > 256 || h < 1 || w < 1 || img_h < 1 || img_w < 1 || crop_h < 1 ||
      crop_w < 1 || crop_h > img_h || crop_w > img_w || out_h < 1 || out_w < 1 || flip < 0 || flip > 2) {
    set_error("seg_postprocess: bad geometry (B %d K %d map %dx%d image %dx%d crop %dx%d out %dx%d flip %d)", batch,
              num_classes, h, w, img_h, img_w, crop_h, crop_w, out_h, out_w, flip);
    return DDP_E_BADCFG;
  }
  SegPostArgs a;
  a.logits = d_scores;
  a.B = batch;
  a.K = num_classes;
  a.h = h;
  a.w = w;
  a.H = img_h;
  a.W = img_w;
  a.ch = crop_h;
  a.cw = crop_w;
  a.oh = out_h;
  a.ow = out_w;
  a.align = align_corners ? 1 : 0;
  a.flip = flip;
  a.seg = d_seg;
  return launch_seg_postprocess(a, static_cast<hipStream_t>(stream));
}

int ddp_seg_aug_postprocess(const ddp_seg_aug* augs, int n_aug, int batch, int num_classes, int out_h, int out_w,
                            int align_corners, unsigned char* d_seg, float* d_prob, void* stream) {
  if (!augs || !d_seg) {
    set_error("seg_aug_postprocess: augs / seg is NULL");
    return DDP_E_NULL;
  }
  if (n_aug < 1 || n_aug > DDP_MAX_AUGS || batch < 1 || num_classes < 1 || num_classes > 256 || out_h < 1 || out_w < 1) {
    set_error("seg_aug_postprocess: bad arguments (n_aug %d B %d K %d out %dx%d)", n_aug, batch, num_classes, out_h, out_w);
    return DDP_E_BADCFG;
  }
  for (int i = 0; i < n_aug; ++i) {
    const ddp_seg_aug& g = augs[i];
    DDP_TRY(check_ptr(g.d_scores, "aug scores"));
    if (g.h < 1 || g.w < 1 || g.img_h < 1 || g.img_w < 1 || g.crop_h < 1 || g.crop_w < 1 || g.crop_h > g.img_h ||
        g.crop_w > g.img_w || g.flip < 0 || g.flip > 2) {
      set_error("seg_aug_postprocess: augmentation %d: bad geometry (map %dx%d image %dx%d crop %dx%d flip %d)", i, g.h, g.w,
                g.img_h, g.img_w, g.crop_h, g.crop_w, g.flip);
      return DDP_E_BADCFG;
    }
  }
  if (d_prob) DDP_TRY(check_ptr(d_prob, "prob"));
  return launch_seg_aug_postprocess(augs, n_aug, batch, num_classes, out_h, out_w, align_corners ? 1 : 0, d_seg, d_prob,
                                    static_cast<hipStream_t>(stream));
}

int ddp_seg_slide_postprocess(const float* const* d_scores, const int* win_y1, const int* win_x1, int n_rows, int n_cols, int batch,
                              int num_classes, int h, int w, int crop_h, int crop_w, int img_h, int img_w, int keep_h, int keep_w,
                              int out_h, int out_w, int align_corners, int flip, int prob_mode, unsigned char* d_seg, float* d_prob,
                              void* stream) {
  if (!d_scores || !win_y1 || !win_x1 || (!d_seg && !d_prob)) {
    set_error("seg_slide_postprocess: scores / window origins / both outputs NULL");
    return DDP_E_NULL;
  }
  if (n_rows < 1 || n_cols < 1 || n_rows * n_cols > DDP_MAX_WINDOWS || batch < 1 || num_classes < 1 || num_classes > 256 || h < 1 ||
      w < 1 || crop_h < 1 || crop_w < 1 || crop_h > img_h || crop_w > img_w || keep_h < 1 || keep_w < 1 || keep_h > img_h ||
      keep_w > img_w || out_h < 1 || out_w < 1 || flip < 0 || flip > 2 || prob_mode < 0 || prob_mode > 2 || (prob_mode && !d_prob)) {
    set_error("seg_slide_postprocess: bad geometry (%dx%d windows of %dx%d on %dx%d, B %d K %d map %dx%d keep %dx%d out %dx%d flip %d "
              "prob_mode %d)", n_rows, n_cols, crop_h, crop_w, img_h, img_w, batch, num_classes, h, w, keep_h, keep_w, out_h, out_w, flip,
              prob_mode);
    return DDP_E_BADCFG;
  }
  // every image pixel covered, by at most 4 window rows / columns (the kernel keeps that many per tap in registers)
  for (int axis = 0; axis < 2; ++axis) {
    const int* o = axis ? win_x1 : win_y1;
    const int n = axis ? n_cols : n_rows, crop = axis ? crop_w : crop_h, size = axis ? img_w : img_h;
    int covered = 0;
    for (int i = 0; i < n; ++i) {
      if (o[i] < 0 || o[i] + crop > size || (i > 0 && o[i] < o[i - 1]) || o[i] > covered) {
        set_error("seg_slide_postprocess: window origins along axis %d must be ascending, inside the image and leave no gap", axis);
        return DDP_E_BADCFG;
      }
      covered = o[i] + crop;
      if (i >= 4 && o[i - 4] + crop > o[i]) {
        set_error("seg_slide_postprocess: more than 4 windows overlap along axis %d (stride < crop / 4)", axis);
        return DDP_E_BADCFG;
      }
    }
    if (covered < size) {
      set_error("seg_slide_postprocess: the windows do not cover the image along axis %d", axis);
      return DDP_E_BADCFG;
    }
  }
  for (int i = 0; i < n_rows * n_cols; ++i) DDP_TRY(check_ptr(d_scores[i], "window scores"));
  if (d_prob) DDP_TRY(check_ptr(d_prob, "prob"));
  return launch_seg_slide_postprocess(d_scores, win_y1, win_x1, n_rows, n_cols, batch, num_classes, h, w, crop_h, crop_w, img_h, img_w,
                                      keep_h, keep_w, out_h, out_w, align_corners ? 1 : 0, flip, prob_mode, d_seg, d_prob,
                                      static_cast<hipStream_t>(stream));
}

int ddp_depth_postprocess(const ddp_depth_aug* augs, int n_aug, int batch, int out_h, int out_w, int align_corners,
                          float min_depth, float max_depth, float* d_out, void* stream) {
  if (!augs) {
    set_error("depth_postprocess: augs is NULL");
    return DDP_E_NULL;
  }
  DDP_TRY(check_ptr(d_out, "out"));
  if (n_aug < 1 || n_aug > DDP_MAX_AUGS || batch < 1 || out_h < 1 || out_w < 1 || !(min_depth <= max_depth)) {
    set_error("depth_postprocess: bad arguments (n_aug %d B %d out %dx%d depth range [%g, %g])", n_aug, batch, out_h, out_w,
              double(min_depth), double(max_depth));
    return DDP_E_BADCFG;
  }
  for (int i = 0; i < n_aug; ++i) {
    const ddp_depth_aug& g = augs[i];
    DDP_TRY(check_ptr(g.d_depth, "aug depth"));
    if (g.h < 1 || g.w < 1 || g.flip < 0 || g.flip > 2) {
      set_error("depth_postprocess: augmentation %d: bad geometry (map %dx%d flip %d)", i, g.h, g.w, g.flip);
      return DDP_E_BADCFG;
    }
  }
  return launch_depth_aug_postprocess(augs, n_aug, batch, out_h, out_w, align_corners ? 1 : 0, min_depth, max_depth, d_out,
                                      static_cast<hipStream_t>(stream));
}

namespace {
struct MsmLayout {
  // weight region first (independent of batch and map sizes: DDP_NECK_WEIGHTS_READY)
  unsigned short* wsplit;
  unsigned char* stream[4];   // per level: its 256 x 256 block of the 1x1 conv as 8 wide stage images
  // activations
  float* a[4];                // level inputs, fp32 fragment-major
  float* y[4];                // per-level conv outputs at the level's own resolution (y[0]: the merged map)
  double* partial;
  float* stats;
  size_t wbytes, abytes, bytes;
};
// weights are carved from wbase, activations from abase (nullptr: sizes only); o->wbytes / o->abytes = the two region sizes
static int msm_layout(int batch, const int* lh, const int* lw, char* wbase, char* abase, MsmLayout* o) {
  if (batch < 1 || !lh || !lw) {
    set_error("neck_msm: bad arguments");
    return DDP_E_BADCFG;
  }
  for (int l = 0; l < 4; ++l)
    if (lh[l] < 1 || lw[l] < 1) {
      set_error("neck_msm: level %d has size %dx%d", l, lh[l], lw[l]);
      return DDP_E_BADCFG;
    }
  char* base = wbase;
  size_t off = 0;
  auto take = [&](size_t nbytes) {
    char* p = base ? base + off : nullptr;
    off += (nbytes + 255) / 256 * 256;
    return p;
  };
  o->wsplit = reinterpret_cast<unsigned short*>(take(size_t(3) * 256 * 256 * 2));
  for (int l = 0; l < 4; ++l) o->stream[l] = reinterpret_cast<unsigned char*>(take(size_t(8) * b3_stage_bytes()));
  o->wbytes = off;
  base = abase;
  off = 0;
  for (int l = 0; l < 4; ++l) {
    const size_t M = size_t(batch) * lh[l] * lw[l], Mp = (M + 255) / 256 * 256;
    o->a[l] = reinterpret_cast<float*>(take(Mp * 256 * sizeof(float)));
    o->y[l] = reinterpret_cast<float*>(take(Mp * 256 * sizeof(float)));
  }
  const size_t chunks = (size_t(lh[0]) * lw[0] + 31) / 32;   // (32-token chunks: the merging kernel writes the statistics itself)
  o->partial = reinterpret_cast<double*>(take(size_t(batch) * chunks * 64 * sizeof(double)));
  o->stats = reinterpret_cast<float*>(take(size_t(batch) * 64 * sizeof(float)));
  o->abytes = off;
  o->bytes = o->wbytes + o->abytes;
  return DDP_OK;
}
}  // namespace

int ddp_neck_msm_workspace(int batch, const int* level_h, const int* level_w, size_t* bytes) {
  if (!bytes) {
    set_error("bytes is NULL");
    return DDP_E_NULL;
  }
  MsmLayout o;
  DDP_TRY(msm_layout(batch, level_h, level_w, nullptr, nullptr, &o));
  *bytes = o.bytes;
  return DDP_OK;
}

namespace {
// the merging itself on fp32 fragment-major level inputs a[0..3] (see ddp_neck_msm)
// (a_nchw: the level inputs are the caller's NCHW tensors, read in place by the stream GEMM - no layout conversion)
int msm_core(const float* const* a_blk, const int* level_h, const int* level_w, int batch, const float* d_conv_w, const float* d_gn_w,
             const float* d_gn_b, int align_corners, int flags, float* d_out, const MsmLayout& o, hipStream_t st, bool a_nchw = false) {
  const int N = level_h[0] * level_w[0];
  if (!(flags & DDP_NECK_WEIGHTS_READY))
    for (int l = 0; l < 4; ++l) {
      DDP_TRY(launch_split_weights(d_conv_w + 256 * l, 1024, 256, 256, o.wsplit, st));
      DDP_TRY(launch_build_stages(o.wsplit, size_t(256) * 256, 256, 256, 0, 1, 8, 0, 2, 1, 0, o.stream[l], st));
    }
  SgemmProblem pr[4];
  for (int l = 0; l < 4; ++l) {
    pr[l].A = a_blk[l];
    pr[l].out = o.y[l];
    pr[l].stream = o.stream[l];
    pr[l].M = batch * level_h[l] * level_w[l];
    pr[l].ns = 8;
    pr[l].conv_h = pr[l].conv_w = 0;
    pr[l].bias = nullptr;
    pr[l].gn_partial = nullptr;                            // (GroupNorm acts on the SUM of the resized level outputs)
    pr[l].gn_N = 0;
    pr[l].nchw_N = a_nchw ? level_h[l] * level_w[l] : 0;
  }
  DDP_TRY(launch_b3_sgemm(pr, 4, 0, 0, st));               // all four levels in one persistent launch
  const float* yl[3] = {o.y[1], o.y[2], o.y[3]};
  DDP_TRY(launch_msm_sum_blk(o.y[0], yl, level_h + 1, level_w + 1, batch, level_h[0], level_w[0], align_corners ? 1 : 0, st, o.partial));
  if (N % 32 == 0) DDP_TRY(launch_gn_final32(o.partial, o.stats, batch, N, 1e-5f, st));      // (statistics fused into the merging kernel)
  else DDP_TRY(launch_gn_stats_blk(o.y[0], o.partial, o.stats, batch, N, 1e-5f, st));
  return launch_gn_apply_nchw_blk(o.y[0], o.stats, d_gn_w, d_gn_b, d_out, batch, N, st);
}
}  // namespace

int ddp_neck_msm(const float* const* d_levels, const int* level_h, const int* level_w, int batch, const float* d_conv_w,
                 const float* d_gn_w, const float* d_gn_b, int align_corners, int flags, float* d_out, void* d_workspace,
                 void* stream) {
  if (!d_levels) {
    set_error("levels is NULL");
    return DDP_E_NULL;
  }
  for (int l = 0; l < 4; ++l) DDP_TRY(check_ptr(d_levels[l], "level"));
  DDP_TRY(check_ptr(d_conv_w, "conv weight"));
  DDP_TRY(check_ptr(d_gn_w, "gn weight"));
  DDP_TRY(check_ptr(d_gn_b, "gn bias"));
  DDP_TRY(check_ptr(d_out, "out"));
  DDP_TRY(check_ptr(d_workspace, "workspace"));
  if (flags & ~DDP_NECK_WEIGHTS_READY) {
    set_error("neck_msm: unknown flags 0x%x", flags);
    return DDP_E_BADCFG;
  }
  MsmLayout o;
  DDP_TRY(msm_layout(batch, level_h, level_w, nullptr, nullptr, &o));
  DDP_TRY(msm_layout(batch, level_h, level_w, static_cast<char*>(d_workspace), static_cast<char*>(d_workspace) + o.wbytes, &o));
  hipStream_t st = static_cast<hipStream_t>(stream);
  // The 1x1 conv over the concatenation of the four resized levels is linear: it equals the sum over the levels of the
  // bilinear resize of (that level's 256-column block of the weight applied at the level's OWN resolution).  Levels 1..3 are
  // 4x / 16x / 64x smaller than level 0, so the contraction shrinks from 4 to 1.33 level-0 maps and the 1024-channel
  // concatenation (6 KiB per token as a split operand) is never built.
  return msm_core(d_levels, level_h, level_w, batch, d_conv_w, d_gn_w, d_gn_b, align_corners, flags, d_out, o, st, true);
}

namespace {
struct FpnLayout {
  // weight region first (sizes depend on the level channels only: DDP_NECK_WEIGHTS_READY)
  float* wpack;
  unsigned short* wsplit;
  unsigned char* lat_stream[4];   // lateral 1x1 conv: C_l / 32 wide stage images
  unsigned char* out_stream[4];   // 3x3 output conv: 72 stage images (tap, 32-channel block)
  // activations, fp32 fragment-major
  float* a[4];                    // level inputs (C_l channels)
  float* y[4];                    // lateral conv output, later the 3x3 conv output
  float* lat[4];                  // laterals after GroupNorm + top-down add
  double* partial[4];             // per level: GroupNorm partial sums (chunks of 32 tokens when fused into the GEMM, else 256)
  float* stats[4];
  size_t wbytes, abytes, bytes;
};
// weights are carved from wbase, activations from abase (nullptr: sizes only); o->wbytes / o->abytes = the two region sizes
static int fpn_layout(const ddp_fpn_level* lv, int batch, char* wbase, char* abase, FpnLayout* o) {
  if (!lv || batch < 1) {
    set_error("neck_fpn: bad arguments");
    return DDP_E_BADCFG;
  }
  char* base = wbase;
  size_t off = 0;
  auto take = [&](size_t nbytes) {
    char* p = base ? base + off : nullptr;
    off += (nbytes + 255) / 256 * 256;
    return p;
  };
  size_t max_w = 2304;
  for (int l = 0; l < 4; ++l) {
    const ddp_fpn_level& v = lv[l];
    if (v.h < 1 || v.w < 1 || v.in_channels < 64 || v.in_channels % 32 || v.in_channels > 4096) {
      set_error("neck_fpn: level %d: %d channels, %dx%d (channels must be a multiple of 32 in [64, 4096])", l, v.in_channels, v.h, v.w);
      return DDP_E_BADCFG;
    }
    if (size_t(v.in_channels) > max_w) max_w = v.in_channels;
  }
  o->wpack = reinterpret_cast<float*>(take(size_t(256) * max_w * 4));
  o->wsplit = reinterpret_cast<unsigned short*>(take(size_t(3) * 256 * max_w * 2));
  for (int l = 0; l < 4; ++l) {
    o->lat_stream[l] = reinterpret_cast<unsigned char*>(take(size_t(lv[l].in_channels / 32) * b3_stage_bytes()));
    o->out_stream[l] = reinterpret_cast<unsigned char*>(take(size_t(72) * b3_stage_bytes()));
  }
  o->wbytes = off;
  base = abase;
  off = 0;
  for (int l = 0; l < 4; ++l) {
    const ddp_fpn_level& v = lv[l];
    const size_t M = size_t(batch) * v.h * v.w, Mp = (M + 255) / 256 * 256;
    o->a[l] = reinterpret_cast<float*>(take(Mp * v.in_channels * 4));
    o->y[l] = reinterpret_cast<float*>(take(Mp * 256 * 4));
    o->lat[l] = reinterpret_cast<float*>(take(Mp * 256 * 4));
    o->stats[l] = reinterpret_cast<float*>(take(size_t(batch) * 64 * 4));
    o->partial[l] = reinterpret_cast<double*>(take(size_t(batch) * ((size_t(v.h) * v.w + 31) / 32) * 64 * sizeof(double)));
  }
  o->abytes = off;
  o->bytes = o->wbytes + o->abytes;
  return DDP_OK;
}
}  // namespace

int ddp_neck_fpn_workspace(const ddp_fpn_level* levels, int batch, size_t* bytes) {
  if (!bytes) {
    set_error("bytes is NULL");
    return DDP_E_NULL;
  }
  FpnLayout o;
  DDP_TRY(fpn_layout(levels, batch, nullptr, nullptr, &o));
  *bytes = o.bytes;
  return DDP_OK;
}

namespace {
// d_out != nullptr: the four outputs as NCHW tensors; else they stay fp32 fragment-major in o.lat[l] (consumed by msm_core)
int fpn_core(const ddp_fpn_level* levels, int batch, const float* const* d_in, float* const* d_out, int flags, const FpnLayout& o,
             hipStream_t st) {
  for (int l = 0; l < 4; ++l) {
    const ddp_fpn_level& v = levels[l];
    DDP_TRY(check_ptr(d_in[l], "level input"));
    if (d_out) DDP_TRY(check_ptr(d_out[l], "level output"));
    DDP_TRY(check_ptr(v.lat_w, "lateral weight"));
    DDP_TRY(check_ptr(v.out_w, "fpn conv weight"));
  }
  // weights -> stage images of the stream GEMM (once per set of weights: DDP_NECK_WEIGHTS_READY skips this)
  if (!(flags & DDP_NECK_WEIGHTS_READY))
    for (int l = 0; l < 4; ++l) {
      const ddp_fpn_level& v = levels[l];
      const int C = v.in_channels;
      DDP_TRY(launch_split_weights(v.lat_w, C, 256, C, o.wsplit, st));
      DDP_TRY(launch_build_stages(o.wsplit, size_t(256) * C, C, 256, 0, 1, C / 32, 0, 2, 1, 0, o.lat_stream[l], st));
      DDP_TRY(launch_pack_conv3x3_scaled(v.out_w, nullptr, o.wpack, 256, 256, st));
      DDP_TRY(launch_split_weights(o.wpack, 2304, 256, 2304, o.wsplit, st));
      DDP_TRY(launch_build_stages(o.wsplit, size_t(256) * 2304, 2304, 256, 0, 1, 72, 0, 2, 1, 0, o.out_stream[l], st));
    }
  // laterals (fpn.py:167-171): 1x1 conv of all four levels in ONE persistent launch of the stream GEMM (k_layer MODE 5; the
  // coarse levels have few tiles of many stages: longest tiles first), GroupNorm statistics per level
  SgemmProblem pr[4];
  for (int l = 0; l < 4; ++l) {
    const ddp_fpn_level& v = levels[3 - l];
    pr[l].A = d_in[3 - l];                                  // the backbone's NCHW level, read in place (k_layer MODE 5, nchw_N)
    pr[l].nchw_N = v.h * v.w;
    pr[l].out = o.y[3 - l];
    pr[l].stream = o.lat_stream[3 - l];
    pr[l].M = batch * v.h * v.w;
    pr[l].ns = v.in_channels / 32;
    pr[l].conv_h = pr[l].conv_w = 0;
    pr[l].bias = nullptr;
    // GroupNorm partial sums in the GEMM epilogue when a wave's 32 tokens cannot straddle two images
    pr[l].gn_N = v.h * v.w;
    pr[l].gn_partial = (v.h * v.w) % 32 == 0 ? o.partial[3 - l] : nullptr;
  }
  DDP_TRY(launch_b3_sgemm(pr, 4, 0, 0, st));
  // GroupNorm + top-down path (:173-185): lat_l = GN(y_l) + nearest_up(lat_{l+1}), coarsest level first, one kernel per level
  // (the statistics of all four levels are finalised by ONE launch when the GEMM epilogue wrote every level's partial sums)
  bool all32 = true;
  int Nl[4];
  for (int l = 0; l < 4; ++l) {
    Nl[l] = levels[l].h * levels[l].w;
    all32 = all32 && Nl[l] % 32 == 0;
  }
  if (all32) DDP_TRY(launch_gn_final32_multi(o.partial, o.stats, Nl, 4, batch, 1e-5f, st));
  for (int l = 3; l >= 0; --l) {
    const ddp_fpn_level& v = levels[l];
    if (all32) {
    } else if ((v.h * v.w) % 32 == 0) DDP_TRY(launch_gn_final32(o.partial[l], o.stats[l], batch, v.h * v.w, 1e-5f, st));
    else DDP_TRY(launch_gn_stats_blk(o.y[l], o.partial[l], o.stats[l], batch, v.h * v.w, 1e-5f, st));
    DDP_TRY(launch_gn_apply_add_blk(o.y[l], o.stats[l], v.lat_gn_w, v.lat_gn_b, l < 3 ? o.lat[l + 1] : nullptr, o.lat[l], batch, v.h,
                                    v.w, l < 3 ? levels[l + 1].h : 1, l < 3 ? levels[l + 1].w : 1, st));
  }
  // outputs (:189-191): 3x3 conv as an implicit GEMM (72 stage images = (tap, 32-channel block); the A fragments of a stage
  // are the fp32 quads of the token shifted by the tap, zero padding), all four levels in one launch, then GroupNorm
  for (int l = 0; l < 4; ++l) {
    const ddp_fpn_level& v = levels[l];
    pr[l].A = o.lat[l];
    pr[l].out = o.y[l];
    pr[l].stream = o.out_stream[l];
    pr[l].M = batch * v.h * v.w;
    pr[l].ns = 72;
    pr[l].conv_h = v.h;
    pr[l].conv_w = v.w;
    pr[l].nchw_N = 0;
    pr[l].gn_N = v.h * v.w;
    pr[l].gn_partial = (v.h * v.w) % 32 == 0 ? o.partial[l] : nullptr;
  }
  DDP_TRY(launch_b3_sgemm(pr, 4, 0, 1, st));
  if (all32) DDP_TRY(launch_gn_final32_multi(o.partial, o.stats, Nl, 4, batch, 1e-5f, st));
  for (int l = 0; l < 4; ++l) {
    const ddp_fpn_level& v = levels[l];
    if (all32) {
    } else if ((v.h * v.w) % 32 == 0) DDP_TRY(launch_gn_final32(o.partial[l], o.stats[l], batch, v.h * v.w, 1e-5f, st));
    else DDP_TRY(launch_gn_stats_blk(o.y[l], o.partial[l], o.stats[l], batch, v.h * v.w, 1e-5f, st));
    if (d_out) DDP_TRY(launch_gn_apply_nchw_blk(o.y[l], o.stats[l], v.out_gn_w, v.out_gn_b, d_out[l], batch, v.h * v.w, st));
    else       // (the laterals are dead once the convolution has run: their buffers take the normalised outputs)
      DDP_TRY(launch_gn_apply_add_blk(o.y[l], o.stats[l], v.out_gn_w, v.out_gn_b, nullptr, o.lat[l], batch, v.h, v.w, 1, 1, st));
  }
  return DDP_OK;
}
}  // namespace

int ddp_neck_fpn(const ddp_fpn_level* levels, int batch, const float* const* d_in, float* const* d_out, int flags, void* d_workspace,
                 void* stream) {
  if (!d_in || !d_out) {
    set_error("neck_fpn: in / out is NULL");
    return DDP_E_NULL;
  }
  DDP_TRY(check_ptr(d_workspace, "workspace"));
  if (flags & ~DDP_NECK_WEIGHTS_READY) {
    set_error("neck_fpn: unknown flags 0x%x", flags);
    return DDP_E_BADCFG;
  }
  FpnLayout o;
  DDP_TRY(fpn_layout(levels, batch, nullptr, nullptr, &o));
  DDP_TRY(fpn_layout(levels, batch, static_cast<char*>(d_workspace), static_cast<char*>(d_workspace) + o.wbytes, &o));
  return fpn_core(levels, batch, d_in, d_out, flags, o, static_cast<hipStream_t>(stream));
}

// FPN followed by MultiStageMerging, as every DDP config chains them (configs/ade/ddp_swin_t_2x8_512x512_160k_ade20k.py neck list):
// the four FPN outputs stay fp32 fragment-major and feed the merging directly - their NCHW form (one write by FPN, one read
// and re-layout by the merging per level) is never produced.
int ddp_neck_fpn_msm_workspace(const ddp_fpn_level* levels, int batch, size_t* bytes) {
  if (!bytes) {
    set_error("bytes is NULL");
    return DDP_E_NULL;
  }
  FpnLayout f;
  DDP_TRY(fpn_layout(levels, batch, nullptr, nullptr, &f));
  int lh[4], lw[4];
  for (int l = 0; l < 4; ++l) {
    lh[l] = levels[l].h;
    lw[l] = levels[l].w;
  }
  MsmLayout m;
  DDP_TRY(msm_layout(batch, lh, lw, nullptr, nullptr, &m));
  *bytes = f.bytes + m.bytes;
  return DDP_OK;
}

int ddp_neck_fpn_msm(const ddp_fpn_level* levels, int batch, const float* const* d_in, const float* d_msm_conv_w,
                     const float* d_msm_gn_w, const float* d_msm_gn_b, int align_corners, int flags, float* d_out, void* d_workspace,
                     void* stream) {
  if (!d_in) {
    set_error("neck_fpn_msm: in is NULL");
    return DDP_E_NULL;
  }
  DDP_TRY(check_ptr(d_msm_conv_w, "msm conv weight"));
  DDP_TRY(check_ptr(d_msm_gn_w, "msm gn weight"));
  DDP_TRY(check_ptr(d_msm_gn_b, "msm gn bias"));
  DDP_TRY(check_ptr(d_out, "out"));
  DDP_TRY(check_ptr(d_workspace, "workspace"));
  if (flags & ~DDP_NECK_WEIGHTS_READY) {
    set_error("neck_fpn_msm: unknown flags 0x%x", flags);
    return DDP_E_BADCFG;
  }
  // workspace = [FPN weights | MSM weights | FPN activations | MSM activations]: both weight regions at offsets that do not
  // depend on the geometry (DDP_NECK_WEIGHTS_READY survives a change of batch / map size)
  FpnLayout f;
  DDP_TRY(fpn_layout(levels, batch, nullptr, nullptr, &f));
  int lh[4], lw[4];
  for (int l = 0; l < 4; ++l) {
    lh[l] = levels[l].h;
    lw[l] = levels[l].w;
  }
  MsmLayout m;
  DDP_TRY(msm_layout(batch, lh, lw, nullptr, nullptr, &m));
  char* wsb = static_cast<char*>(d_workspace);
  const size_t fw = f.wbytes, mw = m.wbytes, fa = f.abytes;
  DDP_TRY(fpn_layout(levels, batch, wsb, wsb + fw + mw, &f));
  DDP_TRY(msm_layout(batch, lh, lw, wsb + fw, wsb + fw + mw + fa, &m));
  hipStream_t st = static_cast<hipStream_t>(stream);
  DDP_TRY(fpn_core(levels, batch, d_in, nullptr, flags, f, st));
  return msm_core(f.lat, lh, lw, batch, d_msm_conv_w, d_msm_gn_w, d_msm_gn_b, align_corners, flags, d_out, m, st);
}

namespace {
struct FcnLayout {
  float *x0, *xb0, *xb1, *wpack, *film, *aff, *logits, *cls_bias;
  unsigned short *a_sb, *wsplit;
  unsigned char* stream;     // stage images of the GEMM being run (72 for a 3x3 conv, 8 for conv_seg)
  int ldl;
  size_t bytes;
};
static int fcn_layout(int maps, int h, int w, int K, char* base, FcnLayout* o) {
  if (maps < 1 || h < 1 || w < 1 || K < 1 || K > 256) {
    set_error("fcn_head: bad geometry (maps %d map %dx%d classes %d)", maps, h, w, K);
    return DDP_E_BADCFG;
  }
  size_t off = 0;
  auto take = [&](size_t nbytes) {
    char* p = base ? base + off : nullptr;
    off += (nbytes + 255) / 256 * 256;
    return p;
  };
  const size_t M = size_t(maps) * h * w, Mp = (M + 255) / 256 * 256;
  o->ldl = (K + 31) / 32 * 32;
  o->x0 = reinterpret_cast<float*>(take(Mp * 256 * 4));        // row-major input (the sampler's concat-conv writes it)
  o->xb0 = reinterpret_cast<float*>(take(Mp * 256 * 4));       // fp32 fragment-major activations, ping / pong
  o->xb1 = reinterpret_cast<float*>(take(Mp * 256 * 4));
  o->a_sb = reinterpret_cast<unsigned short*>(take(Mp * 256 * 6));   // (SB staging of the sampler's concat-conv input)
  o->wpack = reinterpret_cast<float*>(take(size_t(256) * 2304 * 4));
  o->wsplit = reinterpret_cast<unsigned short*>(take(size_t(3) * 256 * 2304 * 2));
  o->stream = reinterpret_cast<unsigned char*>(take(size_t(72) * b3_stage_bytes()));
  o->film = reinterpret_cast<float*>(take(512 * 4));
  o->aff = reinterpret_cast<float*>(take(512 * 4));
  o->cls_bias = reinterpret_cast<float*>(take(256 * 4));
  o->logits = reinterpret_cast<float*>(take(Mp * o->ldl * 4));
  o->bytes = off;
  return DDP_OK;
}
}  // namespace

int ddp_fcn_head_workspace(int maps, int h, int w, int num_classes, size_t* bytes) {
  if (!bytes) {
    set_error("bytes is NULL");
    return DDP_E_NULL;
  }
  FcnLayout o;
  DDP_TRY(fcn_layout(maps, h, w, num_classes, nullptr, &o));
  *bytes = o.bytes;
  return DDP_OK;
}

namespace {
// constants of one head evaluation that depend on (weights, time embedding) only: per conv the 72 stage images of the scaled
// 3x3 weights and the shift vector, conv_seg's 8 images and its padded bias
struct FcnPrepared {
  const unsigned char* conv_stream[8];
  const float* conv_shift[8];
  const unsigned char* cls_stream;
  const float* cls_bias;
};
// one ConvWithTimeModule (fcn_head_with_time.py:205-225): FiLM vector from the time embedding, eval BatchNorm x FiLM folded
// into (scale, shift), the scale folded into the 3x3 weights, those split and laid out as the 72 stage images of the stream
// GEMM.  scratch = o.{film, aff, wpack, wsplit}
int fcn_conv_prepare(const ddp_fcn_conv& c, int i, const float* d_temb, const FcnLayout& o, unsigned char* stream_dst,
                     float* shift_dst, hipStream_t st) {
  DDP_TRY(check_ptr(c.conv_w, "conv weight"));
  if (c.bn_w && (!c.bn_b || !c.bn_mean || !c.bn_var)) {
    set_error("fcn_head: conv %d has an incomplete norm", i);
    return DDP_E_NULL;
  }
  const float* film = nullptr;
  if (d_temb && c.time_w) {      // (:216-221) SiLU -> Linear(1024, 512) -> (scale | shift)
    DDP_TRY(launch_matvec(c.time_w, c.time_b, d_temb, o.film, DDP_TIME_DIM, 512, 1, DDP_TIME_DIM, 512, 2, 0, st));
    film = o.film;
  }
  DDP_TRY(launch_fcn_fold(c.bn_w, c.bn_b, c.bn_mean, c.bn_var, c.bn_eps, c.conv_b, film, o.aff, shift_dst, st));
  DDP_TRY(launch_pack_conv3x3_scaled(c.conv_w, o.aff, o.wpack, 256, 256, st));
  DDP_TRY(launch_split_weights(o.wpack, 2304, 256, 2304, o.wsplit, st));
  return launch_build_stages(o.wsplit, size_t(256) * 2304, 2304, 256, 0, 1, 72, 0, 2, 1, 0, stream_dst, st);
}
// conv_seg (cls_seg; dropout is the identity in eval mode): 1x1, the class rows zero-padded to the GEMM's 256 outputs
int fcn_cls_prepare(const float* d_cls_w, const float* d_cls_b, int num_classes, const FcnLayout& o, unsigned char* stream_dst,
                    float* bias_dst, hipStream_t st) {
  DDP_TRY(launch_split_weights(d_cls_w, 256, num_classes, 256, o.wsplit, st));
  DDP_TRY(launch_build_stages(o.wsplit, size_t(num_classes) * 256, 256, num_classes, 0, 1, 8, 0, 2, 1, 0, stream_dst, st));
  if (hipMemsetAsync(bias_dst, 0, 256 * sizeof(float), st) != hipSuccess ||
      (d_cls_b && hipMemcpyAsync(bias_dst, d_cls_b, size_t(num_classes) * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)) {
    set_error("fcn_head: conv_seg bias copy failed");
    return DDP_E_LAUNCH;
  }
  return DDP_OK;
}

// FCNHeadWithTime on token-major rows: o.x0 (M,256) in -> o.logits (M, ldl) (fcn_head_with_time.py:285-305, eval mode).  Inside,
// the activations are fp32 fragment-major and every convolution runs on the persistent stream GEMM (k_layer MODE 5): the 3x3
// ones as implicit GEMMs of 72 stages with the folded norm x FiLM shift as bias and ReLU in the epilogue, conv_seg as 8 stages.
// `prep`: the weight-side constants built beforehand (the sampler loop: ddp_prepare_fcn); NULL: the time embedding is a
// run-time input (ddp_fcn_head_forward) and they are built here, one conv at a time through the workspace's single stream
// buffer (conv i's images are consumed - stream order - before conv i + 1's are written).
int fcn_head_tokens(const ddp_fcn_conv* convs, int num_convs, int dilation, const float* d_cls_w, const float* d_cls_b,
                    int num_classes, const float* d_temb, int maps, int h, int w, const FcnLayout& o, hipStream_t st,
                    const FcnPrepared* prep = nullptr) {
  const int N = h * w, M = maps * N;
  DDP_TRY(launch_row_to_blk(o.x0, o.xb0, M, st));
  float* cur = o.xb0;
  float* nxt = o.xb1;
  SgemmProblem pr;
  pr.gn_partial = nullptr;
  pr.gn_N = 0;
  pr.nchw_N = 0;
  pr.M = M;
  for (int i = 0; i < num_convs; ++i) {
    if (!prep) DDP_TRY(fcn_conv_prepare(convs[i], i, d_temb, o, o.stream, o.aff + 256, st));
    pr.A = cur;
    pr.out = nxt;
    pr.stream = prep ? prep->conv_stream[i] : o.stream;
    pr.ns = 72;
    pr.conv_h = h;
    pr.conv_w = w;
    pr.bias = prep ? prep->conv_shift[i] : o.aff + 256;
    DDP_TRY(launch_b3_sgemm(&pr, 1, 2, dilation, st));
    float* t = cur;
    cur = nxt;
    nxt = t;
  }
  if (!prep) DDP_TRY(fcn_cls_prepare(d_cls_w, d_cls_b, num_classes, o, o.stream, o.cls_bias, st));
  pr.A = cur;
  pr.out = nxt;
  pr.stream = prep ? prep->cls_stream : o.stream;
  pr.ns = 8;
  pr.conv_h = pr.conv_w = 0;
  pr.bias = prep ? prep->cls_bias : o.cls_bias;
  DDP_TRY(launch_b3_sgemm(&pr, 1, 0, 0, st));
  return launch_blk_to_row(nxt, o.logits, M, st, o.ldl, o.ldl);
}

// sampler loop around the FCN head: the head's own workspace first, then the loop's buffers
struct FcnLoopLayout {
  FcnLayout head;
  float *tin, *u, *hid, *temb, *lut, *wx, *wm, *xtok, *xproj, *mask, *prob, *snoise;
  unsigned short *wx_split, *wm_split, *in_sb;
  // what ddp_prepare_fcn leaves behind for ddp_sample_fcn (besides temb, lut, wx_split, wm_split): per (step, conv) the stage
  // images of the FiLM-scaled 3x3 weights and the shift vector; conv_seg's images and padded bias
  unsigned char *conv_streams, *cls_stream;
  float *conv_shifts, *cls_bias;
  size_t bytes;
};
int fcn_loop_layout(const ddp_cfg* c, int num_convs, char* base, FcnLoopLayout* o) {
  const int R = c->batch * c->randsteps, N = c->h * c->w, Kc = c->num_classes, Cx = c->feat_channels;
  DDP_TRY(fcn_layout(R, c->h, c->w, Kc, base, &o->head));
  size_t off = o->head.bytes;
  auto take = [&](size_t nbytes) {
    char* p = base ? base + off : nullptr;
    off += (nbytes + 255) / 256 * 256;
    return p;
  };
  auto takef = [&](size_t floats) { return reinterpret_cast<float*>(take(floats * sizeof(float))); };
  const size_t M = size_t(R) * N, Mp = (M + 255) / 256 * 256, MB = size_t(c->batch) * N, MBp = (MB + 255) / 256 * 256;
  o->tin = takef(DDP_MAX_STEPS);
  o->u = takef(size_t(c->timesteps) * DDP_SINU_FEATS);
  o->hid = takef(size_t(c->timesteps) * DDP_TIME_DIM);
  o->temb = takef(size_t(c->timesteps) * DDP_TIME_DIM);
  o->lut = takef(size_t(Kc + 1) * 256);
  o->wx = takef(size_t(256) * Cx);
  o->wm = takef(size_t(256) * 256);
  o->wx_split = reinterpret_cast<unsigned short*>(take(size_t(3) * 256 * Cx * 2));
  o->wm_split = reinterpret_cast<unsigned short*>(take(size_t(3) * 256 * 256 * 2));
  o->xtok = takef(MB * Cx);
  o->xproj = takef(MB * 256);
  o->mask = takef(M * 256);
  o->prob = takef(M * o->head.ldl);
  o->snoise = takef(c->sampler == DDP_SAMPLER_DDPM ? M * 256 : 0);
  const size_t sb_a = MBp * Cx, sb_b = Mp * 256;
  o->in_sb = reinterpret_cast<unsigned short*>(take((sb_a > sb_b ? sb_a : sb_b) * 6));
  const size_t per_conv = size_t(72) * b3_stage_bytes();
  o->conv_streams = reinterpret_cast<unsigned char*>(take(size_t(c->timesteps) * num_convs * per_conv));
  o->conv_shifts = takef(size_t(c->timesteps) * (num_convs > 0 ? num_convs : 1) * 256);
  o->cls_stream = reinterpret_cast<unsigned char*>(take(size_t(8) * b3_stage_bytes()));
  o->cls_bias = takef(256);
  o->bytes = off;
  return DDP_OK;
}
FcnPrepared fcn_prepared_of(const FcnLoopLayout& o, int num_convs, int step) {
  FcnPrepared p;
  const size_t per_conv = size_t(72) * b3_stage_bytes();
  for (int i = 0; i < 8; ++i) {
    const bool live = i < num_convs;
    p.conv_stream[i] = live ? o.conv_streams + (size_t(step) * num_convs + i) * per_conv : nullptr;
    p.conv_shift[i] = live ? o.conv_shifts + (size_t(step) * num_convs + i) * 256 : nullptr;
  }
  p.cls_stream = o.cls_stream;
  p.cls_bias = o.cls_bias;
  return p;
}
int validate_fcn_loop(const ddp_cfg* cfg, int num_convs, int dilation) {
  if (!cfg) {
    set_error("cfg is NULL");
    return DDP_E_NULL;
  }
  ddp_cfg c = *cfg;
  c.num_layers = 1;                       // the encoder depth is meaningless here
  c.flags &= ~DDP_FLAG_FCN_PREPARED;      // this loop's own flag (ddp_sample rejects it)
  DDP_TRY(validate(&c));
  if (cfg->task != DDP_TASK_SEG || cfg->head_h != cfg->h || cfg->head_w != cfg->w) {
    set_error("sample_fcn: segmentation only (FCNHeadWithTime is a segmentation head)");
    return DDP_E_BADCFG;
  }
  if (num_convs < 0 || num_convs > 8 || dilation < 1) {
    set_error("sample_fcn: num_convs %d / dilation %d out of range", num_convs, dilation);
    return DDP_E_BADCFG;
  }
  return DDP_OK;
}
}  // namespace

int ddp_fcn_head_forward(const ddp_fcn_conv* convs, int num_convs, int dilation, const float* d_cls_w, const float* d_cls_b,
                         int num_classes, const float* d_feat, const float* d_temb, int maps, int h, int w, float* d_out,
                         void* d_workspace, void* stream) {
  if (num_convs < 0 || num_convs > 8 || dilation < 1 || (num_convs > 0 && !convs)) {
    set_error("fcn_head: num_convs %d / dilation %d out of range", num_convs, dilation);
    return DDP_E_BADCFG;
  }
  DDP_TRY(check_ptr(d_cls_w, "conv_seg weight"));
  DDP_TRY(check_ptr(d_feat, "feat"));
  DDP_TRY(check_ptr(d_out, "out"));
  DDP_TRY(check_ptr(d_workspace, "workspace"));
  FcnLayout o;
  DDP_TRY(fcn_layout(maps, h, w, num_classes, static_cast<char*>(d_workspace), &o));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int N = h * w;
  DDP_TRY(launch_nchw_to_tok(d_feat, o.x0, maps, 256, N, st));
  DDP_TRY(fcn_head_tokens(convs, num_convs, dilation, d_cls_w, d_cls_b, num_classes, d_temb, maps, h, w, o, st));
  return launch_finalize_nchw(o.logits, o.ldl, d_out, maps, 1, N, num_classes, 1.0f, st);
}

int ddp_sample_fcn_workspace(const ddp_cfg* cfg, int num_convs, int dilation, size_t* bytes) {
  DDP_TRY(validate_fcn_loop(cfg, num_convs, dilation));
  if (!bytes) {
    set_error("bytes is NULL");
    return DDP_E_NULL;
  }
  FcnLoopLayout o;
  DDP_TRY(fcn_loop_layout(cfg, num_convs, nullptr, &o));
  *bytes = o.bytes;
  return DDP_OK;
}

namespace {
int check_fcn_loop_args(const ddp_cfg* cfg, const ddp_weights* weights, const ddp_fcn_conv* convs, int num_convs, int dilation,
                        const ddp_step* steps, const void* d_workspace) {
  DDP_TRY(validate_fcn_loop(cfg, num_convs, dilation));
  if (!weights || !steps || (num_convs > 0 && !convs)) {
    set_error("sample_fcn: weights / steps / convs is NULL");
    return DDP_E_NULL;
  }
  DDP_TRY(check_ptr(weights->transform_w, "transform_w"));
  DDP_TRY(check_ptr(weights->transform_b, "transform_b"));
  DDP_TRY(check_ptr(weights->embedding, "embedding"));
  DDP_TRY(check_ptr(weights->head_w, "conv_seg weight"));
  DDP_TRY(check_ptr(weights->time1_w, "time_mlp.1"));
  DDP_TRY(check_ptr(weights->time3_w, "time_mlp.3"));
  return check_ptr(d_workspace, "workspace");
}
int prepare_fcn(const ddp_cfg* cfg, const ddp_weights* weights, const ddp_fcn_conv* convs, int num_convs, const ddp_step* steps,
                const FcnLoopLayout& o, hipStream_t st) {
  const int K = cfg->timesteps, Kc = cfg->num_classes, Cx = cfg->feat_channels;
  // time embeddings of every step, x0 LUT, the two column blocks of the concat-conv
  float tin[DDP_MAX_STEPS];
  for (int s = 0; s < K; ++s) tin[s] = steps[s].time_in;
  DDP_TRY(launch_write_floats(tin, K, o.tin, st));
  DDP_TRY(time_embed_dev(weights, 0, o.tin, K, o.u, o.hid, o.temb, nullptr, st));
  DDP_TRY(launch_build_lut(weights->embedding, o.lut, Kc + 1, cfg->bit_scale, st));
  DDP_TRY(launch_pack_cols(weights->transform_w, Cx + 256, 0, 256, Cx, o.wx, st));
  DDP_TRY(launch_pack_cols(weights->transform_w, Cx + 256, Cx, 256, 256, o.wm, st));
  DDP_TRY(launch_split_weights(o.wx, Cx, 256, Cx, o.wx_split, st));
  DDP_TRY(launch_split_weights(o.wm, 256, 256, 256, o.wm_split, st));
  // the head's weight-side constants: one set of scaled-weight images per (step, conv) - the FiLM scale depends on the step
  const size_t per_conv = size_t(72) * b3_stage_bytes();
  for (int s = 0; s < K; ++s)
    for (int i = 0; i < num_convs; ++i)
      DDP_TRY(fcn_conv_prepare(convs[i], i, o.temb + size_t(s) * DDP_TIME_DIM, o.head,
                               o.conv_streams + (size_t(s) * num_convs + i) * per_conv,
                               o.conv_shifts + (size_t(s) * num_convs + i) * 256, st));
  return fcn_cls_prepare(weights->head_w, weights->head_b, Kc, o.head, o.cls_stream, o.cls_bias, st);
}
}  // namespace

int ddp_prepare_fcn(const ddp_cfg* cfg, const ddp_weights* weights, const ddp_fcn_conv* convs, int num_convs, int dilation,
                    const ddp_step* steps, void* d_workspace, void* stream) {
  DDP_TRY(check_fcn_loop_args(cfg, weights, convs, num_convs, dilation, steps, d_workspace));
  FcnLoopLayout o;
  DDP_TRY(fcn_loop_layout(cfg, num_convs, static_cast<char*>(d_workspace), &o));
  return prepare_fcn(cfg, weights, convs, num_convs, steps, o, static_cast<hipStream_t>(stream));
}

int ddp_sample_fcn(const ddp_cfg* cfg, const ddp_weights* weights, const ddp_fcn_conv* convs, int num_convs, int dilation,
                   const ddp_step* steps, const float* d_x, const float* d_noise, const float* d_step_noise, float* d_out,
                   void* d_workspace, void* stream) {
  DDP_TRY(check_fcn_loop_args(cfg, weights, convs, num_convs, dilation, steps, d_workspace));
  DDP_TRY(check_ptr(d_x, "x"));
  DDP_TRY(check_ptr(d_noise, "noise"));
  DDP_TRY(check_ptr(d_out, "out"));
  if (cfg->sampler == DDP_SAMPLER_DDPM) DDP_TRY(check_ptr(d_step_noise, "step_noise"));
  hipStream_t st = static_cast<hipStream_t>(stream);
  FcnLoopLayout o;
  DDP_TRY(fcn_loop_layout(cfg, num_convs, static_cast<char*>(d_workspace), &o));
  const int B = cfg->batch, r = cfg->randsteps, K = cfg->timesteps, Kc = cfg->num_classes, Cx = cfg->feat_channels;
  const int N = cfg->h * cfg->w, R = B * r, M = R * N;
  if (!(cfg->flags & DDP_FLAG_FCN_PREPARED)) DDP_TRY(prepare_fcn(cfg, weights, convs, num_convs, steps, o, st));
  SplitW wpx, wpm;
  wpx.p = o.wx_split;
  wpx.comp_stride = size_t(256) * Cx;
  wpm.p = o.wm_split;
  wpm.comp_stride = size_t(256) * 256;
  // loop-invariant half of the concat-conv (ddp.py:223-224 with the x columns hoisted), start noise -> token-major
  DDP_TRY(launch_nchw_to_tok(d_x, o.xtok, B, Cx, N, st));
  DDP_TRY(launch_row_to_sb(o.xtok, Cx, o.in_sb, B * N, Cx, st));
  DDP_TRY(launch_b3_linear(o.in_sb, wpx, weights->transform_b, nullptr, 0, 0, 0, o.xproj, 256, B * N, 256, Cx, st, TAG_XPROJ));
  DDP_TRY(launch_nchw_to_tok(d_noise, o.mask, R, 256, N, st));
  for (int s = 0; s < K; ++s) {
    const ddp_step& sp = steps[s];
    // feat = transform(cat[x, mask_t]) -> the head's token-major input
    DDP_TRY(launch_row_to_sb(o.mask, 256, o.in_sb, M, 256, st));
    DDP_TRY(launch_b3_linear(o.in_sb, wpm, nullptr, o.xproj, 256, r * N, N, o.head.x0, 256, M, 256, 256, st, TAG_XPROJ));
    const FcnPrepared prep = fcn_prepared_of(o, num_convs, s);
    DDP_TRY(fcn_head_tokens(convs, num_convs, dilation, weights->head_w, weights->head_b, Kc, o.temb + size_t(s) * DDP_TIME_DIM, R,
                            cfg->h, cfg->w, o.head, st, &prep));
    SegUpdateArgs a;
    a.logits = o.head.logits;
    a.ldl = o.head.ldl;
    a.num_classes = Kc;
    a.lut = o.lut;
    a.mask = o.mask;
    a.prob = o.prob;
    a.prob_mode = cfg->accumulation ? (s == 0 ? 1 : 2) : 0;
    a.step_noise = nullptr;
    a.x0_idx = nullptr;
    a.x0_force = nullptr;
    a.sampler = cfg->sampler;
    a.st = sp;
    a.rows = M;
    if (cfg->sampler == DDP_SAMPLER_DDPM && sp.ddpm_add_noise) {
      DDP_TRY(launch_nchw_to_tok(d_step_noise + size_t(s) * M * 256, o.snoise, R, 256, N, st));
      a.step_noise = o.snoise;
    }
    DDP_TRY(launch_seg_update(a, st));
  }
  if (cfg->accumulation) return launch_finalize_nchw(o.prob, o.head.ldl, d_out, B, r, N, Kc, float(r * K), st);
  return launch_finalize_nchw(o.head.logits, o.head.ldl, d_out, B, r, N, Kc, float(r), st);
}

}  // extern "C"
