/* ddp_mi355x.h - C ABI of libddp_mi355x.so: the MI355X (gfx950) implementation of the DDP
 * multi-step denoising inference loop.
 *
 * The reference has no FFI for this path: the loop is Python/torch
 * (segmentation/mmseg/models/segmentors/ddp.py:215-246 `DDP.ddim_sample`, :248-290 `ddpm_sample`;
 * depth/depth/models/depther/ddp.py:229-247 `DDP.sample`;
 * bev/mmdet3d/models/fusion_models/ddp.py:268-301 `DDP.ddim_sample`) calling
 * `DeformableHeadWithTime.forward` (segmentation/mmseg/models/decode_heads/deformable_head_with_time.py:90-132)
 * and, below it, mmcv's only custom kernel `ext_module.ms_deform_attn_forward`
 * (controlnet/annotator/uniformer/mmcv/ops/multi_scale_deform_attn.py:47-53).  This header is the
 * boundary a maintainer would bind instead (ctypes stub: INTEGRATION.md); `ddp_amd`'s registered
 * drop-in classes call exactly these symbols.
 *
 * Conventions
 *  - plain C, no torch types.  Every pointer named `d_*` or living in `ddp_weights` is a DEVICE
 *    pointer to contiguous fp32, 16-byte aligned.  The caller (PyTorch) owns every buffer,
 *    including the workspace; the library never allocates or frees device memory.
 *  - all work is enqueued on the `hipStream_t` passed as `void* stream` (0 = default stream);
 *    no host synchronisation happens inside any entry point.
 *  - return value: DDP_OK (0) or a negative DDP_E_* code; `ddp_last_error()` gives a message.
 *    No exception crosses the boundary.
 *  - thread-safety: re-entrant per (workspace, stream); no global mutable state except the
 *    thread-local last-error string and the measurement hook at the end of this header
 *    (ddp_profile_*: ONE process-wide session, armed by bench.py only - not thread-safe, see there).
 *  - fixed architecture constants of the reference configs: embed 256, 8 heads x 32, 4 points,
 *    1 level, FFN 1024, time dim 1024, 16 learned sinusoid features.
 */
#ifndef DDP_MI355X_H
#define DDP_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: exactly the entry points declared here are exported */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define DDP_ABI_VERSION 5
#define DDP_MAX_LAYERS 12
#define DDP_MAX_STEPS 64
#define DDP_EMBED 256
#define DDP_HEADS 8
#define DDP_POINTS 4
#define DDP_FFN 1024
#define DDP_TIME_DIM 1024
#define DDP_SINU_FEATS 17 /* 1 + 16 */
#define DDP_SAMP_STRIDE 96 /* per token: 64 sample coords (head,point,xy) + 32 weights */

enum { DDP_OK = 0, DDP_E_BADCFG = -1, DDP_E_ALIGN = -2, DDP_E_LAUNCH = -3, DDP_E_NULL = -4 };
enum { DDP_TASK_SEG = 0, DDP_TASK_DEPTH = 1, DDP_TASK_BEV = 2 };
enum { DDP_SAMPLER_DDIM = 0, DDP_SAMPLER_DDPM = 1 };
/* How the channel contractions are evaluated (both are fp32-accurate; csrc/gemm_f32.h, csrc/gemm_bf16x3.h):
 *   DDP_GEMM_F32_MFMA   exact f32-input MFMA (v_mfma_f32_32x32x2_f32), 157 TFLOP/s ceiling
 *   DDP_GEMM_BF16X3     every fp32 operand is split exactly into 3 bf16 pieces and the 6 significant cross
 *                       products run on the bf16 matrix cores with fp32 accumulation (error <= ~3 * 2^-24 per
 *                       product, i.e. fp32 round-off class), 417 TFLOP/s fp32-equivalent ceiling */
enum { DDP_GEMM_F32_MFMA = 0, DDP_GEMM_BF16X3 = 1 };
/* ddp_cfg.flags (diagnostics, bf16x3 engine): run the decoder layer / the head of a step as the separate tile GEMMs they
 * were fused from (identical arithmetic per contraction; used by same-box A/B runs and by the parity tests that keep
 * the unfused kernels covered).  UNFUSED_LAYER implies the unfused step head and seg tail as well.
 * GATHER_GUESS_ZERO: the LDS-staged gather starts every window from a zero guess of the head's mean sampling offset instead
 * of the mean of its offset biases, which forces its "actual mean is far from the guess: refill" branch (identical results:
 * the window origin only decides which taps are served from LDS).
 * FORCE_X0 (seg, test instrument): teacher forcing.  Every step feeds back the class the CALLER supplies (ddp_x0_trace's
 * buffer, filled before ddp_sample) instead of its own argmax (ddp.py:235); scores, softmax accumulation, update and
 * everything else are unchanged, and the step's OWN argmax is still recorded next to the supplied one.  With the
 * decisions of a reference run supplied, the loop has no discontinuity left and the outputs must agree with that run to
 * rounding (tests/test_full_size_parity.py).  Separate kernel instantiations: the product's tails are not touched. */
enum { DDP_FLAG_UNFUSED_LAYER = 1, DDP_FLAG_UNFUSED_PROLOGUE = 2, DDP_FLAG_RECORD_X0 = 4, DDP_FLAG_GATHER_GUESS_ZERO = 8,
       DDP_FLAG_FCN_PREPARED = 16 /* ddp_sample_fcn: the workspace holds what ddp_prepare_fcn wrote */,
       DDP_FLAG_FORCE_X0 = 32,
       /* depth head variants (depth/depth/models/decode_heads/decode_head.py:252-262; model configuration, not diagnostics): */
       DDP_FLAG_DEPTH_SCALE_UP = 256 /* depth = sigmoid(conv_depth) * eps, eps = max_depth (or 1 with NO_EPS) instead of relu(conv_depth) + eps */,
       DDP_FLAG_DEPTH_NO_EPS = 512 /* use_eps=False: eps = 0 (relu branch) / 1 (scale_up branch) instead of min_depth / max_depth */,
       DDP_FLAG_SB_HEAD = 128 /* the first step's head as the four launches it was fused from (NCHW -> split fragments of x and of
                                  the start noise, the x-projection GEMM, k_layer MODE 2) instead of ONE kernel that reads the
                                  caller's NCHW tensors directly (k_layer MODE 7); A/B runs and parity tests of the separate kernels */,
       DDP_FLAG_UNFUSED_TAIL = 64 /* the step boundary as the launches it was fused from (same-box A/B runs, parity tests of the separate
                                     kernels).  seg: the last decoder layer of a step and the step's tail as two kernels (k_layer MODE 0 +
                                     MODE 4 / 1 instead of MODE 6; identical arithmetic, bit-identical results).  depth: the GEMM step
                                     head (k_layer MODE 3), the 9-tap head GEMM on the SB layer output and k_depth_update per step instead
                                     of k_depth_head + k_layer MODE 10 / MODE 9.  bev: the concat-conv GEMM, grid resampling, head GEMM and
                                     k_bev_update on the 256-channel map per step instead of the u chain (k_bev_u_update, k_bev_q, k_layer
                                     MODE 8).  depth / bev: the same operators regrouped - results agree to rounding, not bit for bit */ };

/* Problem description.  Mirrors the constructor kwargs of the reference `DDP` classes
 * (segmentors/ddp.py:57-67; depther/ddp.py:42-54; fusion_models/ddp.py:67-80). */
typedef struct ddp_cfg {
  int32_t abi_version;    /* must be DDP_ABI_VERSION */
  int32_t task;           /* DDP_TASK_* */
  int32_t sampler;        /* DDP_SAMPLER_* (ddpm: seg only) */
  int32_t batch;          /* B independent images (the reference loop is b = 1 per call) */
  int32_t randsteps;      /* r noise replicas per image */
  int32_t timesteps;      /* K sampling steps, <= DDP_MAX_STEPS */
  int32_t num_layers;     /* L encoder layers, <= DDP_MAX_LAYERS */
  int32_t num_classes;    /* K_cls (seg, bev <= 256); ignored for depth */
  int32_t feat_channels;  /* channels of x (multiple of 32) */
  int32_t h, w;           /* spatial size of x / of the noisy map */
  int32_t head_h, head_w; /* token grid of the encoder: == h,w except bev (grid transform output) */
  int32_t accumulation;   /* seg: average softmax over steps (ddp.py:241-245); bev always accumulates */
  float bit_scale;
  float min_depth, max_depth; /* depth */
  float threshold;            /* bev x0 threshold (fusion_models/ddp.py:290) */
  /* bev grid transform (heads/segm/deformable_head_with_time.py:70-97): per axis (y then x)
   * input [min,max], output first centre and step: c_k = out_first + k * out_step */
  float bev_in_min[2], bev_in_max[2], bev_out_first[2], bev_out_step[2];
  int32_t gemm_mode;          /* DDP_GEMM_* */
  int32_t flags;              /* DDP_FLAG_* (0 = the product path) */
} ddp_cfg;

typedef struct ddp_layer_weights {            /* decode_head.encoder.layers.<l>.* */
  const float *sampling_offsets_w, *sampling_offsets_b;   /* (64,256),(64)  attentions.0.sampling_offsets */
  const float *attention_weights_w, *attention_weights_b; /* (32,256),(32)  attentions.0.attention_weights */
  const float *value_proj_w, *value_proj_b;               /* (256,256),(256) */
  const float *output_proj_w, *output_proj_b;             /* (256,256),(256) */
  const float *ffn0_w, *ffn0_b;                           /* (1024,256),(1024) ffns.0.layers.0.0 */
  const float *ffn1_w, *ffn1_b;                           /* (256,1024),(256)  ffns.0.layers.1 */
  const float *norm0_w, *norm0_b, *norm1_w, *norm1_b;     /* (256) each        norms.{0,1} */
  const float *time_w, *time_b;                           /* (512,1024),(512)  time_mlp.1 ; may be NULL */
} ddp_layer_weights;

typedef struct ddp_weights {
  const float *transform_w, *transform_b; /* (256, Cx+Cm) 1x1 conv as matrix, (256): transform.conv / down.conv */
  const float *time_freq;                 /* (8)         time_mlp.0.weights */
  const float *time1_w, *time1_b;         /* (1024,17),(1024) time_mlp.1 */
  const float *time3_w, *time3_b;         /* (1024,1024),(1024) time_mlp.3 */
  const float *embedding;                 /* (K_cls+1,256) embedding_table.weight; NULL for depth */
  const float *head_w, *head_b;           /* seg/bev conv_seg (K_cls,256),(K_cls); depth conv_depth (1,256,3,3),(1) */
  ddp_layer_weights layers[DDP_MAX_LAYERS];
} ddp_weights;

/* Host-computed per-step schedule scalars.  The cosine log-SNR is ill-conditioned at t = 1, so the
 * Python host layer evaluates the schedule with torch CPU fp32 ops in the reference's op order
 * (segmentors/ddp.py:22-28,225-231) and hands the scalars over; the library never re-derives them.
 *   seg/bev : time_in = log_snr(t_now); alpha, sigma, alpha_next, sigma_next as in ddp.py:229-231
 *   depth   : time_in = t_now; alpha = sqrt(gamma_now), sigma = 1/sqrt(1-gamma_now),
 *             alpha_next = sqrt(gamma_next), sigma_next = sqrt(1-gamma_next)  (depther/ddp.py:220-227)
 *   ddpm    : ddpm_c = -expm1(ls - ls_next), ddpm_std = exp(0.5*log(max(sigma_next^2*c,1e-20))),
 *             ddpm_add_noise = (t_next > 0)                                     (ddp.py:276-284) */
typedef struct ddp_step {
  float time_in, alpha, sigma, alpha_next, sigma_next, ddpm_c, ddpm_std;
  int32_t ddpm_add_noise;
} ddp_step;

const char* ddp_last_error(void);
int ddp_abi_version(void);

/* Size in bytes of the caller-provided workspace for `cfg` (constants + activations). */
int ddp_query_workspace(const ddp_cfg* cfg, size_t* bytes);

/* Fill the constant region of the workspace: time embeddings and per-layer FiLM vectors for every
 * step (ddp.py:31-46,107-112; utils/transformer.py:275-278), sine positional tables folded through
 * the offset/attention projections (utils/transformer.py:78-113), the x0 look-up table
 * (ddp.py:236-237) and packed projection weights.  Must be re-run when weights, (h,w) or the
 * schedule change; `steps` is a HOST array of cfg->timesteps entries. */
int ddp_prepare(const ddp_cfg* cfg, const ddp_weights* weights, const ddp_step* steps,
                void* d_workspace, void* stream);

/* Serving the reference's own test protocol - ONE image per call, a different (h, w) almost every call
 * (segmentation/tools/test.py:214-219 forces samples_per_gpu = 1; configs/_base_/datasets/ade20k.py:23 resizes keep-ratio):
 * the workspace starts with a MODEL region (everything ddp_prepare derives from the weights and the schedule: time
 * embeddings, FiLM, folded affines, x0 LUT and T = LUT.W_m^T, split weight planes, the layer kernels' weight streams) whose
 * size and content do not depend on batch / randsteps / (h, w) / (head_h, head_w) / the bev grid.  When only those change:
 *   - keep a buffer of at least ddp_query_workspace(new cfg) bytes whose first ddp_query_const_workspace(cfg) bytes are the
 *     prepared model region (same buffer, or a device-to-device copy of that prefix into a larger one),
 *   - call ddp_prepare_geometry(new cfg, ...) : L positional-table launches + one memset, no weight is touched.
 * Every other cfg field (task, sampler, timesteps, num_layers, num_classes, feat_channels, gemm_mode, flags, bit_scale, ...)
 * must be unchanged since ddp_prepare. */
int ddp_query_const_workspace(const ddp_cfg* cfg, size_t* bytes);
int ddp_prepare_geometry(const ddp_cfg* cfg, void* d_workspace, void* stream);

/* The whole K-step loop for B images: replaces DDP.ddim_sample / ddpm_sample / sample.
 *   d_x          (B, Cx, h, w)           frozen neck feature, NCHW
 *   d_noise      (B, r, Cm, h, w)        start noise (the reference draws torch.randn in-method)
 *   d_step_noise (K, B, r, Cm, h, w)     per-step noise, ddpm only, else NULL
 *   d_out        seg: (B, K_cls, h, w); depth: (B, 1, h, w); bev: (B, K_cls, head_h, head_w) */
int ddp_sample(const ddp_cfg* cfg, const ddp_weights* weights, const ddp_step* steps,
               const float* d_x, const float* d_noise, const float* d_step_noise, float* d_out,
               void* d_workspace, void* stream);

/* Diagnostic trace (seg, cfg->flags & DDP_FLAG_RECORD_X0): after ddp_sample, *d_idx points at (K, B*r*h*w) uint8 inside
 * the workspace - the x0 class (argmax of the step's scores, ddp.py:235) every step fed back.  The loop's only
 * discontinuity is this discrete choice; with it recorded, parity splits into "same decisions -> outputs agree to
 * rounding" and "decisions differ only where the reference's own top-2 gap is at rounding level"
 * (tests/test_full_size_parity.py).
 * With DDP_FLAG_FORCE_X0 the buffer is (2, K, B*r*h*w): [0] = the decisions to feed back, written by the caller BEFORE
 * ddp_sample (values < num_classes); [1] = the argmax the engine itself found at every step under that forcing.  The library
 * cannot tell whether [0] was filled: a caller that sets DDP_FLAG_FORCE_X0 and does not write it feeds back whatever the
 * workspace held (values >= num_classes are read as class 0) - ddp_amd.engine.DDPEngine refuses to sample until
 * set_x0_decisions() ran.  Both flags are rejected (DDP_E_BADCFG) for tasks other than DDP_TASK_SEG.
 * The workspace LAYOUT (which bytes hold what, incl. this buffer's place) is an implementation detail of one build: it is
 * queried, never assumed, and may change between library builds of the same DDP_ABI_VERSION. */
int ddp_x0_trace(const ddp_cfg* cfg, void* d_workspace, const unsigned char** d_idx);

/* ---- finer-grained entry points (unit tests, and the decode_head plugin surface) ------------- */

/* DeformableHeadWithTime.forward: feat (R,256,hh,wh) NCHW + time embedding (1024) -> head output
 * seg: logits (R,K_cls,hh,wh); depth: metric depth (R,1,hh,wh); bev: sigmoid maps.  Uses step slot 0
 * of the workspace constants for FiLM, recomputed from d_temb.  R = batch*randsteps. */
int ddp_head_forward(const ddp_cfg* cfg, const ddp_weights* weights, const float* d_feat,
                     const float* d_temb, float* d_out, void* d_workspace, void* stream);

/* One deformable attention core (mmcv ms_deform_attn_forward for 1 level):
 *   d_value (R, N, 256) token-major; d_samp (R*N, 96): per token 64 pixel-unit sample coordinates
 *   [head][point][x,y] followed by 32 softmaxed weights [head][point]; d_out (R*N, 256). */
int ddp_msda_forward(const float* d_value, const float* d_samp, float* d_out, int rows, int h, int w,
                     void* stream);

/* The same core computed by the kernel the SAMPLING LOOP runs (k_msda_gather_lds: one head per block, the head's 128-B
 * slices of a window of the zero-padded value map staged in LDS, taps outside the window served from global memory), behind
 * the same plain interface: the entry pads the map, transposes the table to head-major, launches the loop's gather exactly
 * as ddp_sample does and converts its split-bf16 output back to fp32 rows (exact).  d_guess: (8,2) per-head (x, y) guess of
 * the mean sampling offset that positions the LDS window before the table is read (the loop derives it from the offset bias
 * and the positional tables), or NULL = zero guess; results do not depend on it (it only decides which taps hit LDS).
 * This is the entry the adversarial tests use: on-pixel, far-outside, NaN and border-straddling sample points. */
int ddp_msda_forward_lds_workspace(int rows, int h, int w, size_t* bytes);
int ddp_msda_forward_lds(const float* d_value, const float* d_samp, const float* d_guess, float* d_out, int rows, int h, int w,
                         void* d_workspace, void* stream);

/* out[M][N] = A[M][K] * W[N][K]^T + bias  (fp32 MFMA), optional exact GELU. K % 32 == 0. */
int ddp_linear(const float* d_a, const float* d_w, const float* d_bias, float* d_out, int m, int n, int k,
               int gelu, void* stream);

/* The same contraction in the arithmetic of the DEFAULT engine (DDP_GEMM_BF16X3, csrc/gemm_bf16x3.h): A and W are split
 * exactly into three bf16 pieces each and the six significant cross products run on the bf16 matrix cores with fp32
 * accumulation.  Exposed so that the accuracy claim above is a unit test (tests/test_b3_arithmetic.py: error against
 * fp64 <= c * 2^-24 * sum_k |a||w| on normal-range, wide-dynamic-range, cancelling and near-subnormal operands), not a
 * comment.  out[M][N] = A[M][K] * W[N][K]^T + bias; K % 32 == 0, N % 4 == 0; d_workspace from ddp_linear_b3_workspace. */
int ddp_linear_b3_workspace(int m, int n, int k, size_t* bytes);
int ddp_linear_b3(const float* d_a, const float* d_w, const float* d_bias, float* d_out, int m, int n, int k,
                  void* d_workspace, void* stream);

/* time_mlp + per-layer FiLM on device: d_temb (S,1024), d_film (S,L,512) for S time inputs. */
int ddp_time_embed(const ddp_weights* weights, int num_layers, const float* time_in_host, int s,
                   float* d_temb, float* d_film, float* d_scratch /* >= S*(17+1024) floats */, void* stream);

/* DDIM x0-projection + update for seg on token-major buffers (ddp.py:235-239):
 *   d_logits (M, ld_logits), d_lut (K_cls+1... rows of 256), d_mask (M,256) updated in place. */
int ddp_ddim_update_seg(const float* d_logits, int ld_logits, int num_classes, const float* d_lut,
                        float* d_mask, int rows, const ddp_step* step, void* stream);

/* x0 projection alone (segmentors/ddp.py:235-237), NCHW: d_x0 (B,256,h,w) = (sigmoid(embedding[argmax_k d_scores]) * 2 - 1) *
 * bit_scale for d_scores (B,K,h,w).  With the scores of ONE decoder pass at t = 1 this is the self-aligned pre-pass of
 * SelfAlignedDDP (segmentors/self_aligned_ddp.py:150-164: t = 1 head forward -> argmax -> embedding -> bit scaling). */
int ddp_seg_x0_project(const float* d_scores, int batch, int num_classes, int n_pix, const float* d_embedding,
                       float bit_scale, float* d_x0, void* stream);

/* Post-loop epilogue of the segmentor, fused: replaces
 *   resize(out, img.shape[2:]) (segmentors/ddp.py:124-128), whole_inference's crop to img_shape + resize to
 *   ori_shape (encoder_decoder.py:236-248), softmax (:277), flip (:278-285) and argmax (:296).
 * d_scores (B,K,h,w) = ddp_sample's output; image (img_h,img_w) = padded network input; crop = img_shape;
 * out = ori_shape (== crop: no second resize); flip: 0 none, 1 horizontal, 2 vertical.
 * d_seg (B,out_h,out_w) uint8 class indices (first maximum wins).  K <= 256. */
int ddp_seg_postprocess(const float* d_scores, int batch, int num_classes, int h, int w, int img_h, int img_w,
                        int crop_h, int crop_w, int out_h, int out_w, int align_corners, int flip,
                        unsigned char* d_seg, void* stream);

/* Multi-scale / flip test-time augmentation epilogue, fused (encoder_decoder.py:306-331 `aug_test` over :251-287 `inference`
 * over :229-248 `whole_inference` over segmentors/ddp.py:124-128): for every augmentation a
 *   resize of its low-resolution scores to its network input -> crop to img_shape -> resize to ori_shape -> softmax ->
 *   flip undone,
 * then the mean over the augmentations and argmax.  One thread per output pixel walks all augmentations: neither the per-
 * augmentation (B,K,H,W) tensors nor the (B,K,ori_h,ori_w) running sum of the reference are materialised.
 * d_seg (B,out_h,out_w) uint8; d_prob optional (B,K,out_h,out_w) mean probabilities (tests / callers that need them). */
#define DDP_MAX_AUGS 16
typedef struct ddp_seg_aug {
  const float* d_scores;   /* (B,K,h,w): ddp_sample's output for this augmentation */
  int32_t h, w;            /* its map size */
  int32_t img_h, img_w;    /* its (padded) network input */
  int32_t crop_h, crop_w;  /* its img_shape */
  int32_t flip;            /* 0 none, 1 horizontal, 2 vertical: undone on the probabilities */
} ddp_seg_aug;
int ddp_seg_aug_postprocess(const ddp_seg_aug* augs, int n_aug, int batch, int num_classes, int out_h, int out_w,
                            int align_corners, unsigned char* d_seg, float* d_prob, void* stream);

/* Sliding-window inference epilogue, fused (encoder_decoder.py:180-227 `slide_inference` over segmentors/ddp.py:114-129
 * `encode_decode`, then :266-296 `inference` / `simple_test`): the image is covered by a grid of n_rows x n_cols windows of
 * crop_h x crop_w pixels (top-left corners win_y1[i], win_x1[j]: the reference's y1 / x1 after clamping to the image); every
 * window went through the K-step loop on its own and left low-resolution scores (B,K,h,w).  Per output pixel this kernel does
 * what the reference does with full-size tensors: resize each covering window's scores to the window size, sum them in window
 * order (row-major), divide by the number of covering windows, crop to (keep_h, keep_w) = img_shape, resize to (out_h, out_w) =
 * ori_shape, [softmax,] undo the flip, argmax.  Neither the per-window (B,K,crop_h,crop_w) logits nor `preds` / `count_mat` at
 * image size exist.  A pixel may be covered by at most 4 window rows and 4 window columns (stride >= crop / 4).
 * d_scores[i * n_cols + j] (B,K,h,w); d_seg (B,out_h,out_w) uint8 or NULL; d_prob optional (B,K,out_h,out_w):
 * prob_mode 1 = softmax probabilities (`inference`), 2 = the averaged scores themselves (`slide_inference`'s return value). */
#define DDP_MAX_WINDOWS 64
int ddp_seg_slide_postprocess(const float* const* d_scores, const int* win_y1, const int* win_x1, int n_rows, int n_cols, int batch,
                              int num_classes, int h, int w, int crop_h, int crop_w, int img_h, int img_w, int keep_h, int keep_w,
                              int out_h, int out_w, int align_corners, int flip, int prob_mode, unsigned char* d_seg, float* d_prob,
                              void* stream);

/* Post-loop epilogue of the depth toolbox, fused (depth/depth/models/depther/ddp.py:95-109 `encode_decode`: clamp to
 * [min_depth, max_depth], bilinear resize to the network input; encoder_decoder.py:187-194 `inference`: flip undone;
 * :198-209 `simple_test`; :210-229 `aug_test`: running sum over the augmentations in list order, / n).  One thread per 4
 * output pixels walks all augmentations: neither the per-augmentation (B,1,H,W) maps nor the running sum are materialised.
 * With n_aug == 1 this is `simple_test` / `inference` / `encode_decode(rescale=True)` (x / 1 is exact); out == map size:
 * no resize (`rescale=False`).  d_out (B,1,out_h,out_w) fp32. */
typedef struct ddp_depth_aug {
  const float* d_depth;    /* (B,1,h,w): ddp_sample's output (task depth) for this augmentation */
  int32_t h, w;            /* its map size */
  int32_t flip;            /* 0 none, 1 horizontal, 2 vertical: undone on the resized map */
} ddp_depth_aug;
int ddp_depth_postprocess(const ddp_depth_aug* augs, int n_aug, int batch, int out_h, int out_w, int align_corners,
                          float min_depth, float max_depth, float* d_out, void* stream);

/* MultiStageMerging neck (necks/multi_stage_merging.py:40-52): the step that produces the frozen
 * feature x of the sampling loop from the four FPN levels: bilinear resize of every level to level 0's grid, concat
 * (1024 ch), down = ConvModule(1024, 256, 1, bias=False, GroupNorm(32), no activation).
 * d_levels[l] (B,256,h_l,w_l) NCHW; d_conv_w = neck.down.conv.weight (256,1024,1,1); d_gn_w/d_gn_b = neck.down.gn.*;
 * d_out (B,256,h_0,w_0) NCHW.  Workspace from ddp_neck_msm_workspace (caller-owned). */
int ddp_neck_msm_workspace(int batch, const int* level_h, const int* level_w, size_t* bytes);
/* flags of the two neck entries: the workspace starts with a weight region (split weight planes as the stage images of the
 * stream GEMM) whose layout depends on the channel counts only; a caller that runs the same weights again through the same
 * workspace buffer - on the same stream, or after synchronising with the call that packed them - passes
 * DDP_NECK_WEIGHTS_READY and the weights are not re-packed. */
#define DDP_NECK_WEIGHTS_READY 1
int ddp_neck_msm(const float* const* d_levels, const int* level_h, const int* level_w, int batch,
                 const float* d_conv_w, const float* d_gn_w, const float* d_gn_b, int align_corners, int flags, float* d_out,
                 void* d_workspace, void* stream);

/* FPN neck (necks/fpn.py:163-213) as the DDP configs build it: 4 levels, lateral ConvModule(C_l,256,1,
 * bias=False, GN(32), no act), top-down nearest upsample + add, output ConvModule(256,256,3,padding=1,bias=False,GN(32),
 * no act); num_outs = 4 (no extra levels).  d_in[l] (B,C_l,h_l,w_l) NCHW, C_l % 32 == 0, 64 <= C_l <= 4096; d_out[l]
 * (B,256,h_l,w_l) NCHW. */
typedef struct ddp_fpn_level {
  const float* lat_w;      /* lateral_convs.l.conv.weight (256,C_l,1,1) */
  const float* lat_gn_w;   /* lateral_convs.l.gn.weight / bias (256) */
  const float* lat_gn_b;
  const float* out_w;      /* fpn_convs.l.conv.weight (256,256,3,3) */
  const float* out_gn_w;   /* fpn_convs.l.gn.weight / bias (256) */
  const float* out_gn_b;
  int in_channels, h, w;
} ddp_fpn_level;
int ddp_neck_fpn_workspace(const ddp_fpn_level* levels, int batch, size_t* bytes);
int ddp_neck_fpn(const ddp_fpn_level* levels, int batch, const float* const* d_in, float* const* d_out, int flags,
                 void* d_workspace, void* stream);

/* FPN followed by MultiStageMerging - the neck list of every DDP config (configs/ade/ddp_swin_t_2x8_512x512_160k_ade20k.py: neck =
 * [FPN, MultiStageMerging]; the segmentor uses only the merged map): the four FPN outputs stay in the GEMM operand layout and
 * feed the merging directly; their NCHW form is never produced.  d_in[l] as ddp_neck_fpn, the three MSM parameters as
 * ddp_neck_msm, d_out (B,256,h_0,w_0) NCHW. */
int ddp_neck_fpn_msm_workspace(const ddp_fpn_level* levels, int batch, size_t* bytes);
int ddp_neck_fpn_msm(const ddp_fpn_level* levels, int batch, const float* const* d_in, const float* d_msm_conv_w,
                     const float* d_msm_gn_w, const float* d_msm_gn_b, int align_corners, int flags, float* d_out,
                     void* d_workspace, void* stream);

/* FCNHeadWithTime.forward (decode_heads/fcn_head_with_time.py:285-305), eval mode:
 *   x = inputs[0]; for each ConvWithTimeModule: x = ReLU( norm(conv3x3(x)) * (scale + 1) + shift ),
 *   (scale, shift) = Linear(SiLU(temb)).chunk(2)  (:205-225);  out = conv_seg(x)  (cls_seg, dropout is identity in eval).
 * 256 channels in and out; the norm is an eval-mode BatchNorm (running statistics) or absent.  The reference's
 * conv_cat is constructed but never called by _forward_feature (:285-299) and is therefore not part of the path. */
typedef struct ddp_fcn_conv {
  const float* conv_w;   /* convs.i.conv.weight (256,256,3,3) */
  const float* conv_b;   /* convs.i.conv.bias (256) or NULL (bias='auto' with a norm) */
  const float* bn_w;     /* convs.i.bn.weight / bias / running_mean / running_var (256 each), or all NULL */
  const float* bn_b;
  const float* bn_mean;
  const float* bn_var;
  float bn_eps;
  const float* time_w;   /* convs.i.time_mlp.1.weight (512,1024) */
  const float* time_b;   /* convs.i.time_mlp.1.bias (512) */
} ddp_fcn_conv;
int ddp_fcn_head_workspace(int maps, int h, int w, int num_classes, size_t* bytes);
int ddp_fcn_head_forward(const ddp_fcn_conv* convs, int num_convs, int dilation, const float* d_cls_w, const float* d_cls_b,
                         int num_classes, const float* d_feat /* (maps,256,h,w) */, const float* d_temb /* (1024) or NULL */,
                         int maps, int h, int w, float* d_out /* (maps,num_classes,h,w) */, void* d_workspace, void* stream);

/* The K-step sampler with FCNHeadWithTime as the decode head: what `DDP.ddim_sample` / `ddpm_sample`
 * (segmentors/ddp.py:215-290) compute when `_decode_head_forward_test` (:192-196) dispatches to
 * `FCNHeadWithTime.forward_test` (decode_heads/fcn_head_with_time.py:327-343).  Same inputs, outputs, schedule scalars and
 * x0 projection / update / accumulation as ddp_sample; cfg->task must be DDP_TASK_SEG, cfg->num_layers is ignored;
 * `weights` supplies transform, time_mlp, embedding and conv_seg (head_w / head_b), `convs` the head's ConvWithTimeModules. */
int ddp_sample_fcn_workspace(const ddp_cfg* cfg, int num_convs, int dilation, size_t* bytes);
/* Everything of that loop that depends on weights and schedule only - the time embeddings of the K steps, the x0 table, the
 * split column blocks of the concat-conv and, per (step, conv), the FiLM vector (fcn_head_with_time.py:216-221), the folded
 * norm x FiLM affine and the 72 stage images of the scaled 3x3 weights (the FiLM scale is folded INTO the weights, so every
 * step has its own images), plus conv_seg's images - written once into the model region of the workspace.  A caller that
 * samples repeatedly through the same workspace buffer (same stream, or after synchronising with this call) passes
 * DDP_FLAG_FCN_PREPARED in cfg->flags to ddp_sample_fcn and none of these ~12 kernels per (step, conv) runs again; without
 * the flag ddp_sample_fcn calls this itself. */
int ddp_prepare_fcn(const ddp_cfg* cfg, const ddp_weights* weights, const ddp_fcn_conv* convs, int num_convs, int dilation,
                    const ddp_step* steps, void* d_workspace, void* stream);
int ddp_sample_fcn(const ddp_cfg* cfg, const ddp_weights* weights, const ddp_fcn_conv* convs, int num_convs, int dilation,
                   const ddp_step* steps, const float* d_x, const float* d_noise, const float* d_step_noise, float* d_out,
                   void* d_workspace, void* stream);

/* Measurement hook (bench.py roofline leg; not part of the reference surface, NOT thread-safe: one process-wide session;
 * two threads sampling with a session armed would interleave their records): arm HIP-event timing
 * around every launch of one GEMM call site, then read the summed duration and launch count.
 * tag: 1 xproj, 2 feat / step prologue, 3 value_proj, 4 sampling proj, 5 output_proj+LN, 6 FFN fc1, 7 FFN fc2+LN / the
 * layer kernel, 8 head conv / seg tail, 9 deformable gather, 10 last layer of a step + seg tail (one kernel); 255 = all of
 * them at once.  ddp_profile_end synchronises on
 * the recorded events and returns the sum over all records; ddp_profile_read then gives one call site's share
 * (DDP_E_BADCFG for an unknown tag or when no finished session exists). */
int ddp_profile_begin(int tag);
int ddp_profile_end(float* total_ms, int* launches);
int ddp_profile_read(int tag, float* total_ms, int* launches);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* DDP_MI355X_H */
