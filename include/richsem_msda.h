/*
 * richsem_msda.h -- C ABI of the MI355X (gfx950) multi-scale deformable attention library
 * (librichsem_msda.so, built from richsem_amd/csrc by hipcc --offload-arch=gfx950).
 *
 * This is the drop-in boundary for the reference's native extension
 * `MultiScaleDeformableAttention` (reference models/richsem/ops/src/vision.cpp:13-16):
 *
 *   reference (pybind11 / ATen)                                   this library (extern "C")
 *   ------------------------------------------------------------  ---------------------------
 *   ms_deform_attn_forward   src/ms_deform_attn.h:20-39            msda_forward_{f32,f64}
 *     -> ms_deform_attn_cuda_forward  src/cuda/ms_deform_attn_cuda.cu:20-80
 *   ms_deform_attn_backward  src/ms_deform_attn.h:41-61            msda_backward_{f32,f64}
 *     -> ms_deform_attn_cuda_backward src/cuda/ms_deform_attn_cuda.cu:83-153
 *
 * Plain pointers and sizes only; no torch / ATen types.  All data pointers are DEVICE pointers
 * to contiguous row-major tensors with the reference's layouts
 * (src/cuda/ms_deform_attn_cuda.cu:40-48):
 *
 *   value          (N, S, M, D)            S = sum_l H_l*W_l
 *   spatial_shapes (L, 2) int64            (H_l, W_l)                        [device]
 *   level_start    (L)    int64            start row of level l inside S     [device]
 *   sampling_loc   (N, Lq, M, L, P, 2)     (x, y) normalised to [0,1] over the padded map
 *   attn_weight    (N, Lq, M, L, P)
 *   out / grad_out (N, Lq, M*D)
 *   grad_value, grad_sampling_loc, grad_attn_weight: shaped like value / sampling_loc / attn_weight
 *
 * Differences from the reference binding, all on the ownership side:
 *   - outputs are caller-allocated; they need NOT be zero-filled (the reference allocates them
 *     with at::zeros, ms_deform_attn_cuda.cu:54,121-123; here the library zero-fills what its
 *     kernels accumulate into, on `stream`);
 *   - the HIP stream is an explicit argument (the reference takes the current stream,
 *     ms_deform_attn_cuda.cu:65,135); work is enqueued, never synchronised;
 *   - `shapes_host` / `level_start_host` are optional HOST mirrors of the two int64 tensors.
 *     The launch geometry and the argument checks need the level sizes on the host; when the
 *     mirrors are NULL the library copies them from the device, which synchronises `stream`
 *     (correct, slower; a training loop should pass the mirrors, as the Python shim does);
 *   - errors are returned (and described by msda_last_error()), never only printed as in
 *     ms_deform_im2col_cuda.cuh:948-952.
 *
 * `im2col_step` keeps the reference's contract (N % min(N, im2col_step) == 0, otherwise
 * MSDA_ERR_IM2COL_STEP; ms_deform_attn_cuda.cu:50-52).  Images are independent, so the
 * library processes the whole batch in one launch whatever the step is; results are identical.
 *
 * Thread safety: entry points keep no global mutable state except the option table set by
 * msda_set_option(); they may be called concurrently on different streams (forward thread
 * and autograd thread).
 */
#ifndef RICHSEM_MSDA_H
#define RICHSEM_MSDA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RICHSEM_MSDA_ABI_VERSION 8

/* Return codes: 0 = success; negative = argument error detected on the host (nothing was
 * launched); positive = hipError_t reported by the runtime. */
enum {
    MSDA_OK = 0,
    MSDA_ERR_NULL_POINTER = -1,
    MSDA_ERR_BAD_DIMS = -2,      /* non-positive dimension, sum H*W != S, bad level_start   */
    MSDA_ERR_IM2COL_STEP = -3,   /* N % min(N, im2col_step) != 0                            */
    MSDA_ERR_TOO_LARGE = -4,     /* a tensor has >= 2^31 elements (32-bit index math, as the */
                                 /* reference: ms_deform_im2col_cuda.cuh:255-263)           */
    MSDA_ERR_MISALIGNED = -5,    /* a data pointer is not aligned to its element size        */
    MSDA_ERR_NO_DEVICE = -6,     /* no gfx950 device / code object not loadable              */
    MSDA_ERR_BAD_OPTION = -7,
    MSDA_ERR_NOT_ON_CPU = -8     /* the host ("_cpu") variants: declared, as in the reference, and not implemented */
};

typedef void *msda_stream_t; /* hipStream_t; NULL = the default stream */

int msda_abi_version(void);

/* Text of the last error on the calling thread ("" if none). */
const char *msda_last_error(void);

/* Tuning / test hooks.  Keys:
 *   "fwd_variant"     0 = auto, 1 = direct gather kernel, 2 = LDS-window kernel (when applicable), 3 = split kernel (32 lanes per
 *                     (query, head), all gathers of a lane in flight at once: D = 32, L*P a multiple of 4 up to 32; automatic for
 *                     fp32 calls of fewer than 65536 (query, head) items: decoder-shaped calls)
 *   "bwd_variant"     0 = auto, 1 = direct kernel (level-sum windows / row atomics), 4 = routed pixel-stationary kernels
 *                     (sampling points routed to output tiles in ONE pass -- every query block lays its records out in a stretch of
 *                     its own and announces them to the tiles' bins --, then one workgroup per tile: every pixel of grad_value
 *                     written once with plain stores; cost independent of where the points fall; fp32 / bf16 storage, D = 32,
 *                     L <= 4, Lq <= 320 x 256 per (image, head): up to 320 x 128 queries -- E: 22323, the 1280 x 1280 mosaic
 *                     batches: 34000 -- with 8-wave route workgroups, beyond that with 16-wave ones; otherwise the call falls
 *                     back to 1); 5 = row-band kernel (msda_band.h: grad_value, grad_sampling_loc and grad_attn_weight of a
 *                     decoder-shaped call in ONE launch; a measured option -- 101 us against 88 us for 1 on the decoder call,
 *                     profiles/r05_dd_backward.md -- D = 32 only, else falls back to 1).  auto = routed for encoder-shaped calls
 *                     (Lq == S), direct for the rest
 *   "band_lds_kb"     row-band backward: window budget in KB (16..150, default 64 = two workgroups per CU)
 *   "band_hits"       row-band backward: expected listed points per workgroup above which a band is dealt over query slabs (default 400)
 *   "fwd_prep_fused"  1 (default) = msda_forward_prep_* runs decoder-shaped calls as one kernel, 0 = always two, 2 = encoder-shaped
 *                     calls too: the LDS-window kernel reads the raw projection itself (a measured option: 232 us against 139 us
 *                     for the two-kernel form on the encoder call, profiles/r05_f1_ab.txt)
 *   "locality_monitor"  1 (default) = in auto mode the window forward kernel counts the points that miss their window on the
 *                     first 2 calls of a (problem shape, sampling_loc buffer) and on every 64th after; the count comes back
 *                     by an asynchronous copy and is read on a later call (no call waits, nothing is probed during graph
 *                     capture, up to 8 probes in flight).  Share > 8 % -> direct forward (crossover measured on MI355X,
 *                     profiles/r02_locality.md).  0 = auto always takes the window forward kernel when it applies.  Setting
 *                     it forgets what was learnt.  "locality_share_ppm" (get only): last measured share in parts per
 *                     million, -1 = none.  (The backward needs no monitor.)
 *   "rps_tile"        routed backward: largest tile side + 1 (4..16, default 16: tile + one row / column <= 256 pixels)
 *   "rps_max_chunks"  routed backward: chunks of 1536 points one workgroup takes before a tile's points are dealt over
 *                     several workgroups (default 12)
 *   "rps_route_wgs"   routed backward: workgroups per CU of the route pass (default 2: what is resident)
 *   "rps_seg_shift"   routed backward: a pixel's list is walked in units of at most 2^n sampling points (3..11, default 4)
 *   "levelsum_lds_kb" level-sum window size in KB (8..150, default 150 = one workgroup per CU)
 *   "bwd_direct_cpl"  channels per lane of the direct backward kernel (0 = auto, 1, 2, 4)
 *   "tile_region"     side of an LDS-window region, in pixels of the finest level (default 20)
 *   "tile_margin"     window margin around a region, in pixels of the sampled level (default 6)
 *   "tile_grow"       1 (default) = a window grows into the LDS its phase leaves unused, coarsest level first (per-level
 *                     margins >= tile_margin); 0 = every level uses exactly tile_margin
 *   "bwd_levelsum"    1 (default) = direct backward, fp32: grad_value of whole levels (or row bands of a level) is summed
 *                     in an f64 LDS window by its own kernel and written once.  Calls with Lq*P <= 2^20 whose levels fit 16 windows
 *                     hand over ALL levels: no global atomics, no zero-fill, sums exact to fp32 rounding whatever the order.  Otherwise only
 *                     levels that fit LDS whole and receive >= 2 sampling points per pixel.  0 = off (row atomics only)
 *   "bwd_split"       1 (default) = when the level-sum kernel has produced all of grad_value, small calls (as "fwd_variant" 3) take
 *                     grad_sampling_loc / grad_attn_weight from the split kernel (32 lanes per (query, head)); 0 = the 8-lane kernel
 *   "profile_filter"  which calls msda_profile_enable brackets with its event pair: 0 (default) = every call; (kind + 1) * 16 + variant =
 *                     only those (kind 0 forward / 1 backward, variant as in msda_profile_record) -- an event pair costs a call ~4 us
 *   "tile_persist"    persistent workgroups walking the work items (default 512 = 2 per CU; 0 = one workgroup per item)
 *   "tile_debug"      diagnostic bits (stage-stamp kernel selection)
 * Unknown key or value out of range -> MSDA_ERR_BAD_OPTION.  Options change speed, never results. */
int msda_set_option(const char *key, int value);
int msda_get_option(const char *key, int *value);

/* Host-only: the launch plan the LDS-window kernels would use for a problem (no device access).
 * info[0] = 1 if the window kernels apply (fp32, D = 32, L <= 4, Lq == S, levels contiguous), else 0;
 * info[1], info[2] = region grid (rows, cols); info[3] = LDS phases; info[4] = LDS bytes per workgroup;
 * info[5] = workgroups; info[6] = margin; info[7] = largest number of queries in one region. */
int msda_tiled_plan(int N, int S, int M, int D, int L, int Lq, int P, const int64_t *shapes_host,
                    const int64_t *level_start_host, int info[8]);

/* Host-only: what the level-sum backward kernel (fp32 direct backward, option "bwd_levelsum") would take for a problem.
 * info[0] = bit mask of the levels handed over (0 = none; all L bits = no global atomics and no zero-fill at all);
 * info[1] = (level, row band) windows; info[2] = 4-channel slices; info[3] = LDS bytes per workgroup;
 * info[4] = workgroups; info[5] = most rows in one window; info[6..7] = 0. */
int msda_levelsum_plan(int N, int S, int M, int D, int L, int Lq, int P, const int64_t *shapes_host,
                       const int64_t *level_start_host, int info[8]);

/* Diagnostic: when `device_buffer` is non-NULL the LDS-window kernels write shader-clock stamps into it, 16 x 8 bytes
 * per workgroup (buffer >= workgroups x 128 bytes), one per kernel stage; NULL (default) switches it off. */
int msda_debug_stamps(void *device_buffer);

/* Diagnostic: when `device_counter` (one uint32, caller-zeroed) is non-NULL, the LDS-window FORWARD kernel adds to it
 * the number of sampling points that missed their window and took the general path, counted once per 16-channel half
 * (so a call adds at most 2 * N*Lq*M*L*P).  NULL (default) switches it off. */
int msda_debug_stats(void *device_counter);

/* ---- launch profiler (measurement aid; off by default) ------------------------------------
 * When enabled, every forward/backward call brackets its MAIN kernel (not the zero-fill) with
 * a pair of pre-created HIP events recorded on the stream the kernel is launched on.
 * msda_profile_collect() synchronises those events and returns one record per call, oldest
 * first, then clears the log.  No allocation happens on the launch path. */
typedef struct {
    int kind;        /* 0 = forward, 1 = backward                                  */
    int variant;     /* 1 = direct kernel, 2 = tiled kernel                        */
    int dtype_bytes; /* 4 or 8                                                     */
    int N, S, M, D, L, Lq, P;
    float kernel_ms; /* elapsed time of the main kernel                            */
} msda_profile_record;

int msda_profile_enable(int capacity);  /* capacity = max calls to log; 0 disables and frees   */
int msda_profile_collect(msda_profile_record *records, int max_records, int *n_records);

/* ---- forward:  replaces ms_deform_attn_forward (src/ms_deform_attn.h:20-39) ---------------- */
int msda_forward_f32(const float *value, const int64_t *spatial_shapes, const int64_t *level_start,
                     const float *sampling_loc, const float *attn_weight,
                     int N, int S, int M, int D, int L, int Lq, int P, int im2col_step,
                     float *out,
                     const int64_t *shapes_host, const int64_t *level_start_host,
                     msda_stream_t stream);

int msda_forward_f64(const double *value, const int64_t *spatial_shapes, const int64_t *level_start,
                     const double *sampling_loc, const double *attn_weight,
                     int N, int S, int M, int D, int L, int Lq, int P, int im2col_step,
                     double *out,
                     const int64_t *shapes_host, const int64_t *level_start_host,
                     msda_stream_t stream);

/* ---- backward: replaces ms_deform_attn_backward (src/ms_deform_attn.h:41-61) --------------- */
int msda_backward_f32(const float *value, const int64_t *spatial_shapes, const int64_t *level_start,
                      const float *sampling_loc, const float *attn_weight, const float *grad_out,
                      int N, int S, int M, int D, int L, int Lq, int P, int im2col_step,
                      float *grad_value, float *grad_sampling_loc, float *grad_attn_weight,
                      const int64_t *shapes_host, const int64_t *level_start_host,
                      msda_stream_t stream);

int msda_backward_f64(const double *value, const int64_t *spatial_shapes, const int64_t *level_start,
                      const double *sampling_loc, const double *attn_weight, const double *grad_out,
                      int N, int S, int M, int D, int L, int Lq, int P, int im2col_step,
                      double *grad_value, double *grad_sampling_loc, double *grad_attn_weight,
                      const int64_t *shapes_host, const int64_t *level_start_host,
                      msda_stream_t stream);

/* ---- host ("_cpu") variants of both (ABI v8).  The reference declares ms_deform_attn_cpu_forward / _backward
 * (src/cpu/ms_deform_attn_cpu.h:14-31) and implements neither: both bodies are AT_ERROR("Not implement on cpu")
 * (src/cpu/ms_deform_attn_cpu.cpp:17-41), and the dispatcher never reaches them (src/ms_deform_attn.h:38,60 raise
 * "Not implemented on the CPU" for a tensor that is not on the device).  These two do the same: they read no
 * argument, launch nothing, set msda_last_error() to the reference's text and return MSDA_ERR_NOT_ON_CPU.  This
 * library has no CPU implementation of the operator (the restatement under oracle/ is test infrastructure). */
int msda_forward_cpu(const void *value, const int64_t *spatial_shapes, const int64_t *level_start,
                     const void *sampling_loc, const void *attn_weight,
                     int N, int S, int M, int D, int L, int Lq, int P, int im2col_step, void *out);
int msda_backward_cpu(const void *value, const int64_t *spatial_shapes, const int64_t *level_start,
                      const void *sampling_loc, const void *attn_weight, const void *grad_out,
                      int N, int S, int M, int D, int L, int Lq, int P, int im2col_step,
                      void *grad_value, void *grad_sampling_loc, void *grad_attn_weight);

/* ---- bf16 storage, fp32 compute (new capability: the reference dispatches float / double only,
 * src/cuda/ms_deform_attn_cuda.cu:64,134) -------------------------------------------------------
 * value, out, grad_out, grad_value are bfloat16 (raw 16-bit words, same layouts as above);
 * sampling_loc, attn_weight and their gradients stay float32.  Every sum is formed in fp32 (or in
 * f64 LDS windows) and rounded to bf16 ONCE: grad_value is never accumulated in bf16.  Where the
 * kernels accumulate grad_value with atomics, they do so in an fp32 scratch buffer owned by the
 * library (one per device and stream, allocated on first use -- not while the stream is being
 * captured -- and reused); decoder-shaped calls need no scratch.
 * value / out / grad_out / grad_value must be 8-byte aligned for the fast paths (2 bytes minimum). */
int msda_forward_bf16(const uint16_t *value, const int64_t *spatial_shapes, const int64_t *level_start,
                      const float *sampling_loc, const float *attn_weight,
                      int N, int S, int M, int D, int L, int Lq, int P, int im2col_step,
                      uint16_t *out,
                      const int64_t *shapes_host, const int64_t *level_start_host,
                      msda_stream_t stream);

int msda_backward_bf16(const uint16_t *value, const int64_t *spatial_shapes, const int64_t *level_start,
                       const float *sampling_loc, const float *attn_weight, const uint16_t *grad_out,
                       int N, int S, int M, int D, int L, int Lq, int P, int im2col_step,
                       uint16_t *grad_value, float *grad_sampling_loc, float *grad_attn_weight,
                       const int64_t *shapes_host, const int64_t *level_start_host,
                       msda_stream_t stream);

/* ---- module-level element-wise work ------------------------------------------------
 * What reference models/richsem/ops/modules/ms_deform_attn.py:94-109 does with half a dozen PyTorch ops per call --
 * masked_fill of value, softmax over the L*P logits of a (query, head), offsets / normaliser (+ reference points) --
 * and what autograd replays backwards, as one kernel each way plus an in-place row mask.
 *
 *   offsets    raw output of the sampling_offsets projection: row (n, q) at offsets + (n*Lq + q)*off_stride, M*L*P*2 values
 *   logits     raw output of the attention_weights projection: row (n, q) at logits + (n*Lq + q)*log_stride, M*L*P values
 *              (strides in ELEMENTS: both may point into the output of ONE 256 -> 384 projection, stride 384)
 *   ref        reference points (N, Lq, L, ref_dim), ref_dim = 2 (x, y) or 4 (x, y, w, h), contiguous
 *   shapes_host  HOST copy of spatial_shapes (L, 2) int64 -- the offset normaliser (W_l, H_l) of ref_dim = 2
 *   loc, aw    outputs, contiguous: sampling_loc (N, Lq, M, L, P, 2), attn_weight (N, Lq, M, L, P) -- the operator's inputs
 * msda_prep_backward: grad_loc / grad_aw as returned by msda_backward_*, aw as produced by msda_prep_forward; writes
 *   grad_offsets / grad_logits (with their own row strides: they may be the two parts of ONE gradient tensor) and, when
 *   grad_ref is non-NULL, grad_reference_points (N, Lq, L, ref_dim); `offsets` is only read for ref_dim = 4 with grad_ref.
 * msda_mask_rows: x (rows, row_elems), zeroes IN PLACE every row whose mask byte is non-zero (value rows of padded
 *   pixels forward, grad_value rows backward).  Only masked rows cost memory traffic.
 * L <= 16, L*P <= 64.  float32 and float64 (the module's golden vectors are float64). */
int msda_prep_forward_f32(const float *offsets, int64_t off_stride, const float *logits, int64_t log_stride,
                          const float *ref, int ref_dim, const int64_t *shapes_host,
                          int N, int Lq, int M, int L, int P, float *loc, float *aw, msda_stream_t stream);
int msda_prep_forward_f64(const double *offsets, int64_t off_stride, const double *logits, int64_t log_stride,
                          const double *ref, int ref_dim, const int64_t *shapes_host,
                          int N, int Lq, int M, int L, int P, double *loc, double *aw, msda_stream_t stream);
int msda_prep_backward_f32(const float *grad_loc, const float *grad_aw, const float *aw,
                           const float *offsets, int64_t off_stride, const float *ref, int ref_dim,
                           const int64_t *shapes_host, int N, int Lq, int M, int L, int P,
                           float *grad_offsets, int64_t goff_stride, float *grad_logits, int64_t glog_stride,
                           float *grad_ref, msda_stream_t stream);
int msda_prep_backward_f64(const double *grad_loc, const double *grad_aw, const double *aw,
                           const double *offsets, int64_t off_stride, const double *ref, int ref_dim,
                           const int64_t *shapes_host, int N, int Lq, int M, int L, int P,
                           double *grad_offsets, int64_t goff_stride, double *grad_logits, int64_t glog_stride,
                           double *grad_ref, msda_stream_t stream);
/* the same with a bfloat16 projection: offsets / logits (and their gradients) are raw 16-bit bf16 words -- the output of a bf16
 * GEMM -- while locations, weights, reference points and their gradients are float32 and all arithmetic is fp32 */
int msda_prep_forward_bf16(const uint16_t *offsets, int64_t off_stride, const uint16_t *logits, int64_t log_stride,
                           const float *ref, int ref_dim, const int64_t *shapes_host, int N, int Lq, int M, int L, int P,
                           float *loc, float *aw, msda_stream_t stream);
int msda_prep_backward_bf16(const float *grad_loc, const float *grad_aw, const float *aw,
                            const uint16_t *offsets, int64_t off_stride, const float *ref, int ref_dim,
                            const int64_t *shapes_host, int N, int Lq, int M, int L, int P,
                            uint16_t *grad_offsets, int64_t goff_stride, uint16_t *grad_logits, int64_t glog_stride,
                            float *grad_ref, msda_stream_t stream);
/* msda_forward_prep_*: what msda_prep_forward_* followed by msda_forward_* compute, behind ONE entry point (SURVEY.md section 8f rank 1:
 * "softmax + location math fused into the gather kernel"; reference ops/modules/ms_deform_attn.py:97-114).  Arguments as those two;
 * sampling_loc / attn_weight are OUTPUTS (the module's backward needs them) and must not be NULL.  Decoder-shaped calls (Lq != S,
 * L*P <= 32) run as one kernel that resolves the sampling points from the raw projection and never re-reads sampling_loc; other calls
 * run the two kernels one after the other (the encoder-shaped forward keeps its LDS-window kernel and locality monitor).
 * msda_set_option("fwd_prep_fused", 0) forces the two-kernel form everywhere. */
int msda_forward_prep_f32(const float *value, const int64_t *spatial_shapes, const int64_t *level_start,
                          const float *offsets, int64_t off_stride, const float *logits, int64_t log_stride,
                          const float *ref, int ref_dim, int N, int S, int M, int D, int L, int Lq, int P, int im2col_step,
                          float *out, float *sampling_loc, float *attn_weight,
                          const int64_t *shapes_host, const int64_t *level_start_host, msda_stream_t stream);
int msda_forward_prep_f64(const double *value, const int64_t *spatial_shapes, const int64_t *level_start,
                          const double *offsets, int64_t off_stride, const double *logits, int64_t log_stride,
                          const double *ref, int ref_dim, int N, int S, int M, int D, int L, int Lq, int P, int im2col_step,
                          double *out, double *sampling_loc, double *attn_weight,
                          const int64_t *shapes_host, const int64_t *level_start_host, msda_stream_t stream);
int msda_forward_prep_bf16(const uint16_t *value, const int64_t *spatial_shapes, const int64_t *level_start,
                           const uint16_t *offsets, int64_t off_stride, const uint16_t *logits, int64_t log_stride,
                           const float *ref, int ref_dim, int N, int S, int M, int D, int L, int Lq, int P, int im2col_step,
                           uint16_t *out, float *sampling_loc, float *attn_weight,
                           const int64_t *shapes_host, const int64_t *level_start_host, msda_stream_t stream);
int msda_mask_rows_f32(float *x, const uint8_t *mask, int64_t rows, int row_elems, msda_stream_t stream);
int msda_mask_rows_f64(double *x, const uint8_t *mask, int64_t rows, int row_elems, msda_stream_t stream);
int msda_mask_rows_bf16(uint16_t *x, const uint8_t *mask, int64_t rows, int row_elems, msda_stream_t stream);

/* ---- criterion plumbing: the all-negative term of the sigmoid focal loss, one pass each way -----------------
 * sigmoid_focal_loss at a negative entry is (1 - alpha) p^2 softplus(x) (reference models/richsem/richsem.py:1124-1160 through
 * sigmoid_focal_loss); the criterion sums it over whole logit tensors with one weight per row (query), and corrects the positive entries
 * separately.  logits (rows, C) float32 contiguous; row_weight (rows) float32 (0 = the row carries no loss).
 * msda_focal_neg_sum_f32 writes *n_partial <= max_partial fp64 partial sums (add them up); msda_focal_neg_grad_f32 writes
 * grad_logits = gscale[0] * d(sum)/d(logits) for every element (gscale: a device scalar, the upstream gradient). */
int msda_focal_neg_sum_f32(const float *logits, const float *row_weight, int64_t rows, int C, float alpha, double *partial, int max_partial,
                           int *n_partial, msda_stream_t stream);
int msda_focal_neg_grad_f32(const float *logits, const float *row_weight, int64_t rows, int C, float alpha, const float *gscale,
                            float *grad_logits, msda_stream_t stream);

/* The criterion's per-pair tails, one launch each (K pairs, float32): loss[0] <- the weighted sum, grad <- its gradient w.r.t. the
 * predictions (multiply by the incoming scalar gradient).  msda_box_pair_loss_f32: sum_k w[k] (c_l1 |p_k - t_k|_1 + c_giou (1 - GIoU(p_k,
 * t_k))) for boxes (cx, cy, w, h) (SetCriterion.loss_boxes, models/richsem/richsem.py:1162-1188; util/box_ops.py:9-64 on the diagonal;
 * gradients with torch's conventions for |x|, max / min and clamp); msda_focal_pos_sum_f32: sum_k w[k] (alpha (1 - q)^2 softplus(-x_k) -
 * (1 - alpha) q^2 softplus(x_k)), q = sigmoid(x_k) -- a positive entry's share of the sigmoid focal loss (richsem.py:1124-1160) minus the
 * all-negative term that msda_focal_neg_sum_f32 counted for it. */
int msda_box_pair_loss_f32(const float *pred, const float *target, const float *weight, int K, float c_l1, float c_giou, float *loss, float *grad_pred,
                           msda_stream_t stream);
int msda_focal_pos_sum_f32(const float *x, const float *weight, int K, float alpha, float *loss, float *grad_x, msda_stream_t stream);

/* ---- integer part of the contrastive-denoising set-up (reference
 * models/richsem/dn_components.py:42-71, 131-179): bit-exact int64 / bool results ---------------------------------
 * msda_dn_indices_i64: cum = exclusive prefix of the per-image box counts (batch + 1 int64 on the device), total = cum[batch],
 *   groups2 = 2 * dn_number.  Writes total * groups2 entries: known_bid[i] = image of box i % total,
 *   map_known_indice[i] = index of that box inside its image + single_pad * (i / total).
 * msda_dn_attn_mask_u8: (tgt_size x tgt_size) bytes, 1 = masked: columns < pad_size are hidden from rows >= pad_size, and
 *   from rows of another denoising group (group = index / group_pad). */
int msda_dn_indices_i64(const int64_t *cum, int batch, int64_t total, int groups2, int64_t single_pad, int64_t *known_bid,
                        int64_t *map_known_indice, msda_stream_t stream);
int msda_dn_attn_mask_u8(uint8_t *mask, int64_t tgt_size, int64_t pad_size, int64_t group_pad, msda_stream_t stream);

/* Top-k query selection (reference models/richsem/deformable_transformer.py:370-372: torch.topk(scores, k, dim=1)[1]): for every
 * row of `scores` (rows x n, float32) the indices -- and, if `values` is non-NULL, the scores -- of its k largest elements in
 * descending order; equal scores lowest index first.  n <= 36864, k <= 1024 (one workgroup holds a row in LDS). */
int msda_topk_f32(const float *scores, int rows, int n, int k, int64_t *indices, float *values, msda_stream_t stream);

/* The decoder's positional query embedding, gen_sineembed_for_position (models/richsem/utils.py:142-168) in one launch (the reference:
 * ~15 element-wise ops per decoder layer): boxes (tokens, >= dims) f32 (x, y[, w, h]) with a row stride of ld floats; out (tokens,
 * dims * pe_dim) bf16 in the reference's order (y, x[, w, h]), channel 2k = sin, 2k + 1 = cos of coordinate * 2 pi /
 * temperature^(2k / pe_dim).  dims = 2 | 4, pe_dim even.  No gradient (the boxes it is applied to are detached, :779-804). */
/* Backward of a 256 -> n linear layer with n <= 8 (the last layer of the box heads, MLP(256, 256, 4, 3): models/richsem/utils.py:110-122,
 * deformable_transformer.py:779-804): dy (T, n) bf16 contiguous, x (T, 256) bf16, w (n, 256) f32 -> dx (T, 256) bf16 (or NULL), dw (n, 256)
 * f32, db (n) f32 (or NULL): two streaming kernels instead of three GEMMs with a 4-wide operand. */
int msda_narrow_linear_backward_bf16(const uint16_t *dy, const uint16_t *x, const float *w, int T, int n, uint16_t *dx, float *dw, float *db,
                                     msda_stream_t stream);

/* The decoder's box update, y = sigmoid(delta + inverse_sigmoid(ref)) (models/richsem/deformable_transformer.py:779-804,
 * richsem.py:705-715; inverse_sigmoid of util/misc.py:605-609 with its eps) as one launch, and its gradient w.r.t. delta
 * (grad_y * y * (1 - y)) as one more: delta / grad_delta (n) bf16 or f32, ref, y, grad_y (n) f32.  ref is taken as a constant (the
 * reference detaches it between layers). */
int msda_box_refine_forward(const void *delta, int delta_is_bf16, const float *ref, float eps, int64_t n, float *y, msda_stream_t stream);
int msda_box_refine_backward(const float *grad_y, const float *y, int64_t n, void *grad_delta, int delta_is_bf16, msda_stream_t stream);
/* ... with the gradient w.r.t. ref as well (ref, grad_ref (n) f32; the heads' boxes of decoder layers 1..5, whose reference is not detached:
 * models/richsem/richsem.py:705-715): grad_ref = grad_delta * d inverse_sigmoid(ref) / d ref, the clamps' gradients as torch takes them */
int msda_box_refine_backward_ref(const float *grad_y, const float *y, int64_t n, void *grad_delta, int delta_is_bf16, const float *ref, float eps,
                                 float *grad_ref, msda_stream_t stream);

int msda_sine_embed_bf16(const float *boxes, int ld, int tokens, int dims, int pe_dim, float temperature, uint16_t *out, msda_stream_t stream);

/* ROIAlign forward (reference models/richsem/richsem.py:750, :878:
 * detectron2.layers.ROIAlign(output_size, spatial_scale, sampling_ratio = 0, aligned = True) on the frozen CLIP feature map).
 * input (N, C, H, W) contiguous; rois (K, 5) = (batch index, x1, y1, x2, y2) in input pixels; output (K, C, pooled_h, pooled_w).
 * detectron2's published algorithm (= torchvision.ops.roi_align); no gradient (the teacher is frozen). */
int msda_roi_align_forward_f32(const float *input, const float *rois, int K, int N, int C, int H, int W, int pooled_h,
                               int pooled_w, double spatial_scale, int sampling_ratio, int aligned, float *output,
                               msda_stream_t stream);
int msda_roi_align_forward_f64(const double *input, const double *rois, int K, int N, int C, int H, int W, int pooled_h,
                               int pooled_w, double spatial_scale, int sampling_ratio, int aligned, double *output,
                               msda_stream_t stream);

/* ---- convolution forward on the matrix cores with the frozen-BatchNorm affine, residual add and ReLU in its epilogue (SURVEY.md
 * section 8a rows a10 / a11; reference clip/model.py:10-56, :94-167 -- the frozen CLIP teacher called at models/richsem/richsem.py:628 --
 * and models/richsem/backbone.py:20-56 around the ResNet-50 convolutions; csrc/conv_mfma.hip) -------------------------------------
 *     out[n, ho, wo, co] = act( scale[co] * sum_{kh, kw, ci} x[n, ho s + kh - p, wo s + kw - p, ci] w[co, ci, kh, kw] + shift[co]
 *                               (+ residual[n, ho, wo, co]) ),   act = relu or identity
 * Activations NHWC bf16, fp32 accumulation, fp32 scale / shift (BatchNorm folded: scale = w rsqrt(var + eps), shift = b - mean scale;
 * a plain bias is scale = 1, shift = bias).  C_out % 16 == 0; C_in % 32 == 0, or any C_in with KH KW C_in <= 512 (the 3-channel stems:
 * operand fragments gathered element by element); all pointers 16-byte aligned (x: 2-byte for few-channel inputs); residual may be NULL.
 * msda_conv_pack_weight: weight (C_out, C_in, KH, KW) fp32 (torch layout) -> msda_conv_packed_elems uint16 in MFMA fragment order
 * (repack when the weight changes).  msda_conv_set_tiling forces the per-wave tile (channel tiles in {1, 2, 4, 8, 16}, pixel tiles in
 * {1, 2, 3}; 0 = chosen per call so that the grid fills the chip) -- results do not depend on it beyond summation order. */
int msda_conv_set_tiling(int co_tiles, int pixel_tiles);
/* ... and msda_conv_set_ring the operand prefetch of the C_in % 64 == 0 kernels: both operands through LDS rings of `slots` iterations
 * (LDS DMA; csrc/conv_mfma.hip, conv_ring_kernel): -1 never, 0 chosen per call, 3 / 4 / 6 wherever that kernel applies.  Same results. */
int msda_conv_set_ring(int slots);
int msda_conv_packed_elems(int Cout, int Cin, int KH, int KW, int64_t *elems);
int msda_conv_pack_weight(const float *weight, int Cout, int Cin, int KH, int KW, uint16_t *packed, msda_stream_t stream);
int msda_conv_forward_bf16(const uint16_t *x, const uint16_t *packed_weight, const float *scale, const float *shift,
                           const uint16_t *residual, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int relu,
                           uint16_t *out, msda_stream_t stream);

/* The same with a k split for problems that leave most of the chip idle (few output pixels, long k: ResNet-50's layer4 3 x 3 at batch 2,
 * the stride-2 projection of C5): msda_conv_forward_workspace_bytes reports how many bytes of ZEROED device memory (an fp32 image of the
 * output) the split needs -- 0 where the problem is not split --, msda_conv_forward_ws_bf16 takes them (or NULL: no split): the k
 * slices add their raw sums there and a second kernel applies the epilogue.  Results differ from the unsplit call by fp32 summation
 * order only. */
int msda_conv_forward_workspace_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int64_t *bytes);
int msda_conv_forward_ws_bf16(const uint16_t *x, const uint16_t *packed_weight, const float *scale, const float *shift,
                              const uint16_t *residual, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int relu,
                              uint16_t *out, void *workspace, msda_stream_t stream);

/* GroupNorm with 8 channels per group on NHWC bf16 (nn.GroupNorm(32, 256) of the input projections, models/richsem/richsem.py:301,
 * :307): x (N, HW, C) bf16 with C = 8 * groups; gamma, beta (C) f32; stats: N * (C / 8) * 2 doubles of device scratch; out_f32 and / or
 * out_bf16 (N, HW, C), either may be NULL.  Sums of x and x^2 in fp32 per thread, combined in fp64.  Forward only. */
int msda_groupnorm8_nhwc_bf16(const uint16_t *x, const float *gamma, const float *beta, float eps, int N, int HW, int C, double *stats,
                              float *out_f32, uint16_t *out_bf16, msda_stream_t stream);
/* Its backward (the trainable input projections): x, dy (N, HW, C) bf16; stats as the forward left them; bstats: N * (C / 8) * 16 doubles
 * of device scratch; dx (N, HW, C) bf16; dgamma, dbeta (C) f32, either may be NULL.  Replaces autograd's GroupNorm backward on the
 * channels-first fp32 view (three permuting copies and five kernels per level). */
int msda_groupnorm8_backward_nhwc_bf16(const uint16_t *x, const uint16_t *dy, const float *gamma, float eps, int N, int HW, int C,
                                       const double *stats, double *bstats, uint16_t *dx, float *dgamma, float *dbeta, msda_stream_t stream);

/* Pooling on NHWC bf16 activations (nn.AvgPool2d(k) of the CLIP ResNet, clip/model.py:24, :36, :115; MaxPool2d(3, 2, 1) after
 * torchvision's ResNet stem): is_max = 0: mean over k x k windows at `stride`, pad must be 0; is_max = 1: maximum with implicit -inf
 * padding.  out (N, Ho, Wo, C) with Ho = (H + 2 pad - k) / stride + 1.  C % 8 == 0, 16-byte aligned pointers.  Forward only. */
int msda_pool_nhwc_bf16(const uint16_t *x, int N, int H, int W, int C, int k, int stride, int pad, int is_max, uint16_t *out,
                        msda_stream_t stream);

/* Gradient of msda_conv_forward_bf16 w.r.t. its input (for the backbone's trained stages), by the same kernel: a stride-1
 * convolution of the output gradient -- read as if zero-upsampled by `stride` -- with the flipped, transposed weight.
 * dy (N, Ho, Wo, Cout) bf16 (already multiplied by the ReLU mask); packed_weight_t = msda_conv_pack_weight(w_t, Cin, Cout, KH, KW) with
 * w_t[ci][co][kh][kw] = scale[co] * w[co][ci][KH-1-kh][KW-1-kw]; dx (N, H, W, Cin) bf16, every element written.  Square kernels,
 * Cout % 32 == 0, Cin % 16 == 0, pad <= KH - 1. */
int msda_conv_dgrad_bf16(const uint16_t *dy, const uint16_t *packed_weight_t, int N, int Ho, int Wo, int Cout, int Cin, int KH, int KW,
                         int stride, int pad, int H, int W, uint16_t *dx, msda_stream_t stream);

/* ... and its k-split form (see msda_conv_forward_ws_bf16; k = KH KW C_out here, the image is dx's) */
int msda_conv_dgrad_workspace_bytes(int N, int Ho, int Wo, int Cout, int Cin, int KH, int KW, int stride, int pad, int H, int W, int64_t *bytes);
int msda_conv_dgrad_ws_bf16(const uint16_t *dy, const uint16_t *packed_weight_t, int N, int Ho, int Wo, int Cout, int Cin, int KH, int KW,
                            int stride, int pad, int H, int W, uint16_t *dx, void *workspace, msda_stream_t stream);
/* ... with the element-wise work that follows an input gradient in a residual network in its epilogue: dx = mask(conv_dgrad(dy) + add).
 * add (N, H, W, Cin) bf16 or NULL: a second gradient of the same tensor (the identity branch of a bottleneck); relu_out (N, H, W, Cin)
 * bf16 or NULL: the tensor whose gradient this is, when it is the output of a ReLU -- dx is zeroed where it is not positive, i.e. dx is
 * the gradient at that ReLU's INPUT (what aten::threshold_backward would make of it in a pass of its own). */
int msda_conv_dgrad_fused_bf16(const uint16_t *dy, const uint16_t *packed_weight_t, int N, int Ho, int Wo, int Cout, int Cin, int KH, int KW,
                               int stride, int pad, int H, int W, const uint16_t *add, const uint16_t *relu_out, uint16_t *dx,
                               void *workspace, msda_stream_t stream);

/* Gradient of msda_conv_forward_bf16 w.r.t. its weight (csrc/conv_wgrad.hip): dw[co][kh][kw][ci] = sum over output pixels of
 * dz[n, ho, wo, co] * x[n, ho stride + kh - pad, wo stride + kw - pad, ci].  dz (N, Ho, Wo, Cout) bf16 = gradient at the CONVOLUTION's
 * output (ReLU mask applied); x (N, H, W, Cin) bf16; dw (Cout, KH, KW, Cin) fp32, every element written.  The pixels are split into
 * chunks where the (tap, channel block) grid alone cannot fill the chip; the chunks' partial sums go through `workspace`
 * (msda_conv_wgrad_workspace_bytes; may be NULL when that is 0) and a second kernel.  dbias (Cout) fp32, optional (NULL: not wanted):
 * sum over the output pixels of dz, formed on the way.  scale (Cout) fp32, optional: dw[co] is multiplied by scale[co] (the frozen affine
 * that follows the convolution commutes with the pixel sum).  torch_layout != 0: dw is written as (Cout, Cin, KH, KW), nn.Conv2d's layout.
 * Cout % 128 == 0, Cin % 128 == 0. */
int msda_conv_wgrad_workspace_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int64_t *bytes);
/* Tuning / tests: 1 (default) = operand stages prefetched three ahead through an LDS ring (LDS DMA), 0 = register-staged; same results */
int msda_conv_set_wgrad_ring(int on);
int msda_conv_wgrad_bf16(const uint16_t *dz, const uint16_t *x, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                         int pad, float *dw, float *dbias, const float *scale, int torch_layout, void *workspace, msda_stream_t stream);

/* Several weight gradients in ONE launch of the product kernel and one of the reduction (a bottleneck block's three or four,
 * models/richsem/backbone.py:59-92 trains layer2-4): each problem is msda_conv_wgrad_bf16's, its result in nn.Conv2d's layout
 * (Cout, Cin, KH, KW), multiplied by scale[co] when scale is not NULL; dbias as msda_conv_wgrad_bf16's (optional).  The problems share the chip in proportion to
 * their work instead of each being cut into ~512 short workgroups.  n <= 8. */
typedef struct {
    const uint16_t *dz, *x;
    float *dw;
    const float *scale;
    float *dbias;      /* (Cout) fp32: sum over the pixels of dz, or NULL */
    int N, H, W, Cin, Cout, KH, KW, stride, pad;
} msda_wgrad_problem;
int msda_conv_wgrad_group_workspace_bytes(const msda_wgrad_problem *problems, int n, int64_t *bytes);
int msda_conv_wgrad_group_bf16(const msda_wgrad_problem *problems, int n, void *workspace, msda_stream_t stream);

/* ---- two-stage query selection: row maxima of the class logits without the logits (reference
 * models/richsem/deformable_transformer.py:368-372 with the CLIP-text classifier models/richsem/richsem.py:176-184 in its shipped
 * configuration: bias-free linear projection Wp (proj x 256), text embeddings t_c) ------------------------------------------------
 *     score[token] = max_c  exp(logit_scale) * (Wp x / |Wp x|) . (t_c / |t_c|)  =  scale * max_c (G x)_c / sqrt(x . (A x))
 * with G = T^ Wp (classes x 256) and A = Wp^T Wp (256 x 256), both formed by the caller in fp32 whenever the weights change
 * (csrc/cls_mfma.hip).  msda_cls_pack lays [G; A] out in MFMA fragment order as bf16 hi + lo parts (msda_cls_packed_elems uint16
 * elements); msda_cls_max_scores computes the scores of `tokens` rows of x (fp32, or bf16 when x_is_bf16) with bf16 MFMAs on the
 * split operands: parts = 2 keeps the weights' lo part (fp32-level accuracy with fp32 x), parts = 1 is a plain bf16 product.
 * d_model must be 256; classes <= 8192; x and packed 16-byte aligned.  Forward only (the selection carries no gradient). */
int msda_cls_packed_elems(int classes, int64_t *elems);
int msda_cls_pack(const float *G, int classes, const float *A, int d_model, uint16_t *packed, msda_stream_t stream);
int msda_cls_max_scores(const void *x, int x_is_bf16, const uint16_t *packed, int tokens, int d_model, int classes, float scale,
                        int parts, float *scores, msda_stream_t stream);

/* ---- attention core of CLIP's AttentionPool2d for its single query token (reference
 * clip/model.py:58-91, called at models/richsem/richsem.py:753 on the ROIAlign output) -----------------------------------
 * With one query per head the key / value projections move to the other side of the attention (csrc/msda_attnpool.h): the caller
 * forms u[k, h, :] = head_dim^-1/2 * Wk_h^T q[k, h, :] with a library GEMM, this kernel computes per (ROI k, head h)
 *     x_0 = mean_t feat[k, :, t] + pos[0],  x_{t+1} = feat[k, :, t] + pos[t + 1]          (tokens, never materialised)
 *     a = softmax_t(u[k, h] . x_t),   z[k, h, :] = sum_t a_t x_t
 * and the caller finishes with Wv_h z[k, h] + bv_h and the output projection.
 * u, z (K, H, C), or (H, K, C) when head_major != 0; feat (K, C, T) as msda_roi_align_forward_* writes it; pos (T + 1, C);
 * spos = u . pos^T, rows as u, T + 1 columns (one library GEMM of the caller's); T <= 256. */
int msda_attnpool_core_f32(const float *u, const float *feat, const float *pos, const float *spos, int K, int H, int C, int T,
                           int head_major, float *z, msda_stream_t stream);
int msda_attnpool_core_f64(const double *u, const double *feat, const double *pos, const double *spos, int K, int H, int C, int T,
                           int head_major, double *z, msda_stream_t stream);

/* ---- the Hungarian matcher's cost blocks (reference models/richsem/matcher.py:49-78 with
 * util/box_ops.py:9-59) --------------------------------------------------------------------------------------------
 *     C[b][q][t] = w_bbox * |box_q - box_t|_1 + w_class * (focal-style class cost at label_t) + w_giou * (-GIoU(box_q, box_t))
 * for query q of image b against target t of THE SAME image only (the blocks the reference keeps, matcher.py:76-77), with the
 * reference's arithmetic in the reference's order.  logits (B, Q, C); boxes (B, Q, 4) cxcywh; tgt_ids (n_targets) int64;
 * tgt_boxes (n_targets, 4) cxcywh; tgt_offsets (B + 1) int64 ON THE DEVICE = exclusive prefix of the per-image target counts
 * (tgt_offsets[B] == n_targets).  cost: Q * n_targets elements; image b's (Q x T_b) row-major block starts at element
 * Q * tgt_offsets[b].  Several decoder outputs are matched by calling this once per output into consecutive slices of one buffer
 * and copying that buffer to the host once.  A label outside [0, C) gives NaN in its column. */
int msda_matcher_cost_f32(const float *logits, const float *boxes, const int64_t *tgt_ids, const float *tgt_boxes,
                          const int64_t *tgt_offsets, int B, int Q, int C, int64_t n_targets, double w_class, double w_bbox,
                          double w_giou, double alpha, float *cost, msda_stream_t stream);
int msda_matcher_cost_f64(const double *logits, const double *boxes, const int64_t *tgt_ids, const double *tgt_boxes,
                          const int64_t *tgt_offsets, int B, int Q, int C, int64_t n_targets, double w_class, double w_bbox,
                          double w_giou, double alpha, double *cost, msda_stream_t stream);

/* ---- feed-forward block of the transformer layers on the matrix cores ----------
 *     out = LayerNorm(x + W2 . relu(W1 . x + b1) + b2)
 * reference: models/richsem/deformable_transformer.py:862-866 (encoder forward_ffn), :940-944 (decoder forward_ffn), with
 * activation = relu and the dropouts inactive (p = 0.0 in the shipped configs, or eval mode).  bf16 storage, fp32
 * accumulation; the hidden activation is rounded to bf16 once (as a bf16 nn.Linear would), everything after the second
 * product (bias, residual, LayerNorm) is fp32 and rounded once at the store.  New capability: the reference runs fp32.
 *   x, out       (tokens, d_model) bf16, row-major;  d_model must be 256
 *   w1           (d_ffn, d_model) bf16 = linear1.weight;  b1 (d_ffn) f32 = linear1.bias;  d_ffn % 32 == 0, <= 4096
 *   w2_packed    linear2.weight (d_model, d_ffn) bf16 after msda_ffn_pack_w2_bf16 (a fixed permutation of the hidden
 *                columns inside every group of 32: repack whenever the weight changes);  b2 (d_model) f32
 *   ln_weight, ln_bias (d_model) f32, eps as nn.LayerNorm.  All pointers 16-byte aligned. */
/* Training: the forward that also writes the LayerNorm's 1 / sqrt(var + eps) per token (rstd: `tokens` floats) and its normalised input
 * yhat = (y - mean) * rstd ((tokens, 256) bf16) -- either may be NULL -- and the first step of the backward: from dy (gradient of out),
 * yhat and rstd the gradient dz at the LayerNorm's input (= gradient of the residual x and of the second product's output, bf16) and the
 * token sums grad_ln_weight = sum dy * yhat, grad_ln_bias = sum dy, grad_b2 = sum dz (f32, 256 each, zeroed by the call).  The products
 * of the backward are msda_lin256_forward_bf16 (recomputed hidden activation, dH with the ReLU mask), msda_conv_wgrad_bf16 (weight
 * gradients) and one library GEMM (dx = dz + dHm W1), see richsem_amd/functions/ffn.py. */
int msda_ffn_forward_train_bf16(const uint16_t *x, const uint16_t *w1, const float *b1, const uint16_t *w2_packed, const float *b2,
                                const float *ln_weight, const float *ln_bias, float eps, int tokens, int d_model, int d_ffn,
                                uint16_t *out, float *rstd, uint16_t *yhat, msda_stream_t stream);
int msda_ffn_ln_backward_bf16(const uint16_t *dy, const uint16_t *yhat, const float *rstd, const float *ln_weight, int tokens, int d_model,
                              uint16_t *dz, float *grad_ln_weight, float *grad_ln_bias, float *grad_b2, msda_stream_t stream);
/* Residual add + LayerNorm of the transformer layers (reference deformable_transformer.py:876-877: src = norm1(src + dropout1(src2)),
 * with the dropout inactive): out = LayerNorm(a + b) over 256 channels; a, b (b may be NULL), out (tokens, 256) bf16; ln_weight, ln_bias
 * f32; rstd (tokens) f32 and yhat (tokens, 256) bf16 for the backward (either may be NULL), which is msda_ffn_ln_backward_bf16: its dz is
 * the gradient of a and of b. */
int msda_add_layernorm_forward_bf16(const uint16_t *a, const uint16_t *b, const float *ln_weight, const float *ln_bias, float eps, int tokens,
                                    int d_model, uint16_t *out, float *rstd, uint16_t *yhat, msda_stream_t stream);

/* out = act(x W^T + b) for in_features = 256 on the matrix cores (csrc/lin256_mfma.hip; bf16 storage, fp32 accumulation): the two
 * token-parallel products of the feed-forward block's backward.  msda_lin256_pack_bf16: W (out_features, 256) bf16 row-major -> the same
 * number of elements in MFMA fragment order (out_features % 64 == 0).  epilogue 0: acc + bias (bias may be NULL); 1: relu(acc + bias);
 * 2: acc where relu_mask > 0, else 0 (relu_mask (tokens, out_features) bf16: the forward's hidden activation); 3: acc + bias with the
 * rows of masked tokens zeroed (relu_mask then points to `tokens` bytes, non-zero = masked: MSDeformAttn's value projection with its
 * padding mask, ops/modules/ms_deform_attn.py:94-96).  16-byte aligned pointers (the byte mask of epilogue 3: any alignment). */
int msda_lin256_pack_bf16(const uint16_t *w, int out_features, int in_features, uint16_t *packed, msda_stream_t stream);
int msda_lin256_forward_bf16(const uint16_t *x, const uint16_t *packed_w, const float *bias, const uint16_t *relu_mask, int epilogue,
                             int tokens, int in_features, int out_features, uint16_t *out, msda_stream_t stream);
/* Several 256 -> 256 layers stacked into one product (out_features = 256 * layers, W and bias concatenated along the outputs): out is
 * `layers` separate contiguous (tokens, 256) matrices, one after the other; row_mask (tokens bytes, non-zero = zero that token's rows)
 * may be NULL.  The decoder's cross-attention value projections of all its layers: the memory is read once. */
int msda_lin256_forward_stacked_bf16(const uint16_t *x, const uint16_t *packed_w, const float *bias, const uint8_t *row_mask, int tokens,
                                     int in_features, int out_features, uint16_t *out, msda_stream_t stream);
/* Masked multi-head self-attention of the decoder's queries (csrc/attn_mfma.hip; reference: nn.MultiheadAttention of
 * DeformableTransformerDecoderLayer, models/richsem/deformable_transformer.py:907, :974-978): softmax(q k^T / sqrt(32) + mask) v per
 * (image, head), head dimension 32, bf16 storage, fp32 softmax -- new capability (the reference runs torch's fp32 attention).
 * Token (i, b) is row i * bs + b of q / k / v (sequence-first, batch_first = 0) or row b * nq + i (batch_first = 1), `ld*` elements
 * apart, a head's 32 channels at column 32 h; out / dout the same rows, heads * 32 wide, contiguous; lse (bs * heads, ceil32(nq)) f32: log2-sum-exp per query, written by the forward, read by
 * the backward; mask_bits (nq, ceil(nq / 32)) uint32: bit j of word (q, kb) set = query q must not attend to key 32 kb + j;
 * maskt_bits the same of the transposed mask (both NULL = no mask); workspace: msda_attn_workspace_bytes(nq, bs, heads) bytes.
 * 16-byte aligned pointers, ld* multiples of 8.  Every element of out / dq / dk / dv is written. */
int64_t msda_attn_workspace_bytes(int nq, int bs, int heads);
int msda_attn_forward_bf16(const uint16_t *q, int ldq, const uint16_t *k, int ldk, const uint16_t *v, int ldv, const uint32_t *mask_bits,
                           int nq, int bs, int batch_first, int heads, uint16_t *out, float *lse, void *workspace, msda_stream_t stream);
int msda_attn_backward_bf16(const uint16_t *q, int ldq, const uint16_t *k, int ldk, const uint16_t *v, int ldv, const uint16_t *out,
                            const uint16_t *dout, const float *lse, const uint32_t *mask_bits, const uint32_t *maskt_bits, int nq, int bs,
                            int batch_first, int heads, uint16_t *dq, int lddq, uint16_t *dk, int lddk, uint16_t *dv, int lddv, void *workspace,
                            msda_stream_t stream);
/* The same product for fp32 tensors at fp32-level accuracy (both operands split into bf16 hi + lo parts, three bf16 MFMAs per tile;
 * csrc/lin256_mfma.hip): out (tokens, out_features) f32 = x (tokens, 256) f32 . W^T + bias.  msda_lin256_pack_f32: W (out_features, 256)
 * f32 -> 2 * out_features * 256 uint16 (hi and lo parts in fragment order); out_features % 32 == 0.  The MSDeformAttn module's fp32
 * projections (reference ops/modules/ms_deform_attn.py:52-56). */
int msda_lin256_pack_f32(const float *w, int out_features, int in_features, uint16_t *packed, msda_stream_t stream);
int msda_lin256_forward_f32(const float *x, const uint16_t *packed_w, const float *bias, int tokens, int in_features, int out_features,
                            float *out, msda_stream_t stream);
/* Diagnostic: non-NULL = the kernel adds up the shader clocks wave 0 of every workgroup spends per loop stage (wait for the
 * weight tile, barrier, first product, relu + conversion, second product) into 8 x 8 bytes per workgroup; NULL = off. */
int msda_ffn_debug_stamps(void *device_buffer);
int msda_ffn_pack_w2_bf16(const uint16_t *w2, int d_model, int d_ffn, uint16_t *w2_packed, msda_stream_t stream);
int msda_ffn_forward_bf16(const uint16_t *x, const uint16_t *w1, const float *b1, const uint16_t *w2_packed, const float *b2,
                          const float *ln_weight, const float *ln_bias, float eps, int tokens, int d_model, int d_ffn,
                          uint16_t *out, msda_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RICHSEM_MSDA_H */
