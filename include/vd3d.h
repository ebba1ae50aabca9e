/*
 * vd3d.h -- C ABI of libvd3d_hip.so: the MI355X (gfx950) depth-image-based stereo renderer.
 *
 * This is the drop-in boundary for ONE hot path of VisionDepth3D (reference tree
 * /root/reference, Python): the per-frame 2D->3D chain of core/render_3d.py.
 * The reference has no FFI; the boundary is a pair of Python call sites
 * (SURVEY.md 8(b)).  Each entry point below names the reference interface it replaces.
 *
 * Conventions
 *  - plain C, no torch types.  All image pointers are DEVICE (HBM) pointers unless a
 *    name ends in _host.  Sizes are in pixels.  Planes are dense row-major.
 *  - "rgb_chw"  = float32 [3][h][w], R,G,B planes, values in [0,1]  (reference frame_to_tensor layout,
 *                 core/render_3d.py:135-138)
 *  - "bgr_hwc"  = uint8  [h][w][3], B,G,R interleaved                (OpenCV frame layout,
 *                 core/render_3d.py:289-291)
 *  - every call is ASYNCHRONOUS on the ctx's HIP stream and performs no host
 *    synchronisation; temporal-tracker scalars live in device memory inside the
 *    ctx and are advanced by a device-side scalar stage.  Use vd3d_sync() or
 *    vd3d_state_export() to observe results on the host.
 *  - return value: 0 on success, negative vd3d_status on error; vd3d_last_error()
 *    returns a thread-local message.
 */
#ifndef VD3D_H
#define VD3D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 6 (round 6): new entry points vd3d_gemm_x3_* / vd3d_attention_x3_* (no struct changed); vd3d_debug_tune answers only knobs 3 and 4 unless built with
 *    -DVD3D_DEV_KNOBS; aten_threads / aten_sum_threads accept 1 .. 1024.
 * 5 (round 5): vd3d_shift_params gained aten_threads / reserved0 at its end (vd3d_render_params embeds it: its later fields moved by 8 bytes); vd3d_torch_math_aten.
 * 4 (round 5): vd3d_render_params::reserved0 became aten_sum_threads (same layout). */
#define VD3D_ABI_VERSION 6

typedef enum vd3d_status {
  VD3D_OK = 0,
  VD3D_E_INVALID = -1,    /* bad argument (the reference raises AssertionError/ValueError) */
  VD3D_E_HIP = -2,        /* HIP runtime error */
  VD3D_E_NOMEM = -3,
  VD3D_E_UNSUPPORTED = -4 /* valid in the reference, not built yet here (fails loudly, never falls back) */
} vd3d_status;

/* output_format strings of format_3d_output, core/render_3d.py:837-860 */
typedef enum vd3d_format {
  VD3D_FMT_HALF_SBS = 0,
  VD3D_FMT_FULL_SBS = 1,
  VD3D_FMT_VR = 2,
  VD3D_FMT_ANAGLYPH = 3,
  VD3D_FMT_INTERLACED = 4
} vd3d_format;

/* depth input encodings accepted by vd3d_render_frame */
typedef enum vd3d_depth_fmt {
  VD3D_DEPTH_F32 = 0,     /* float32 [h][w] already in [0,1] (precomputed; BASELINE configs 1 and 3) */
  VD3D_DEPTH_BGR_U8 = 1,  /* uint8 [h][w][3] depth-video frame; depth_to_tensor, core/render_3d.py:140-143 */
  VD3D_DEPTH_GRAY_U8 = 2  /* uint8 [h][w]: same as BGR_U8 with B=G=R (BGR2GRAY is the identity on gray) */
} vd3d_depth_fmt;

/*
 * Keyword parameters of pixel_shift_cuda(), core/render_3d.py:561-590, same names,
 * same meaning, same defaults (vd3d_shift_params_default).  Python floats are doubles
 * and are narrowed to float32 at the same point torch narrows them.
 */
typedef struct vd3d_shift_params {
  double fg_shift, mg_shift, bg_shift;     /* positional args 5-7 */
  int32_t blur_ksize;                      /* 9   */
  int32_t use_subject_tracking;            /* 1   */
  int32_t enable_floating_window;          /* 1   */
  int32_t enable_feathering;               /* 1   */
  int32_t enable_edge_masking;             /* 1   */
  int32_t enable_dynamic_convergence;      /* 1   */
  double feather_strength;                 /* 10.0 */
  double max_pixel_shift_percent;          /* 0.02 */
  double parallax_balance;                 /* 0.8  */
  double zero_parallax_strength;           /* 0.0  */
  double convergence_strength;             /* 0.0  */
  double depth_pop_gamma;                  /* 0.85 */
  double depth_pop_mid;                    /* 0.50 */
  double depth_stretch_lo;                 /* 0.05 */
  double depth_stretch_hi;                 /* 0.95 */
  double fg_pop_multiplier;                /* 1.20 */
  double bg_push_multiplier;               /* 1.10 */
  double subject_lock_strength;            /* 1.00 */
  /* Extension, not a pixel_shift_cuda keyword (round 5).  N >= 1: reproduce ATen's SCALAR TAILS for a reference process that runs torch with N
   * intra-op threads (torch.get_num_threads()): torch.pow / torch.sigmoid on a float32 CPU plane split its n elements over min(N, ceil(n / 32768))
   * threads and send the last (chunk length mod 32) elements of every chunk through libm (std::pow in double with the unrounded Python exponent,
   * glibc's expf) instead of the SLEEF vector bodies (core/render_3d.py:209,517,620).  No plane a video produces has such a tail (1920x1080 and
   * 3840x2160 split evenly over 1 .. 96 threads); odd sizes do.  0 (the default): SLEEF values everywhere.  Inside vd3d_render_frame and the sharded entry
   * points vd3d_render_params::aten_sum_threads is used instead of this field.
   * HOST ISA ASSUMPTION of the mode (both fields): the reference's torch is an x86-64 build that dispatches to its AVX-512 kernels (elementwise loops step
   * two 16-lane vectors: tail = chunk length mod 32; an AVX2-only host would have mod 16) on glibc >= 2.27 (expf = the FMA ifunc variant, restated and checked
   * on all 2^32 inputs against glibc 2.35).  For a reference on another ISA the mode stays deterministic but is no longer that reference's bits at sizes with
   * tails.  Range 0 .. 1024.  The Python shims (pixel_shift_cuda, render_sbs_3d) pass torch.get_num_threads() of the calling process by default (round 6). */
  int32_t aten_threads;                    /* 0 */
  int32_t reserved0;
} vd3d_shift_params;

/*
 * Per-clip constants of the render_sbs_3d loop body, core/render_3d.py:1227-1419.
 * The geometry block is what lines 1074-1138 and 1236-1259 derive once per clip
 * (host helper: visiondepth3d_amd.geometry.plan_geometry mirrors them).
 */
typedef struct vd3d_render_params {
  /* geometry */
  int32_t src_w, src_h;          /* decoded frame size */
  int32_t crop_x, crop_y;        /* centre crop to the target aspect (:1236-1248) */
  int32_t crop_w, crop_h;
  int32_t eye_w, eye_h;          /* target_eye_w/h : size both tensors are resized to (:1250-1263) */
  int32_t warp_w, warp_h;        /* resized_width/height: size pixel_shift_cuda warps at (:1285) */
  int32_t fit_w, fit_h;          /* per_eye_w/h: size each eye is fitted to before muxing (:1409-1417) */
  int32_t out_w, out_h;          /* muxed frame size */
  int32_t format;                /* vd3d_format */
  int32_t auto_crop_black_bars;  /* :1230-1234: detect_black_bars + crop per frame, then the aspect crop of :1236-1248 is
                                    re-derived per frame ON DEVICE from target_ratio (crop_x/y/w/h are then ignored) */
  /* sliders */
  vd3d_shift_params shift;       /* fg/mg/bg are the UNSCALED slider values; dyn_scale/ipd are applied per frame */
  double ipd_factor;             /* :1283,1308 */
  double dof_strength;           /* max_sigma of apply_dof_cuda; 0 disables (:1340) */
  double sharpness_factor;       /* apply_sharpening factor (:1406) */
  double color_saturation, color_contrast, color_brightness; /* apply_color_grade (:1362-1365) */
  double target_ratio;           /* aspect_ratios[selected_aspect_ratio] (:1072); used when auto_crop_black_bars */
  /* Association of the DOF Gaussian levels (apply_dof_cuda -> torchvision gaussian_blur, core/render_3d.py:806).  The reference
   * blurs with a dense k x k depthwise convolution whose summation order belongs to the convolution library of the machine it
   * runs on.  1 (what vd3d_render_params_default sets): the dense convolution in the order PyTorch's CPU build (oneDNN) uses --
   * row-major taps, one fused multiply-add per tap -- which reproduces the CPU reference's blur BIT FOR BIT
   * (tests/test_oracle_vs_golden.py::test_b2_attribution*); fused into the finishing kernel.
   * 0: separable columns-then-rows sums -- 4x less arithmetic, identical to <= 6e-7 in float, which the reference's own uint8
   * truncation turns into +-1 LSB on ~0.5 % of samples before the sharpen stage (gain ~4.5): an opt-in fast mode. */
  int32_t dof_dense_conv;
  /* Float32 summation order of the two `torch.mean` calls on the path (compute_dynamic_parallax_scale :418, compute_motion_metric :928; round 5).
   * The reference's values are those of ATen's float32 cascade sum, which depends on the number of intra-op threads torch runs with
   * (TensorIterator splits reductions of >= 32 768 elements across them).  N >= 1: reproduce torch with N threads bit for bit
   * (torch.get_num_threads() of the reference process; its default is the machine's core count).  0 (what vd3d_render_params_default sets):
   * the correctly rounded mean of the exact sum -- independent of any thread count, within one float32 ULP of every N. */
  int32_t aten_sum_threads;                /* also the N of vd3d_shift_params::aten_threads for the frames of this clip */
} vd3d_render_params;

/*
 * All temporal-tracker scalars of the reference, made explicit (the reference keeps four of
 * them as never-reset module singletons, core/render_3d.py:284-285,500,511).
 */
typedef struct vd3d_state {
  /* FloatingWindowTracker :479-500 (alpha 0.97) */
  double fw_prev_offset;
  int32_t fw_frame_counter;
  /* DepthPercentileEMA :233-262 (float32 0-d tensors in the reference) */
  int32_t ema_valid;
  float ema_lo, ema_hi;
  /* ConvergenceEMA :273-280 (alpha 0.97) */
  int32_t conv_valid;
  int32_t bar_prev_width;        /* FloatingBarEaser :502-511 */
  double conv_val;
  /* FocalDepthTracker :895-922 */
  int32_t focal_valid;
  int32_t smooth_valid;          /* ShiftSmoother :463-477 */
  double focal;
  double sm_fg, sm_mg, sm_bg;
  /* plane state validity: TemporalDepthFilter.prev_depth (:220-229) and prev_depth_tensor (:1181,1463) */
  int32_t tdf_valid;
  int32_t prev_depth_valid;
} vd3d_state;

/* Per-frame scalars the device-side scalar stage produced (diagnostics, multi-GPU exchange, tests). */
typedef struct vd3d_frame_scalars {
  float q_lo, q_hi;              /* quantile(d,0.02/0.98) of the filtered depth (a5 inputs) */
  float ema_lo, ema_hi;          /* after the EMA */
  float mean_c, var_c;           /* centre-crop mean / unbiased variance (a6) */
  double dyn_scale;
  float s_norm;                  /* estimate_subject_depth(normalised eye-res depth): focal candidate, bar input */
  float mad;                     /* mean |d_t - d_{t-1}| (a15) */
  float s0, q05, q95, s1;        /* pixel_shift_cuda internals */
  float zpo_raw;                 /* zero-parallax offset before the tracker */
  double zpo;                    /* after FloatingWindowTracker */
  double focal;
  double stable_zero;
  int32_t bar_width;
  int32_t bar_side;              /* 0 none, 1 = mask right columns, 2 = mask left columns */
  int32_t collapse;              /* DepthPercentileEMA "hi-lo<1e-5" guard taken */
  int32_t crop_top, crop_bottom; /* detect_black_bars result of this frame (0,0 when auto crop is off) */
  int32_t reserved;
} vd3d_frame_scalars;

typedef struct vd3d_ctx vd3d_ctx;

/* ---- lifecycle -------------------------------------------------------------------------- */
int vd3d_abi_version(void);
const char* vd3d_last_error(void);
void vd3d_shift_params_default(vd3d_shift_params* p);   /* defaults of core/render_3d.py:569-589 */
void vd3d_render_params_default(vd3d_render_params* p);

/* One ctx per device per thread.  `stream` is a hipStream_t: NULL = the device's default (null) stream, which is
 * what PyTorch enqueues on by default, so tensors and vd3d calls stay ordered; VD3D_STREAM_PRIVATE = create and own a
 * private non-blocking stream (the caller then orders against it with events / vd3d_sync). */
#define VD3D_STREAM_PRIVATE ((void*)(intptr_t)-1)
int vd3d_ctx_create(int device, void* stream, vd3d_ctx** out);
int vd3d_ctx_destroy(vd3d_ctx* ctx);
int vd3d_sync(vd3d_ctx* ctx);                            /* hipStreamSynchronize */
/* The context's stream handle.  ESCAPE SEMANTICS: for a VD3D_STREAM_PRIVATE context the handle is from now on assumed to be held by the
 * caller (PyTorch wraps it and may record events on it when tensors marked with record_stream() are freed, possibly after the context
 * is gone), so vd3d_ctx_destroy no longer destroys that stream -- it is left to the process.  Query it once and keep it; a context whose
 * stream was never queried destroys it.  The same holds for vd3d_ctx_pixel_stream. */
void* vd3d_ctx_stream(vd3d_ctx* ctx);
/* re-target a context created on a caller-owned stream (not VD3D_STREAM_PRIVATE): later calls enqueue on `stream`, which is first
 * ordered behind everything the context has enqueued so far */
int vd3d_ctx_set_stream(vd3d_ctx* ctx, void* stream);
void* vd3d_ctx_pixel_stream(vd3d_ctx* ctx);              /* second stream of vd3d_set_pixel_overlap (NULL before it was enabled) */
void* vd3d_ctx_pixel_stream_k(vd3d_ctx* ctx, int k);     /* pixel stream k of vd3d_set_pixel_overlap(ctx, n), k < n <= 4 (k = 0: the one above) */

/* ---- tracker state (replaces the module singletons; lets ranks exchange it) -- */
int vd3d_state_reset(vd3d_ctx* ctx);                     /* fresh process + fresh render */
int vd3d_state_new_clip(vd3d_ctx* ctx);                  /* what render_sbs_3d re-creates per call (:1174-1182) */
int vd3d_state_export(vd3d_ctx* ctx, vd3d_state* host_out);      /* synchronises */
int vd3d_state_import(vd3d_ctx* ctx, const vd3d_state* host_in);
/* plane state: TemporalDepthFilter.prev_depth and prev_depth_tensor, float32 [eye_h][eye_w] device planes */
int vd3d_state_planes(vd3d_ctx* ctx, float** tdf_prev, float** norm_prev, int* eye_h, int* eye_w);
int vd3d_last_scalars(vd3d_ctx* ctx, vd3d_frame_scalars* host_out); /* synchronises */

/* ---- B1: pixel_shift_cuda, core/render_3d.py:561-712 --------------------------------------
 * rgb_chw [3][in_h][in_w] f32, depth [in_h][in_w] f32 (device) -> left/right bgr_hwc u8 [H][W][3]
 * (device), optional final_shift f32 [H][W].  Mutates the FloatingWindowTracker in ctx state
 * exactly as the reference mutates its module global (:651). */
int vd3d_pixel_shift(vd3d_ctx* ctx, const float* rgb_chw, const float* depth, int in_h, int in_w,
                     int W, int H, const vd3d_shift_params* p,
                     uint8_t* left_bgr, uint8_t* right_bgr, float* shift_or_null);

/* ---- B2: one iteration of the render_sbs_3d loop body, core/render_3d.py:1227-1419 ---------
 * frame_bgr u8 [src_h][src_w][3], depth per depth_fmt (device) -> out u8 [out_h][out_w][3] (device). */
int vd3d_render_frame(vd3d_ctx* ctx, const uint8_t* frame_bgr, const void* depth, int depth_fmt,
                      const vd3d_render_params* p, uint8_t* out_bgr);

/* The same loop iteration for a frame listed by skip_blank_frames (core/render_3d.py:1046-1060, branch :1278-1281): both eyes are
 * the raw source frame (source size), pixel_shift_cuda / ipd scaling / focal tracker / DOF / colour grade are skipped, the depth
 * filters, ShiftSmoother, dynamic scale and floating-window bars advance; sharpen, fit and mux run on the source-sized frame.
 * Which frames are blank is the caller's knowledge (the reference asks ffmpeg's blackdetect, core/ffmpeg_blackdetect.py:23-81). */
int vd3d_render_frame_blank(vd3d_ctx* ctx, const uint8_t* frame_bgr, const void* depth, int depth_fmt,
                            const vd3d_render_params* p, uint8_t* out_bgr);

/* ---- frame sharding: ONE clip over G GPUs, bit-identical to 1 GPU --------------------------------------------
 * The reference renders strictly frame by frame (core/render_3d.py:1194-1463); its temporal state is a plane EMA
 * (TemporalDepthFilter :220-229), a percentile EMA (:233-262) and non-linear scalar trackers (:463-511,895-922).  Here a "step" is
 * a window of n <= 512 consecutive frames cut into G contiguous chunks, one per rank.  Per step every rank runs
 *   [vd3d_tdf_plane_import]   the filtered plane of the frame before its chunk, received from the previous chunk's owner
 *                             (one eye-size float32 plane per chunk boundary, point to point);
 *   vd3d_shard2_p1            per own frame, in order: ingest + plane EMA + exact q.02 / q.98 of the filtered plane;
 *   [vd3d_tdf_plane_export]   the plane after its last frame, sent on to the next chunk's owner;
 *   all-gather of 2 floats per frame, vd3d_shard2_r1: DepthPercentileEMA replayed on every rank;
 *   vd3d_shard2_p3            per own frame: every other measurement (4 x int64 per frame), planes kept in the frame's slot;
 *   all-gather of 4 x int64 per frame, vd3d_shard2_r2: the remaining trackers replayed in frame order, own slots patched;
 *   vd3d_shard_pixels[_blank] per own frame: shift plane, warp, DOF, grade, sharpen, mux.
 * Data-path traffic per step and rank: <= 40 B per frame of records + one plane per chunk boundary. */
int vd3d_shard_begin(vd3d_ctx* ctx, const vd3d_render_params* p, int n_slots);
int vd3d_shard_pixels(vd3d_ctx* ctx, int slot, const vd3d_render_params* p, uint8_t* out_bgr);
/* the pixel pass of an own frame listed by skip_blank_frames (core/render_3d.py:1278-1281): both eyes are the source frame */
int vd3d_shard_pixels_blank(vd3d_ctx* ctx, int slot, const uint8_t* frame_bgr, const vd3d_render_params* p, uint8_t* out_bgr);
/* Overlapped pixel passes (no reference counterpart: the reference renders strictly frame by frame, core/render_3d.py:1194-1463).
 * enable != 0: vd3d_shard_pixels is enqueued on a second stream of the context, ordered after all work enqueued so far; the
 * measurement chain of the next step (vd3d_shard2_p1 .. r2; latency-bound scans) then overlaps the pixel kernels of this one.  A call
 * that overwrites a slot first waits for the pixel pass still reading it, so callers alternate between two slot sets to get the overlap.
 * Outputs are complete after vd3d_sync, or -- for consumers ordered on the context's stream -- after vd3d_join_pixels. */
/* enable = 0: off; 1: one pixel stream; 2 .. 4: consecutive pixel passes go round-robin over that many streams, each with its own shift
 * plane and warped eyes, so the kernels of neighbouring frames share the CUs (a consumer of the muxed frames orders itself behind
 * vd3d_join_pixels / vd3d_wait_pixels, or behind every vd3d_ctx_pixel_stream_k). */
int vd3d_set_pixel_overlap(vd3d_ctx* ctx, int enable);
int vd3d_join_pixels(vd3d_ctx* ctx);
int vd3d_wait_pixels(vd3d_ctx* ctx, int slot);   /* host waits for the outstanding overlapped pixel pass of `slot` (no-op if none) */

/* with auto_crop_black_bars: vd3d_shard2_p0 per own frame (crop rectangle {x,y,w,h} -> device int[4]), all-gather of the
 * rectangles, vd3d_shard2_set_crops(frame order) -- before vd3d_shard2_p1 of the step */
int vd3d_shard2_p0(vd3d_ctx* ctx, const uint8_t* frame_bgr, const vd3d_render_params* p, int* crop_out_dev);
int vd3d_shard2_set_crops(vd3d_ctx* ctx, const int* crops_all_dev, int n);
int vd3d_shard2_p1(vd3d_ctx* ctx, const uint8_t* frame_bgr, const void* depth, int depth_fmt,
                   const vd3d_render_params* p, int step_idx, int slot, float* q_out_dev);
/* TemporalDepthFilter.prev_depth (float32 [eye_h][eye_w]) out of / into the context: the chunk-boundary hand-off.  import with
 * valid = 0 installs "no previous frame" (what a fresh clip starts from). */
int vd3d_tdf_plane_export(vd3d_ctx* ctx, float* dst_dev, int eye_h, int eye_w);
int vd3d_tdf_plane_import(vd3d_ctx* ctx, const float* src_dev, int eye_h, int eye_w, int valid);
int vd3d_shard2_r1(vd3d_ctx* ctx, const float* q_all_dev, int n);
int vd3d_shard2_p3(vd3d_ctx* ctx, int slot, int step_idx, const vd3d_render_params* p, long long* m_out_dev);
/* Batched forms for the own frames of a step, which are consecutive (slots slot0 .. slot0 + n - 1, step indices step_idx0 ..,
 * n <= 16): P1 in two launches (the plane EMA -- a per-pixel recurrence of core/render_3d.py:225-229 -- walks the frames inside the
 * ingest kernel, the exact-quantile passes of the n frames run side by side), P3 in five launches for all n frames.  Same results
 * as n single calls; frames_bgr / depths are HOST arrays of n device pointers, q_out_dev = float[n][2], m_out_dev = int64[n][4]. */
int vd3d_shard2_p1_batch(vd3d_ctx* ctx, const uint8_t* const* frames_bgr, const void* const* depths, int depth_fmt,
                         const vd3d_render_params* p, int step_idx0, int slot0, int n, float* q_out_dev);
int vd3d_shard2_p3_batch(vd3d_ctx* ctx, int slot0, int step_idx0, int n, const vd3d_render_params* p, long long* m_out_dev);
/* blank_host_or_null[t] != 0: frame t of the step is in the skip_blank_frames set (no ipd scaling, no focal / FloatingWindowTracker
 * update for it, core/render_3d.py:1278-1281,1334-1337) */
int vd3d_shard2_r2(vd3d_ctx* ctx, const long long* m_all_dev, const int* own_slot_host, const uint8_t* blank_host_or_null, int n,
                   const vd3d_render_params* p);

/* ---- heal_missing_pixels(warped_frame, warped_depth, original_frame, edge_mask, heal_strength=0.5), core/render_3d.py:431-459 (a23):
 * the gradient-based hole fill.  warped_chw / original_chw / out_chw: float32 [3][H][W] (rgb_chw), edge_mask_or_null: float32 [H][W]
 * or NULL; the reference's warped_depth argument is unused by the reference and has no counterpart.  The reference's render
 * loop never calls this function, so it is an optional stage here too: callers that want it run it between B1 and the
 * finishing stage.  out may not alias the inputs. */
int vd3d_heal_missing_pixels(vd3d_ctx* ctx, const float* warped_chw, const float* original_chw, const float* edge_mask_or_null,
                             int H, int W, double heal_strength, float* out_chw);

/* ---- optional NV12 wire format at the frame I/O boundary.  The reference moves bgr24 (cv2.VideoCapture.read() in,
 * `ffmpeg -f rawvideo -pix_fmt bgr24` out, core/render_3d.py:987,1222,1143-1163,1422-1427) and leaves colour conversion to ffmpeg on
 * the host; with a decoder that emits NV12 and `-pix_fmt nv12` on the encoder pipe the wire carries 1.5 bytes per pixel and the
 * conversion runs on the frame already in HBM.  BT.601 limited range, 20-bit fixed point (OpenCV's cvtColor constants = swscale's
 * default matrix); no reference counterpart to be bit-exact with.  h and w even; pitches in bytes; uv_plane holds h/2 rows of
 * interleaved (U, V). */
int vd3d_nv12_to_bgr(vd3d_ctx* ctx, const uint8_t* y_plane, long long y_pitch, const uint8_t* uv_plane, long long uv_pitch, int h, int w,
                     uint8_t* out_bgr);
int vd3d_bgr_to_nv12(vd3d_ctx* ctx, const uint8_t* bgr, int h, int w, uint8_t* y_plane, long long y_pitch, uint8_t* uv_plane,
                     long long uv_pitch);

/* ---- cv2.resize(src, (dw, dh), interpolation=cv2.INTER_CUBIC) on uint8 images with cn = 1 or 3 interleaved channels: the resize of
 * the uint8 depth map back to the source size when the depth tab runs at an explicit inference size (a24,
 * core/render_depth.py:1914-1917) and the size changes of the up-scale stage (core/merged_pipeline.py:260-264).  OpenCV's
 * fixed-point bicubic (A = -0.75, 11-bit coefficients, replicate border, one rounding).  src != dst. */
int vd3d_resize_cubic_u8(vd3d_ctx* ctx, const uint8_t* src, int sh, int sw, int cn, uint8_t* dst, int dh, int dw);
/* cv2.resize(src, (dw, dh), interpolation=cv2.INTER_AREA) on uint8 BGR (3 interleaved channels): run_esrgan's input_res_pct < 100
 * (core/merged_pipeline.py:246-248) and pad_to_aspect_ratio (core/render_3d.py:121).  Down-scale ratios up to 10. */
int vd3d_resize_area_u8(vd3d_ctx* ctx, const uint8_t* src_bgr, int sh, int sw, uint8_t* dst_bgr, int dh, int dw);
/* cv2.resize(src, (dw, dh)) with OpenCV's default interpolation (INTER_LINEAR, fixed point) on uint8 BGR: format_3d_output's VR branch */
int vd3d_resize_linear_u8(vd3d_ctx* ctx, const uint8_t* src_bgr, int sh, int sw, uint8_t* dst_bgr, int dh, int dw);
/* format_3d_output(left, right, fmt) (core/render_3d.py:837-860) on two uint8 BGR eyes of h x w (device pointers).  out_bgr: [h][2 w][3] for the
 * SBS formats, [1600][2880][3] for VR (eyes of another size are resized with INTER_LINEAR first, like the reference), [h][w][3] for anaglyph
 * and interlaced. */
int vd3d_format_3d_output(vd3d_ctx* ctx, const uint8_t* left_bgr, const uint8_t* right_bgr, int h, int w, int format, uint8_t* out_bgr);

/* ---- up-scale stage glue (core/merged_pipeline.py:219-236).  The network between the two calls is the caller's
 * (visiondepth3d_amd.upscale runs it on PyTorch-ROCm).
 *   preprocess : preprocess_esr :219-223 -- BGR uint8 rows (pitch_bytes apart, so a tile crop needs no copy) -> RGB / 255 as
 *                float32, bf16 or fp16 (the reference's ONNX models are fp16), planar [3][h][w] or channels-last [h][w][3]
 *   postprocess: postprocess_esr :225-229 -- RGB float32 [h][w] planes (or channels-last) -> clip(0,1) * 255 truncated, BGR uint8;
 *                only the (cy, cx, ch, cw) window is written (the tile centre of _esrgan_tiled :266-284)
 *   add_weighted: blend_images :231-236 -- cv2.addWeighted(a, alpha, b, beta, gamma) on uint8 */
int vd3d_esr_preprocess(vd3d_ctx* ctx, int dtype, const uint8_t* frame_bgr, long long pitch_bytes, int h, int w, int channels_last,
                        void* out_rgb);
int vd3d_esr_postprocess(vd3d_ctx* ctx, const float* pred_rgb, int h, int w, int channels_last, int cy, int cx, int ch, int cw,
                         uint8_t* out_bgr, long long pitch_bytes);
int vd3d_add_weighted_u8(vd3d_ctx* ctx, const uint8_t* a, double alpha, const uint8_t* b, double beta, double gamma, long long n,
                         uint8_t* out);
/* run_rife's glue (core/merged_pipeline.py:195-218): concatenate_images + preprocess_rife -- two uint8 BGR frames / 255 stacked to six
 * channels in the frames' own order, float32 / bf16 / fp16, planar [6][h][w] or channels-last; and the output side -- float32
 * [3][h][w] (or channels-last) -> clip(0,1) * 255 truncated, channel order untouched.  The interpolation network between the two is
 * the caller's (the reference's RIFE ONNX graph is not in /root/reference). */
int vd3d_rife_preprocess(vd3d_ctx* ctx, int dtype, const uint8_t* frame1_bgr, const uint8_t* frame2_bgr, int h, int w, int channels_last,
                         void* out6);
int vd3d_rife_postprocess(vd3d_ctx* ctx, const float* pred3, int h, int w, int channels_last, uint8_t* out_bgr);

/* One body layer of the up-scale network on the matrix cores: y = PReLU(conv3x3(x, stride 1, zero padding 1) + bias), 64 -> 64 channels,
 * fp16 NHWC [H][W][64] in and out, float32 accumulate (v_mfma_f32_32x32x16_f16).  The reference runs these layers inside its ONNX
 * session (core/merged_pipeline.py:250-252); realesr-general-x4v3 has 32 of them.
 *   w_frag: the layer's weight Wt[oc][ic][kh][kw] pre-arranged in fragment order, fp16 [36 steps][2][64 lanes][8]:
 *           step = (kh*3 + kw)*4 + kc, element [step][t][l][j] = Wt[32 t + (l & 31)][16 kc + 8 (l >> 5) + j][kh][kw]
 *           (visiondepth3d_amd.upscale.conv_weight_fragments builds it)
 *   bias, slope_or_null: float32 [64] (slope NULL = no activation).  All pointers 16-byte aligned; x != y. */
int vd3d_conv3x3_c64_f16(vd3d_ctx* ctx, const void* x_nhwc, int H, int W, const void* w_frag, const float* bias, const float* slope_or_null,
                         void* y_nhwc);
/* The other two layers of the compact (SRVGG) networks the reference's model list holds (VisionDepth3D.py:1094-1098), also inside its ONNX
 * session (core/merged_pipeline.py:250-252):
 *   head: y = PReLU(conv3x3(x) + bias), 3 -> 64 channels, x fp16 NHWC [H][W][3] (vd3d_esr_preprocess, channels_last), y fp16 NHWC [H][W][64];
 *         w27x64: float32 [27][64], row (kh*3 + kw)*3 + ic, column oc = Wt[oc][ic][kh][kw]; float32 accumulate in that row order.
 *   tail: the 64 -> 3 r^2 convolution is vd3d_conv3x3_c64_f16 with weight fragments and bias zero-padded to 64 output channels and no activation;
 *         vd3d_esr_tail_f32 then applies pixel_shuffle(r) and adds the nearest-neighbour up-sampled input (the fp16 addition of the network)
 *         and writes the float32 planar prediction [3][r H][r W] that vd3d_esr_postprocess takes.  r = 2 or 4. */
int vd3d_conv3x3_head_f16(vd3d_ctx* ctx, const void* x_nhwc3, int H, int W, const float* w27x64, const float* bias, const float* slope_or_null,
                          void* y_nhwc64);
int vd3d_esr_tail_f32(vd3d_ctx* ctx, const void* t_nhwc64, const void* x_nhwc3, int H, int W, int r, float* out_planar);

/* ---- depth hand-off (a24): transformers' bicubic post-process to (H,W) + convert_depth_to_grayscale
 * (core/render_depth.py:585-611,1914-1916) for a batch of B predictions [B][ph][pw] float32 -> uint8 [B][H][W].
 * Replaces the reference's 8-bit depth video on disk while keeping its quantisation. */
int vd3d_depth_handoff(vd3d_ctx* ctx, const float* pred, int B, int ph, int pw, int H, int W, int invert, uint8_t* out_gray);

/* element type of the depth network's activations (a25).  The reference loads its Hugging Face depth models with
 * AutoModelForDepthEstimation.from_pretrained(checkpoint) and no dtype, i.e. float32 (core/render_depth.py:758-759,823-824):
 * VD3D_DT_F32 is the like-for-like precision, VD3D_DT_BF16 the optional reduced-precision mode. */
typedef enum vd3d_dtype { VD3D_DT_BF16 = 0, VD3D_DT_F32 = 1, VD3D_DT_F16 = 2 /* vd3d_esr_preprocess only */ } vd3d_dtype;

/* ---- depth-net input preparation (a25, core/render_depth.py:1106-1119 -> DPTImageProcessor): B uint8 BGR frames
 * [B][H][W][3] -> antialiased bicubic resize to (th,tw), 1/255, (x-mean)/std, RGB, `dtype`, NHWC [B][th][tw][3].
 * Returns VD3D_E_UNSUPPORTED when the down-scale factor exceeds the kernel's tap budget (scale > ~5.5). */
int vd3d_depth_preprocess(vd3d_ctx* ctx, const uint8_t* frames_bgr, int B, int H, int W, int th, int tw,
                          const float* mean3_host, const float* std3_host, int dtype, void* out_nhwc);

/* Transformer-block glue of the depth network (a25): s = x + y, n = LayerNorm(s)*gamma + beta on [rows][cols] device arrays
 * of `dtype` in one pass (y == NULL: LayerNorm only, out_sum unused).  cols in {384, 768, 1024} (DA-V2 S/B/L), else
 * VD3D_E_UNSUPPORTED.  Replaces torch's add + layer_norm kernel pair inside DepthPipe's fused backbone. */
int vd3d_add_layernorm(vd3d_ctx* ctx, int dtype, const void* x, const void* y_or_null, const void* gamma, const void* beta,
                       float eps, int64_t rows, int cols, void* out_sum, void* out_norm);

/* The linear layers and the attention of the depth network's transformer blocks (a25; core/render_depth.py:1106-1119 runs them in float32 through the
 * Hugging Face pipeline) as SPLIT-operand GEMMs on the 16-bit matrix cores -- OPT-IN modes of DepthPipe (gemm="bf16x3" | "fp16x2"; the default stays
 * hipBLASLt's float32 GEMM + PyTorch's float32 attention).  gfx950 has no TF32 and its float32-input MFMA runs at 1/16 of the bf16 / fp16 rate.
 *   Y[M][N] = X[M][K] . W[N][K]^T + bias[N]   (row-major float32 device arrays, leading dimensions K / K / N; epilogue 1: + exact GELU)
 * mode VD3D_X3_BF16X3: every float32 operand is split EXACTLY into three bf16 terms (8 + 8 + 8 significant bits); six of the nine term products -- all but
 *   x2 w3, x3 w2, x3 w3, together <= 2^-23 |x w| -- go through v_mfma_f32_32x32x16_bf16 with float32 accumulation: float32-faithful (the error of a float32
 *   GEMM in another summation order; tests/test_hip_gemm.py vs float64 beside hipBLASLt), 6 MFMAs per MAC.  NaN / Inf inputs give NaN.
 * mode VD3D_X3_FP16X2: every operand as two fp16 terms, h1 = fp16(x), h2 = fp16(x - h1), round to nearest: 22 significant bits (|x - h1 - h2| <= 2^-22 |x|
 *   while h2 is a normal fp16 number, i.e. |x| >= 2^-2; below that the absolute error is <= 2^-25); three products x1 w1 + x1 w2 + x2 w1 -- HALF the matrix
 *   work.  Weight rows are pre-scaled by an exact power of two (undone in the epilogue) so that their second terms stay normal; activations are used as
 *   they are and must stay below 65 504 in magnitude (fp16's range; larger values give Inf / NaN).  Per product up to ~3 x 2^-22 relative instead of one
 *   float32 rounding; in a K >= 64 dot product that sits below the float32 accumulation error both modes share (measured beside hipBLASLt in the tests).
 * The weights are split and packed once: vd3d_gemm_x3_weight_bytes(N, K, mode) bytes (< 0: K is not a positive multiple of 16, or an unknown mode), filled
 * by vd3d_gemm_x3_pack_weights; the image is opaque, belongs to its mode and is only valid for this library version. */
enum { VD3D_GEMM_EPI_NONE = 0, VD3D_GEMM_EPI_GELU = 1 };
enum { VD3D_X3_BF16X3 = 0, VD3D_X3_FP16X2 = 1 };
int64_t vd3d_gemm_x3_weight_bytes(int N, int K, int mode);
int vd3d_gemm_x3_pack_weights(vd3d_ctx* ctx, const float* W, int N, int K, int mode, void* image);
int vd3d_gemm_x3(vd3d_ctx* ctx, const float* X, int64_t M, int K, const void* w_image, int N, int mode, const float* bias_or_null, int epilogue, float* Y);
/* The attention of the same blocks in the same arithmetic: out[b][t][h][:] = softmax_t'(q[b][t][h] . k[b][t'][h] * scale) v[b][t'][h] with
 * q / k / v = qkv[b][t][0 / 1 / 2][h][:], qkv the contiguous float32 output [B][T][3][H][D] of the fused QKV linear, out [B][T][H][D] float32.
 * Q, K, V and the probabilities are split like the GEMM operands of `mode` (fp16x2: q, k, v scaled by 2^4 and the probabilities by 2^10 before their split,
 * exactly, undone in the logits' scale and the final normalisation), both matrix products run on the matrix cores with float32 accumulation, the online
 * softmax is float32 (exp2 of the pre-scaled logits).  D = 64 only (every DINOv2 size), else VD3D_E_UNSUPPORTED.
 * `workspace`: vd3d_attention_x3_workspace_bytes(B, T, H, D, mode) bytes of device memory (the split images), 16-byte aligned, owned by the caller. */
int64_t vd3d_attention_x3_workspace_bytes(int B, int T, int H, int D, int mode);
int vd3d_attention_x3(vd3d_ctx* ctx, const float* qkv, int B, int T, int H, int D, float scale, int mode, void* workspace, int64_t workspace_bytes, float* out);

/* The 3 x 3 convolutions of the DPT neck / head in the fp16x2 arithmetic (the third piece of DepthPipe(gemm="fp16x2")): stride 1, zero padding 1, no bias
 * (DepthPipe runs them bias-free with a glue launch behind each), float32 channels_last: X [B][H][W][Cin] -> Y [B][H][W][Cout], W the module's float32
 * weight [Cout][Cin][3][3].  Cin a multiple of 16, Cout 32, 64 or 128 (the fusion stage and the head's two 3 x 3 convolutions of DA-V2-Small / -Base), else
 * VD3D_E_UNSUPPORTED -- the caller keeps the library convolution for those.  Same arithmetic contract as vd3d_gemm_x3 in mode VD3D_X3_FP16X2 (operands as two
 * fp16 terms, three MFMA products, float32 accumulation, weights pre-scaled per output channel by an exact power of two, |x| < 65 504).
 * The weights are split and packed once into vd3d_conv3x3_x2_weight_bytes(Cin, Cout) bytes (< 0: unsupported shape). */
int64_t vd3d_conv3x3_x2_weight_bytes(int Cin, int Cout);
int vd3d_conv3x3_x2_pack_weights(vd3d_ctx* ctx, const float* W, int Cin, int Cout, void* image);
int vd3d_conv3x3_x2(vd3d_ctx* ctx, const float* X, int B, int H, int W, int Cin, const void* w_image, int Cout, float* Y);

/* F.interpolate(mode="bilinear", align_corners=True) of an NHWC (channels_last) tensor [B][ih][iw][C] -> [B][oh][ow][C] of
 * `dtype`, C a multiple of 8 (bf16) / 4 (f32): the up-samplings of the DPT neck / head (a25). */
int vd3d_upsample_bilinear_nhwc(vd3d_ctx* ctx, int dtype, const void* in, void* out, int B, int ih, int iw, int oh, int ow, int C);

/* Float32 glue between the library convolutions of the DPT neck / head (a25; transformers DepthAnythingPreActResidualLayer /
 * FeatureFusionLayer / DepthEstimationHead as the reference's pipeline runs them, core/render_depth.py:1106-1119).  NHWC maps of n_pix
 * pixels x C channels (C a multiple of 4).
 *   vd3d_nhwc_bias_act_f32:  v = y [+ bias[c]] [+ r1]; [v = r2 + v]; [v = max(v, 0)] -> out (may alias y); [relu_out = max(v, 0)]
 *   vd3d_upsample_bilinear_bias_nhwc_f32:  vd3d_upsample_bilinear_nhwc + bias[c] on the interpolated value (up(conv + b) == up(conv) + b)
 *   vd3d_dpt_head_tail_f32:  out[p] = max(b3 + sum_c w3[c] * max(y[p][c] + b2[c], 0), 0) * scale -- conv2's bias, ReLU, the 1x1 conv3, its
 *                            bias, the final ReLU and max_depth of the head in one pass; C in {16, 32, 64} */
int vd3d_nhwc_bias_act_f32(vd3d_ctx* ctx, const float* y, const float* bias_or_null, const float* r1_or_null, const float* r2_or_null, int relu,
                           int64_t n_pix, int C, float* out, float* relu_out_or_null);
int vd3d_upsample_bilinear_bias_nhwc_f32(vd3d_ctx* ctx, const float* in, const float* bias, float* out, int B, int ih, int iw, int oh, int ow, int C);
int vd3d_dpt_head_tail_f32(vd3d_ctx* ctx, const float* y, const float* b2, const float* w3, float b3, float scale, int64_t n_pix, int C, float* out);

/* ---- preview visualisers: generate_preview_image, core/preview_utils.py:23-84, the exactly defined types.
 * left / right: uint8 BGR [h][w][3] eyes (outputs of vd3d_pixel_shift).  out: [h][w][3], except HSBS: [h][2*(w/2)][3].
 * The colour-mapped heat-maps and the arrow overlay (OpenCV colour-map tables / line rasteriser) return VD3D_E_UNSUPPORTED. */
typedef enum vd3d_preview {
  VD3D_PREVIEW_INTERLACED = 0,   /* even rows left, odd rows right */
  VD3D_PREVIEW_HSBS = 1,         /* cv2.resize(eye, (w/2, h)) (INTER_LINEAR, 11-bit fixed point) + hstack */
  VD3D_PREVIEW_LR_DIFF = 2,      /* cv2.absdiff */
  VD3D_PREVIEW_FEATHER_BLEND = 3,/* the left eye */
  VD3D_PREVIEW_RED_BLUE = 4      /* (right.B, right.G, left.R) */
} vd3d_preview;
int vd3d_preview_image(vd3d_ctx* ctx, int type, const uint8_t* left_bgr, const uint8_t* right_bgr, int h, int w, uint8_t* out_bgr);
/* The colour-mapped types of the same function (core/preview_utils.py:42-66): type 0 "Shift Heatmap" (cv2.normalize NORM_MINMAX),
 * 1 "Shift Heatmap (Abs)", 2 "Shift Heatmap (Clipped +-5px)", 3 "Feather Mask"; shift_map: float32 [h][w] (pixel_shift_cuda's third return),
 * lut_bgr_dev: 256 x 3 uint8 BGR table in device memory -- OpenCV's COLORMAP_JET (types 0-2) / COLORMAP_BONE (3) as the caller obtained it
 * (cv2.applyColorMap(np.arange(256, dtype=np.uint8), cmap)); the library holds no copy of those tables. */
int vd3d_preview_heatmap(vd3d_ctx* ctx, int type, const float* shift_map, int h, int w, const uint8_t* lut_bgr_dev, uint8_t* out_bgr);
/* "Overlay Arrows" (core/preview_utils.py:74-82): the left eye with a green cv2.arrowedLine((x, y) -> (x + int(10 shift), y), tipLength 0.3)
 * at every 20th pixel of every 20th row where |int(10 shift)| > 1.  left_bgr != out_bgr. */
int vd3d_preview_arrows(vd3d_ctx* ctx, const uint8_t* left_bgr, const float* shift_map, int h, int w, uint8_t* out_bgr);

/* ---- stage entry points (the pieces B2 is made of; exported for tests / profiling / sharded runner) */
/* apply_dof_cuda + apply_color_grade + tensor_to_frame + side bars + apply_sharpening + fit + mux
 * (core/render_3d.py:1340-1419) on two u8 eyes. depth_norm is the eye-res normalised depth. */
int vd3d_finish_frame(vd3d_ctx* ctx, const uint8_t* left_bgr, const uint8_t* right_bgr,
                      const float* depth_norm, int eye_h, int eye_w,
                      const vd3d_render_params* p, double focal_depth, int bar_width, int bar_side,
                      uint8_t* out_bgr);
/* exact order statistics on a device plane: torch.quantile semantics (float32 rank, two-branch lerp) */
int vd3d_quantiles(vd3d_ctx* ctx, const float* plane, int64_t n, const float* q_host, int nq, float* out_host);
/* estimate_subject_depth, core/render_3d.py:145-172 (synchronises; test/diagnostic entry) */
int vd3d_subject_depth(vd3d_ctx* ctx, const float* plane, int H, int W, float* out_host);
/* detect_black_bars, core/render_3d.py:293-316, on a device-resident uint8 BGR frame (synchronises; test/diagnostic
 * entry -- vd3d_render_frame runs the same kernel asynchronously when auto_crop_black_bars is set) */
int vd3d_detect_black_bars(vd3d_ctx* ctx, const uint8_t* frame_bgr, int h, int w, int* top_host, int* bottom_host);
/* device-to-device streaming copy used as the measured-peak yardstick for roofline.frac */
int vd3d_stream_copy(vd3d_ctx* ctx, const void* src, void* dst, size_t bytes);
/* The three transcendental operators of the path, elementwise on a device float32 array, with the VALUES torch's CPU kernels give
 * (which is what the reference computes with): op 0 = torch.pow(x, param) for x >= 0 (_signed_pow core/render_3d.py:517, the layer
 * weight :620; SLEEF Sleef_powf_u10 and ATen's special exponents), op 1 = torch.sigmoid(x) (:209; 1 / (1 + Sleef_expf_u10(0 - x))),
 * op 2 = torch.sqrt(x) (:206, :349, :440; MKL VML vsSqrt -- one ULP low on 0.6 % of the inputs).  The kernels of the chain call the
 * same device functions; this entry exists so that they can be compared against the oracle / torch on arbitrary inputs. */
int vd3d_torch_math(vd3d_ctx* ctx, int op, const float* x, float param, float* out, long long n);
/* Ops 0 and 1 of the above as a torch process with `aten_threads` intra-op threads computes them on a contiguous n-element float32 CPU tensor: the SLEEF
 * values, and on the scalar tails of every thread's chunk (vd3d_shift_params::aten_threads) libm's -- (float) pow((double) x, param) with the unrounded
 * Python exponent, 1 / (1 + expf(-x)) with glibc's expf.  aten_threads 0: no tails; < 0: every element takes the tail arithmetic (diagnostic).  The shaping and
 * shift kernels call the same device functions.  n < 2^32. */
int vd3d_torch_math_aten(vd3d_ctx* ctx, int op, const float* x, double param, float* out, long long n, int aten_threads);
/* HIP-event profiling of the stages on the ctx stream ("frame", "ingest", "select_eye", "select_dc", "shape",
 * "select_s1", "warp" (= "shift" + "w1", the fused warp kernel alone), "finish", "handoff", "advance", "pixel_shift", "stream_copy").  vd3d_last_stage_ms = average ms per call since
 * profiling was enabled (-1 if never seen); both getters synchronise. */
int vd3d_set_profiling(vd3d_ctx* ctx, int enable);
/* Route selectors for parity tests and, in development libraries only, launch-policy knobs.  Results NEVER depend on any of them (tested).
 * Product library: which = 3 -- routes of the finishing stage (bit 0: fused kernel in front of a fit it does not take, bit 1: its epilogue as the sharpen / fit /
 * mux kernel behind the unfused DOF kernels; default 3) and which = 4 -- 1: feather_strength <= 0 still runs the mask / window-sum / blend kernels instead of the
 * exact no-feather warp (default 0).  Process-wide plain switches: set them while no frame is in flight.  Every other knob (0 - 2, 5 - 9: tile heights and
 * orders, chain grouping, the parked persistent kernels) returns VD3D_E_UNSUPPORTED unless the library was built with -DVD3D_DEV_KNOBS
 * (bash tools/build_ab.sh dev -DVD3D_DEV_KNOBS); which = -1 queries that (0 = development knobs present).  The VD3D_TUNE environment variable of the Python
 * loader is honoured by development libraries only. */
int vd3d_debug_tune(int which, int value);
float vd3d_last_stage_ms(vd3d_ctx* ctx, const char* stage);
long vd3d_stage_calls(vd3d_ctx* ctx, const char* stage);
/* host-only (works without a GPU): the float32 Gaussian weights the DOF kernels receive for one blur level -- torchvision's _get_gaussian_kernel1d as torch
 * evaluates it on the CPU, incl. torch.sum's summation order (core/render_3d.py:798-806); odd k <= 31.  Tests compare it with torch and the oracle. */
int vd3d_debug_gaussian_kernel1d(int k, float sigma, float* out_host);
/* host-only: the library's restatement of torch.exp on a float32 CPU tensor (oneMKL vsExp, not the rounded exponential) that the DOF weights use */
int vd3d_debug_exp_torch(const float* x_host, float* out_host, long long n);
/* internal planes of the last call (device pointers owned by ctx; tests only).  S: the render / step path without feathering computes the shift values inside the
 * warp kernel and does not write this plane (round 6); vd3d_pixel_shift always does. */
int vd3d_debug_planes(vd3d_ctx* ctx, float** D, float** S, uint8_t** L, uint8_t** R, float** rgb_eye, float** dn_cur);

#ifdef __cplusplus
}
#endif
#endif /* VD3D_H */
