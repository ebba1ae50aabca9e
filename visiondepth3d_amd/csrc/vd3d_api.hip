// vd3d_api.hip -- C ABI of libvd3d_hip.so (include/vd3d.h).  Host-side orchestration only:
// every entry point enqueues kernels on the ctx stream and returns; no host synchronisation
// except where the header says so.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "vd3d_kernels.h"

#define VD3D_EXPORT extern "C" __attribute__((visibility("default")))

static thread_local char g_err[512] = "";
static int set_err(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}
#define HIPCHK(x)                                                                                     \
  do {                                                                                                \
    hipError_t e_ = (x);                                                                              \
    if (e_ != hipSuccess) return set_err(VD3D_E_HIP, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

struct vd_prof_rec { std::string name; hipEvent_t a, b; };

struct vd3d_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false, own_stream_escaped = false;
  vd_dev_work* work = nullptr;
  uint32_t* histA = nullptr;  // [VD_NJOBS][VD_NB_A] followed by histB [VD_NJOBS][VD_MAX_T][VD_NB_B]
  uint32_t* histB = nullptr;
  size_t hist_bytes = 0;
  // eye-res planes
  int eye_h = 0, eye_w = 0;
  float* rgb_eye = nullptr; float* tdf = nullptr; float* dn[2] = {nullptr, nullptr};
  int dn_cur = 0;
  // warp-res planes
  int H = 0, W = 0;
  float *dc = nullptr;   // curved depth plane (fused chain)
  float *D = nullptr, *S = nullptr, *e2L = nullptr, *e2R = nullptr, *bL = nullptr, *bR = nullptr;
  float *E2 = nullptr;   // [H][W][2]: gradient mask of both eyes, k_e2w -> W1
  uint8_t *L = nullptr, *R = nullptr, *gL = nullptr, *gR = nullptr;
  uint8_t* gLR = nullptr; size_t gLR_cap = 0;   // [H][2W][3] sharpened eyes side by side (E1 in front of a fit it does not take)
  uint32_t* mm = nullptr; int mm_cap = 0;
  uint32_t* rowflag = nullptr; int rowflag_cap = 0;   // k_autocrop: one flag per source row
  uint8_t* blank_eye = nullptr; size_t blank_cap = 0; // skip_blank_frames: the side-masked source frame (source size)
  // vd3d_render_params::aten_sum_threads > 0: piece plan of the two torch.mean sums (vd3d_atensum.hip) for the current eye size / thread count, and the
  // per-frame piece sums of a batch.  Plans are never overwritten (a queued kernel may still read one): a new key gets new buffers.
  int aten_eh = 0, aten_ew = 0, aten_T = 0, aten_n_small = 0, aten_n_big = 0, aten_nr_crop = 0, aten_nr_mad = 0;
  int* aten_plan = nullptr; float* aten_scratch = nullptr;
  struct vd_aten_entry { int eh, ew, T, n_small, n_big, nr_crop, nr_mad; int* plan; float* scratch; };
  std::vector<vd_aten_entry> aten_cache;   // plans of other (eye size, thread count) keys this context has used (re-used, bounded at 8)
  // the N-thread ATen mode (round 5): planes resized by ATen's premultiplied-weight kernel ahead of W1 (RGB, [3][H][W]) and of the finishing kernels (depth, [H][W])
  float* pm_rgb = nullptr; size_t pm_rgb_cap = 0;
  float* pm_dd = nullptr; size_t pm_dd_cap = 0;
  uint8_t* fmt_eyes = nullptr; size_t fmt_cap = 0;    // vd3d_format_3d_output's own pair of resized VR eyes (never the shared warp-res planes: ADVICE r4)
  bool crop_scalars_dirty = false;                    // fs.crop_top/bottom hold a previous auto-crop result
  // frame sharding (three-phase protocol): per-slot planes of the frames this rank owns inside the current step
  int n_slots = 0, slot_eh = 0, slot_ew = 0, slot_H = 0, slot_W = 0;
  std::vector<float*> slot_rgb, slot_dn, slot_D;
  std::vector<float*> slot_tdf, slot_tdfp;   // measure/replay protocol: filtered plane of the own frame and of the frame before it
  std::vector<const float*> slot_prev;       // where P1 left "the filtered plane of the frame before slot's frame" (slot_tdfp[slot], or the previous slot's slot_tdf inside a batch)
  // batched steps: per frame of a batch (<= VD_MAX_BATCH) its own histogram arena and curved-depth plane, so that the frames' chain
  // kernels run side by side in one launch
  int n_batch = 0; uint32_t* bhist = nullptr; std::vector<float*> bdc;
  float* etab = nullptr;                     // [VD_MAX_STEP + 1][VD_ETAB] replayed normalisation table of the current step
  int* crop_tab = nullptr;                   // [VD_MAX_STEP][4] exchanged auto-crop rectangles of the current step
  bool crop_tab_set = false;
  vd_dev_work* slot_work = nullptr;
  // overlapped pixel passes (vd3d_set_pixel_overlap): vd3d_shard_pixels runs on pix_stream behind the measurement chain of the
  // NEXT step, which stays on `stream`; slot_done[slot] guards the slot's planes against being overwritten too early
  hipStream_t pix_stream = nullptr; bool pix_overlap = false; bool pix_pending = false; bool pix_stream_escaped = false;
  hipEvent_t ev_chain = nullptr, ev_pix_last = nullptr;
  // more than one pixel stream (vd3d_set_pixel_overlap(ctx, n), n = 2 .. VD_MAX_PIX): consecutive pixel passes go round-robin over n
  // streams, each with its own shift plane and warped eyes, so the kernels of neighbouring frames (k_shift / W1 / E1: different LDS, VGPR and
  // latency profiles) share the CUs and the tail of one launch is covered by the next frame's kernels.  Stream 0 = pix_stream with the
  // context's S / L / R.  A pass that needs the context's SHARED fallback planes (unfused warp / finish) first waits for the other
  // streams and is waited for by every later pass (pix_excl_*).
#define VD_MAX_PIX 4
  int n_pix = 1; unsigned pix_rr = 0; int cur_pix = -1; bool cur_excl = false;
  hipStream_t pix_x[VD_MAX_PIX - 1] = {nullptr, nullptr, nullptr}; bool pix_x_escaped[VD_MAX_PIX - 1] = {false, false, false};
  hipEvent_t ev_pix_last_x[VD_MAX_PIX - 1] = {nullptr, nullptr, nullptr}; bool pix_pending_x[VD_MAX_PIX - 1] = {false, false, false};
  float* E2_x[VD_MAX_PIX - 1] = {nullptr, nullptr, nullptr};
  float* S_x[VD_MAX_PIX - 1] = {nullptr, nullptr, nullptr}; uint8_t* L_x[VD_MAX_PIX - 1] = {nullptr, nullptr, nullptr}; uint8_t* R_x[VD_MAX_PIX - 1] = {nullptr, nullptr, nullptr};
  int x_H = 0, x_W = 0;
  hipEvent_t ev_excl = nullptr; bool excl_set = false;
  std::vector<hipEvent_t> slot_done; std::vector<char> slot_busy;
  // dense DOF weight table of the fused finishing kernel (vd_finish_consts::w2) in device memory + the host copy it was uploaded from
  // Tables are never overwritten (a finish kernel of an earlier dof_strength may still be queued on any of the context's streams): every new
  // parameter set gets its own 1.3 KB device table, filled before anything can read it
  struct w2_tab { float host[4][81]; float* dev; };
  std::vector<w2_tab> w2_tabs; int w2_cur = -1;
  // ... and the general table of the unfused dense kernel (k_dof_grade4: any odd tap count up to 31), same never-overwrite rule
  struct wk_tab { int ksz[4]; float kern[4][32]; float* dev; };
  std::vector<wk_tab> wk_tabs; int wk_cur = -1;
  // profiling
  bool profiling = false;
  std::vector<vd_prof_rec> recs;
  std::vector<hipEvent_t> ev_pool;
  std::map<std::string, std::pair<double, long>> acc;
};

struct StageTimer {
  vd3d_ctx* c; vd_prof_rec r; bool on;
  StageTimer(vd3d_ctx* ctx, const char* name) : c(ctx), on(ctx->profiling) {
    if (!on) return;
    r.name = name;
    auto get = [&]() { hipEvent_t e; if (!c->ev_pool.empty()) { e = c->ev_pool.back(); c->ev_pool.pop_back(); } else (void)hipEventCreate(&e); return e; };
    r.a = get(); r.b = get();
    (void)hipEventRecord(r.a, c->stream);
  }
  ~StageTimer() {
    if (!on) return;
    (void)hipEventRecord(r.b, c->stream);
    c->recs.push_back(r);
  }
};

static void prof_collect(vd3d_ctx* c) {
  for (auto& r : c->recs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      auto& a = c->acc[r.name];
      a.first += ms; a.second += 1;
    }
    c->ev_pool.push_back(r.a); c->ev_pool.push_back(r.b);
  }
  c->recs.clear();
}

// main stream waits until the overlapped pixel pass that still reads `slot` has finished (no-op without overlap)
static int wait_slot(vd3d_ctx* c, int slot) {
  if (c->pix_overlap && slot >= 0 && slot < (int)c->slot_busy.size() && c->slot_busy[slot]) {
    HIPCHK(hipStreamWaitEvent(c->stream, c->slot_done[slot], 0));
    c->slot_busy[slot] = 0;
  }
  return 0;
}
// main stream waits for every outstanding overlapped pixel pass (they share L / R / S and write the caller's outputs)
static int join_pixels(vd3d_ctx* c) {
  bool any = c->pix_pending;
  if (c->pix_pending) {
    HIPCHK(hipStreamWaitEvent(c->stream, c->ev_pix_last, 0));
    c->pix_pending = false;
  }
  for (int k = 0; k < VD_MAX_PIX - 1; ++k)
    if (c->pix_pending_x[k]) {
      HIPCHK(hipStreamWaitEvent(c->stream, c->ev_pix_last_x[k], 0));
      c->pix_pending_x[k] = false; any = true;
    }
  if (any) std::fill(c->slot_busy.begin(), c->slot_busy.end(), 0);
  c->excl_set = false;
  return 0;
}
// A pixel pass on one of several pixel streams is about to touch the context's shared fallback planes (e2 / b / graded eyes): order it
// behind the passes already enqueued on the other pixel streams; the caller records ev_excl afterwards, which every later pass waits for.
static int pix_exclusive(vd3d_ctx* c) {
  if (c->cur_pix < 0 || c->n_pix < 2 || c->cur_excl) return 0;
  if (c->cur_pix != 0 && c->pix_pending) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_pix_last, 0));
  for (int k = 0; k < VD_MAX_PIX - 1; ++k)
    if (k + 1 != c->cur_pix && c->pix_pending_x[k]) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_pix_last_x[k], 0));
  c->cur_excl = true;
  return 0;
}

template <class T> static hipError_t re_alloc(T** p, size_t n) {
  if (*p) { hipError_t e = hipFree(*p); *p = nullptr; if (e != hipSuccess) return e; }
  return n ? hipMalloc((void**)p, n * sizeof(T)) : hipSuccess;
}

static int ensure_eye(vd3d_ctx* c, int eh, int ew) {
  if (c->eye_h == eh && c->eye_w == ew) return 0;
  size_t ne = (size_t)eh * ew;
  HIPCHK(re_alloc(&c->rgb_eye, 3 * ne));
  HIPCHK(re_alloc(&c->tdf, ne));
  HIPCHK(re_alloc(&c->dn[0], ne));
  HIPCHK(re_alloc(&c->dn[1], ne));
  if (c->eye_h != 0) {   // the planes were reallocated: TemporalDepthFilter.prev_depth / prev_depth_tensor no longer exist
    const int32_t zero2[2] = {0, 0};
    static_assert(offsetof(vd3d_state, prev_depth_valid) == offsetof(vd3d_state, tdf_valid) + sizeof(int32_t), "adjacent flags");
    HIPCHK(hipMemcpyAsync(&c->work->st.tdf_valid, zero2, sizeof zero2, hipMemcpyHostToDevice, c->stream));
  }
  c->eye_h = eh; c->eye_w = ew; c->dn_cur = 0;
  return 0;
}
static int ensure_work(vd3d_ctx* c, int H, int W) {
  if (c->H == H && c->W == W) return 0;
  size_t n = (size_t)H * W;
  HIPCHK(re_alloc(&c->D, n)); HIPCHK(re_alloc(&c->S, n)); HIPCHK(re_alloc(&c->dc, n));
  HIPCHK(re_alloc(&c->e2L, n)); HIPCHK(re_alloc(&c->e2R, n)); HIPCHK(re_alloc(&c->E2, 2 * n));
  HIPCHK(re_alloc(&c->bL, n)); HIPCHK(re_alloc(&c->bR, n));
  HIPCHK(re_alloc(&c->L, 3 * n)); HIPCHK(re_alloc(&c->R, 3 * n));
  HIPCHK(re_alloc(&c->gL, 3 * n)); HIPCHK(re_alloc(&c->gR, 3 * n));
  c->H = H; c->W = W;
  return 0;
}

// ---------------------------------------------------------------------------------------------
VD3D_EXPORT int vd3d_abi_version(void) { return VD3D_ABI_VERSION; }
VD3D_EXPORT const char* vd3d_last_error(void) { return g_err; }

VD3D_EXPORT void vd3d_shift_params_default(vd3d_shift_params* p) {
  memset(p, 0, sizeof *p);
  p->blur_ksize = 9; p->use_subject_tracking = 1; p->enable_floating_window = 1; p->enable_feathering = 1;
  p->enable_edge_masking = 1; p->enable_dynamic_convergence = 1; p->feather_strength = 10.0;
  p->max_pixel_shift_percent = 0.02; p->parallax_balance = 0.8; p->zero_parallax_strength = 0.0;
  p->convergence_strength = 0.0; p->depth_pop_gamma = 0.85; p->depth_pop_mid = 0.50; p->depth_stretch_lo = 0.05;
  p->depth_stretch_hi = 0.95; p->fg_pop_multiplier = 1.20; p->bg_push_multiplier = 1.10; p->subject_lock_strength = 1.00;
  p->aten_threads = 0; p->reserved0 = 0;
}
VD3D_EXPORT void vd3d_render_params_default(vd3d_render_params* p) {
  memset(p, 0, sizeof *p);
  vd3d_shift_params_default(&p->shift);
  // render_sbs_3d's own keyword defaults (core/render_3d.py:949-984)
  p->shift.feather_strength = 0.0; p->shift.blur_ksize = 1; p->shift.use_subject_tracking = 0; p->shift.enable_floating_window = 0;
  p->ipd_factor = 1.0; p->dof_strength = 0.0; p->sharpness_factor = 0.0;
  p->color_saturation = 1.0; p->color_contrast = 1.0; p->color_brightness = 0.0;
  p->format = VD3D_FMT_HALF_SBS;
  p->dof_dense_conv = 1;   // the reference's convolution order (parity mode) is the default; 0 = separable levels
}

VD3D_EXPORT int vd3d_ctx_create(int device, void* stream, vd3d_ctx** out) {
  if (!out) return set_err(VD3D_E_INVALID, "out is NULL");
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return set_err(VD3D_E_INVALID, "device %d out of range (%d visible)", device, ndev);
  HIPCHK(hipSetDevice(device));
  vd3d_ctx* c = new vd3d_ctx();
  c->device = device;
  if (stream == VD3D_STREAM_PRIVATE) { HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
  else c->stream = (hipStream_t)stream;  // NULL = the default stream
  HIPCHK(hipMalloc((void**)&c->work, sizeof(vd_dev_work)));
  HIPCHK(hipMemsetAsync(c->work, 0, sizeof(vd_dev_work), c->stream));
  const size_t nA = (size_t)VD_NJOBS * VD_NB_A, nB = (size_t)VD_NJOBS * VD_MAX_T * (VD_NB_B + VD_NB_BC);
  c->hist_bytes = (nA + nB) * sizeof(uint32_t);
  HIPCHK(hipMalloc((void**)&c->histA, c->hist_bytes));
  c->histB = c->histA + nA;
  *out = c;
  return 0;
}
VD3D_EXPORT int vd3d_ctx_destroy(vd3d_ctx* c) {
  if (!c) return 0;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  prof_collect(c);
  for (auto e : c->ev_pool) (void)hipEventDestroy(e);
  void* ptrs[] = {c->work, c->histA, c->rgb_eye, c->tdf, c->dn[0], c->dn[1], c->D, c->S, c->e2L, c->e2R, c->E2, c->bL, c->bR, c->L, c->R, c->gL, c->gR, c->gLR, c->mm, c->dc, c->rowflag, c->etab, c->crop_tab, c->blank_eye, c->fmt_eyes, c->aten_plan, c->aten_scratch, c->pm_rgb, c->pm_dd};
  for (const auto& e : c->aten_cache) { (void)hipFree(e.plan); (void)hipFree(e.scratch); }
  for (auto& t : c->w2_tabs) (void)hipFree(t.dev);
  for (auto& t : c->wk_tabs) (void)hipFree(t.dev);
  for (void* p : ptrs) if (p) (void)hipFree(p);
  for (auto* v : {&c->slot_rgb, &c->slot_dn, &c->slot_D, &c->slot_tdf, &c->slot_tdfp})
    for (float* q : *v) (void)hipFree(q);
  if (c->slot_work) (void)hipFree(c->slot_work);
  for (float* q : c->bdc) (void)hipFree(q);
  if (c->bhist) (void)hipFree(c->bhist);
  for (int k = 0; k < VD_MAX_PIX - 1; ++k) {
    if (c->pix_x[k]) { (void)hipStreamSynchronize(c->pix_x[k]); if (!c->pix_x_escaped[k]) (void)hipStreamDestroy(c->pix_x[k]); }
    if (c->ev_pix_last_x[k]) (void)hipEventDestroy(c->ev_pix_last_x[k]);
    if (c->S_x[k]) (void)hipFree(c->S_x[k]);
    if (c->E2_x[k]) (void)hipFree(c->E2_x[k]);
    if (c->L_x[k]) (void)hipFree(c->L_x[k]);
    if (c->R_x[k]) (void)hipFree(c->R_x[k]);
  }
  if (c->ev_excl) (void)hipEventDestroy(c->ev_excl);
  if (c->pix_stream) { (void)hipStreamSynchronize(c->pix_stream); if (!c->pix_stream_escaped) (void)hipStreamDestroy(c->pix_stream); }
  for (auto e : c->slot_done) (void)hipEventDestroy(e);
  if (c->ev_chain) (void)hipEventDestroy(c->ev_chain);
  if (c->ev_pix_last) (void)hipEventDestroy(c->ev_pix_last);
  if (c->own_stream && !c->own_stream_escaped) (void)hipStreamDestroy(c->stream);
  delete c;
  return 0;
}
VD3D_EXPORT int vd3d_sync(vd3d_ctx* c) {
  if (c->pix_stream) HIPCHK(hipStreamSynchronize(c->pix_stream));
  for (int k = 0; k < VD_MAX_PIX - 1; ++k) if (c->pix_x[k]) HIPCHK(hipStreamSynchronize(c->pix_x[k]));
  HIPCHK(hipStreamSynchronize(c->stream));
  prof_collect(c);
  return 0;
}
VD3D_EXPORT void* vd3d_ctx_stream(vd3d_ctx* c) {
  if (c && c->own_stream) c->own_stream_escaped = true;   // see vd3d_ctx_pixel_stream
  return c ? (void*)c->stream : nullptr;
}
// Re-target a context that enqueues on a caller-owned stream (e.g. whatever stream is current in PyTorch at call time).  Work already
// enqueued stays on the old stream; the new stream is ordered behind it with an event, so the context's planes are never raced.
VD3D_EXPORT int vd3d_ctx_set_stream(vd3d_ctx* c, void* stream) {
  if (!c) return set_err(VD3D_E_INVALID, "NULL context");
  if (c->own_stream) return set_err(VD3D_E_INVALID, "the context owns a private stream (VD3D_STREAM_PRIVATE)");
  if ((hipStream_t)stream == c->stream) return 0;
  HIPCHK(hipSetDevice(c->device));
  int rc = join_pixels(c);
  if (rc) return rc;
  hipEvent_t e;
  HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  HIPCHK(hipEventRecord(e, c->stream));
  HIPCHK(hipStreamWaitEvent((hipStream_t)stream, e, 0));
  HIPCHK(hipEventDestroy(e));
  c->stream = (hipStream_t)stream;
  return 0;
}
// The handle escapes to the caller (PyTorch wraps it and may record events on it when tensors marked with record_stream() are freed, possibly
// after the context is gone): from then on the stream is never destroyed (vd3d_ctx_destroy leaves it to the process).
VD3D_EXPORT void* vd3d_ctx_pixel_stream(vd3d_ctx* c) {
  if (!c) return nullptr;
  if (c->pix_stream) c->pix_stream_escaped = true;
  return (void*)c->pix_stream;
}
// pixel stream k of vd3d_set_pixel_overlap(ctx, n) (k = 0: vd3d_ctx_pixel_stream); NULL beyond the streams created so far
VD3D_EXPORT void* vd3d_ctx_pixel_stream_k(vd3d_ctx* c, int k) {
  if (!c || k < 0 || k >= VD_MAX_PIX) return nullptr;
  if (k == 0) return vd3d_ctx_pixel_stream(c);
  if (c->pix_x[k - 1]) c->pix_x_escaped[k - 1] = true;
  return (void*)c->pix_x[k - 1];
}

// ---- state ------------------------------------------------------------------------------------
VD3D_EXPORT int vd3d_state_export(vd3d_ctx* c, vd3d_state* out) {
  HIPCHK(hipMemcpyAsync(out, &c->work->st, sizeof(vd3d_state), hipMemcpyDeviceToHost, c->stream));
  return vd3d_sync(c);
}
VD3D_EXPORT int vd3d_state_import(vd3d_ctx* c, const vd3d_state* in) {
  HIPCHK(hipMemcpyAsync(&c->work->st, in, sizeof(vd3d_state), hipMemcpyHostToDevice, c->stream));
  return vd3d_sync(c);
}
VD3D_EXPORT int vd3d_state_reset(vd3d_ctx* c) {
  vd3d_state z;
  memset(&z, 0, sizeof z);
  return vd3d_state_import(c, &z);
}
VD3D_EXPORT int vd3d_state_new_clip(vd3d_ctx* c) {  // what render_sbs_3d re-creates per call (:1174-1182)
  vd3d_state s;
  int rc = vd3d_state_export(c, &s);
  if (rc) return rc;
  s.smooth_valid = 0; s.tdf_valid = 0; s.prev_depth_valid = 0; s.focal_valid = 0;
  return vd3d_state_import(c, &s);
}
VD3D_EXPORT int vd3d_state_planes(vd3d_ctx* c, float** tdf_prev, float** norm_prev, int* eye_h, int* eye_w) {
  if (tdf_prev) *tdf_prev = c->tdf;
  if (norm_prev) *norm_prev = c->dn[c->dn_cur ^ 1];  // the plane the NEXT frame treats as d_{t-1}
  if (eye_h) *eye_h = c->eye_h;
  if (eye_w) *eye_w = c->eye_w;
  return 0;
}
VD3D_EXPORT int vd3d_last_scalars(vd3d_ctx* c, vd3d_frame_scalars* out) {
  HIPCHK(hipMemcpyAsync(out, &c->work->fs, sizeof(vd3d_frame_scalars), hipMemcpyDeviceToHost, c->stream));
  return vd3d_sync(c);
}

// the current plan goes back into the context's cache / a cached plan becomes current
static void aten_cache_put(vd3d_ctx* c) {
  if (!c->aten_plan) return;
  c->aten_cache.push_back(vd3d_ctx::vd_aten_entry{c->aten_eh, c->aten_ew, c->aten_T, c->aten_n_small, c->aten_n_big, c->aten_nr_crop, c->aten_nr_mad, c->aten_plan, c->aten_scratch});
  c->aten_plan = nullptr; c->aten_scratch = nullptr; c->aten_eh = c->aten_ew = c->aten_T = -1;
}
static bool aten_cache_get(vd3d_ctx* c, int eh, int ew, int T) {
  for (size_t i = 0; i < c->aten_cache.size(); ++i) {
    const vd3d_ctx::vd_aten_entry e = c->aten_cache[i];
    if (e.eh != eh || e.ew != ew || e.T != T) continue;
    c->aten_cache.erase(c->aten_cache.begin() + (long)i);
    c->aten_eh = eh; c->aten_ew = ew; c->aten_T = T;
    c->aten_n_small = e.n_small; c->aten_n_big = e.n_big; c->aten_nr_crop = e.nr_crop; c->aten_nr_mad = e.nr_mad;
    c->aten_plan = e.plan; c->aten_scratch = e.scratch;
    return true;
  }
  return false;
}

// vd3d_render_params::aten_sum_threads > 0: the two torch.mean calls of the loop body in ATen's float32 summation order (vd3d_atensum.hip)
static int aten_setup(vd3d_ctx* c, const vd3d_render_params* p, vd_stage_args* a) {
  a->aten_threads = 0;
  const int T = p->aten_sum_threads;
  if (T <= 0) return 0;
  if (c->aten_eh != p->eye_h || c->aten_ew != p->eye_w || c->aten_T != T) {
    // a small per-key cache: a context that alternates between eye sizes / thread counts re-uses its plans instead of allocating a new pair
    // per change; a replaced key's buffers stay valid (queued launches may still read them) and are handed back out when the key returns
    aten_cache_put(c);
    if (!aten_cache_get(c, p->eye_h, p->eye_w, T)) {
      std::vector<int> flat;
      int ns = 0, nb = 0, nrc = 0, nrm = 0;
      if (!vd_aten_plan_build(p->eye_h, p->eye_w, T, flat, &ns, &nb, &nrc, &nrm))
        return set_err(VD3D_E_UNSUPPORTED, "aten_sum_threads %d with %dx%d eyes: outside the restated range (1 .. 1024 threads)", T, p->eye_w, p->eye_h);
      if (c->aten_cache.size() >= 8) {   // bounded: drain the stream once, then free the oldest entry
        HIPCHK(hipStreamSynchronize(c->stream));
        if (c->pix_stream) HIPCHK(hipStreamSynchronize(c->pix_stream));
        for (int k = 0; k < VD_MAX_PIX - 1; ++k) if (c->pix_x[k]) HIPCHK(hipStreamSynchronize(c->pix_x[k]));
        (void)hipFree(c->aten_cache.front().plan); (void)hipFree(c->aten_cache.front().scratch);
        c->aten_cache.erase(c->aten_cache.begin());
      }
      c->aten_plan = nullptr; c->aten_scratch = nullptr;
      HIPCHK(hipMalloc((void**)&c->aten_plan, flat.size() * sizeof(int)));
      HIPCHK(hipMemcpy(c->aten_plan, flat.data(), flat.size() * sizeof(int), hipMemcpyHostToDevice));   // a fresh buffer: nothing queued reads it yet
      HIPCHK(hipMalloc((void**)&c->aten_scratch, (size_t)VD_MAX_BATCH * (size_t)(ns + nb) * sizeof(float)));
      c->aten_eh = p->eye_h; c->aten_ew = p->eye_w; c->aten_T = T;
      c->aten_n_small = ns; c->aten_n_big = nb; c->aten_nr_crop = nrc; c->aten_nr_mad = nrm;
    }
  }
  a->aten_threads = T; a->aten_n_small = c->aten_n_small; a->aten_n_big = c->aten_n_big; a->aten_nr_crop = c->aten_nr_crop; a->aten_nr_mad = c->aten_nr_mad;
  a->aten_plan = c->aten_plan; a->aten_scratch = c->aten_scratch;
  return 0;
}

// ---- shared middle section: shaped depth, s1, shift, feather, warp --------------------------------
static int check_shift_params(const vd3d_shift_params* p, int H, int W) {
  if (W < 2 || H < 2) return set_err(VD3D_E_INVALID, "warp size %dx%d too small", W, H);
  if ((long long)H * W >= (1ll << 31)) return set_err(VD3D_E_INVALID, "warp size too large");
  if (p->blur_ksize < 1) return set_err(VD3D_E_INVALID, "blur_ksize must be >= 1 (avg_pool2d raises)");
  if (p->blur_ksize > PL_KMAX_HOST) return set_err(VD3D_E_UNSUPPORTED, "blur_ksize %d > %d not built", p->blur_ksize, PL_KMAX_HOST);
  return 0;
}

// The shift parameters as the WARP stage sees them.  feather_shift_edges (core/render_3d.py:328-374) with feather_strength <= 0 is an exact no-op:
// edge_mask = clamp(grad * fs, 0, 1) = 0 everywhere (the gradient magnitude is a finite non-negative number), its k x k average is 0, and
// shifted * (1 - 0) + original * 0 = shifted (the samples are non-negative, so no signed zero survives; the final clamp(0, 1) only touches values the
// uint8 conversion of tensor_to_frame maps to the same byte) -- so the mask kernel (k_e2w), the window sums and the blend are not run at all.  This is
// the GUI's own default configuration (VisionDepth3D.py:1405-1453: feather_strength 0.0, blur_ksize 1).  k_shift keeps the caller's feather_strength:
// suppress_artifacts_with_edge_mask uses it on its own (:204-216).  vd3d_debug_tune(4, 1) keeps the long way (A/B, tests).
static int g_feather0_long = 0;
static vd3d_shift_params warp_stage_params(const vd3d_shift_params& sp) {
  vd3d_shift_params w = sp;
  if (w.enable_feathering && w.feather_strength <= 0.0 && !g_feather0_long) w.enable_feathering = 0;
  return w;
}

// the context's own planes and control block as a batch of one frame (sequential entry points)
static vd_batch batch_of_one(vd3d_ctx* c) {
  vd_batch b;
  memset(&b, 0, sizeof b);
  b.n = 1; b.w_main = c->work;
  vd_batch_frame& F = b.f[0];
  F.w = c->work; F.histA = c->histA; F.histB = c->histB;
  F.rgb_eye = c->rgb_eye; F.tdf = c->tdf; F.tdf_prev = c->tdf;   // the plane EMA updates c->tdf in place
  F.dc = c->dc; F.D = c->D;
  return b;
}

// N-thread ATen mode: a resize ATen runs with its premultiplied-weight kernel (vd_interp_premult) is done here, into a scratch plane of the context, and the
// consumer (W1, the finishing kernels) is then called with identity geometry.  The scratch is shared by the context's pixel streams: callers on one of them go through
// pix_exclusive first.
static int premult_resize(vd3d_ctx* c, int C, const float** plane, int* ih, int* iw, int H, int W, int aten_threads, float** scratch, size_t* cap) {
  if ((*ih == H && *iw == W) || !vd_interp_premult(C, H, W, aten_threads)) return 0;
  const size_t need = (size_t)C * H * W;
  if (*cap < need) { HIPCHK(hipDeviceSynchronize()); HIPCHK(re_alloc(scratch, need)); *cap = need; }
  vd_launch_interp_planes(c->stream, *plane, C, *ih, *iw, *scratch, H, W, 1);
  *plane = *scratch; *ih = H; *iw = W;
  return 0;
}

static int run_shift_and_warp(vd3d_ctx* c, const float* rgb, const float* depth_plane, int ih, int iw, int W, int H,
                              const vd3d_shift_params& sp, vd_stage_args a, bool skip_pixels = false) {
  hipStream_t s = c->stream;
  {
    StageTimer t(c, "select_dc");   // fused chain: norm + stage1 (A1), b1 (B1), shape (+A2), b2 (B2)
    vd_batch b = batch_of_one(c);   // render path: depth_plane = dn_cur (written by K3a from c->tdf); bare pixel_shift_cuda: the caller's plane
    b.f[0].dn = const_cast<float*>(depth_plane);
    b.f[0].dn_prev = a.have_eye ? c->dn[c->dn_cur ^ 1] : nullptr;
    vd_launch_chain_work(s, b, a.have_eye, ih, iw, H, W, (float)sp.depth_pop_mid, (float)sp.depth_pop_gamma, a);
  }
  if (!skip_pixels) { StageTimer t(c, "warp");
    { int rcp = premult_resize(c, 3, &rgb, &ih, &iw, H, W, sp.aten_threads, &c->pm_rgb, &c->pm_rgb_cap); if (rcp) return rcp; }
    const vd3d_shift_params spw = warp_stage_params(sp);   // feather_strength <= 0: the exact no-feather kernels
    const bool pre = spw.enable_feathering && vd_warp_fused_ok(ih, iw, H, W, spw);   // k_e2w -> W1 (see shard_pixels_impl)
    // without feathering W1 computes the shift values of its own tile (round 6); this entry point keeps the plane (its callers can ask for it)
    const bool fold = vd_warp_fold_ok(ih, iw, H, W, spw, sp);
    const vd_shift_fold ff = {c->work, sp, c->S};
    if (!fold) { StageTimer t1(c, "shift"); vd_launch_shift(s, c->D, H, W, c->work, sp, c->S); }
    bool fused = false;
    { StageTimer t2(c, "w1");
      if (pre) { StageTimer t3(c, "e2w"); vd_launch_e2w(s, c->D, c->S, H, W, (float)spw.feather_strength, c->E2); }
      fused = vd_launch_warp_fused(s, rgb, ih, iw, c->D, c->S, H, W, spw, c->L, c->R, pre ? c->E2 : nullptr, fold ? &ff : nullptr); }
    if (!fused && fold) { StageTimer t1(c, "shift"); vd_launch_shift(s, c->D, H, W, c->work, sp, c->S); }
    if (!fused) {   // blur sizes / frame sizes the fused kernel refuses (its LDS tile would not fit): one stage per kernel
    if (spw.enable_feathering) {
      vd_launch_e2(s, c->D, c->S, H, W, (float)spw.feather_strength, c->e2L, c->e2R);
      vd_launch_pool(s, c->e2L, c->e2R, H, W, spw.blur_ksize, c->bL, c->bR);
    }
    vd_launch_warp(s, rgb, ih, iw, c->S, c->bL, c->bR, H, W, spw.enable_feathering ? 1 : 0, c->L, c->R);
    }
  }
  HIPCHK(hipGetLastError());
  return 0;
}

VD3D_EXPORT int vd3d_pixel_shift(vd3d_ctx* c, const float* rgb_chw, const float* depth, int in_h, int in_w, int W, int H,
                                 const vd3d_shift_params* p, uint8_t* left_bgr, uint8_t* right_bgr, float* shift_or_null) {
  if (!c || !rgb_chw || !depth || !p || !left_bgr || !right_bgr) return set_err(VD3D_E_INVALID, "NULL argument");
  if (in_h < 1 || in_w < 1) return set_err(VD3D_E_INVALID, "bad input size");
  int rc = check_shift_params(p, H, W);
  if (rc) return rc;
  HIPCHK(hipSetDevice(c->device));
  if ((rc = join_pixels(c))) return rc;
  if ((rc = ensure_work(c, H, W))) return rc;
  StageTimer tf(c, "pixel_shift");
  HIPCHK(hipMemsetAsync(c->histA, 0, c->hist_bytes, c->stream));
  vd_stage_args a;
  memset(&a, 0, sizeof a);
  a.have_eye = 0; a.W = W; a.H = H; a.shift = *p; a.ipd_factor = 0.0;
  rc = run_shift_and_warp(c, rgb_chw, depth, in_h, in_w, W, H, *p, a);
  if (rc) return rc;
  const size_t n = (size_t)H * W;
  HIPCHK(hipMemcpyAsync(left_bgr, c->L, 3 * n, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(right_bgr, c->R, 3 * n, hipMemcpyDeviceToDevice, c->stream));
  if (shift_or_null) HIPCHK(hipMemcpyAsync(shift_or_null, c->S, n * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
  return 0;
}

// ---- finishing constants ------------------------------------------------------------------------
static float linspace_host(float start, float end, int steps, int i) {
  if (steps == 1) return start;
  float step = (end - start) / (float)(steps - 1);
  if (i < steps / 2) return fmaf(step, (float)i, start);
  return fmaf(-step, (float)(steps - 1 - i), end);
}
// torch.sum of a contiguous float32 vector in ATen's CPU order (256-bit reduction kernels: 8 lanes, four interleaved vector accumulators, the n % 8 tail
// first, then the lanes in order; below 8 elements four interleaved scalar accumulators; n < 512) -- what `pdf.sum()` of torchvision's _get_gaussian_kernel1d evaluates to.  Same rule as the oracle's sum_aten_f32
// (written independently there); tests/test_aten_restatements.py pins both against torch.sum and against each other through vd3d_debug_gaussian_kernel1d.
static float host_sum_aten(const float* v, int n) {
  if (n < 8) {   // no full vector: the scalar variant (four interleaved scalar accumulators, leftovers into the first)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int i = 0;
    for (; i + 4 <= n; i += 4) { s0 += v[i]; s1 += v[i + 1]; s2 += v[i + 2]; s3 += v[i + 3]; }
    for (; i < n; ++i) s0 += v[i];
    return ((s0 + s1) + s2) + s3;
  }
  const int nv = n >> 3, nq = nv >> 2;
  float acc[4][8];
  for (int j = 0; j < 4; ++j)
    for (int l = 0; l < 8; ++l) acc[j][l] = 0.f;
  for (int i = 0; i < nv; ++i) {
    float* a = acc[i < 4 * nq ? (i & 3) : 0];
    for (int l = 0; l < 8; ++l) a[l] += v[8 * i + l];
  }
  for (int l = 0; l < 8; ++l) acc[0][l] = ((acc[0][l] + acc[1][l]) + acc[2][l]) + acc[3][l];
  float fin = 0.f;
  for (int i = 8 * nv; i < n; ++i) fin += v[i];
  for (int l = 0; l < 8; ++l) fin += acc[0][l];
  return fin;
}
// torch.exp of a float32 on the CPU = oneMKL VML vsExp (high accuracy, AVX-512 path): a table-driven float32 routine, NOT the rounded
// exponential (1.4 % of the inputs of [-8, 0] are an ULP off).  Host-side restatement for the DOF weights (the oracle holds an independent one
// with literal tables; tests/test_aten_restatements.py compares both with torch on every blur level): 2^(j/32) as a float32 head and a
// relative float32 tail, both derived here from their definitions in extended precision; range reduction t = floor(32 x log2 e) / 32 (the
// library's fused multiply-add rounds toward zero onto a positive shifter), r = x - t ln 2 in two fused steps, a degree-3 polynomial tail.
static float host_exp_torch(float x) {
  if (!(fabsf(x) < 87.0f)) return (float)exp((double)x);   // the library's special-case path is not restated (never reached: x in [-8.5, 0])
  struct exp_tabs {   // function-local static of class type: C++11 initialises it once, thread-safely (one context per thread is a supported pattern)
    float th[32], tl[32];
    exp_tabs() {
      for (int j = 0; j < 32; ++j) {
        const long double v = exp2l((long double)j / 32.0L);
        th[j] = (float)v;
        tl[j] = (float)((v - (long double)th[j]) / (long double)th[j]);
      }
    }
  };
  static const exp_tabs tabs;
  const float* th = tabs.th; const float* tl = tabs.tl;
  union { uint32_t u; float f; } c;
  c.u = 0x3fb8aa3bu; const float l2e = c.f;            // log2(e) as the library rounds it
  c.u = 0x3f317218u; const float ln2_hi = c.f;         // ln 2 = ln2_hi - 1.9046542e-09
  c.u = 0x3e2aabf3u; const float c3 = c.f;
  c.u = 0x3f0000f6u; const float c2 = c.f;
  const double t = floor((double)x * (double)l2e * 32.0) * (1.0 / 32.0);
  const long long m32 = (long long)floor((double)x * (double)l2e * 32.0);
  const int j = (int)(m32 & 31);                        // two's complement: the non-negative residue
  const float n = (float)t;
  float r = fmaf(n, -ln2_hi, x);
  r = fmaf(n, 1.9046542e-09f, r);
  const float q = fmaf(fmaf(r, c3, c2), r * r, tl[j] + r);
  const float sc = ldexpf(th[j], (int)floor(t));
  return fmaf(q, sc, sc);
}
// torchvision _get_gaussian_kernel1d in float32 (core/render_3d.py:798-806): k <= 2 DF_RMAX_HOST + 1 taps into out
static void host_gaussian_kernel1d(int k, float sigma, float* out) {
  const float half = (float)((k - 1) * 0.5);
  for (int i = 0; i < k; ++i) {
    float x = linspace_host(-half, half, k, i);
    float t = x / sigma;
    out[i] = host_exp_torch(-0.5f * (t * t));
  }
  const float sum = host_sum_aten(out, k);
  for (int i = 0; i < k; ++i) out[i] = out[i] / sum;
}
// host-only: the product's restatement of torch.exp (MKL vsExp) on n floats -- compared with torch and with the oracle's on the CPU
VD3D_EXPORT int vd3d_debug_exp_torch(const float* x_host, float* out_host, long long n) {
  if (n < 0 || (n > 0 && (!x_host || !out_host))) return set_err(VD3D_E_INVALID, "vd3d_debug_exp_torch: host pointers");
  for (long long i = 0; i < n; ++i) out_host[i] = host_exp_torch(x_host[i]);
  return 0;
}
VD3D_EXPORT int vd3d_debug_gaussian_kernel1d(int k, float sigma, float* out_host) {   // host-only (no GPU needed): the weights the DOF kernels are given
  if (k < 1 || k > 2 * DF_RMAX_HOST + 1 || !(k & 1) || !out_host || !(sigma > 0.f)) return set_err(VD3D_E_INVALID, "gaussian_kernel1d: odd k <= %d, sigma > 0", 2 * DF_RMAX_HOST + 1);
  host_gaussian_kernel1d(k, sigma, out_host);
  return 0;
}

static int make_finish_consts(const vd3d_render_params* p, vd_finish_consts* fc) {
  memset(fc, 0, sizeof *fc);
  const int NL = 5;
  if (p->dof_strength > 0.0) {
    fc->nlev = NL - 1;
    for (int l = 1; l < NL; ++l) {  // apply_dof_cuda :798-806 + torchvision _get_gaussian_kernel1d
      float sigma = linspace_host(0.f, (float)p->dof_strength, NL, l);
      int k = (int)(2 * ceil(2 * (double)sigma) + 1);
      if (k > 2 * DF_RMAX_HOST + 1) return set_err(VD3D_E_UNSUPPORTED, "dof_strength %.3f needs a %d-tap Gaussian (> %d)", p->dof_strength, k, 2 * DF_RMAX_HOST + 1);
      fc->ksz[l - 1] = k;
      host_gaussian_kernel1d(k, sigma, fc->kern[l - 1]);
      if (k <= 9)   // dense association: weight of tap (i, j) = the float32 product of the two 1-D weights (what torchvision's outer product holds)
        for (int i = 0; i < k; ++i)
          for (int j = 0; j < k; ++j) fc->w2[l - 1][i * k + j] = fc->kern[l - 1][i] * fc->kern[l - 1][j];
    }
  }
  fc->fw = (float)(0.35 + 1e-6);
  fc->imax = (float)((NL - 1) - 1e-6);
  fc->sat = (float)p->color_saturation; fc->con = (float)p->color_contrast; fc->bri = (float)p->color_brightness;
  // apply_sharpening :719-728: float32 kernel / np.sum (numpy pairwise order for 9 elements)
  const float c0 = (float)(5.0 + p->sharpness_factor);
  float ks = ((0.f + -1.f) + (0.f + -1.f)) + ((c0 + -1.f) + (0.f + -1.f));
  ks = ks + 0.f;
  fc->sharp_kn = ks != 0.f ? -1.f / ks : -1.f;
  fc->sharp_kc = ks != 0.f ? c0 / ks : c0;
  return 0;
}

#define VD_AREA_MAXT_HOST 12
static int check_fit(const vd3d_render_params* p) {
  // VR: format_3d_output's cv2.resize(eye,(1440,1600)) is the identity only for the 1440x1600 eyes render_sbs_3d makes (:1104,1127)
  if (p->format == VD3D_FMT_VR && (p->fit_w != 1440 || p->fit_h != 1600))
    return set_err(VD3D_E_UNSUPPORTED, "VR with a %dx%d eye canvas would need format_3d_output's INTER_LINEAR resize (not built)", p->fit_w, p->fit_h);
  if (p->format < 0 || p->format > VD3D_FMT_INTERLACED) return set_err(VD3D_E_INVALID, "unknown format %d", p->format);
  int in_w, in_h;
  if (p->format == VD3D_FMT_HALF_SBS) { in_w = p->fit_w; in_h = p->fit_h; }
  else {
    const double ta = (double)p->fit_w / p->fit_h, ca = (double)p->warp_w / p->warp_h;
    if (ca > ta) { in_w = p->fit_w; in_h = (int)(p->fit_w / ca); }
    else { in_h = p->fit_h; in_w = (int)(ca * p->fit_h); }
  }
  if (in_w < 1 || in_h < 1) return set_err(VD3D_E_INVALID, "empty fit %dx%d", in_w, in_h);
  const bool upscale = in_w > p->warp_w || in_h > p->warp_h;   // OpenCV: linear machinery with area-mode coefficients (k_sharp_mux)
  if (!upscale && (p->warp_w % in_w || p->warp_h % in_h) && ((double)p->warp_w / in_w > VD_AREA_MAXT_HOST - 2 || (double)p->warp_h / in_h > VD_AREA_MAXT_HOST - 2))
    return set_err(VD3D_E_UNSUPPORTED, "fractional INTER_AREA ratio above %d not built", VD_AREA_MAXT_HOST - 2);
  const int mux_w = (p->format == VD3D_FMT_HALF_SBS || p->format == VD3D_FMT_FULL_SBS || p->format == VD3D_FMT_VR) ? 2 * p->fit_w : p->fit_w;
  if (p->out_w != mux_w || p->out_h != p->fit_h) return set_err(VD3D_E_INVALID, "out size %dx%d does not match mux %dx%d", p->out_w, p->out_h, mux_w, p->fit_h);
  return 0;
}

static int g_fused_fit = getenv("VD3D_FUSED_FIT") ? atoi(getenv("VD3D_FUSED_FIT")) : 3;   // bit 0: E1 in front of a fit it does not take; bit 1: k_sharp_fit behind k_dof_grade4.  VD3D_FUSED_FIT=0 / vd3d_debug_tune(3, 0): the unfused DOF / grade kernels in front of every fit E1 does not take (A/B, tests)
static int run_finish(vd3d_ctx* c, const uint8_t* L, const uint8_t* R, const float* dn, int eh, int ew,
                      const vd3d_render_params* p, const vd_finish_consts& fc, float focal, int use_override, int bw, int bs,
                      uint8_t* out, const vd_dev_work* wk = nullptr) {
  StageTimer t(c, "finish");
  if (!wk) wk = c->work;
  if (fc.nlev && !(eh == p->warp_h && ew == p->warp_w) && vd_interp_premult(1, p->warp_h, p->warp_w, p->aten_sum_threads)) {   // depth_for_dof (:1347-1350) by ATen's other kernel
    int rcx = pix_exclusive(c); if (rcx) return rcx;
    if ((rcx = premult_resize(c, 1, &dn, &eh, &ew, p->warp_h, p->warp_w, p->aten_sum_threads, &c->pm_dd, &c->pm_dd_cap))) return rcx;
  }
  const int dense = (p->dof_dense_conv && fc.nlev) ? 1 : 0;   // the reference's dense k x k conv order
  const float* d_w2 = nullptr;
  if (dense) {
    if (c->w2_cur < 0 || memcmp(c->w2_tabs[c->w2_cur].host, fc.w2, sizeof fc.w2) != 0) {   // first frame or a new dof_strength (rare)
      c->w2_cur = -1;
      for (size_t i = 0; i < c->w2_tabs.size(); ++i)
        if (memcmp(c->w2_tabs[i].host, fc.w2, sizeof fc.w2) == 0) c->w2_cur = (int)i;
      if (c->w2_cur < 0) {
        if (c->w2_tabs.size() >= 64) {   // a caller sweeping the slider: drain every stream of the context once, then recycle all tables
          HIPCHK(hipDeviceSynchronize());
          for (auto& t : c->w2_tabs) (void)hipFree(t.dev);
          c->w2_tabs.clear();
        }
        vd3d_ctx::w2_tab t;
        memcpy(t.host, fc.w2, sizeof fc.w2);
        HIPCHK(hipMalloc((void**)&t.dev, sizeof fc.w2));
        HIPCHK(hipMemcpy(t.dev, t.host, sizeof fc.w2, hipMemcpyHostToDevice));   // a fresh buffer: nothing queued reads it yet
        c->w2_tabs.push_back(t);
        c->w2_cur = (int)c->w2_tabs.size() - 1;
      }
    }
    d_w2 = c->w2_tabs[c->w2_cur].dev;
  }
  if (vd_launch_finish_fused(c->stream, L, R, dn, eh, ew, *p, fc, wk, focal, use_override, bw, bs, out, dense, d_w2)) {
    HIPCHK(hipGetLastError());
    return 0;
  }
  { int rcx = pix_exclusive(c); if (rcx) return rcx; }   // gL / gR / gLR are shared by the context's pixel streams
  // E1 refused the FIT only (a fractional or up-scaling INTER_AREA ratio, the VR canvas): it still runs -- 1:1 into a side-by-side scratch of
  // sharpened eyes -- and the fit / mux kernel reads that instead of sharpening two graded planes itself (round 4; same bytes:
  // tests/test_hip_widen.py runs both ways)
  bool taps_ok = (g_fused_fit & 1) != 0;
  for (int l = 0; l < fc.nlev; ++l) if (fc.ksz[l] > 9 || fc.ksz[l] < 3) taps_ok = false;
  if (taps_ok && (p->warp_w & 3) == 0) {
    vd3d_render_params q = *p;
    q.format = VD3D_FMT_HALF_SBS; q.fit_w = p->warp_w; q.fit_h = p->warp_h; q.out_w = 2 * p->warp_w; q.out_h = p->warp_h;
    const size_t need = (size_t)6 * p->warp_w * p->warp_h;
    if (c->gLR_cap < need) { HIPCHK(hipDeviceSynchronize()); HIPCHK(re_alloc(&c->gLR, need)); c->gLR_cap = need; }
    if (vd_launch_finish_fused(c->stream, L, R, dn, eh, ew, q, fc, wk, focal, use_override, bw, bs, c->gLR, dense, d_w2)) {
      vd_launch_sharp_mux(c->stream, c->gLR, c->gLR + (size_t)3 * p->warp_w, *p, fc, out, 2 * p->warp_w);
      HIPCHK(hipGetLastError());
      return 0;
    }
  }
  const float* d_wk = nullptr;
  if (dense) {   // k_dof_grade4's weights: [level][row][32] device table, looked up / built like d_w2 above
    auto same = [&](const vd3d_ctx::wk_tab& t) { return memcmp(t.ksz, fc.ksz, sizeof fc.ksz) == 0 && memcmp(t.kern, fc.kern, sizeof fc.kern) == 0; };
    if (c->wk_cur < 0 || !same(c->wk_tabs[c->wk_cur])) {
      c->wk_cur = -1;
      for (size_t i = 0; i < c->wk_tabs.size(); ++i) if (same(c->wk_tabs[i])) c->wk_cur = (int)i;
      if (c->wk_cur < 0) {
        if (c->wk_tabs.size() >= 64) {
          HIPCHK(hipDeviceSynchronize());
          for (auto& t : c->wk_tabs) (void)hipFree(t.dev);
          c->wk_tabs.clear();
        }
        vd3d_ctx::wk_tab t;
        memcpy(t.ksz, fc.ksz, sizeof fc.ksz); memcpy(t.kern, fc.kern, sizeof fc.kern);
        std::vector<float> host(VD_D4_WTAB_FLOATS, 0.f);
        for (int l = 0; l < fc.nlev; ++l)
          for (int i = 0; i < fc.ksz[l]; ++i)
            for (int j = 0; j < fc.ksz[l]; ++j) host[((size_t)l * 31 + i) * 32 + j] = fc.kern[l][i] * fc.kern[l][j];   // float32 product, as torchvision's outer product holds it
        HIPCHK(hipMalloc((void**)&t.dev, sizeof(float) * VD_D4_WTAB_FLOATS));
        HIPCHK(hipMemcpy(t.dev, host.data(), sizeof(float) * VD_D4_WTAB_FLOATS, hipMemcpyHostToDevice));   // a fresh buffer: nothing queued reads it yet
        c->wk_tabs.push_back(t);
        c->wk_cur = (int)c->wk_tabs.size() - 1;
      }
    }
    d_wk = c->wk_tabs[c->wk_cur].dev;
  }
  if (dense) vd_launch_dof_grade_dense(c->stream, L, R, dn, eh, ew, p->warp_h, p->warp_w, fc, wk, focal, use_override, bw, bs, c->gL, c->gR, d_wk);   // both eyes per launch (round 6)
  else {
    vd_launch_dof_grade(c->stream, L, dn, eh, ew, p->warp_h, p->warp_w, fc, wk, focal, use_override, bw, bs, c->gL, dense, d_wk);
    vd_launch_dof_grade(c->stream, R, dn, eh, ew, p->warp_h, p->warp_w, fc, wk, focal, use_override, bw, bs, c->gR, dense, d_wk);
  }
  // graded planes -> sharpen + fit + mux: the fused kernel's epilogue as a kernel of its own where its fit conditions hold, else the per-pixel one
  if (!((g_fused_fit & 2) && vd_launch_sharp_fit(c->stream, c->gL, c->gR, *p, fc, out))) vd_launch_sharp_mux(c->stream, c->gL, c->gR, *p, fc, out);
  HIPCHK(hipGetLastError());
  return 0;
}

VD3D_EXPORT int vd3d_finish_frame(vd3d_ctx* c, const uint8_t* left_bgr, const uint8_t* right_bgr, const float* depth_norm,
                                  int eye_h, int eye_w, const vd3d_render_params* p, double focal_depth, int bar_width,
                                  int bar_side, uint8_t* out_bgr) {
  if (!c || !left_bgr || !right_bgr || !depth_norm || !p || !out_bgr) return set_err(VD3D_E_INVALID, "NULL argument");
  int rc = check_fit(p);
  if (rc) return rc;
  vd_finish_consts fc;
  if ((rc = make_finish_consts(p, &fc))) return rc;
  HIPCHK(hipSetDevice(c->device));
  if ((rc = join_pixels(c))) return rc;
  if ((rc = ensure_work(c, p->warp_h, p->warp_w))) return rc;
  return run_finish(c, left_bgr, right_bgr, depth_norm, eye_h, eye_w, p, fc, (float)focal_depth, 1, bar_width, bar_side, out_bgr);
}

// ---- B2 ---------------------------------------------------------------------------------------
static int render_frame_impl(vd3d_ctx* c, const uint8_t* frame_bgr, const void* depth, int depth_fmt,
                             const vd3d_render_params* p, uint8_t* out_bgr, bool blank) {
  if (!c || !depth || !p || !frame_bgr || !out_bgr) return set_err(VD3D_E_INVALID, "NULL argument");
  if (depth_fmt < 0 || depth_fmt > VD3D_DEPTH_GRAY_U8) return set_err(VD3D_E_INVALID, "bad depth_fmt %d", depth_fmt);
  if (!p->auto_crop_black_bars &&
      (p->crop_x < 0 || p->crop_y < 0 || p->crop_w < 1 || p->crop_h < 1 || p->crop_x + p->crop_w > p->src_w || p->crop_y + p->crop_h > p->src_h))
    return set_err(VD3D_E_INVALID, "crop window outside the frame");
  if (p->auto_crop_black_bars && (!(p->target_ratio > 0.0) || p->src_w < 1 || p->src_h < 1))
    return set_err(VD3D_E_INVALID, "auto_crop_black_bars needs target_ratio > 0");
  if (p->eye_w < 2 || p->eye_h < 2) return set_err(VD3D_E_INVALID, "eye size too small");
  int rc = check_fit(p);
  if (rc) return rc;
  // blank frame: the source-sized frame itself goes through sharpen / fit / mux (:1279-1281, :1406-1419)
  vd3d_render_params pb = *p;
  if (blank) {
    pb.warp_w = p->src_w; pb.warp_h = p->src_h;
    if (p->src_w < 1 || p->src_h < 1) return set_err(VD3D_E_INVALID, "bad source size");
    if ((rc = check_fit(&pb))) return rc;
  }
  // render_sbs_3d forwards literals for the pop/lock controls and never forwards parallax_balance (:1284-1331)
  vd3d_shift_params sp = p->shift;
  sp.parallax_balance = 0.8; sp.depth_pop_gamma = 0.85; sp.depth_pop_mid = 0.50; sp.depth_stretch_lo = 0.05;
  sp.depth_stretch_hi = 0.95; sp.fg_pop_multiplier = 1.20; sp.bg_push_multiplier = 1.10; sp.subject_lock_strength = 1.00;
  sp.aten_threads = p->aten_sum_threads;   // ATen's scalar tails for the same reference process (include/vd3d.h)
  if ((rc = check_shift_params(&sp, p->warp_h, p->warp_w))) return rc;
  vd_finish_consts fc;
  if ((rc = make_finish_consts(p, &fc))) return rc;
  HIPCHK(hipSetDevice(c->device));
  if ((rc = join_pixels(c))) return rc;   // the unsharded pixel pass shares L / R / S with overlapped ones
  if ((rc = ensure_eye(c, p->eye_h, p->eye_w))) return rc;
  if ((rc = ensure_work(c, p->warp_h, p->warp_w))) return rc;
  hipStream_t s = c->stream;
  const long long ne = (long long)p->eye_h * p->eye_w;
  StageTimer tf(c, "frame");
  vd_stage_args a;
  memset(&a, 0, sizeof a);
  a.have_eye = 1; a.W = p->warp_w; a.H = p->warp_h; a.n_eye = ne;
  { const char* e = getenv("VD3D_DBG"); a.dbg = e ? atoi(e) : 0; }
  a.blank = blank ? 1 : 0;
  a.n_crop = (long long)(p->eye_h * 3 / 4 - p->eye_h / 4) * (long long)(p->eye_w * 3 / 4 - p->eye_w / 4);
  a.ipd_factor = p->ipd_factor; a.shift = sp;
  if ((rc = aten_setup(c, p, &a))) return rc;
  float* dn_cur = c->dn[c->dn_cur];
  if (p->auto_crop_black_bars) {   // :1230-1248 decided on device, no host round trip
    if (p->src_h > c->rowflag_cap) { HIPCHK(re_alloc(&c->rowflag, (size_t)p->src_h)); c->rowflag_cap = p->src_h; }
    vd_launch_autocrop(s, frame_bgr, p->src_h, p->src_w, p->target_ratio, c->rowflag, c->work);
    c->crop_scalars_dirty = true;
  } else if (c->crop_scalars_dirty) {
    HIPCHK(hipMemsetAsync(&c->work->fs.crop_top, 0, 2 * sizeof(int32_t), s));
    c->crop_scalars_dirty = false;
  }
  {
    StageTimer t(c, "select_eye");   // fused chain: ingest (+A0), b0 (B0); eye stats ride on the next launch
    HIPCHK(hipMemsetAsync(c->histA, 0, c->hist_bytes, s));
    vd_batch b = batch_of_one(c);
    b.f[0].frame = frame_bgr; b.f[0].depth = depth;
    vd_launch_chain_eye(s, b, depth_fmt, *p, a);
  }
  // a blank frame still runs the whole select chain (its eye-res half carries the filters and the bars; the work-res half only
  // computes unused pop-shaping constants -- blank frames are rare and this keeps one code path), but no shift map and no warp
  rc = run_shift_and_warp(c, c->rgb_eye, dn_cur, p->eye_h, p->eye_w, p->warp_w, p->warp_h, sp, a, blank);
  if (rc) return rc;
  if (blank) {
    StageTimer t(c, "finish");
    const size_t nb = (size_t)p->src_h * p->src_w * 3;
    if (nb > c->blank_cap) { HIPCHK(re_alloc(&c->blank_eye, nb)); c->blank_cap = nb; }
    vd_launch_blank_eye(s, frame_bgr, p->src_h, p->src_w, c->work, c->blank_eye);
    vd_launch_sharp_mux(s, c->blank_eye, c->blank_eye, pb, fc, out_bgr);
    HIPCHK(hipGetLastError());
  } else {
    rc = run_finish(c, c->L, c->R, dn_cur, p->eye_h, p->eye_w, p, fc, 0.f, 0, 0, 0, out_bgr);
    if (rc) return rc;
  }
  c->dn_cur ^= 1;
  return 0;
}

VD3D_EXPORT int vd3d_render_frame(vd3d_ctx* c, const uint8_t* frame_bgr, const void* depth, int depth_fmt,
                                  const vd3d_render_params* p, uint8_t* out_bgr) {
  return render_frame_impl(c, frame_bgr, depth, depth_fmt, p, out_bgr, false);
}

// The frame is in the skip_blank_frames set (core/render_3d.py:1046-1060, 1278-1281): both eyes are the raw source frame; the
// depth filters, ShiftSmoother, dynamic parallax scale and the floating-window bars advance, pixel_shift_cuda (FloatingWindowTracker),
// the ipd scaling, the focal tracker, DOF and the colour grade do not run.  Sharpen / fit / mux work on the source-sized frame.
VD3D_EXPORT int vd3d_render_frame_blank(vd3d_ctx* c, const uint8_t* frame_bgr, const void* depth, int depth_fmt,
                                        const vd3d_render_params* p, uint8_t* out_bgr) {
  return render_frame_impl(c, frame_bgr, depth, depth_fmt, p, out_bgr, true);
}

// ---- frame sharding, three-phase protocol (visiondepth3d_amd/sharded.py) ----------------------------------
VD3D_EXPORT int vd3d_shard_begin(vd3d_ctx* c, const vd3d_render_params* p, int n_slots) {
  if (!c || !p || n_slots < 1 || n_slots > VD_MAX_STEP) return set_err(VD3D_E_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  int rc;
  if ((rc = ensure_eye(c, p->eye_h, p->eye_w))) return rc;
  if ((rc = ensure_work(c, p->warp_h, p->warp_w))) return rc;
  const bool same = c->n_slots >= n_slots && c->slot_eh == p->eye_h && c->slot_ew == p->eye_w && c->slot_H == p->warp_h && c->slot_W == p->warp_w;
  if (same) return 0;
  if (c->pix_stream) HIPCHK(hipStreamSynchronize(c->pix_stream));
  for (int k = 0; k < VD_MAX_PIX - 1; ++k) { if (c->pix_x[k]) HIPCHK(hipStreamSynchronize(c->pix_x[k])); c->pix_pending_x[k] = false; }
  c->pix_pending = false; c->excl_set = false; std::fill(c->slot_busy.begin(), c->slot_busy.end(), 0);
  HIPCHK(hipStreamSynchronize(c->stream));
  for (auto q : c->slot_rgb) (void)hipFree(q);
  for (auto q : c->slot_dn) (void)hipFree(q);
  for (auto q : c->slot_D) (void)hipFree(q);
  for (auto q : c->slot_tdf) (void)hipFree(q);
  for (auto q : c->slot_tdfp) (void)hipFree(q);
  c->slot_rgb.clear(); c->slot_dn.clear(); c->slot_D.clear(); c->slot_tdf.clear(); c->slot_tdfp.clear();
  for (auto q : c->bdc) (void)hipFree(q);
  c->bdc.clear();
  if (c->bhist) { (void)hipFree(c->bhist); c->bhist = nullptr; }
  c->n_batch = 0;
  const size_t ne = (size_t)p->eye_h * p->eye_w, n = (size_t)p->warp_h * p->warp_w;
  for (int i = 0; i < n_slots; ++i) {
    float *a = nullptr, *b = nullptr, *d = nullptr, *t0 = nullptr, *t1 = nullptr;
    HIPCHK(hipMalloc((void**)&a, 3 * ne * sizeof(float))); HIPCHK(hipMalloc((void**)&b, ne * sizeof(float)));
    HIPCHK(hipMalloc((void**)&d, n * sizeof(float)));
    HIPCHK(hipMalloc((void**)&t0, ne * sizeof(float))); HIPCHK(hipMalloc((void**)&t1, ne * sizeof(float)));
    c->slot_rgb.push_back(a); c->slot_dn.push_back(b); c->slot_D.push_back(d); c->slot_tdf.push_back(t0); c->slot_tdfp.push_back(t1);
  }
  if (!c->etab) HIPCHK(hipMalloc((void**)&c->etab, (size_t)(VD_MAX_STEP + 1) * VD_ETAB * sizeof(float)));
  if (!c->crop_tab) HIPCHK(hipMalloc((void**)&c->crop_tab, (size_t)VD_MAX_STEP * 4 * sizeof(int)));
  HIPCHK(re_alloc(&c->slot_work, (size_t)n_slots));
  c->slot_prev.assign((size_t)n_slots, nullptr);
  for (int i = 0; i < n_slots; ++i) c->slot_prev[i] = c->slot_tdfp[i];
  c->n_batch = n_slots < VD_MAX_BATCH ? n_slots : VD_MAX_BATCH;
  HIPCHK(hipMalloc((void**)&c->bhist, (size_t)c->n_batch * c->hist_bytes));
  for (int i = 0; i < c->n_batch; ++i) { float* q = nullptr; HIPCHK(hipMalloc((void**)&q, n * sizeof(float))); c->bdc.push_back(q); }
  c->n_slots = n_slots; c->slot_eh = p->eye_h; c->slot_ew = p->eye_w; c->slot_H = p->warp_h; c->slot_W = p->warp_w;
  return 0;
}
// phase 3: shift plane, fused warp and finishing kernels of one owned frame from its slot -> muxed frame
static int shard_pixels_impl(vd3d_ctx* c, int slot, const vd3d_render_params* p, uint8_t* out_bgr, const uint8_t* blank_frame_bgr) {
  if (!c || !p || !out_bgr || slot < 0 || slot >= c->n_slots) return set_err(VD3D_E_INVALID, "bad argument");
  int rc = check_fit(p);
  if (rc) return rc;
  vd3d_render_params pb = *p;
  if (blank_frame_bgr) {   // the source-sized frame itself goes through sharpen / fit / mux (:1279-1281, :1406-1419)
    pb.warp_w = p->src_w; pb.warp_h = p->src_h;
    if (p->src_w < 1 || p->src_h < 1) return set_err(VD3D_E_INVALID, "bad source size");
    if ((rc = check_fit(&pb))) return rc;
  }
  vd3d_shift_params sp = p->shift;
  sp.parallax_balance = 0.8; sp.depth_pop_gamma = 0.85; sp.depth_pop_mid = 0.50; sp.depth_stretch_lo = 0.05;
  sp.depth_stretch_hi = 0.95; sp.fg_pop_multiplier = 1.20; sp.bg_push_multiplier = 1.10; sp.subject_lock_strength = 1.00;
  sp.aten_threads = p->aten_sum_threads;   // ATen's scalar tails for the same reference process (include/vd3d.h)
  vd_finish_consts fc;
  if ((rc = make_finish_consts(p, &fc))) return rc;
  HIPCHK(hipSetDevice(c->device));
  if (p->warp_h != c->slot_H || p->warp_w != c->slot_W || p->eye_h != c->slot_eh || p->eye_w != c->slot_ew)
    return set_err(VD3D_E_INVALID, "vd3d_shard_pixels: %dx%d (eye %dx%d) is not the size of vd3d_shard_begin (%dx%d, eye %dx%d)", p->warp_w, p->warp_h,
                   p->eye_w, p->eye_h, c->slot_W, c->slot_H, c->slot_ew, c->slot_eh);
  if (c->H != p->warp_h || c->W != p->warp_w) {
    // another entry point (vd3d_pixel_shift / vd3d_render_frame at another size) re-sized the context's shared warp-res planes since
    // vd3d_shard_begin: bring them back before a pixel pass writes H x W planes into them (hipFree synchronises the device)
    if ((rc = join_pixels(c))) return rc;
    if ((rc = ensure_work(c, p->warp_h, p->warp_w))) return rc;
  }
  // overlapped mode: this pass runs on pix_stream behind everything enqueued on the main stream so far (the slot's measurements and
  // the replay that patched its constants); the main stream is free to start the next step's measurement chain meanwhile
  struct StreamSwap {
    vd3d_ctx* c; hipStream_t saved;
    explicit StreamSwap(vd3d_ctx* ctx) : c(ctx), saved(ctx->stream) {}
    ~StreamSwap() { c->stream = saved; c->cur_pix = -1; c->cur_excl = false; }
  } swap(c);
  const int H = p->warp_h, W = p->warp_w;
  int k = 0;
  float* S = c->S; uint8_t* L = c->L; uint8_t* R = c->R; float* E2 = c->E2;
  if (c->pix_overlap) {
    while ((int)c->slot_done.size() <= slot) {
      hipEvent_t e;
      HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      c->slot_done.push_back(e); c->slot_busy.push_back(0);
    }
    if (c->n_pix > 1 && !blank_frame_bgr) {
      k = (int)(c->pix_rr++ % (unsigned)c->n_pix);
      if (c->x_H != H || c->x_W != W) {   // the extra streams' planes follow the warp size (rare: drain them first)
        for (int j = 0; j < VD_MAX_PIX - 1; ++j) {
          if (c->pix_x[j]) HIPCHK(hipStreamSynchronize(c->pix_x[j]));
          HIPCHK(re_alloc(&c->S_x[j], (size_t)0)); HIPCHK(re_alloc(&c->L_x[j], (size_t)0)); HIPCHK(re_alloc(&c->R_x[j], (size_t)0));
          HIPCHK(re_alloc(&c->E2_x[j], (size_t)0));
        }
        c->x_H = H; c->x_W = W;
      }
      if (k > 0) {
        const size_t n = (size_t)H * W;
        if (!c->S_x[k - 1]) {
          HIPCHK(re_alloc(&c->S_x[k - 1], n)); HIPCHK(re_alloc(&c->L_x[k - 1], 3 * n)); HIPCHK(re_alloc(&c->R_x[k - 1], 3 * n));
          HIPCHK(re_alloc(&c->E2_x[k - 1], 2 * n));
        }
        S = c->S_x[k - 1]; L = c->L_x[k - 1]; R = c->R_x[k - 1]; E2 = c->E2_x[k - 1];
      }
    }
    hipStream_t sk = k == 0 ? c->pix_stream : c->pix_x[k - 1];
    HIPCHK(hipEventRecord(c->ev_chain, c->stream));
    HIPCHK(hipStreamWaitEvent(sk, c->ev_chain, 0));
    if (c->excl_set) HIPCHK(hipStreamWaitEvent(sk, c->ev_excl, 0));   // a pass that used the shared fallback planes is still the latest such user
    c->stream = sk;
    c->cur_pix = k; c->cur_excl = false;
    if (blank_frame_bgr) { int rcx = pix_exclusive(c); if (rcx) return rcx; }   // blank_eye is a shared plane
  }
  hipStream_t s = c->stream;
  const vd_dev_work* wk = &c->slot_work[slot];
  if (blank_frame_bgr) {
    StageTimer t(c, "finish");
    const size_t nb = (size_t)p->src_h * p->src_w * 3;
    if (nb > c->blank_cap) { HIPCHK(re_alloc(&c->blank_eye, nb)); c->blank_cap = nb; }
    vd_launch_blank_eye(s, blank_frame_bgr, p->src_h, p->src_w, wk, c->blank_eye);
    vd_launch_sharp_mux(s, c->blank_eye, c->blank_eye, pb, fc, out_bgr);
    HIPCHK(hipGetLastError());
  } else {
  { StageTimer t(c, "warp");
    // fused path with feathering: k_e2w writes the gradient mask of both eyes, W1 starts at its window sums
    const vd3d_shift_params spw = warp_stage_params(sp);   // feather_strength <= 0 (the GUI's default): W1 without mask, window sums and blend -- exact
    const float* rgb = c->slot_rgb[slot];
    int rih = p->eye_h, riw = p->eye_w;
    if (!(rih == H && riw == W) && vd_interp_premult(3, H, W, sp.aten_threads)) {   // N-thread ATen mode: the eye -> warp resize of :595 by ATen's other kernel
      if ((rc = pix_exclusive(c))) return rc;
      if ((rc = premult_resize(c, 3, &rgb, &rih, &riw, H, W, sp.aten_threads, &c->pm_rgb, &c->pm_rgb_cap))) return rc;
    }
    const bool pre = spw.enable_feathering && vd_warp_fused_ok(rih, riw, H, W, spw);
    // without feathering W1 computes the shift values of its own tile (round 6): no k_shift launch, no S plane (8 N bytes of traffic less)
    const bool fold = vd_warp_fold_ok(rih, riw, H, W, spw, sp);
    const vd_shift_fold ff = {wk, sp, nullptr};
    if (!fold) { StageTimer t1(c, "shift"); vd_launch_shift(s, c->slot_D[slot], H, W, wk, sp, S); }
    bool fused;
    { StageTimer t2(c, "w1");
      if (pre) { StageTimer t3(c, "e2w"); vd_launch_e2w(s, c->slot_D[slot], S, H, W, (float)spw.feather_strength, E2); }
      fused = vd_launch_warp_fused(s, rgb, rih, riw, c->slot_D[slot], S, H, W, spw, L, R, pre ? E2 : nullptr, fold ? &ff : nullptr); }
    if (!fused && fold) { StageTimer t1(c, "shift"); vd_launch_shift(s, c->slot_D[slot], H, W, wk, sp, S); }
    if (!fused) {
      if ((rc = pix_exclusive(c))) return rc;
      if (spw.enable_feathering) {
        vd_launch_e2(s, c->slot_D[slot], S, H, W, (float)spw.feather_strength, c->e2L, c->e2R);
        vd_launch_pool(s, c->e2L, c->e2R, H, W, spw.blur_ksize, c->bL, c->bR);
      }
      vd_launch_warp(s, rgb, rih, riw, S, c->bL, c->bR, H, W, spw.enable_feathering ? 1 : 0, L, R);
    }
  }
  HIPCHK(hipGetLastError());
  rc = run_finish(c, L, R, c->slot_dn[slot], p->eye_h, p->eye_w, p, fc, 0.f, 0, 0, 0, out_bgr, wk);
  if (rc) return rc;
  }
  if (c->pix_overlap) {
    HIPCHK(hipEventRecord(c->slot_done[slot], c->stream));
    if (k == 0) { HIPCHK(hipEventRecord(c->ev_pix_last, c->stream)); c->pix_pending = true; }
    else { HIPCHK(hipEventRecord(c->ev_pix_last_x[k - 1], c->stream)); c->pix_pending_x[k - 1] = true; }
    if (c->cur_excl) { HIPCHK(hipEventRecord(c->ev_excl, c->stream)); c->excl_set = true; }
    c->slot_busy[slot] = 1;
  }
  return 0;
}

VD3D_EXPORT int vd3d_shard_pixels(vd3d_ctx* c, int slot, const vd3d_render_params* p, uint8_t* out_bgr) {
  return shard_pixels_impl(c, slot, p, out_bgr, nullptr);
}
// the pixel pass of an own frame that is in the skip_blank_frames set (core/render_3d.py:1278-1281): both eyes are the source frame
VD3D_EXPORT int vd3d_shard_pixels_blank(vd3d_ctx* c, int slot, const uint8_t* frame_bgr, const vd3d_render_params* p, uint8_t* out_bgr) {
  if (!frame_bgr) return set_err(VD3D_E_INVALID, "NULL frame");
  return shard_pixels_impl(c, slot, p, out_bgr, frame_bgr);
}

// Overlapped pixel passes: with enable != 0, vd3d_shard_pixels is enqueued on a second stream of the context, ordered after all work
// enqueued so far, and returns; the measurement chain of the NEXT step (vd3d_shard2_p1 ... r2, latency-bound) then runs concurrently with
// the pixel kernels of this one.  Slots are guarded: a call that overwrites a slot first waits for the pixel pass that still reads it, so
// a caller that alternates between two slot sets gets the overlap and a caller that does not gets the sequential order.  Outputs are
// complete after vd3d_sync, or, for consumers ordered on the context's stream, after vd3d_join_pixels.
VD3D_EXPORT int vd3d_set_pixel_overlap(vd3d_ctx* c, int enable) {
  if (!c) return set_err(VD3D_E_INVALID, "NULL context");
  if (enable < 0 || enable > VD_MAX_PIX) return set_err(VD3D_E_INVALID, "vd3d_set_pixel_overlap: 0 (off) or 1 .. %d pixel streams", VD_MAX_PIX);
  HIPCHK(hipSetDevice(c->device));
  int rc = join_pixels(c);
  if (rc) return rc;
  if (enable && !c->pix_stream) {
    HIPCHK(hipStreamCreateWithFlags(&c->pix_stream, hipStreamNonBlocking));   // stream priorities: measured, no effect either way
    HIPCHK(hipEventCreateWithFlags(&c->ev_chain, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_pix_last, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_excl, hipEventDisableTiming));
  }
  for (int k = 1; k < enable; ++k)
    if (!c->pix_x[k - 1]) {
      HIPCHK(hipStreamCreateWithFlags(&c->pix_x[k - 1], hipStreamNonBlocking));
      HIPCHK(hipEventCreateWithFlags(&c->ev_pix_last_x[k - 1], hipEventDisableTiming));
    }
  c->pix_overlap = enable != 0;
  c->n_pix = enable > 1 ? enable : 1;
  return 0;
}
VD3D_EXPORT int vd3d_join_pixels(vd3d_ctx* c) {
  if (!c) return set_err(VD3D_E_INVALID, "NULL context");
  HIPCHK(hipSetDevice(c->device));
  return join_pixels(c);
}
// host-side wait for the overlapped pixel pass of one slot (its output frame is then complete); no-op if none is outstanding
VD3D_EXPORT int vd3d_wait_pixels(vd3d_ctx* c, int slot) {
  if (!c || slot < 0) return set_err(VD3D_E_INVALID, "bad argument");
  // not gated on slot_busy: a later chain call may already have ordered the first stream behind this pass (clearing the flag)
  // while the host has not waited yet; synchronising a completed or never-recorded event returns at once
  if (slot < (int)c->slot_done.size()) HIPCHK(hipEventSynchronize(c->slot_done[slot]));
  return 0;
}


// ---- frame sharding, measure / replay protocol (visiondepth3d_amd/sharded.py: MeasureReplaySharder) ----------
// Replicated work per foreign frame = ONE small kernel (the TemporalDepthFilter plane EMA); everything else is measured by the
// owner, exchanged as a few numbers per frame and replayed as scalar recurrences on every rank.
static int shard2_args(vd3d_ctx* c, const vd3d_render_params* p, vd_stage_args* a, vd3d_shift_params* sp) {
  *sp = p->shift;
  sp->parallax_balance = 0.8; sp->depth_pop_gamma = 0.85; sp->depth_pop_mid = 0.50; sp->depth_stretch_lo = 0.05;
  sp->depth_stretch_hi = 0.95; sp->fg_pop_multiplier = 1.20; sp->bg_push_multiplier = 1.10; sp->subject_lock_strength = 1.00;
  sp->aten_threads = p->aten_sum_threads;
  memset(a, 0, sizeof *a);
  a->have_eye = 1; a->W = p->warp_w; a->H = p->warp_h; a->n_eye = (long long)p->eye_h * p->eye_w;
  a->n_crop = (long long)(p->eye_h * 3 / 4 - p->eye_h / 4) * (long long)(p->eye_w * 3 / 4 - p->eye_w / 4);
  a->ipd_factor = p->ipd_factor; a->shift = *sp; a->etab = c->etab;
  { static int dbg = -1; if (dbg < 0) { const char* e = getenv("VD3D_DBG"); dbg = e ? atoi(e) : 0; } a->dbg = dbg; }   // timing probes only (vd3d_kernels.h)
  return aten_setup(c, p, a);
}
// P1, for every OWN frame of the step in frame order (a rank owns a contiguous chunk of the step): ingest (RGB kept in the slot),
// TemporalDepthFilter plane EMA, exact q.02 / q.98 of the filtered plane written to q_out_dev[0..1]; the filtered planes of this
// frame and of the frame before it are kept in the slot.  The plane EMA carries over from the previous frame of the clip, wherever
// it was rendered: vd3d_tdf_plane_import installs the plane the previous chunk's owner exported.
// Batched form: n consecutive own frames (slots slot0 .. slot0 + n - 1, step indices step_idx0 ..) in TWO launches -- K1 walks the
// frames inside every workgroup (the EMA is a per-pixel recurrence), K2 runs the frames side by side; per-launch costs are paid once.
static int shard2_p1_impl(vd3d_ctx* c, const uint8_t* const* frames_bgr, const void* const* depths, int depth_fmt, const vd3d_render_params* p,
                          int step_idx0, int slot0, int n, float* q_out_dev) {
  if (!c || !frames_bgr || !depths || !p || !q_out_dev || n < 1 || step_idx0 < 0 || step_idx0 + n > VD_MAX_STEP) return set_err(VD3D_E_INVALID, "bad argument");
  if (slot0 < 0 || slot0 + n > c->n_slots) return set_err(VD3D_E_INVALID, "slots %d..%d out of range (vd3d_shard_begin)", slot0, slot0 + n - 1);
  if (n > c->n_batch) return set_err(VD3D_E_INVALID, "batch of %d frames > %d", n, c->n_batch);
  for (int j = 0; j < n; ++j) if (!frames_bgr[j] || !depths[j]) return set_err(VD3D_E_INVALID, "an own frame needs the frame and its depth plane");
  if (depth_fmt < 0 || depth_fmt > VD3D_DEPTH_GRAY_U8) return set_err(VD3D_E_INVALID, "bad depth_fmt %d", depth_fmt);
  if (p->auto_crop_black_bars && !c->crop_tab_set)
    return set_err(VD3D_E_INVALID, "auto_crop_black_bars in a sharded step: call vd3d_shard2_p0 on the own frames and vd3d_shard2_set_crops first");
  HIPCHK(hipSetDevice(c->device));
  for (int j = 0; j < n; ++j) { int rcw = wait_slot(c, slot0 + j); if (rcw) return rcw; }
  hipStream_t s = c->stream;
  vd_stage_args a; vd3d_shift_params sp;
  { int rca = shard2_args(c, p, &a, &sp); if (rca) return rca; }
  a.crop_tab = p->auto_crop_black_bars ? c->crop_tab : nullptr;
  const size_t ne = (size_t)p->eye_h * p->eye_w;
  StageTimer t(c, "p1_own");
  a.shard = 3;
  // the plane before the first frame of the batch is the context's filter state; inside the batch every frame reads its predecessor's slot
  HIPCHK(hipMemcpyAsync(c->slot_tdfp[slot0], c->tdf, ne * sizeof(float), hipMemcpyDeviceToDevice, s));
  HIPCHK(hipMemsetAsync(&c->slot_work[slot0], 0, (size_t)n * sizeof(vd_dev_work), s));   // tickets, fixed-point sums, select jobs of the frames
  HIPCHK(hipMemsetAsync(c->bhist, 0, (size_t)n * c->hist_bytes, s));
  vd_batch b;
  memset(&b, 0, sizeof b);
  b.n = n; b.w_main = c->work;
  const size_t nA = (size_t)VD_NJOBS * VD_NB_A;
  for (int j = 0; j < n; ++j) {
    vd_batch_frame& F = b.f[j];
    const int slot = slot0 + j;
    F.w = &c->slot_work[slot];
    F.histA = c->bhist + (size_t)j * (c->hist_bytes / sizeof(uint32_t)); F.histB = F.histA + nA;
    F.frame = frames_bgr[j]; F.depth = depths[j];
    F.rgb_eye = c->slot_rgb[slot]; F.tdf = c->slot_tdf[slot];
    F.tdf_prev = j == 0 ? c->slot_tdfp[slot0] : c->slot_tdf[slot - 1];
    c->slot_prev[slot] = F.tdf_prev;
    F.q_out = q_out_dev + 2 * j; F.shard_idx = step_idx0 + j;
  }
  vd_launch_chain_eye(s, b, depth_fmt, *p, a);
  HIPCHK(hipMemcpyAsync(c->tdf, c->slot_tdf[slot0 + n - 1], ne * sizeof(float), hipMemcpyDeviceToDevice, s));   // the filter state after the batch
  HIPCHK(hipGetLastError());
  return 0;
}
VD3D_EXPORT int vd3d_shard2_p1(vd3d_ctx* c, const uint8_t* frame_bgr, const void* depth, int depth_fmt, const vd3d_render_params* p,
                               int step_idx, int slot, float* q_out_dev) {
  return shard2_p1_impl(c, &frame_bgr, &depth, depth_fmt, p, step_idx, slot, 1, q_out_dev);
}
VD3D_EXPORT int vd3d_shard2_p1_batch(vd3d_ctx* c, const uint8_t* const* frames_bgr, const void* const* depths, int depth_fmt,
                                     const vd3d_render_params* p, int step_idx0, int slot0, int n, float* q_out_dev) {
  return shard2_p1_impl(c, frames_bgr, depths, depth_fmt, p, step_idx0, slot0, n, q_out_dev);
}
// The chunk hand-off of the plane state (one eye-size float32 plane per chunk boundary): export copies
// TemporalDepthFilter.prev_depth to a caller buffer (which the caller sends to the owner of the next chunk), import installs a
// received plane and marks it valid (valid = 0: "no previous frame", the state of a fresh clip).
VD3D_EXPORT int vd3d_tdf_plane_export(vd3d_ctx* c, float* dst_dev, int eye_h, int eye_w) {
  if (!c || !dst_dev || !c->tdf || eye_h != c->eye_h || eye_w != c->eye_w) return set_err(VD3D_E_INVALID, "bad argument (vd3d_shard_begin first)");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipMemcpyAsync(dst_dev, c->tdf, (size_t)eye_h * eye_w * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
  return 0;
}
VD3D_EXPORT int vd3d_tdf_plane_import(vd3d_ctx* c, const float* src_dev, int eye_h, int eye_w, int valid) {
  if (!c || !src_dev || !c->tdf || eye_h != c->eye_h || eye_w != c->eye_w) return set_err(VD3D_E_INVALID, "bad argument (vd3d_shard_begin first)");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipMemcpyAsync(c->tdf, src_dev, (size_t)eye_h * eye_w * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
  const int32_t v = valid ? 1 : 0;
  HIPCHK(hipMemcpyAsync(&c->work->st.tdf_valid, &v, sizeof v, hipMemcpyHostToDevice, c->stream));   // pageable 4-byte source: staged by the runtime
  return 0;
}
// P0 (only with auto_crop_black_bars), own frames: detect_black_bars + the per-frame aspect crop -> crop_out_dev[0..3] = {x, y, w, h}.
// The rectangles of ALL frames of the step are then all-gathered by the caller and installed with vd3d_shard2_set_crops.
VD3D_EXPORT int vd3d_shard2_p0(vd3d_ctx* c, const uint8_t* frame_bgr, const vd3d_render_params* p, int* crop_out_dev) {
  if (!c || !frame_bgr || !p || !crop_out_dev) return set_err(VD3D_E_INVALID, "NULL argument");
  if (!(p->target_ratio > 0.0) || p->src_w < 1 || p->src_h < 1) return set_err(VD3D_E_INVALID, "auto_crop_black_bars needs target_ratio > 0");
  HIPCHK(hipSetDevice(c->device));
  if (p->src_h > c->rowflag_cap) { HIPCHK(re_alloc(&c->rowflag, (size_t)p->src_h)); c->rowflag_cap = p->src_h; }
  vd_launch_autocrop(c->stream, frame_bgr, p->src_h, p->src_w, p->target_ratio, c->rowflag, c->work, crop_out_dev);
  c->crop_scalars_dirty = true;
  HIPCHK(hipGetLastError());
  return 0;
}
VD3D_EXPORT int vd3d_shard2_set_crops(vd3d_ctx* c, const int* crops_all_dev, int n) {
  if (!c || !crops_all_dev || n < 1 || n > VD_MAX_STEP || !c->crop_tab) return set_err(VD3D_E_INVALID, "bad argument (vd3d_shard_begin first)");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipMemcpyAsync(c->crop_tab, crops_all_dev, (size_t)n * 4 * sizeof(int), hipMemcpyDeviceToDevice, c->stream));
  c->crop_tab_set = true;
  return 0;
}
// R1: replay DepthPercentileEMA over the n frames of the step from the exchanged quantiles q_all_dev[n][2] (frame order)
VD3D_EXPORT int vd3d_shard2_r1(vd3d_ctx* c, const float* q_all_dev, int n) {
  if (!c || !q_all_dev || n < 1 || n > VD_MAX_STEP || !c->etab) return set_err(VD3D_E_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  vd_launch_shard2_r1(c->stream, c->work, q_all_dev, n, c->etab);
  HIPCHK(hipGetLastError());
  return 0;
}
// P3, own frames only: normalise, eye-res statistics, warp-res select chain, shaped depth plane, s1 -> measurements
// m_out_dev[0..3] = {sum1, sum2, sum_mad, (s_norm | s1 << 32)}; planes and shape constants stay in the slot.
// Batched form: the frames are independent once R1 has replayed the normalisation table, so n frames go through K3a ... K6 side by side
// (five launches per group of `VD3D_CHAIN_GROUP` frames; a group's curved-depth planes stay cache-resident between its launches).
static int g_chain_group = -1;   // frames per group of the batched P3 (VD3D_CHAIN_GROUP / vd3d_debug_tune)
static int shard2_p3_impl(vd3d_ctx* c, int slot0, int step_idx0, int n, const vd3d_render_params* p, long long* m_out_dev) {
  if (!c || !p || !m_out_dev || n < 1 || slot0 < 0 || slot0 + n > c->n_slots || step_idx0 < 0 || step_idx0 + n > VD_MAX_STEP) return set_err(VD3D_E_INVALID, "bad argument");
  if (n > c->n_batch) return set_err(VD3D_E_INVALID, "batch of %d frames > %d", n, c->n_batch);
  HIPCHK(hipSetDevice(c->device));
  hipStream_t s = c->stream;
  vd_stage_args a; vd3d_shift_params sp;
  int rc;
  if ((rc = shard2_args(c, p, &a, &sp))) return rc;
  if ((rc = check_shift_params(&sp, p->warp_h, p->warp_w))) return rc;
  a.shard = 3;
  for (int j = 0; j < n; ++j) if ((rc = wait_slot(c, slot0 + j))) return rc;
  int& group = g_chain_group;
  if (group < 0) { const char* e = getenv("VD3D_CHAIN_GROUP"); group = e ? atoi(e) : VD_MAX_BATCH; }
  if (group < 1) group = 1;
  if (group > VD_MAX_BATCH) group = VD_MAX_BATCH;
  StageTimer t(c, "p3_own");
  const size_t nA = (size_t)VD_NJOBS * VD_NB_A;
  for (int j0 = 0; j0 < n; j0 += group) {
    const int g = n - j0 < group ? n - j0 : group;
    HIPCHK(hipMemsetAsync(c->bhist, 0, (size_t)g * c->hist_bytes, s));   // the select jobs of these frames start from empty histograms
    vd_batch b;
    memset(&b, 0, sizeof b);
    b.n = g; b.w_main = c->work;
    for (int j = 0; j < g; ++j) {
      vd_batch_frame& F = b.f[j];
      const int slot = slot0 + j0 + j;
      F.w = &c->slot_work[slot];
      F.histA = c->bhist + (size_t)j * (c->hist_bytes / sizeof(uint32_t)); F.histB = F.histA + nA;
      F.tdf = c->slot_tdf[slot]; F.dn = c->slot_dn[slot]; F.dn_prev = c->slot_prev[slot];
      F.dc = c->bdc[j]; F.D = c->slot_D[slot];
      F.m_out = m_out_dev + 4 * (j0 + j); F.shard_idx = step_idx0 + j0 + j;
    }
    vd_launch_chain_work(s, b, 1, p->eye_h, p->eye_w, p->warp_h, p->warp_w, (float)sp.depth_pop_mid, (float)sp.depth_pop_gamma, a);
  }
  HIPCHK(hipGetLastError());
  return 0;
}
VD3D_EXPORT int vd3d_shard2_p3(vd3d_ctx* c, int slot, int step_idx, const vd3d_render_params* p, long long* m_out_dev) {
  return shard2_p3_impl(c, slot, step_idx, 1, p, m_out_dev);
}
VD3D_EXPORT int vd3d_shard2_p3_batch(vd3d_ctx* c, int slot0, int step_idx0, int n, const vd3d_render_params* p, long long* m_out_dev) {
  return shard2_p3_impl(c, slot0, step_idx0, n, p, m_out_dev);
}
// R2: replay every remaining tracker over the n frames of the step from the exchanged measurements m_all_dev[n][4] (frame order);
// own_slot_host[t] = slot of frame t on this rank or -1.  Afterwards vd3d_shard_pixels(slot) renders the own frames.
VD3D_EXPORT int vd3d_shard2_r2(vd3d_ctx* c, const long long* m_all_dev, const int* own_slot_host, const uint8_t* blank_host_or_null, int n,
                               const vd3d_render_params* p) {
  if (!c || !m_all_dev || !own_slot_host || !p || n < 1 || n > VD_MAX_STEP) return set_err(VD3D_E_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  vd_stage_args a; vd3d_shift_params sp;
  { int rca = shard2_args(c, p, &a, &sp); if (rca) return rca; }
  for (int t = 0; t < n; ++t) { int rc = wait_slot(c, own_slot_host[t]); if (rc) return rc; }
  StageTimer t(c, "replay");
  vd_launch_shard2_r2(c->stream, c->work, m_all_dev, c->etab, own_slot_host, blank_host_or_null, n, c->slot_work, a);
  HIPCHK(hipGetLastError());
  return 0;
}

// ---- diagnostics / tests --------------------------------------------------------------------------
VD3D_EXPORT int vd3d_debug_planes(vd3d_ctx* c, float** D, float** S, uint8_t** L, uint8_t** R, float** rgb_eye, float** dn_cur) {
  if (D) *D = c->D;
  if (S) *S = c->S;
  if (L) *L = c->L;
  if (R) *R = c->R;
  if (rgb_eye) *rgb_eye = c->rgb_eye;
  if (dn_cur) *dn_cur = c->dn[c->dn_cur ^ 1];  // the plane the last render_frame wrote
  return 0;
}

VD3D_EXPORT int vd3d_quantiles(vd3d_ctx* c, const float* plane, int64_t n, const float* q_host, int nq, float* out_host) {
  if (!c || !plane || !q_host || !out_host || n < 1 || nq < 1) return set_err(VD3D_E_INVALID, "bad argument");
  if (n >= (1ll << 31)) return set_err(VD3D_E_INVALID, "n too large");
  HIPCHK(hipSetDevice(c->device));
  for (int i = 0; i < nq; i += 2) {
    vd_stage_args a;
    memset(&a, 0, sizeof a);
    // reuse the work-res quantile job through the stretch_lo/hi slots
    a.shift.depth_stretch_lo = q_host[i];
    a.shift.depth_stretch_hi = q_host[i + 1 < nq ? i + 1 : i];
    HIPCHK(hipMemsetAsync(c->histA, 0, c->hist_bytes, c->stream));
    vd_launch_hist_eye_d(c->stream, false, plane, n, c->work, c->histA, c->histB);
    a.stage = VD_ST_AQ; vd_launch_scalar_stage(c->stream, c->work, c->histA, c->histB, a);
    vd_launch_hist_eye_d(c->stream, true, plane, n, c->work, c->histA, c->histB);
    a.stage = VD_ST_BQ; vd_launch_scalar_stage(c->stream, c->work, c->histA, c->histB, a);
    float res[2];
    HIPCHK(hipMemcpyAsync(res, &c->work->fs.q_lo, 2 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    out_host[i] = res[0];
    if (i + 1 < nq) out_host[i + 1] = res[1];
  }
  return 0;
}

VD3D_EXPORT int vd3d_subject_depth(vd3d_ctx* c, const float* plane, int H, int W, float* out_host) {
  if (!c || !plane || !out_host || H < 1 || W < 1) return set_err(VD3D_E_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  vd_stage_args a;
  memset(&a, 0, sizeof a);
  HIPCHK(hipMemsetAsync(c->histA, 0, c->hist_bytes, c->stream));
  vd_launch_hist_work_s1(c->stream, false, plane, H, W, c->work, c->histA, c->histB);
  a.stage = VD_ST_A2; vd_launch_scalar_stage(c->stream, c->work, c->histA, c->histB, a);
  vd_launch_hist_work_s1(c->stream, true, plane, H, W, c->work, c->histA, c->histB);
  a.stage = VD_ST_BS; vd_launch_scalar_stage(c->stream, c->work, c->histA, c->histB, a);
  HIPCHK(hipMemcpyAsync(out_host, &c->work->fs.s1, sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

// a23: heal_missing_pixels, core/render_3d.py:431-459 (optional stage; the reference's loop never calls it)
VD3D_EXPORT int vd3d_heal_missing_pixels(vd3d_ctx* c, const float* warped_chw, const float* original_chw, const float* edge_mask_or_null,
                                         int H, int W, double heal_strength, float* out_chw) {
  if (!c || !warped_chw || !original_chw || !out_chw || H < 1 || W < 1) return set_err(VD3D_E_INVALID, "bad argument");
  if (out_chw == warped_chw || out_chw == original_chw) return set_err(VD3D_E_INVALID, "out may not alias an input (3x3 neighbourhood reads)");
  if ((long long)H * W * 3 >= (1ll << 31)) return set_err(VD3D_E_INVALID, "frame too large");
  HIPCHK(hipSetDevice(c->device));
  StageTimer t(c, "heal");
  vd_launch_heal(c->stream, warped_chw, original_chw, edge_mask_or_null, H, W, (float)heal_strength, out_chw);
  HIPCHK(hipGetLastError());
  return 0;
}

// cv2.resize(u8, (dw, dh), interpolation=cv2.INTER_CUBIC): a24 with an explicit inference size (core/render_depth.py:1917) and the
// up-scale stage's size changes (core/merged_pipeline.py:260-264)
VD3D_EXPORT int vd3d_resize_cubic_u8(vd3d_ctx* c, const uint8_t* src, int sh, int sw, int cn, uint8_t* dst, int dh, int dw) {
  if (!c || !src || !dst || sh < 1 || sw < 1 || dh < 1 || dw < 1 || src == dst) return set_err(VD3D_E_INVALID, "bad argument");
  if ((long long)sh * sw * cn >= (1ll << 31) || (long long)dh * dw * cn >= (1ll << 31)) return set_err(VD3D_E_INVALID, "image too large");
  HIPCHK(hipSetDevice(c->device));
  StageTimer t(c, "resize_cubic");
  if (!vd_launch_resize_cubic_u8(c->stream, src, sh, sw, cn, dst, dh, dw)) return set_err(VD3D_E_UNSUPPORTED, "resize_cubic_u8: %d channels (1 or 3)", cn);
  HIPCHK(hipGetLastError());
  return 0;
}

VD3D_EXPORT int vd3d_resize_area_u8(vd3d_ctx* c, const uint8_t* src_bgr, int sh, int sw, uint8_t* dst_bgr, int dh, int dw) {
  if (!c || !src_bgr || !dst_bgr || sh < 1 || sw < 1 || dh < 1 || dw < 1 || src_bgr == dst_bgr) return set_err(VD3D_E_INVALID, "bad argument");
  if ((long long)sh * sw * 3 >= (1ll << 31) || (long long)dh * dw * 3 >= (1ll << 31)) return set_err(VD3D_E_INVALID, "image too large");
  HIPCHK(hipSetDevice(c->device));
  StageTimer t(c, "resize_area");
  if (!vd_launch_resize_area_u8(c->stream, src_bgr, sh, sw, dst_bgr, dh, dw))
    return set_err(VD3D_E_UNSUPPORTED, "resize_area_u8: %dx%d -> %dx%d exceeds the tap budget (ratio <= 10)", sw, sh, dw, dh);
  HIPCHK(hipGetLastError());
  return 0;
}

// cv2.resize(src, (dw, dh)) -- INTER_LINEAR, OpenCV's default -- on uint8 BGR: format_3d_output's VR branch (core/render_3d.py:846-849)
VD3D_EXPORT int vd3d_resize_linear_u8(vd3d_ctx* c, const uint8_t* src_bgr, int sh, int sw, uint8_t* dst_bgr, int dh, int dw) {
  if (!c || !src_bgr || !dst_bgr || sh < 1 || sw < 1 || dh < 1 || dw < 1 || src_bgr == dst_bgr) return set_err(VD3D_E_INVALID, "bad argument");
  if ((long long)sh * sw * 3 >= (1ll << 31) || (long long)dh * dw * 3 >= (1ll << 31)) return set_err(VD3D_E_INVALID, "image too large");
  HIPCHK(hipSetDevice(c->device));
  vd_launch_resize_linear_u8(c->stream, src_bgr, sh, sw, dst_bgr, dh, dw);
  HIPCHK(hipGetLastError());
  return 0;
}

// format_3d_output(left, right, fmt), core/render_3d.py:837-860, on two uint8 BGR eyes of h x w: Half-/Full-SBS = side by side, VR = both eyes
// resized to 1440 x 1600 (INTER_LINEAR) side by side, Red-Cyan Anaglyph = generate_anaglyph_3d, Passive Interlaced = even rows left / odd rows
// right.  out: [h][2w][3] (SBS), [1600][2880][3] (VR), [h][w][3] (anaglyph, interlaced).  The mux runs through k_sharp_mux with the identity
// sharpen kernel (kc = 1, kn = 0: exact) and a 1:1 fit.
VD3D_EXPORT int vd3d_format_3d_output(vd3d_ctx* c, const uint8_t* left_bgr, const uint8_t* right_bgr, int h, int w, int format, uint8_t* out_bgr) {
  if (!c || !left_bgr || !right_bgr || !out_bgr || h < 1 || w < 1) return set_err(VD3D_E_INVALID, "bad argument");
  if (format < 0 || format > VD3D_FMT_INTERLACED) return set_err(VD3D_E_INVALID, "unknown format %d", format);
  HIPCHK(hipSetDevice(c->device));
  int rc = join_pixels(c);
  if (rc) return rc;
  const uint8_t *L = left_bgr, *R = right_bgr;
  int eh = h, ew = w;
  if (format == VD3D_FMT_VR && (h != 1600 || w != 1440)) {
    // two 1440 x 1600 eyes in a scratch of this entry point's own (grow-only): the context's warp-res planes belong to the render path, whose
    // slots may be in flight at another size (an ensure_work() here would shrink them under a queued pixel pass)
    const size_t ne = (size_t)1600 * 1440 * 3;
    if (c->fmt_cap < 2 * ne) { HIPCHK(re_alloc(&c->fmt_eyes, 2 * ne)); c->fmt_cap = 2 * ne; }
    vd_launch_resize_linear_u8(c->stream, left_bgr, h, w, c->fmt_eyes, 1600, 1440);
    vd_launch_resize_linear_u8(c->stream, right_bgr, h, w, c->fmt_eyes + ne, 1600, 1440);
    L = c->fmt_eyes; R = c->fmt_eyes + ne; eh = 1600; ew = 1440;
  }
  vd3d_render_params p;
  vd3d_render_params_default(&p);
  p.format = format; p.warp_w = ew; p.warp_h = eh; p.fit_w = ew; p.fit_h = eh;
  p.out_w = (format == VD3D_FMT_HALF_SBS || format == VD3D_FMT_FULL_SBS || format == VD3D_FMT_VR) ? 2 * ew : ew; p.out_h = eh;
  vd_finish_consts fc;
  memset(&fc, 0, sizeof fc);
  fc.sharp_kn = 0.f; fc.sharp_kc = 1.f;
  // identity fit: format_3d_output stacks / interleaves / mixes the eyes as they are (np.hstack, row slices, the anaglyph matrix) -- NOT the
  // render loop's pad_to_aspect_ratio, whose int(aspect * h) truncates to w - 1 for about one eye size in twenty (61x7, 1920x804, ...)
  vd_launch_sharp_mux(c->stream, L, R, p, fc, out_bgr, 0, true);
  HIPCHK(hipGetLastError());
  return 0;
}

// preprocess_esr / postprocess_esr / blend_images, core/merged_pipeline.py:219-236
VD3D_EXPORT int vd3d_esr_preprocess(vd3d_ctx* c, int dtype, const uint8_t* frame_bgr, long long pitch_bytes, int h, int w, int channels_last,
                                    void* out_rgb) {
  if (!c || !frame_bgr || !out_rgb || h < 1 || w < 1 || pitch_bytes < 3ll * w) return set_err(VD3D_E_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  StageTimer t(c, "esr_pre");
  if (!vd_launch_esr_pre(c->stream, dtype, frame_bgr, pitch_bytes, h, w, channels_last ? 1 : 0, out_rgb)) return set_err(VD3D_E_INVALID, "bad dtype %d", dtype);
  HIPCHK(hipGetLastError());
  return 0;
}

VD3D_EXPORT int vd3d_esr_postprocess(vd3d_ctx* c, const float* pred_rgb, int h, int w, int channels_last, int cy, int cx, int ch, int cw,
                                     uint8_t* out_bgr, long long pitch_bytes) {
  if (!c || !pred_rgb || !out_bgr || h < 1 || w < 1 || cy < 0 || cx < 0 || ch < 1 || cw < 1 || cy + ch > h || cx + cw > w || pitch_bytes < 3ll * cw)
    return set_err(VD3D_E_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  StageTimer t(c, "esr_post");
  vd_launch_esr_post(c->stream, pred_rgb, h, w, channels_last ? 1 : 0, cy, cx, ch, cw, out_bgr, pitch_bytes);
  HIPCHK(hipGetLastError());
  return 0;
}

// run_rife's glue, core/merged_pipeline.py:195-218
VD3D_EXPORT int vd3d_rife_preprocess(vd3d_ctx* c, int dtype, const uint8_t* frame1_bgr, const uint8_t* frame2_bgr, int h, int w, int channels_last,
                                     void* out6) {
  if (!c || !frame1_bgr || !frame2_bgr || !out6 || h < 1 || w < 1) return set_err(VD3D_E_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  if (!vd_launch_rife_pre(c->stream, dtype, frame1_bgr, frame2_bgr, h, w, channels_last ? 1 : 0, out6)) return set_err(VD3D_E_INVALID, "bad dtype %d", dtype);
  HIPCHK(hipGetLastError());
  return 0;
}
VD3D_EXPORT int vd3d_rife_postprocess(vd3d_ctx* c, const float* pred3, int h, int w, int channels_last, uint8_t* out_bgr) {
  if (!c || !pred3 || !out_bgr || h < 1 || w < 1) return set_err(VD3D_E_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  vd_launch_rife_post(c->stream, pred3, h, w, channels_last ? 1 : 0, out_bgr);
  HIPCHK(hipGetLastError());
  return 0;
}

VD3D_EXPORT int vd3d_add_weighted_u8(vd3d_ctx* c, const uint8_t* a, double alpha, const uint8_t* b, double beta, double gamma, long long n,
                                     uint8_t* out) {
  if (!c || !a || !b || !out || n < 1) return set_err(VD3D_E_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  StageTimer t(c, "add_weighted");
  vd_launch_add_weighted_u8(c->stream, a, (float)alpha, b, (float)beta, (float)gamma, n, out);
  HIPCHK(hipGetLastError());
  return 0;
}

// a24: depth-net prediction [B][ph][pw] float32 -> uint8 depth planes [B][H][W] (bicubic post-process + per-frame min-max)
VD3D_EXPORT int vd3d_depth_handoff(vd3d_ctx* c, const float* pred, int B, int ph, int pw, int H, int W, int invert, uint8_t* out_gray) {
  if (!c || !pred || !out_gray || B < 1 || ph < 1 || pw < 1 || H < 1 || W < 1) return set_err(VD3D_E_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  if (B > c->mm_cap) { HIPCHK(re_alloc(&c->mm, (size_t)3 * B)); c->mm_cap = B; }
  StageTimer t(c, "handoff");
  vd_launch_depth_handoff(c->stream, pred, B, ph, pw, H, W, invert ? 1 : 0, c->mm, out_gray);
  HIPCHK(hipGetLastError());
  return 0;
}

VD3D_EXPORT int vd3d_depth_preprocess(vd3d_ctx* c, const uint8_t* frames_bgr, int B, int H, int W, int th, int tw,
                                      const float* mean3_host, const float* std3_host, int dtype, void* out_nhwc) {
  if (dtype != VD3D_DT_BF16 && dtype != VD3D_DT_F32) return set_err(VD3D_E_INVALID, "bad dtype %d", dtype);
  if (!c || !frames_bgr || !mean3_host || !std3_host || !out_nhwc || B < 1 || H < 1 || W < 1 || th < 1 || tw < 1)
    return set_err(VD3D_E_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  StageTimer t(c, "depth_prep");
  if (!vd_launch_depth_prep(c->stream, frames_bgr, B, H, W, th, tw, mean3_host, std3_host, dtype, out_nhwc))
    return set_err(VD3D_E_UNSUPPORTED, "depth_preprocess: %dx%d -> %dx%d exceeds the antialias tap budget", W, H, tw, th);
  HIPCHK(hipGetLastError());
  return 0;
}

VD3D_EXPORT int vd3d_add_layernorm(vd3d_ctx* c, int dtype, const void* x, const void* y_or_null, const void* gamma, const void* beta,
                                   float eps, int64_t rows, int cols, void* out_sum, void* out_norm) {
  if (dtype != VD3D_DT_BF16 && dtype != VD3D_DT_F32) return set_err(VD3D_E_INVALID, "bad dtype %d", dtype);
  if (!c || !x || !gamma || !beta || !out_norm || rows < 1 || (y_or_null && !out_sum)) return set_err(VD3D_E_INVALID, "bad argument");
  if (!vd_launch_add_layernorm(c->stream, dtype, x, y_or_null, gamma, beta, eps, (long long)rows, cols, out_sum, out_norm))
    return set_err(VD3D_E_UNSUPPORTED, "add_layernorm: cols %d not in {384,768,1024}", cols);
  HIPCHK(hipGetLastError());
  return 0;
}

VD3D_EXPORT int64_t vd3d_conv3x3_x2_weight_bytes(int Cin, int Cout) { return (int64_t)vd_conv3x3_x2_weight_bytes(Cin, Cout); }

VD3D_EXPORT int vd3d_conv3x3_x2_pack_weights(vd3d_ctx* c, const float* W, int Cin, int Cout, void* image) {
  if (!c || !W || !image) return set_err(VD3D_E_INVALID, "bad argument");
  if (!vd_launch_conv3x3_x2_pack(c->stream, W, Cin, Cout, image))
    return set_err(VD3D_E_UNSUPPORTED, "conv3x3_x2: C_in %d must be a positive multiple of 16 and C_out %d one of 64, 128", Cin, Cout);
  HIPCHK(hipGetLastError());
  return 0;
}

VD3D_EXPORT int vd3d_conv3x3_x2(vd3d_ctx* c, const float* X, int B, int H, int W, int Cin, const void* w_image, int Cout, float* Y) {
  if (!c || !X || !w_image || !Y) return set_err(VD3D_E_INVALID, "bad argument");
  if (!vd_launch_conv3x3_x2(c->stream, X, B, H, W, Cin, w_image, Cout, Y))
    return set_err(VD3D_E_UNSUPPORTED, "conv3x3_x2: C_in %d (multiple of 16), C_out %d (64 | 128), B %d <= 65535, 16-byte aligned input and image", Cin, Cout, B);
  HIPCHK(hipGetLastError());
  return 0;
}

VD3D_EXPORT int64_t vd3d_attention_x3_workspace_bytes(int B, int T, int H, int D, int mode) { return (int64_t)vd_attn_x3_workspace_bytes(B, T, H, D, mode); }

VD3D_EXPORT int vd3d_attention_x3(vd3d_ctx* c, const float* qkv, int B, int T, int H, int D, float scale, int mode, void* workspace, int64_t workspace_bytes, float* out) {
  if (!c || !qkv || !workspace || !out) return set_err(VD3D_E_INVALID, "bad argument");
  const long long need = vd_attn_x3_workspace_bytes(B, T, H, D, mode);
  if (need < 0) return set_err(VD3D_E_UNSUPPORTED, "attention_x3: head size %d not built (64), an unknown mode %d or an empty shape (B %d T %d H %d)", D, mode, B, T, H);
  if (workspace_bytes < need) return set_err(VD3D_E_INVALID, "attention_x3: workspace of %lld bytes, %lld needed", (long long)workspace_bytes, need);
  if (!vd_launch_attn_x3(c->stream, qkv, B, T, H, D, scale, workspace, out, mode))
    return set_err(VD3D_E_UNSUPPORTED, "attention_x3: qkv / workspace / out must be 16-byte aligned, B * H <= 65535");
  HIPCHK(hipGetLastError());
  return 0;
}

VD3D_EXPORT int64_t vd3d_gemm_x3_weight_bytes(int N, int K, int mode) { return (int64_t)vd_gemm_x3_weight_bytes(N, K, mode); }

VD3D_EXPORT int vd3d_gemm_x3_pack_weights(vd3d_ctx* c, const float* W, int N, int K, int mode, void* image) {
  if (!c || !W || !image) return set_err(VD3D_E_INVALID, "bad argument");
  if (!vd_launch_gemm_x3_pack_w(c->stream, W, N, K, image, mode)) return set_err(VD3D_E_UNSUPPORTED, "gemm_x3: K %d must be a positive multiple of 16 (N %d), mode %d in {0, 1}", K, N, mode);
  HIPCHK(hipGetLastError());
  return 0;
}

VD3D_EXPORT int vd3d_gemm_x3(vd3d_ctx* c, const float* X, int64_t M, int K, const void* w_image, int N, int mode, const float* bias_or_null, int epilogue, float* Y) {
  if (!c || !X || !w_image || !Y || M < 1) return set_err(VD3D_E_INVALID, "bad argument");
  if (epilogue != VD3D_GEMM_EPI_NONE && epilogue != VD3D_GEMM_EPI_GELU) return set_err(VD3D_E_INVALID, "gemm_x3: unknown epilogue %d", epilogue);
  if (!vd_launch_gemm_x3(c->stream, X, (long long)M, K, w_image, N, bias_or_null, epilogue, Y, mode))
    return set_err(VD3D_E_UNSUPPORTED, "gemm_x3: K %d must be a positive multiple of 16, mode %d in {0, 1}, X and the weight image 16-byte aligned", K, mode);
  HIPCHK(hipGetLastError());
  return 0;
}

VD3D_EXPORT int vd3d_upsample_bilinear_nhwc(vd3d_ctx* c, int dtype, const void* in, void* out, int B, int ih, int iw, int oh, int ow, int C) {
  if (dtype != VD3D_DT_BF16 && dtype != VD3D_DT_F32) return set_err(VD3D_E_INVALID, "bad dtype %d", dtype);
  if (!c || !in || !out || B < 1 || ih < 1 || iw < 1) return set_err(VD3D_E_INVALID, "bad argument");
  if (!vd_launch_upsample_bilinear_nhwc(c->stream, dtype, in, out, B, ih, iw, oh, ow, C))
    return set_err(VD3D_E_UNSUPPORTED, "upsample_bilinear_nhwc: C not a multiple of 8 (bf16) / 4 (f32) or output smaller than 2x2");
  HIPCHK(hipGetLastError());
  return 0;
}

// DPT neck / head glue in float32 (vd3d_netops.hip): NHWC maps of n_pix pixels x C channels
VD3D_EXPORT int vd3d_nhwc_bias_act_f32(vd3d_ctx* c, const float* y, const float* bias_or_null, const float* r1_or_null, const float* r2_or_null, int relu,
                                       int64_t n_pix, int C, float* out, float* relu_out_or_null) {
  if (!c || !y || !out || n_pix < 1 || C < 4) return set_err(VD3D_E_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  if (!vd_launch_bias_act_f32(c->stream, y, bias_or_null, r1_or_null, r2_or_null, relu, (long long)n_pix, C, out, relu_out_or_null))
    return set_err(VD3D_E_UNSUPPORTED, "nhwc_bias_act: C %d not a multiple of 4", C);
  HIPCHK(hipGetLastError());
  return 0;
}
VD3D_EXPORT int vd3d_upsample_bilinear_bias_nhwc_f32(vd3d_ctx* c, const float* in, const float* bias, float* out, int B, int ih, int iw, int oh, int ow, int C) {
  if (!c || !in || !bias || !out || B < 1 || ih < 1 || iw < 1) return set_err(VD3D_E_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  if (!vd_launch_upsample_bilinear_bias_nhwc_f32(c->stream, in, bias, out, B, ih, iw, oh, ow, C))
    return set_err(VD3D_E_UNSUPPORTED, "upsample_bilinear_bias_nhwc: C not a multiple of 4 or output smaller than 2x2");
  HIPCHK(hipGetLastError());
  return 0;
}
VD3D_EXPORT int vd3d_dpt_head_tail_f32(vd3d_ctx* c, const float* y, const float* b2, const float* w3, float b3, float scale, int64_t n_pix, int C, float* out) {
  if (!c || !y || !b2 || !w3 || !out || n_pix < 1) return set_err(VD3D_E_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  if (!vd_launch_head_tail_f32(c->stream, y, b2, w3, b3, scale, (long long)n_pix, C, out))
    return set_err(VD3D_E_UNSUPPORTED, "dpt_head_tail: C %d not in {16, 32, 64}", C);
  HIPCHK(hipGetLastError());
  return 0;
}

VD3D_EXPORT int vd3d_preview_image(vd3d_ctx* c, int type, const uint8_t* left_bgr, const uint8_t* right_bgr, int h, int w, uint8_t* out_bgr) {
  if (!c || !left_bgr || !right_bgr || !out_bgr || h < 1 || w < 1) return set_err(VD3D_E_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  if (!vd_launch_preview(c->stream, type, left_bgr, right_bgr, h, w, out_bgr))
    return set_err(VD3D_E_UNSUPPORTED, "preview type %d (colour-mapped heat-maps / arrow overlay need OpenCV's tables and rasteriser)", type);
  HIPCHK(hipGetLastError());
  return 0;
}

// the colour-mapped previews of generate_preview_image (core/preview_utils.py:42-66); lut_bgr_dev: the caller's 256 x 3 BGR table in HBM
VD3D_EXPORT int vd3d_preview_heatmap(vd3d_ctx* c, int type, const float* shift_map, int h, int w, const uint8_t* lut_bgr_dev, uint8_t* out_bgr) {
  if (!c || !shift_map || !lut_bgr_dev || !out_bgr || h < 1 || w < 1) return set_err(VD3D_E_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  if (c->mm_cap < 1) { HIPCHK(re_alloc(&c->mm, (size_t)3)); c->mm_cap = 1; }
  if (!vd_launch_preview_heatmap(c->stream, type, shift_map, h, w, lut_bgr_dev, c->mm, out_bgr))
    return set_err(VD3D_E_UNSUPPORTED, "heat-map type %d (0 shift, 1 |shift|, 2 clipped, 3 feather mask)", type);
  HIPCHK(hipGetLastError());
  return 0;
}

// body layer of the up-scale network on the matrix cores (vd3d_conv.hip)
VD3D_EXPORT int vd3d_conv3x3_c64_f16(vd3d_ctx* c, const void* x_nhwc, int H, int W, const void* w_frag, const float* bias, const float* slope_or_null,
                                     void* y_nhwc) {
  if (!c || !x_nhwc || !w_frag || !bias || !y_nhwc || H < 1 || W < 1 || x_nhwc == y_nhwc) return set_err(VD3D_E_INVALID, "bad argument");
  if ((long long)H * W * 128 >= (1ll << 32)) return set_err(VD3D_E_INVALID, "conv3x3_c64_f16: activation larger than 4 GB");
  if (((uintptr_t)x_nhwc | (uintptr_t)y_nhwc | (uintptr_t)w_frag | (uintptr_t)bias | (uintptr_t)slope_or_null) & 15)
    return set_err(VD3D_E_INVALID, "conv3x3_c64_f16: pointers must be 16-byte aligned");
  HIPCHK(hipSetDevice(c->device));
  StageTimer t(c, "conv3x3");
  if (!vd_launch_conv3x3_c64_f16(c->stream, x_nhwc, H, W, w_frag, bias, slope_or_null, y_nhwc)) return set_err(VD3D_E_HIP, "conv3x3_c64_f16: LDS attribute");
  HIPCHK(hipGetLastError());
  return 0;
}

VD3D_EXPORT int vd3d_conv3x3_head_f16(vd3d_ctx* c, const void* x_nhwc3, int H, int W, const float* w27x64, const float* bias, const float* slope_or_null,
                                      void* y_nhwc64) {
  if (!c || !x_nhwc3 || !w27x64 || !bias || !y_nhwc64 || H < 1 || W < 1) return set_err(VD3D_E_INVALID, "conv3x3_head_f16: bad argument");
  if ((long long)H * W * 128 >= (1ll << 32)) return set_err(VD3D_E_INVALID, "conv3x3_head_f16: activation larger than 4 GB");
  if ((reinterpret_cast<uintptr_t>(y_nhwc64) | reinterpret_cast<uintptr_t>(w27x64)) & 15) return set_err(VD3D_E_INVALID, "conv3x3_head_f16: y and w must be 16-byte aligned");
  HIPCHK(hipSetDevice(c->device));
  StageTimer t(c, "conv_head");
  vd_launch_conv3x3_head_f16(c->stream, x_nhwc3, H, W, w27x64, bias, slope_or_null, y_nhwc64);
  HIPCHK(hipGetLastError());
  return 0;
}

VD3D_EXPORT int vd3d_esr_tail_f32(vd3d_ctx* c, const void* t_nhwc64, const void* x_nhwc3, int H, int W, int r, float* out_planar) {
  if (!c || !t_nhwc64 || !x_nhwc3 || !out_planar || H < 1 || W < 1 || (r != 2 && r != 4)) return set_err(VD3D_E_INVALID, "esr_tail_f32: bad argument (r = 2 or 4)");
  if (reinterpret_cast<uintptr_t>(t_nhwc64) & 15) return set_err(VD3D_E_INVALID, "esr_tail_f32: t must be 16-byte aligned");
  HIPCHK(hipSetDevice(c->device));
  StageTimer t(c, "esr_tail");
  vd_launch_esr_tail_f32(c->stream, t_nhwc64, x_nhwc3, H, W, r, out_planar);
  HIPCHK(hipGetLastError());
  return 0;
}

// optional NV12 wire format at the frame I/O boundary (vd3d_nv12.hip)
VD3D_EXPORT int vd3d_nv12_to_bgr(vd3d_ctx* c, const uint8_t* y_plane, long long y_pitch, const uint8_t* uv_plane, long long uv_pitch, int h, int w,
                                 uint8_t* out_bgr) {
  if (!c || !y_plane || !uv_plane || !out_bgr || h < 2 || w < 2 || (h & 1) || (w & 1) || y_pitch < w || uv_pitch < w)
    return set_err(VD3D_E_INVALID, "bad argument (NV12 needs even h and w)");
  HIPCHK(hipSetDevice(c->device));
  StageTimer t(c, "nv12_in");
  vd_launch_nv12_to_bgr(c->stream, y_plane, uv_plane, h, w, y_pitch, uv_pitch, out_bgr);
  HIPCHK(hipGetLastError());
  return 0;
}
VD3D_EXPORT int vd3d_bgr_to_nv12(vd3d_ctx* c, const uint8_t* bgr, int h, int w, uint8_t* y_plane, long long y_pitch, uint8_t* uv_plane,
                                 long long uv_pitch) {
  if (!c || !y_plane || !uv_plane || !bgr || h < 2 || w < 2 || (h & 1) || (w & 1) || y_pitch < w || uv_pitch < w)
    return set_err(VD3D_E_INVALID, "bad argument (NV12 needs even h and w)");
  HIPCHK(hipSetDevice(c->device));
  StageTimer t(c, "nv12_out");
  vd_launch_bgr_to_nv12(c->stream, bgr, h, w, y_plane, uv_plane, y_pitch, uv_pitch);
  HIPCHK(hipGetLastError());
  return 0;
}

VD3D_EXPORT int vd3d_preview_arrows(vd3d_ctx* c, const uint8_t* left_bgr, const float* shift_map, int h, int w, uint8_t* out_bgr) {
  if (!c || !left_bgr || !shift_map || !out_bgr || h < 1 || w < 1 || left_bgr == out_bgr) return set_err(VD3D_E_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  vd_launch_preview_arrows(c->stream, left_bgr, shift_map, h, w, out_bgr);
  HIPCHK(hipGetLastError());
  return 0;
}

VD3D_EXPORT int vd3d_detect_black_bars(vd3d_ctx* c, const uint8_t* frame_bgr, int h, int w, int* top_host, int* bottom_host) {
  if (!c || !frame_bgr || !top_host || !bottom_host || h < 1 || w < 1) return set_err(VD3D_E_INVALID, "bad argument");
  HIPCHK(hipSetDevice(c->device));
  if (h > c->rowflag_cap) { HIPCHK(re_alloc(&c->rowflag, (size_t)h)); c->rowflag_cap = h; }
  vd_launch_autocrop(c->stream, frame_bgr, h, w, 16.0 / 9.0, c->rowflag, c->work);
  c->crop_scalars_dirty = true;
  int32_t tb[2];
  HIPCHK(hipMemcpyAsync(tb, &c->work->fs.crop_top, sizeof tb, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  *top_host = tb[0]; *bottom_host = tb[1];
  return 0;
}

VD3D_EXPORT int vd3d_stream_copy(vd3d_ctx* c, const void* src, void* dst, size_t bytes) {
  HIPCHK(hipSetDevice(c->device));
  StageTimer t(c, "stream_copy");
  vd_launch_stream_copy(c->stream, src, dst, bytes);
  HIPCHK(hipGetLastError());
  return 0;
}

VD3D_EXPORT int vd3d_torch_math(vd3d_ctx* c, int op, const float* x, float param, float* out, long long n) {
  if (!c) return set_err(VD3D_E_INVALID, "NULL context");
  if (op < 0 || op > 2 || n < 0 || (n > 0 && (!x || !out))) return set_err(VD3D_E_INVALID, "vd3d_torch_math: op 0..2, n >= 0, device pointers");
  HIPCHK(hipSetDevice(c->device));
  if (n) vd_launch_torch_math(c->stream, op, x, param, out, n);
  HIPCHK(hipGetLastError());
  return 0;
}

VD3D_EXPORT int vd3d_torch_math_aten(vd3d_ctx* c, int op, const float* x, double param, float* out, long long n, int aten_threads) {
  if (!c) return set_err(VD3D_E_INVALID, "NULL context");
  if (op < 0 || op > 1 || n < 0 || n >= (1ll << 32) || (n > 0 && (!x || !out)))
    return set_err(VD3D_E_INVALID, "vd3d_torch_math_aten: op 0..1, 0 <= n < 2^32, device pointers");
  HIPCHK(hipSetDevice(c->device));
  if (n) vd_launch_torch_math_aten(c->stream, op, x, param, out, n, aten_threads);
  HIPCHK(hipGetLastError());
  return 0;
}

// development probe (tools/probe_step.py): 0 = workgroup divisor of the batched chain kernels, 1 = frames per P3 group
// Process-wide launch-policy switches.  The PRODUCT library keeps the two ROUTE selectors the parity tests use to prove that two code paths give the same
// bytes (3: fused / unfused finishing routes, 4: the long way round feather_strength 0) -- plain ints written before any frame is in flight, read per call;
// everything else (tile heights, tile orders, the parked persistent kernels, chain grouping) exists only in development libraries built with
// -DVD3D_DEV_KNOBS (tools/build_ab.sh dev -DVD3D_DEV_KNOBS), and so does the VD3D_TUNE environment variable (round 6).
// which == -1: query -- 0 when the development knobs are compiled in, VD3D_E_UNSUPPORTED when not.
VD3D_EXPORT int vd3d_debug_tune(int which, int value) {
  if (which == 3) { g_fused_fit = value; return 0; }
  if (which == 4) { g_feather0_long = value; return 0; }
#ifdef VD3D_DEV_KNOBS
  if (which == -1) return 0;
  if (which == 0) vd_set_batch_grid_div(value);
  else if (which == 1) g_chain_group = value;
  else if (which == 2) vd_set_warp_pre_th(value);
  else if (which == 5) vd_set_conv_mode(value);
  else if (which == 6) vd_set_finish_persist(value);
  else if (which == 7) vd_set_warp_nofeather_th(value);
  else if (which == 8) vd_set_finish_xcd(value);
  else if (which == 9) vd_set_warp_order(value);
  else return set_err(VD3D_E_INVALID, "vd3d_debug_tune: unknown knob %d", which);
  return 0;
#else
  (void)value;
  if (which == -1 || (which >= 0 && which <= 9)) return set_err(VD3D_E_UNSUPPORTED, "vd3d_debug_tune(%d): development knob, this library was built without -DVD3D_DEV_KNOBS", which);
  return set_err(VD3D_E_INVALID, "vd3d_debug_tune: unknown knob %d", which);
#endif
}

VD3D_EXPORT int vd3d_set_profiling(vd3d_ctx* c, int enable) {
  vd3d_sync(c);
  c->profiling = enable != 0;
  if (enable) c->acc.clear();
  return 0;
}
// average milliseconds per call of the named stage since profiling was enabled; -1 if never seen.  Synchronises.
VD3D_EXPORT float vd3d_last_stage_ms(vd3d_ctx* c, const char* stage) {
  vd3d_sync(c);
  auto it = c->acc.find(stage);
  if (it == c->acc.end() || it->second.second == 0) return -1.f;
  return (float)(it->second.first / (double)it->second.second);
}
VD3D_EXPORT long vd3d_stage_calls(vd3d_ctx* c, const char* stage) {
  vd3d_sync(c);
  auto it = c->acc.find(stage);
  return it == c->acc.end() ? 0 : it->second.second;
}
