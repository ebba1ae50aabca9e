// vd3d_kernels.h -- launch prototypes + small PODs shared by the .hip translation units.
#pragma once
#include <hip/hip_runtime.h>

#include "vd3d_work.h"

enum { VD_ST_A0 = 0, VD_ST_B0, VD_ST_A1, VD_ST_B1, VD_ST_A2, VD_ST_B2,
       VD_ST_AQ, VD_ST_BQ, VD_ST_BS };  // AQ/BQ: generic quantile pair, BS: bare subject depth (test entry points)
#define PL_KMAX_HOST 129  // largest blur_ksize of the fallback pool kernel (LDS tile (16 + k - 1) x (64 + k - 1) floats = 108 KB at 129; the fused
                          // warp kernel takes k <= ~17 at 4K, larger windows run e2 / pool / warp as separate launches)
#define DF_RMAX_HOST 15   // largest Gaussian radius of the DOF kernel

#define VD_ETAB 8
struct vd_stage_args {
  int stage;
  int have_eye;        // 1: full render_frame chain (eye-res stages exist); 0: bare pixel_shift_cuda
  int W, H;            // warp size
  long long n_eye;     // eye_h*eye_w
  long long n_crop;    // centre-crop population of compute_dynamic_parallax_scale
  double ipd_factor;
  int shard;           // 0 = sequential frame; 3 = own frame of a sharded step (measurements only, no tracker touched)
  int shard_idx;       // frame index inside the sharded step
  float* q_out;        // shard == 3: {q_lo, q_hi} of this frame (device, 2 floats)
  long long* m_out;    // shard == 3: {sum1, sum2, sum_mad, (s_norm | s1 << 32)} of this frame (device, 4 x int64)
  const int* crop_tab; // shard 3/4 with auto_crop_black_bars: per-frame crop rectangles {x, y, w, h} of the step (device), else NULL
  const float* etab;   // shard == 3: per-frame normalisation table of the step, VD_ETAB floats per entry:
                       // entry t = {ema_lo, ema_den, collapse, have_prev, ema_hi} in force BEFORE frame t
  int dbg;             // development probes (timing only, results are garbage): bit0 = skip the last-workgroup scalar stage, bit1 = skip ticket +
                       // fences, bit2 = skip the LDS -> global histogram flush, bit3 = skip the LDS histogram adds
  int blank;           // skip_blank_frames hit (core/render_3d.py:1278-1281): no ipd scaling, no FloatingWindowTracker / focal update
  // vd3d_render_params::aten_sum_threads > 0 (vd3d_atensum.hip): torch.mean's float32 summation order for the dynamic parallax scale and the motion metric
  int aten_threads, aten_n_small, aten_n_big, aten_nr_crop, aten_nr_mad;
  const int* aten_plan;   // device: [pieces][4] = {plane offset, length, range, job}
  float* aten_scratch;    // device: [frames of the batch][pieces] piece sums
  vd3d_shift_params shift;
};

// One frame of a (possibly batched) select-chain launch: every chain kernel takes a vd_batch by value and workgroup row blockIdx.y (K1: a
// loop inside the workgroup, the plane EMA being a recurrence over the frames) works on frame f[blockIdx.y].  The sequential entry points pass
// a batch of one (w = the context's control block, planes = the context's); the sharded step passes the own frames of the step, each with
// its slot's control block, histograms and planes, so that the per-launch costs (launch gap, last-workgroup scan, scalar stage) are paid
// once per step and the scans of the frames run side by side.
#define VD_MAX_BATCH 16
struct vd_batch_frame {
  vd_dev_work* w;          // control block of this frame (tickets, select jobs, fixed-point sums, derived constants)
  uint32_t* histA;         // [VD_NJOBS][VD_NB_A]
  uint32_t* histB;         // [VD_NJOBS][VD_MAX_T][VD_NB_B] + coarse level
  const uint8_t* frame;    // K1: source frame (u8 BGR) and depth plane
  const void* depth;
  float* rgb_eye;          // K1 writes [3][eh][ew]
  float* tdf;              // filtered depth plane of THIS frame (K1 writes; K2 / K3a read)
  const float* tdf_prev;   // filtered plane of the frame before (K1: the EMA's previous value; may alias tdf: in-place update)
  float* dn;               // normalised eye-res plane (K3a writes; K3b / K4 read); bare pixel_shift_cuda: the caller's depth plane
  const float* dn_prev;    // K3a MAD: previous normalised plane (sequential) / previous FILTERED plane (measure-replay sharding)
  float* dc;               // curved depth, warp resolution (K3b writes; K4 / K5 read)
  float* D;                // shaped depth, warp resolution (K5 writes; K6 and the pixel pass read)
  float* q_out;            // shard == 3: {q_lo, q_hi} of this frame
  long long* m_out;        // shard == 3: the frame's measurement record
  int shard_idx, pad_;
};
struct vd_batch {
  int n, pad_;
  vd_dev_work* w_main;     // the context's control block: TemporalDepthFilter validity before frame 0; marked valid by K2's last workgroup
  vd_batch_frame f[VD_MAX_BATCH];
};

#ifdef __HIPCC__
#include "vd3d_dev.h"
// enhance_curvature(.,0.08) + clamp (core/render_3d.py:599-601) of the bilinearly resized depth (:596)
VD_DEV float vd_curved_depth(const float* __restrict__ dn, int ih, int iw, int H, int W, int y, int x) {
  float d;
  if (ih == H && iw == W) {
    d = dn[(size_t)y * W + x];
  } else {
    vd_tap ty = vd_interp_tap(ih, H, y), tx = vd_interp_tap(iw, W, x);
    const float* r0 = dn + (size_t)ty.i0 * iw;
    const float* r1 = dn + (size_t)ty.i1 * iw;
    d = vd_bilerp(r0[tx.i0], r0[tx.i1], r1[tx.i0], r1[tx.i1], tx.w0, tx.w1, ty.w0, ty.w1);
  }
  const float xx = vd_lin11(W, x), yy = vd_lin11(H, y);
  const float curv = 1.f - (xx * xx + yy * yy);
  return vd_clamp(d + curv * (float)0.08, 0.f, 1.f);
}

// frame_to_tensor / depth_to_tensor (:135-143) + centre crop (:1236-1248) + resize to the eye size (:1262-1263) for one
// eye-res pixel, and the TemporalDepthFilter update (:225-229) of that pixel in place.  Returns the new filtered value.
VD_DEV float vd_depth_at(const void* depth, int fmt, size_t idx) {
  if (fmt == VD3D_DEPTH_F32) return ((const float*)depth)[idx];
  if (fmt == VD3D_DEPTH_GRAY_U8) return vd_u8_unit((float)((const uint8_t*)depth)[idx]);   // == v / 255.0f for every uint8 v
  const uint8_t* p = (const uint8_t*)depth + idx * 3;  // cv2.COLOR_BGR2GRAY fixed point
  int g = (p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + 8192) >> 14;
  return vd_u8_unit((float)g);
}
VD_DEV float vd_ingest_pixel(const uint8_t* __restrict__ frame, const void* __restrict__ depth, int fmt, const vd3d_render_params& p,
                             int tdf_valid, float* __restrict__ rgb_eye, const float* tdf_prev, float* tdf, int ey, int ex) {
  const vd_tap ty = vd_interp_tap(p.crop_h, p.eye_h, ey), tx = vd_interp_tap(p.crop_w, p.eye_w, ex);
  const size_t i00 = (size_t)(ty.i0 + p.crop_y) * p.src_w + (tx.i0 + p.crop_x);
  const size_t i01 = (size_t)(ty.i0 + p.crop_y) * p.src_w + (tx.i1 + p.crop_x);
  const size_t i10 = (size_t)(ty.i1 + p.crop_y) * p.src_w + (tx.i0 + p.crop_x);
  const size_t i11 = (size_t)(ty.i1 + p.crop_y) * p.src_w + (tx.i1 + p.crop_x);
  const size_t ne = (size_t)p.eye_h * p.eye_w, o = (size_t)ey * p.eye_w + ex;
  // the N-thread ATen mode (round 5): which of ATen's two bilinear kernels resizes the 3-channel frame / the 1-channel depth at this eye size
  const bool pm_c = vd_interp_premult(3, p.eye_h, p.eye_w, p.aten_sum_threads), pm_d = vd_interp_premult(1, p.eye_h, p.eye_w, p.aten_sum_threads);
  if (frame) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {  // output plane c = R,G,B ; source byte 2-c
      const int sc = 2 - c;
      float p00 = vd_u8_unit((float)frame[i00 * 3 + sc]), p01 = vd_u8_unit((float)frame[i01 * 3 + sc]);
      float p10 = vd_u8_unit((float)frame[i10 * 3 + sc]), p11 = vd_u8_unit((float)frame[i11 * 3 + sc]);
      rgb_eye[c * ne + o] = vd_bilerp_sel(pm_c, p00, p01, p10, p11, tx.w0, tx.w1, ty.w0, ty.w1);
    }
  }
  const float cur = vd_bilerp_sel(pm_d, vd_depth_at(depth, fmt, i00), vd_depth_at(depth, fmt, i01), vd_depth_at(depth, fmt, i10),
                                  vd_depth_at(depth, fmt, i11), tx.w0, tx.w1, ty.w0, ty.w1);
  const float prev = tdf_valid ? tdf_prev[o] : cur;   // tdf_prev may be tdf itself (in-place) or the plane this thread wrote one frame ago
  const float nv = 0.5f * prev + (float)(1 - 0.5) * cur;
  tdf[o] = nv;
  return nv;
}
#endif

// ---- vd3d_select.hip (fused chain)
void vd_launch_shard2_r1(hipStream_t s, vd_dev_work* w, const float* q_all, int n, float* etab);
void vd_launch_shard2_r2(hipStream_t s, vd_dev_work* w, const long long* m_all, const float* etab, const int* own_slot_host,
                         const uint8_t* blank_host_or_null, int n, vd_dev_work* slot_work, const vd_stage_args& a);
void vd_launch_autocrop(hipStream_t s, const uint8_t* frame, int h, int wd, double target_ratio, uint32_t* rowflag, vd_dev_work* w,
                        int* crop_out = nullptr);   // crop_out: optional device int[4] copy of the rectangle (sharded steps)
void vd_set_batch_grid_div(int v);   // tuning probe (vd3d_debug_tune)
// K1 + K2 over the frames of b in frame order (one launch each)
void vd_launch_chain_eye(hipStream_t s, const vd_batch& b, int fmt, const vd3d_render_params& p, const vd_stage_args& a);
// K3a (have_eye) + K3b + K4 + K5 + K6 over the frames of b side by side (one launch each, grid.y = b.n)
void vd_launch_chain_work(hipStream_t s, const vd_batch& b, int have_eye, int ih, int iw, int H, int W, float mid, float gamma,
                          const vd_stage_args& a);

// ---- vd3d_select.hip
void vd_launch_hist_eye_d(hipStream_t s, bool passB, const float* tdf, long long n, vd_dev_work* w, uint32_t* histA, uint32_t* histB);
void vd_launch_hist_work_s1(hipStream_t s, bool passB, const float* D, int H, int W, vd_dev_work* w, uint32_t* histA, uint32_t* histB);
void vd_launch_scalar_stage(hipStream_t s, vd_dev_work* w, const uint32_t* histA, const uint32_t* histB, const vd_stage_args& a);

// ---- vd3d_planes.hip
struct vd_finish_consts {
  int nlev;                 // number of blurred levels (4 when dof on, 0 when off)
  int ksz[4];               // kernel sizes
  float kern[4][32];        // 1-D Gaussian weights per level (torchvision _get_gaussian_kernel1d)
  float w2[4][81];          // levels of <= 9 taps: the dense K x K weights fl(k1[i] * k1[j]), row-major with pitch K.  The host keeps a copy
                            // of this table in device memory (vd3d_ctx::d_w2); E1 reads it through scalar loads, so the FMAs take the
                            // weight as a scalar operand: no weight registers, no per-row weight products in the kernel
  float fw, imax;           // focus_width + 1e-6, (N-1) - 1e-6
  float sat, con, bri;
  float sharp_kn, sharp_kc; // normalised sharpen taps
};
// launch constants of the shift plane (k_shift, and W1 when it computes the shift of its own tile: round 6)
struct vd_shift_consts {
  float mid, fg, mg, bg, fgm, bgm, pb, half_width, fs, ma, mb;
  int edge;
};
vd_shift_consts vd_shift_consts_of(const vd3d_shift_params& p, int W);
void vd_launch_shift(hipStream_t s, const float* D, int H, int W, const vd_dev_work* w, const vd3d_shift_params& p, float* S);
// W1 without feathering can compute the shift plane of its own tile (no halo of S is needed there) instead of reading the plane k_shift wrote: `work` = the frame's
// control block (layer shifts, zero-parallax, clamp, convergence), `S_out` = where to keep the plane for a caller who wants it, or NULL.
struct vd_shift_fold { const vd_dev_work* work; vd3d_shift_params sp; float* S_out; };
bool vd_warp_fold_ok(int ih, int iw, int H, int W, const vd3d_shift_params& warp_p, const vd3d_shift_params& shift_p);
void vd_launch_e2(hipStream_t s, const float* D, const float* S, int H, int W, float fs, float* e2L, float* e2R);
void vd_launch_pool(hipStream_t s, const float* e2L, const float* e2R, int H, int W, int k, float* bL, float* bR);
void vd_launch_warp(hipStream_t s, const float* rgb, int ih, int iw, const float* S, const float* bL, const float* bR, int H, int W,
                    int feather, uint8_t* L, uint8_t* R);
bool vd_launch_warp_fused(hipStream_t s, const float* rgb, int ih, int iw, const float* D, const float* S, int H, int W,
                          const vd3d_shift_params& p, uint8_t* L, uint8_t* R, const float* E2 = nullptr, const vd_shift_fold* fold = nullptr);
bool vd_warp_fused_ok(int ih, int iw, int H, int W, const vd3d_shift_params& p);
// k_e2w (vd3d_warp.hip): gradient mask plane E2[H][W][2] (left, right eye) of feather_shift_edges from the shaped depth and the shift plane
void vd_launch_e2w(hipStream_t s, const float* D, const float* S, int H, int W, float feather_strength, float* E2);
void vd_set_warp_pre_th(int th);
void vd_set_warp_order(int v);
void vd_set_warp_nofeather_th(int th);
#ifdef __cplusplus
#include <vector>
// vd3d_atensum.hip: piece plan of the two torch.mean sums for one eye size and torch thread count (host vectors; the caller uploads them)
bool vd_aten_plan_build(int eh, int ew, int T, std::vector<int>& pieces_flat, int* n_small, int* n_big, int* nr_crop, int* nr_mad);
#endif
void vd_launch_aten_sums(hipStream_t s, const vd_batch& b, const vd_stage_args& a);
void vd_set_finish_persist(int k);
void vd_set_finish_xcd(int on);   // vd3d_finish.hip: 0 = one tile per workgroup, k > 0 = persistent fused finishing kernel, k workgroups per CU
void vd_set_conv_mode(int v);   // vd3d_conv.hip: < 0 one tile per workgroup (rounds 2 - 4), >= 0 persistent kernel with a phase skew of v microseconds
void vd_launch_dof_grade(hipStream_t s, const uint8_t* eye_in, const float* dn, int eh, int ew, int H, int W,
                         const vd_finish_consts& fc, const vd_dev_work* w, float focal_override, int use_override,
                         int bar_width, int bar_side, uint8_t* eye_out, int dense = 0, const float* dense_wtab = nullptr);
void vd_launch_dof_grade_dense(hipStream_t s, const uint8_t* L_in, const uint8_t* R_in, const float* dn, int eh, int ew, int H, int W,
                               const vd_finish_consts& fc, const vd_dev_work* w, float focal_override, int use_override,
                               int bar_width, int bar_side, uint8_t* L_out, uint8_t* R_out, const float* wtab);
#define VD_D4_WTAB_FLOATS (4 * 31 * 32)   // dense weight table of k_dof_grade4: [level 4][row 31][pitch 32] = fl(k1[i] * k1[j])
// sharpen + fit + mux of two graded eyes with the fused finishing kernel's epilogue (vd3d_finish.hip); false: not its fit (k_sharp_mux then)
bool vd_launch_sharp_fit(hipStream_t s, const uint8_t* gL, const uint8_t* gR, const vd3d_render_params& p, const vd_finish_consts& fc, uint8_t* out);
// presharp_pitch > 0: gL / gR hold sharpened eyes already (row pitch in pixels): fit + mux only
// identity_fit: the eyes fill the fit rectangle 1:1 (warp == fit; vd3d_format_3d_output) -- no pad_to_aspect_ratio geometry
void vd_launch_sharp_mux(hipStream_t s, const uint8_t* gL, const uint8_t* gR, const vd3d_render_params& p,
                         const vd_finish_consts& fc, uint8_t* out, int presharp_pitch = 0, bool identity_fit = false);
void vd_launch_stream_copy(hipStream_t s, const void* src, void* dst, size_t bytes);
// F.interpolate(bilinear, align_corners=False) of C float32 planes; premult: ATen's premultiplied-weight kernel (vd_bilerp_pm) instead of the nested form
void vd_launch_interp_planes(hipStream_t s, const float* src, int C, int ih, int iw, float* dst, int oh, int ow, int premult);
void vd_launch_torch_math_aten(hipStream_t s, int op, const float* x, double p, float* out, long long n, int threads);
void vd_launch_torch_math(hipStream_t s, int op, const float* x, float p, float* out, long long n);
void vd_launch_blank_eye(hipStream_t s, const uint8_t* src, int h, int w, const vd_dev_work* wk, uint8_t* dst);

bool vd_launch_preview(hipStream_t s, int type, const uint8_t* L, const uint8_t* R, int h, int w, uint8_t* out);
// ---- vd3d_depthprep.hip
bool vd_launch_depth_prep(hipStream_t s, const uint8_t* frames, int B, int H, int W, int th, int tw, const float mean[3],
                          const float stdv[3], int dtype, void* out_nhwc);
// ---- vd3d_netops.hip
// vd3d_gemm.hip: the transformer linears as a split-bf16 (bf16x3, six products) GEMM with float32 accumulation
long long vd_gemm_x3_weight_bytes(int N, int K, int mode);
bool vd_launch_gemm_x3_pack_w(hipStream_t s, const float* W, int N, int K, void* img, int mode);
bool vd_launch_gemm_x3(hipStream_t s, const float* X, long long M, int K, const void* wimg, int N, const float* bias, int epilogue, float* Y, int mode);
// vd3d_conv2.hip: 3 x 3 convolution (stride 1, padding 1, no bias) of float32 NHWC maps in the fp16x2 arithmetic
long long vd_conv3x3_x2_weight_bytes(int Cin, int Cout);
bool vd_launch_conv3x3_x2_pack(hipStream_t s, const float* W, int Cin, int Cout, void* img);
bool vd_launch_conv3x3_x2(hipStream_t s, const float* X, int B, int H, int W, int Cin, const void* wimg, int Cout, float* Y);
// vd3d_attn.hip: softmax(Q K^T scale) V with both products as split-bf16 MFMA work
long long vd_attn_x3_workspace_bytes(int B, int T, int H, int D, int mode);
bool vd_launch_attn_x3(hipStream_t s, const float* qkv, int B, int T, int H, int D, float scale, void* ws, float* out, int mode);
bool vd_launch_add_layernorm(hipStream_t s, int dtype, const void* x, const void* y, const void* gamma, const void* beta, float eps,
                             long long rows, int cols, void* out_sum, void* out_norm);
bool vd_launch_upsample_bilinear_nhwc(hipStream_t s, int dtype, const void* in, void* out, int B, int ih, int iw, int oh, int ow, int C);
bool vd_launch_bias_act_f32(hipStream_t s, const float* y, const float* bias, const float* r1, const float* r2, int relu, long long n_pix, int C,
                            float* out, float* relu_out);
bool vd_launch_upsample_bilinear_bias_nhwc_f32(hipStream_t s, const float* in, const float* bias, float* out, int B, int ih, int iw, int oh, int ow, int C);
bool vd_launch_head_tail_f32(hipStream_t s, const float* y, const float* b2, const float* w3, float b3, float scale, long long n_pix, int C, float* out);
// ---- vd3d_handoff.hip
void vd_launch_depth_handoff(hipStream_t s, const float* pred, int B, int ph, int pw, int H, int W, int invert, uint32_t* mm,
                             uint8_t* out);

// ---- vd3d_finish.hip
bool vd_launch_finish_fused(hipStream_t s, const uint8_t* L, const uint8_t* R, const float* dn, int eh, int ew,
                            const vd3d_render_params& p, const vd_finish_consts& fc, const vd_dev_work* w, float focal,
                            int use_override, int bar_w, int bar_s, uint8_t* out, int dense, const float* w2_dev);

// ---- vd3d_conv.hip
bool vd_launch_conv3x3_c64_f16(hipStream_t s, const void* x, int H, int W, const void* wfrag, const float* bias, const float* slope_or_null, void* y);
void vd_launch_conv3x3_head_f16(hipStream_t s, const void* x, int H, int W, const float* w27, const float* bias, const float* slope_or_null, void* y);
void vd_launch_esr_tail_f32(hipStream_t s, const void* t, const void* x, int H, int W, int r, float* out);
// ---- vd3d_nv12.hip
void vd_launch_nv12_to_bgr(hipStream_t s, const uint8_t* y, const uint8_t* uv, int h, int w, long long y_pitch, long long uv_pitch, uint8_t* out);
void vd_launch_bgr_to_nv12(hipStream_t s, const uint8_t* bgr, int h, int w, uint8_t* y, uint8_t* uv, long long y_pitch, long long uv_pitch);
// ---- vd3d_heatmap.hip
bool vd_launch_preview_heatmap(hipStream_t s, int type, const float* shift, int h, int w, const uint8_t* lut_dev, uint32_t* mm, uint8_t* out);
void vd_launch_preview_arrows(hipStream_t s, const uint8_t* left, const float* shift, int h, int w, uint8_t* out);
// ---- vd3d_upscale.hip
bool vd_launch_resize_cubic_u8(hipStream_t s, const uint8_t* src, int sh, int sw, int cn, uint8_t* dst, int dh, int dw);
bool vd_launch_resize_area_u8(hipStream_t s, const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw);
void vd_launch_resize_linear_u8(hipStream_t s, const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw);
bool vd_launch_rife_pre(hipStream_t s, int dtype, const uint8_t* f1, const uint8_t* f2, int h, int w, int hwc, void* out);
void vd_launch_rife_post(hipStream_t s, const float* pred, int h, int w, int hwc, uint8_t* dst);
bool vd_launch_esr_pre(hipStream_t s, int dtype, const uint8_t* src, long long pitch, int h, int w, int hwc, void* out);
void vd_launch_esr_post(hipStream_t s, const float* pred, int h, int w, int hwc, int cy, int cx, int ch, int cw, uint8_t* dst, long long pitch);
void vd_launch_add_weighted_u8(hipStream_t s, const uint8_t* a, float alpha, const uint8_t* b, float beta, float gamma, long long n, uint8_t* out);

// ---- vd3d_heal.hip
void vd_launch_heal(hipStream_t s, const float* warped, const float* orig, const float* edge_or_null, int H, int W, float hs, float* out);
