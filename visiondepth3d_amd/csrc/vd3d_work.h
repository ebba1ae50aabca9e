// vd3d_work.h -- device-resident control block shared by the kernels and the host API.
#pragma once
#include <stdint.h>

#include "../../include/vd3d.h"

#define VD_NB_A 16384u  // float bit-pattern prefix (top 16 bits) bins; values in [0,1] use 0..0x3F80
#define VD_NB_B 65536u  // low 16 bits
#define VD_NB_BC 256u   // coarse level of the low-16 histogram (low16 >> 8): scan B walks 256 + 256 bins, not 65536
#define VD_MAX_T 4      // distinct target prefixes per select job
#define VD_NJOBS 5
#define VD_MAX_STEP 512  // frames per sharded step (world * frames-per-rank)

// select jobs
enum {
  VD_J_EYE_Q = 0,   // quantile(.02,.98) of clamp(filtered depth), eye-res, all pixels      (a5)
  VD_J_EYE_SUBJ = 1, // estimate_subject_depth(normalised eye-res depth)                    (a15/a18)
  VD_J_WORK_Q = 2,  // quantile(.05,.95) of curved depth, warp-res, all pixels               (a10)
  VD_J_WORK_S0 = 3, // estimate_subject_depth(curved depth)                                  (a11 step 3)
  VD_J_WORK_S1 = 4  // estimate_subject_depth(shaped depth)                                  (a11 step 5)
};

struct vd_sel_ctl {
  uint64_t count;          // population size
  uint64_t ranks[4];       // requested 0-based ranks
  uint64_t rank_rem[4];    // rank inside its target prefix bin
  uint32_t nranks;
  uint32_t ntargets;
  uint32_t tprefix[VD_MAX_T];
  uint32_t rank_t[4];      // rank -> target slot
  float val[4];            // resolved order statistics
  float w[2];              // quantile lerp weights (rank - floor(rank))
  uint32_t peak_bin;       // 64-bin histogram arg-max (first max)
  uint32_t fallback;       // subject: fewer than 20 valid samples
};

struct vd_dev_work {
  vd3d_state st;
  vd3d_frame_scalars fs;
  vd_sel_ctl job[VD_NJOBS];
  uint32_t ticket[8];             // last-workgroup arrival counters of the fused chain kernels (self re-arming)
  long long sum1, sum2, sum_mad;  // 2^-40 fixed-point sums (centre crop of the normalised depth; |d_t - d_{t-1}|)
  // constants derived by the scalar stages, consumed by the plane kernels
  float ema_lo, ema_den;
  int32_t collapse;
  int32_t shp_stretch;
  float shp_lo, shp_den, shp_subj_s;
  float fg, mg, bg;        // per-frame layer shifts narrowed to float32 (after smoother * dyn_scale * ipd)
  double fg_d, mg_d, bg_d;
  float zpo_f;
  int32_t have_zpo;
  float msn;
  float conv;
  int32_t have_conv;
  float focal;
  int32_t bar_width, bar_side;
  int32_t acrop[4];        // crop_x, crop_y, crop_w, crop_h of THIS frame when auto_crop_black_bars is on (k_autocrop)
  float aten_sum_mean, aten_sum_mad;   // vd3d_render_params::aten_sum_threads > 0: torch.sum's float32 value of the centre crop / of |d_t - d_{t-1}| (vd3d_atensum.hip)
};
