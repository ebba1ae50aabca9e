// vd3d_select.hip -- exact order statistics on HBM-resident float32 planes (gfx950).
//
// Replaces the reference's torch.quantile / torch.median / torch.histc call sites
// (core/render_3d.py:154-170,249-250,536-537; 51 % of the reference's CPU time is aten::sort there).
// No sort: values live in [0,1], so the float bit pattern is a monotone uint32 key.
//   pass A  : 16-bit-prefix histogram (16257 live bins) privatised in LDS (65 KB / job), plain returnless ds_add
//             (measured faster than ballot / shuffle aggregation), one global atomic per non-empty bin per workgroup;
//   scan A  : one workgroup prefix-scans the bins, derives the requested ranks on device
//             (torch.quantile's float32 rank arithmetic, the lower-median index, the 64-bin histc arg-max
//             which is an exact aggregation of the prefix bins) and records <= 4 target prefixes;
//   pass B  : elements whose prefix is a target add to a 65536-bin histogram of their low 16 bits
//             (global atomics: <= 2 wave-leader aggregation rounds -- a constant plane costs N/64 atomics -- then one plain
//             atomic per remaining lane: the hits are one value band of the plane, usually with distinct low bits);
//   scan B  : locates each rank inside its target bin -> the exact float, then the scalar stage runs.
// Everything stays on the device: no host synchronisation anywhere in the frame.  The file also holds the fused select chain
// (K0-K6: passes riding on the producer kernels, scans + scalar stages run by the last workgroup) and the replay kernels of
// the two frame-sharding protocols.
#include "vd3d_dev.h"
#include "vd3d_kernels.h"

#define NBL 16272  // LDS bins per job (>= 0x3F80 + 1)

// ------------------------------------------------------------------------------------------------
// pass A / pass B over a value functor.  F::get(i, v, in_all, in_crop)
// ------------------------------------------------------------------------------------------------
template <class F, int J_ALL, int J_CROP>
__global__ __launch_bounds__(1024) void k_hist_a(F f, uint32_t* __restrict__ histA) {
  __shared__ uint32_t h0[J_ALL >= 0 ? NBL : 1];
  __shared__ uint32_t h1[J_CROP >= 0 ? NBL : 1];
  const int tid = threadIdx.x;
  for (int b = tid; b < NBL; b += 1024) {
    if (J_ALL >= 0) h0[b] = 0;
    if (J_CROP >= 0) h1[b] = 0;
  }
  __syncthreads();
  const long long n = f.count();
  const long long stride = (long long)gridDim.x * 1024;
  for (long long base = (long long)blockIdx.x * 1024; base < n; base += stride) {
    const long long i = base + tid;
    float v = 0.f;
    bool in_all = false, in_crop = false;
    if (i < n) f.get(i, v, in_all, in_crop);
    unsigned key = __float_as_uint(v) >> 16;
    key = key < (NBL - 1) ? key : (NBL - 1);
    if (J_ALL >= 0) vd_lds_hist_add(h0, key, in_all);
    if (J_CROP >= 0) vd_lds_hist_add(h1, key, in_crop);
  }
  __syncthreads();
  for (int b = tid; b < NBL; b += 1024) {
    if (J_ALL >= 0) { uint32_t c = h0[b]; if (c) atomicAdd(&histA[(size_t)J_ALL * VD_NB_A + b], c); }
    if (J_CROP >= 0) { uint32_t c = h1[b]; if (c) atomicAdd(&histA[(size_t)J_CROP * VD_NB_A + b], c); }
  }
}

// coarse (low16>>8) counts live behind the fine arena: histBC = histB + VD_NJOBS*VD_MAX_T*VD_NB_B, [job][target][256]
VD_DEV uint32_t* vd_histbc(uint32_t* histB) { return histB + (size_t)VD_NJOBS * VD_MAX_T * VD_NB_B; }
VD_DEV const uint32_t* vd_histbc(const uint32_t* histB) { return histB + (size_t)VD_NJOBS * VD_MAX_T * VD_NB_B; }

template <class F, int J_ALL, int J_CROP>
__global__ __launch_bounds__(256) void k_hist_b(F f, const vd_dev_work* __restrict__ w, uint32_t* __restrict__ histB) {
  const long long n = f.count();
  const long long stride = (long long)gridDim.x * 256;
  uint32_t nt0 = 0, nt1 = 0, tp0[VD_MAX_T], tp1[VD_MAX_T];
  if (J_ALL >= 0) { nt0 = w->job[J_ALL].ntargets; for (int t = 0; t < VD_MAX_T; ++t) tp0[t] = w->job[J_ALL].tprefix[t]; }
  if (J_CROP >= 0) { nt1 = w->job[J_CROP].ntargets; for (int t = 0; t < VD_MAX_T; ++t) tp1[t] = w->job[J_CROP].tprefix[t]; }
  for (long long base = (long long)blockIdx.x * 256; base < n; base += stride) {
    const long long i = base + threadIdx.x;
    float v = 0.f;
    bool in_all = false, in_crop = false;
    if (i < n) f.get(i, v, in_all, in_crop);
    const unsigned bits = __float_as_uint(v);
    const unsigned pre = bits >> 16, low = bits & 0xffffu;
    if (J_ALL >= 0) {
      bool hit = false; unsigned key = 0;
      for (uint32_t t = 0; t < nt0; ++t) if (in_all && pre == tp0[t]) { hit = true; key = (t << 16) | low; }
      vd_hist_add_agg(histB + (size_t)J_ALL * VD_MAX_T * VD_NB_B, key, hit);
      vd_hist_add_agg(vd_histbc(histB) + (size_t)J_ALL * VD_MAX_T * VD_NB_BC, key >> 8, hit);
    }
    if (J_CROP >= 0) {
      bool hit = false; unsigned key = 0;
      for (uint32_t t = 0; t < nt1; ++t) if (in_crop && pre == tp1[t]) { hit = true; key = (t << 16) | low; }
      vd_hist_add_agg(histB + (size_t)J_CROP * VD_MAX_T * VD_NB_B, key, hit);
      vd_hist_add_agg(vd_histbc(histB) + (size_t)J_CROP * VD_MAX_T * VD_NB_BC, key >> 8, hit);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// value functors
// ------------------------------------------------------------------------------------------------
VD_DEV bool vd_in_subject_crop(int y, int x, int H, int W, float v) {  // core/render_3d.py:154-157
  return y >= H / 5 && y < H * 4 / 5 && x >= W / 5 && x < W * 4 / 5 && v > 0.05f && v < 0.95f;
}

struct FEyeD {  // clamp(filtered depth): DepthPercentileEMA input (:247)
  const float* tdf; long long n;
  VD_DEV long long count() const { return n; }
  VD_DEV void get(long long i, float& v, bool& in_all, bool& in_crop) const {
    v = vd_clamp(tdf[i], 0.f, 1.f); in_all = true; in_crop = false;
  }
};
struct FPlaneSubj {  // a stored plane, masked centre crop membership (estimate_subject_depth)
  const float* p; int H, W;
  VD_DEV long long count() const { return (long long)H * W; }
  VD_DEV void get(long long i, float& v, bool& in_all, bool& in_crop) const {
    v = p[i];
    const int y = (int)((unsigned)i / (unsigned)W), x = (int)((unsigned)i - (unsigned)y * (unsigned)W);
    in_all = false; in_crop = vd_in_subject_crop(y, x, H, W, v);
  }
};
struct FWorkDc {  // curved depth at warp resolution, recomputed from the eye-res plane
  const float* dn; int ih, iw, H, W;
  VD_DEV long long count() const { return (long long)H * W; }
  VD_DEV void get(long long i, float& v, bool& in_all, bool& in_crop) const {
    const int y = (int)((unsigned)i / (unsigned)W), x = (int)((unsigned)i - (unsigned)y * (unsigned)W);
    v = vd_curved_depth(dn, ih, iw, H, W, y, x);
    in_all = true; in_crop = vd_in_subject_crop(y, x, H, W, v);
  }
};

static inline int hist_grid(long long n, int per_block, int cap) {
  long long g = (n + per_block - 1) / per_block;
  if (g > cap) g = cap;
  return (int)(g < 1 ? 1 : g);
}

void vd_launch_hist_eye_d(hipStream_t s, bool passB, const float* tdf, long long n, vd_dev_work* w, uint32_t* histA, uint32_t* histB) {
  FEyeD f{tdf, n};
  if (!passB) hipLaunchKernelGGL((k_hist_a<FEyeD, VD_J_EYE_Q, -1>), dim3(hist_grid(n, 4096, 512)), dim3(1024), 0, s, f, histA);
  else hipLaunchKernelGGL((k_hist_b<FEyeD, VD_J_EYE_Q, -1>), dim3(hist_grid(n, 1024, 2048)), dim3(256), 0, s, f, w, histB);
}
void vd_launch_hist_work_s1(hipStream_t s, bool passB, const float* D, int H, int W, vd_dev_work* w, uint32_t* histA, uint32_t* histB) {
  FPlaneSubj f{D, H, W};
  long long n = (long long)H * W;
  if (!passB) hipLaunchKernelGGL((k_hist_a<FPlaneSubj, -1, VD_J_WORK_S1>), dim3(hist_grid(n, 4096, 512)), dim3(1024), 0, s, f, histA);
  else hipLaunchKernelGGL((k_hist_b<FPlaneSubj, -1, VD_J_WORK_S1>), dim3(hist_grid(n, 1024, 2048)), dim3(256), 0, s, f, w, histB);
}

// ------------------------------------------------------------------------------------------------
// single-workgroup scans
// ------------------------------------------------------------------------------------------------
// exclusive prefix of one uint32 per thread over a 1024-thread workgroup
VD_DEV uint32_t block_excl_scan(uint32_t v, uint32_t* wsum /*[16]*/, uint32_t& total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t inc = v;
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t t = (uint32_t)__shfl_up((int)inc, off, 64);
    if (lane >= off) inc += t;
  }
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  uint32_t wbase = 0, tot = 0;
  for (int i = 0; i < 16; ++i) {
    uint32_t s = wsum[i];
    if (i < wv) wbase += s;
    tot += s;
  }
  __syncthreads();
  total = tot;
  return wbase + inc - v;
}

enum { SEL_QUANT = 0, SEL_SUBJECT = 1 };

// scan A for one job: count, ranks, target prefixes (+ 64-bin arg-max for subject jobs)
VD_DEV void scan_a_job(vd_sel_ctl* c, const uint32_t* __restrict__ hist, int kind, float q0, float q1,
                       uint32_t* sm /* >= 16 + 64 + 8 words */) {
  const int tid = threadIdx.x;
  uint32_t loc[16];
  uint32_t tsum = 0;
  {
    const uint4* hv = reinterpret_cast<const uint4*>(hist) + tid * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint4 q = hv[i];
      loc[4 * i] = q.x; loc[4 * i + 1] = q.y; loc[4 * i + 2] = q.z; loc[4 * i + 3] = q.w;
      tsum += q.x + q.y + q.z + q.w;
    }
  }
  uint32_t* wsum = sm;
  uint32_t* h64 = sm + 16;
  if (tid < 64) h64[tid] = 0;
  uint32_t total;
  uint32_t base = block_excl_scan(tsum, wsum, total);
  if (kind == SEL_SUBJECT) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (loc[i]) {
        float v = __uint_as_float((uint32_t)(tid * 16 + i) << 16);
        int b = (int)(v * 64.0f);
        b = b > 63 ? 63 : b;
        atomicAdd(&h64[b], loc[i]);
      }
  }
  __syncthreads();
  if (tid == 0) {
    c->count = total;
    c->fallback = 0;
    c->ntargets = 0;
    if (kind == SEL_QUANT) {
      // torch.quantile: rank = float32(q) * float32(n-1), floor/ceil, weight = rank - floor
      const float nm1 = (float)((long long)total - 1);
      const float r0 = q0 * nm1, r1 = q1 * nm1;
      c->nranks = 4;
      c->ranks[0] = (uint64_t)floorf(r0); c->ranks[1] = (uint64_t)ceilf(r0);
      c->ranks[2] = (uint64_t)floorf(r1); c->ranks[3] = (uint64_t)ceilf(r1);
      c->w[0] = r0 - floorf(r0); c->w[1] = r1 - floorf(r1);
    } else {
      c->nranks = 1;
      c->ranks[0] = total ? ((uint64_t)total - 1) / 2 : 0;  // torch.median: lower median
      if (total < 20) { c->fallback = 1; c->nranks = 0; }
      int peak = 0;
      for (int b = 1; b < 64; ++b) if (h64[b] > h64[peak]) peak = b;  // first maximum
      c->peak_bin = (uint32_t)peak;
    }
    for (uint32_t r = 0; r < c->nranks; ++r)
      if (c->ranks[r] >= total) c->ranks[r] = total ? total - 1 : 0;
  }
  __syncthreads();
  __threadfence_block();
  const uint32_t nr = c->nranks;
  for (uint32_t r = 0; r < nr; ++r) {
    const uint64_t rk = c->ranks[r];
    if (rk >= base && rk < (uint64_t)base + tsum) {
      uint64_t acc = base;
      for (int i = 0; i < 16; ++i) {
        if (rk < acc + loc[i]) {
          sm[80 + r] = (uint32_t)(tid * 16 + i);  // prefix
          c->rank_rem[r] = rk - acc;
          break;
        }
        acc += loc[i];
      }
    }
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t nt = 0;
    for (uint32_t r = 0; r < nr; ++r) {
      uint32_t p = sm[80 + r], slot = nt;
      for (uint32_t t = 0; t < nt; ++t) if (c->tprefix[t] == p) slot = t;
      if (slot == nt) c->tprefix[nt++] = p;
      c->rank_t[r] = slot;
    }
    c->ntargets = nt;
  }
  __syncthreads();
}

// scan B for one job: resolve each rank inside its target's low-16 histogram through the coarse level.
// The 1024 threads split into 4 groups of 256; group r resolves rank r (all ranks in parallel).
VD_DEV void scan_b_job(vd_sel_ctl* c, const uint32_t* __restrict__ histB_job, const uint32_t* __restrict__ histBC_job,
                       uint32_t* sm /* >= 4*(4+1) words */) {
  const int tid = threadIdx.x, grp = tid >> 8, t = tid & 255, lane = tid & 63, wv = t >> 6;
  const uint32_t nr = c->nranks;
  uint32_t* gs = sm + grp * 8;  // per group: [0..3] wave sums, [4] coarse bin, [5] remaining rank
  for (int level = 0; level < 2; ++level) {
    uint32_t v = 0;
    uint64_t rk = 0;
    if ((uint32_t)grp < nr) {
      const uint32_t slot = c->rank_t[grp];
      if (level == 0) { v = histBC_job[(size_t)slot * VD_NB_BC + t]; rk = c->rank_rem[grp]; }
      else { v = histB_job[(size_t)slot * VD_NB_B + gs[4] * 256u + t]; rk = gs[5]; }
    }
    // exclusive scan over the 256 threads of the group (4 waves)
    uint32_t inc = v;
    for (int off = 1; off < 64; off <<= 1) {
      uint32_t u = (uint32_t)__shfl_up((int)inc, off, 64);
      if (lane >= off) inc += u;
    }
    __syncthreads();  // previous level's reads of gs[] are done
    if (lane == 63) gs[wv] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int i = 0; i < wv; ++i) base += gs[i];
    base += inc - v;
    __syncthreads();
    if ((uint32_t)grp < nr && rk >= base && rk < (uint64_t)base + v) {
      if (level == 0) { gs[4] = (uint32_t)t; gs[5] = (uint32_t)(rk - base); }
      else c->val[grp] = __uint_as_float((c->tprefix[c->rank_t[grp]] << 16) | (gs[4] << 8) | (uint32_t)t);
    }
    __syncthreads();
  }
  __threadfence_block();
  __syncthreads();
}

VD_DEV float quantile_lerp(float va, float vb, float w) {  // ATen lerp: two-branch form, each branch ONE fused multiply-add (round 5: oracle quantile_from_ranks)
  float diff = vb - va;
  return (w < 0.5f) ? vd_fma(w, diff, va) : vd_fma(-diff, 1.f - w, vb);
}
VD_DEV float subject_from_job(const vd_sel_ctl* c) {  // core/render_3d.py:159-172
  if (c->fallback) return 0.5f;
  const float bin_width = (float)(1.0 / 64);
  float subject = ((float)c->peak_bin + 0.5f) * bin_width;
  float s = 0.7f * subject + 0.3f * c->val[0];
  return vd_clamp(s, 0.f, 1.f);
}

// ------------------------------------------------------------------------------------------------
// scalar recurrences (Python-float == float64 arithmetic), ports of the tracker classes
// ------------------------------------------------------------------------------------------------
VD_DEV double fw_smooth_offset(vd3d_state* st, double cur, double threshold) {  // :485-498
  const double alpha = 0.97;
  double delta = fabs(cur - st->fw_prev_offset);
  if (delta < threshold) return st->fw_prev_offset;
  st->fw_prev_offset = alpha * st->fw_prev_offset + (1 - alpha) * cur;
  st->fw_frame_counter += 1;
  if (st->fw_frame_counter >= 100) {
    double v = st->fw_prev_offset;
    v = v < 1.0 ? v : 1.0;
    v = v > -1.0 ? v : -1.0;
    st->fw_prev_offset = v;
    st->fw_frame_counter = 0;
  }
  return st->fw_prev_offset;
}
VD_DEV double focal_update(vd3d_state* st, double motion, double cand) {  // :895-922
  double m = motion < 0.0 ? 0.0 : (motion > 1.0 ? 1.0 : motion);
  const double alpha = 0.10 + 0.20 * m, deadband = 0.03, max_step = 0.02;
  double c = cand;
  if (!st->focal_valid) { st->focal = c; st->focal_valid = 1; return c; }
  if (fabs(c - st->focal) < deadband) c = st->focal;
  double nf = (1.0 - alpha) * st->focal + alpha * c;
  double delta = nf - st->focal;
  if (delta > max_step) nf = st->focal + max_step;
  else if (delta < -max_step) nf = st->focal - max_step;
  nf = nf < 1.0 ? nf : 1.0;
  st->focal = nf > 0.0 ? nf : 0.0;
  return st->focal;
}

// pixel_shift_cuda scalars :633-671 once s1 is known
VD_DEV void shift_scalars(vd_dev_work* w, const vd3d_shift_params& p, int W, float s1, double fg_d, double mg_d, double bg_d) {
  const float fg = (float)fg_d, mg = (float)mg_d, bg = (float)bg_d;
  const float fgm = (float)p.fg_pop_multiplier, bgm = (float)p.bg_push_multiplier, pb = (float)p.parallax_balance;
  const double half_width_d = (double)W / 2.0;
  const float half_width = (float)half_width_d;
  w->fg = fg; w->mg = mg; w->bg = bg;
  w->have_zpo = 0; w->zpo_f = 0.f;
  w->fs.zpo_raw = 0.f; w->fs.zpo = 0.0;
  if (p.use_subject_tracking) {
    float adj = s1 * pb;
    float z = ((((-adj) * fg) * fgm + ((-adj) * mg)) + ((adj * bg) * bgm)) / half_width;
    z = z * (float)p.subject_lock_strength;
    z = z - (float)p.zero_parallax_strength;
    if (p.enable_floating_window) {
      float sw = vd_clamp(1.0f - s1 * 2.0f, 0.5f, 1.0f);
      z = z * sw;
      z = vd_clamp(z, -0.35f, 0.35f);
      w->fs.zpo_raw = z;
      double zd = fw_smooth_offset(&w->st, (double)z, 0.0015);
      w->fs.zpo = zd;
      w->zpo_f = (float)zd;
    } else {
      w->fs.zpo_raw = z;
      w->fs.zpo = (double)z;
      w->zpo_f = z;
    }
    w->have_zpo = 1;
  }
  w->msn = (float)(((double)W * p.max_pixel_shift_percent) / half_width_d);
  w->have_conv = 0; w->conv = 0.f;
  if (p.convergence_strength != 0.0) {
    double cn;
    if (p.enable_dynamic_convergence) cn = (double)(s1 * (float)p.convergence_strength);
    else cn = p.convergence_strength;
    w->conv = (float)(cn / half_width_d);
    w->have_conv = 1;
  }
}

// ------------------------------------------------------------------------------------------------
// the scalar stage kernel: one workgroup, runs between plane passes
// ------------------------------------------------------------------------------------------------
VD_DEV void run_scalar_stage(vd_dev_work* w, const uint32_t* histA, const uint32_t* histB, const vd_stage_args& a, uint32_t* sm) {
  const int tid = threadIdx.x;
  // The select control blocks live in LDS for the duration of the stage: the scans' rank/target bookkeeping is a chain
  // of dependent single-thread accesses, ~1 us each against global memory (measured: 15 us per stage before).
  __shared__ vd_sel_ctl lcs[VD_NJOBS];
  {
    const uint32_t* g = reinterpret_cast<const uint32_t*>(&w->job[0]);
    uint32_t* l = reinterpret_cast<uint32_t*>(&lcs[0]);
    for (int i = tid; i < (int)(sizeof(lcs) / 4); i += blockDim.x) l[i] = g[i];
  }
  __syncthreads();
  switch (a.stage) {
    case VD_ST_A0:
      scan_a_job(&lcs[VD_J_EYE_Q], histA + (size_t)VD_J_EYE_Q * VD_NB_A, SEL_QUANT, (float)0.02, (float)0.98, sm);
      break;
    case VD_ST_B0: {
      vd_sel_ctl* c = &lcs[VD_J_EYE_Q];
      scan_b_job(c, histB + (size_t)VD_J_EYE_Q * VD_MAX_T * VD_NB_B, vd_histbc(histB) + (size_t)VD_J_EYE_Q * VD_MAX_T * VD_NB_BC, sm);
      if (tid == 0) {  // DepthPercentileEMA.normalize :249-261
        float lo = quantile_lerp(c->val[0], c->val[1], c->w[0]);
        float hi = quantile_lerp(c->val[2], c->val[3], c->w[1]);
        vd3d_state* st = &w->st;
        w->fs.q_lo = lo; w->fs.q_hi = hi;
        if (a.shard == 3) {   // measure/replay sharding: the quantiles are exchanged, the EMA runs in k_shard2_r1 on every rank
          a.q_out[0] = lo; a.q_out[1] = hi;
          st->tdf_valid = 1;
          w->sum1 = 0; w->sum2 = 0; w->sum_mad = 0;
        } else
        if ((hi - lo) < 1e-5f) {
          w->collapse = 1;
        } else {
          w->collapse = 0;
          if (!st->ema_valid) { st->ema_lo = lo; st->ema_hi = hi; st->ema_valid = 1; }
          else {
            const float al = (float)0.92, be = (float)(1 - 0.92);
            st->ema_lo = al * st->ema_lo + be * lo;
            st->ema_hi = al * st->ema_hi + be * hi;
          }
        }
        if (a.shard != 3) {
          w->ema_lo = st->ema_lo;
          w->ema_den = (st->ema_hi - st->ema_lo) + (float)1e-6;
          w->fs.ema_lo = st->ema_lo; w->fs.ema_hi = st->ema_hi; w->fs.collapse = w->collapse;
          st->tdf_valid = 1;
          w->sum1 = 0; w->sum2 = 0; w->sum_mad = 0;
        }
      }
    } break;
    case VD_ST_A1:
      if (a.have_eye) scan_a_job(&lcs[VD_J_EYE_SUBJ], histA + (size_t)VD_J_EYE_SUBJ * VD_NB_A, SEL_SUBJECT, 0.f, 0.f, sm);
      scan_a_job(&lcs[VD_J_WORK_Q], histA + (size_t)VD_J_WORK_Q * VD_NB_A, SEL_QUANT, (float)a.shift.depth_stretch_lo,
                 (float)a.shift.depth_stretch_hi, sm);
      scan_a_job(&lcs[VD_J_WORK_S0], histA + (size_t)VD_J_WORK_S0 * VD_NB_A, SEL_SUBJECT, 0.f, 0.f, sm);
      break;
    case VD_ST_B1: {
      if (a.have_eye) scan_b_job(&lcs[VD_J_EYE_SUBJ], histB + (size_t)VD_J_EYE_SUBJ * VD_MAX_T * VD_NB_B, vd_histbc(histB) + (size_t)VD_J_EYE_SUBJ * VD_MAX_T * VD_NB_BC, sm);
      scan_b_job(&lcs[VD_J_WORK_Q], histB + (size_t)VD_J_WORK_Q * VD_MAX_T * VD_NB_B, vd_histbc(histB) + (size_t)VD_J_WORK_Q * VD_MAX_T * VD_NB_BC, sm);
      scan_b_job(&lcs[VD_J_WORK_S0], histB + (size_t)VD_J_WORK_S0 * VD_MAX_T * VD_NB_B, vd_histbc(histB) + (size_t)VD_J_WORK_S0 * VD_MAX_T * VD_NB_BC, sm);
      if (tid == 0) {
        vd_sel_ctl* cq = &lcs[VD_J_WORK_Q];
        const float lo = quantile_lerp(cq->val[0], cq->val[1], cq->w[0]);
        const float hi = quantile_lerp(cq->val[2], cq->val[3], cq->w[1]);
        const float s0 = subject_from_job(&lcs[VD_J_WORK_S0]);
        w->fs.q05 = lo; w->fs.q95 = hi; w->fs.s0 = s0;
        // shape_depth_for_pop constants :536-553
        const int stretch = !((hi - lo) < 1e-5f);
        const float den = (hi - lo) + (float)1e-6;
        const float subj = vd_clamp(s0, 0.f, 1.f);
        w->shp_stretch = stretch; w->shp_lo = lo; w->shp_den = den;
        w->shp_subj_s = stretch ? vd_clamp((subj - lo) / den, 0.f, 1.f) : subj;
        if (a.have_eye && a.shard == 3) {   // measurements only: the recurrences run in k_shard2_r2
          w->fs.s_norm = subject_from_job(&lcs[VD_J_EYE_SUBJ]);
          a.m_out[0] = w->sum1; a.m_out[1] = w->sum2; a.m_out[2] = w->sum_mad;
          if (a.aten_threads > 0) {   // torch-order float32 sums travel instead of the exact |difference| sum (which the mean of :928 then does not use)
            reinterpret_cast<float*>(&a.m_out[2])[0] = w->aten_sum_mean; reinterpret_cast<float*>(&a.m_out[2])[1] = w->aten_sum_mad;
          }
          reinterpret_cast<float*>(&a.m_out[3])[0] = w->fs.s_norm;
          w->sum1 = 0; w->sum2 = 0; w->sum_mad = 0;
        } else
        if (a.have_eye) {
          vd3d_state* st = &w->st;
          w->fs.s_norm = subject_from_job(&lcs[VD_J_EYE_SUBJ]);
          // compute_dynamic_parallax_scale :412-427 from the exact fixed-point sums
          const double n = (double)a.n_crop;
          const double s1d = (double)w->sum1 / VD_FX, s2d = (double)w->sum2 / VD_FX;
          float mean = (float)(s1d / n);
          if (a.aten_threads > 0 && a.n_crop > 0) mean = w->aten_sum_mean / (float)a.n_crop;   // torch.mean = ATen's float32 cascade sum / float(n)
          const float var = (float)((s2d - s1d * s1d / n) / (a.n_crop > 1 ? n - 1.0 : 1.0));
          const float nv = vd_clamp(var / (mean + 1e-5f), 0.f, 1.f);
          const float scale = (float)0.90 + nv * (float)(1.15 - 0.90);
          w->fs.mean_c = mean; w->fs.var_c = var; w->fs.dyn_scale = (double)scale;
          // ShiftSmoother :470-477 then :1276,1308
          double fg = a.shift.fg_shift, mg = a.shift.mg_shift, bg = a.shift.bg_shift;
          const double al = 0.15;
          if (!st->smooth_valid) { st->sm_fg = fg; st->sm_mg = mg; st->sm_bg = bg; st->smooth_valid = 1; }
          else {
            st->sm_fg = al * fg + (1 - al) * st->sm_fg;
            st->sm_mg = al * mg + (1 - al) * st->sm_mg;
            st->sm_bg = al * bg + (1 - al) * st->sm_bg;
          }
          fg = st->sm_fg; mg = st->sm_mg; bg = st->sm_bg;
          fg *= (double)scale; mg *= (double)scale; bg *= (double)scale;
          if (a.ipd_factor != 0.0 && !a.blank) { fg *= a.ipd_factor; mg *= a.ipd_factor; bg *= a.ipd_factor; }
          w->fg_d = fg; w->mg_d = mg; w->bg_d = bg;
          // compute_motion_metric :924-929
          w->fs.mad = 0.f;
          double motion = 0.0;
          if (st->prev_depth_valid && !a.blank) {
            float mad = (float)(((double)w->sum_mad / VD_FX) / (double)a.n_eye);
            if (a.aten_threads > 0 && a.n_eye > 0) mad = w->aten_sum_mad / (float)a.n_eye;
            w->fs.mad = mad;
            double m = (double)mad * 4.0;
            motion = m < 0.0 ? 0.0 : (m > 1.0 ? 1.0 : m);
          }
          if (!a.blank) w->fs.focal = focal_update(st, motion, (double)w->fs.s_norm);   // :1334-1337 sit in the non-blank branch
          else w->fs.focal = st->focal;
          w->focal = (float)w->fs.focal;
          st->prev_depth_valid = 1;
        } else {
          w->fg_d = a.shift.fg_shift; w->mg_d = a.shift.mg_shift; w->bg_d = a.shift.bg_shift;
        }
      }
    } break;
    case VD_ST_AQ:
      scan_a_job(&lcs[VD_J_EYE_Q], histA + (size_t)VD_J_EYE_Q * VD_NB_A, SEL_QUANT, (float)a.shift.depth_stretch_lo,
                 (float)a.shift.depth_stretch_hi, sm);
      break;
    case VD_ST_BQ: {
      vd_sel_ctl* c = &lcs[VD_J_EYE_Q];
      scan_b_job(c, histB + (size_t)VD_J_EYE_Q * VD_MAX_T * VD_NB_B, vd_histbc(histB) + (size_t)VD_J_EYE_Q * VD_MAX_T * VD_NB_BC, sm);
      if (tid == 0) {
        w->fs.q_lo = quantile_lerp(c->val[0], c->val[1], c->w[0]);
        w->fs.q_hi = quantile_lerp(c->val[2], c->val[3], c->w[1]);
      }
    } break;
    case VD_ST_BS:
      scan_b_job(&lcs[VD_J_WORK_S1], histB + (size_t)VD_J_WORK_S1 * VD_MAX_T * VD_NB_B, vd_histbc(histB) + (size_t)VD_J_WORK_S1 * VD_MAX_T * VD_NB_BC, sm);
      if (tid == 0) w->fs.s1 = subject_from_job(&lcs[VD_J_WORK_S1]);
      break;
    case VD_ST_A2:
      scan_a_job(&lcs[VD_J_WORK_S1], histA + (size_t)VD_J_WORK_S1 * VD_NB_A, SEL_SUBJECT, 0.f, 0.f, sm);
      break;
    case VD_ST_B2: {
      scan_b_job(&lcs[VD_J_WORK_S1], histB + (size_t)VD_J_WORK_S1 * VD_MAX_T * VD_NB_B, vd_histbc(histB) + (size_t)VD_J_WORK_S1 * VD_MAX_T * VD_NB_BC, sm);
      if (tid == 0) {
        {
          const float s1 = subject_from_job(&lcs[VD_J_WORK_S1]);
          w->fs.s1 = s1;
          if (a.shard == 3) reinterpret_cast<float*>(&a.m_out[3])[1] = s1;   // sharded: every tracker is replayed after the exchange
          else if (!a.blank) shift_scalars(w, a.shift, a.W, s1, w->fg_d, w->mg_d, w->bg_d);   // blank: pixel_shift_cuda never runs
        }
        if (a.have_eye && a.shard != 3) {  // floating-window bars :1390-1403
          vd3d_state* st = &w->st;
          const float s = w->fs.s_norm;
          const float rz = ((((-s) * (float)w->fg_d) + ((-s) * (float)w->mg_d)) + (s * (float)w->bg_d)) /
                           (float)((double)a.W / 2 + 1e-6);
          if (!st->conv_valid) { st->conv_val = (double)rz; st->conv_valid = 1; }
          else st->conv_val = 0.97 * st->conv_val + (1 - 0.97) * (double)rz;
          const double sz = st->conv_val;
          w->fs.stable_zero = sz;
          int bw = 0, side = 0;
          if (a.shift.enable_floating_window && a.shift.use_subject_tracking) {
            int raw_bar = (int)(fabs(sz) * a.W * 0.75);
            st->bar_prev_width = (int)(0.85 * st->bar_prev_width + (1 - 0.85) * raw_bar);
            bw = st->bar_prev_width < 80 ? st->bar_prev_width : 80;
            bw = bw > 0 ? bw : 0;
            if (sz > 0.005) side = 1; else if (sz < -0.005) side = 2;
          }
          w->bar_width = bw; w->bar_side = side;
          w->fs.bar_width = bw; w->fs.bar_side = side;
        }
        if (a.blank) {   // pixel_shift_cuda did not run for this frame: its reported scalars are those of "no call"
          w->fs.s0 = 0.f; w->fs.q05 = 0.f; w->fs.q95 = 0.f; w->fs.s1 = 0.f; w->fs.zpo_raw = 0.f; w->fs.zpo = 0.0;
        }
      }
    } break;
  }
  __syncthreads();
  {
    uint32_t* g = reinterpret_cast<uint32_t*>(&w->job[0]);
    const uint32_t* l = reinterpret_cast<const uint32_t*>(&lcs[0]);
    for (int i = tid; i < (int)(sizeof(lcs) / 4); i += blockDim.x) g[i] = l[i];
  }
}


__global__ __launch_bounds__(1024) void k_scalar_stage(vd_dev_work* w, const uint32_t* __restrict__ histA,
                                                       const uint32_t* __restrict__ histB, vd_stage_args a) {
  __shared__ uint32_t sm[128];
  run_scalar_stage(w, histA, histB, a, sm);
}

void vd_launch_scalar_stage(hipStream_t s, vd_dev_work* w, const uint32_t* histA, const uint32_t* histB, const vd_stage_args& a) {
  hipLaunchKernelGGL(k_scalar_stage, dim3(1), dim3(1024), 0, s, w, histA, histB, a);
}


// ================================================================================================
// Fused select chain (default path): pass-A histograms ride on the producer kernels, eye-res and warp-res work share
// one launch, and the scan + scalar stage that used to be a separate 1-workgroup launch is executed by the LAST
// workgroup to finish (agent-scope release -> ticket -> acquire, cdna_hip_programming.md Guideline 16).
// Per frame: 6 launches instead of 17 for everything up to the shift plane.
// ================================================================================================
// last-arrival ticket: returns true (workgroup-uniformly) in exactly one workgroup, after every other workgroup's
// global writes/atomics are visible to it.
VD_DEV bool last_workgroup(uint32_t* counter, uint32_t* sflag, int dbg = 0) {
  if (dbg & 2) return false;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores / returnless atomics have left
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // ROCm 7.2 may drop the wait after buffer_wbl2: restate it
    const uint32_t t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t last = (t == gridDim.x - 1) ? 1u : 0u;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm for the next frame
    }
    *sflag = last;
  }
  __syncthreads();
  return *sflag != 0u;
}

VD_DEV void lds_hist_flush(const uint32_t* h, uint32_t* g) {
  for (int b = threadIdx.x; b < NBL; b += blockDim.x) { const uint32_t c = h[b]; if (c) atomicAdd(&g[b], c); }
}
// LDS histogram adds of four consecutive pixels: neighbours of a smooth plane mostly share their bin, so runs of equal keys are merged into one
// atomic with the summed increment (increments add linearly, also the packed 0x10001 ones): up to 4x fewer LDS atomics and bank-conflict cycles
// (K3b: 4.8e7 conflict cycles per 16-frame launch before)
VD_DEV void vd_lds_hist_add4(uint32_t* hist, const unsigned k[4], unsigned inc[4]) {
#pragma unroll
  for (int q = 1; q < 4; ++q)
    if (k[q] == k[q - 1]) { inc[q] += inc[q - 1]; inc[q - 1] = 0u; }
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (inc[q]) atomicAdd(&hist[k[q]], inc[q]);
}
VD_DEV unsigned key_a(float v) { unsigned k = __float_as_uint(v) >> 16; return k < (NBL - 1) ? k : (NBL - 1); }

// Grid-stride walk over a float plane for the pass kernels: 4 consecutive pixels per thread and iteration (one 16-byte load, one
// integer division) when the row length allows it -- these loops were latency-bound with one 4-byte load in flight per
// thread (16 dependent round trips per thread at 4K).  body(valid, y, x, v) is called uniformly by every lane (it may ballot).
// VEC4 = false for the pass-B kernels: their hits are one value band of a smooth plane, i.e. spatially clustered, and the
// 16-byte walk would put a band on 4x fewer workgroups (measured 2x slower).
template <bool VEC4, int NT = 1024, class Body>
VD_DEV void vd_plane_walk(const float* __restrict__ p, long long n, int W, int wg, int nwg, Body body) {
  if (VEC4 && (W & 3) == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {
    const long long n4 = n >> 2;
    for (long long b4 = (long long)wg * 1024; b4 < n4; b4 += (long long)nwg * 1024) {
      const long long i4 = b4 + threadIdx.x;
      const bool ok = i4 < n4;
      vd_f4 v = {0.f, 0.f, 0.f, 0.f};
      int y = 0, x = 0;
      if (ok) {
        v = reinterpret_cast<const vd_f4*>(p)[i4];
        const unsigned i = (unsigned)i4 * 4u;
        y = (int)(i / (unsigned)W); x = (int)(i - (unsigned)y * (unsigned)W);
      }
      body(ok, y, x, v.x); body(ok, y, x + 1, v.y); body(ok, y, x + 2, v.z); body(ok, y, x + 3, v.w);
    }
  } else {
    // four grid-strided elements per trip, their loads issued together: with one 4-byte load in flight per thread these passes
    // were pure latency (a thread walks n / (nwg * 1024) ~ 16 dependent round trips at 4K); the visiting ORDER is irrelevant to
    // a histogram, the spatial spread over workgroups stays what the pass-B kernels want (VEC4 = false, see above)
    constexpr int U = 4;
    const long long stride = (long long)nwg * NT;
    for (long long base = (long long)wg * NT; base < n; base += U * stride) {
      float v[U]; bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long i = base + u * stride + threadIdx.x;
        ok[u] = i < n;
        v[u] = ok[u] ? p[i] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long i = base + u * stride + threadIdx.x;
        int y = 0, x = 0;
        if (ok[u]) { y = (int)((unsigned)i / (unsigned)W); x = (int)((unsigned)i - (unsigned)y * (unsigned)W); }
        body(ok[u], y, x, v[u]);
      }
    }
  }
}

// Walk of the sub-rectangle rows [y0, y1) x columns [xa, xb) of a float plane with row length W in 16-byte elements (W, xa, xb multiples
// of 4, base 16-byte aligned: vd_walk4_ok).  Element e of the rectangle = row y0 + e / wq, columns xa + 4 (e % wq) .. + 3, wq = (xb - xa) / 4;
// workgroup wg of nwg takes elements wg * NT + tid + k * nwg * NT, (row, column) are carried incrementally (ONE integer division per thread,
// the per-element division of vd_plane_walk was a fifth of the pass-B kernels' instructions), two 16-byte loads are in flight per thread
// (the 4-byte walk reached 1.6 - 2.6 TB/s).  body(ok, y, x, v): x = column of v.x; called uniformly by every lane (it may ballot).
VD_DEV bool vd_walk4_ok(const float* p, int W) { return (W & 3) == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
template <int NT = 1024, class Body>
VD_DEV void vd_rect_walk4(const float* __restrict__ p, int W, int y0, int y1, int xa, int xb, int wg, int nwg, Body body) {
  const unsigned wq = (unsigned)(xb - xa) >> 2;
  const unsigned ne = (y1 > y0 && wq) ? (unsigned)(y1 - y0) * wq : 0u;
  const unsigned stride = (unsigned)nwg * NT;
  const unsigned sy = wq ? stride / wq : 0u, sx = wq ? stride - sy * wq : 0u;
  unsigned e = (unsigned)wg * NT + threadIdx.x;
  unsigned r = wq ? e / wq : 0u, c = wq ? e - r * wq : 0u;
  for (unsigned base = (unsigned)wg * NT; base < ne; base += 2u * stride, e += 2u * stride) {
    unsigned r1 = r + sy, c1 = c + sx;
    if (c1 >= wq) { c1 -= wq; ++r1; }
    const bool ok0 = e < ne, ok1 = e + stride < ne;
    vd_f4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
    const int ya = y0 + (int)r, xa0 = xa + 4 * (int)c, yb = y0 + (int)r1, xb0 = xa + 4 * (int)c1;
    if (ok0) v0 = *reinterpret_cast<const vd_f4*>(p + (size_t)ya * W + xa0);
    if (ok1) v1 = *reinterpret_cast<const vd_f4*>(p + (size_t)yb * W + xb0);
    body(ok0, ya, xa0, v0);
    body(ok1, yb, xb0, v1);
    r = r1 + sy; c = c1 + sx;
    if (c >= wq) { c -= wq; ++r; }
  }
}
// bounding rectangle (columns rounded out to multiples of 4) of vd_in_subject_crop's region
VD_DEV void vd_subject_rect(int H, int W, int* y0, int* y1, int* xa, int* xb) {
  *y0 = H / 5; *y1 = H * 4 / 5;
  *xa = (W / 5) & ~3;
  const int xe = (W * 4 / 5 + 3) & ~3;
  *xb = xe < W ? xe : W;
}


// K0 (only when auto_crop_black_bars): detect_black_bars (:293-316) + crop_black_bars_torch (:318-326) + the aspect crop of
// :1236-1248, decided on device.  One wave per source row: integer sum of the cv2 RGB2GRAY of the frame after the
// reference's float32 round trip ((v/255)*255 truncated); row is "content" iff sum > 10*w (== np.mean(row) > 10).
// Last workgroup: first / last content row -> top / bottom -> crop rectangle in vd_dev_work::acrop.
__global__ __launch_bounds__(256) void k_autocrop(const uint8_t* __restrict__ frame, int h, int wd, double target_ratio,
                                                  uint32_t* __restrict__ rowflag, vd_dev_work* w, int* __restrict__ crop_out) {
  __shared__ uint32_t sflag;
  __shared__ int s_first, s_last;
  const int lane = threadIdx.x & 63;
  const int y = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (y < h) {
    unsigned sum = 0;
    for (int x = lane; x < wd; x += 64) {
      const uint8_t* px = frame + ((size_t)y * wd + x) * 3;
      const unsigned b = (unsigned)(uint8_t)(vd_u8_unit((float)px[0]) * 255.0f), g = (unsigned)(uint8_t)(vd_u8_unit((float)px[1]) * 255.0f),
                     r = (unsigned)(uint8_t)(vd_u8_unit((float)px[2]) * 255.0f);
      sum += (r * 4899u + g * 9617u + b * 1868u + 8192u) >> 14;
    }
    for (int off = 32; off > 0; off >>= 1) sum += (unsigned)__shfl_down((int)sum, off, 64);
    if (lane == 0) rowflag[y] = sum > 10u * (unsigned)wd ? 1u : 0u;
  }
  if (!last_workgroup(&w->ticket[6], &sflag)) return;
  if (threadIdx.x == 0) { s_first = h; s_last = -1; }
  __syncthreads();
  int first = h, last = -1;
  for (int r = threadIdx.x; r < h; r += 256)
    if (rowflag[r]) { first = min(first, r); last = max(last, r); }
  if (first < h) { atomicMin(&s_first, first); atomicMax(&s_last, last); }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int top = s_last < 0 ? 0 : s_first, bottom = s_last < 0 ? 0 : h - s_last - 1;
    int y0 = 0, hh = h;
    if (top + bottom < h) { y0 = top; hh = h - top - bottom; }
    int cx = 0, cy = y0, cw = wd, ch = hh;
    const double cr = (double)wd / (double)hh;
    if (fabs(cr - target_ratio) > 0.01) {
      if (cr > target_ratio) { const int nw = (int)((double)hh * target_ratio); cx = (wd - nw) / 2; cw = nw; }
      else { const int nh = (int)((double)wd / target_ratio); cy = y0 + (hh - nh) / 2; ch = nh; }
    }
    w->acrop[0] = cx; w->acrop[1] = cy; w->acrop[2] = cw; w->acrop[3] = ch;
    w->fs.crop_top = top; w->fs.crop_bottom = bottom;
    if (crop_out) { crop_out[0] = cx; crop_out[1] = cy; crop_out[2] = cw; crop_out[3] = ch; }
  }
}
void vd_launch_autocrop(hipStream_t s, const uint8_t* frame, int h, int wd, double target_ratio, uint32_t* rowflag, vd_dev_work* w,
                        int* crop_out) {
  hipLaunchKernelGGL(k_autocrop, dim3((h + 3) / 4), dim3(256), 0, s, frame, h, wd, target_ratio, rowflag, w, crop_out);
}

struct vd_targets { uint32_t nt, tp[VD_MAX_T], tmin, tspan; };   // tmin / tspan: hull of the target prefixes (empty: tmin = ~0)
VD_DEV vd_targets load_targets(const vd_sel_ctl* c, int dbg = 0) {
  vd_targets t; t.nt = (dbg & 16) ? 0u : c->ntargets;   // timing probe: bit4 = no pass-B hits at all (no global atomics)
  uint32_t lo = 0xffffffffu, hi = 0u;
  for (int i = 0; i < VD_MAX_T; ++i) {
    t.tp[i] = c->tprefix[i];
    if ((uint32_t)i < t.nt) { lo = t.tp[i] < lo ? t.tp[i] : lo; hi = t.tp[i] > hi ? t.tp[i] : hi; }
  }
  t.tmin = lo; t.tspan = t.nt ? hi - lo : 0u;
  return t;
}
VD_DEV void hist_b_add(uint32_t* histB, int job, const vd_targets& c, float v, bool member) {
  const unsigned bits = __float_as_uint(v);
  bool hit = false; unsigned key = 0;
  for (uint32_t t = 0; t < c.nt; ++t) if (member && (bits >> 16) == c.tp[t]) { hit = true; key = (t << 16) | (bits & 0xffffu); }
  if (!__any(hit)) return;   // the hits are one value band of a smooth plane: most waves have none
  vd_hist_add_agg(histB + (size_t)job * VD_MAX_T * VD_NB_B, key, hit);
  vd_hist_add_agg(vd_histbc(histB) + (size_t)job * VD_MAX_T * VD_NB_BC, key >> 8, hit);
}

// (Round 4, measured and not kept: pass B as per-job candidate LISTS -- wave-aggregated (key, count) entries behind one returning atomic per wave, the
// ranks resolved by the scanning workgroup with two LDS histogram passes over the list.  The device-scope atomics of the hits do bound these kernels
// (hits muted: K4 398 -> 212 us per 16 frames), but the single-workgroup list scan cost more than it saved: K4 866, K6 322 us.)
// four consecutive elements of one row: one range test per element against the hull of the job's target prefixes decides whether the wave
// has any candidate at all (a smooth plane: almost never); only then the per-element path with its membership test runs
template <class Member>   // member(q) -> bool, evaluated for candidates only
VD_DEV void hist_b_add4(uint32_t* histB, int job, const vd_targets& c, vd_f4 v, bool ok, Member member) {
  bool cand = false;
#pragma unroll
  for (int q = 0; q < 4; ++q) cand |= ((__float_as_uint(v[q]) >> 16) - c.tmin) <= c.tspan;
  if (!__any(ok && cand)) return;
#pragma unroll
  for (int q = 0; q < 4; ++q) hist_b_add(histB, job, c, v[q], ok && member(q));
}

// per-frame view of the launch arguments: the record destinations and the step index of frame F
VD_DEV vd_stage_args frame_args(vd_stage_args a, const vd_batch_frame& F, int stage) {
  a.stage = stage; a.shard_idx = F.shard_idx; a.q_out = F.q_out; a.m_out = F.m_out;
  return a;
}

// K1: ingest (+ TemporalDepthFilter): a pure streaming kernel since round 4 (pass A of J0 is K1a below -- with the histogram, its flush
// and the ticket's release fence inside, a batched launch paid them once per frame and workgroup: 51 us per 4K frame for 91 MB).  The plane
// EMA is a per-pixel recurrence over the frames, so a batched launch walks its frames INSIDE the thread, in order: the thread that wrote
// pixel o of frame f - 1 is the one that reads it as the previous value of frame f (same thread, same address: program order).
// Fast path (the render loop's exact 2:1 eye resize, rows 8-byte aligned): one thread = 4 consecutive eye pixels = a 2 x 8 block of source
// pixels, fetched with 8-byte (frame: 24 B per row) and 16-byte (float32 depth: 32 B per row) loads; same float32 expressions as
// vd_ingest_pixel (vd_interp_tap(2 n, n, o) = taps 2 o, 2 o + 1 with weights 1 - 0.5, 0.5).
struct vd_ingest_fast { int on; };
VD_DEV float vd_depth_u8(unsigned g) { return vd_u8_unit((float)g); }
__global__ __launch_bounds__(256) void k_chain_ingest(vd_batch b, int fmt, vd3d_render_params p, vd_stage_args a, int fast) {
  const int valid0 = b.w_main->st.tdf_valid;
  const int ew = p.eye_w, eh = p.eye_h;
  if (fast) {
    const unsigned gpr = (unsigned)ew >> 2, ng = gpr * (unsigned)eh;          // groups of 4 eye pixels
    for (unsigned g = blockIdx.x * 256u + threadIdx.x; g < ng; g += gridDim.x * 256u) {
      const unsigned ey = g / gpr, ex = (g - ey * gpr) * 4u;
      const size_t o = (size_t)ey * ew + ex, ne = (size_t)eh * ew;
      const float w1 = 0.5f, w0 = 1.f - w1;
      for (int fi = 0; fi < b.n; ++fi) {
        const vd_batch_frame& F = b.f[fi];
        const size_t srow = (size_t)(2 * ey + p.crop_y) * p.src_w + (2 * ex + p.crop_x);
        // ---- frame: 8 BGR pixels of two source rows
        uint32_t fr[2][6];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const uint2* q = reinterpret_cast<const uint2*>(F.frame + (srow + (size_t)r * p.src_w) * 3);
          const uint2 q0 = q[0], q1 = q[1], q2 = q[2];
          fr[r][0] = q0.x; fr[r][1] = q0.y; fr[r][2] = q1.x; fr[r][3] = q1.y; fr[r][4] = q2.x; fr[r][5] = q2.y;
        }
        // ---- depth: 8 samples of two source rows
        float dp[2][8];
        if (fmt == VD3D_DEPTH_F32) {
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const vd_f4* q = reinterpret_cast<const vd_f4*>((const float*)F.depth + srow + (size_t)r * p.src_w);
            const vd_f4 a0 = q[0], a1 = q[1];
            dp[r][0] = a0.x; dp[r][1] = a0.y; dp[r][2] = a0.z; dp[r][3] = a0.w; dp[r][4] = a1.x; dp[r][5] = a1.y; dp[r][6] = a1.z; dp[r][7] = a1.w;
          }
        } else {   // VD3D_DEPTH_GRAY_U8: 8 bytes per row
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const uint2 q = *reinterpret_cast<const uint2*>((const uint8_t*)F.depth + srow + (size_t)r * p.src_w);
#pragma unroll
            for (int k = 0; k < 4; ++k) { dp[r][k] = vd_depth_u8((q.x >> (8 * k)) & 0xffu); dp[r][4 + k] = vd_depth_u8((q.y >> (8 * k)) & 0xffu); }
          }
        }
        const int tdf_valid = fi == 0 ? valid0 : 1;
        vd_f4 prev = {0.f, 0.f, 0.f, 0.f};
        if (tdf_valid) prev = *reinterpret_cast<const vd_f4*>(F.tdf_prev + o);
        // byte k of a row's 24: pixel k / 3, channel BGR[k % 3]
        auto fb = [&](int r, int k) { return vd_u8_unit((float)((fr[r][k >> 2] >> (8 * (k & 3))) & 0xffu)); };
#pragma unroll
        for (int c = 0; c < 3; ++c) {   // output plane c = R, G, B; source byte 2 - c
          vd_f4 v;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int k0 = 3 * (2 * q) + (2 - c), k1 = 3 * (2 * q + 1) + (2 - c);
            v[q] = vd_bilerp(fb(0, k0), fb(0, k1), fb(1, k0), fb(1, k1), w0, w1, w0, w1);
          }
          *reinterpret_cast<vd_f4*>(F.rgb_eye + (size_t)c * ne + o) = v;
        }
        vd_f4 nv;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float cur = vd_bilerp(dp[0][2 * q], dp[0][2 * q + 1], dp[1][2 * q], dp[1][2 * q + 1], w0, w1, w0, w1);
          const float pv = tdf_valid ? prev[q] : cur;
          nv[q] = 0.5f * pv + (float)(1 - 0.5) * cur;
        }
        *reinterpret_cast<vd_f4*>(F.tdf + o) = nv;
      }
    }
    return;
  }
  const long long n = (long long)eh * ew;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int ey = (int)((unsigned)i / (unsigned)ew), ex = (int)((unsigned)i - (unsigned)ey * (unsigned)ew);
    for (int fi = 0; fi < b.n; ++fi) {
      const vd_batch_frame& F = b.f[fi];
      if (p.auto_crop_black_bars) {   // per-frame rectangle: k_autocrop's (sequential) or the exchanged table of a sharded step
        const int* cr = a.crop_tab ? a.crop_tab + 4 * F.shard_idx : F.w->acrop;
        p.crop_x = cr[0]; p.crop_y = cr[1]; p.crop_w = cr[2]; p.crop_h = cr[3];
      }
      vd_ingest_pixel(F.frame, F.depth, fmt, p, fi == 0 ? valid0 : 1, F.rgb_eye, F.tdf_prev, F.tdf, ey, ex);
    }
  }
}

// K1a: pass A of J0 (q.02 / q.98 of the clamped filtered plane) over the frames of the batch side by side; last workgroup: scan A0
__global__ __launch_bounds__(1024) void k_chain_a0(vd_batch b, long long n, int ew, vd_stage_args a) {
  __shared__ uint32_t h0[NBL];
  __shared__ uint32_t sm[128];
  const vd_batch_frame& F = b.f[blockIdx.y];
  for (int bb = threadIdx.x; bb < NBL; bb += 1024) h0[bb] = 0;
  __syncthreads();
  if (ew > 0 && vd_walk4_ok(F.tdf, ew)) {
    vd_rect_walk4(F.tdf, ew, 0, (int)(n / ew), 0, ew, (int)blockIdx.x, (int)gridDim.x, [&](bool ok, int, int, vd_f4 d) {
      unsigned k[4], inc[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { k[q] = key_a(vd_clamp(d[q], 0.f, 1.f)); inc[q] = (ok && !(a.dbg & 8)) ? 1u : 0u; }
      vd_lds_hist_add4(h0, k, inc);
    });
  } else {
    for (long long base = (long long)blockIdx.x * 1024; base < n; base += (long long)gridDim.x * 1024) {
      const long long i = base + threadIdx.x;
      const float v = i < n ? F.tdf[i] : 0.f;
      vd_lds_hist_add(h0, key_a(vd_clamp(v, 0.f, 1.f)), i < n && !(a.dbg & 8));
    }
  }
  __syncthreads();
  if (!(a.dbg & 4)) lds_hist_flush(h0, F.histA + (size_t)VD_J_EYE_Q * VD_NB_A);
  if (last_workgroup(&F.w->ticket[0], &sm[127], a.dbg) && !(a.dbg & 1)) run_scalar_stage(F.w, F.histA, F.histB, frame_args(a, F, VD_ST_A0), sm);
}

// K2: pass B of J0; last workgroup: scan B0 + DepthPercentileEMA
__global__ __launch_bounds__(1024) void k_chain_b0(vd_batch b, long long n, int ew, vd_stage_args a) {
  __shared__ uint32_t sm[128];
  const vd_batch_frame& F = b.f[blockIdx.y];
  vd_dev_work* w = F.w;
  uint32_t* histB = F.histB;
  const vd_targets tq = load_targets(&w->job[VD_J_EYE_Q], a.dbg);
  if (ew > 0 && vd_walk4_ok(F.tdf, ew)) {
    vd_rect_walk4(F.tdf, ew, 0, (int)(n / ew), 0, ew, (int)blockIdx.x, (int)gridDim.x, [&](bool ok, int, int, vd_f4 d) {
      vd_f4 v;
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = vd_clamp(d[q], 0.f, 1.f);
      hist_b_add4(histB, VD_J_EYE_Q, tq, v, ok, [&](int) { return true; });
    });
  } else
  vd_plane_walk<false>(F.tdf, n, 1, (int)blockIdx.x, (int)gridDim.x, [&](bool ok, int, int, float d) {
    hist_b_add(histB, VD_J_EYE_Q, tq, vd_clamp(d, 0.f, 1.f), ok);
  });
  if (last_workgroup(&w->ticket[1], &sm[127], a.dbg) && !(a.dbg & 1)) {
    run_scalar_stage(w, F.histA, histB, frame_args(a, F, VD_ST_B0), sm);
    if (threadIdx.x == 0 && b.w_main != w) b.w_main->st.tdf_valid = 1;   // batched step: the filter state lives in the context's block
  }
}

// curved depth at warp resolution straight from the filtered plane (normalisation recomputed per tap, so this
// does not depend on the dn plane another workgroup of the same launch is writing) or from a given plane (B1 API)
struct FWorkSrc {
  const float* src; int ih, iw, H, W, norm, collapse; float lo, den;
  int pm;   // round 5: ATen's premultiplied-weight bilinear kernel (N-thread ATen mode, warp planes with H + W <= 128)
  float scale_h, scale_w, step_x, step_y;   // hoisted loop invariants (same float32 expressions as vd_interp_tap / vd_lin11)
  VD_DEV float tap(size_t idx) const {
    float d = src[idx];
    if (!norm) return d;
    d = vd_clamp(d, 0.f, 1.f);
    return collapse ? d : vd_clamp((d - lo) / den, 0.f, 1.f);
  }
  VD_DEV float at(int y, int x) const {
    float d;
    if (ih == H && iw == W) d = tap((size_t)y * W + x);
    else {
      const bool two = !pm && 2 * ih == H && 2 * iw == W;   // exact 2:1 (Half-SBS): closed-form taps, same values
      const vd_tap ty = two ? vd_tap21(ih, y) : vd_interp_tap_s(ih, H, scale_h, y);
      const vd_tap tx = two ? vd_tap21(iw, x) : vd_interp_tap_s(iw, W, scale_w, x);
      d = vd_bilerp_sel(pm != 0, tap((size_t)ty.i0 * iw + tx.i0), tap((size_t)ty.i0 * iw + tx.i1), tap((size_t)ty.i1 * iw + tx.i0),
                        tap((size_t)ty.i1 * iw + tx.i1), tx.w0, tx.w1, ty.w0, ty.w1);
    }
    const float xx = vd_lin11_step(step_x, W, x), yy = vd_lin11_step(step_y, H, y);
    const float curv = 1.f - (xx * xx + yy * yy);
    return vd_clamp(d + curv * (float)0.08, 0.f, 1.f);
  }
  // at(y, x .. x + 3) for the exact 2:1 resize of a plain plane (norm == 0), x a multiple of 4 with 4 <= x and x + 6 <= W: the four
  // pixels share the eye columns x/2 - 1 .. x/2 + 2 and their taps are a parity rule (vd_tap21: even x -> (i0, w1) = (x/2 - 1, 0.75),
  // odd x -> ((x - 1)/2, 0.25)): 8 loads and 24 operations instead of 16 taps.  Same association as vd_bilerp, same values as at().
  VD_DEV vd_f4 at4_21(int y, int x) const {
    const vd_tap ty = vd_tap21(ih, y);
    const float* r0 = src + (size_t)ty.i0 * iw + ((x >> 1) - 1);
    const float* r1 = src + (size_t)ty.i1 * iw + ((x >> 1) - 1);
    const float p0[4] = {r0[0], r0[1], r0[2], r0[3]}, p1[4] = {r1[0], r1[1], r1[2], r1[3]};
    const float a0 = vd_fma(p0[0], 0.25f, 0.75f * p0[1]), b0 = vd_fma(p1[0], 0.25f, 0.75f * p1[1]);
    const float a1 = vd_fma(p0[1], 0.75f, 0.25f * p0[2]), b1 = vd_fma(p1[1], 0.75f, 0.25f * p1[2]);
    const float a2 = vd_fma(p0[1], 0.25f, 0.75f * p0[2]), b2 = vd_fma(p1[1], 0.25f, 0.75f * p1[2]);
    const float a3 = vd_fma(p0[2], 0.75f, 0.25f * p0[3]), b3 = vd_fma(p1[2], 0.75f, 0.25f * p1[3]);
    const float d[4] = {vd_fma(a0, ty.w0, ty.w1 * b0), vd_fma(a1, ty.w0, ty.w1 * b1), vd_fma(a2, ty.w0, ty.w1 * b2),
                        vd_fma(a3, ty.w0, ty.w1 * b3)};
    const float yy = vd_lin11_step(step_y, H, y), yy2 = yy * yy;
    vd_f4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float xx = vd_lin11_step(step_x, W, x + q);
      const float curv = 1.f - (xx * xx + yy2);
      v[q] = vd_clamp(d[q] + curv * (float)0.08, 0.f, 1.f);
    }
    return v;
  }
};

// K3a (eye resolution; skipped by the bare pixel_shift_cuda entry point): normalise the filtered plane -> dn_cur, the exact centre-crop sums,
//     the MAD against the previous plane, pass A of J1.  No ticket: its scan runs with K3b's.
__global__ __launch_bounds__(1024) void k_chain_norm(vd_batch b, int eh, int ew, vd_stage_args a) {
  __shared__ uint32_t h1[NBL];
  __shared__ long long part[3][16];
  const vd_batch_frame& F = b.f[blockIdx.y];
  const float* __restrict__ tdf = F.tdf; float* __restrict__ dn_cur = F.dn; const float* __restrict__ dn_prev = F.dn_prev;
  vd_dev_work* w = F.w; uint32_t* histA = F.histA;
  a.shard_idx = F.shard_idx;
  // normalisation scalars: stage B0's (device) or, in the measure/replay sharding, this frame's row of the replayed table;
  // there dn_prev is the PREVIOUS FILTERED plane and is normalised on the fly with the previous frame's row
  const bool m3 = a.shard == 3;
  const float* e_cur = m3 ? a.etab + VD_ETAB * (a.shard_idx + 1) : nullptr;
  const float* e_prv = m3 ? a.etab + VD_ETAB * a.shard_idx : nullptr;
  const float lo = m3 ? e_cur[0] : w->ema_lo, den = m3 ? e_cur[1] : w->ema_den;
  const int collapse = m3 ? (int)e_cur[2] : w->collapse;
  for (int b = threadIdx.x; b < NBL; b += 1024) h1[b] = 0;
  __syncthreads();
  const long long n = (long long)eh * ew;
  const int have_prev = m3 ? (int)e_prv[3] : w->st.prev_depth_valid;
  const float p_lo = m3 ? e_prv[0] : 0.f, p_den = m3 ? e_prv[1] : 1.f;
  const int p_col = m3 ? (int)e_prv[2] : 0;
  long long s1 = 0, s2 = 0, sd = 0;
  for (long long base = (long long)blockIdx.x * 1024; base < n; base += (long long)gridDim.x * 1024) {
    const long long i = base + threadIdx.x;
    float v = 0.f; bool in_crop = false;
    if (i < n) {
      const float d = vd_clamp(tdf[i], 0.f, 1.f);
      v = collapse ? d : vd_clamp((d - lo) / den, 0.f, 1.f);
      dn_cur[i] = v;
      const int y = (int)((unsigned)i / (unsigned)ew), x = (int)((unsigned)i - (unsigned)y * (unsigned)ew);
      if (y >= eh / 4 && y < eh * 3 / 4 && x >= ew / 4 && x < ew * 3 / 4) {
        const double dv = (double)v;
        s1 += vd_fx40(dv); s2 += vd_fx40(dv * dv);
      }
      if (have_prev) {
        float vp = dn_prev[i];
        if (m3) { const float dp = vd_clamp(vp, 0.f, 1.f); vp = p_col ? dp : vd_clamp((dp - p_lo) / p_den, 0.f, 1.f); }
        sd += vd_fx40((double)fabsf(v - vp));
      }
      in_crop = vd_in_subject_crop(y, x, eh, ew, v);
    }
    vd_lds_hist_add(h1, key_a(v), in_crop && !(a.dbg & 8));
  }
  s1 = vd_wave_sum_ll(s1); s2 = vd_wave_sum_ll(s2); sd = vd_wave_sum_ll(sd);
  if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = s1; part[1][threadIdx.x >> 6] = s2; part[2][threadIdx.x >> 6] = sd; }
  __syncthreads();
  if (threadIdx.x < 3) {
    long long v = 0;
    for (int i = 0; i < 16; ++i) v += part[threadIdx.x][i];
    long long* dst = threadIdx.x == 0 ? &w->sum1 : (threadIdx.x == 1 ? &w->sum2 : &w->sum_mad);
    if (v) atomicAdd((unsigned long long*)dst, (unsigned long long)v);
  }
  if (!(a.dbg & 4)) lds_hist_flush(h1, histA + (size_t)VD_J_EYE_SUBJ * VD_NB_A);
}

// K3b (warp resolution): curved depth of the NORMALISED plane -> dc, pass A of J2 (all pixels) + J3 (subject crop); last workgroup: scan A1.
// One LDS histogram serves both jobs: the low half-word counts every pixel of the bin, the high half-word the pixels inside the subject
// crop (a workgroup sees at most VD_K3_MAX_PX < 65536 pixels: the host sizes the grid), so a pixel costs ONE ds_add and the kernel needs
// 65 KB of LDS instead of 130 KB -- two workgroups per CU.  With the exact 2:1 resize of Half-SBS four pixels share their taps (at4_21).
#define VD_K3_MAX_PX 61440
__global__ __launch_bounds__(1024) void k_chain_stage1(vd_batch b, FWorkSrc f, vd_stage_args a) {
  __shared__ uint32_t hp[NBL];
  __shared__ uint32_t sm[128];
  const vd_batch_frame& F = b.f[blockIdx.y];
  f.src = F.dn;
  float* __restrict__ dc = F.dc; vd_dev_work* w = F.w; uint32_t* histA = F.histA; const uint32_t* histB = F.histB;
  for (int b = threadIdx.x; b < NBL; b += 1024) hp[b] = 0;
  __syncthreads();
  const long long n = (long long)f.H * f.W;
  const int nwg = (int)gridDim.x, wg = (int)blockIdx.x;
  if ((f.W & 3) == 0 && (reinterpret_cast<uintptr_t>(dc) & 15) == 0) {
    const bool two = !f.pm && !f.norm && 2 * f.ih == f.H && 2 * f.iw == f.W;
    const long long n4 = n >> 2;
    for (long long b4 = (long long)wg * 1024; b4 < n4; b4 += (long long)nwg * 1024) {
      const long long i4 = b4 + threadIdx.x;
      const bool ok = i4 < n4;
      vd_f4 v = {0.f, 0.f, 0.f, 0.f};
      int y = 0, x = 0;
      if (ok) {
        const unsigned i = (unsigned)i4 * 4u;
        y = (int)(i / (unsigned)f.W); x = (int)(i - (unsigned)y * (unsigned)f.W);
        if (two && x >= 4 && x + 6 <= f.W) v = f.at4_21(y, x);
        else { v.x = f.at(y, x); v.y = f.at(y, x + 1); v.z = f.at(y, x + 2); v.w = f.at(y, x + 3); }
        reinterpret_cast<vd_f4*>(dc)[i4] = v;   // curved depth plane: pass B and the shape kernel stream it
      }
      unsigned k[4], inc[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        k[q] = key_a(v[q]);
        inc[q] = (ok && !(a.dbg & 8)) ? (vd_in_subject_crop(y, x + q, f.H, f.W, v[q]) ? 0x10001u : 1u) : 0u;
      }
      vd_lds_hist_add4(hp, k, inc);
    }
  } else {
    for (long long base = (long long)wg * 1024; base < n; base += (long long)nwg * 1024) {
      const long long i = base + threadIdx.x;
      if (i < n) {
        const int y = (int)((unsigned)i / (unsigned)f.W), x = (int)((unsigned)i - (unsigned)y * (unsigned)f.W);
        const float v = f.at(y, x);
        dc[i] = v;   // curved depth plane: pass B and the shape kernel stream it instead of re-deriving it from 4 taps
        atomicAdd(&hp[key_a(v)], vd_in_subject_crop(y, x, f.H, f.W, v) ? 0x10001u : 1u);
      }
    }
  }
  __syncthreads();
  if (!(a.dbg & 4))
  for (int b = threadIdx.x; b < NBL; b += 1024) {
    const uint32_t c = hp[b];
    if (c & 0xffffu) atomicAdd(&histA[(size_t)VD_J_WORK_Q * VD_NB_A + b], c & 0xffffu);
    if (c >> 16) atomicAdd(&histA[(size_t)VD_J_WORK_S0 * VD_NB_A + b], c >> 16);
  }
  if (last_workgroup(&w->ticket[2], &sm[127], a.dbg) && !(a.dbg & 1)) run_scalar_stage(w, histA, histB, frame_args(a, F, VD_ST_A1), sm);
}

// K4: [eye] pass B of J1 on the stored dn plane | [work] pass B of J2 + J3 ; last workgroup: scan B1 + its scalar stage
__global__ __launch_bounds__(1024) void k_chain_b1(vd_batch b, int eh, int ew, int n_eye_wg, FWorkSrc f, vd_stage_args a) {
  __shared__ uint32_t sm[128];
  const vd_batch_frame& F = b.f[blockIdx.y];
  const float* __restrict__ dn_cur = F.dn; const float* __restrict__ dc = F.dc;
  vd_dev_work* w = F.w; const uint32_t* histA = F.histA; uint32_t* histB = F.histB;
  const vd_targets t_eye = load_targets(&w->job[VD_J_EYE_SUBJ], a.dbg), t_q = load_targets(&w->job[VD_J_WORK_Q], a.dbg),
                   t_s0 = load_targets(&w->job[VD_J_WORK_S0], a.dbg);
  if ((int)blockIdx.x < n_eye_wg) {
    if (vd_walk4_ok(dn_cur, ew)) {   // the eye-res subject job only counts pixels of the centre crop: walk that rectangle
      int y0, y1, xa, xb;
      vd_subject_rect(eh, ew, &y0, &y1, &xa, &xb);
      vd_rect_walk4(dn_cur, ew, y0, y1, xa, xb, (int)blockIdx.x, n_eye_wg, [&](bool ok, int y, int x, vd_f4 v) {
        hist_b_add4(histB, VD_J_EYE_SUBJ, t_eye, v, ok, [&](int q) { return vd_in_subject_crop(y, x + q, eh, ew, v[q]); });
      });
    } else
    vd_plane_walk<false>(dn_cur, (long long)eh * ew, ew, (int)blockIdx.x, n_eye_wg, [&](bool ok, int y, int x, float v) {
      hist_b_add(histB, VD_J_EYE_SUBJ, t_eye, v, ok && vd_in_subject_crop(y, x, eh, ew, v));
    });
  } else {
    if (vd_walk4_ok(dc, f.W)) {
      vd_rect_walk4(dc, f.W, 0, f.H, 0, f.W, (int)blockIdx.x - n_eye_wg, (int)gridDim.x - n_eye_wg, [&](bool ok, int y, int x, vd_f4 v) {
        hist_b_add4(histB, VD_J_WORK_Q, t_q, v, ok, [&](int) { return true; });
        hist_b_add4(histB, VD_J_WORK_S0, t_s0, v, ok, [&](int q) { return vd_in_subject_crop(y, x + q, f.H, f.W, v[q]); });
      });
    } else
    vd_plane_walk<false>(dc, (long long)f.H * f.W, f.W, (int)blockIdx.x - n_eye_wg, (int)gridDim.x - n_eye_wg, [&](bool ok, int y, int x, float v) {
      hist_b_add(histB, VD_J_WORK_Q, t_q, v, ok);
      hist_b_add(histB, VD_J_WORK_S0, t_s0, v, ok && vd_in_subject_crop(y, x, f.H, f.W, v));
    });
  }
  if (last_workgroup(&w->ticket[3], &sm[127], a.dbg) && !(a.dbg & 1)) run_scalar_stage(w, histA, histB, frame_args(a, F, VD_ST_B1), sm);
}

// K5: shape_depth_for_pop -> D plane + pass A of J4 ; last workgroup: scan A2
// TAILS (round 5): the plane has ATen scalar tails for the reference's thread count (vd_tails_of; no video size does) -- those pixels take libm's pow
template <bool TAILS>
__global__ __launch_bounds__(1024) void k_chain_shape(vd_batch b, FWorkSrc f, float mid, float gamma, vd_stage_args a, vd_tails tl) {
  __shared__ uint32_t h1[NBL];
  __shared__ uint32_t sm[128];
  const vd_batch_frame& F = b.f[blockIdx.y];
  const float* __restrict__ dc = F.dc; float* __restrict__ D = F.D;
  vd_dev_work* w = F.w; uint32_t* histA = F.histA; const uint32_t* histB = F.histB;
  for (int b = threadIdx.x; b < NBL; b += 1024) h1[b] = 0;
  __syncthreads();
  const long long n = (long long)f.H * f.W;
  const int stretch = w->shp_stretch;
  const float lo = w->shp_lo, den = w->shp_den, subj_s = w->shp_subj_s;
  const double gamma_d = a.shift.depth_pop_gamma;   // the Python float: ATen's scalar lambda takes it unrounded
  auto shape1 = [&](float d, unsigned i) {   // shape_depth_for_pop :519-558 for one pixel
    const float ds = stretch ? vd_clamp((d - lo) / den, 0.f, 1.f) : d;
    const float centered = (ds - subj_s) + mid;
    const float t = centered - mid;
    const float sgn = (t > 0.f) ? 1.f : ((t < 0.f) ? -1.f : 0.f);
    if (TAILS) { if (vd_in_tail(tl, i)) return vd_clamp(sgn * vd_pow_tail(fabsf(t), gamma_d) + mid, 0.f, 1.f); }
    return vd_clamp(sgn * vd_pow_torch(fabsf(t), gamma, c_vd_rs14) + mid, 0.f, 1.f);   // torch.pow: SLEEF's value
  };
  if ((f.W & 3) == 0 && ((reinterpret_cast<uintptr_t>(dc) | reinterpret_cast<uintptr_t>(D)) & 15) == 0) {
    const long long n4 = n >> 2;   // 4 pixels per thread: one 16-byte load / store, one integer division, 4 independent pow chains
    for (long long b4 = (long long)blockIdx.x * 1024; b4 < n4; b4 += (long long)gridDim.x * 1024) {
      const long long i4 = b4 + threadIdx.x;
      const bool ok = i4 < n4;
      vd_f4 v = {0.f, 0.f, 0.f, 0.f};
      int y = 0, x = 0;
      if (ok) {
        const vd_f4 d = reinterpret_cast<const vd_f4*>(dc)[i4];
        const unsigned i0 = (unsigned)i4 * 4u;
        v.x = shape1(d.x, i0); v.y = shape1(d.y, i0 + 1u); v.z = shape1(d.z, i0 + 2u); v.w = shape1(d.w, i0 + 3u);
        reinterpret_cast<vd_f4*>(D)[i4] = v;
        const unsigned i = (unsigned)i4 * 4u;
        y = (int)(i / (unsigned)f.W); x = (int)(i - (unsigned)y * (unsigned)f.W);
      }
      unsigned k[4], inc[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { k[q] = key_a(v[q]); inc[q] = (ok && !(a.dbg & 8) && vd_in_subject_crop(y, x + q, f.H, f.W, v[q])) ? 1u : 0u; }
      vd_lds_hist_add4(h1, k, inc);
    }
  } else {
    for (long long base = (long long)blockIdx.x * 1024; base < n; base += (long long)gridDim.x * 1024) {
      const long long i = base + threadIdx.x;
      float v = 0.f; bool in_crop = false;
      if (i < n) {
        const int y = (int)((unsigned)i / (unsigned)f.W), x = (int)((unsigned)i - (unsigned)y * (unsigned)f.W);
        v = shape1(dc[i], (unsigned)i);
        D[i] = v;
        in_crop = vd_in_subject_crop(y, x, f.H, f.W, v);
      }
      vd_lds_hist_add(h1, key_a(v), in_crop);
    }
  }
  __syncthreads();
  if (!(a.dbg & 4)) lds_hist_flush(h1, histA + (size_t)VD_J_WORK_S1 * VD_NB_A);
  if (last_workgroup(&w->ticket[4], &sm[127], a.dbg) && !(a.dbg & 1)) run_scalar_stage(w, histA, histB, frame_args(a, F, VD_ST_A2), sm);
}

// K6: pass B of J4 on the D plane ; last workgroup: scan B2 + tracker recurrences
__global__ __launch_bounds__(1024) void k_chain_b2(vd_batch b, int H, int W, vd_stage_args a) {
  __shared__ uint32_t sm[128];
  const vd_batch_frame& F = b.f[blockIdx.y];
  const float* __restrict__ D = F.D; vd_dev_work* w = F.w; const uint32_t* histA = F.histA; uint32_t* histB = F.histB;
  const long long n = (long long)H * W;
  const vd_targets t_s1 = load_targets(&w->job[VD_J_WORK_S1], a.dbg);
  if (vd_walk4_ok(D, W)) {   // the shaped-depth subject job only counts pixels of the centre crop: walk that rectangle (36 % of the plane)
    int y0, y1, xa, xb;
    vd_subject_rect(H, W, &y0, &y1, &xa, &xb);
    vd_rect_walk4(D, W, y0, y1, xa, xb, (int)blockIdx.x, (int)gridDim.x, [&](bool ok, int y, int x, vd_f4 v) {
      hist_b_add4(histB, VD_J_WORK_S1, t_s1, v, ok, [&](int q) { return vd_in_subject_crop(y, x + q, H, W, v[q]); });
    });
  } else
  vd_plane_walk<false>(D, n, W, (int)blockIdx.x, (int)gridDim.x, [&](bool ok, int y, int x, float v) {
    hist_b_add(histB, VD_J_WORK_S1, t_s1, v, ok && vd_in_subject_crop(y, x, H, W, v));
  });
  if (last_workgroup(&w->ticket[5], &sm[127], a.dbg) && !(a.dbg & 1)) run_scalar_stage(w, histA, histB, frame_args(a, F, VD_ST_B2), sm);
}

static inline int chain_grid(long long n, int per_wg, int cap) {
  long long g = (n + per_wg - 1) / per_wg;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// workgroups per frame of a batched launch: `per_wg` elements per workgroup, at most `cap`, shrunk when many frames share the launch
// (the frames of a step fill the chip together: fewer, longer workgroups per frame = fewer LDS histogram flushes and tickets)
static int g_batch_div = -1;
void vd_set_batch_grid_div(int v) { g_batch_div = v < 1 ? 1 : v; }
static inline int batch_grid(long long n, int per_wg, int cap, int nframes) {
  if (g_batch_div < 0) { const char* e = getenv("VD3D_BATCH_GRID_DIV"); g_batch_div = e ? atoi(e) : 8; if (g_batch_div < 1) g_batch_div = 1; }   // 8: measured best of 1 .. 64 at 4K, 16 frames (profiles/r04_chain_batched.md)
  int g = chain_grid(n, per_wg, cap);
  if (nframes >= 4) g = (g + g_batch_div - 1) / g_batch_div;
  return g < 1 ? 1 : g;
}

void vd_launch_chain_eye(hipStream_t s, const vd_batch& b, int fmt, const vd3d_render_params& p, const vd_stage_args& a) {
  const long long ne = (long long)p.eye_h * p.eye_w;
  // K1 fast path: exact 2:1 eye resize of a fixed crop window, 4 eye pixels per thread through 8- / 16-byte loads
  int fast = !vd_interp_premult(3, p.eye_h, p.eye_w, p.aten_sum_threads) &&   // the N-thread ATen mode resizes such eyes with its premultiplied-weight kernel: generic path
             !p.auto_crop_black_bars && p.crop_w == 2 * p.eye_w && p.crop_h == 2 * p.eye_h && (p.eye_w & 3) == 0 &&
             (fmt == VD3D_DEPTH_F32 || fmt == VD3D_DEPTH_GRAY_U8) && ((3 * (long long)p.src_w) & 7) == 0 &&
             ((3 * ((long long)p.crop_y * p.src_w + p.crop_x)) & 7) == 0;
  const long long dsz = fmt == VD3D_DEPTH_F32 ? 4 : 1;
  if (fast && (((dsz * p.src_w) & (fmt == VD3D_DEPTH_F32 ? 15 : 7)) || ((dsz * ((long long)p.crop_y * p.src_w + p.crop_x)) & (fmt == VD3D_DEPTH_F32 ? 15 : 7)))) fast = 0;
  for (int j = 0; fast && j < b.n; ++j) {
    const vd_batch_frame& F = b.f[j];
    if ((reinterpret_cast<uintptr_t>(F.frame) & 7) || (reinterpret_cast<uintptr_t>(F.depth) & 15) || (reinterpret_cast<uintptr_t>(F.rgb_eye) & 15) ||
        (reinterpret_cast<uintptr_t>(F.tdf) & 15) || (reinterpret_cast<uintptr_t>(F.tdf_prev) & 15) || ((ne * 4) & 15)) fast = 0;
  }
  const long long items = fast ? ne / 4 : ne;
  hipLaunchKernelGGL(k_chain_ingest, dim3(chain_grid(items, 256, 4096)), dim3(256), 0, s, b, fmt, p, a, fast);
  hipLaunchKernelGGL(k_chain_a0, dim3(batch_grid(ne, 8192, 256, b.n), b.n), dim3(1024), 0, s, b, ne, p.eye_w, a);
  hipLaunchKernelGGL(k_chain_b0, dim3(batch_grid(ne, 4096, 256, b.n), b.n), dim3(1024), 0, s, b, ne, p.eye_w, a);
}

// have_eye: the render path (F.tdf = filtered plane -> F.dn normalised); else the bare pixel_shift_cuda entry point (F.dn = the caller's plane)
void vd_launch_chain_work(hipStream_t s, const vd_batch& b, int have_eye, int ih, int iw, int H, int W, float mid, float gamma,
                          const vd_stage_args& a) {
  FWorkSrc f;
  f.src = nullptr; f.ih = ih; f.iw = iw; f.H = H; f.W = W; f.norm = 0; f.collapse = 0; f.lo = 0.f; f.den = 1.f;
  f.pm = vd_interp_premult(1, H, W, a.shift.aten_threads) ? 1 : 0;
  f.scale_h = (float)ih / (float)H; f.scale_w = (float)iw / (float)W;
  f.step_x = W > 1 ? (1.f - (-1.f)) / (float)(W - 1) : 0.f; f.step_y = H > 1 ? (1.f - (-1.f)) / (float)(H - 1) : 0.f;
  const long long ne = (long long)ih * iw, n = (long long)H * W;
  const unsigned nf = (unsigned)b.n;
  const int eye_wg = have_eye ? batch_grid(ne, 4096, 256, b.n) : 0;
  // K3b: 65 KB of LDS -> two workgroups per CU; every workgroup walks <= VD_K3_MAX_PX pixels (its packed histogram counts in 16 bits)
  int work_wg = batch_grid(n, 16384, 512, b.n);
  { const long long need = (n + VD_K3_MAX_PX - 4096 - 1) / (VD_K3_MAX_PX - 4096); if (work_wg < need) work_wg = (int)need; }
  const int eye_wg_b = have_eye ? batch_grid(ne, 4096, 128, b.n) : 0;
  const int work_wg_b = batch_grid(n, 4096, 512, b.n);   // K4: streams the stored curved-depth plane (one release fence per workgroup)
  // K3a normalises the eye-res plane (its own launch: K3b then samples plain values -- no per-tap division, taps shared by four pixels)
  if (have_eye) hipLaunchKernelGGL(k_chain_norm, dim3(eye_wg, nf), dim3(1024), 0, s, b, ih, iw, a);
  if (have_eye && a.aten_threads > 0) vd_launch_aten_sums(s, b, a);   // torch.mean's summation order (vd3d_atensum.hip): read by K4's scalar stage
  hipLaunchKernelGGL(k_chain_stage1, dim3(work_wg, nf), dim3(1024), 0, s, b, f, a);
  hipLaunchKernelGGL(k_chain_b1, dim3(eye_wg_b + work_wg_b, nf), dim3(1024), 0, s, b, ih, iw, eye_wg_b, f, a);
  const vd_tails tl = vd_tails_of((unsigned long long)n, vd_pow_is_special(gamma) ? 0 : a.shift.aten_threads);
  if (tl.on) hipLaunchKernelGGL(k_chain_shape<true>, dim3(batch_grid(n, 4096, 512, b.n), nf), dim3(1024), 0, s, b, f, mid, gamma, a, tl);
  else hipLaunchKernelGGL(k_chain_shape<false>, dim3(batch_grid(n, 4096, 512, b.n), nf), dim3(1024), 0, s, b, f, mid, gamma, a, tl);
  hipLaunchKernelGGL(k_chain_b2, dim3(batch_grid(n, 4096, 512, b.n), nf), dim3(1024), 0, s, b, H, W, a);
}

struct vd_own_slots { short v[VD_MAX_STEP]; unsigned char blank[VD_MAX_STEP]; };  // by value in the kernel arguments (1.5 KB): no staging copy
// ================================================================================================
// Measure / replay frame sharding: owners measure, every rank replays the scalar recurrences.
// k_shard2_r1: DepthPercentileEMA (:249-261) over the exchanged (q_lo, q_hi) of ALL frames of the step, in frame order.
//   etab[t] = {ema_lo, ema_den, collapse, have_prev} in force BEFORE frame t (etab[0] = carry-in), etab[t+1] after it.
// ================================================================================================
__global__ void k_shard2_r1(vd_dev_work* w, const float* __restrict__ q_all, int n, float* __restrict__ etab) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  vd3d_state* st = &w->st;
  etab[0] = w->ema_lo; etab[1] = w->ema_den; etab[2] = (float)w->collapse; etab[3] = (float)st->prev_depth_valid; etab[4] = st->ema_hi;
  for (int t = 0; t < n; ++t) {
    const float lo = q_all[2 * t], hi = q_all[2 * t + 1];
    int collapse;
    if ((hi - lo) < 1e-5f) collapse = 1;
    else {
      collapse = 0;
      if (!st->ema_valid) { st->ema_lo = lo; st->ema_hi = hi; st->ema_valid = 1; }
      else {
        const float al = (float)0.92, be = (float)(1 - 0.92);
        st->ema_lo = al * st->ema_lo + be * lo;
        st->ema_hi = al * st->ema_hi + be * hi;
      }
    }
    float* e = etab + VD_ETAB * (t + 1);
    e[0] = st->ema_lo; e[1] = (st->ema_hi - st->ema_lo) + (float)1e-6; e[2] = (float)collapse; e[3] = 1.f; e[4] = st->ema_hi;
  }
}
void vd_launch_shard2_r1(hipStream_t s, vd_dev_work* w, const float* q_all, int n, float* etab) {
  hipLaunchKernelGGL(k_shard2_r1, dim3(1), dim3(64), 0, s, w, q_all, n, etab);
}

// k_shard2_r2: every remaining recurrence of the loop body, in frame order, from the exchanged measurements
//   m_all[t] = {sum1, sum2, sum_mad, (s_norm | s1 << 32)}: dynamic parallax scale + ShiftSmoother (:412-427,470-477,1276,1308),
//   motion metric + FocalDepthTracker (:895-929), pixel_shift_cuda scalars incl. FloatingWindowTracker (:633-671),
//   ConvergenceEMA + FloatingBarEaser (:1390-1403).  Own frames get their constants patched into their slot.
__global__ void k_shard2_r2(vd_dev_work* w, const long long* __restrict__ m_all, const float* __restrict__ etab, vd_own_slots own, int n,
                            vd_dev_work* slot_work, vd_stage_args a) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  vd3d_state* st = &w->st;
  for (int t = 0; t < n; ++t) {
    const long long sum1 = m_all[4 * t], sum2 = m_all[4 * t + 1], sum_mad = m_all[4 * t + 2];
    const float s_norm = reinterpret_cast<const float*>(&m_all[4 * t + 3])[0], s1 = reinterpret_cast<const float*>(&m_all[4 * t + 3])[1];
    const int have_prev = (int)etab[VD_ETAB * t + 3];
    // ---- stage B1
    w->fs.s_norm = s_norm;
    const double nn = (double)a.n_crop;
    const double s1d = (double)sum1 / VD_FX, s2d = (double)sum2 / VD_FX;
    float mean = (float)(s1d / nn);
    if (a.aten_threads > 0 && a.n_crop > 0) mean = reinterpret_cast<const float*>(&m_all[4 * t + 2])[0] / (float)a.n_crop;
    const float var = (float)((s2d - s1d * s1d / nn) / (a.n_crop > 1 ? nn - 1.0 : 1.0));
    const float nv = vd_clamp(var / (mean + 1e-5f), 0.f, 1.f);
    const float scale = (float)0.90 + nv * (float)(1.15 - 0.90);
    w->fs.mean_c = mean; w->fs.var_c = var; w->fs.dyn_scale = (double)scale;
    double fg = a.shift.fg_shift, mg = a.shift.mg_shift, bg = a.shift.bg_shift;
    const double al = 0.15;
    if (!st->smooth_valid) { st->sm_fg = fg; st->sm_mg = mg; st->sm_bg = bg; st->smooth_valid = 1; }
    else {
      st->sm_fg = al * fg + (1 - al) * st->sm_fg;
      st->sm_mg = al * mg + (1 - al) * st->sm_mg;
      st->sm_bg = al * bg + (1 - al) * st->sm_bg;
    }
    const bool blank = own.blank[t] != 0;   // skip_blank_frames hit (:1278-1281): no ipd scaling, no focal / FloatingWindowTracker update
    fg = st->sm_fg; mg = st->sm_mg; bg = st->sm_bg;
    fg *= (double)scale; mg *= (double)scale; bg *= (double)scale;
    if (a.ipd_factor != 0.0 && !blank) { fg *= a.ipd_factor; mg *= a.ipd_factor; bg *= a.ipd_factor; }
    w->fg_d = fg; w->mg_d = mg; w->bg_d = bg;
    w->fs.mad = 0.f;
    double motion = 0.0;
    if (have_prev && !blank) {
      float mad = (float)(((double)sum_mad / VD_FX) / (double)a.n_eye);
      if (a.aten_threads > 0 && a.n_eye > 0) mad = reinterpret_cast<const float*>(&m_all[4 * t + 2])[1] / (float)a.n_eye;
      w->fs.mad = mad;
      const double m = (double)mad * 4.0;
      motion = m < 0.0 ? 0.0 : (m > 1.0 ? 1.0 : m);
    }
    if (!blank) w->fs.focal = focal_update(st, motion, (double)s_norm);
    else w->fs.focal = st->focal;
    w->focal = (float)w->fs.focal;
    // ---- stage B2
    w->fs.s1 = blank ? 0.f : s1;
    if (!blank) shift_scalars(w, a.shift, a.W, s1, fg, mg, bg);
    else { w->fs.zpo_raw = 0.f; w->fs.zpo = 0.0; }
    {
      const float s = s_norm;
      const float rz = ((((-s) * (float)w->fg_d) + ((-s) * (float)w->mg_d)) + (s * (float)w->bg_d)) / (float)((double)a.W / 2 + 1e-6);
      if (!st->conv_valid) { st->conv_val = (double)rz; st->conv_valid = 1; }
      else st->conv_val = 0.97 * st->conv_val + (1 - 0.97) * (double)rz;
      const double sz = st->conv_val;
      w->fs.stable_zero = sz;
      int bw = 0, side = 0;
      if (a.shift.enable_floating_window && a.shift.use_subject_tracking) {
        const int raw_bar = (int)(fabs(sz) * a.W * 0.75);
        st->bar_prev_width = (int)(0.85 * st->bar_prev_width + (1 - 0.85) * raw_bar);
        bw = st->bar_prev_width < 80 ? st->bar_prev_width : 80;
        bw = bw > 0 ? bw : 0;
        if (sz > 0.005) side = 1; else if (sz < -0.005) side = 2;
      }
      w->bar_width = bw; w->bar_side = side;
      w->fs.bar_width = bw; w->fs.bar_side = side;
    }
    const int sl = own.v[t];
    if (sl >= 0) {
      vd_dev_work* d = &slot_work[sl];
      d->fg = w->fg; d->mg = w->mg; d->bg = w->bg;
      d->zpo_f = w->zpo_f; d->have_zpo = w->have_zpo; d->msn = w->msn; d->conv = w->conv; d->have_conv = w->have_conv;
      d->focal = w->focal; d->bar_width = w->bar_width; d->bar_side = w->bar_side;
      const float* e = etab + VD_ETAB * (t + 1);
      vd3d_frame_scalars* fs = &d->fs;   // diagnostics of the slot = what vd3d_last_scalars reports for a sequential frame
      fs->ema_lo = e[0]; fs->ema_hi = e[4]; fs->collapse = (int)e[2];
      fs->mean_c = mean; fs->var_c = var; fs->dyn_scale = (double)scale; fs->s_norm = s_norm; fs->mad = w->fs.mad;
      fs->s1 = w->fs.s1; fs->zpo_raw = w->fs.zpo_raw; fs->zpo = w->fs.zpo; fs->focal = w->fs.focal; fs->stable_zero = w->fs.stable_zero;
      fs->bar_width = w->bar_width; fs->bar_side = w->bar_side;
    }
  }
  const float* e = etab + VD_ETAB * n;   // what a sequential run leaves behind for the next frame
  w->ema_lo = e[0]; w->ema_den = e[1]; w->collapse = (int)e[2];
  st->prev_depth_valid = 1;
}
void vd_launch_shard2_r2(hipStream_t s, vd_dev_work* w, const long long* m_all, const float* etab, const int* own_slot_host,
                         const uint8_t* blank_host_or_null, int n, vd_dev_work* slot_work, const vd_stage_args& a) {
  vd_own_slots own;
  for (int t = 0; t < VD_MAX_STEP; ++t) {
    own.v[t] = (short)(t < n ? own_slot_host[t] : -1);
    own.blank[t] = (unsigned char)((blank_host_or_null && t < n && blank_host_or_null[t]) ? 1 : 0);
  }
  hipLaunchKernelGGL(k_shard2_r2, dim3(1), dim3(64), 0, s, w, m_all, etab, own, n, slot_work, a);
}
