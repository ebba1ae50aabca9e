// vd3d_planes.hip -- plane kernels of the per-frame DIBR chain (v0: one reference stage per kernel,
// intermediate planes in HBM; arithmetic identical to oracle/vd3d_oracle.c).
//
// Reference call sites (core/render_3d.py): frame_to_tensor/depth_to_tensor :135-143, loop resize :1262-1263,
// TemporalDepthFilter :220-229, DepthPercentileEMA :241-262, shape_depth_for_pop :519-558, shift build :620-680,
// suppress_artifacts_with_edge_mask :198-216, grid_sample :684-701, feather_shift_edges :328-374,
// apply_dof_cuda :769-834, apply_color_grade :734-767, apply_side_mask :885-892, apply_sharpening :717-732,
// INTER_AREA fit :1409-1417, format_3d_output :837-883.
#include "vd3d_dev.h"
#include "vd3d_kernels.h"

// ------------------------------------------------------------------------------------------------
// shift plane: layer weights, zero-parallax, clamp, convergence, edge-mask suppression
// ------------------------------------------------------------------------------------------------
#define SH_TW 64
#define SH_TH 16
template <bool TAILS>   // round 5: ATen's scalar tails (vd_tails_of) -- those pixels take glibc's expf in the sigmoid and libm's pow in the layer weight
__global__ __launch_bounds__(256) void k_shift(const float* __restrict__ D, int H, int W, const vd_dev_work* __restrict__ w,
                                               vd_shift_consts c, float* __restrict__ S, vd_tails tl) {
  __shared__ __attribute__((aligned(16))) float em[SH_TH + 4][SH_TW + 4];
  __shared__ int2 rs14[64];
  const int x0 = blockIdx.x * SH_TW, y0 = blockIdx.y * SH_TH;
  if (c.edge) {
    vd_stage_rs14(rs14, threadIdx.x, 256);
    __syncthreads();
    for (int t = threadIdx.x; t < (SH_TH + 4) * (SH_TW + 4); t += 256) {
      const int ty = t / (SH_TW + 4), tx = t - ty * (SH_TW + 4);
      const int y = y0 - 2 + ty, x = x0 - 2 + tx;
      float e = 0.f;  // zero padding of avg_pool2d
      if (y >= 0 && y < H && x >= 0 && x < W) {
        const float cc = D[(size_t)y * W + x];
        const float dx = x > 0 ? fabsf(cc - D[(size_t)y * W + x - 1]) : 0.f;
        const float dy = y > 0 ? fabsf(cc - D[(size_t)(y - 1) * W + x]) : 0.f;
        const float g = vd_sqrt_torch(dx * dx + dy * dy, rs14);      // torch.sqrt / torch.sigmoid: the CPU libraries' values
        const float z = ((g - (float)0.02) * c.fs) * 5.f;
        float sg;
        if (TAILS && vd_in_tail(tl, (unsigned)y * (unsigned)W + (unsigned)x)) sg = vd_sigmoid_tail(z);
        else sg = vd_sigmoid_torch(z);
        e = 1.f - sg;
      }
      em[ty][tx] = e;
    }
    __syncthreads();
  }
  const float fgf = w->fg, mgf = w->mg, bgf = w->bg;
  // one thread = 4 consecutive pixels of one row (SH_TH * SH_TW / 4 == 256 threads)
  const int ty = threadIdx.x / (SH_TW / 4), tx = (threadIdx.x - ty * (SH_TW / 4)) * 4;
  const int y = y0 + ty;
  if (y >= H) return;
  vd_f4 s5 = {0.f, 0.f, 0.f, 0.f};
  if (c.edge) {
    // avg_pool2d(5, 1, 2) in ATen's order (cpu_avg_pool2d): ONE float32 running sum over the window, row-major; the zero padding
    // adds exact zeros.  Four outputs share the 8-column window of each tile row (two 16-byte LDS reads).
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const vd_f4 a = *reinterpret_cast<const vd_f4*>(&em[ty + i][tx]), b = *reinterpret_cast<const vd_f4*>(&em[ty + i][tx + 4]);
      const float win[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
      for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) s5[q] += win[q + j];
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int x = x0 + tx + q;
    if (x >= W) break;
    const float Dv = D[(size_t)y * W + x];
    float p15;
    if (TAILS && vd_in_tail(tl, (unsigned)y * (unsigned)W + (unsigned)x)) p15 = vd_pow_tail(1.0f - Dv, 1.5);
    else p15 = vd_pow15_torch(1.0f - Dv);
    const float fgw = vd_clamp(p15, 0.f, 1.f);
    const float mgw = vd_clamp(1.0f - fabsf(Dv - c.mid) * 3.0f, 0.f, 1.f);
    const float bgw = vd_clamp(Dv, 0.f, 1.f);
    const float raw = ((fgw * fgf) * c.fgm + mgw * mgf) + (bgw * bgf) * c.bgm;
    float sft = (raw * c.pb) / c.half_width;
    if (w->have_zpo) sft = sft - w->zpo_f;
    sft = vd_clamp(sft, -w->msn, w->msn);
    if (w->have_conv) sft = sft - w->conv;
    if (c.edge) {
      const float sm = s5[q] / 25.f;
      sft = c.ma * sft + c.mb * (sft * sm);
    }
    S[(size_t)y * W + x] = sft;
  }
}
vd_shift_consts vd_shift_consts_of(const vd3d_shift_params& p, int W) {
  vd_shift_consts c;
  c.mid = (float)p.depth_pop_mid;
  c.fgm = (float)p.fg_pop_multiplier; c.bgm = (float)p.bg_push_multiplier; c.pb = (float)p.parallax_balance;
  c.half_width = (float)((double)W / 2.0);
  c.fs = (float)p.feather_strength;
  double ms = p.feather_strength / 10.0;  // np.clip(feather_strength/10, .05, .3)
  ms = ms < 0.05 ? 0.05 : (ms > 0.3 ? 0.3 : ms);
  c.ma = (float)(1.0 - ms); c.mb = (float)ms;
  c.edge = p.enable_edge_masking ? 1 : 0;
  c.fg = c.mg = c.bg = 0.f;
  return c;
}
void vd_launch_shift(hipStream_t s, const float* D, int H, int W, const vd_dev_work* w, const vd3d_shift_params& p, float* S) {
  const vd_shift_consts c = vd_shift_consts_of(p, W);
  const vd_tails tl = vd_tails_of((unsigned long long)H * W, p.aten_threads);
  if (tl.on) hipLaunchKernelGGL(k_shift<true>, dim3((W + SH_TW - 1) / SH_TW, (H + SH_TH - 1) / SH_TH), dim3(256), 0, s, D, H, W, w, c, S, tl);
  else hipLaunchKernelGGL(k_shift<false>, dim3((W + SH_TW - 1) / SH_TW, (H + SH_TH - 1) / SH_TH), dim3(256), 0, s, D, H, W, w, c, S, tl);
}

// ------------------------------------------------------------------------------------------------
// warped-depth gradient mask e2 per eye (feather_shift_edges :347-352 on grid_sample(D))
// ------------------------------------------------------------------------------------------------
VD_DEV float warped_depth(const float* __restrict__ D, const float* __restrict__ S, int H, int W, int y, int x, float sign) {
  const float s = S[(size_t)y * W + x];
  float gx = vd_lin11(W, x);
  gx = sign > 0.f ? gx + s : gx - s;
  const vd_gs g = vd_gs_params(gx, vd_lin11(H, y), W, H);
  const float* r0 = D + (size_t)g.yn * W;
  const float vnw = r0[g.xw], vne = g.e_ok ? r0[g.xw + 1] : 0.f;
  float vsw = 0.f, vse = 0.f;
  if (g.s_ok) { vsw = r0[W + g.xw]; vse = g.e_ok ? r0[W + g.xw + 1] : 0.f; }
  return vd_gs_combine(g, vnw, vne, vsw, vse);
}
__global__ __launch_bounds__(256) void k_e2(const float* __restrict__ D, const float* __restrict__ S, int H, int W, float fs,
                                            float* __restrict__ e2L, float* __restrict__ e2R) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= W || y >= H) return;
#pragma unroll
  for (int eye = 0; eye < 2; ++eye) {
    const float sign = eye == 0 ? 1.f : -1.f;
    const float c = warped_depth(D, S, H, W, y, x, sign);
    const float gx = x > 0 ? c - warped_depth(D, S, H, W, y, x - 1, sign) : 0.f;
    const float gy = y > 0 ? c - warped_depth(D, S, H, W, y - 1, x, sign) : 0.f;
    const float g = vd_sqrt_torch(gx * gx + gy * gy, c_vd_rs14);
    (eye == 0 ? e2L : e2R)[(size_t)y * W + x] = vd_clamp(g * fs, 0.f, 1.f);
  }
}
void vd_launch_e2(hipStream_t s, const float* D, const float* S, int H, int W, float fs, float* e2L, float* e2R) {
  hipLaunchKernelGGL(k_e2, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), 0, s, D, S, H, W, fs, e2L, e2R);
}

// ------------------------------------------------------------------------------------------------
// k x k zero-padded window average in ATen's order (cpu_avg_pool2d: one float32 running sum over the window, row-major), window
// start -(k/2).  Fallback kernel of the unfused path (blur sizes / frame sizes the fused warp kernel refuses).
// ------------------------------------------------------------------------------------------------
#define PL_TW 64
#define PL_TH 16
__global__ __launch_bounds__(256) void k_pool(const float* __restrict__ e2L, const float* __restrict__ e2R, int H, int W, int k,
                                              float* __restrict__ bL, float* __restrict__ bR) {
  extern __shared__ float lds[];
  const int r = k / 2;
  const int tw = PL_TW + k - 1, th = PL_TH + k - 1;
  float* tile = lds;                 // [th][tw]
  const int x0 = blockIdx.x * PL_TW, y0 = blockIdx.y * PL_TH;
  const float div = (float)(k * k);
  for (int eye = 0; eye < 2; ++eye) {
    const float* src = eye == 0 ? e2L : e2R;
    float* dst = eye == 0 ? bL : bR;
    for (int t = threadIdx.x; t < th * tw; t += 256) {
      const int ty = t / tw, tx = t - ty * tw;
      const int y = y0 - r + ty, x = x0 - r + tx;
      tile[t] = (y >= 0 && y < H && x >= 0 && x < W) ? src[(size_t)y * W + x] : 0.f;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < PL_TH * PL_TW; t += 256) {
      const int ty = t / PL_TW, tx = t - ty * PL_TW;
      const int y = y0 + ty, x = x0 + tx;
      if (y >= H || x >= W) continue;
      float s = 0.f;
      for (int i = 0; i < k; ++i)
        for (int j = 0; j < k; ++j) s += tile[(ty + i) * tw + tx + j];
      dst[(size_t)y * W + x] = s / div;
    }
    __syncthreads();
  }
}
void vd_launch_pool(hipStream_t s, const float* e2L, const float* e2R, int H, int W, int k, float* bL, float* bR) {
  const int tw = PL_TW + k - 1, th = PL_TH + k - 1;
  size_t lds = sizeof(float) * ((size_t)th * tw);
  if (lds > 64 * 1024) {   // large blur_ksize: the tile needs the opt-in LDS range
    static bool attr[64] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr[dev]) {
      (void)hipFuncSetAttribute((const void*)k_pool, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr[dev] = true;
    }
  }
  hipLaunchKernelGGL(k_pool, dim3((W + PL_TW - 1) / PL_TW, (H + PL_TH - 1) / PL_TH), dim3(256), lds, s, e2L, e2R, H, W, k, bL, bR);
}

// ------------------------------------------------------------------------------------------------
// warp + feather blend + tensor_to_frame
// ------------------------------------------------------------------------------------------------
VD_DEV float rgb_at(const float* __restrict__ pl, int ih, int iw, int H, int W, int y, int x) {  // F.interpolate :595
  if (ih == H && iw == W) return pl[(size_t)y * W + x];
  const vd_tap ty = vd_interp_tap(ih, H, y), tx = vd_interp_tap(iw, W, x);
  const float* r0 = pl + (size_t)ty.i0 * iw;
  const float* r1 = pl + (size_t)ty.i1 * iw;
  return vd_bilerp(r0[tx.i0], r0[tx.i1], r1[tx.i0], r1[tx.i1], tx.w0, tx.w1, ty.w0, ty.w1);
}
__global__ __launch_bounds__(256) void k_warp(const float* __restrict__ rgb, int ih, int iw, const float* __restrict__ S,
                                              const float* __restrict__ bL, const float* __restrict__ bR, int H, int W, int feather,
                                              uint8_t* __restrict__ L, uint8_t* __restrict__ R) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= W || y >= H) return;
  const size_t o = (size_t)y * W + x, ni = (size_t)ih * iw;
  const float s = S[o];
  const float gx0 = vd_lin11(W, x), gy = vd_lin11(H, y);
  float orig[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) orig[c] = rgb_at(rgb + c * ni, ih, iw, H, W, y, x);
#pragma unroll
  for (int eye = 0; eye < 2; ++eye) {
    const vd_gs g = vd_gs_params(eye == 0 ? gx0 + s : gx0 - s, gy, W, H);
    const float b = feather ? (eye == 0 ? bL : bR)[o] : 0.f;
    uint8_t* out = (eye == 0 ? L : R) + o * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* pl = rgb + c * ni;
      const float vnw = rgb_at(pl, ih, iw, H, W, g.yn, g.xw);
      const float vne = g.e_ok ? rgb_at(pl, ih, iw, H, W, g.yn, g.xw + 1) : 0.f;
      float vsw = 0.f, vse = 0.f;
      if (g.s_ok) {
        vsw = rgb_at(pl, ih, iw, H, W, g.yn + 1, g.xw);
        vse = g.e_ok ? rgb_at(pl, ih, iw, H, W, g.yn + 1, g.xw + 1) : 0.f;
      }
      float v = vd_gs_combine(g, vnw, vne, vsw, vse);
      if (feather) v = vd_clamp(v * (1.0f - b) + orig[c] * b, 0.f, 1.f);
      out[2 - c] = (uint8_t)(v * 255.0f);  // truncation; RGB -> BGR
    }
  }
}
void vd_launch_warp(hipStream_t s, const float* rgb, int ih, int iw, const float* S, const float* bL, const float* bR, int H, int W,
                    int feather, uint8_t* L, uint8_t* R) {
  hipLaunchKernelGGL(k_warp, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), 0, s, rgb, ih, iw, S, bL, bR, H, W, feather, L, R);
}

// ------------------------------------------------------------------------------------------------
// DOF (separable Gaussian levels in LDS) + colour grade + truncate + side bars
// ------------------------------------------------------------------------------------------------
#define DF_TW 32
#define DF_TH 16
#define DF_RMAX 15
// Separable levels (vd3d_render_params::dof_dense_conv = 0: a throughput mode outside the 1-LSB bar); the reference's dense order is k_dof_grade4 below.
__global__ __launch_bounds__(512) void k_dof_grade(const uint8_t* __restrict__ eye_in, const float* __restrict__ dn, int eh, int ew,
                                                   int H, int W, vd_finish_consts fc, const vd_dev_work* __restrict__ w,
                                                   float focal_override, int use_override, int bar_width_o, int bar_side_o,
                                                   uint8_t* __restrict__ eye_out) {
  extern __shared__ float lds[];
  const int R = fc.nlev ? fc.ksz[fc.nlev - 1] / 2 : 0;
  const int tw = DF_TW + 2 * R, th = DF_TH + 2 * R;
  float* tile = lds;                        // [3][th][tw]
  float* vb = lds + (size_t)3 * th * tw;    // [3][DF_TH][tw] vertical sums (the vertical pass runs first)
  const int x0 = blockIdx.x * DF_TW, y0 = blockIdx.y * DF_TH;
  for (int t = threadIdx.x; t < th * tw; t += 512) {
    const int ty = t / tw, tx = t - ty * tw;
    const int y = vd_reflect(y0 - R + ty, H), x = vd_reflect(x0 - R + tx, W);
    const uint8_t* px = eye_in + ((size_t)y * W + x) * 3;
    tile[0 * th * tw + t] = vd_u8_unit((float)px[2]);
    tile[1 * th * tw + t] = vd_u8_unit((float)px[1]);
    tile[2 * th * tw + t] = vd_u8_unit((float)px[0]);
  }
  __syncthreads();
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int x = x0 + tx, y = y0 + ty;
  const bool live = x < W && y < H;
  float vlo[3], vhi[3];
  int lo = 0;
  float alpha = 0.f;
  if (fc.nlev) {
    const float focal = use_override ? focal_override : w->focal;
    float dd = 0.f;
    if (live) {  // depth_for_dof = F.interpolate(depth_tensor -> (H,W)) :1347-1350
      if (eh == H && ew == W) dd = dn[(size_t)y * W + x];
      else {
        const vd_tap ay = vd_interp_tap(eh, H, y), ax = vd_interp_tap(ew, W, x);
        const float* r0 = dn + (size_t)ay.i0 * ew;
        const float* r1 = dn + (size_t)ay.i1 * ew;
        dd = vd_bilerp(r0[ax.i0], r0[ax.i1], r1[ax.i0], r1[ax.i1], ax.w0, ax.w1, ay.w0, ay.w1);
      }
    }
    const float bw = vd_clamp(fabsf(dd - focal) / fc.fw, 0.f, 1.f);
    const float bi = vd_clamp(bw * (float)fc.nlev, 0.f, fc.imax);
    lo = (int)floorf(bi);
    lo = lo > fc.nlev - 1 ? fc.nlev - 1 : (lo < 0 ? 0 : lo);
    alpha = bi - (float)lo;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) { vlo[c] = tile[c * th * tw + (ty + R) * tw + tx + R]; vhi[c] = vlo[c]; }
  for (int l = 0; l < fc.nlev; ++l) {  // level l+1 of the reference's stack
    const int k = fc.ksz[l], r = k / 2;
    __syncthreads();
    for (int t = threadIdx.x; t < 3 * DF_TH * tw; t += 512) {   // vertical sums for every tile column
      const int c = t / (DF_TH * tw), rem = t - c * DF_TH * tw;
      const int vy = rem / tw, vx = rem - vy * tw;
      const float* col = tile + (size_t)c * th * tw + (vy + R - r) * tw + vx;
      vb[t] = vd_gauss_sym_rt(fc.kern[l], k, col, tw);
    }
    __syncthreads();
    if (l + 1 == lo || l + 1 == lo + 1) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float* row = vb + (size_t)c * DF_TH * tw + ty * tw + tx + R - r;
        const float s = vd_gauss_sym_rt(fc.kern[l], k, row, 1);
        if (l + 1 == lo) vlo[c] = s; else vhi[c] = s;
      }
    }
  }
  if (!live) return;
  float rgbv[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v = vlo[c];
    if (fc.nlev) v = vd_clamp((1.0f - alpha) * vlo[c] + alpha * vhi[c], 0.f, 1.f);
    rgbv[c] = v;
  }
  // apply_color_grade :750-767
  const float luma = ((float)0.2126 * rgbv[0] + (float)0.7152 * rgbv[1]) + (float)0.0722 * rgbv[2];
  const int bar_w = use_override ? bar_width_o : w->bar_width, bar_s = use_override ? bar_side_o : w->bar_side;
  const bool masked = bar_w > 0 && ((bar_s == 2 && x < bar_w) || (bar_s == 1 && x >= W - bar_w));
  uint8_t* out = eye_out + ((size_t)y * W + x) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v = luma + (rgbv[c] - luma) * fc.sat;
    v = 0.5f + (v - 0.5f) * fc.con;
    v = v + fc.bri;
    v = vd_clamp(v, 0.f, 1.f);
    out[2 - c] = masked ? (uint8_t)0 : (uint8_t)(v * 255.0f);
  }
}
// Dense levels (the reference's order: one k x k window per output, taps row-major, one FMA per tap from 0, weight = fl(k1[i] * k1[j]);
// vd3d_render_params::dof_dense_conv = 1, the default) for the Gaussians the fused finishing kernel does not take: more than 9 taps (dof_strength
// > 2, up to 31 taps; a VR canvas or a fractional fit alone keeps the fused kernel, vd3d_api.hip run_finish).  Round 4: one thread = 4 consecutive
// pixels (64 x 32 tile, 512 threads).  For every tap row the K + 3 window columns are read ONCE (a single LDS dword each) and feed the four
// outputs' chains with a sliding window of four weights; a window column outside an output's K taps would meet a zero weight -- fma(v, 0, acc)
// is exact here, acc never is -0 -- so it can be issued (the run-time loop) or left out (the instantiations): same bits.  (One pixel per thread
// until round 3: 1 387 us per 4K frame pair at dof_strength 3.0 against the fused kernel's 252 at 2.0.)  Each output's taps arrive in ascending
// (i, j) order.
// Second pass: the tap count is a template parameter (one instantiation per odd K <= 21, the GUI slider's range) and the weights come from a device table
// (vd3d_ctx::wk_tabs: [level][row i][32] = fl(k1[i] * k1[j]), row pitch 32 floats) through wave-uniform loads, i.e. as SCALAR operands of the
// FMAs: the window of K + 3 columns sits in registers, the sliding weights are register names, the zero-weight products at the row ends are
// not issued (exact: see above) -- 4 K FMAs per tap row and channel next to (K + 3) / 2 ds_read2 and K / 4 scalar loads, where the run-time
// loop spent 12 instructions per 4 FMAs (two LDS reads, three weight moves, loop control).
#define D4_TW 64
#define D4_TH 32
#define D4_WP 32                    // row pitch of the weight table
#define D4_WL (31 * D4_WP)          // floats per level
// level `level` arrives: it replaces the running value of the pixels whose lo level it is, and is blended in where it is lo + 1 (:822-834)
VD_DEV void d4_fold(float a0, float a1, float a2, float a3, int level, const int lo[4], const float alpha[4], float* res) {
  const float acc[4] = {a0, a1, a2, a3};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (level == lo[q]) res[q] = acc[q];
    if (level == lo[q] + 1) res[q] = (1.0f - alpha[q]) * res[q] + alpha[q] * acc[q];
  }
}
template <int K>
VD_DEV void d4_level(const float* __restrict__ tile, int th, int twp, int ty, int tx, int R, const float* __restrict__ wl, int level, const int lo[4],
                     const float alpha[4], float res[3][4]) {
  constexpr int r = K / 2;
#pragma unroll 1
  for (int c = 0; c < 3; ++c) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 1
    for (int i = 0; i < K; ++i) {
      const float* t0 = tile + (size_t)(c * th + ty + R - r + i) * twp + tx + R - r;   // window column 0 = tap 0 of pixel 0
      const float* wr = wl + i * D4_WP;                                                // wave-uniform: scalar loads
      float v[K + 3], wv[K];
#pragma unroll
      for (int j = 0; j < K; ++j) wv[j] = wr[j];
#pragma unroll
      for (int jj = 0; jj < K + 3; ++jj) v[jj] = t0[jj];
#pragma unroll
      for (int jj = 0; jj < K + 3; ++jj) {   // window column jj: tap jj of pixel 0, jj - 1 of pixel 1, ... (each output's taps in ascending order)
        if (jj < K) a0 = vd_fma(v[jj], wv[jj], a0);
        if (jj >= 1 && jj - 1 < K) a1 = vd_fma(v[jj], wv[jj - 1], a1);
        if (jj >= 2 && jj - 2 < K) a2 = vd_fma(v[jj], wv[jj - 2], a2);
        if (jj >= 3 && jj - 3 < K) a3 = vd_fma(v[jj], wv[jj - 3], a3);
      }
    }
    d4_fold(a0, a1, a2, a3, level, lo, alpha, res[c]);
  }
}
// any tap count (run-time loops; the instantiations above cover the GUI's range, this one keeps the register budget of the kernel theirs)
VD_DEV void d4_level_any(int k, const float* __restrict__ tile, int th, int twp, int ty, int tx, int R, const float* __restrict__ wl, int level,
                         const int lo[4], const float alpha[4], float res[3][4]) {
  const int r = k / 2;
  for (int c = 0; c < 3; ++c) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int i = 0; i < k; ++i) {
      const float* t0 = tile + (size_t)(c * th + ty + R - r + i) * twp + tx + R - r;
      const float* wr = wl + i * D4_WP;
      float w1 = 0.f, w2 = 0.f, w3 = 0.f;
      for (int jj = 0; jj < k + 3; ++jj) {   // sliding window of four weights; the zero products at the row ends are exact (acc never is -0)
        const float v = t0[jj], w0 = jj < k ? wr[jj] : 0.f;
        a0 = vd_fma(v, w0, a0); a1 = vd_fma(v, w1, a1); a2 = vd_fma(v, w2, a2); a3 = vd_fma(v, w3, a3);
        w3 = w2; w2 = w1; w1 = w0;
      }
    }
    d4_fold(a0, a1, a2, a3, level, lo, alpha, res[c]);
  }
}
// Round 6: both eyes in one launch (blockIdx.z; the second eye's workgroups fill the first one's tail), and the tile load in straight-line batches of D4_LU pixels per
// thread -- the one-pixel-per-iteration loop it replaces was a chain of th * tw / 512 = 7 .. 12 dependent global round trips per thread (the wait pattern W1's analysis
// found, profiles/r06_w1_phases.md).  m_tw = ceil(2^32 / tw): umulhi(t, m_tw) == t / tw for t, tw < 2^16.
#define D4_LU 4
__global__ __launch_bounds__(512) void k_dof_grade4(const uint8_t* __restrict__ eyeL_in, const uint8_t* __restrict__ eyeR_in, const float* __restrict__ dn, int eh, int ew,
                                                    int H, int W, vd_finish_consts fc, const vd_dev_work* __restrict__ w,
                                                    float focal_override, int use_override, int bar_width_o, int bar_side_o,
                                                    uint8_t* __restrict__ eyeL_out, uint8_t* __restrict__ eyeR_out, int twp, const float* __restrict__ wtab, uint32_t m_tw) {
  extern __shared__ float lds[];
  const uint8_t* __restrict__ eye_in = blockIdx.z ? eyeR_in : eyeL_in;
  uint8_t* __restrict__ eye_out = blockIdx.z ? eyeR_out : eyeL_out;
  const int R = fc.nlev ? fc.ksz[fc.nlev - 1] / 2 : 0;
  const int tw = D4_TW + 2 * R, th = D4_TH + 2 * R;
  float* tile = lds;                          // [3][th][twp], twp = 1 mod 4: the four rows of a wave's lanes fall into distinct banks
  const int x0 = blockIdx.x * D4_TW, y0 = blockIdx.y * D4_TH;
  const int ntile = th * tw;
#pragma unroll 1
  for (int t0 = threadIdx.x; t0 < ntile; t0 += D4_LU * 512) {
    uint8_t b[D4_LU][3]; int dst[D4_LU];
#pragma unroll
    for (int j = 0; j < D4_LU; ++j) {
      const int t = t0 + j * 512, tc = min(t, ntile - 1);           // past the end: the last pixel again (loaded, not stored)
      const int ty = (int)__umulhi((uint32_t)tc, m_tw), tx = tc - ty * tw;
      const int y = vd_reflect(y0 - R + ty, H), x = vd_reflect(x0 - R + tx, W);
      const uint8_t* px = eye_in + ((size_t)y * W + x) * 3;
      b[j][0] = px[0]; b[j][1] = px[1]; b[j][2] = px[2];
      dst[j] = t < ntile ? ty * twp + tx : -1;
    }
#pragma unroll
    for (int j = 0; j < D4_LU; ++j) {
      if (dst[j] >= 0) {
        tile[0 * th * twp + dst[j]] = vd_u8_unit((float)b[j][2]);
        tile[1 * th * twp + dst[j]] = vd_u8_unit((float)b[j][1]);
        tile[2 * th * twp + dst[j]] = vd_u8_unit((float)b[j][0]);
      }
    }
  }
  __syncthreads();
  const int ty = threadIdx.x >> 4, tx = (threadIdx.x & 15) * 4;
  const int y = y0 + ty, xs = x0 + tx;
  if (y >= H || xs >= W) return;
  int lo[4] = {0, 0, 0, 0};
  float alpha[4] = {0.f, 0.f, 0.f, 0.f};
  int need = 0;   // bit l: level l is the lo or the hi level of one of the four pixels
  if (fc.nlev) {
    const float focal = use_override ? focal_override : w->focal;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int x = min(xs + q, W - 1);
      float dd;   // depth_for_dof = F.interpolate(depth_tensor -> (H,W)) :1347-1350
      if (eh == H && ew == W) dd = dn[(size_t)y * W + x];
      else {
        const vd_tap ay = vd_interp_tap(eh, H, y), ax = vd_interp_tap(ew, W, x);
        const float* r0 = dn + (size_t)ay.i0 * ew;
        const float* r1 = dn + (size_t)ay.i1 * ew;
        dd = vd_bilerp(r0[ax.i0], r0[ax.i1], r1[ax.i0], r1[ax.i1], ax.w0, ax.w1, ay.w0, ay.w1);
      }
      const float bw = vd_clamp(fabsf(dd - focal) / fc.fw, 0.f, 1.f);
      const float bi = vd_clamp(bw * (float)fc.nlev, 0.f, fc.imax);
      int l = (int)floorf(bi);
      l = l > fc.nlev - 1 ? fc.nlev - 1 : (l < 0 ? 0 : l);
      lo[q] = l; alpha[q] = bi - (float)l;
      need |= (1 << l) | (1 << (l + 1));
    }
  }
  float res[3][4];   // level 0 = the pixel itself; replaced by level lo, then blended with level lo + 1 (levels arrive in ascending order)
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int q = 0; q < 4; ++q) res[c][q] = tile[(c * th + ty + R) * twp + tx + R + q];
  for (int l = 0; l < fc.nlev; ++l) {
    if (!((need >> (l + 1)) & 1)) continue;
    const float* wl = wtab + (size_t)l * D4_WL;
    switch (fc.ksz[l]) {   // compile-time tap count per instantiation
#define D4_CASE(K) case K: d4_level<K>(tile, th, twp, ty, tx, R, wl, l + 1, lo, alpha, res); break;
      D4_CASE(3) D4_CASE(5) D4_CASE(7) D4_CASE(9) D4_CASE(11) D4_CASE(13) D4_CASE(15) D4_CASE(17) D4_CASE(19) D4_CASE(21)
#undef D4_CASE
      default: d4_level_any(fc.ksz[l], tile, th, twp, ty, tx, R, wl, l + 1, lo, alpha, res); break;   // 23 .. 31 taps (dof_strength > 5, beyond the GUI's slider)
    }
  }
  const int bar_w = use_override ? bar_width_o : w->bar_width, bar_s = use_override ? bar_side_o : w->bar_side;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int x = xs + q;
    if (x >= W) break;
    float rgbv[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) rgbv[c] = fc.nlev ? vd_clamp(res[c][q], 0.f, 1.f) : res[c][q];
    // apply_color_grade :750-767
    const float luma = ((float)0.2126 * rgbv[0] + (float)0.7152 * rgbv[1]) + (float)0.0722 * rgbv[2];
    const bool masked = bar_w > 0 && ((bar_s == 2 && x < bar_w) || (bar_s == 1 && x >= W - bar_w));
    uint8_t* out = eye_out + ((size_t)y * W + x) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float v = luma + (rgbv[c] - luma) * fc.sat;
      v = 0.5f + (v - 0.5f) * fc.con;
      v = v + fc.bri;
      v = vd_clamp(v, 0.f, 1.f);
      out[2 - c] = masked ? (uint8_t)0 : (uint8_t)(v * 255.0f);
    }
  }
}
// dense levels, both eyes (R_in / R_out may be NULL: one eye)
void vd_launch_dof_grade_dense(hipStream_t s, const uint8_t* L_in, const uint8_t* R_in, const float* dn, int eh, int ew, int H, int W,
                               const vd_finish_consts& fc, const vd_dev_work* w, float focal_override, int use_override,
                               int bar_width, int bar_side, uint8_t* L_out, uint8_t* R_out, const float* wtab) {
  const int R = fc.nlev ? fc.ksz[fc.nlev - 1] / 2 : 0;
  const int tw = D4_TW + 2 * R, th = D4_TH + 2 * R;
  int twp = tw;
  while ((twp & 3) != 1) ++twp;
  const size_t lds = sizeof(float) * 3 * (size_t)th * twp;
  static bool attr[64] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && !attr[dev]) {
    (void)hipFuncSetAttribute((const void*)k_dof_grade4, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr[dev] = true;
  }
  const uint32_t m_tw = (uint32_t)(((1ull << 32) + (uint64_t)tw - 1) / (uint64_t)tw);
  hipLaunchKernelGGL(k_dof_grade4, dim3((W + D4_TW - 1) / D4_TW, (H + D4_TH - 1) / D4_TH, R_in ? 2 : 1), dim3(512), lds, s, L_in, R_in, dn, eh, ew, H, W, fc, w,
                     focal_override, use_override, bar_width, bar_side, L_out, R_out, twp, wtab, m_tw);
}
void vd_launch_dof_grade(hipStream_t s, const uint8_t* eye_in, const float* dn, int eh, int ew, int H, int W,
                         const vd_finish_consts& fc, const vd_dev_work* w, float focal_override, int use_override,
                         int bar_width, int bar_side, uint8_t* eye_out, int dense, const float* wtab) {
  const int R = fc.nlev ? fc.ksz[fc.nlev - 1] / 2 : 0;
  if (dense) {
    vd_launch_dof_grade_dense(s, eye_in, nullptr, dn, eh, ew, H, W, fc, w, focal_override, use_override, bar_width, bar_side, eye_out, nullptr, wtab);
    return;
  }
  const int tw = DF_TW + 2 * R, th = DF_TH + 2 * R;
  const dim3 g((W + DF_TW - 1) / DF_TW, (H + DF_TH - 1) / DF_TH);
  const size_t lds = sizeof(float) * 3 * ((size_t)th * tw + (size_t)DF_TH * tw);
  hipLaunchKernelGGL(k_dof_grade, g, dim3(512), lds, s, eye_in, dn, eh, ew, H, W, fc, w, focal_override, use_override, bar_width,
                     bar_side, eye_out);
}

// ------------------------------------------------------------------------------------------------
// sharpen (filter2D 3x3) + fit (integer-ratio INTER_AREA / pad) + mux
// ------------------------------------------------------------------------------------------------
// pre > 0: g already holds SHARPENED pixels with a row pitch of `pre` pixels (the fused finishing kernel ran 1:1 into a side-by-side scratch,
// round 4): the fit / mux below is all that is left to do
template <bool PRE>
VD_DEV uint8_t sharp_at(const uint8_t* __restrict__ g, int H, int W, int pre, int y, int x, int c, float kn, float kc) {
  if (PRE) return g[((size_t)y * pre + x) * 3 + c];
  const int yu = vd_reflect(y - 1, H), yd = vd_reflect(y + 1, H), xl = vd_reflect(x - 1, W), xr = vd_reflect(x + 1, W);
  float s = 0.f;
  s += kn * (float)g[((size_t)yu * W + x) * 3 + c];
  s += kn * (float)g[((size_t)y * W + xl) * 3 + c];
  s += kc * (float)g[((size_t)y * W + x) * 3 + c];
  s += kn * (float)g[((size_t)y * W + xr) * 3 + c];
  s += kn * (float)g[((size_t)yd * W + x) * 3 + c];
  return vd_sat_rne_u8(s);
}
struct vd_mux_geom {
  int H, W;            // sharpened eye size (warp size)
  int fit_w, fit_h;    // padded eye canvas
  int in_w, in_h;      // resized image placed inside the canvas
  int xo, yo;          // its offset
  int fx, fy;          // integer down-scale factors
  int out_w, out_h, format;
  int frac;            // 1: non-integer (or mixed) INTER_AREA down-scale, generic area table path; 2: some dimension up-scales
  int pre;             // > 0: the inputs are sharpened already, row pitch in pixels (see sharp_at)
  double sx, sy;       // OpenCV's scale = 1./((double)dsize/ssize)
};
template <bool PRE>
__global__ __launch_bounds__(256) void k_sharp_mux(const uint8_t* __restrict__ gL, const uint8_t* __restrict__ gR, vd_mux_geom m,
                                                   float kn, float kc, uint8_t* __restrict__ out) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= m.fit_w || y >= m.fit_h) return;
  uint8_t px[2][3];
  const int ix = x - m.xo, iy = y - m.yo;
  const bool inside = ix >= 0 && ix < m.in_w && iy >= 0 && iy < m.in_h;
  // Round 6, the VR path (pre-sharpened eyes, fractional INTER_AREA ratio): the area taps are computed ONCE per output pixel and every source pixel's three bytes are
  // read together -- the generic loop below evaluates vd_area_taps and walks the ny x nx window per eye AND per channel.  Per channel the same float32 sums in the
  // same order (h = h + s * alpha over the row, acc = acc + h * beta over the rows).
  const bool fast_area = PRE && m.frac == 1;
  if (fast_area) {
    float ax[VD_AREA_MAXT], ay[VD_AREA_MAXT];
    int x0s = 0, y0s = 0, nx = 0, ny = 0;
    if (inside) { nx = vd_area_taps(m.W, m.sx, ix, &x0s, ax); ny = vd_area_taps(m.H, m.sy, iy, &y0s, ay); }
#pragma unroll
    for (int eye = 0; eye < 2; ++eye) {
      const uint8_t* g = eye == 0 ? gL : gR;
      float acc[3] = {0.f, 0.f, 0.f};
      for (int j = 0; j < ny; ++j) {
        float h[3] = {0.f, 0.f, 0.f};
        const uint8_t* row = g + ((size_t)(y0s + j) * m.pre + x0s) * 3;
        for (int k = 0; k < nx; ++k) {
          const uint8_t b0 = row[3 * k], b1 = row[3 * k + 1], b2 = row[3 * k + 2];
          h[0] = h[0] + (float)b0 * ax[k]; h[1] = h[1] + (float)b1 * ax[k]; h[2] = h[2] + (float)b2 * ax[k];
        }
        acc[0] = acc[0] + h[0] * ay[j]; acc[1] = acc[1] + h[1] * ay[j]; acc[2] = acc[2] + h[2] * ay[j];
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) px[eye][c] = inside ? vd_sat_rne_u8(acc[c]) : (uint8_t)0;
    }
  }
#pragma unroll
  for (int eye = 0; eye < 2 && !fast_area; ++eye) {
    const uint8_t* g = eye == 0 ? gL : gR;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      uint8_t v = 0;
      if (inside && m.frac == 2) {  // INTER_AREA asked to up-scale: hal::resize's linear path with area-mode coefficients (11-bit fixed point)
        int xi, xa0, xa1, yi, yb0, yb1;
        vd_area_lin_coef(m.W, m.in_w, ix, &xi, &xa0, &xa1);
        vd_area_lin_coef(m.H, m.in_h, iy, &yi, &yb0, &yb1);
        const int x1 = xi + 1 < m.W ? xi + 1 : m.W - 1, y1 = yi + 1 < m.H ? yi + 1 : m.H - 1;
        const int r0 = (int)sharp_at<PRE>(g, m.H, m.W, m.pre, yi, xi, c, kn, kc) * xa0 + (int)sharp_at<PRE>(g, m.H, m.W, m.pre, yi, x1, c, kn, kc) * xa1;
        const int r1 = (int)sharp_at<PRE>(g, m.H, m.W, m.pre, y1, xi, c, kn, kc) * xa0 + (int)sharp_at<PRE>(g, m.H, m.W, m.pre, y1, x1, c, kn, kc) * xa1;
        const int q = (((yb0 * (r0 >> 4)) >> 16) + ((yb1 * (r1 >> 4)) >> 16) + 2) >> 2;
        v = (uint8_t)(q < 0 ? 0 : (q > 255 ? 255 : q));
      } else if (inside && m.frac) {  // ResizeArea_<uchar,float>: per source row sum_k S*alpha (float32), then sum_j row*beta
        float ax[VD_AREA_MAXT], ay[VD_AREA_MAXT];
        int x0s, y0s;
        const int nx = vd_area_taps(m.W, m.sx, ix, &x0s, ax), ny = vd_area_taps(m.H, m.sy, iy, &y0s, ay);
        float acc = 0.f;
        for (int j = 0; j < ny; ++j) {
          float h = 0.f;
          for (int k = 0; k < nx; ++k) h = h + (float)sharp_at<PRE>(g, m.H, m.W, m.pre, y0s + j, x0s + k, c, kn, kc) * ax[k];
          acc = acc + h * ay[j];
        }
        v = vd_sat_rne_u8(acc);
      } else if (inside) {
        if (m.fx == 1 && m.fy == 1) v = sharp_at<PRE>(g, m.H, m.W, m.pre, iy, ix, c, kn, kc);
        else {
          int sum = 0;
          for (int j = 0; j < m.fy; ++j)
            for (int i = 0; i < m.fx; ++i) sum += sharp_at<PRE>(g, m.H, m.W, m.pre, iy * m.fy + j, ix * m.fx + i, c, kn, kc);
          if (m.fx == 2 && m.fy == 2) v = (uint8_t)((sum + 2) >> 2);
          else v = vd_sat_rne_u8((float)sum * (1.f / (float)(m.fx * m.fy)));
        }
      }
      px[eye][c] = v;
    }
  }
  if (m.format == VD3D_FMT_HALF_SBS || m.format == VD3D_FMT_FULL_SBS || m.format == VD3D_FMT_VR) {
    uint8_t* o0 = out + ((size_t)y * m.out_w + x) * 3;
    uint8_t* o1 = out + ((size_t)y * m.out_w + x + m.fit_w) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) { o0[c] = px[0][c]; o1[c] = px[1][c]; }
  } else if (m.format == VD3D_FMT_INTERLACED) {
    uint8_t* o0 = out + ((size_t)y * m.out_w + x) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) o0[c] = px[y & 1][c];
  } else {  // Dubois anaglyph on the BGR-ordered planes as-is (:866-883)
    const float l0 = vd_u8_unit((float)px[0][0]), l1 = vd_u8_unit((float)px[0][1]), l2 = vd_u8_unit((float)px[0][2]);
    const float r0 = vd_u8_unit((float)px[1][0]), r1 = vd_u8_unit((float)px[1][1]), r2 = vd_u8_unit((float)px[1][2]);
    const float red = ((float)0.4561 * l0 + (float)0.5005 * l1) + (float)0.1762 * l2;
    const float green = ((float)0.3764 * r0 + (float)0.7616 * r1) - (float)0.1876 * r2;
    const float blue = ((float)-0.0401 * r0 - (float)0.1126 * r1) + (float)1.2723 * r2;
    uint8_t* o0 = out + ((size_t)y * m.out_w + x) * 3;
    o0[0] = (uint8_t)(vd_clamp(red, 0.f, 1.f) * 255.0f);
    o0[1] = (uint8_t)(vd_clamp(green, 0.f, 1.f) * 255.0f);
    o0[2] = (uint8_t)(vd_clamp(blue, 0.f, 1.f) * 255.0f);
  }
}
void vd_launch_sharp_mux(hipStream_t s, const uint8_t* gL, const uint8_t* gR, const vd3d_render_params& p,
                         const vd_finish_consts& fc, uint8_t* out, int presharp_pitch, bool identity_fit) {
  vd_mux_geom m;
  m.pre = presharp_pitch;
  m.H = p.warp_h; m.W = p.warp_w; m.fit_w = p.fit_w; m.fit_h = p.fit_h;
  m.out_w = p.out_w; m.out_h = p.out_h; m.format = p.format;
  if (p.format == VD3D_FMT_HALF_SBS || (identity_fit && p.warp_w == p.fit_w && p.warp_h == p.fit_h)) {  // cv2.resize straight to (per_eye_w, per_eye_h) :1413
    m.in_w = p.fit_w; m.in_h = p.fit_h; m.xo = 0; m.yo = 0;
  } else {  // pad_to_aspect_ratio :101-131
    const double ta = (double)p.fit_w / p.fit_h, ca = (double)p.warp_w / p.warp_h;
    if (ca > ta) { m.in_w = p.fit_w; m.in_h = (int)(p.fit_w / ca); }
    else { m.in_h = p.fit_h; m.in_w = (int)(ca * p.fit_h); }
    m.xo = (p.fit_w - m.in_w) / 2; m.yo = (p.fit_h - m.in_h) / 2;
  }
  m.fx = m.in_w > 0 ? p.warp_w / m.in_w : 1; m.fy = m.in_h > 0 ? p.warp_h / m.in_h : 1;
  m.sx = 1.0 / ((double)m.in_w / p.warp_w); m.sy = 1.0 / ((double)m.in_h / p.warp_h);
  m.frac = (p.warp_w % m.in_w || p.warp_h % m.in_h) ? 1 : 0;
  if (m.in_w > p.warp_w || m.in_h > p.warp_h) m.frac = 2;
  const dim3 g((p.fit_w + 63) / 64, (p.fit_h + 3) / 4);
  if (presharp_pitch > 0) hipLaunchKernelGGL(k_sharp_mux<true>, g, dim3(256), 0, s, gL, gR, m, fc.sharp_kn, fc.sharp_kc, out);
  else hipLaunchKernelGGL(k_sharp_mux<false>, g, dim3(256), 0, s, gL, gR, m, fc.sharp_kn, fc.sharp_kc, out);
}

// blank frame (skip_blank_frames, core/render_3d.py:1278-1281 + :1398-1403): both eyes are the raw source frame with the floating-window
// side mask applied; the bar comes from the device-resident frame constants (no host round trip)
__global__ __launch_bounds__(256) void k_blank_eye(const uint8_t* __restrict__ src, int h, int w, const vd_dev_work* __restrict__ wk,
                                                   uint8_t* __restrict__ dst) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w || y >= h) return;
  const int bw = min(wk->bar_width, w), bs = wk->bar_side;
  const bool masked = bw > 0 && ((bs == 2 && x < bw) || (bs == 1 && x >= w - bw));
  const size_t i = ((size_t)y * w + x) * 3;
  dst[i] = masked ? 0 : src[i]; dst[i + 1] = masked ? 0 : src[i + 1]; dst[i + 2] = masked ? 0 : src[i + 2];
}
void vd_launch_blank_eye(hipStream_t s, const uint8_t* src, int h, int w, const vd_dev_work* wk, uint8_t* dst) {
  hipLaunchKernelGGL(k_blank_eye, dim3((w + 63) / 64, (h + 3) / 4), dim3(256), 0, s, src, h, w, wk, dst);
}

// ------------------------------------------------------------------------------------------------
// streaming copy: the measured-peak yardstick for roofline.frac.  ONE 16-byte element per thread, no loop: of the access patterns tried
// on MI355X (tools/ubench_valu.hip, 1 GiB: grid-stride with four loads in flight 4.4-5.2 TB/s, contiguous chunk per workgroup 5.6-5.9,
// hipMemcpyAsync 5.5) this is the one that reaches the guide's 6.2-6.3 TB/s.
// ------------------------------------------------------------------------------------------------
// elementwise torch-CPU math exactly as the chain's kernels evaluate it (vd3d_torch_math; test / diagnostic entry)
__global__ __launch_bounds__(256) void k_torch_math(int op, const float* __restrict__ x, float p, float* __restrict__ out, long long n) {
  __shared__ int2 rs14[64];
  vd_stage_rs14(rs14, threadIdx.x, 256);
  __syncthreads();
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float v = x[i];
    out[i] = op == 0 ? vd_pow_torch(v, p, rs14) : op == 1 ? vd_sigmoid_torch(v) : vd_sqrt_torch(v, rs14);
  }
}
// F.interpolate(bilinear, align_corners=False) of C planes into HBM (round 5).  The render kernels resize inside themselves; this one exists for the N-thread
// ATen mode, where small outputs (and 3-channel inputs of a one-thread torch) take ATen's premultiplied-weight kernel: the plane is resized here and the consumer
// is called with identity geometry.
__global__ __launch_bounds__(256) void k_interp_planes(const float* __restrict__ src, int ih, int iw, float* __restrict__ dst, int oh, int ow, int pm) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= ow || y >= oh) return;
  const float* pl = src + (size_t)blockIdx.z * ih * iw;
  const vd_tap ty = vd_interp_tap(ih, oh, y), tx = vd_interp_tap(iw, ow, x);
  const float* r0 = pl + (size_t)ty.i0 * iw;
  const float* r1 = pl + (size_t)ty.i1 * iw;
  dst[((size_t)blockIdx.z * oh + y) * ow + x] = vd_bilerp_sel(pm != 0, r0[tx.i0], r0[tx.i1], r1[tx.i0], r1[tx.i1], tx.w0, tx.w1, ty.w0, ty.w1);
}
void vd_launch_interp_planes(hipStream_t s, const float* src, int C, int ih, int iw, float* dst, int oh, int ow, int premult) {
  hipLaunchKernelGGL(k_interp_planes, dim3((ow + 63) / 64, (oh + 3) / 4, C), dim3(256), 0, s, src, ih, iw, dst, oh, ow, premult);
}

// the same with ATen's scalar tails for `threads` intra-op threads (threads < 0: every element takes the tail arithmetic -- diagnostic)
__global__ __launch_bounds__(256) void k_torch_math_aten(int op, const float* __restrict__ x, double p, float* __restrict__ out, long long n, vd_tails tl) {
  __shared__ int2 rs14[64];
  vd_stage_rs14(rs14, threadIdx.x, 256);
  __syncthreads();
  const float pf = (float)p;
  const bool special = vd_pow_is_special(pf);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float v = x[i];
    const bool tail = tl.on && vd_in_tail(tl, (unsigned)i);
    out[i] = op == 0 ? (tail && !special ? vd_pow_tail(v, p) : vd_pow_torch(v, pf, rs14)) : (tail ? vd_sigmoid_tail(v) : vd_sigmoid_torch(v));
  }
}
void vd_launch_torch_math_aten(hipStream_t s, int op, const float* x, double p, float* out, long long n, int threads) {
  const long long nb = (n + 255) / 256;
  const vd_tails tl = threads < 0 ? vd_tails_all((unsigned long long)n) : vd_tails_of((unsigned long long)n, threads);
  hipLaunchKernelGGL(k_torch_math_aten, dim3((unsigned)(nb < 16384 ? (nb > 0 ? nb : 1) : 16384)), dim3(256), 0, s, op, x, p, out, n, tl);
}
void vd_launch_torch_math(hipStream_t s, int op, const float* x, float p, float* out, long long n) {
  const long long nb = (n + 255) / 256;
  hipLaunchKernelGGL(k_torch_math, dim3((unsigned)(nb < 16384 ? (nb > 0 ? nb : 1) : 16384)), dim3(256), 0, s, op, x, p, out, n);
}

typedef unsigned int vd_u4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_stream_copy(const vd_u4* __restrict__ src, vd_u4* __restrict__ dst, size_t n16) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n16) dst[i] = src[i];
}
void vd_launch_stream_copy(hipStream_t s, const void* src, void* dst, size_t bytes) {
  const size_t n16 = bytes / 16;
  const size_t per = (size_t)1 << 31;   // elements per launch (the grid dimension is 32-bit)
  for (size_t off = 0; off < n16; off += per) {
    const size_t n = n16 - off < per ? n16 - off : per;
    hipLaunchKernelGGL(k_stream_copy, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const vd_u4*)src + off, (vd_u4*)dst + off, n);
  }
  if (bytes % 16) (void)hipMemcpyAsync((char*)dst + n16 * 16, (const char*)src + n16 * 16, bytes % 16, hipMemcpyDeviceToDevice, s);
}

// ------------------------------------------------------------------------------------------------
// preview visualisers (core/preview_utils.py:23-84): one thread per output pixel
// ------------------------------------------------------------------------------------------------
// hal::resize INTER_LINEAR coefficients for 8-bit (11-bit fixed point), one destination index
VD_DEV void vd_lin_coef(int ssize, int dsize, int d, int* idx, int* a0, int* a1) {
  const double scale = 1.0 / ((double)dsize / ssize);
  float fx = (float)((d + 0.5) * scale - 0.5);
  int sx = (int)floorf(fx);
  fx -= (float)sx;
  if (sx < 0) { fx = 0.f; sx = 0; }
  if (sx >= ssize - 1) { fx = 0.f; sx = ssize - 1; }
  *idx = sx;
  *a0 = (int)rintf((1.f - fx) * 2048.f);
  *a1 = (int)rintf(fx * 2048.f);
}
__global__ __launch_bounds__(256) void k_preview(int type, const uint8_t* __restrict__ L, const uint8_t* __restrict__ R, int h, int w,
                                                 uint8_t* __restrict__ out) {
  const int ow = type == VD3D_PREVIEW_HSBS ? 2 * (w / 2) : w;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= ow || y >= h) return;
  uint8_t* o = out + ((size_t)y * ow + x) * 3;
  const size_t i = ((size_t)y * w + x) * 3;
  switch (type) {
    case VD3D_PREVIEW_INTERLACED: { const uint8_t* s = (y & 1) ? R : L; o[0] = s[i]; o[1] = s[i + 1]; o[2] = s[i + 2]; } break;
    case VD3D_PREVIEW_LR_DIFF:
#pragma unroll
      for (int c = 0; c < 3; ++c) { const int d = (int)L[i + c] - (int)R[i + c]; o[c] = (uint8_t)(d < 0 ? -d : d); }
      break;
    case VD3D_PREVIEW_FEATHER_BLEND: o[0] = L[i]; o[1] = L[i + 1]; o[2] = L[i + 2]; break;
    case VD3D_PREVIEW_RED_BLUE: o[0] = R[i]; o[1] = R[i + 1]; o[2] = L[i + 2]; break;
    case VD3D_PREVIEW_HSBS: {
      const int hw = w / 2;
      const uint8_t* s = x < hw ? L : R;
      int xi, a0, a1;
      vd_lin_coef(w, hw, x < hw ? x : x - hw, &xi, &a0, &a1);
      const int x1 = xi + 1 < w ? xi + 1 : w - 1;
#pragma unroll
      for (int c = 0; c < 3; ++c) {   // height unchanged: beta = (2048, 0)
        const int r0 = (int)s[((size_t)y * w + xi) * 3 + c] * a0 + (int)s[((size_t)y * w + x1) * 3 + c] * a1;
        const int q = (((2048 * (r0 >> 4)) >> 16) + 2) >> 2;
        o[c] = (uint8_t)(q < 0 ? 0 : (q > 255 ? 255 : q));
      }
    } break;
  }
}
bool vd_launch_preview(hipStream_t s, int type, const uint8_t* L, const uint8_t* R, int h, int w, uint8_t* out) {
  if (type < VD3D_PREVIEW_INTERLACED || type > VD3D_PREVIEW_RED_BLUE) return false;
  const int ow = type == VD3D_PREVIEW_HSBS ? 2 * (w / 2) : w;
  if (ow < 1) return false;
  hipLaunchKernelGGL(k_preview, dim3((ow + 63) / 64, (h + 3) / 4), dim3(256), 0, s, type, L, R, h, w, out);
  return true;
}
