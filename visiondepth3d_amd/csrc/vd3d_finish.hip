// vd3d_finish.hip -- E1, the fused finishing kernel: apply_dof_cuda + apply_color_grade + tensor_to_frame +
// apply_side_mask + apply_sharpening + INTER_AREA fit + SBS / interlaced mux in ONE launch for both eyes
// (core/render_3d.py:1340-1419).  Replaces k_dof_grade x2 + k_sharp_mux and their two graded planes (-12N B of HBM).
//
// Per 64x16 tile of sharpened pixels (384 threads, blockIdx.z = eye):
//   load    (16+2+8) x (64+8+8) reflect-padded u8 tile -> /255 -> float planes in LDS (row pitch 80 floats, 16 B aligned)
//   level l H-pass: one task = (channel,row,4-pixel strip): 3x ds_read_b128 window, k-tap sums in the reference order,
//           one ds_write_b128;   V-pass: one thread = one strip of the 18x72 graded region, k x ds_read_b128
//   blend the two levels each pixel needs, grade, truncate, side bars -> packed BGR0 dwords in LDS
//   epilogue: 3x3 sharpen (5 dword reads give all 3 channels), integer-ratio box average, 12-byte packed stores
// Arithmetic identical to k_dof_grade / k_sharp_mux (and the oracle): same association, no contraction.
// Fast path conditions (else the unfused kernels run): Gaussian taps <= 9 (dof_strength <= 2), fit factors in {1,2,4},
// format in {Half-SBS, Full-SBS, Passive Interlaced}.
#include "vd3d_dev.h"
#include "vd3d_kernels.h"

#define FF_TW 64
#define FF_TH 16
#define FF_R 4                      // max Gaussian radius of the fast path
#define FF_GW (FF_TW + 8)           // graded region width  (4-pixel halo each side: aligned strips; 1 is needed)
#define FF_GH (FF_TH + 2)           // graded region height (1-pixel halo)
#define FF_IW (FF_GW + 2 * FF_R)    // input tile width  = 80
#define FF_IH (FF_GH + 2 * FF_R)    // input tile height = 26
#define FF_NS (FF_GW / 4)           // strips per row = 18
#define FF_NT 384

struct vd_ff_args {
  int H, W, eh, ew;
  int fit_w, fit_h, in_w, in_h, xo, yo, fx, fy, out_w, format;
  int use_override, bar_w, bar_s;
  float focal;
};

// one Gaussian level: K = 9 - 2*OFF taps.  H-pass over (channel,row,strip) tasks, then V-pass for the strips that need it.
template <int OFF>
VD_DEV void ff_level(float (*tile)[FF_IH][FF_IW], float (*hb)[FF_IH][FF_GW], const float* __restrict__ kern, int tid, bool need,
                     int sy, int ss, int level, const int lo[4], float vlo[3][4], float vhi[3][4]) {
  constexpr int K = 2 * (FF_R - OFF) + 1;
  float kw[K];
#pragma unroll
  for (int t = 0; t < K; ++t) kw[t] = kern[t];
  for (int t = tid; t < 3 * FF_IH * FF_NS; t += FF_NT) {
    const int c = t / (FF_IH * FF_NS), rem = t - c * FF_IH * FF_NS, row = rem / FF_NS, s = rem - row * FF_NS;
    const float4* wp = reinterpret_cast<const float4*>(&tile[c][row][4 * s]);
    const float4 w0 = wp[0], w1 = wp[1], w2 = wp[2];
    const float win[12] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float sacc = 0.f;
#pragma unroll
      for (int tt = 0; tt < K; ++tt) sacc += kw[tt] * win[j + OFF + tt];
      o[j] = sacc;
    }
    *reinterpret_cast<float4*>(&hb[c][row][4 * s]) = make_float4(o[0], o[1], o[2], o[3]);
  }
  __syncthreads();
  if (need) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tt = 0; tt < K; ++tt) {
        const float4 v = *reinterpret_cast<const float4*>(&hb[c][sy + OFF + tt][4 * ss]);
        o[0] += kw[tt] * v.x; o[1] += kw[tt] * v.y; o[2] += kw[tt] * v.z; o[3] += kw[tt] * v.w;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (level == lo[q]) vlo[c][q] = o[q];
        if (level == lo[q] + 1) vhi[c][q] = o[q];
      }
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(FF_NT) void k_finish_fused(const uint8_t* __restrict__ eyeL, const uint8_t* __restrict__ eyeR,
                                                        const float* __restrict__ dn, vd_finish_consts fc, vd_ff_args a,
                                                        const vd_dev_work* __restrict__ w, uint8_t* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float tile[3][FF_IH][FF_IW];
  __shared__ __attribute__((aligned(16))) float hb[3][FF_IH][FF_GW];
  __shared__ uint32_t gb[FF_GH][FF_GW + 1];   // odd pitch: the epilogue's column walks stay conflict-free
  __shared__ float lut[256];                  // v / 255.0f (true division), computed once per workgroup
  __shared__ int lvl_mask;                    // levels any pixel of this tile needs
  const int eye = blockIdx.z;
  const uint8_t* __restrict__ src = eye == 0 ? eyeL : eyeR;
  const int H = a.H, W = a.W;
  const int x0 = blockIdx.x * FF_TW, y0 = blockIdx.y * FF_TH;
  const int gx0 = x0 - 4, gy0 = y0 - 1;            // graded region origin
  const int ix0 = gx0 - FF_R, iy0 = gy0 - FF_R;    // input tile origin
  const int tid = threadIdx.x;

  if (tid < 256) lut[tid] = (float)tid / 255.0f;
  if (tid == 0) lvl_mask = 0;
  __syncthreads();
  for (int t = tid; t < FF_IH * FF_IW; t += FF_NT) {
    const int ty = t / FF_IW, tx = t - ty * FF_IW;
    const int y = vd_reflect(iy0 + ty, H), x = vd_reflect(ix0 + tx, W);
    const uint8_t* px = src + ((size_t)y * W + x) * 3;
    tile[0][ty][tx] = lut[px[2]];
    tile[1][ty][tx] = lut[px[1]];
    tile[2][ty][tx] = lut[px[0]];
  }
  // per-strip setup (threads 0..323 own one 4-pixel strip of the graded region)
  const bool strip = tid < FF_GH * FF_NS;
  const int sy = tid / FF_NS, ss = tid - sy * FF_NS;   // graded row, strip index
  const int gy = gy0 + sy, gxs = gx0 + 4 * ss;
  int lo[4] = {0, 0, 0, 0};
  float alpha[4] = {0.f, 0.f, 0.f, 0.f};
  int lmin = 9, lmax = -1;
  if (strip && fc.nlev) {
    const float focal = a.use_override ? a.focal : w->focal;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int y = min(max(gy, 0), H - 1), x = min(max(gxs + q, 0), W - 1);  // halo pixels outside the image are never used
      float dd;
      if (a.eh == H && a.ew == W) dd = dn[(size_t)y * W + x];
      else {
        const vd_tap ay = vd_interp_tap(a.eh, H, y), ax = vd_interp_tap(a.ew, W, x);
        const float* r0 = dn + (size_t)ay.i0 * a.ew;
        const float* r1 = dn + (size_t)ay.i1 * a.ew;
        dd = vd_bilerp(r0[ax.i0], r0[ax.i1], r1[ax.i0], r1[ax.i1], ax.w0, ax.w1, ay.w0, ay.w1);
      }
      const float bw = vd_clamp(fabsf(dd - focal) / fc.fw, 0.f, 1.f);
      const float bi = vd_clamp(bw * (float)fc.nlev, 0.f, fc.imax);
      int l = (int)floorf(bi);
      l = l > fc.nlev - 1 ? fc.nlev - 1 : (l < 0 ? 0 : l);
      lo[q] = l; alpha[q] = bi - (float)l;
      lmin = min(lmin, l); lmax = max(lmax, l + 1);
    }
    int m = 0;
    for (int l = max(lmin, 1); l <= lmax; ++l) m |= 1 << l;
    if (m) atomicOr(&lvl_mask, m);
  }
  __syncthreads();
  const int need_mask = lvl_mask;
  float vlo[3][4], vhi[3][4];
  if (strip) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float4 v = *reinterpret_cast<const float4*>(&tile[c][sy + FF_R][4 * ss + FF_R]);
      vlo[c][0] = v.x; vlo[c][1] = v.y; vlo[c][2] = v.z; vlo[c][3] = v.w;
#pragma unroll
      for (int q = 0; q < 4; ++q) vhi[c][q] = vlo[c][q];
    }
  }
  for (int l = 0; l < fc.nlev; ++l) {  // level l+1 of the reference's stack
    if (!(need_mask >> (l + 1) & 1)) continue;  // no pixel of this tile blends with this level (workgroup-uniform)
    const int off = FF_R - fc.ksz[l] / 2;
    const bool need = strip && l + 1 >= lmin && l + 1 <= lmax;
    switch (off) {  // compile-time tap count => all register indexing is static
      case 0: ff_level<0>(tile, hb, fc.kern[l], tid, need, sy, ss, l + 1, lo, vlo, vhi); break;
      case 1: ff_level<1>(tile, hb, fc.kern[l], tid, need, sy, ss, l + 1, lo, vlo, vhi); break;
      case 2: ff_level<2>(tile, hb, fc.kern[l], tid, need, sy, ss, l + 1, lo, vlo, vhi); break;
      default: ff_level<3>(tile, hb, fc.kern[l], tid, need, sy, ss, l + 1, lo, vlo, vhi); break;
    }
  }
  if (strip) {  // blend, grade (:750-767), truncate, side bars (:885-892)
    const int bar_w = a.use_override ? a.bar_w : w->bar_width, bar_s = a.use_override ? a.bar_s : w->bar_side;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float rgbv[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float v = vlo[c][q];
        if (fc.nlev) v = vd_clamp((1.0f - alpha[q]) * vlo[c][q] + alpha[q] * vhi[c][q], 0.f, 1.f);
        rgbv[c] = v;
      }
      const float luma = ((float)0.2126 * rgbv[0] + (float)0.7152 * rgbv[1]) + (float)0.0722 * rgbv[2];
      const int x = gxs + q;
      const bool masked = bar_w > 0 && ((bar_s == 2 && x < bar_w) || (bar_s == 1 && x >= W - bar_w));
      uint32_t pk = 0;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float v = luma + (rgbv[c] - luma) * fc.sat;
        v = 0.5f + (v - 0.5f) * fc.con;
        v = v + fc.bri;
        v = vd_clamp(v, 0.f, 1.f);
        const uint32_t u = masked ? 0u : (uint32_t)(uint8_t)(v * 255.0f);
        pk |= u << (8 * (2 - c));  // byte 0 = B, 1 = G, 2 = R
      }
      gb[sy][4 * ss + q] = pk;
    }
  }
  __syncthreads();
  // epilogue: sharpen (:717-732) + integer-ratio INTER_AREA (:1413) + mux; one task = 4 consecutive output pixels
  const int ow = FF_TW / a.fx, oh = FF_TH / a.fy;      // output pixels produced by this tile
  const int ox0 = x0 / a.fx, oy0 = y0 / a.fy;
  const float kn = fc.sharp_kn, kc = fc.sharp_kc;
  const float scale = 1.f / (float)(a.fx * a.fy);
  for (int t = tid; t < oh * (ow / 4); t += FF_NT) {
    const int tq = t / oh, ty = t - tq * oh;   // row-fastest: lanes walk a column of the odd-pitch gb tile
    const int oy = oy0 + ty;
    if (oy >= a.in_h) continue;
    uint32_t pack[3] = {0, 0, 0};
    int nvalid = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ox = ox0 + tq * 4 + q;
      if (ox >= a.in_w) break;
      ++nvalid;
      int sum[3] = {0, 0, 0};
      for (int j = 0; j < a.fy; ++j)
        for (int i = 0; i < a.fx; ++i) {
          const int y = oy * a.fy + j, x = ox * a.fx + i;     // sharpened pixel (inside this tile)
          const int yu = vd_reflect(y - 1, H), yd = vd_reflect(y + 1, H), xl = vd_reflect(x - 1, W), xr = vd_reflect(x + 1, W);
          const uint32_t pu = gb[yu - gy0][x - gx0], pl = gb[y - gy0][xl - gx0], pc = gb[y - gy0][x - gx0];
          const uint32_t pr = gb[y - gy0][xr - gx0], pd = gb[yd - gy0][x - gx0];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const int sh = 8 * c;
            float sacc = 0.f;
            sacc += kn * (float)((pu >> sh) & 0xffu);
            sacc += kn * (float)((pl >> sh) & 0xffu);
            sacc += kc * (float)((pc >> sh) & 0xffu);
            sacc += kn * (float)((pr >> sh) & 0xffu);
            sacc += kn * (float)((pd >> sh) & 0xffu);
            sum[c] += (int)vd_sat_rne_u8(sacc);
          }
        }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        uint8_t v;
        if (a.fx == 1 && a.fy == 1) v = (uint8_t)sum[c];
        else if (a.fx == 2 && a.fy == 2) v = (uint8_t)((sum[c] + 2) >> 2);
        else v = vd_sat_rne_u8((float)sum[c] * scale);
        const int bi = 3 * q + c;
        pack[bi >> 2] |= (uint32_t)v << (8 * (bi & 3));
      }
    }
    if (a.format == VD3D_FMT_INTERLACED && (((oy + a.yo) & 1) != eye)) continue;
    const int oxq = ox0 + tq * 4 + a.xo + ((a.format == VD3D_FMT_INTERLACED) ? 0 : eye * a.fit_w);
    uint8_t* o = out + ((size_t)(oy + a.yo) * a.out_w + oxq) * 3;
    if (nvalid == 4 && ((size_t)(o - out) & 3) == 0) {
      uint32_t* o32 = reinterpret_cast<uint32_t*>(o);
      o32[0] = pack[0]; o32[1] = pack[1]; o32[2] = pack[2];
    } else {
      for (int bi = 0; bi < 3 * nvalid; ++bi) o[bi] = (uint8_t)(pack[bi >> 2] >> (8 * (bi & 3)));
    }
  }
}

// returns false when the fast path does not apply (caller runs the unfused kernels)
bool vd_launch_finish_fused(hipStream_t s, const uint8_t* L, const uint8_t* R, const float* dn, int eh, int ew,
                            const vd3d_render_params& p, const vd_finish_consts& fc, const vd_dev_work* w, float focal,
                            int use_override, int bar_w, int bar_s, uint8_t* out) {
  if (!(p.format == VD3D_FMT_HALF_SBS || p.format == VD3D_FMT_FULL_SBS || p.format == VD3D_FMT_INTERLACED)) return false;
  for (int l = 0; l < fc.nlev; ++l) if (fc.ksz[l] > 2 * FF_R + 1 || fc.ksz[l] < 3) return false;
  vd_ff_args a;
  a.H = p.warp_h; a.W = p.warp_w; a.eh = eh; a.ew = ew;
  a.fit_w = p.fit_w; a.fit_h = p.fit_h; a.out_w = p.out_w; a.format = p.format;
  if (p.format == VD3D_FMT_HALF_SBS) { a.in_w = p.fit_w; a.in_h = p.fit_h; a.xo = 0; a.yo = 0; }
  else {
    const double ta = (double)p.fit_w / p.fit_h, ca = (double)p.warp_w / p.warp_h;
    if (ca > ta) { a.in_w = p.fit_w; a.in_h = (int)(p.fit_w / ca); }
    else { a.in_h = p.fit_h; a.in_w = (int)(ca * p.fit_h); }
    a.xo = (p.fit_w - a.in_w) / 2; a.yo = (p.fit_h - a.in_h) / 2;
  }
  a.fx = p.warp_w / a.in_w; a.fy = p.warp_h / a.in_h;
  if (!((a.fx == 1 || a.fx == 2 || a.fx == 4) && (a.fy == 1 || a.fy == 2 || a.fy == 4))) return false;
  if ((FF_TW / a.fx) % 4) return false;
  a.use_override = use_override; a.bar_w = bar_w; a.bar_s = bar_s; a.focal = focal;
  if (a.xo || a.yo || a.in_w != p.fit_w || a.in_h != p.fit_h)  // pad_to_aspect_ratio canvas (:124): black background
    (void)hipMemsetAsync(out, 0, (size_t)p.out_w * p.out_h * 3, s);
  dim3 g((p.warp_w + FF_TW - 1) / FF_TW, (p.warp_h + FF_TH - 1) / FF_TH, 2);
  hipLaunchKernelGGL(k_finish_fused, g, dim3(FF_NT), 0, s, L, R, dn, fc, a, w, out);
  return true;
}
