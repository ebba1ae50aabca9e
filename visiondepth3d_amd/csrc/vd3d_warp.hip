// vd3d_warp.hip -- W1, the fused parallax-warp kernel (the kernel BASELINE.json:north_star sets the HBM target on).
//
// One launch does what the reference does with 4 grid_sample calls + 2 feather_shift_edges
// (core/render_3d.py:684-712): warped-depth gradient mask -> k x k separable window average -> RGB warp of both
// eyes -> feather blend -> tensor_to_frame truncation, with NO intermediate planes in HBM.
//
// Per 64x32 output tile (512 threads, 2 workgroups per CU):
//   phase A warped depth of both eyes on the (TH+k) x (TW+k) halo: S loads then D gathers (L2), fixed unroll for ILP
//   (the eye-res RGB tile is requested from HBM before phase A into registers and lands in LDS afterwards)
//   phase B e2 = clamp(|grad WD| * fs, 0, 1)            phase C horizontal k-sums (ascending x)
//   phase D vertical k-sums -> b, RGB samples = nested bilinear (resize of :595 inside the grid_sample of :697)
//           read from the LDS eye tile, one row per wave (row-uniform taps; exact skip of the south samples when the
//           sample row is integral), blend, truncate, shuffle-packed 12-byte stores per 4 lanes.
// Arithmetic is identical to the unfused v0 kernels (same helpers, same association) => bit-exact vs the oracle.
//
// Algorithmic HBM bytes per stereo pair: read RGB 3N (eye-res f32 x3 at N/4) + D 4N + S 4N, write 6N.
#include "vd3d_dev.h"
#include "vd3d_kernels.h"

#define WF_TW 64
#define WF_TH 32
#define WF_NT 512
#define WF_PF 16  // RGB tile elements prefetched per thread (registers) while phase A runs

struct vd_wf_args {
  int ih, iw, H, W, k, feather, bound;  // bound: rigorous host-side bound on |pixel shift| (+ margin)
  int er_max, ec_max;                   // eye tile capacity (rows, cols) when resizing
  uint32_t m_ww, m_ew, m_ec, m_erec;    // ceil(2^32/d) reciprocals: q = umulhi(t, m) is exact for t, d < 2^16
  float fs, scale_h, scale_w;
};
VD_DEV int wf_div(int t, uint32_t m) { return (int)__umulhi((uint32_t)t, m); }

VD_DEV vd_tap wf_tap(int in, int out, float scale, int o) {  // vd_interp_tap with the scale hoisted
  vd_tap t;
  if (in == out) { t.i0 = o; t.i1 = o; t.w0 = 1.f; t.w1 = 0.f; return t; }
  float src = scale * ((float)o + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  int i0 = (int)floorf(src);
  if (i0 > in - 1) i0 = in - 1;
  float l1 = vd_clamp(src - (float)i0, 0.f, 1.f);
  t.i0 = i0; t.i1 = i0 + (i0 < in - 1 ? 1 : 0); t.w1 = l1; t.w0 = 1.f - l1;
  return t;
}

// LDS map (floats):  wd[2][wh*ww] (later hs[2][eh*TW]) | e2[2][eh*ew] | tile[3*er*ec]
#define WF_AI 6  // phase-A positions per thread ((TH+k)(TW+k) <= WF_AI*WF_NT for k <= 9; larger k loops)
template <bool RESIZE>
__global__ __launch_bounds__(WF_NT) void k_warp_fused(const float* __restrict__ rgb, const float* __restrict__ D,
                                                      const float* __restrict__ S, vd_wf_args a, uint8_t* __restrict__ L,
                                                      uint8_t* __restrict__ R) {
  extern __shared__ float lds[];
  const int H = a.H, W = a.W, k = a.k, r = k / 2;
  const int x0 = blockIdx.x * WF_TW, y0 = blockIdx.y * WF_TH;
  const int ww = WF_TW + k, wh = WF_TH + k;          // wd region
  const int ew = WF_TW + k - 1, eh = WF_TH + k - 1;  // e2 region
  float* wd = lds;                                   // [2][wh*ww]   (later hs [2][eh*WF_TW])
  float* e2 = lds + 2 * wh * ww;                     // [2][eh*ew]
  float* tile = a.feather ? e2 + 2 * eh * ew : lds;  // [3][er][ec] (the feather buffers are not allocated when feathering is off)
  const int tid = threadIdx.x;
  const int wy0 = y0 - r - 1, wx0 = x0 - r - 1;

  // eye-res RGB tile: global -> registers now, registers -> LDS after phase A (latency hidden behind phase A)
  int er0 = 0, ec0 = 0, er = 0, ec = 0;
  float pf[WF_PF];
  if (RESIZE) {
    const int ya = max(y0 - 1, 0), yb = min(y0 + WF_TH, H - 1);
    const int xa = max(x0 - a.bound - 1, 0), xb = min(x0 + WF_TW + a.bound + 1, W - 1);
    er0 = wf_tap(a.ih, H, a.scale_h, ya).i0; er = wf_tap(a.ih, H, a.scale_h, yb).i1 - er0 + 1;
    ec0 = wf_tap(a.iw, W, a.scale_w, xa).i0; ec = wf_tap(a.iw, W, a.scale_w, xb).i1 - ec0 + 1;
    er = min(er, a.er_max); ec = a.ec_max;  // fixed row pitch (host constant) so the reciprocals apply
    ec0 = min(ec0, a.iw - ec); ec0 = max(ec0, 0);
    const size_t ni = (size_t)a.ih * a.iw;
#pragma unroll
    for (int j = 0; j < WF_PF; ++j) {
      const int t = tid + j * WF_NT;
      float v = 0.f;
      if (t < 3 * er * ec) {
        const int c = t >= 2 * er * ec ? 2 : (t >= er * ec ? 1 : 0);
        const int rem = t - c * er * ec, ty = wf_div(rem, a.m_ec), tx = rem - ty * ec;
        if (ec0 + tx < a.iw) v = rgb[c * ni + (size_t)(er0 + ty) * a.iw + (ec0 + tx)];
      }
      pf[j] = v;
    }
  }
  if (a.feather) {
    // phase A: warped depth of both eyes on the (TH+k) x (TW+k) halo region (grid_sample of D, :700-701).
    // Two passes with a fixed unroll so all S loads, then all D gathers, are in flight together.
    for (int base = 0; base < wh * ww; base += WF_AI * WF_NT) {
      float sv[WF_AI];
#pragma unroll
      for (int j = 0; j < WF_AI; ++j) {
        const int t = base + tid + j * WF_NT;
        const int ty = wf_div(t, a.m_ww), tx = t - ty * ww;
        const int y = wy0 + ty, x = wx0 + tx;
        sv[j] = (t < wh * ww && y >= 0 && y < H && x >= 0 && x < W) ? S[(size_t)y * W + x] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < WF_AI; ++j) {
        const int t = base + tid + j * WF_NT;
        if (t < wh * ww) {
          const int ty = wf_div(t, a.m_ww), tx = t - ty * ww;
          const int y = wy0 + ty, x = wx0 + tx;
          float vl = 0.f, vr = 0.f;
          if (y >= 0 && y < H && x >= 0 && x < W) {
            const float gx = vd_lin11(W, x), gy = vd_lin11(H, y);
#pragma unroll
            for (int eye = 0; eye < 2; ++eye) {
              const vd_gs g = vd_gs_params(eye == 0 ? gx + sv[j] : gx - sv[j], gy, W, H);
              const float* r0 = D + (size_t)g.yn * W + g.xw;
              const float vnw = r0[0], vne = g.e_ok ? r0[1] : 0.f;
              float v;
              if (g.s_ok && (g.sw != 0.f || g.se != 0.f)) {
                const float vsw = r0[W], vse = g.e_ok ? r0[W + 1] : 0.f;
                v = vd_gs_combine(g, vnw, vne, vsw, vse);
              } else {
                v = vd_fma(vne, g.ne, vnw * g.nw);   // sw = se = 0 exactly: the south samples add +0
              }
              if (eye == 0) vl = v; else vr = v;
            }
          }
          wd[t] = vl; wd[wh * ww + t] = vr;
        }
      }
    }
  }
  if (RESIZE) {
#pragma unroll
    for (int j = 0; j < WF_PF; ++j) {
      const int t = tid + j * WF_NT;
      if (t < 3 * er * ec) tile[t] = pf[j];
    }
    const size_t ni = (size_t)a.ih * a.iw;
    for (int t = tid + WF_PF * WF_NT; t < 3 * er * ec; t += WF_NT) {  // only for very large shift bounds
      const int c = t / (er * ec), rem = t - c * er * ec, ty = rem / ec, tx = rem - ty * ec;
      tile[t] = (ec0 + tx < a.iw) ? rgb[c * ni + (size_t)(er0 + ty) * a.iw + (ec0 + tx)] : 0.f;
    }
  }
  __syncthreads();
  if (a.feather) {
    // phase B: e2 = clamp(|grad WD| * fs, 0, 1) (:347-352), zero outside the image (avg_pool2d zero padding)
    for (int t = tid; t < eh * ew; t += WF_NT) {
      const int ty = wf_div(t, a.m_ew), tx = t - ty * ew;
      const int y = y0 - r + ty, x = x0 - r + tx;
      float el = 0.f, er_ = 0.f;
      if (y >= 0 && y < H && x >= 0 && x < W) {
#pragma unroll
        for (int eye = 0; eye < 2; ++eye) {
          const float* wv = wd + eye * wh * ww + (ty + 1) * ww + (tx + 1);
          const float c = wv[0];
          const float gx = x > 0 ? c - wv[-1] : 0.f;
          const float gy = y > 0 ? c - wv[-ww] : 0.f;
          const float v = vd_clamp(sqrtf(gx * gx + gy * gy) * a.fs, 0.f, 1.f);
          if (eye == 0) el = v; else er_ = v;
        }
      }
      e2[t] = el; e2[eh * ew + t] = er_;
    }
    __syncthreads();
    // phase C: horizontal window sums (ascending x) into the dead wd buffer
    float* hs = wd;
    for (int t = tid; t < eh * WF_TW; t += WF_NT) {
      const int ty = t >> 6, tx = t & 63;
#pragma unroll
      for (int eye = 0; eye < 2; ++eye) {
        const float* row = e2 + eye * eh * ew + ty * ew + tx;
        float sacc = 0.f;
        for (int j = 0; j < k; ++j) sacc += row[j];
        hs[eye * eh * WF_TW + t] = sacc;
      }
    }
  }
  __syncthreads();
  // phase D: one wave = 64 consecutive pixels of ONE row per iteration, so everything that depends on y only
  // (sample rows yn / yn+1, their resize taps, the vertical weights) is wave-uniform.  ~70 % of rows have an exactly
  // integral sample row (n == 0): there sw = se = 0 and the two south samples contribute exactly +0 -> skipped
  // (bit-exact: fma(v, 0, acc) == acc for finite v).
  const float* hs = wd;
  const float div = (float)(k * k);
  const size_t ni = (size_t)a.ih * a.iw;
  const int lane = tid & 63, wv = tid >> 6;
  for (int ty = wv; ty < WF_TH; ty += WF_NT / 64) {
    const int y = y0 + ty;
    if (y >= H) break;
    const float gy = vd_lin11(H, y);
    const vd_gs grow = vd_gs_params(0.f, gy, W, H);      // row part: yn, s_ok and whether n == 0
    const bool south = grow.s_ok && (grow.sw != 0.f || grow.se != 0.f);   // n != 0 (e + w == 1, so not both products vanish)
    const vd_tap tyo = wf_tap(a.ih, H, a.scale_h, y);
    const vd_tap ty0 = wf_tap(a.ih, H, a.scale_h, grow.yn), ty1 = wf_tap(a.ih, H, a.scale_h, min(grow.yn + 1, H - 1));
    const int x = x0 + lane;
    uint32_t pL = 0, pR = 0;
    if (x < W) {
      const size_t o = (size_t)y * W + x;
      float b[2] = {0.f, 0.f};
      if (a.feather) {
#pragma unroll
        for (int eye = 0; eye < 2; ++eye) {
          const float* col = hs + eye * eh * WF_TW + ty * WF_TW + lane;
          float sacc = 0.f;
          for (int i = 0; i < k; ++i) sacc += col[i * WF_TW];
          b[eye] = sacc / div;
        }
      }
      const float s = S[o];
      const float gx0 = vd_lin11(W, x);
      const vd_gs gl = vd_gs_params(gx0 + s, gy, W, H), gr = vd_gs_params(gx0 - s, gy, W, H);
      if (RESIZE) {
        const vd_tap txo = wf_tap(a.iw, W, a.scale_w, x);
        const vd_tap tl0 = wf_tap(a.iw, W, a.scale_w, gl.xw), tl1 = wf_tap(a.iw, W, a.scale_w, min(gl.xw + 1, W - 1));
        const vd_tap tr0 = wf_tap(a.iw, W, a.scale_w, gr.xw), tr1 = wf_tap(a.iw, W, a.scale_w, min(gr.xw + 1, W - 1));
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float* tc = tile + c * er * ec;
          auto smp = [&](const vd_tap& yy, const vd_tap& xx) {
            const float* q0 = tc + (yy.i0 - er0) * ec - ec0;
            const float* q1 = tc + (yy.i1 - er0) * ec - ec0;
            return vd_bilerp(q0[xx.i0], q0[xx.i1], q1[xx.i0], q1[xx.i1], xx.w0, xx.w1, yy.w0, yy.w1);
          };
          const float orig = smp(tyo, txo);
#pragma unroll
          for (int eye = 0; eye < 2; ++eye) {
            const vd_gs& g = eye == 0 ? gl : gr;
            const vd_tap& xa = eye == 0 ? tl0 : tr0;
            const vd_tap& xb = eye == 0 ? tl1 : tr1;
            const float vnw = smp(ty0, xa);
            const float vne = g.e_ok ? smp(ty0, xb) : 0.f;
            float v;
            if (south) {
              const float vsw = smp(ty1, xa);
              const float vse = g.e_ok ? smp(ty1, xb) : 0.f;
              v = vd_gs_combine(g, vnw, vne, vsw, vse);
            } else {
              v = vd_fma(vne, g.ne, vnw * g.nw);   // == vd_gs_combine with sw = se = 0 (or no south row)
            }
            if (a.feather) v = vd_clamp(v * (1.0f - b[eye]) + orig * b[eye], 0.f, 1.f);
            const uint32_t u = (uint32_t)(uint8_t)(v * 255.0f);
            if (eye == 0) pL |= u << (8 * (2 - c)); else pR |= u << (8 * (2 - c));
          }
        }
      } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float* pl = rgb + c * ni;
          const float orig = pl[o];
#pragma unroll
          for (int eye = 0; eye < 2; ++eye) {
            const vd_gs& g = eye == 0 ? gl : gr;
            const float* r0 = pl + (size_t)g.yn * W;
            const float vnw = r0[g.xw], vne = g.e_ok ? r0[g.xw + 1] : 0.f;
            float v;
            if (south) {
              const float vsw = r0[W + g.xw], vse = g.e_ok ? r0[W + g.xw + 1] : 0.f;
              v = vd_gs_combine(g, vnw, vne, vsw, vse);
            } else {
              v = vd_fma(vne, g.ne, vnw * g.nw);
            }
            if (a.feather) v = vd_clamp(v * (1.0f - b[eye]) + orig * b[eye], 0.f, 1.f);
            const uint32_t u = (uint32_t)(uint8_t)(v * 255.0f);
            if (eye == 0) pL |= u << (8 * (2 - c)); else pR |= u << (8 * (2 - c));
          }
        }
      }
    }
    // pack 4 lanes x 3 bytes into 3 dwords (lanes 4j, 4j+1, 4j+2 store) : pX = B | G<<8 | R<<16
    const uint32_t nL = (uint32_t)__shfl_down((int)pL, 1, 64), nR = (uint32_t)__shfl_down((int)pR, 1, 64);
    const int q = lane & 3;
    const int xq = x0 + (lane & ~3);
    const bool full = (xq + 3 < W) && ((((size_t)y * W + xq) * 3) % 4 == 0);
    if (full) {
      if (q < 3) {
        uint32_t dL, dR;
        if (q == 0) { dL = pL | (nL << 24); dR = pR | (nR << 24); }
        else if (q == 1) { dL = (pL >> 8) | (nL << 16); dR = (pR >> 8) | (nR << 16); }
        else { dL = (pL >> 16) | (nL << 8); dR = (pR >> 16) | (nR << 8); }
        reinterpret_cast<uint32_t*>(L + ((size_t)y * W + xq) * 3)[q] = dL;
        reinterpret_cast<uint32_t*>(R + ((size_t)y * W + xq) * 3)[q] = dR;
      }
    } else if (x < W) {
      uint8_t* ol = L + ((size_t)y * W + x) * 3;
      uint8_t* orr = R + ((size_t)y * W + x) * 3;
      ol[0] = (uint8_t)pL; ol[1] = (uint8_t)(pL >> 8); ol[2] = (uint8_t)(pL >> 16);
      orr[0] = (uint8_t)pR; orr[1] = (uint8_t)(pR >> 8); orr[2] = (uint8_t)(pR >> 16);
    }
  }
}

// returns false when the fused kernel cannot be used (tiles would not fit the 160 KB LDS): caller falls back to v0
bool vd_launch_warp_fused(hipStream_t s, const float* rgb, int ih, int iw, const float* D, const float* S, int H, int W,
                          const vd3d_shift_params& p, uint8_t* L, uint8_t* R) {
  vd_wf_args a;
  a.ih = ih; a.iw = iw; a.H = H; a.W = W; a.k = p.enable_feathering ? p.blur_ksize : 1; a.feather = p.enable_feathering ? 1 : 0;
  a.fs = (float)p.feather_strength;
  a.scale_h = (float)ih / (float)H; a.scale_w = (float)iw / (float)W;
  // |final shift| <= clamp bound + |convergence| (edge-mask blend is a convex shrink); pixels = S * (W-1)/2
  const double half_width = (double)W / 2.0;
  const double smax = ((double)W * p.max_pixel_shift_percent) / half_width + fabs(p.convergence_strength) / half_width;
  a.bound = (int)ceil(smax * (double)(W - 1) / 2.0 * 1.0001) + 2;
  const bool resize = !(ih == H && iw == W);
  a.er_max = a.ec_max = 0;
  const int k = a.k;
  size_t fl = a.feather ? (size_t)2 * (WF_TH + k) * (WF_TW + k) + (size_t)2 * (WF_TH + k - 1) * (WF_TW + k - 1) : 0;
  if (resize) {
    a.er_max = (int)ceil((WF_TH + 2) * (double)a.scale_h) + 3;
    a.ec_max = (int)ceil((WF_TW + 2 * a.bound + 3) * (double)a.scale_w) + 3;
    a.ec_max |= 1;                       // odd row pitch: the 4-rows-per-wave gathers of phase D land on distinct banks
    if (a.ec_max > iw) a.ec_max = iw;
    fl += (size_t)3 * a.er_max * a.ec_max;
  }
  auto magic = [](int d) { return (uint32_t)(((1ull << 32) + (uint64_t)d - 1) / (uint64_t)(d > 0 ? d : 1)); };
  a.m_ww = magic(WF_TW + k); a.m_ew = magic(WF_TW + k - 1); a.m_ec = magic(a.ec_max > 0 ? a.ec_max : 1); a.m_erec = 0;
  if ((WF_TH + k) * (WF_TW + k) >= 65536 || 3 * a.er_max * a.ec_max >= 65536) return false;
  const size_t bytes = fl * sizeof(float);
  if (bytes > 78 * 1024) return false;  // keep 2 workgroups per CU
  dim3 g((W + WF_TW - 1) / WF_TW, (H + WF_TH - 1) / WF_TH);
  if (resize) {
    static bool attr1 = false;
    if (!attr1) { (void)hipFuncSetAttribute((const void*)k_warp_fused<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr1 = true; }
    hipLaunchKernelGGL(k_warp_fused<true>, g, dim3(WF_NT), bytes, s, rgb, D, S, a, L, R);
  } else {
    static bool attr0 = false;
    if (!attr0) { (void)hipFuncSetAttribute((const void*)k_warp_fused<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr0 = true; }
    hipLaunchKernelGGL(k_warp_fused<false>, g, dim3(WF_NT), bytes, s, rgb, D, S, a, L, R);
  }
  return true;
}
