// vd3d_warp.hip -- W1, the fused parallax-warp kernel (the kernel BASELINE.json:north_star sets the HBM target on).
//
// One launch does what the reference does with 4 grid_sample calls + 2 feather_shift_edges
// (core/render_3d.py:684-712): warped-depth gradient mask -> k x k separable window average -> RGB warp of both
// eyes -> feather blend -> tensor_to_frame truncation, with NO intermediate planes in HBM.
//
// Per 64x32 output tile (512 threads, 3 workgroups per CU at 4K / k = 9: the LDS buffers alias, see the kernel):
//   tables  row taps (wave-uniform) and column taps of the resize, grid_sample row parts -> LDS, once per tile
//   phase A warped depth of BOTH eyes (one packed-f32 vector) on the (TH+k) x (TW+k) halo: S loads, then D gathers (L2)
//   phase B e2 = clamp(|grad WD| * fs, 0, 1)            phase C horizontal k-sums (ascending x)
//           (the eye-res RGB tile is prefetched into registers during B / C and lands over the dead wd / e2 buffers)
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
  int ncol;                             // entries of the column-tap table (WF_TW + 2*bound + 6)
  int tab_off;                          // float offset of the tables in LDS
  uint32_t m_ww, m_ew, m_ec, m_erec;    // ceil(2^32/d) reciprocals: q = umulhi(t, m) is exact for t, d < 2^16
  float fs, scale_h, scale_w;
  float step_x, step_y;                 // linspace steps (1-(-1))/(float)(W-1), .../(H-1): vd_lin11_step
};
// LDS tables (built once per tile, so the per-pixel phases only do table look-ups):
//   rowA[wh][4]   per halo row of phase A : yn, n, 1-n, south flag                      (grid_sample row part)
//   rowD[TH][16]  per tile row of phase D : 3 resize taps (orig / yn / yn+1) as tile row offsets + weights, n, 1-n, south, yn
//   colT[ncol][2] per warp-res column     : resize tap of that column as tile column offset + weight (i1 = i0+1: the tile
//                                           keeps a duplicate of the last image column)
#define WF_RD 16
VD_DEV int wf_div(int t, uint32_t m) { return (int)__umulhi((uint32_t)t, m); }

VD_DEV vd_tap wf_tap(int in, int out, float scale, int o) {  // vd_interp_tap with the scale hoisted
  vd_tap t;
  if (in == out) { t.i0 = o; t.i1 = o; t.w0 = 1.f; t.w1 = 0.f; return t; }
  float src = vd_fma(scale, (float)o + 0.5f, -0.5f);   // area_pixel_compute_source_index: ONE fused multiply-add in ATen's builds
  if (src < 0.f) src = 0.f;
  int i0 = (int)floorf(src);
  if (i0 > in - 1) i0 = in - 1;
  float l1 = vd_clamp_fin(src - (float)i0, 0.f, 1.f);
  t.i0 = i0; t.i1 = i0 + (i0 < in - 1 ? 1 : 0); t.w1 = l1; t.w0 = 1.f - l1;
  return t;
}

// ---- both eyes as ONE packed-float32 vector (x = left eye, y = right eye): every float operation of the two grid_sample
// chains is issued once as v_pk_mul/add/fma_f32; each element is rounded exactly like the scalar helper it mirrors.
struct wf_gs2 { int xw[2]; vd_f2 nw, ne, sw, se; bool e_ok[2]; };
// vd_gs_params for (gx + s, gx - s) sharing the row part (yn, n, s_ = 1 - n computed by the caller)
VD_DEV wf_gs2 wf_gs_params2(float gx, float s, float n, float sr, int W) {
  wf_gs2 p;
  const vd_f2 g = {gx + s, gx - s};
  vd_f2 ix = (g + 1.f) * ((float)(W - 1) / 2.f);
  ix.x = fminf((float)(W - 1), fmaxf(ix.x, 0.f)); ix.y = fminf((float)(W - 1), fmaxf(ix.y, 0.f));
  const vd_f2 xw = {floorf(ix.x), floorf(ix.y)};
  const vd_f2 w = ix - xw, e = 1.f - w;
  p.nw = sr * e; p.ne = sr * w; p.sw = n * e; p.se = n * w;
  p.xw[0] = (int)xw.x; p.xw[1] = (int)xw.y;
  p.e_ok[0] = (p.xw[0] + 1) < W; p.e_ok[1] = (p.xw[1] + 1) < W;
  return p;
}
// row part of vd_gs_params: clamped iy -> yn, n = iy - yn, s = 1 - n
VD_DEV void wf_gs_row(float gy, int H, int* yn, float* n, float* sr, bool* s_ok) {
  float iy = (gy + 1.f) * ((float)(H - 1) / 2.f);
  iy = fminf((float)(H - 1), fmaxf(iy, 0.f));
  const float f = floorf(iy);
  *n = iy - f; *sr = 1.f - (iy - f); *yn = (int)f; *s_ok = ((int)f + 1) < H;
}
// RGB tile: prefetched registers -> LDS (+ the rare overflow elements straight from global)
#define WF_STORE_TILE                                                                                                   \
  {                                                                                                                     \
    _Pragma("unroll") for (int j = 0; j < WF_PF; ++j) {                                                                 \
      const int t = tid + j * WF_NT;                                                                                    \
      if (t < 3 * er * ec) tile[t] = pf[j];                                                                             \
    }                                                                                                                   \
    const unsigned ni_ = (unsigned)a.ih * (unsigned)a.iw;                                                               \
    for (int t = tid + WF_PF * WF_NT; t < 3 * er * ec; t += WF_NT) { /* only for very large shift bounds */            \
      const int c = t / (er * ec), rem = t - c * er * ec, ty = rem / ec, tx = rem - ty * ec;                            \
      tile[t] = rgb[(unsigned)c * ni_ + (unsigned)(er0 + ty) * (unsigned)a.iw + (unsigned)min(ec0 + tx, a.iw - 1)];    \
    }                                                                                                                   \
  }
// LDS map (floats):  wd2[wh*ww][2] (later hs2[eh*TW][2]) | e2_2[eh*ew][2] | tile[3*er*ec] | rowD | rowA | colT      ([..][2] = eyes)
#define WF_AI 6  // phase-A positions per thread ((TH+k)(TW+k) <= WF_AI*WF_NT for k <= 9; larger k loops)
template <bool RESIZE, bool FEATHER>
__global__ __launch_bounds__(WF_NT) void k_warp_fused(const float* __restrict__ rgb, const float* __restrict__ D,
                                                      const float* __restrict__ S, vd_wf_args a, uint8_t* __restrict__ L,
                                                      uint8_t* __restrict__ R) {
  extern __shared__ float lds[];
  const int H = a.H, W = a.W, k = a.k, r = k / 2;
  const int x0 = blockIdx.x * WF_TW, y0 = blockIdx.y * WF_TH;
  const int ww = WF_TW + k, wh = WF_TH + k;          // wd region
  const int ew = WF_TW + k - 1, eh = WF_TH + k - 1;  // e2 region
  // LDS aliasing (floats): wd2 [0, 2*wh*ww) is dead after phase B and becomes hs2 [0, 2*eh*TW); e2_2 follows wd2 and is dead
  // after phase C; the RGB tile is parked in registers until then and lands at [2*eh*TW, ...) over the dead tail of wd2 and
  // e2_2.  Live maximum = hs2 + tile (+ tables) = 53 KB at 4K / k = 9  =>  3 workgroups per CU instead of 2.
  vd_f2* wd = reinterpret_cast<vd_f2*>(lds);                       // [wh*ww]   (later hs [eh*WF_TW])
  vd_f2* e2 = reinterpret_cast<vd_f2*>(lds + 2 * wh * ww);         // [eh*ew]
  float* tile = FEATHER ? lds + 2 * eh * WF_TW : lds;              // [3][er][ec]
  float* rowD = lds + a.tab_off;                                    // [WF_TH][WF_RD], 16 B aligned (host: after the aliased buffers)
  float* rowA = rowD + WF_TH * WF_RD;                               // [wh][4]
  float* colT = rowA + wh * 4;                                      // [ncol][2]
  const int tid = threadIdx.x;
  const int wy0 = y0 - r - 1, wx0 = x0 - r - 1;
  const int cb = max(x0 - a.bound - 2, 0);                          // first warp-res column of colT

  // eye-res RGB tile: global -> registers now, registers -> LDS after phase A (latency hidden behind phase A)
  int er0 = 0, ec0 = 0, er = 0, ec = 0;
  float pf[WF_PF];
  if (RESIZE) {
    const int ya = max(y0 - 1, 0), yb = min(y0 + WF_TH, H - 1);
    const int xa = max(x0 - a.bound - 1, 0), xb = min(x0 + WF_TW + a.bound + 1, W - 1);
    er0 = wf_tap(a.ih, H, a.scale_h, ya).i0; er = wf_tap(a.ih, H, a.scale_h, yb).i1 - er0 + 1;
    ec0 = wf_tap(a.iw, W, a.scale_w, xa).i0; ec = wf_tap(a.iw, W, a.scale_w, xb).i1 - ec0 + 1;
    er = min(er, a.er_max); ec = a.ec_max;  // fixed row pitch (host constant) so the reciprocals apply
    ec0 = min(ec0, a.iw + 1 - ec); ec0 = max(ec0, 0);   // tile column iw - ec0 (if inside) duplicates the last image column
    if (!FEATHER) {
    const unsigned ni = (unsigned)a.ih * (unsigned)a.iw;
    // element t = tid + j*WF_NT of the [3*er][ec] tile, (row, tx) advanced incrementally (no integer division / 32-bit multiply)
    const int pq = WF_NT / ec, pr = WF_NT - pq * ec;
    int prow = wf_div(tid, a.m_ec), ptx = tid - prow * ec;
#pragma unroll
    for (int j = 0; j < WF_PF; ++j) {
      float v = 0.f;
      if (prow < 3 * er) {
        const int c = prow >= 2 * er ? 2 : (prow >= er ? 1 : 0);
        const int ty = prow - c * er;
        const unsigned pb = c == 2 ? 2u * ni : (c == 1 ? ni : 0u);
        v = rgb[pb + __umul24((unsigned)(er0 + ty), (unsigned)a.iw) + (unsigned)min(ec0 + ptx, a.iw - 1)];
      }
      pf[j] = v;
      prow += pq; ptx += pr;
      if (ptx >= ec) { ptx -= ec; ++prow; }
    }
      }
  }
  // ---- tables
  if (tid < wh) {   // phase-A rows
    const int y = wy0 + tid;
    int yn = 0; float n = 0.f, sr = 1.f; bool s_ok = false;
    if (y >= 0 && y < H) wf_gs_row(vd_lin11_step(a.step_y, H, y), H, &yn, &n, &sr, &s_ok);
    float* t = rowA + tid * 4;
    t[0] = __int_as_float(yn); t[1] = n; t[2] = sr; t[3] = __int_as_float((s_ok && n != 0.f) ? 1 : 0);
  } else if (tid >= 64 && tid < 64 + WF_TH) {   // phase-D rows
    const int ty = tid - 64, y = min(y0 + ty, H - 1);
    int yn; float n, sr; bool s_ok;
    wf_gs_row(vd_lin11_step(a.step_y, H, y), H, &yn, &n, &sr, &s_ok);
    float* t = rowD + ty * WF_RD;
    const vd_tap to = wf_tap(a.ih, H, a.scale_h, y), t0 = wf_tap(a.ih, H, a.scale_h, yn), t1 = wf_tap(a.ih, H, a.scale_h, min(yn + 1, H - 1));
    t[0] = __int_as_float((to.i0 - er0) * ec); t[1] = __int_as_float((to.i1 - er0) * ec); t[2] = to.w0; t[3] = to.w1;
    t[4] = __int_as_float((t0.i0 - er0) * ec); t[5] = __int_as_float((t0.i1 - er0) * ec); t[6] = t0.w0; t[7] = t0.w1;
    t[8] = __int_as_float((t1.i0 - er0) * ec); t[9] = __int_as_float((t1.i1 - er0) * ec); t[10] = t1.w0; t[11] = t1.w1;
    t[12] = n; t[13] = sr; t[14] = __int_as_float((s_ok && n != 0.f) ? 1 : 0); t[15] = __int_as_float(yn);
  }
  if (RESIZE) {
    for (int j = tid; j < a.ncol; j += WF_NT) {
      const vd_tap t = wf_tap(a.iw, W, a.scale_w, min(cb + j, W - 1));
      colT[2 * j] = __int_as_float(t.i0 - ec0); colT[2 * j + 1] = t.w1;
    }
  }
  __syncthreads();
  if (FEATHER) {
    // phase A: warped depth of both eyes on the (TH+k) x (TW+k) halo region (grid_sample of D, :700-701).
    // Two passes with a fixed unroll so all S loads, then all D gathers, are in flight together.
    for (int base = 0; base < wh * ww; base += WF_AI * WF_NT) {
      float sv[WF_AI];
#pragma unroll
      for (int j = 0; j < WF_AI; ++j) {
        const int t = base + tid + j * WF_NT;
        const int ty = wf_div(t, a.m_ww), tx = t - ty * ww;
        const int y = wy0 + ty, x = wx0 + tx;
        sv[j] = (t < wh * ww && y >= 0 && y < H && x >= 0 && x < W) ? S[__umul24((unsigned)y, (unsigned)W) + (unsigned)x] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < WF_AI; ++j) {
        const int t = base + tid + j * WF_NT;
        if (t < wh * ww) {
          const int ty = wf_div(t, a.m_ww), tx = t - ty * ww;
          const int y = wy0 + ty, x = wx0 + tx;
          vd_f2 v = {0.f, 0.f};
          if (y >= 0 && y < H && x >= 0 && x < W) {
            const float gx = vd_lin11_step(a.step_x, W, x);
            const vd_f4 rt = *reinterpret_cast<const vd_f4*>(rowA + ty * 4);
            const int yn = __float_as_int(rt.x);
            const wf_gs2 g = wf_gs_params2(gx, sv[j], rt.y, rt.z, W);
            const unsigned rb = __umul24((unsigned)yn, (unsigned)W);
            const float* r0 = D + (rb + (unsigned)g.xw[0]);
            const float* r1 = D + (rb + (unsigned)g.xw[1]);
            const vd_f2 vnw = {r0[0], r1[0]};
            const vd_f2 vne = {g.e_ok[0] ? r0[1] : 0.f, g.e_ok[1] ? r1[1] : 0.f};
            // vd_gs_combine; when n == 0 (or no south row) sw = se = 0 exactly and the south samples add +0
            vd_f2 acc = vd_vfma(vne, g.ne, vnw * g.nw);
            if (__float_as_int(rt.w)) {
              const vd_f2 vsw = {r0[W], r1[W]};
              const vd_f2 vse = {g.e_ok[0] ? r0[W + 1] : 0.f, g.e_ok[1] ? r1[W + 1] : 0.f};
              acc = vd_vfma(vse, g.se, vd_vfma(vsw, g.sw, acc));
            }
            v = acc;
          }
          wd[t] = v;
        }
      }
    }
  }
  if (RESIZE && !FEATHER) {
    WF_STORE_TILE
  }
  __syncthreads();
  if (FEATHER && RESIZE) {   // tile prefetch: in flight during phases B and C only (keeps the register footprint of phase A small)
    const unsigned ni = (unsigned)a.ih * (unsigned)a.iw;
    // element t = tid + j*WF_NT of the [3*er][ec] tile, (row, tx) advanced incrementally (no integer division / 32-bit multiply)
    const int pq = WF_NT / ec, pr = WF_NT - pq * ec;
    int prow = wf_div(tid, a.m_ec), ptx = tid - prow * ec;
#pragma unroll
    for (int j = 0; j < WF_PF; ++j) {
      float v = 0.f;
      if (prow < 3 * er) {
        const int c = prow >= 2 * er ? 2 : (prow >= er ? 1 : 0);
        const int ty = prow - c * er;
        const unsigned pb = c == 2 ? 2u * ni : (c == 1 ? ni : 0u);
        v = rgb[pb + __umul24((unsigned)(er0 + ty), (unsigned)a.iw) + (unsigned)min(ec0 + ptx, a.iw - 1)];
      }
      pf[j] = v;
      prow += pq; ptx += pr;
      if (ptx >= ec) { ptx -= ec; ++prow; }
    }
    }
  if (FEATHER) {
    // phase B: e2 = clamp(|grad WD| * fs, 0, 1) (:347-352), zero outside the image (avg_pool2d zero padding)
    const int bq = WF_NT / ew, br = WF_NT - bq * ew;
    int ty = wf_div(tid, a.m_ew), tx = tid - ty * ew;
    for (int t = tid; t < eh * ew; t += WF_NT) {
      const int y = y0 - r + ty, x = x0 - r + tx;
      vd_f2 e = {0.f, 0.f};
      if (y >= 0 && y < H && x >= 0 && x < W) {
        const vd_f2* wv = wd + __umul24((unsigned)(ty + 1), (unsigned)ww) + (tx + 1);
        const vd_f2 c = wv[0];
        const vd_f2 z = {0.f, 0.f};
        const vd_f2 gx = x > 0 ? c - wv[-1] : z;
        const vd_f2 gy = y > 0 ? c - wv[-ww] : z;
        const vd_f2 q = gx * gx + gy * gy;
        const vd_f2 m = vd_f2{sqrtf(q.x), sqrtf(q.y)} * a.fs;
        e.x = vd_clamp_fin(m.x, 0.f, 1.f); e.y = vd_clamp_fin(m.y, 0.f, 1.f);
      }
      e2[t] = e;
      ty += bq; tx += br;
      if (tx >= ew) { tx -= ew; ++ty; }
    }
    __syncthreads();
    // phase C: horizontal window sums (ascending x) into the dead wd buffer
    vd_f2* hs = wd;
    for (int t = tid; t < eh * WF_TW; t += WF_NT) {
      const int ty = t >> 6, tx = t & 63;
      const vd_f2* row = e2 + ty * ew + tx;
      vd_f2 sacc = {0.f, 0.f};
      for (int j = 0; j < k; ++j) sacc += row[j];
      hs[t] = sacc;
    }
    if (RESIZE) {
      __syncthreads();   // every read of e2 is done: the tile may overwrite it
      WF_STORE_TILE
    }
  }
  __syncthreads();
  // phase D: one wave = 64 consecutive pixels of ONE row per iteration, so everything that depends on y only
  // (sample rows yn / yn+1, their resize taps, the vertical weights) is wave-uniform and comes from rowD.  ~70 % of rows
  // have an exactly integral sample row (n == 0): there sw = se = 0 and the two south samples contribute exactly +0 ->
  // skipped (bit-exact: fma(v, 0, acc) == acc for finite v).
  const vd_f2* hs = wd;
  const float div = (float)(k * k);
  const unsigned ni = (unsigned)a.ih * (unsigned)a.iw;
  const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int ty = wv; ty < WF_TH; ty += WF_NT / 64) {
    const int y = y0 + ty;
    if (y >= H) break;
    const vd_f4* rt = reinterpret_cast<const vd_f4*>(rowD + ty * WF_RD);
    const vd_f4 ro = rt[0], ra = rt[1], rb = rt[2], rs = rt[3];
    const float n = rs.x, sr = rs.y;
    const bool south = __float_as_int(rs.z) != 0;
    const int yn = __float_as_int(rs.w);
    const int x = x0 + lane;
    uint32_t pL = 0, pR = 0;
    if (x < W) {
      const unsigned o = __umul24((unsigned)y, (unsigned)W) + (unsigned)x;
      vd_f2 b = {0.f, 0.f};
      if (FEATHER) {
        const vd_f2* col = hs + ty * WF_TW + lane;
        vd_f2 sacc = {0.f, 0.f};
        for (int i = 0; i < k; ++i) sacc += col[i * WF_TW];
        b.x = sacc.x / div; b.y = sacc.y / div;
      }
      const float s = S[o];
      const float gx0 = vd_lin11_step(a.step_x, W, x);
      const wf_gs2 g = wf_gs_params2(gx0, s, n, sr, W);
      const vd_f2 omb = 1.0f - b;
      if (RESIZE) {
        // column taps from the table: entry j = resize tap of warp-res column cb + j; xw+1 is the next entry
        const vd_f2* ct = reinterpret_cast<const vd_f2*>(colT);
        const vd_f2 eo = ct[x - cb];
        const vd_f2 eaL = ct[g.xw[0] - cb], ebL = ct[g.xw[0] - cb + 1], eaR = ct[g.xw[1] - cb], ebR = ct[g.xw[1] - cb + 1];
        const int iaL = __float_as_int(eaL.x), ibL = __float_as_int(ebL.x), iaR = __float_as_int(eaR.x), ibR = __float_as_int(ebR.x);
        const vd_f2 wa1 = {eaL.y, eaR.y}, wb1 = {ebL.y, ebR.y};
        const vd_f2 wa0 = 1.f - wa1, wb0 = 1.f - wb1;
        const int io = __float_as_int(eo.x);
        const float wo1 = eo.y, wo0 = 1.f - eo.y;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float* tc = tile + c * er * ec;
          // nested bilinear sample of BOTH eyes: tile rows r0o / r1o (wave-uniform), tile columns i (and i+1) per eye
          auto smp2 = [&](const vd_f4& rw, int iL, int iR, const vd_f2& w0, const vd_f2& w1) {
            const float* q0 = tc + __float_as_int(rw.x);
            const float* q1 = tc + __float_as_int(rw.y);
            const vd_f2 p00 = {q0[iL], q0[iR]}, p01 = {q0[iL + 1], q0[iR + 1]};
            const vd_f2 p10 = {q1[iL], q1[iR]}, p11 = {q1[iL + 1], q1[iR + 1]};
            const vd_f2 ua = vd_vfma(p00, w0, w1 * p01);
            const vd_f2 ub = vd_vfma(p10, w0, w1 * p11);
            return vd_vfma(ua, (vd_f2)(rw.z), rw.w * ub);
          };
          float orig;
          {
            const float* q0 = tc + __float_as_int(ro.x);
            const float* q1 = tc + __float_as_int(ro.y);
            orig = vd_bilerp(q0[io], q0[io + 1], q1[io], q1[io + 1], wo0, wo1, ro.z, ro.w);
          }
          const vd_f2 vnw = smp2(ra, iaL, iaR, wa0, wa1);
          vd_f2 vne = smp2(ra, ibL, ibR, wb0, wb1);
          vne.x = g.e_ok[0] ? vne.x : 0.f; vne.y = g.e_ok[1] ? vne.y : 0.f;
          vd_f2 v = vd_vfma(vne, g.ne, vnw * g.nw);
          if (south) {
            const vd_f2 vsw = smp2(rb, iaL, iaR, wa0, wa1);
            vd_f2 vse = smp2(rb, ibL, ibR, wb0, wb1);
            vse.x = g.e_ok[0] ? vse.x : 0.f; vse.y = g.e_ok[1] ? vse.y : 0.f;
            v = vd_vfma(vse, g.se, vd_vfma(vsw, g.sw, v));
          }
          if (FEATHER) { v = v * omb + orig * b; v.x = vd_clamp_fin(v.x, 0.f, 1.f); v.y = vd_clamp_fin(v.y, 0.f, 1.f); }
          const vd_f2 u = v * 255.0f;
          pL |= (uint32_t)(uint8_t)u.x << (8 * (2 - c));
          pR |= (uint32_t)(uint8_t)u.y << (8 * (2 - c));
        }
      } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float* pl = rgb + (unsigned)c * ni;
          const float orig = pl[o];
          const float* r0 = pl + __umul24((unsigned)yn, (unsigned)W);
          const vd_f2 vnw = {r0[g.xw[0]], r0[g.xw[1]]};
          const vd_f2 vne = {g.e_ok[0] ? r0[g.xw[0] + 1] : 0.f, g.e_ok[1] ? r0[g.xw[1] + 1] : 0.f};
          vd_f2 v = vd_vfma(vne, g.ne, vnw * g.nw);
          if (south) {
            const vd_f2 vsw = {r0[W + g.xw[0]], r0[W + g.xw[1]]};
            const vd_f2 vse = {g.e_ok[0] ? r0[W + g.xw[0] + 1] : 0.f, g.e_ok[1] ? r0[W + g.xw[1] + 1] : 0.f};
            v = vd_vfma(vse, g.se, vd_vfma(vsw, g.sw, v));
          }
          if (FEATHER) { v = v * omb + orig * b; v.x = vd_clamp_fin(v.x, 0.f, 1.f); v.y = vd_clamp_fin(v.y, 0.f, 1.f); }
          const vd_f2 u = v * 255.0f;
          pL |= (uint32_t)(uint8_t)u.x << (8 * (2 - c));
          pR |= (uint32_t)(uint8_t)u.y << (8 * (2 - c));
        }
      }
    }
    // pack 4 lanes x 3 bytes into 3 dwords (lanes 4j, 4j+1, 4j+2 store) : pX = B | G<<8 | R<<16
    const uint32_t nL = (uint32_t)__shfl_down((int)pL, 1, 64), nR = (uint32_t)__shfl_down((int)pR, 1, 64);
    const int q = lane & 3;
    const int xq = x0 + (lane & ~3);
    const unsigned ob = (__umul24((unsigned)y, (unsigned)W) + (unsigned)xq) * 3u;
    const bool full = (xq + 3 < W) && (ob % 4u == 0);
    if (full) {
      if (q < 3) {
        uint32_t dL, dR;
        if (q == 0) { dL = pL | (nL << 24); dR = pR | (nR << 24); }
        else if (q == 1) { dL = (pL >> 8) | (nL << 16); dR = (pR >> 8) | (nR << 16); }
        else { dL = (pL >> 16) | (nL << 8); dR = (pR >> 16) | (nR << 8); }
        reinterpret_cast<uint32_t*>(L + ob)[q] = dL;
        reinterpret_cast<uint32_t*>(R + ob)[q] = dR;
      }
    } else if (x < W) {
      const unsigned o1 = ((unsigned)y * (unsigned)W + (unsigned)x) * 3u;
      uint8_t* ol = L + o1;
      uint8_t* orr = R + o1;
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
  size_t sz_tile = 0;
  if (resize) {
    a.er_max = (int)ceil((WF_TH + 2) * (double)a.scale_h) + 3;
    a.ec_max = (int)ceil((WF_TW + 2 * a.bound + 3) * (double)a.scale_w) + 3;
    a.ec_max += 1;                       // room for the duplicate of the last image column (table taps use i1 = i0 + 1)
    a.ec_max |= 1;                       // odd row pitch: the 4-rows-per-wave gathers of phase D land on distinct banks
    if (a.ec_max > iw + 1) a.ec_max = iw + 1;
    sz_tile = (size_t)3 * a.er_max * a.ec_max;
  }
  // aliased layout (see the kernel): max(wd2 + e2_2, hs2 + tile) when feathering, else the tile alone
  size_t fl = sz_tile;
  if (a.feather) {
    const size_t sz_wd = (size_t)2 * (WF_TH + k) * (WF_TW + k), sz_e2 = (size_t)2 * (WF_TH + k - 1) * (WF_TW + k - 1);
    const size_t sz_hs = (size_t)2 * (WF_TH + k - 1) * WF_TW;
    fl = sz_wd + sz_e2 > sz_hs + sz_tile ? sz_wd + sz_e2 : sz_hs + sz_tile;
  }
  a.ncol = WF_TW + 2 * a.bound + 6;
  fl = (fl + 3) & ~(size_t)3;            // tables start 16 B aligned (ds_read_b128)
  a.tab_off = (int)fl;
  fl += (size_t)WF_TH * WF_RD + (size_t)(WF_TH + k) * 4 + (size_t)2 * a.ncol;
  if (fl & 3) fl += 4 - (fl & 3);
  a.step_x = (1.f - (-1.f)) / (float)(W - 1); a.step_y = (1.f - (-1.f)) / (float)(H - 1);
  auto magic = [](int d) { return (uint32_t)(((1ull << 32) + (uint64_t)d - 1) / (uint64_t)(d > 0 ? d : 1)); };
  a.m_ww = magic(WF_TW + k); a.m_ew = magic(WF_TW + k - 1); a.m_ec = magic(a.ec_max > 0 ? a.ec_max : 1); a.m_erec = 0;
  if ((WF_TH + k) * (WF_TW + k) >= 65536 || 3 * a.er_max * a.ec_max >= 65536) return false;
  if (H >= (1 << 24) || W >= (1 << 24) || (unsigned long long)H * W * 3ull >= (1ull << 32) || (unsigned long long)ih * iw * 3ull >= (1ull << 32)) return false;  // 24-bit multiplies, 32-bit offsets
  const size_t bytes = fl * sizeof(float);
  if (bytes > 78 * 1024) return false;  // keep >= 2 workgroups per CU (3 when <= 53 KB: 4K / k = 9 needs 53.1 KB)
  dim3 g((W + WF_TW - 1) / WF_TW, (H + WF_TH - 1) / WF_TH);
  static bool attr[64] = {false};   // per device: the attribute belongs to the device's copy of the code object
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && !attr[dev]) {
    (void)hipFuncSetAttribute((const void*)k_warp_fused<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k_warp_fused<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k_warp_fused<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k_warp_fused<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr[dev] = true;
  }
  if (resize && a.feather) hipLaunchKernelGGL((k_warp_fused<true, true>), g, dim3(WF_NT), bytes, s, rgb, D, S, a, L, R);
  else if (resize) hipLaunchKernelGGL((k_warp_fused<true, false>), g, dim3(WF_NT), bytes, s, rgb, D, S, a, L, R);
  else if (a.feather) hipLaunchKernelGGL((k_warp_fused<false, true>), g, dim3(WF_NT), bytes, s, rgb, D, S, a, L, R);
  else hipLaunchKernelGGL((k_warp_fused<false, false>), g, dim3(WF_NT), bytes, s, rgb, D, S, a, L, R);
  return true;
}
