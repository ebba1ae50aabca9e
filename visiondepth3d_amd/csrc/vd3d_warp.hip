// vd3d_warp.hip -- W1, the fused parallax-warp kernel (the kernel BASELINE.json:north_star sets the HBM target on).
//
// One launch does what the reference does with 4 grid_sample calls + 2 feather_shift_edges
// (core/render_3d.py:684-712): warped-depth gradient mask -> k x k separable window average -> RGB warp of both
// eyes -> feather blend -> tensor_to_frame truncation, with NO intermediate planes in HBM.
//
// Per 64x32 output tile (512 threads = 8 waves, 3 workgroups per CU at 4K / k = 9: the LDS buffers alias, see the kernel):
//   tables  row taps (wave-uniform) and column taps of the resize, grid_sample row parts -> LDS, once per tile
//   phase A warped depth of BOTH eyes (one packed-f32 vector) on the (TH+k) x (TW+k) halo.  Mapping: ONE WAVE = ONE ROW, lane =
//           column, so everything that depends on y (sample row, its weights, the south flag, all row base addresses) is
//           scalar and everything that depends on x only (the linspace coordinate) is computed once per lane; the k leftover
//           columns right of the first 64 are walked transposed (wave = column, lane = row).  ~35 VALU per position, was ~150.
//   phase B e2 = clamp(|grad WD| * fs, 0, 1)
//   phase C b = avg_pool2d(e2, k, 1, k/2) in ATen's order: ONE float32 running sum per output over the k x k window, row-major
//           (cpu_avg_pool2d; any other association differs in the last bits of ~70 % of the outputs and shows as 1-LSB eye
//           differences on noisy frames).  One thread = 4 consecutive pixels x both eyes: the (k+3)-column
//           window of every e2 row comes from 16-byte LDS reads, 4 independent packed chains; exact 3-operation division by k*k.
//   phase D RGB samples = nested bilinear (resize of :595 inside the grid_sample of :697) read from pre-interpolated rows in LDS,
//           one row per wave (row-uniform taps; exact skip of the south samples when the sample row is integral), blend, truncate,
//           shuffle-packed 12-byte stores per 4 lanes.
// Arithmetic is identical to the unfused v0 kernels (same helpers, same association) => bit-exact vs the oracle.
//
// Round 6 (profiles/r06_w1_phases.md): the kernel is bound by DEPENDENT WAITS at 2.5 resident workgroups per CU -- not by HBM, not by issue slots -- so phase D reads a
// row's LDS samples in one batch, packs its 12-byte store groups by DPP + v_alignbit, takes its row parameters from a per-geometry table by s_load; without feathering
// the kernel has two barriers, deals D1 flat over the threads and (SHIFT) computes k_shift's values for its own tile, so the frame has no k_shift launch and no S plane.
// Rule learnt the hard way: a load under a wave-uniform `if` makes hipcc put `s_waitcnt vmcnt` at the join -- between loads that do not depend on each other; every load
// batch in this file is therefore straight-line code on clamped indices, with the guard applied to the VALUE afterwards.
//
// Algorithmic HBM bytes per stereo pair: read RGB 3N (eye-res f32 x3 at N/4) + D 4N + S 4N, write 6N.
#include <cstdio>
#include <mutex>
#include "vd3d_dev.h"
#include "vd3d_kernels.h"

#define WF_TW 64
// tile height WF_TH is a template parameter of the kernel; threads = 16 per tile row.  64x32 tiles (512 threads, 3 workgroups per CU); 64x16
// tiles (256 threads, 5 per CU) measured 170 vs 155 us at 4K: the mask phases' halo grows from 1.46x to 1.78x.
#define WF_AB 3   // halo rows of phase A a wave walks together (loads of all of them in flight)

struct vd_wf_args {
  int ih, iw, H, W, k, feather, bound;  // bound: rigorous host-side bound on |pixel shift| (+ margin)
  int er_max;                           // eye-res rows one tile touches (exact maximum over the tile rows, host-computed)
  int nch;                              // warp-res columns of the pre-interpolated rows Hh: WF_TW + 2*bound + 2
  int tab_off;                          // float offset of the tables in LDS
  int e2_off;                           // float offset of the e2 plane in LDS (multiple of 4)
  int fastdiv;                          // 1: x / (k*k) as q0 = x*rc, r = fma(-q0, kk, x), q = fma(r, rc, q0) -- verified exhaustively
  uint32_t m_ew;                        // ceil(2^32/d) reciprocal: q = umulhi(t, m) is exact for t, d < 2^16
  float fs, scale_h, scale_w;
  float step_x, step_y;                 // linspace steps (1-(-1))/(float)(W-1), .../(H-1): vd_lin11_step
  float kk, rc_kk;                      // (float)(k*k) and its correctly rounded reciprocal
  int ntx, ntiles, per, xcd;            // tile grid: tiles per row, tile count, tiles per XCD band, band order on (vd_xcd_tile)
};
// LDS tables (built once per tile, so the per-pixel phases only do table look-ups):
//   rowA[wh][4]   per halo row of phase A : yn, n, 1-n, south flag                      (grid_sample row part)
//   rowD[TH][16]  (until round 5; the LDS region is still reserved) per tile row of phase D: 3 resize taps (orig / yn / yn+1) + weights, n, 1-n, south, yn.
//                 Round 6: ONE table per geometry in global memory (k_wf_rowtab, absolute eye-res rows), read by s_load in phase D2
//   colT[nch][2]  per warp-res column     : resize tap of that column (absolute eye-res column, weight of the next one)
#define WF_RD 16
VD_DEV int wf_div(int t, uint32_t m) { return (int)__umulhi((uint32_t)t, m); }
VD_DEV int wf_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
VD_DEV float wf_unif(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

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
// warped depth of both eyes at one position (grid_sample of D with the shift S, :700-701): the caller supplies the row part
// (drow = D + yn * W, n, sr, south) and the column part (gx); in the main block of phase A the former is scalar and the latter is
// loop-invariant per lane, in the leftover block it is the other way round.
VD_DEV vd_f2 wf_warped_depth(const float* __restrict__ drow, float s, float gx, float n, float sr, bool south, int W) {
  const wf_gs2 g = wf_gs_params2(gx, s, n, sr, W);
  const float* r0 = drow + g.xw[0];
  const float* r1 = drow + g.xw[1];
  const vd_f2 vnw = {r0[0], r1[0]};
  const vd_f2 vne = {g.e_ok[0] ? r0[1] : 0.f, g.e_ok[1] ? r1[1] : 0.f};
  // vd_gs_combine; when n == 0 (or no south row) sw = se = 0 exactly and the south samples add +0
  vd_f2 acc = vd_vfma(vne, g.ne, vnw * g.nw);
  if (south) {
    const vd_f2 vsw = {r0[W], r1[W]};
    const vd_f2 vse = {g.e_ok[0] ? r0[W + 1] : 0.f, g.e_ok[1] ? r1[W + 1] : 0.f};
    acc = vd_vfma(vse, g.se, vd_vfma(vsw, g.sw, acc));
  }
  return acc;
}

// The same in two halves, so that a wave can have the gathers of several positions in flight before it combines any of them.
struct wf_wdl { vd_f2 w, vnw, vne, vsw, vse; bool e0, e1; };   // vne / vse: RAW east samples, zeroed by e0 / e1 in the combine
VD_DEV wf_wdl wf_wd_issue(const float* __restrict__ drow, float s, float gx, bool south, int W) {
  wf_wdl l;
  const vd_f2 g = {gx + s, gx - s};
  vd_f2 ix = (g + 1.f) * ((float)(W - 1) / 2.f);
  ix.x = fminf((float)(W - 1), fmaxf(ix.x, 0.f)); ix.y = fminf((float)(W - 1), fmaxf(ix.y, 0.f));
  const vd_f2 xw = {floorf(ix.x), floorf(ix.y)};
  l.w = ix - xw;
  const int x0 = (int)xw.x, x1 = (int)xw.y;
  const bool e0 = (x0 + 1) < W, e1 = (x1 + 1) < W;
  const float* r0 = drow + x0;
  const float* r1 = drow + x1;
  // branch-free per lane (the east neighbour of the last column is read as the column itself and then zeroed), so that the
  // loads of several positions stay in flight together; `south` is wave-uniform where this is used
  const int o0 = e0 ? 1 : 0, o1 = e1 ? 1 : 0;
  l.vnw = vd_f2{r0[0], r1[0]};
  l.vne = vd_f2{r0[o0], r1[o1]};
  l.vsw = vd_f2{0.f, 0.f}; l.vse = vd_f2{0.f, 0.f};
  if (south) {
    l.vsw = vd_f2{r0[W], r1[W]};
    l.vse = vd_f2{r0[W + o0], r1[W + o1]};
  }
  l.e0 = e0; l.e1 = e1;   // nothing above consumes a loaded value: no s_waitcnt in the issue half
  return l;
}
// Unconditional on purpose (no `if (south)`): with the south samples zero-filled the two extra fused multiply-adds add +0 to a
// non-negative accumulator, i.e. nothing -- and a branch here would let the optimiser fuse this half back onto the issue half.
VD_DEV vd_f2 wf_wd_combine(const wf_wdl& l, float n, float sr) {
  const vd_f2 e = 1.f - l.w;
  const vd_f2 vne = {l.e0 ? l.vne.x : 0.f, l.e1 ? l.vne.y : 0.f}, vse = {l.e0 ? l.vse.x : 0.f, l.e1 ? l.vse.y : 0.f};
  const vd_f2 acc = vd_vfma(vne, sr * l.w, l.vnw * (sr * e));
  return vd_vfma(vse, n * l.w, vd_vfma(l.vsw, n * e, acc));
}

// LDS map (floats):  phases A-C: wd2[wh*ww][2] (later hs2[eh*TW][2]) | e2_2[eh*ew][2]      ([..][2] = eyes)
//                    phase D   : Hh[3][er_max][nch] over the same region (RESIZE)   | then, never aliased: rowD | rowA / colT
#ifndef WF_OCC_ATTR
#define WF_OCC_ATTR   // A/B builds: -DWF_OCC_ATTR='__attribute__((amdgpu_waves_per_eu(8, 8)))'
#endif
#ifndef WF_HB
#define WF_HB 5    // D1 without feathering: Hh elements a thread builds together (2 * WF_HB loads in flight)
#endif
#define WF_HBC 6   // D1 with feathering: Hh rows of its chunk a wave builds together (round 4: 14 measured slower)
VD_STAMP_DECL(wf_stamps);
#ifdef VD_PHASE_STAMPS
extern "C" __attribute__((visibility("default"))) int vd3d_debug_stamps_w1(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(wf_stamps), sizeof(wf_stamps)); }
#endif
VD_OCC_DECL(wf_occ, vd3d_debug_occ_w1)
// torch.sqrt of a finite x >= 0 (vd_sqrt_torch): zero -- every flat pixel -- and the never-reached x < 2^-100 take the rounded root
VD_DEV float wf_sqrt_torch(float x, const int2* __restrict__ tab) {
  const float xs = fmaxf(x, 0x1p-100f);
  const float y = vd_rsqrt14(xs, tab), s = xs * y, h = 0.5f * y;
  float r = vd_fma(vd_fma(-s, s, xs), h, s);
  if (x < 0x1p-100f) r = sqrtf(x);
  return r;
}
// ================================================================================================================================
// W0 `k_e2w` (round 4): the feather gradient mask of both eyes as its own launch -- phases A and B of W1 moved out of its tile.  In W1 they
// ran on the (TH + k) x (TW + k) halo of every tile (1.46x the pixels at 4K / k = 9) behind two dependent global round trips (S, then the
// D gathers) that three resident workgroups could not cover (33 % + 20 % of a workgroup's life: 56 us of 164 per 4K frame).  Here:
//   wd    grid_sample(D, x -/+ S) of both eyes (:700-701) on the tile + one row / column (the gradient's neighbours): 1.05x, 17 KB of LDS,
//         every thread's S loads, then all its D gathers, in flight together
//   e2    clamp(|grad wd| * feather_strength, 0, 1) (:347-352) -> E2[y][x] = (left, right), 8 bytes per pixel; one thread = 4 pixels
// W1 then starts at its phase C with a plain, coalesced tile load of E2.  Same expressions as W1's phases A / B (which the unfused
// fallback and the no-mask variants keep).  (A first version also folded the shift plane itself in -- k_shift_e2: its edge-term tile and
// pow chains had to be recomputed on the mask's halo and the launch took 177 us against k_shift's 64 + this kernel's: not kept.)
#define EW_TW 64
#define EW_TH 32
#define EW_NT 512
#define EW_U 5                         // positions per thread: ceil((TH + 1) (TW + 1) / NT)
struct vd_ew_args { int H, W, vec; float fs, step_x, step_y; };
__global__ __launch_bounds__(EW_NT) void k_e2w(const float* __restrict__ D, const float* __restrict__ S, vd_ew_args a, vd_f2* __restrict__ E2) {
  constexpr int PW = EW_TW + 1, PH = EW_TH + 1, NP = PW * PH;
  static_assert(EW_U * EW_NT >= NP, "positions per thread");
  __shared__ __attribute__((aligned(16))) vd_f2 wd[PH][PW + 3];      // pitch 68 pairs: rows stay 16-byte aligned
  __shared__ __attribute__((aligned(16))) vd_f4 rowT[PH];            // per region row: grid_sample row part (yn, n, 1 - n, south flag)
  __shared__ int2 rs14[64];
  const int H = a.H, W = a.W;
  const int x0 = blockIdx.x * EW_TW, y0 = blockIdx.y * EW_TH;
  const int tid = threadIdx.x;
  vd_stage_rs14(rs14, tid, EW_NT);
  if (tid >= EW_NT - PH) {   // the last PH threads: one region row each
    const int r = tid - (EW_NT - PH), yy = y0 - 1 + r;
    int yn = 0; float n = 0.f, sr = 1.f; bool s_ok = false;
    if (yy >= 0 && yy < H) wf_gs_row(vd_lin11_step(a.step_y, H, yy), H, &yn, &n, &sr, &s_ok);
    rowT[r] = vd_f4{__int_as_float(yn), n, sr, __int_as_float((s_ok && n != 0.f) ? 1 : 0)};
  }
  // positions (row r, column c) of the (TH + 1) x (TW + 1) region: image pixel (y0 - 1 + r, x0 - 1 + c)
  float sv[EW_U]; int py[EW_U], px[EW_U]; bool ok[EW_U];
#pragma unroll
  for (int u = 0; u < EW_U; ++u) {
    const int t = tid + u * EW_NT;
    const int r = t / PW, c = t - r * PW;
    py[u] = y0 - 1 + r; px[u] = x0 - 1 + c;
    ok[u] = t < NP && py[u] >= 0 && py[u] < H && px[u] >= 0 && px[u] < W;
    sv[u] = ok[u] ? S[(unsigned)py[u] * (unsigned)W + (unsigned)px[u]] : 0.f;
  }
  __syncthreads();   // row table complete (the S loads above are in flight across it)
#pragma unroll
  for (int u = 0; u < EW_U; ++u) {
    const int t = tid + u * EW_NT;
    if (t >= NP) continue;
    const int r = t / PW, c = t - r * PW;
    vd_f2 v = {0.f, 0.f};
    if (ok[u]) {   // W1 phase A: row part from the table (wf_gs_row), then wf_warped_depth
      const vd_f4 rt = rowT[r];
      v = wf_warped_depth(D + (unsigned)__float_as_int(rt.x) * (unsigned)W, sv[u], vd_lin11_step(a.step_x, W, px[u]), rt.y, rt.z,
                          __float_as_int(rt.w) != 0, W);
    }
    wd[r][c] = v;
  }
  __syncthreads();
  // e2: thread = (tile row, strip of 4 pixels)
  const int ty = tid >> 4, tx = (tid & 15) * 4;
  const int y = y0 + ty, xs = x0 + tx;
  if (y >= H || xs >= W) return;
  vd_f2 e[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int x = xs + q;
    e[q] = vd_f2{0.f, 0.f};
    if (x < W) {
      const vd_f2 cc = wd[ty + 1][tx + q + 1];
      const vd_f2 z = {0.f, 0.f};
      const vd_f2 gx = x > 0 ? cc - wd[ty + 1][tx + q] : z;
      const vd_f2 gy = y > 0 ? cc - wd[ty][tx + q + 1] : z;
      const vd_f2 qq = gx * gx + gy * gy;
      const vd_f2 m = vd_f2{wf_sqrt_torch(qq.x, rs14), wf_sqrt_torch(qq.y, rs14)} * a.fs;   // torch.sqrt = MKL vsSqrt, not the rounded root
      e[q].x = vd_clamp_fin(m.x, 0.f, 1.f); e[q].y = vd_clamp_fin(m.y, 0.f, 1.f);
    }
  }
  const size_t o = (size_t)y * W + xs;
  if (a.vec) {
    vd_f4* dst = reinterpret_cast<vd_f4*>(E2 + o);
    dst[0] = vd_f4{e[0].x, e[0].y, e[1].x, e[1].y};
    dst[1] = vd_f4{e[2].x, e[2].y, e[3].x, e[3].y};
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) if (xs + q < W) E2[o + q] = e[q];
  }
}
void vd_launch_e2w(hipStream_t s, const float* D, const float* S, int H, int W, float fs, float* E2) {
  vd_ew_args a;
  a.H = H; a.W = W; a.fs = fs;
  a.vec = ((W & 3) == 0 && (reinterpret_cast<uintptr_t>(E2) & 31) == 0) ? 1 : 0;
  a.step_x = (1.f - (-1.f)) / (float)(W - 1); a.step_y = (1.f - (-1.f)) / (float)(H - 1);
  hipLaunchKernelGGL(k_e2w, dim3((W + EW_TW - 1) / EW_TW, (H + EW_TH - 1) / EW_TH), dim3(EW_NT), 0, s, D, S, a, reinterpret_cast<vd_f2*>(E2));
}

// PRE (round 4): the gradient mask e2 of both eyes comes from the E2 plane k_e2w wrote (8 bytes per pixel); the tile's e2 region is a
// plain coalesced load and the kernel starts at phase C -- no rowA table, no phase A / B, no D plane.
// SHIFT (round 6, only without feathering): the tile computes its own shift values -- k_shift's arithmetic, expression for expression, on the tile + the 5 x 5 halo of
// the edge mask -- instead of reading the plane k_shift wrote; S is then the plane to WRITE for a caller who wants it (or NULL), sc / work are k_shift's arguments.
template <bool RESIZE, bool FEATHER, int WF_TH, bool PRE, bool SHIFT>
__global__ __launch_bounds__(WF_TH * 16) WF_OCC_ATTR void k_warp_fused(const float* __restrict__ rgb, const float* __restrict__ D,
                                                      float* __restrict__ S, vd_wf_args a, uint8_t* __restrict__ L,
                                                      uint8_t* __restrict__ R, const vd_f2* __restrict__ E2, const float* __restrict__ rowtab,
                                                      vd_shift_consts sc, const vd_dev_work* __restrict__ work) {
  constexpr int WF_NT = WF_TH * 16, WF_NW = WF_NT / 64;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ int2 rs14[64];                          // VRSQRT14 table of vd_sqrt_torch (phase B)
  const int H = a.H, W = a.W, k = a.k, r = k / 2;
  int tby, tbx;
  if (a.xcd == 2) {   // round 5 (vd3d_debug_tune(9, 2)): tile rows dealt to the XCDs round-robin (vd_xcd_tile_rows, one row per group) instead of one contiguous band each
    vd_xcd_tile_rows(blockIdx.x, a.ntx, 1, 1, &tby, &tbx);
    if (tby * a.ntx >= a.ntiles) return;
  } else {
    const int tile = vd_xcd_tile(blockIdx.x, a.per, a.xcd);
    if (tile >= a.ntiles) return;                      // padding workgroup of the last band (workgroup-uniform, before any barrier)
    tby = tile / a.ntx; tbx = tile - tby * a.ntx;
  }
  const int x0 = tbx * WF_TW, y0 = tby * WF_TH;
  const int ww = WF_TW + k, wh = WF_TH + k;          // wd region
  const int ew = WF_TW + k - 1, eh = WF_TH + k - 1;  // e2 region
  const int ewp = ew + ((2 - ew) & 3);               // e2 row pitch in eye pairs, = 2 mod 4: rows stay 16-byte aligned and consecutive rows
                                                     // are 4 banks (mod 8) apart, which makes the 16-byte window reads of phase C conflict-free
  // LDS aliasing (floats): wd2 [0, 2*wh*ww) is dead after phase B and becomes bb2 [0, 2*TH*TW) (the blend weights); e2_2 follows wd2
  // and is dead after phase C; bb2 is consumed into registers (the weights of the wave's four rows) before the pre-interpolated RGB
  // rows Hh land over the whole region.  Live maximum at 4K / k = 9: max(wd2 + e2_2 = 45.9 KB, Hh = 47.3 KB) + 3.8 KB of tables = 51 KB
  // => 3 workgroups per CU.
  vd_f2* wd = reinterpret_cast<vd_f2*>(lds);                       // [wh*ww]   (later hs [eh*WF_TW])
  vd_f2* e2 = reinterpret_cast<vd_f2*>(lds + a.e2_off);            // [eh][ewp], 16 B aligned
  float* Hh = lds;                                                  // [3][er_max][nch]
  float* rowD = lds + a.tab_off;                                    // [WF_TH][WF_RD], 16 B aligned (host: after the aliased buffers)
  float* rowA = rowD + WF_TH * WF_RD;                               // [wh][4]        (phase A only)
  float* colT = rowA;                                               // [nch][2]       (built after phase C: rowA is dead by then)
  const int tid = threadIdx.x;
  VD_STAMP(wf_stamps, 0, false);
  VD_OCC_IN(wf_occ);
  const int lane = tid & 63, wv = wf_uni(tid >> 6);
  // The shift values of this wave's phase-D rows are requested NOW (consumed after phase C): their maximum over the tile decides which 64-column
  // chunks of the pre-interpolated rows Hh phase D1 has to build at all (round 4) -- the LDS holds the worst case the parameters allow (+- 79 px at
  // 4K), a tile of an ordinary frame samples +- 10 .. 20 px around itself.
  __shared__ __attribute__((aligned(16))) unsigned smax_w[8];      // one slot per wave (round 6: no zeroing, no atomics -> no barrier of its own).  32 bytes: the dynamic LDS
                                                                   // behind it must stay 16-byte aligned (ds_read_b128 everywhere; a 4-byte static variable in front of it
                                                                   // cost 3x the kernel time: misaligned 16-byte LDS accesses)
  float sD[WF_TH / WF_NW];
  if (!SHIFT) {
#pragma unroll
    for (int j = 0; j < WF_TH / WF_NW; ++j) {
      const int y = y0 + wv + j * WF_NW, x = x0 + lane;
      sD[j] = (y < H && x < W) ? S[(unsigned)y * (unsigned)W + (unsigned)x] : 0.f;
    }
  }
  const int wy0 = y0 - r - 1, wx0 = x0 - r - 1;
  const int cb = max(x0 - a.bound, 0);                              // first warp-res column of Hh / colT
  const int nch = a.nch;
  const unsigned ni = (unsigned)a.ih * (unsigned)a.iw;

  int er0 = 0;
  if (RESIZE) er0 = wf_tap(a.ih, H, a.scale_h, max(y0 - 1, 0)).i0;  // first eye-res row the tile touches
  // ---- tables
  if (FEATHER && !PRE && tid < wh) {   // phase-A rows
    const int y = wy0 + tid;
    int yn = 0; float n = 0.f, sr = 1.f; bool s_ok = false;
    if (y >= 0 && y < H) wf_gs_row(vd_lin11_step(a.step_y, H, y), H, &yn, &n, &sr, &s_ok);
    float* t = rowA + tid * 4;
    t[0] = __int_as_float(yn); t[1] = n; t[2] = sr; t[3] = __int_as_float((s_ok && n != 0.f) ? 1 : 0);
  }
  if (((FEATHER && !PRE) || SHIFT) && tid >= WF_NT - 64) rs14[tid - (WF_NT - 64)] = c_vd_rs14[tid - (WF_NT - 64)];
  if (FEATHER && PRE) {
    // the e2 region of the tile straight from the E2 plane (zero outside the image = avg_pool2d's padding); all loads of a thread in flight.  Round 6: straight-line
    // (indices and coordinates clamped, the value zeroed afterwards) -- with the loads under `if (inside)` hipcc put a `s_waitcnt vmcnt(0)` at the first join, i.e. the S
    // loads issued at the top of the kernel were waited for BEFORE the first E2 load went out: two dependent round trips where one was meant
    constexpr int NLD = 6;
    for (int t0 = tid; t0 < eh * ew; t0 += NLD * WF_NT) {
      vd_f2 ev[NLD]; int dst[NLD];
#pragma unroll
      for (int j = 0; j < NLD; ++j) {
        const int t = t0 + j * WF_NT;
        const int tc = min(t, eh * ew - 1);
        const int ty = wf_div(tc, a.m_ew), tx = tc - ty * ew;
        const int y = y0 - r + ty, x = x0 - r + tx;
        const bool in = y >= 0 && y < H && x >= 0 && x < W;
        const vd_f2 v = E2[(unsigned)min(max(y, 0), H - 1) * (unsigned)W + (unsigned)min(max(x, 0), W - 1)];
        ev[j] = vd_f2{in ? v.x : 0.f, in ? v.y : 0.f};
        dst[j] = t < eh * ew ? ty * ewp + tx : -1;
      }
#pragma unroll
      for (int j = 0; j < NLD; ++j) if (dst[j] >= 0) e2[dst[j]] = ev[j];
    }
  }
  if (FEATHER) __syncthreads();   // (without feathering nothing was written yet: the kernel's first barrier is the one behind the tile maximum below)
  VD_STAMP(wf_stamps, 1, false);
  if (FEATHER && !PRE) {
    // phase A, main block: wave = halo row (scalar row part), lane = the first 64 halo columns (gx once per lane).  The phase is
    // LATENCY-bound (two dependent global round trips per position: S, then the D gathers), so WF_AB rows are walked together: all
    // their S loads are issued first, then all their D gathers, then the combines (measured: 12 -> 4 exposed round trips per wave).
    {
      const int x = wx0 + lane;
      const bool xin = x >= 0 && x < W;
      const int xc = min(max(x, 0), W - 1);              // lanes left / right of the image load a valid column and are zeroed below
      const float gx = vd_lin11_step(a.step_x, W, xc);
      for (int tb = wv; tb < wh; tb += WF_AB * WF_NW) {
        float sv[WF_AB];
#pragma unroll
        for (int j = 0; j < WF_AB; ++j) {
          const int ty = tb + j * WF_NW, y = wy0 + ty;
          sv[j] = (ty < wh && y >= 0 && y < H) ? S[(unsigned)y * (unsigned)W + (unsigned)xc] : 0.f;   // wave-uniform guard
        }
        wf_wdl ld[WF_AB];
        float rn[WF_AB], rsr[WF_AB]; bool rso[WF_AB], rok[WF_AB];
#pragma unroll
        for (int j = 0; j < WF_AB; ++j) {
          const int ty = tb + j * WF_NW, y = wy0 + ty;
          rok[j] = ty < wh && y >= 0 && y < H;   // wave-uniform
          rn[j] = 0.f; rsr[j] = 1.f; rso[j] = false;
          ld[j].w = ld[j].vnw = ld[j].vne = ld[j].vsw = ld[j].vse = vd_f2{0.f, 0.f}; ld[j].e0 = ld[j].e1 = false;
          if (rok[j]) {
            const vd_f4 rt = *reinterpret_cast<const vd_f4*>(rowA + ty * 4);
            const int yn = wf_uni(__float_as_int(rt.x));
            rn[j] = wf_unif(rt.y); rsr[j] = wf_unif(rt.z);
            rso[j] = wf_uni(__float_as_int(rt.w)) != 0;
            ld[j] = wf_wd_issue(D + (unsigned)yn * (unsigned)W, sv[j], gx, rso[j], W);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        vd_f2 vv[WF_AB];
#pragma unroll
        for (int j = 0; j < WF_AB; ++j) {   // rows outside the image hold zero-filled samples -> 0
          const vd_f2 v = wf_wd_combine(ld[j], rn[j], rsr[j]);
          vv[j].x = xin ? v.x : 0.f; vv[j].y = xin ? v.y : 0.f;
        }
#pragma unroll
        for (int j = 0; j < WF_AB; ++j) {
          const int ty = tb + j * WF_NW;
          if (ty < wh) wd[ty * ww + lane] = vv[j];
        }
      }
    }
    // phase A, leftover block: the k halo columns right of the first 64, transposed (wave = column, lane = halo row)
    for (int c = wv; c < ww - 64; c += WF_NW) {
      const int tx = 64 + c, x = wx0 + tx;          // wave-uniform
      const bool xin = x >= 0 && x < W;
      const float gx = vd_lin11_step(a.step_x, W, min(max(x, 0), W - 1));
      for (int ty = lane; ty < wh; ty += 64) {
        const int y = wy0 + ty;
        vd_f2 v = {0.f, 0.f};
        if (xin && y >= 0 && y < H) {
          const vd_f4 rt = *reinterpret_cast<const vd_f4*>(rowA + ty * 4);
          const int yn = __float_as_int(rt.x);
          v = wf_warped_depth(D + (unsigned)yn * (unsigned)W, S[(unsigned)y * (unsigned)W + (unsigned)x], gx, rt.y, rt.z,
                              __float_as_int(rt.w) != 0, W);
        }
        wd[ty * ww + tx] = v;
      }
    }
  }
  if (FEATHER && !PRE) __syncthreads();
  VD_STAMP(wf_stamps, 2, false);
  if (FEATHER && !PRE) {
    // phase B: e2 = clamp(|grad WD| * fs, 0, 1) (:347-352), zero outside the image (avg_pool2d zero padding)
    const int bq = WF_NT / ew, br = WF_NT - bq * ew;
    int ty = wf_div(tid, a.m_ew), tx = tid - ty * ew;
    for (int t = tid; t < eh * ew; t += WF_NT) {
      const int y = y0 - r + ty, x = x0 - r + tx;
      vd_f2 e = {0.f, 0.f};
      if (y >= 0 && y < H && x >= 0 && x < W) {
        const vd_f2* wvp = wd + __umul24((unsigned)(ty + 1), (unsigned)ww) + (tx + 1);
        const vd_f2 c = wvp[0];
        const vd_f2 z = {0.f, 0.f};
        const vd_f2 gx = x > 0 ? c - wvp[-1] : z;
        const vd_f2 gy = y > 0 ? c - wvp[-ww] : z;
        const vd_f2 q = gx * gx + gy * gy;
        const vd_f2 m = vd_f2{wf_sqrt_torch(q.x, rs14), wf_sqrt_torch(q.y, rs14)} * a.fs;   // torch.sqrt = MKL vsSqrt, not the rounded root
        e.x = vd_clamp_fin(m.x, 0.f, 1.f); e.y = vd_clamp_fin(m.y, 0.f, 1.f);
      }
      e2[ty * ewp + tx] = e;
      ty += bq; tx += br;
      if (tx >= ew) { tx -= ew; ++ty; }
    }
    __syncthreads();
  }
  VD_STAMP(wf_stamps, 3, false);
  if (FEATHER) {
    // phase C: blend weights b (see the header) into the dead wd buffer; thread = (tile row, strip of 4 pixels), both eyes packed
    {
      vd_f2* bb = wd;   // [WF_TH][WF_TW]
      const int ty = tid >> 4, tx0 = (tid & 15) * 4;
      vd_f2 s0 = {0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
      if (k == 9) {
#pragma unroll
        for (int i = 0; i < 9; ++i) {
          const vd_f4* row = reinterpret_cast<const vd_f4*>(e2 + (ty + i) * ewp + tx0);
          vd_f2 wn[12];
#pragma unroll
          for (int h = 0; h < 6; ++h) { const vd_f4 v = row[h]; wn[2 * h] = vd_f2{v.x, v.y}; wn[2 * h + 1] = vd_f2{v.z, v.w}; }
#pragma unroll
          for (int j = 0; j < 9; ++j) { s0 += wn[j]; s1 += wn[j + 1]; s2 += wn[j + 2]; s3 += wn[j + 3]; }
        }
      } else {
        for (int i = 0; i < k; ++i) {
          const vd_f2* row = e2 + (ty + i) * ewp + tx0;
          vd_f2 w0 = row[0], w1 = row[1], w2 = row[2];
          for (int j = 0; j < k; ++j) {
            const vd_f2 w3 = row[j + 3];
            s0 += w0; s1 += w1; s2 += w2; s3 += w3;
            w0 = w1; w1 = w2; w2 = w3;
          }
        }
      }
      vd_f2 sv[4] = {s0, s1, s2, s3};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (a.fastdiv) {   // correctly rounded s / (k*k) in three packed operations (tools/verify_fastdiv.c: exhaustive)
          const vd_f2 q0 = sv[q] * a.rc_kk;
          const vd_f2 rr = vd_vfma(-q0, (vd_f2)(a.kk), sv[q]);
          sv[q] = vd_vfma(rr, (vd_f2)(a.rc_kk), q0);
        } else {
          sv[q].x = sv[q].x / a.kk; sv[q].y = sv[q].y / a.kk;
        }
      }
      vd_f4* dst = reinterpret_cast<vd_f4*>(bb + ty * WF_TW + tx0);
      dst[0] = vd_f4{sv[0].x, sv[0].y, sv[1].x, sv[1].y};
      dst[1] = vd_f4{sv[2].x, sv[2].y, sv[3].x, sv[3].y};
    }
  }
  if (SHIFT) {
    // ---- the shift values of the tile (core/render_3d.py:620-680 + suppress_artifacts_with_edge_mask :198-216), as vd3d_planes.hip:k_shift computes them.  Three
    // short phases in LDS that the Hh rows overwrite later: dT = the depth tile + halo (3 left / up, 2 right / down: the 5 x 5 pool's window + the gradient's
    // neighbours), em = 1 - sigmoid(...) of the gradient magnitude on the pool's window (zero outside the image = avg_pool2d's padding), sS = the shift values.
    constexpr int DT_W = WF_TW + 8, DT_H = WF_TH + 5, EM_W = WF_TW + 4, EM_H = WF_TH + 4;   // dT pitch 72 floats, em pitch 68 (= k_shift's: 16-byte rows)
    float* dT = lds;                                  // [DT_H][DT_W]: dT[ty][tx] = D[y0 - 3 + ty][x0 - 3 + tx]
    float* em = lds + ((DT_H * DT_W + 3) & ~3);       // [EM_H][EM_W]: em[ty][tx] at (y0 - 2 + ty, x0 - 2 + tx)
    float* sS = em + EM_H * EM_W;                     // [WF_TH][WF_TW]
    {
      constexpr int NE = DT_H * (WF_TW + 5), NL = (NE + WF_NT - 1) / WF_NT;
      float dv[NL];
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        const int t = min(tid + j * WF_NT, NE - 1);
        const int ty = t / (WF_TW + 5), tx = t - ty * (WF_TW + 5);
        const int y = y0 - 3 + ty, x = x0 - 3 + tx;
        const bool in = y >= 0 && y < H && x >= 0 && x < W;
        const float v = D[(unsigned)min(max(y, 0), H - 1) * (unsigned)W + (unsigned)min(max(x, 0), W - 1)];
        dv[j] = in ? v : 0.f;
      }
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        const int t = tid + j * WF_NT;
        if (t < NE) { const int ty = t / (WF_TW + 5), tx = t - ty * (WF_TW + 5); dT[ty * DT_W + tx] = dv[j]; }
      }
    }
    __syncthreads();
    if (sc.edge) {
      for (int t = tid; t < EM_H * EM_W; t += WF_NT) {
        const int ty = t / EM_W, tx = t - ty * EM_W;
        const int y = y0 - 2 + ty, x = x0 - 2 + tx;
        float e = 0.f;  // zero padding of avg_pool2d
        if (y >= 0 && y < H && x >= 0 && x < W) {
          const float* dp = dT + (ty + 1) * DT_W + (tx + 1);
          const float cc = dp[0];
          const float dx = x > 0 ? fabsf(cc - dp[-1]) : 0.f;
          const float dy = y > 0 ? fabsf(cc - dp[-DT_W]) : 0.f;
          const float g = vd_sqrt_torch(dx * dx + dy * dy, rs14);
          const float z = ((g - (float)0.02) * sc.fs) * 5.f;
          e = 1.f - vd_sigmoid_torch(z);
        }
        em[t] = e;
      }
      __syncthreads();
    }
    {
      const float fgf = work->fg, mgf = work->mg, bgf = work->bg;
      // one thread = 4 consecutive pixels of one row (WF_TH * WF_TW / 4 == WF_NT threads)
      const int ty = tid / (WF_TW / 4), tx = (tid - ty * (WF_TW / 4)) * 4;
      const int y = y0 + ty;
      vd_f4 s5 = {0.f, 0.f, 0.f, 0.f};
      if (sc.edge) {
        // avg_pool2d(5, 1, 2) in ATen's order (cpu_avg_pool2d): ONE float32 running sum over the window, row-major; the zero padding adds exact zeros
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          const vd_f4 wa = *reinterpret_cast<const vd_f4*>(em + (ty + i) * EM_W + tx), wb = *reinterpret_cast<const vd_f4*>(em + (ty + i) * EM_W + tx + 4);
          const float win[8] = {wa[0], wa[1], wa[2], wa[3], wb[0], wb[1], wb[2], wb[3]};
#pragma unroll
          for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) s5[q] += win[q + j];
        }
      }
      vd_f4 so = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int x = x0 + tx + q;
        if (x < W && y < H) {
          const float Dv = dT[(ty + 3) * DT_W + (tx + q + 3)];
          const float p15 = vd_pow15_torch(1.0f - Dv);
          const float fgw = vd_clamp(p15, 0.f, 1.f);
          const float mgw = vd_clamp(1.0f - fabsf(Dv - sc.mid) * 3.0f, 0.f, 1.f);
          const float bgw = vd_clamp(Dv, 0.f, 1.f);
          const float raw = ((fgw * fgf) * sc.fgm + mgw * mgf) + (bgw * bgf) * sc.bgm;
          float sft = (raw * sc.pb) / sc.half_width;
          if (work->have_zpo) sft = sft - work->zpo_f;
          sft = vd_clamp(sft, -work->msn, work->msn);
          if (work->have_conv) sft = sft - work->conv;
          if (sc.edge) {
            const float sm = s5[q] / 25.f;
            sft = sc.ma * sft + sc.mb * (sft * sm);
          }
          so[q] = sft;
          if (S) S[(unsigned)y * (unsigned)W + (unsigned)x] = sft;
        }
      }
      *reinterpret_cast<vd_f4*>(sS + ty * WF_TW + tx) = so;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < WF_TH / WF_NW; ++j) sD[j] = sS[(wv + j * WF_NW) * WF_TW + lane];   // zero outside the frame, like the plane loads of the other variants
  }
  {   // tile maximum of |S| (non-negative floats order like their bit patterns; a NaN would sort above everything -> the full range is built)
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < WF_TH / WF_NW; ++j) m = fmaxf(m, fabsf(sD[j]));
    unsigned mb = __float_as_uint(m);
    for (int off = 32; off > 0; off >>= 1) mb = max(mb, (unsigned)__shfl_xor((int)mb, off, 64));
    if (lane == 0) smax_w[wv] = mb;
  }
  auto build_colT = [&]() {   // column taps of the warp-res columns cb .. cb + nch - 1: (absolute eye-res column i0, weight of i0 + 1)
    for (int j = tid; j < nch; j += WF_NT) {
      const vd_tap t = wf_tap(a.iw, W, a.scale_w, min(cb + j, W - 1));
      colT[2 * j] = __int_as_float(t.i0); colT[2 * j + 1] = t.w1;
    }
  };
  if (RESIZE && !FEATHER) build_colT();   // no blend-weight buffer to wait for: the column taps share the barrier of the tile maximum (round 6: five barriers -> two)
  __syncthreads();
  VD_STAMP(wf_stamps, 4, false);
  // phase D0: the feather weights b of the wave's rows into registers -- bb2 dies here
  vd_f2 bD[WF_TH / WF_NW];
#pragma unroll
  for (int j = 0; j < WF_TH / WF_NW; ++j) {
    vd_f2 b = {0.f, 0.f};
    if (FEATHER) b = wd[(wv + j * WF_NW) * WF_TW + lane];
    bD[j] = b;
  }
  if (RESIZE) {
    if (FEATHER) {
      build_colT();
      __syncthreads();   // bb2 fully consumed, colT complete
    }
    VD_STAMP(wf_stamps, 5, false);
    // phase D1: Hh[c][r][X] = the HORIZONTAL half of the resize of :595 for warp-res column cb + X and eye-res row er0 + r:
    //   fma(p[i0], 1 - w1, w1 * p[i0 + 1])  -- exactly the first two operations of ATen's bilinear (rows, then columns), computed
    // once and shared by every sample that needs it (~4.4 per element) instead of inside each of them.  One wave = one 64-column
    // chunk (its column taps live in two registers), WF_HB rows of loads in flight.
    {
      // columns of Hh this tile can sample: its own 64 plus the tile's largest shift in pixels (same margin rule as the host's `bound`) plus the east
      // neighbour of the bilinear pair; chunks outside are never read and are not built
      unsigned smb = 0u;
#pragma unroll
      for (int j = 0; j < WF_NW; ++j) smb = max(smb, smax_w[j]);
      const float smax = __uint_as_float((unsigned)wf_uni((int)smb));   // wave-uniform: the chunk / row assignment below stays scalar
      int bt = a.bound;
      if (smax < 1.0f) bt = min(a.bound, (int)ceilf(smax * ((float)(W - 1) * 0.5f) * 1.0001f) + 3);
      if (!FEATHER) {
      // Round 6, the kernel without feathering: the elements (channel, eye-res row, column) are dealt FLAT over the workgroup's threads, WF_HB per thread and batch, every
      // load of a batch in flight before the first is used, in straight-line code.  The per-chunk form below (one wave = one cb-aligned 64-column chunk, (channel, row)
      // pieces walked in wave-uniform loops) stood at 9 000 of a workgroup's 18 600 cycles at 4K in the phase stamps, and stayed there when its loads were a quarter as
      // many (one 64-lane row segment + ds_bpermute, tools/r06/w1_d1_segments.patch), when they all hit the caches and -- 5 800 cycles -- when there were none: the
      // time was the scalar loop control, the waits hipcc puts at the joins of wave-uniform branches and a 3 : 3 : 2 split of the waves over three chunks, not memory.
      // Measured at 4K (us per launch, no feathering / feathering): per-chunk 57.1 / 89.4, flat 52.2 / 92.6, (channel, row, 64-column group) items dealt round-robin
      // over the waves 56.3 / 94.8 -- the flat form pays ~25 VALU per element for its index arithmetic, which the feathered kernel's window sums compete for: each
      // instantiation keeps the form that is faster for it.
      const int X_lo = max(x0 - bt - 1 - cb, 0), X_hi = min(x0 + WF_TW + bt + 1 - cb, nch - 1);   // first / last Hh column this tile can sample
      const int ncol = X_hi - X_lo + 1, er = a.er_max;
      const uint32_t mcol = 0xFFFFFFFFu / (uint32_t)ncol + 1u;      // ceil(2^32 / ncol): umulhi(e, mcol) == e / ncol for e, ncol < 2^16
      const int total = 3 * er * ncol;
      const vd_f2* colT2 = reinterpret_cast<const vd_f2*>(colT);
#pragma unroll 1
      for (int e0 = tid; e0 < total; e0 += WF_HB * WF_NT) {
        float p0[WF_HB], p1[WF_HB], w1v[WF_HB]; int dst[WF_HB];
#pragma unroll
        for (int j = 0; j < WF_HB; ++j) {
          const int e = e0 + j * WF_NT;
          const int ee = min(e, total - 1);                          // past the end: the last element again (loaded, not stored)
          const int pc = wf_div(ee, mcol), X = X_lo + (ee - pc * ncol);
          const vd_f2 ct = colT2[X];
          const int i0 = __float_as_int(ct.x), i1 = min(i0 + 1, a.iw - 1);
          const int c = (pc >= er ? 1 : 0) + (pc >= 2 * er ? 1 : 0), rr = pc - c * er;
          const float* srow = rgb + ((unsigned)c * ni + (unsigned)min(er0 + rr, a.ih - 1) * (unsigned)a.iw);
          p0[j] = srow[i0]; p1[j] = srow[i1];
          w1v[j] = ct.y;
          dst[j] = e < total ? pc * nch + X : -1;
        }
#pragma unroll
        for (int j = 0; j < WF_HB; ++j)
          if (dst[j] >= 0) Hh[dst[j]] = vd_fma(p0[j], 1.f - w1v[j], w1v[j] * p1[j]);
      }
      } else {
      const int c_lo = max(x0 - bt - 1 - cb, 0) >> 6, c_hi = min(x0 + WF_TW + bt + 1 - cb, nch - 1) >> 6;
      const int nchk = c_hi - c_lo + 1;                       // 1 .. (nch + 63) / 64 chunks to build (<= WF_NW: host check)
      const int cw = wv % nchk, chunk = c_lo + cw, rstart = wv / nchk, rstep = (WF_NW - cw + nchk - 1) / nchk;
      const int X = chunk * 64 + lane;
      const bool xok = X < nch;
      const vd_f2 ct = xok ? reinterpret_cast<const vd_f2*>(colT)[X] : vd_f2{0.f, 0.f};
      const int i0 = __float_as_int(ct.x), i1 = min(i0 + 1, a.iw - 1);
      const float w1 = ct.y, w0 = 1.f - ct.y;
      const int er = a.er_max;
      int c = 0, rr = rstart;
      while (rr >= er) { rr -= er; ++c; }
      while (c < 3) {
        float p0[WF_HBC], p1[WF_HBC]; int dst[WF_HBC];
#pragma unroll
        for (int j = 0; j < WF_HBC; ++j) {
          dst[j] = -1; p0[j] = 0.f; p1[j] = 0.f;
          if (c < 3) {   // wave-uniform
            const float* srow = rgb + ((unsigned)c * ni + (unsigned)min(er0 + rr, a.ih - 1) * (unsigned)a.iw);
            if (xok) { p0[j] = srow[i0]; p1[j] = srow[i1]; }
            dst[j] = (c * er + rr) * nch;
            rr += rstep;
            while (rr >= er && c < 3) { rr -= er; ++c; }
          }
        }
#pragma unroll
        for (int j = 0; j < WF_HBC; ++j)
          if (dst[j] >= 0 && xok) Hh[dst[j] + X] = vd_fma(p0[j], w0, w1 * p1[j]);
      }
      }
    }
  }
  __syncthreads();
  VD_STAMP(wf_stamps, 6, false);
  // phase D2: one wave = 64 consecutive pixels of ONE row per iteration, so everything that depends on y only (sample rows yn / yn+1,
  // their resize taps, the vertical weights) is wave-uniform and comes from rowD.  ~70 % of rows have an exactly integral sample row
  // (n == 0): there sw = se = 0 and the two south samples contribute exactly +0 -> skipped (bit-exact: fma(v, 0, acc) == acc).
  int jD = 0;
  // (the table row of the NEXT iteration is requested a row ahead: an s_load's round trip would otherwise be exposed at the top of each of a wave's four rows)
  const vd_f4* rt0 = reinterpret_cast<const vd_f4*>(rowtab + (size_t)(unsigned)min(y0 + wv, H - 1) * WF_RD);
  vd_f4 ro_n = rt0[0], ra_n = rt0[1], rb_n = rt0[2], rs_n = rt0[3];
#pragma unroll 1
  for (int ty = wv; ty < WF_TH; ty += WF_NW, ++jD) {
    const int y = y0 + ty;
    if (y >= H) break;
    const float s_row = sD[0];   // rotate the prefetched per-row values (static register indices, compact loop body)
    const vd_f2 b = bD[0];
#pragma unroll
    for (int j = 0; j + 1 < WF_TH / WF_NW; ++j) { sD[j] = sD[j + 1]; bD[j] = bD[j + 1]; }
    // row parameters: 16 floats per frame row from the per-geometry table k_wf_rowtab wrote (round 6) -- a wave-uniform address, so they arrive by s_load in
    // SGPRs; the LDS table this replaces cost 3 ds_read_b128 + 11 v_readfirstlane per row of a kernel that is bound by its VALU issue
    const vd_f4 ro = ro_n, ra = ra_n, rb = rb_n, rs = rs_n;
    {
      const vd_f4* rt = reinterpret_cast<const vd_f4*>(rowtab + (size_t)(unsigned)min(y + WF_NW, H - 1) * WF_RD);
      ro_n = rt[0]; ra_n = rt[1]; rb_n = rt[2]; rs_n = rt[3];
    }
    const float n = rs.x, sr = rs.y;
    const bool south = __float_as_int(rs.z) != 0;
    const int yn = __float_as_int(rs.w);
    const int x = x0 + lane;
    uint32_t pL = 0, pR = 0;
    if (x < W) {
      const unsigned o = (unsigned)y * (unsigned)W + (unsigned)x;
      const float s = s_row;
      const float gx0 = vd_lin11_step(a.step_x, W, x);
      const wf_gs2 g = wf_gs_params2(gx0, s, n, sr, W);
      const vd_f2 omb = 1.0f - b;
      if (RESIZE) {
        typedef const __attribute__((address_space(3))) char* wf_lds_cp;
        typedef const __attribute__((address_space(3))) float* wf_lds_fp;
        const wf_lds_cp hb = (wf_lds_cp)Hh;
        // byte offsets: column part per lane, row / channel part wave-uniform
        const int cs = a.er_max * nch * 4;                            // channel stride
        const int xo = (x - cb) * 4, xl = (g.xw[0] - cb) * 4, xr = (g.xw[1] - cb) * 4;
        const int rp = nch * 4;                                       // bytes per Hh row; the table holds absolute eye-res rows
        const int o_r0 = (__float_as_int(ro.x) - er0) * rp, o_r1 = (__float_as_int(ro.y) - er0) * rp;
        const int a_r0 = (__float_as_int(ra.x) - er0) * rp, a_r1 = (__float_as_int(ra.y) - er0) * rp;
        const int b_r0 = (__float_as_int(rb.x) - er0) * rp, b_r1 = (__float_as_int(rb.y) - er0) * rp;
        const float o_w0 = ro.z, o_w1 = ro.w, a_w0 = ra.z, a_w1 = ra.w;
        const float b_w0 = rb.z, b_w1 = rb.w;
        // Round 6: (i) the horizontal half of grid_sample per EYE on the (west, east) register pair a ds_read2_b32 delivers, in scalar v_mul / v_fma -- the packed form
        // wanted (left, right) pairs and paid a v_mov + v_cndmask per element to transpose; the east guards are gone because an eye whose east neighbour is outside the
        // image has ix == W - 1 exactly, i.e. ne == se == +0, and fma(finite, +0, p) == fma(0, +0, p) for the non-negative p here.  (ii) All twelve reads of a row's
        // north samples (three channels x two eyes x two resize rows) are issued before the first is used, then the twelve south ones under ONE branch: a wave used to
        // walk channel by channel, six to ten dependent LDS round trips per row, and with six waves per SIMD those round trips -- not the instruction count -- were
        // what a row cost (removing 14 VALU per row by moving the row table to SGPRs changed nothing; removing the waits did).
        auto ld2 = [&](int off) { return vd_f2{*(wf_lds_fp)(hb + off), *(wf_lds_fp)(hb + off + 4)}; };
        vd_f2 v[3];
        {
          vd_f2 a0L[3], a1L[3], a0R[3], a1R[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            a0L[c] = ld2(c * cs + a_r0 + xl); a1L[c] = ld2(c * cs + a_r1 + xl);
            a0R[c] = ld2(c * cs + a_r0 + xr); a1R[c] = ld2(c * cs + a_r1 + xr);
          }
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const vd_f2 nL = vd_vfma(a0L[c], (vd_f2)(a_w0), a_w1 * a1L[c]), nR = vd_vfma(a0R[c], (vd_f2)(a_w0), a_w1 * a1R[c]);
            v[c] = vd_f2{vd_fma(nL.y, g.ne.x, nL.x * g.nw.x), vd_fma(nR.y, g.ne.y, nR.x * g.nw.y)};
          }
        }
        if (south) {
          vd_f2 b0L[3], b1L[3], b0R[3], b1R[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            b0L[c] = ld2(c * cs + b_r0 + xl); b1L[c] = ld2(c * cs + b_r1 + xl);
            b0R[c] = ld2(c * cs + b_r0 + xr); b1R[c] = ld2(c * cs + b_r1 + xr);
          }
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const vd_f2 sL = vd_vfma(b0L[c], (vd_f2)(b_w0), b_w1 * b1L[c]), sR = vd_vfma(b0R[c], (vd_f2)(b_w0), b_w1 * b1R[c]);
            v[c].x = vd_fma(sL.y, g.se.x, vd_fma(sL.x, g.sw.x, v[c].x));
            v[c].y = vd_fma(sR.y, g.se.y, vd_fma(sR.x, g.sw.y, v[c].y));
          }
        }
        if (FEATHER) {
          float o0[3], o1[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) { o0[c] = *(wf_lds_fp)(hb + c * cs + o_r0 + xo); o1[c] = *(wf_lds_fp)(hb + c * cs + o_r1 + xo); }
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float orig = vd_fma(o0[c], o_w0, o_w1 * o1[c]);
            v[c] = v[c] * omb + orig * b; v[c].x = vd_clamp_fin(v[c].x, 0.f, 1.f); v[c].y = vd_clamp_fin(v[c].y, 0.f, 1.f);
          }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const vd_f2 u = v[c] * 255.0f;
          pL |= (uint32_t)(uint8_t)u.x << (8 * (3 - c));
          pR |= (uint32_t)(uint8_t)u.y << (8 * (3 - c));
        }
      } else
      {
        // eyes at warp resolution: the samples are direct global gathers.  Round 6: the twelve north gathers (three channels x two eyes x west / east) are issued
        // before the first is used, the twelve south ones under one branch, none of them guarded -- the east column is xw + 1 where that is inside the image and xw
        // itself where it is not (there ne == se == +0: the finite sample adds nothing).  The guarded, channel-by-channel form compiled to `s_waitcnt vmcnt(0)` behind
        // every channel's north and south group: up to six dependent GLOBAL round trips per row, four rows per wave.
        const int xe0 = g.xw[0] + (g.e_ok[0] ? 1 : 0), xe1 = g.xw[1] + (g.e_ok[1] ? 1 : 0);
        const unsigned rowo = (unsigned)yn * (unsigned)W;
        float orig[3] = {0.f, 0.f, 0.f};
        vd_f2 v[3];
        {
          vd_f2 vnw[3], vne[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float* r0 = rgb + (unsigned)c * ni + rowo;
            vnw[c] = vd_f2{r0[g.xw[0]], r0[g.xw[1]]};
            vne[c] = vd_f2{r0[xe0], r0[xe1]};
            if (FEATHER) orig[c] = rgb[(unsigned)c * ni + o];
          }
#pragma unroll
          for (int c = 0; c < 3; ++c) v[c] = vd_vfma(vne[c], g.ne, vnw[c] * g.nw);
        }
        if (south) {
          vd_f2 vsw[3], vse[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float* r1 = rgb + (unsigned)c * ni + rowo + (unsigned)W;
            vsw[c] = vd_f2{r1[g.xw[0]], r1[g.xw[1]]};
            vse[c] = vd_f2{r1[xe0], r1[xe1]};
          }
#pragma unroll
          for (int c = 0; c < 3; ++c) v[c] = vd_vfma(vse[c], g.se, vd_vfma(vsw[c], g.sw, v[c]));
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          if (FEATHER) { v[c] = v[c] * omb + orig[c] * b; v[c].x = vd_clamp_fin(v[c].x, 0.f, 1.f); v[c].y = vd_clamp_fin(v[c].y, 0.f, 1.f); }
          const vd_f2 u = v[c] * 255.0f;
          pL |= (uint32_t)(uint8_t)u.x << (8 * (3 - c));
          pR |= (uint32_t)(uint8_t)u.y << (8 * (3 - c));
        }
      }
    }
    // pack 4 lanes x 3 bytes into 3 dwords (lanes 4j, 4j+1, 4j+2 store).  pX = (B | G<<8 | R<<16) << 8: with the pixel in the upper three bytes, dword q of the
    // 12-byte group is ONE v_alignbit of (east neighbour's pixel : own pixel) by 8q + 8 bits; the neighbour comes by DPP row_shl:1 (a lane with q < 3 never
    // sits at the end of its row of 16) -- round 6, was two ds_bpermute and a three-way divergent select
    const uint32_t nL = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pL, 0x101, 0xf, 0xf, true) >> 8;
    const uint32_t nR = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pR, 0x101, 0xf, 0xf, true) >> 8;
    const int q = lane & 3;
    const int xq = x0 + (lane & ~3);
    const unsigned ob = ((unsigned)y * (unsigned)W + (unsigned)xq) * 3u;
    const bool full = (xq + 3 < W) && (ob % 4u == 0);
    if (full) {
      if (q < 3) {
        const uint32_t sh = 8u * (uint32_t)q + 8u;
        reinterpret_cast<uint32_t*>(L + ob)[q] = __builtin_amdgcn_alignbit(nL, pL, sh);
        reinterpret_cast<uint32_t*>(R + ob)[q] = __builtin_amdgcn_alignbit(nR, pR, sh);
      }
    } else if (x < W) {
      const unsigned o1 = ((unsigned)y * (unsigned)W + (unsigned)x) * 3u;
      uint8_t* ol = L + o1;
      uint8_t* orr = R + o1;
      ol[0] = (uint8_t)(pL >> 8); ol[1] = (uint8_t)(pL >> 16); ol[2] = (uint8_t)(pL >> 24);
      orr[0] = (uint8_t)(pR >> 8); orr[1] = (uint8_t)(pR >> 16); orr[2] = (uint8_t)(pR >> 24);
    }
  }
  VD_STAMP(wf_stamps, 7, true);
  VD_OCC_OUT(wf_occ);
}

// blur_ksize values whose k*k passed tools/verify_fastdiv.c (all floats in [0, k*k], 3-operation division == IEEE division)
static bool wf_fastdiv_ok(int k) {   // every float in [0, k*k] checked; inexact for k in {6, 10, 12, 14, 18, 20, 22, 24, 26, 28, 30}
  if (k < 1 || k > 33) return false;
  static const unsigned long long bad = (1ull << 6) | (1ull << 10) | (1ull << 12) | (1ull << 14) | (1ull << 18) | (1ull << 20) | (1ull << 22) |
                                        (1ull << 24) | (1ull << 26) | (1ull << 28) | (1ull << 30);
  return !((bad >> k) & 1ull);
}

// tile height of W1 with a precomputed mask (tuning probe: vd3d_debug_tune(2, 16 | 32)); without the mask phases in the tile the halo no longer
// multiplies their arithmetic, so flatter tiles (less LDS per workgroup, more workgroups per CU) become an option
// Tile order of W1 (vd3d_debug_tune(9, v)): 0 plain row-major, 1 one contiguous band of tiles per XCD (round 3), 2 tile rows round-robin over the XCDs, -1 (default) by path:
// the feathered path keeps the bands (its 9 x 9 halo re-reads stay in one L2: fabric traffic 3.3x -> 1.3x, same time in all three orders, 152 .. 155 us), the no-feather path of
// round 5 has no halo worth keeping and runs plain (4K: 66.9 us against 74 .. 75.5 in bands and 69.8 round-robin, gpurun_out/r05c21).
static int g_wf_order = -1;
void vd_set_warp_order(int v) { g_wf_order = v < -1 ? -1 : (v > 2 ? 2 : v); }
static int g_wf_pre_th = 32;
void vd_set_warp_pre_th(int th) { g_wf_pre_th = th == 16 ? 16 : 32; }
// ... and of W1 without feathering (round 5: no mask halo at all, vd3d_debug_tune(7, 16 | 32))
#ifndef WF_NOFEATHER_TH
#define WF_NOFEATHER_TH 32
#endif
static int g_wf_nf_th = WF_NOFEATHER_TH;
void vd_set_warp_nofeather_th(int th) { g_wf_nf_th = th == 16 ? 16 : 32; }

// ---- per-geometry row table of phase D2 (round 6): for every frame row y the three vertical resize taps (row y itself / the grid_sample rows yn, yn + 1) as
// absolute eye-res rows + weights, and the grid_sample row part (n, 1 - n, south flag, yn) -- the same helpers the kernel used to call per tile.
__global__ __launch_bounds__(256) void k_wf_rowtab(float* __restrict__ tab, int H, int ih, float scale_h, float step_y) {
  const int y = blockIdx.x * 256 + threadIdx.x;
  if (y >= H) return;
  int yn; float n, sr; bool s_ok;
  wf_gs_row(vd_lin11_step(step_y, H, y), H, &yn, &n, &sr, &s_ok);
  float* t = tab + (size_t)y * WF_RD;
  const vd_tap to = wf_tap(ih, H, scale_h, y), t0 = wf_tap(ih, H, scale_h, yn), t1 = wf_tap(ih, H, scale_h, min(yn + 1, H - 1));
  t[0] = __int_as_float(to.i0); t[1] = __int_as_float(to.i1); t[2] = to.w0; t[3] = to.w1;
  t[4] = __int_as_float(t0.i0); t[5] = __int_as_float(t0.i1); t[6] = t0.w0; t[7] = t0.w1;
  t[8] = __int_as_float(t1.i0); t[9] = __int_as_float(t1.i1); t[10] = t1.w0; t[11] = t1.w1;
  t[12] = n; t[13] = sr; t[14] = __int_as_float((s_ok && n != 0.f) ? 1 : 0); t[15] = __int_as_float(yn);
}
// One table per (device, H, ih): written once, on the stream of the launch that first needs it, and that stream is drained before the pointer is handed out --
// the pixel streams of a context are not ordered against each other.  Eight geometries are kept per process; the ninth replaces the oldest (hipFree waits
// for the device).  138 KB at 4K.
struct wf_rowtab_entry { int dev, H, ih; float* tab; };
static std::mutex g_wf_rt_mu;
static wf_rowtab_entry g_wf_rt[8];
static int g_wf_rt_n = 0, g_wf_rt_next = 0;
static const float* wf_rowtab_get(hipStream_t s, int H, int ih, float scale_h, float step_y) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(g_wf_rt_mu);
  for (int i = 0; i < g_wf_rt_n; ++i)
    if (g_wf_rt[i].dev == dev && g_wf_rt[i].H == H && g_wf_rt[i].ih == ih) return g_wf_rt[i].tab;
  float* tab = nullptr;
  if (hipMalloc(&tab, (size_t)H * WF_RD * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  hipLaunchKernelGGL(k_wf_rowtab, dim3((H + 255) / 256), dim3(256), 0, s, tab, H, ih, scale_h, step_y);
  if (hipStreamSynchronize(s) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(tab); return nullptr; }
  int slot;
  if (g_wf_rt_n < 8) slot = g_wf_rt_n++;
  else { slot = g_wf_rt_next++ & 7; (void)hipFree(g_wf_rt[slot].tab); }
  g_wf_rt[slot] = wf_rowtab_entry{dev, H, ih, tab};
  return tab;
}
// returns false when the fused kernel cannot be used (tiles would not fit the 160 KB LDS): caller falls back to v0.
// E2 != NULL: the gradient mask was computed by k_e2w (PRE variants); plan_only: decide, do not launch.
template <int WF_TH>
static bool warp_fused_impl(hipStream_t s, const float* rgb, int ih, int iw, const float* D, const float* S, int H, int W,
                            const vd3d_shift_params& p, uint8_t* L, uint8_t* R, const float* E2, bool plan_only, const vd_shift_fold* fold = nullptr) {
  constexpr int WF_NT = WF_TH * 16, WF_NW = WF_NT / 64;
  const bool pre = E2 != nullptr && p.enable_feathering;
  vd_wf_args a;
  a.ih = ih; a.iw = iw; a.H = H; a.W = W; a.k = p.enable_feathering ? p.blur_ksize : 1; a.feather = p.enable_feathering ? 1 : 0;
  a.fs = (float)p.feather_strength;
  a.scale_h = (float)ih / (float)H; a.scale_w = (float)iw / (float)W;
  // |final shift| <= clamp bound + |convergence| (edge-mask blend is a convex shrink); pixels = S * (W-1)/2
  const double half_width = (double)W / 2.0;
  const double smax = ((double)W * p.max_pixel_shift_percent) / half_width + fabs(p.convergence_strength) / half_width;
  a.bound = (int)ceil(smax * (double)(W - 1) / 2.0 * 1.0001) + 2;
  const bool resize = !(ih == H && iw == W);
  const int k = a.k;
  a.kk = (float)(k * k); a.rc_kk = 1.0f / a.kk; a.fastdiv = wf_fastdiv_ok(k) ? 1 : 0;
  a.er_max = 0; a.nch = 0; a.e2_off = 0;
  size_t sz_hh = 0;
  if (resize) {
    // exact number of eye-res rows a tile touches: the same float32 tap arithmetic as the kernel (wf_tap), over every tile row
    auto tap_i0 = [&](int in, int out, float scale, int o) {
      if (in == out) return o;
      float src = fmaf(scale, (float)o + 0.5f, -0.5f);
      if (src < 0.f) src = 0.f;
      int i0 = (int)floorf(src);
      return i0 > in - 1 ? in - 1 : i0;
    };
    for (int y0 = 0; y0 < H; y0 += WF_TH) {
      const int ya = y0 - 1 > 0 ? y0 - 1 : 0, yb = y0 + WF_TH < H - 1 ? y0 + WF_TH : H - 1;
      const int i0 = tap_i0(ih, H, a.scale_h, ya), j0 = tap_i0(ih, H, a.scale_h, yb);
      const int i1 = j0 + (j0 < ih - 1 ? 1 : 0);
      if (i1 - i0 + 1 > a.er_max) a.er_max = i1 - i0 + 1;
    }
    a.nch = WF_TW + 2 * a.bound + 2;
    if ((a.nch + 63) / 64 > WF_NW) return false;   // one wave per 64-column chunk of the Hh build
    sz_hh = (size_t)3 * a.er_max * a.nch;
  }
  // aliased layout (see the kernel): max(wd2 + e2_2, Hh) when feathering, else Hh alone; with a precomputed mask wd2 is only the bb2 buffer
  size_t fl = sz_hh;
  if (a.feather) {
    const int ew = WF_TW + k - 1, ewp = ew + ((2 - ew) & 3);
    const size_t sz_wd = pre ? (size_t)2 * WF_TH * WF_TW : (((size_t)2 * (WF_TH + k) * (WF_TW + k) + 3) & ~(size_t)3);
    const size_t sz_e2 = (size_t)2 * (WF_TH + k - 1) * ewp;
    a.e2_off = (int)sz_wd;
    fl = sz_wd + sz_e2 > sz_hh ? sz_wd + sz_e2 : sz_hh;
  }
  if (fold) {   // the shift phases' scratch (depth tile + halo, edge-mask tile, shift tile) lives where the Hh rows land later
    if (a.feather || WF_TH != 32) return false;
    const size_t sz_sh = (((size_t)(WF_TH + 5) * (WF_TW + 8) + 3) & ~(size_t)3) + (size_t)(WF_TH + 4) * (WF_TW + 4) + (size_t)WF_TH * WF_TW;
    if (sz_sh > fl) fl = sz_sh;
  }
  fl = (fl + 3) & ~(size_t)3;            // tables start 16 B aligned (ds_read_b128)
  a.tab_off = (int)fl;
  const size_t t_rowA = pre ? 0 : (size_t)(WF_TH + k) * 4, t_colT = (size_t)2 * a.nch;
  fl += (size_t)WF_TH * WF_RD + (t_rowA > t_colT ? t_rowA : t_colT);
  if (fl & 3) fl += 4 - (fl & 3);
  a.step_x = (1.f - (-1.f)) / (float)(W - 1); a.step_y = (1.f - (-1.f)) / (float)(H - 1);
  auto magic = [](int d) { return (uint32_t)(((1ull << 32) + (uint64_t)d - 1) / (uint64_t)(d > 0 ? d : 1)); };
  a.m_ew = magic(WF_TW + k - 1);
  if ((WF_TH + k) * (WF_TW + k) >= 65536) return false;
  if (WF_TH + k > WF_NT || 128 + WF_TH > WF_NT) return false;   // the table phase maps one thread per halo row / tile row
  if (H >= (1 << 24) || W >= (1 << 24) || (unsigned long long)H * W * 3ull >= (1ull << 32) || (unsigned long long)ih * iw * 3ull >= (1ull << 32)) return false;  // 32-bit offsets
  const size_t bytes = fl * sizeof(float);
  if (bytes > 78 * 1024) return false;  // keep >= 2 workgroups per CU (3 when <= 53 KB: 4K / k = 9 needs 51 KB)
  if (plan_only) return true;
  a.ntx = (W + WF_TW - 1) / WF_TW;
  a.ntiles = a.ntx * ((H + WF_TH - 1) / WF_TH);
  a.xcd = g_wf_order >= 0 ? g_wf_order : (a.feather ? 1 : 0);
  a.per = (a.ntiles + 7) / 8;
  const int nrows = a.ntiles / a.ntx;
  dim3 g(a.xcd == 2 ? 8 * ((nrows + 7) / 8) * a.ntx : (a.xcd ? 8 * a.per : a.ntiles));
  static bool attr[64] = {false};   // per device: the attribute belongs to the device's copy of the code object
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && !attr[dev]) {
    // dynamic LDS limit = the CU's 160 KB minus the kernel's static LDS (the VRSQRT14 table); a failed call would otherwise surface as a
    // sticky "invalid argument" at the next hipGetLastError
#define WF_ATTR(R_, F_, P_, S_)                                                                                                        \
  do {                                                                                                                             \
    hipFuncAttributes fa_;                                                                                                         \
    size_t st_ = 0;                                                                                                                \
    if (hipFuncGetAttributes(&fa_, (const void*)k_warp_fused<R_, F_, WF_TH, P_, S_>) == hipSuccess) st_ = fa_.sharedSizeBytes;         \
    if (hipFuncSetAttribute((const void*)k_warp_fused<R_, F_, WF_TH, P_, S_>, hipFuncAttributeMaxDynamicSharedMemorySize,              \
                            (int)(160 * 1024 - st_)) != hipSuccess) {                                                              \
      (void)hipGetLastError();                                                                                                     \
      fprintf(stderr, "vd3d: hipFuncSetAttribute(k_warp_fused, max dynamic LDS) failed; using the unfused warp kernels\n");        \
      return false;                                                                                                                \
    }                                                                                                                              \
  } while (0)
    WF_ATTR(true, true, false, false); WF_ATTR(true, false, false, false); WF_ATTR(false, true, false, false); WF_ATTR(false, false, false, false);
    WF_ATTR(true, true, true, false); WF_ATTR(false, true, true, false);
    if (WF_TH == 32) { WF_ATTR(true, false, false, WF_TH == 32); WF_ATTR(false, false, false, WF_TH == 32); }
#undef WF_ATTR
    attr[dev] = true;
  }
  const vd_f2* e2p = reinterpret_cast<const vd_f2*>(E2);
  const float* rowtab = wf_rowtab_get(s, H, ih, a.scale_h, a.step_y);
  if (!rowtab) return false;
  float* Sp = const_cast<float*>(S);
  vd_shift_consts sc = {};
  const vd_dev_work* work = nullptr;
#define WF_GO(R_, F_, P_, S_) hipLaunchKernelGGL((k_warp_fused<R_, F_, WF_TH, P_, S_>), g, dim3(WF_NT), bytes, s, rgb, D, Sp, a, L, R, e2p, rowtab, sc, work)
  if (fold) {
    sc = vd_shift_consts_of(fold->sp, W); work = fold->work; Sp = fold->S_out;
    if (resize) WF_GO(true, false, false, WF_TH == 32); else WF_GO(false, false, false, WF_TH == 32);
  }
  else if (resize && a.feather && pre) WF_GO(true, true, true, false);
  else if (a.feather && pre) WF_GO(false, true, true, false);
  else if (resize && a.feather) WF_GO(true, true, false, false);
  else if (resize) WF_GO(true, false, false, false);
  else if (a.feather) WF_GO(false, true, false, false);
  else WF_GO(false, false, false, false);
#undef WF_GO
  return true;
}
bool vd_launch_warp_fused(hipStream_t s, const float* rgb, int ih, int iw, const float* D, const float* S, int H, int W,
                          const vd3d_shift_params& p, uint8_t* L, uint8_t* R, const float* E2, const vd_shift_fold* fold) {
  if (fold) return !p.enable_feathering && warp_fused_impl<32>(s, rgb, ih, iw, D, S, H, W, p, L, R, nullptr, false, fold);
  if (E2 && p.enable_feathering && g_wf_pre_th == 16 && warp_fused_impl<16>(s, rgb, ih, iw, D, S, H, W, p, L, R, E2, true))
    return warp_fused_impl<16>(s, rgb, ih, iw, D, S, H, W, p, L, R, E2, false);
  if (!p.enable_feathering && g_wf_nf_th == 16 && warp_fused_impl<16>(s, rgb, ih, iw, D, S, H, W, p, L, R, nullptr, true))
    return warp_fused_impl<16>(s, rgb, ih, iw, D, S, H, W, p, L, R, nullptr, false);
  return warp_fused_impl<32>(s, rgb, ih, iw, D, S, H, W, p, L, R, E2, false);
}
// would vd_launch_warp_fused take this frame?  (the caller then runs k_e2w for the mask plane first)
bool vd_warp_fused_ok(int ih, int iw, int H, int W, const vd3d_shift_params& p) {
  return warp_fused_impl<32>(nullptr, nullptr, ih, iw, nullptr, nullptr, H, W, p, nullptr, nullptr, reinterpret_cast<const float*>(16), true);
}
// ... and with the shift plane folded in?  Not in the N-thread ATen mode on planes that have scalar tails (k_shift<true>'s libm arithmetic is not instantiated here), not
// with the development knob that selects the flat tile
#ifndef WF_FOLD_DEFAULT
#define WF_FOLD_DEFAULT 1
#endif
static int g_wf_fold = WF_FOLD_DEFAULT;   // A/B builds: -DWF_FOLD_DEFAULT=0
bool vd_warp_fold_ok(int ih, int iw, int H, int W, const vd3d_shift_params& warp_p, const vd3d_shift_params& shift_p) {
  if (!g_wf_fold || warp_p.enable_feathering || g_wf_nf_th == 16) return false;
  if (vd_tails_of((unsigned long long)H * W, shift_p.aten_threads).on) return false;
  vd_shift_fold f = {nullptr, shift_p, nullptr};
  return warp_fused_impl<32>(nullptr, nullptr, ih, iw, nullptr, nullptr, H, W, warp_p, nullptr, nullptr, nullptr, true, &f);
}
