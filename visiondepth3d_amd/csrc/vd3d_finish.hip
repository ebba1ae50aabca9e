// vd3d_finish.hip -- E1, the fused finishing kernel: apply_dof_cuda + apply_color_grade + tensor_to_frame +
// apply_side_mask + apply_sharpening + INTER_AREA fit + SBS / interlaced mux in ONE launch for both eyes
// (core/render_3d.py:1340-1419).  Replaces k_dof_grade x2 + k_sharp_mux and their two graded planes (-12N B of HBM).
//
// Two geometries of one kernel template (<DENSE, WIDE>; <true, false> = dense levels in the 64x16 geometry serves fit factor 4):
//   <true, true>  DENSE (default, the reference's dense k x k convolution order): 64x30 tile of sharpened pixels, 576 threads
//           = 9 waves.  Waves 0-7: thread = (graded row, 4-pixel strip) of the 64 x 32 block of graded pixels the tile's own columns need
//           (4 rows x 16 strips per wave, every lane busy).  Wave 8: thread = ONE pixel of the two halo columns x0 - 1 / x0 + 64 that only the
//           3x3 sharpen of the edge columns reads (32 rows x 2 sides) -- a quarter of a strip's work, in its own wave so that it costs a
//           quarter.  8.3 wave-executions of strip code per 1 920 pixels (round 2: 64x16 tile, 20 strips x 3 rows per wave with 4 idle lanes
//           and two full halo strips: 6 per 1 024; -26 %).
//   <false, false> separable levels (opt-in): 64x16 tile, 384 threads = 6 waves x 3 rows x 20 strips; a strip's horizontal neighbours are the
//           adjacent LANES (DPP wave shifts), so the halo strips stay in the row.
// Common to both:
//   load    reflect-padded u8 tile -> v/255 (exact 3-op form) -> planar float tile in LDS (interior tiles: 12-byte groups, ds_write_b128)
//   blur weight per strip: the 2:1 depth resize of Half-SBS shares its taps between the four pixels (8 loads, 24 operations); the division by
//           the focus width is the verified 3-operation form (tools/verify_fastdiv_fw.c)
//   level l dense: K x K window per output, taps row-major, one FMA per tap from 0, weight fl(k1[i] * k1[j]); the strip's 12-column window
//           comes from three ds_read_b128 per row and channel; only the strips that blend with a level run it
//           separable: V-pass on float4, H-pass through DPP
//   blend the two levels each pixel needs, grade, truncate, side bars -> packed BGR0 dwords in LDS
//   epilogue: 3x3 sharpen on float4 (4 pixels per lane), integer-ratio box average, 12-byte packed stores;
//           interior tiles with fit (1,1) / (2,1) take the vector path, everything else the generic per-pixel one.
// Arithmetic identical to k_dof_grade / k_sharp_mux and the oracle.
// Fast path conditions (else the unfused kernels run): Gaussian taps <= 9 (dof_strength <= 2), fit factors in {1,2,4},
// format in {Half-SBS, Full-SBS, Passive Interlaced, Red-Cyan Anaglyph (round 4: each eye's workgroups store their own bytes of the anaglyph)}.
// (Round 4, measured and not kept: writing the halo-column windows during the tile load and publishing the level set per wave, i.e. ONE barrier
// between the load phase and the levels instead of two: 252 -> 259 us at 4K.  Tile heights 22 / 14 at 80 VGPRs: 300 / 332 us.  Two tiles per
// workgroup with the second tile's loads requested during the first one's epilogue: hipcc needs 185 VGPRs for the loop, and 81 for the
// one-tile instantiation of the same source -- one register over the three-workgroup budget -- so the single-tile kernel stays as it is.)
#include <mutex>
#include "vd3d_dev.h"
#include "vd3d_kernels.h"

#define FF_TW 64
#define FF_R 4                      // max Gaussian radius of the fast path == strip width
#define FF_GW (FF_TW + 8)           // graded region width  (4-pixel halo each side: aligned strips; 1 is needed)
#define FF_IW (FF_GW + 2 * FF_R)    // input tile width  = 80
#define FF_IS (FF_IW / 4)           // strips per tile row = 20 (strip 0 and 19 are input halo, 1 and 18 the graded halo columns)
#define FF_GP 76                    // pitch of the graded dword tile (multiple of 4: b128 rows)
// tile rows per XCD group (vd_xcd_tile_rows).  Round 5: ONE row per group -- a tile costs 16 k .. 53 k cycles depending on the DOF levels its pixels need, every XCD
// gets exactly an eighth of the tiles, and the launch ends with the XCD whose rows were the most expensive: with two rows per group E1 took 250 us per 4K frame
// pair, with one 231 (plain row-major order, neighbouring tiles on different XCDs: 235; four rows: 310 on the harness scene).  The vertical halo rows are then
// fetched once per XCD -- bytes this kernel has to spare (0.04 of HBM).  profiles/r05_e1_persistent.md
#ifndef FF_XG
#define FF_XG 1
#endif
#ifndef FF_WIDE_TH
#define FF_WIDE_TH 26               // tile height of the wide geometry (A/B builds: -DFF_WIDE_TH=30 / 22 / 14, tools/build_ab.sh)
#endif
// geometry per instantiation (DENSE / separable): tile height, graded rows (1-pixel halo), input rows, threads
// TH = 16: the round-2 geometry (6 waves x 3 rows x 20 strips, 4 idle lanes).  "Wide" thread mapping: (TH + 2) / 4 strip waves of 4 rows x
// 16 strips + 1 halo-pixel wave.  TH = 26 -> 7 + 1 = 8 waves, two per SIMD: with the kernel's 88 VGPRs (5 waves per SIMD) TWO workgroups are
// resident per CU.  TH = 30 (9 waves, the first wide geometry) has the fewest instructions per pixel but one SIMD carries 3 waves of a
// workgroup, a second workgroup would need 6 there, and the residency probe (tools/probe_phases.py: HW_ID + s_memrealtime per workgroup)
// showed exactly ONE workgroup per CU at any time: 425 us per 4K frame against 341 us for TH = 26 (TH = 22: 396 us).
template <int TH_> struct ff_geo {
  static constexpr int TH = TH_;
  static constexpr bool WIDE = TH_ != 16;
  static constexpr int GH = TH + 2;
  static constexpr int IH = GH + 2 * FF_R;
  static constexpr int NSW = GH / 4;                               // strip waves (wide mapping)
  static constexpr int NT = WIDE ? 64 * (NSW + 1) : 384;
  static constexpr int HS = IH * 9 + ((16 - (IH * 9) % 32 + 32) % 32);   // halo-window plane of one side: IH rows x 9 floats, padded to 16 mod 32 banks
  static constexpr int HC = 2 * HS;                                // ... of one channel (both sides)
  static_assert(!WIDE || (GH % 4 == 0 && 2 * GH <= 64), "wide mapping: whole strip waves and one halo wave");
};
#define FF_FW_STD 0.350001007f      // (float)(0.35 + 1e-6): focus_width of the render loop (core/render_3d.py:1357-1360) + :794's 1e-6

struct vd_ff_args {
  int H, W, eh, ew;
  int fit_w, fit_h, in_w, in_h, xo, yo, fx, fy, out_w, format;
  int use_override, bar_w, bar_s;
  float focal;
  int ntx, nty, xcd;                // tile grid per eye; XCD row-group order on (vd_xcd_tile_rows)
};


// neighbour lanes through DPP (gfx9 wave shifts): value of lane-1 / lane+1 (own value at the wave edge, never used there)
VD_DEV float ff_from_left(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
}
VD_DEV float ff_from_right(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
}

// The two levels a pixel blends (:822-834: level lo and lo + 1, weight alpha) arrive in ascending order, so ONE running value per pixel and
// channel is enough: it starts as the unblurred pixel (level 0), is replaced when level lo arrives and becomes (1 - alpha) lo + alpha hi --
// the reference's expression, same operations -- when level lo + 1 does (every pixel's hi level is in its tile's level set).  Half the
// registers of keeping lo and hi apart (24 -> 12 per strip).
VD_DEV float ff_fold(float res, float o, float alpha, int level, int lo) {
  if (level == lo) res = o;
  if (level == lo + 1) res = (1.0f - alpha) * res + alpha * o;
  return res;
}

// one Gaussian level: K = 9 - 2*OFF taps, vertical then horizontal.  Executed by every lane of the wave (the DPP exchange
// needs the neighbours' vertical sums); `mine` = this strip blends with this level.
template <int OFF, int IH>
VD_DEV void ff_level(const float (*tile)[IH][FF_IW], const float* __restrict__ kern, bool active, bool mine, int sy, int ss, int level,
                     const int lo[4], const vd_f4& alpha, vd_f4 vres[3]) {
  constexpr int K = 2 * (FF_R - OFF) + 1;
  float kw[K];
#pragma unroll
  for (int t = 0; t < K; ++t) kw[t] = kern[t];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    vd_f4 vs = {0.f, 0.f, 0.f, 0.f};
    if (active) {   // vd_gauss_sym with the loads consumed pair by pair (sched_barrier: keep at most one pair of rows in flight,
                    // the kernel is register-limited and other waves hide the LDS latency)
      constexpr int r = K / 2;
      auto ld = [&](int tt) { return *reinterpret_cast<const vd_f4*>(&tile[c][sy + OFF + tt][4 * ss]); };
      vd_f4 acc = kw[0] * (ld(0) + ld(K - 1));
#pragma unroll
      for (int tt = 1; tt < r; ++tt) {
        __builtin_amdgcn_sched_barrier(0);
        acc = vd_vfma((vd_f4)(kw[tt]), ld(tt) + ld(K - 1 - tt), acc);
      }
      __builtin_amdgcn_sched_barrier(0);
      vs = vd_vfma((vd_f4)(kw[r]), ld(r), acc);
    }
    float win[12];   // vertical sums at tile columns 4(ss-1) .. 4(ss+1)+3
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      win[4 + q] = vs[q];
      if (q >= OFF) win[q] = ff_from_left(vs[q]);            // only the columns the K-tap window reaches
      if (q < 4 - OFF) win[8 + q] = ff_from_right(vs[q]);
    }
    if (mine) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float o = vd_gauss_sym<K, float>(kw, &win[q + OFF]);
        vres[c][q] = ff_fold(vres[c][q], o, alpha[q], level, lo[q]);
      }
    }
  }
}

// The same level in the reference's DENSE association (vd3d_render_params::dof_dense_conv): one K x K window per
// output, taps row-major, one FMA per tap from 0, weight = fl(k1[i] * k1[j]) -- what F.conv2d (oneDNN) computes on the CPU.  The strip's
// 12-column window comes straight from the LDS tile (three ds_read_b128 per row and channel), the weight row is rebuilt per tile row
// (K multiplies amortised over 12 K FMAs); four independent accumulation chains per channel.  No lane exchange: only the strips that
// blend with this level run it.
template <int OFF, int IH>
VD_DEV void ff_level_dense(const float (*tile)[IH][FF_IW], const float* __restrict__ w2, bool mine, int sy, int ss, int level, const int lo[4],
                           const vd_f4& alpha, vd_f4 vres[3]) {
  constexpr int K = 2 * (FF_R - OFF) + 1;
  if (!mine) return;
  vd_f4 acc[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) acc[c] = (vd_f4){0.f, 0.f, 0.f, 0.f};
  const float* rp0 = &tile[0][sy + OFF][4 * ss - 4];
  vd_f4 n0 = *reinterpret_cast<const vd_f4*>(rp0), n1 = *reinterpret_cast<const vd_f4*>(rp0 + 4), n2 = *reinterpret_cast<const vd_f4*>(rp0 + 8);
#pragma unroll
  for (int i = 0; i < K; ++i) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const vd_f4 w0 = n0, w1 = n1, w2v = n2;
      if (i + 1 < K || c + 1 < 3) {
        const float* rp = &tile[c + 1 < 3 ? c + 1 : 0][sy + OFF + (c + 1 < 3 ? i : i + 1)][4 * ss - 4];
        n0 = *reinterpret_cast<const vd_f4*>(rp); n1 = *reinterpret_cast<const vd_f4*>(rp + 4); n2 = *reinterpret_cast<const vd_f4*>(rp + 8);
      }
      __builtin_amdgcn_sched_barrier(0);
      const float win[12] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3], w2v[0], w2v[1], w2v[2], w2v[3]};
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const float wt = w2[i * K + j];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[c][q] = vd_fma(win[q + OFF + j], wt, acc[c][q]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int q = 0; q < 4; ++q) vres[c][q] = ff_fold(vres[c][q], acc[c][q], alpha[q], level, lo[q]);
}

// One pixel of a halo column (graded row sy, side 0 = column x0 - 1 / 1 = column x0 + 64) in the same dense association, from the
// 9-column halo windows `hal` (see the kernel): single-dword LDS reads, one accumulation chain per channel.  Result folded into element 0 of vres.
template <int OFF, int FF_HS>
VD_DEV void ff_level_dense_px(const float* __restrict__ hal, const float* __restrict__ w2, bool mine, int sy, int side, int level, int lo0,
                              float alpha0, vd_f4 vres[3]) {
  constexpr int K = 2 * (FF_R - OFF) + 1;
  if (!mine) return;
  float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < K; ++i) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* rp = hal + c * (2 * FF_HS) + side * FF_HS + (sy + OFF + i) * 9 + OFF;
#pragma unroll
      for (int j = 0; j < K; ++j) acc[c] = vd_fma(rp[j], w2[i * K + j], acc[c]);
    }
    __builtin_amdgcn_sched_barrier(0);   // keep one window row of loads in flight: the kernel is register-limited (2 workgroups per CU)
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) vres[c][0] = ff_fold(vres[c][0], acc[c], alpha0, level, lo0);
}

VD_DEV float ff_byte(uint32_t v, int sh) { return (float)((v >> sh) & 0xffu); }   // v_cvt_f32_ubyteN

// 3x3 sharpen (:717-732) of 4 consecutive pixels of one row of the graded dword tile (interior: no reflection needed)
VD_DEV void ff_sharp4(const uint32_t (*gb)[FF_GP], int gy, int gc, float kn, float kc, int s[3][4]) {
  const uint4 U = *reinterpret_cast<const uint4*>(&gb[gy - 1][gc]);
  const uint4 C = *reinterpret_cast<const uint4*>(&gb[gy][gc]);
  const uint4 D = *reinterpret_cast<const uint4*>(&gb[gy + 1][gc]);
  const uint32_t lf = gb[gy][gc - 1], rt = gb[gy][gc + 4];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int sh = 8 * c;
    const vd_f4 u = {ff_byte(U.x, sh), ff_byte(U.y, sh), ff_byte(U.z, sh), ff_byte(U.w, sh)};
    const vd_f4 d = {ff_byte(D.x, sh), ff_byte(D.y, sh), ff_byte(D.z, sh), ff_byte(D.w, sh)};
    const vd_f4 m = {ff_byte(C.x, sh), ff_byte(C.y, sh), ff_byte(C.z, sh), ff_byte(C.w, sh)};
    const vd_f4 l = {ff_byte(lf, sh), m.x, m.y, m.z}, r = {m.y, m.z, m.w, ff_byte(rt, sh)};
    // 0 + kn*u + kn*l + kc*m + kn*r + kn*d in that order; the leading "0 +" only decides the sign of an all-zero sum
    vd_f4 acc = kn * u;
    acc = acc + kn * l;
    acc = acc + kc * m;
    acc = acc + kn * r;
    acc = acc + kn * d;
#pragma unroll
    for (int q = 0; q < 4; ++q) s[c][q] = (int)vd_sat_rne_u8(acc[q]);
  }
}

// Red-Cyan anaglyph of generate_anaglyph_3d (:866-883) on the sharpened / fitted values fv[c][q] (c = byte order of the eye, B G R) of up to 4
// consecutive output pixels: the reference splits the BGR frame and CALLS the planes r, g, b, so output byte 0 is a function of the left eye's three
// bytes only and output bytes 1 and 2 of the right eye's only -- each eye's workgroup stores its own bytes, no cross-eye exchange.  Same
// expressions as k_sharp_mux.
VD_DEV void ff_anaglyph_store(uint8_t* o, int eye, const int fv[3][4], int nvalid) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (q >= nvalid) break;
    const float p0 = vd_u8_unit((float)fv[0][q]), p1 = vd_u8_unit((float)fv[1][q]), p2 = vd_u8_unit((float)fv[2][q]);
    if (eye == 0) {
      const float red = ((float)0.4561 * p0 + (float)0.5005 * p1) + (float)0.1762 * p2;
      o[3 * q] = (uint8_t)(vd_clamp(red, 0.f, 1.f) * 255.0f);
    } else {
      const float green = ((float)0.3764 * p0 + (float)0.7616 * p1) - (float)0.1876 * p2;
      const float blue = ((float)-0.0401 * p0 - (float)0.1126 * p1) + (float)1.2723 * p2;
      o[3 * q + 1] = (uint8_t)(vd_clamp(green, 0.f, 1.f) * 255.0f);
      o[3 * q + 2] = (uint8_t)(vd_clamp(blue, 0.f, 1.f) * 255.0f);
    }
  }
}

// epilogue of the finishing kernels: 3x3 sharpen (:717-732) on the tile of graded dwords `gb` (row 0 = image row gy0, column 0 = image column gx0),
// integer-ratio INTER_AREA (:1413), mux (SBS halves / interlaced rows / anaglyph bytes).  Called by k_finish_fused behind its grade phase and by
// k_sharp_fit on planes that are graded already.
template <int FF_TH, int FF_NT>
VD_DEV void ff_epilogue_k(const uint32_t (*gb)[FF_GP], const vd_ff_args& a, float kn, float kc, int eye, int x0, int y0, int gx0, int gy0,
                          int tid, uint8_t* __restrict__ out) {
  const int H = a.H, W = a.W;
  // epilogue: sharpen (:717-732) + integer-ratio INTER_AREA (:1413) + mux
  const int ow = FF_TW / a.fx, oh = FF_TH / a.fy;      // output pixels produced by this tile
  const int ox0 = x0 / a.fx, oy0 = y0 / a.fy;
  // vector path: fits (1, 1), (2, 1) and -- round 5 -- (2, 2): the GUI's own default on a 4K source (Full-SBS = 1920 x 1080 eyes, VisionDepth3D.py:1405-1453)
  const bool out_interior = x0 >= 1 && x0 + FF_TW <= W - 1 && y0 >= 1 && y0 + FF_TH <= H - 1 &&
                            ((a.fy == 1 && (a.fx == 1 || a.fx == 2)) || (a.fy == 2 && a.fx == 2 && (FF_TH & 1) == 0)) && ox0 + ow <= a.in_w && oy0 + oh <= a.in_h;
  if (out_interior) {
    // one task = 4 consecutive OUTPUT pixels of one row = fx groups of 4 sharpened pixels (x fy rows); 12-byte packed store
    const int ngrp = ow / 4;                          // 16 (fx = 1) or 8 (fx = 2)
    for (int t = tid; t < oh * ngrp; t += FF_NT) {
      const int ty = t / ngrp, m = t - ty * ngrp;
      const int oy = oy0 + ty;
      if (a.format == VD3D_FMT_INTERLACED && (((oy + a.yo) & 1) != eye)) continue;
      uint32_t pack[3] = {0, 0, 0};
      int fv[3][4];
      if (a.fx == 1) {
        ff_sharp4(gb, ty + 1, 4 + 4 * m, kn, kc, fv);
      } else if (a.fy == 2) {   // 2 x 2 box of sharpened bytes: OpenCV's ResizeAreaFastVec, (sum + 2) >> 2 (same as the per-pixel path below)
        // left half of the four output pixels, then the right half: two tiles of sharpened values live at a time, not four (the kernel's 80-VGPR
        // budget = three workgroups per CU is set by the dense levels; all four at once took 96)
#pragma unroll
        for (int hx = 0; hx < 2; ++hx) {
          int sa[3][4], sb[3][4];
          ff_sharp4(gb, 2 * ty + 1, 4 + 8 * m + 4 * hx, kn, kc, sa);
          ff_sharp4(gb, 2 * ty + 2, 4 + 8 * m + 4 * hx, kn, kc, sb);
#pragma unroll
          for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int c = 0; c < 3; ++c) fv[c][2 * hx + q] = ((sa[c][2 * q] + sa[c][2 * q + 1]) + (sb[c][2 * q] + sb[c][2 * q + 1]) + 2) >> 2;
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        int s0[3][4], s1[3][4];
        ff_sharp4(gb, ty + 1, 4 + 8 * m, kn, kc, s0);
        ff_sharp4(gb, ty + 1, 8 + 8 * m, kn, kc, s1);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const int sum = q < 2 ? s0[c][2 * q] + s0[c][2 * q + 1] : s1[c][2 * q - 4] + s1[c][2 * q - 3];
            fv[c][q] = (int)vd_sat_rne_u8((float)sum * 0.5f);
          }
      }
      const bool single = a.format == VD3D_FMT_INTERLACED || a.format == VD3D_FMT_ANAGLYPH;   // one eye-sized canvas
      const int oxq = ox0 + 4 * m + a.xo + (single ? 0 : eye * a.fit_w);
      uint8_t* o = out + ((size_t)(oy + a.yo) * a.out_w + oxq) * 3;
      if (a.format == VD3D_FMT_ANAGLYPH) { ff_anaglyph_store(o, eye, fv, 4); continue; }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < 3; ++c) { const int bi = 3 * q + c; pack[bi >> 2] |= (uint32_t)fv[c][q] << (8 * (bi & 3)); }
      if ((reinterpret_cast<uintptr_t>(o) & 3) == 0) {
        uint32_t* o32 = reinterpret_cast<uint32_t*>(o);
        o32[0] = pack[0]; o32[1] = pack[1]; o32[2] = pack[2];
      } else {
        for (int bi = 0; bi < 12; ++bi) o[bi] = (uint8_t)(pack[bi >> 2] >> (8 * (bi & 3)));
      }
    }
    return;
  }
  const float scale = 1.f / (float)(a.fx * a.fy);
  for (int t = tid; t < oh * (ow / 4); t += FF_NT) {
    const int tq = t / oh, ty = t - tq * oh;
    const int oy = oy0 + ty;
    if (oy >= a.in_h) continue;
    uint32_t pack[3] = {0, 0, 0};
    int fv[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
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
        fv[c][q] = (int)v;
      }
    }
    if (a.format == VD3D_FMT_INTERLACED && (((oy + a.yo) & 1) != eye)) continue;
    const bool single = a.format == VD3D_FMT_INTERLACED || a.format == VD3D_FMT_ANAGLYPH;
    const int oxq = ox0 + tq * 4 + a.xo + (single ? 0 : eye * a.fit_w);
    uint8_t* o = out + ((size_t)(oy + a.yo) * a.out_w + oxq) * 3;
    if (a.format == VD3D_FMT_ANAGLYPH) { ff_anaglyph_store(o, eye, fv, nvalid); continue; }
    if (nvalid == 4 && ((size_t)(o - out) & 3) == 0) {
      uint32_t* o32 = reinterpret_cast<uint32_t*>(o);
      o32[0] = pack[0]; o32[1] = pack[1]; o32[2] = pack[2];
    } else {
      for (int bi = 0; bi < 3 * nvalid; ++bi) o[bi] = (uint8_t)(pack[bi >> 2] >> (8 * (bi & 3)));
    }
  }
}

template <int FF_TH, int FF_NT>
VD_DEV void ff_epilogue(const uint32_t (*gb)[FF_GP], const vd_ff_args& a, const vd_finish_consts& fc, int eye, int x0, int y0, int gx0, int gy0,
                        int tid, uint8_t* __restrict__ out) {
  ff_epilogue_k<FF_TH, FF_NT>(gb, a, fc.sharp_kn, fc.sharp_kc, eye, x0, y0, gx0, gy0, tid, out);
}

#ifndef FF_OCC_ATTR
#define FF_OCC_ATTR   // A/B builds: -DFF_OCC_ATTR='__attribute__((amdgpu_waves_per_eu(8, 8)))' forces the 64-VGPR budget
#endif
VD_STAMP_DECL(ff_stamps);
#ifdef VD_PHASE_STAMPS
extern "C" __attribute__((visibility("default"))) int vd3d_debug_stamps_e1(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ff_stamps), sizeof(ff_stamps)); }
#endif
VD_OCC_DECL(ff_occ, vd3d_debug_occ_e1)

template <bool DENSE, int TH_>
__global__ __launch_bounds__(ff_geo<TH_>::NT) FF_OCC_ATTR void k_finish_fused(const uint8_t* __restrict__ eyeL, const uint8_t* __restrict__ eyeR,
                                                        const float* __restrict__ dn, vd_finish_consts fc, vd_ff_args a,
                                                        const vd_dev_work* __restrict__ w, const float* __restrict__ w2g,
                                                        uint8_t* __restrict__ out) {
  constexpr bool WIDE = ff_geo<TH_>::WIDE;
  static_assert(DENSE || !WIDE, "the separable levels exchange vertical sums between adjacent lanes: 64x16 geometry only");
  constexpr int FF_TH = ff_geo<TH_>::TH, FF_GH = ff_geo<TH_>::GH, FF_IH = ff_geo<TH_>::IH, FF_NT = ff_geo<TH_>::NT;
  constexpr int FF_HS = ff_geo<TH_>::HS, FF_HC = ff_geo<TH_>::HC, FF_NSW = ff_geo<TH_>::NSW;
  __shared__ __attribute__((aligned(16))) float tile[3][FF_IH][FF_IW];
  __shared__ __attribute__((aligned(16))) uint32_t gb[FF_GH][FF_GP];
  // WIDE: the 9-column windows of the two halo columns, copied out of the tile with a 9-float pitch: the halo-pixel wave reads single
  // dwords, and in the tile itself its 32 lanes of a half-wave would share 4 banks (one column, pitch 16 mod 32).  Here lane (row, side)
  // reads bank (9 row + 16 side + j) mod 32: all different.
  __shared__ float hal[WIDE ? 3 * FF_HC : 1];
  __shared__ int lvl_mask;                    // levels any pixel of this tile needs
  int trow, tbx;                                   // tile row over both eyes (0 .. 2 nty - 1) and tile column
  vd_xcd_tile_rows(blockIdx.x, a.ntx, FF_XG, a.xcd, &trow, &tbx);
  if (trow >= 2 * a.nty) return;                   // padding workgroup of the last group (workgroup-uniform, before any barrier)
  const int eye = trow >= a.nty ? 1 : 0, tby = trow - eye * a.nty;
  const uint8_t* __restrict__ src = eye == 0 ? eyeL : eyeR;
  const int H = a.H, W = a.W;
  const int x0 = tbx * FF_TW, y0 = tby * FF_TH;
  const int gx0 = x0 - 4, gy0 = y0 - 1;            // graded region origin
  const int ix0 = gx0 - FF_R, iy0 = gy0 - FF_R;    // input tile origin
  const int tid = threadIdx.x;
  VD_STAMP(ff_stamps, 0, false);
  VD_OCC_IN(ff_occ);

  const int lane = tid & 63, wv = tid >> 6;
  bool active, strip, halo_px = false;
  int sy, ss, hq = 0;                             // hq: the live pixel of a halo-column strip (WIDE: 3 = left column x0 - 1, 0 = right column x0 + 64)
  if (WIDE) {
    // waves 0-7: thread = (graded row 4 wv + lane % 4, strip 2 + lane / 4) -- the 64 x 32 graded pixels of the tile's own columns.  Row
    // fastest: with the 80-float tile pitch (16 banks mod 64) every 16-lane group of a ds_read_b128 then covers the 64 banks exactly
    // (row-major lanes measured 18x the bank-conflict cycles: two rows of a group alias 32 banks);
    // wave 8: thread = one pixel of the halo columns (graded row lane / 2; even lanes the left column, odd lanes the right one)
    active = true; strip = true;
    if (wv < FF_NSW) { sy = 4 * wv + (lane & 3); ss = 2 + (lane >> 2); }
    else {
      sy = min(lane >> 1, FF_GH - 1); halo_px = true; ss = (lane & 1) ? FF_IS - 2 : 1; hq = (lane & 1) ? 0 : 3;
      strip = (lane >> 1) < FF_GH;                 // 2 GH halo pixels; the rest of the wave idles (TH = 14)
    }
  } else {
    // thread = (graded row sy, tile strip ss); wave v owns rows 3v..3v+2; strips 1..18 are the graded region
    active = lane < 3 * FF_IS;
    sy = active ? 3 * wv + lane / FF_IS : 0; ss = active ? lane % FF_IS : 0;
    strip = active && ss >= 1 && ss <= FF_IS - 2;
  }
  const int gy = gy0 + sy, gxs = ix0 + 4 * ss;    // image coordinates of the strip's first pixel
  // the frame scalars of the device-resident trackers: requested at the top (scalar loads), consumed two and three barriers later
  const float w_focal = a.use_override ? a.focal : w->focal;
  const int w_bar_w = a.use_override ? a.bar_w : w->bar_width, w_bar_s = a.use_override ? a.bar_s : w->bar_side;
  // The depth samples of the blur weight (exact 2:1 interior strips: 8 loads) are requested NOW, before the tile load, and consumed after
  // the first barrier: at two resident workgroups per CU their round trip was the longest exposed wait of the kernel (s_memtime stamps:
  // 42 % of a workgroup's life sat in the phase that used to issue them).
  const int yc = min(max(gy, 0), H - 1);   // halo pixels outside the image are never used
  const bool fast21 = strip && fc.nlev && 2 * a.eh == H && 2 * a.ew == W && gxs >= 4 && gxs + 6 <= W;
  vd_tap ay21 = {0, 0, 0.f, 0.f};
  float p0[4] = {0.f, 0.f, 0.f, 0.f}, p1[4] = {0.f, 0.f, 0.f, 0.f};
  if (fast21) {
    ay21 = vd_tap21(a.eh, yc);
    const int c0 = (gxs >> 1) - 1;
    const float* r0 = dn + (size_t)ay21.i0 * a.ew + c0;
    const float* r1 = dn + (size_t)ay21.i1 * a.ew + c0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { p0[j] = r0[j]; p1[j] = r1[j]; }
  }
  if (tid == 0) lvl_mask = 0;
  const bool in_interior = ix0 >= 0 && ix0 + FF_IW <= W && iy0 >= 0 && iy0 + FF_IH <= H && (W & 3) == 0 &&
                           (reinterpret_cast<uintptr_t>(src) & 3) == 0;
  if (in_interior) {
    // one task = 4 pixels of one row: 3 dword loads (12 B = 4 BGR pixels), three ds_write_b128.  A thread owns tasks tid and tid + NT
    // (IH * IS <= 2 NT): both tasks' loads are requested before either is converted, so the phase waits for ONE global round trip, not two.
    static_assert(FF_IH * FF_IS <= 2 * FF_NT, "two tile-load tasks per thread");
    uint32_t ld[2][3] = {{0u, 0u, 0u}, {0u, 0u, 0u}};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int t = tid + k * FF_NT;
      if (t < FF_IH * FF_IS) {
        const int ty = t / FF_IS, g = t - ty * FF_IS;
        const uint32_t* p0 = reinterpret_cast<const uint32_t*>(src + ((size_t)(iy0 + ty) * W + ix0 + 4 * g) * 3);
        ld[k][0] = p0[0]; ld[k][1] = p0[1]; ld[k][2] = p0[2];
      }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int t = tid + k * FF_NT;
      if (t < FF_IH * FF_IS) {
        const int ty = t / FF_IS, g = t - ty * FF_IS;
        const uint32_t a0 = ld[k][0], a1 = ld[k][1], a2 = ld[k][2];
        // byte k of the 12-byte group: pixel k/3, channel BGR[k%3]
#define FF_B(k) ff_byte((k) < 4 ? a0 : ((k) < 8 ? a1 : a2), 8 * ((k) & 3))
#pragma unroll
        for (int c = 0; c < 3; ++c) {   // plane 0 = R (byte 2), 1 = G (byte 1), 2 = B (byte 0)
          const int bo = 2 - c;
          const vd_f4 v4 = {vd_u8_unit(FF_B(0 + bo)), vd_u8_unit(FF_B(3 + bo)), vd_u8_unit(FF_B(6 + bo)), vd_u8_unit(FF_B(9 + bo))};
          *reinterpret_cast<vd_f4*>(&tile[c][ty][4 * g]) = v4;
        }
#undef FF_B
      }
    }
  } else {
    // border tiles (reflected coordinates, byte loads): all of a thread's pixels in flight together (round 6: the one-pixel-per-iteration loop was a chain of five dependent
    // global round trips, and the border tiles were the launch's slowest workgroups)
    constexpr int NPX = (FF_IH * FF_IW + FF_NT - 1) / FF_NT;
    uint8_t bb[NPX][3];
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
      const int t = min(tid + j * FF_NT, FF_IH * FF_IW - 1);
      const int ty = t / FF_IW, tx = t - ty * FF_IW;
      const int y = vd_reflect(iy0 + ty, H), x = vd_reflect(ix0 + tx, W);
      const uint8_t* px = src + ((size_t)y * W + x) * 3;
      bb[j][0] = px[0]; bb[j][1] = px[1]; bb[j][2] = px[2];
    }
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
      const int t = tid + j * FF_NT;
      if (t < FF_IH * FF_IW) {
        const int ty = t / FF_IW, tx = t - ty * FF_IW;
        tile[0][ty][tx] = vd_u8_unit((float)bb[j][2]);
        tile[1][ty][tx] = vd_u8_unit((float)bb[j][1]);
        tile[2][ty][tx] = vd_u8_unit((float)bb[j][0]);
      }
    }
  }
  int lo[4] = {0, 0, 0, 0};
  vd_f4 alpha = {0.f, 0.f, 0.f, 0.f};
  int lmin = 9, lmax = -1;
  __syncthreads();   // lvl_mask = 0 visible before the atomicOr below; tile complete
  // Round 6: the depth samples requested at the top are pinned HERE.  Without this hipcc hoists their first consumers (the 2:1 interpolation below) to right behind the
  // loads, inside the `if (fast21)` block, with an `s_waitcnt vmcnt(0)` -- the kernel then sat out their round trip BEFORE issuing the tile loads: two dependent trips
  // where the source meant one (found in the ISA after W1's wait analysis, profiles/r06_w1_phases.md).  Measured: 229.6 us against 229.4 -- E1 issues at 0.78 of the VALU rate
  // and its other resident workgroups covered the wait either way; kept because it is what the source says.
#pragma unroll
  for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(p0[j]), "+v"(p1[j]));
  VD_STAMP(ff_stamps, 1, false);
  if (WIDE && DENSE) {   // halo-column windows (consumed after the next barrier)
    for (int t = tid; t < 3 * 2 * FF_IH; t += FF_NT) {
      const int c = t / (2 * FF_IH), rem = t - c * (2 * FF_IH), side = rem / FF_IH, row = rem - side * FF_IH;
      const float* srcp = &tile[c][row][side ? (4 * (FF_IS - 2)) - FF_R : (4 + 3) - FF_R];   // window of tile column 72 / 7
      float* dstp = &hal[c * FF_HC + side * FF_HS + row * 9];
#pragma unroll
      for (int j = 0; j < 9; ++j) dstp[j] = srcp[j];
    }
  }
  int my_mask = 0;
  if (strip && fc.nlev) {
    const float focal = w_focal;
    float dd[4];
    if (fast21) {
      // exact 2:1 (Half-SBS), strip away from the left / right border: the four pixels x = 4m .. 4m+3 share the eye columns
      // 2m-1 .. 2m+2 and their taps are a parity rule (vd_tap21: even x -> (i0, w1) = (x/2 - 1, 0.75), odd x -> ((x-1)/2, 0.25)),
      // so the strip costs 8 loads (issued at the top of the kernel) and 24 operations instead of 4 independent bilinear samples.
      // Same association as vd_bilerp: a = fma(p[i0], w0, w1 * p[i1]) per row, then fma(a, wy0, wy1 * b).
      const vd_tap ay = ay21;
      const float a0 = vd_fma(p0[0], 0.25f, 0.75f * p0[1]), b0 = vd_fma(p1[0], 0.25f, 0.75f * p1[1]);
      const float a1 = vd_fma(p0[1], 0.75f, 0.25f * p0[2]), b1 = vd_fma(p1[1], 0.75f, 0.25f * p1[2]);
      const float a2 = vd_fma(p0[1], 0.25f, 0.75f * p0[2]), b2 = vd_fma(p1[1], 0.25f, 0.75f * p1[2]);
      const float a3 = vd_fma(p0[2], 0.75f, 0.25f * p0[3]), b3 = vd_fma(p1[2], 0.75f, 0.25f * p1[3]);
      dd[0] = vd_fma(a0, ay.w0, ay.w1 * b0); dd[1] = vd_fma(a1, ay.w0, ay.w1 * b1);
      dd[2] = vd_fma(a2, ay.w0, ay.w1 * b2); dd[3] = vd_fma(a3, ay.w0, ay.w1 * b3);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int x = min(max(gxs + q, 0), W - 1);
        if (a.eh == H && a.ew == W) dd[q] = dn[(size_t)yc * W + x];
        else if (2 * a.eh == H && 2 * a.ew == W) {   // exact 2:1 at the frame border
          const vd_tap ay = vd_tap21(a.eh, yc), ax = vd_tap21(a.ew, x);
          const float* r0 = dn + (size_t)ay.i0 * a.ew;
          const float* r1 = dn + (size_t)ay.i1 * a.ew;
          dd[q] = vd_bilerp(r0[ax.i0], r0[ax.i1], r1[ax.i0], r1[ax.i1], ax.w0, ax.w1, ay.w0, ay.w1);
        } else {
          const vd_tap ay = vd_interp_tap(a.eh, H, yc), ax = vd_interp_tap(a.ew, W, x);
          const float* r0 = dn + (size_t)ay.i0 * a.ew;
          const float* r1 = dn + (size_t)ay.i1 * a.ew;
          dd[q] = vd_bilerp(r0[ax.i0], r0[ax.i1], r1[ax.i0], r1[ax.i1], ax.w0, ax.w1, ay.w0, ay.w1);
        }
      }
    }
    // blur weight |d - focal| / fw: for the loop's focus width (fw == FF_FW_STD) the division is the verified 3-operation form
    // (tools/verify_fastdiv_fw.c: every float in [1e-30, 1] equals the IEEE quotient; 0 maps to 0; the 4.3 M floats below 2.2e-32,
    // where the residual underflows, take the IEEE division)
    const bool fastfw = fc.fw == FF_FW_STD;
    const float rcfw = 1.0f / FF_FW_STD;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float ad = fabsf(dd[q] - focal);
      float qv;
      if (fastfw && (ad >= 1e-30f || ad == 0.f)) {
        const float q0 = ad * rcfw;
        qv = vd_fma(vd_fma(-q0, FF_FW_STD, ad), rcfw, q0);
      } else qv = ad / fc.fw;
      const float bw = vd_clamp_fin(qv, 0.f, 1.f);
      const float bi = vd_clamp_fin(bw * (float)fc.nlev, 0.f, fc.imax);
      int l = (int)floorf(bi);
      l = l > fc.nlev - 1 ? fc.nlev - 1 : (l < 0 ? 0 : l);
      lo[q] = l; alpha[q] = bi - (float)l;
      if (!WIDE || !halo_px || q == hq) { lmin = min(lmin, l); lmax = max(lmax, l + 1); }
    }
    if (WIDE && halo_px && hq == 3) { lo[0] = lo[3]; alpha[0] = alpha[3]; }   // a halo-column thread keeps its one pixel in element 0
    for (int l = max(lmin, 1); l <= lmax; ++l) my_mask |= 1 << l;
  }
  {
    // The tile's level set: OR of the lanes' masks.  One same-address LDS atomic per LANE serialises 512 ways; reduce inside the wave first
    // (one ballot per level bit, the result is wave-uniform) and let one lane per wave publish it: 8 atomics per workgroup.
    int wmask = 0;
#pragma unroll
    for (int l = 1; l <= 4; ++l)
      if (__ballot((my_mask >> l) & 1)) wmask |= 1 << l;
    if ((tid & 63) == 0 && wmask) atomicOr(&lvl_mask, wmask);
  }
  vd_f4 vres[3];   // level 0 = the pixel itself; ff_fold turns it into the blend of the two levels the pixel needs
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    vres[c] = *reinterpret_cast<const vd_f4*>(&tile[c][sy + FF_R][4 * ss]);
    if (WIDE && halo_px && hq == 3) vres[c][0] = vres[c][3];
  }
  __syncthreads();
  VD_STAMP(ff_stamps, 2, false);
  const int need_mask = lvl_mask;
  for (int l = 0; l < fc.nlev; ++l) {  // level l+1 of the reference's stack
    if (!(need_mask >> (l + 1) & 1)) continue;  // no pixel of this tile blends with this level (workgroup-uniform)
    const int off = FF_R - fc.ksz[l] / 2;
    const bool mine = (my_mask >> (l + 1)) & 1;
    if (DENSE) {
      if (WIDE && wv == FF_NSW) {   // wave-uniform: the halo-pixel wave
        const int side = lane & 1;
        switch (off) {
          case 0: ff_level_dense_px<0, FF_HS>(hal, w2g + 81 * l, mine, sy, side, l + 1, lo[0], alpha[0], vres); break;
          case 1: ff_level_dense_px<1, FF_HS>(hal, w2g + 81 * l, mine, sy, side, l + 1, lo[0], alpha[0], vres); break;
          case 2: ff_level_dense_px<2, FF_HS>(hal, w2g + 81 * l, mine, sy, side, l + 1, lo[0], alpha[0], vres); break;
          default: ff_level_dense_px<3, FF_HS>(hal, w2g + 81 * l, mine, sy, side, l + 1, lo[0], alpha[0], vres); break;
        }
        continue;
      }
      switch (off) {
        case 0: ff_level_dense<0, FF_IH>(tile, w2g + 81 * l, mine, sy, ss, l + 1, lo, alpha, vres); break;
        case 1: ff_level_dense<1, FF_IH>(tile, w2g + 81 * l, mine, sy, ss, l + 1, lo, alpha, vres); break;
        case 2: ff_level_dense<2, FF_IH>(tile, w2g + 81 * l, mine, sy, ss, l + 1, lo, alpha, vres); break;
        default: ff_level_dense<3, FF_IH>(tile, w2g + 81 * l, mine, sy, ss, l + 1, lo, alpha, vres); break;
      }
      continue;
    }
    switch (off) {  // compile-time tap count => all register indexing is static
      case 0: ff_level<0, FF_IH>(tile, fc.kern[l], active, mine, sy, ss, l + 1, lo, alpha, vres); break;
      case 1: ff_level<1, FF_IH>(tile, fc.kern[l], active, mine, sy, ss, l + 1, lo, alpha, vres); break;
      case 2: ff_level<2, FF_IH>(tile, fc.kern[l], active, mine, sy, ss, l + 1, lo, alpha, vres); break;
      default: ff_level<3, FF_IH>(tile, fc.kern[l], active, mine, sy, ss, l + 1, lo, alpha, vres); break;
    }
  }
  VD_STAMP(ff_stamps, 3, false);
  if (strip) {  // blend, grade (:750-767), truncate, side bars (:885-892); float4 = the strip's 4 pixels
    const int bar_w = w_bar_w, bar_s = w_bar_s;
    vd_f4 rgbv[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      vd_f4 v = vres[c];
      if (fc.nlev) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = vd_clamp_fin(v[q], 0.f, 1.f);
      }
      rgbv[c] = v;
    }
    const vd_f4 luma = ((float)0.2126 * rgbv[0] + (float)0.7152 * rgbv[1]) + (float)0.0722 * rgbv[2];
    uint32_t pk[4] = {0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      vd_f4 v = luma + (rgbv[c] - luma) * fc.sat;
      v = 0.5f + (v - 0.5f) * fc.con;
      v = v + fc.bri;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float vc = vd_clamp_fin(v[q], 0.f, 1.f);
        pk[q] |= (uint32_t)(uint8_t)(vc * 255.0f) << (8 * (2 - c));  // byte 0 = B, 1 = G, 2 = R
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int x = gxs + ((WIDE && halo_px) ? hq : q);   // a halo-column thread: element 0 is pixel hq of its strip
      const bool masked = bar_w > 0 && ((bar_s == 2 && x < bar_w) || (bar_s == 1 && x >= W - bar_w));
      if (masked) pk[q] = 0u;
    }
    if (WIDE && halo_px) gb[sy][4 * (ss - 1) + hq] = pk[0];
    else *reinterpret_cast<uint4*>(&gb[sy][4 * (ss - 1)]) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
  }
  __syncthreads();
  VD_STAMP(ff_stamps, 4, false);
  ff_epilogue<FF_TH, FF_NT>(gb, a, fc, eye, x0, y0, gx0, gy0, tid, out);
  VD_STAMP(ff_stamps, 5, true);
  VD_OCC_OUT(ff_occ);
}

#ifdef VD3D_DEV_KNOBS   // round 6: the parked persistent variant (measured 7 % slower, profiles/r05_e1_persistent.md) is built only into development libraries
// ================================================================================================================================
// Round 5: E1 as a PERSISTENT kernel (k_finish_fused_p; dense levels, 64x26 geometry: the default route; k_finish_fused above stays for the
// separable levels, fit factor 4 and as the A/B reference, vd3d_debug_tune(6, 0)).  What round 4's stamps showed: 28 % of a workgroup's life is the
// tile load -- a global round trip nothing overlaps, three workgroups per CU notwithstanding -- and the launch span is set by the CUs that drew
// 45 tiles while others drew 33.  Here:
//   * one workgroup per CU slot (3 x CUs workgroups) walks its tiles -- virtual block ids b, b + grid, b + 2 grid ... through the same XCD
//     row-group order as before (a multiple of 8 apart: every workgroup stays on its XCD's tile rows), so every slot gets the same number of
//     tiles (+- 1) spread over the whole picture;
//   * the NEXT tile's raw bytes travel by LDS-DMA (global_load_lds_dwordx3: 12 bytes = 4 BGR pixels per lane, no registers) into a staging
//     buffer that aliases the halo-window buffer -- dead between the levels and the next tile's windows -- while the current tile's epilogue
//     (sharpen + fit + mux + stores) runs; the next iteration converts them LDS -> planar float tile without a global round trip.  A tile at
//     the image border (reflect padding per pixel) is loaded the old way at the top of its iteration.
// Arithmetic: the same device functions in the same order as k_finish_fused<true, 26>: identical bytes.
// ================================================================================================================================
VD_STAMP_DECL(ffp_stamps);
template <int TH_>
__global__ __launch_bounds__(ff_geo<TH_>::NT) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_finish_fused_p(const uint8_t* __restrict__ eyeL, const uint8_t* __restrict__ eyeR,
                                                          const float* __restrict__ dn, vd_finish_consts fc, vd_ff_args a,
                                                          const vd_dev_work* __restrict__ w, const float* __restrict__ w2g_,
                                                          uint8_t* __restrict__ out, int nvb, int* __restrict__ ctr) {
  static_assert(ff_geo<TH_>::WIDE, "persistent kernel: wide geometry only");
  constexpr int FF_TH = ff_geo<TH_>::TH, FF_GH = ff_geo<TH_>::GH, FF_IH = ff_geo<TH_>::IH, FF_NT = ff_geo<TH_>::NT;
  constexpr int FF_HS = ff_geo<TH_>::HS, FF_HC = ff_geo<TH_>::HC, FF_NSW = ff_geo<TH_>::NSW;
  constexpr int NTASK = FF_IH * FF_IS;                     // 12-byte groups of the input tile: 36 x 20 = 720
#ifndef FF_DMA12
#define FF_DMA12 0   // 0 (default; measured correct): three dword DMAs per group into three planes.  1: one 12-byte DMA per group -- its LDS image is NOT 12 bytes per lane on gfx950 (byte compare fails from the first staged tile on)   // (global_load_lds_dwordx3, LDS image = 12 bytes per lane); 0: three dword DMAs per group into three planes
#endif
  constexpr int STG_P = ((NTASK + 63) / 64) * 64;          // plane pitch of the dword-plane layout
  constexpr int STAGE_RAW = FF_DMA12 ? NTASK * 3 : 3 * STG_P;
  constexpr int STAGE_F = (STAGE_RAW > 3 * FF_HC ? STAGE_RAW : 3 * FF_HC);   // floats: the raw tile (8 640 B) and the halo windows (8 064 B) share this buffer
  static_assert(NTASK <= 2 * FF_NT, "two tile-load tasks per thread");
  __shared__ __attribute__((aligned(16))) float tile[3][FF_IH][FF_IW];
  __shared__ __attribute__((aligned(16))) uint32_t gb[FF_GH][FF_GP];
  __shared__ __attribute__((aligned(16))) float hal[STAGE_F];
  __shared__ __attribute__((aligned(16))) int lvl_mask[4];   // [0]: levels any pixel of this tile needs (16 bytes: keeps the arrays 16-byte aligned)
  uint32_t* stage = reinterpret_cast<uint32_t*>(hal);
  const int H0 = a.H, W0 = a.W;
  const float w_focal = a.use_override ? a.focal : w->focal;
  const int w_bar_w = a.use_override ? a.bar_w : w->bar_width, w_bar_s = a.use_override ? a.bar_s : w->bar_side;
  const bool src_ok = (W0 & 3) == 0 && ((reinterpret_cast<uintptr_t>(eyeL) | reinterpret_cast<uintptr_t>(eyeR)) & 3) == 0;

  // tile of virtual block vb (-1: a padding block of the XCD order); interior = 12-byte-group loads / LDS-DMA possible
  auto decode = [&](int vb, int* eye, int* x0, int* y0) -> bool {
    int trow, tbx;
    vd_xcd_tile_rows(vb, a.ntx, FF_XG, a.xcd, &trow, &tbx);
    if (trow >= 2 * a.nty) return false;
    *eye = trow >= a.nty ? 1 : 0;
    *x0 = tbx * FF_TW; *y0 = (trow - *eye * a.nty) * FF_TH;
    return true;
  };
  auto interior = [&](int x0, int y0) -> bool {
    const int ix0 = x0 - 4 - FF_R, iy0 = y0 - 1 - FF_R;
    return src_ok && ix0 >= 0 && ix0 + FF_IW <= W0 && iy0 >= 0 && iy0 + FF_IH <= H0;
  };

  // DYNAMIC tile queue: the first tile of a workgroup is its own block index, every further one comes from a launch-wide counter (ctr[0], starts at
  // 0; virtual block = gridDim.x + ticket).  A static interleave was measured 20 % SLOWER than the one-tile kernel: tiles cost 16 k .. 53 k cycles
  // depending on the DOF levels they need, and the launch ends with the unluckiest of 768 slots.  ctr[1] counts finished workgroups; the last one
  // zeroes both words for the launch that gets this counter pair next (the host hands out pairs round-robin, so concurrent launches on other
  // streams never share one).
  auto next_ticket = [&](int* eye_, int* x0_, int* y0_) -> int {   // called by ONE lane
    int v;
    do { v = (int)gridDim.x + atomicAdd(ctr, 1); } while (v < nvb && !decode(v, eye_, x0_, y0_));
    return v;
  };
  auto leave = [&]() { if (threadIdx.x == 0 && atomicAdd(ctr + 1, 1) == (int)gridDim.x - 1) { atomicExch(ctr, 0); atomicExch(ctr + 1, 0); } };
  int vb = blockIdx.x;
  int eye = 0, x0 = 0, y0 = 0;
  if (!decode(vb, &eye, &x0, &y0)) {                       // a padding block of the XCD order: take a ticket (workgroup-uniform branch)
    if (threadIdx.x == 0) { int e_, x_, y_; lvl_mask[1] = next_ticket(&e_, &x_, &y_); }
    __syncthreads();
    vb = lvl_mask[1];
    __syncthreads();
    if (vb >= nvb || !decode(vb, &eye, &x0, &y0)) { leave(); return; }
  }
  bool staged = false;                                     // this tile's raw bytes are in `stage` (put there by the previous iteration)
  VD_STAMP(ffp_stamps, 0, false);
  while (true) {
    // The thread mapping (see k_finish_fused) is tile-invariant, and so is everything derived from it -- LDS addresses of the tile-load tasks, of the
    // strips, of the epilogue's tasks: left alone, the optimiser hoists all of it out of the tile loop into ~40 registers that the dense levels
    // need (122 VGPRs instead of 80 = two workgroups per CU instead of three).  An opaque per-iteration copy of the thread index makes it
    // recompute them per tile: a few dozen integer operations.
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    tid &= 1023;
    // ... and the same for the frame geometry: the scalar expressions derived from it (tap scales, border tests, 64-bit row pitches) otherwise pile
    // up in SGPRs that spill into VGPR lanes
    int H = H0, W = W0;
    vd_ff_args ai = a;
    asm volatile("" : "+s"(H), "+s"(W), "+s"(ai.eh), "+s"(ai.ew));
    asm volatile("" : "+s"(ai.fit_w), "+s"(ai.fit_h), "+s"(ai.in_w), "+s"(ai.in_h), "+s"(ai.xo), "+s"(ai.yo), "+s"(ai.fx), "+s"(ai.fy), "+s"(ai.out_w), "+s"(ai.format));
    ai.H = H; ai.W = W;
    // (uniform values that need VALU instructions -- 1 / (fx fy), the blur weight's division by fc.fw, integer divisions by run-time constants -- are
    // hoisted as whole VGPRs otherwise: a register per scalar)
    float f_fw = fc.fw, f_imax = fc.imax, f_sat = fc.sat, f_con = fc.con, f_bri = fc.bri, f_kn = fc.sharp_kn, f_kc = fc.sharp_kc;
    int f_nlev = fc.nlev;
    asm volatile("" : "+s"(f_fw), "+s"(f_imax), "+s"(f_sat), "+s"(f_con), "+s"(f_bri), "+s"(f_kn), "+s"(f_kc), "+s"(f_nlev));
    uint8_t* outp = out;
    const uint8_t* eL = eyeL; const uint8_t* eR = eyeR; const float* dnp = dn;
    int zo2 = 0;
    asm volatile("" : "+s"(zo2));
    outp += zo2; eL += zo2; eR += zo2; dnp += zo2;
    const int lane = tid & 63, wv = tid >> 6;
    const int wv_u = __builtin_amdgcn_readfirstlane(wv);
    bool strip = true, halo_px = false;
    int sy, ss, hq = 0;
    if (wv < FF_NSW) { sy = 4 * wv + (lane & 3); ss = 2 + (lane >> 2); }
    else {
      sy = min(lane >> 1, FF_GH - 1); halo_px = true; ss = (lane & 1) ? FF_IS - 2 : 1; hq = (lane & 1) ? 0 : 3;
      strip = (lane >> 1) < FF_GH;
    }
    const uint8_t* __restrict__ src = eye == 0 ? eL : eR;
    const int gx0 = x0 - 4, gy0 = y0 - 1;
    const int ix0 = gx0 - FF_R, iy0 = gy0 - FF_R;
    const int gy = gy0 + sy, gxs = ix0 + 4 * ss;
    int zoff = 0;
    asm volatile("" : "+s"(zoff));                         // per-iteration opaque zero: the weight table's scalar loads are not hoisted out of the tile loop
    const float* w2g = w2g_ + zoff;
    // depth samples of the blur weight: requested now, consumed after the first barrier
    const int yc = min(max(gy, 0), H - 1);
    const bool fast21 = strip && f_nlev && 2 * ai.eh == H && 2 * ai.ew == W && gxs >= 4 && gxs + 6 <= W;
    vd_tap ay21 = {0, 0, 0.f, 0.f};
    float p0[4] = {0.f, 0.f, 0.f, 0.f}, p1[4] = {0.f, 0.f, 0.f, 0.f};
    if (fast21) {
      ay21 = vd_tap21(ai.eh, yc);
      const int c0 = (gxs >> 1) - 1;
      const float* r0 = dnp + (size_t)ay21.i0 * ai.ew + c0;
      const float* r1 = dnp + (size_t)ay21.i1 * ai.ew + c0;
#pragma unroll
      for (int j = 0; j < 4; ++j) { p0[j] = r0[j]; p1[j] = r1[j]; }
    }
    if (tid == 0) lvl_mask[0] = 0;
    const bool in_interior = interior(x0, y0);
    if (staged || in_interior) {
      uint32_t ld[2][3] = {{0u, 0u, 0u}, {0u, 0u, 0u}};
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int t = tid + k * FF_NT;
        if (t < NTASK) {
          if (staged) {
            if (FF_DMA12) { ld[k][0] = stage[3 * t]; ld[k][1] = stage[3 * t + 1]; ld[k][2] = stage[3 * t + 2]; }
            else { ld[k][0] = stage[t]; ld[k][1] = stage[STG_P + t]; ld[k][2] = stage[2 * STG_P + t]; }
          }
          else {
            const int ty = t / FF_IS, g = t - ty * FF_IS;
            const uint32_t* pp = reinterpret_cast<const uint32_t*>(src + ((size_t)(iy0 + ty) * W + ix0 + 4 * g) * 3);
            ld[k][0] = pp[0]; ld[k][1] = pp[1]; ld[k][2] = pp[2];
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int t = tid + k * FF_NT;
        if (t < NTASK) {
          const int ty = t / FF_IS, g = t - ty * FF_IS;
          const uint32_t a0 = ld[k][0], a1 = ld[k][1], a2 = ld[k][2];
#define FF_B(k) ff_byte((k) < 4 ? a0 : ((k) < 8 ? a1 : a2), 8 * ((k) & 3))
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const int bo = 2 - c;
            const vd_f4 v4 = {vd_u8_unit(FF_B(0 + bo)), vd_u8_unit(FF_B(3 + bo)), vd_u8_unit(FF_B(6 + bo)), vd_u8_unit(FF_B(9 + bo))};
            *reinterpret_cast<vd_f4*>(&tile[c][ty][4 * g]) = v4;
          }
#undef FF_B
        }
      }
    } else {
      for (int t = tid; t < FF_IH * FF_IW; t += FF_NT) {
        const int ty = t / FF_IW, tx = t - ty * FF_IW;
        const int y = vd_reflect(iy0 + ty, H), x = vd_reflect(ix0 + tx, W);
        const uint8_t* px = src + ((size_t)y * W + x) * 3;
        tile[0][ty][tx] = vd_u8_unit((float)px[2]);
        tile[1][ty][tx] = vd_u8_unit((float)px[1]);
        tile[2][ty][tx] = vd_u8_unit((float)px[0]);
      }
    }
    // this workgroup's NEXT tile: the ticket is requested here (a device-scope atomic: ~2 us) and looked at only before the third barrier
    int ticket = nvb;
    if (tid == 0) ticket = (int)gridDim.x + atomicAdd(ctr, 1);
    int lo[4] = {0, 0, 0, 0};
    vd_f4 alpha = {0.f, 0.f, 0.f, 0.f};
    int lmin = 9, lmax = -1;
    __syncthreads();   // lvl_mask = 0 visible; tile complete; the staging buffer has been consumed (it becomes the halo-window buffer now)
    VD_STAMP(ffp_stamps, 1, false);
    for (int t = tid; t < 3 * 2 * FF_IH; t += FF_NT) {   // halo-column windows (consumed after the next barrier)
      const int c = t / (2 * FF_IH), rem = t - c * (2 * FF_IH), side = rem / FF_IH, row = rem - side * FF_IH;
      const float* srcp = &tile[c][row][side ? (4 * (FF_IS - 2)) - FF_R : (4 + 3) - FF_R];
      float* dstp = &hal[c * FF_HC + side * FF_HS + row * 9];
#pragma unroll
      for (int j = 0; j < 9; ++j) dstp[j] = srcp[j];
    }
    int my_mask = 0;
    if (strip && f_nlev) {
      const float focal = w_focal;
      float dd[4];
      if (fast21) {
        const vd_tap ay = ay21;
        const float a0 = vd_fma(p0[0], 0.25f, 0.75f * p0[1]), b0 = vd_fma(p1[0], 0.25f, 0.75f * p1[1]);
        const float a1 = vd_fma(p0[1], 0.75f, 0.25f * p0[2]), b1 = vd_fma(p1[1], 0.75f, 0.25f * p1[2]);
        const float a2 = vd_fma(p0[1], 0.25f, 0.75f * p0[2]), b2 = vd_fma(p1[1], 0.25f, 0.75f * p1[2]);
        const float a3 = vd_fma(p0[2], 0.75f, 0.25f * p0[3]), b3 = vd_fma(p1[2], 0.75f, 0.25f * p1[3]);
        dd[0] = vd_fma(a0, ay.w0, ay.w1 * b0); dd[1] = vd_fma(a1, ay.w0, ay.w1 * b1);
        dd[2] = vd_fma(a2, ay.w0, ay.w1 * b2); dd[3] = vd_fma(a3, ay.w0, ay.w1 * b3);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int x = min(max(gxs + q, 0), W - 1);
          if (ai.eh == H && ai.ew == W) dd[q] = dnp[(size_t)yc * W + x];
          else if (2 * ai.eh == H && 2 * ai.ew == W) {
            const vd_tap ay = vd_tap21(ai.eh, yc), ax = vd_tap21(ai.ew, x);
            const float* r0 = dnp + (size_t)ay.i0 * ai.ew;
            const float* r1 = dnp + (size_t)ay.i1 * ai.ew;
            dd[q] = vd_bilerp(r0[ax.i0], r0[ax.i1], r1[ax.i0], r1[ax.i1], ax.w0, ax.w1, ay.w0, ay.w1);
          } else {
            const vd_tap ay = vd_interp_tap(ai.eh, H, yc), ax = vd_interp_tap(ai.ew, W, x);
            const float* r0 = dnp + (size_t)ay.i0 * ai.ew;
            const float* r1 = dnp + (size_t)ay.i1 * ai.ew;
            dd[q] = vd_bilerp(r0[ax.i0], r0[ax.i1], r1[ax.i0], r1[ax.i1], ax.w0, ax.w1, ay.w0, ay.w1);
          }
        }
      }
      const bool fastfw = f_fw == FF_FW_STD;
      const float rcfw = 1.0f / FF_FW_STD;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float ad = fabsf(dd[q] - focal);
        float qv;
        if (fastfw && (ad >= 1e-30f || ad == 0.f)) {
          const float q0 = ad * rcfw;
          qv = vd_fma(vd_fma(-q0, FF_FW_STD, ad), rcfw, q0);
        } else qv = ad / f_fw;
        const float bw = vd_clamp_fin(qv, 0.f, 1.f);
        const float bi = vd_clamp_fin(bw * (float)f_nlev, 0.f, f_imax);
        int l = (int)floorf(bi);
        l = l > f_nlev - 1 ? f_nlev - 1 : (l < 0 ? 0 : l);
        lo[q] = l; alpha[q] = bi - (float)l;
        if (!halo_px || q == hq) { lmin = min(lmin, l); lmax = max(lmax, l + 1); }
      }
      if (halo_px && hq == 3) { lo[0] = lo[3]; alpha[0] = alpha[3]; }
      for (int l = max(lmin, 1); l <= lmax; ++l) my_mask |= 1 << l;
    }
    {
      int wmask = 0;
#pragma unroll
      for (int l = 1; l <= 4; ++l)
        if (__ballot((my_mask >> l) & 1)) wmask |= 1 << l;
      if ((tid & 63) == 0 && wmask) atomicOr(&lvl_mask[0], wmask);
    }
    vd_f4 vres[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      vres[c] = *reinterpret_cast<const vd_f4*>(&tile[c][sy + FF_R][4 * ss]);
      if (halo_px && hq == 3) vres[c][0] = vres[c][3];
    }
    __syncthreads();
    VD_STAMP(ffp_stamps, 2, false);
    const int need_mask = lvl_mask[0];
    for (int l = 0; l < f_nlev; ++l) {
      if (!(need_mask >> (l + 1) & 1)) continue;
      const int off = FF_R - fc.ksz[l] / 2;
      const bool mine = (my_mask >> (l + 1)) & 1;
      if (wv == FF_NSW) {   // wave-uniform: the halo-pixel wave
        const int side = lane & 1;
        switch (off) {
          case 0: ff_level_dense_px<0, FF_HS>(hal, w2g + 81 * l, mine, sy, side, l + 1, lo[0], alpha[0], vres); break;
          case 1: ff_level_dense_px<1, FF_HS>(hal, w2g + 81 * l, mine, sy, side, l + 1, lo[0], alpha[0], vres); break;
          case 2: ff_level_dense_px<2, FF_HS>(hal, w2g + 81 * l, mine, sy, side, l + 1, lo[0], alpha[0], vres); break;
          default: ff_level_dense_px<3, FF_HS>(hal, w2g + 81 * l, mine, sy, side, l + 1, lo[0], alpha[0], vres); break;
        }
        continue;
      }
      switch (off) {
        case 0: ff_level_dense<0, FF_IH>(tile, w2g + 81 * l, mine, sy, ss, l + 1, lo, alpha, vres); break;
        case 1: ff_level_dense<1, FF_IH>(tile, w2g + 81 * l, mine, sy, ss, l + 1, lo, alpha, vres); break;
        case 2: ff_level_dense<2, FF_IH>(tile, w2g + 81 * l, mine, sy, ss, l + 1, lo, alpha, vres); break;
        default: ff_level_dense<3, FF_IH>(tile, w2g + 81 * l, mine, sy, ss, l + 1, lo, alpha, vres); break;
      }
    }
    VD_STAMP(ffp_stamps, 3, false);
    if (strip) {
      const int bar_w = w_bar_w, bar_s = w_bar_s;
      vd_f4 rgbv[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        vd_f4 v = vres[c];
        if (f_nlev) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = vd_clamp_fin(v[q], 0.f, 1.f);
        }
        rgbv[c] = v;
      }
      const vd_f4 luma = ((float)0.2126 * rgbv[0] + (float)0.7152 * rgbv[1]) + (float)0.0722 * rgbv[2];
      uint32_t pk[4] = {0, 0, 0, 0};
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        vd_f4 v = luma + (rgbv[c] - luma) * f_sat;
        v = 0.5f + (v - 0.5f) * f_con;
        v = v + f_bri;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float vc = vd_clamp_fin(v[q], 0.f, 1.f);
          pk[q] |= (uint32_t)(uint8_t)(vc * 255.0f) << (8 * (2 - c));
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int x = gxs + (halo_px ? hq : q);
        const bool masked = bar_w > 0 && ((bar_s == 2 && x < bar_w) || (bar_s == 1 && x >= W - bar_w));
        if (masked) pk[q] = 0u;
      }
      if (halo_px) gb[sy][4 * (ss - 1) + hq] = pk[0];
      else *reinterpret_cast<uint4*>(&gb[sy][4 * (ss - 1)]) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    }
    if (tid == 0) {
      int e_, x_, y_;
      while (ticket < nvb && !decode(ticket, &e_, &x_, &y_)) ticket = (int)gridDim.x + atomicAdd(ctr, 1);   // padding blocks of the XCD order (rare)
      lvl_mask[1] = ticket;
    }
    __syncthreads();   // graded tile complete; the levels are done: the halo-window buffer is dead and takes the next tile's raw bytes
    VD_STAMP(ffp_stamps, 4, false);
    // next tile of this workgroup; its raw bytes by LDS-DMA while the epilogue below runs
    int nvb_ = lvl_mask[1], neye = 0, nx0 = 0, ny0 = 0;
    nvb_ = __builtin_amdgcn_readfirstlane(nvb_);
    const bool more = nvb_ < nvb && decode(nvb_, &neye, &nx0, &ny0);
    const bool nstaged = more && interior(nx0, ny0);       // workgroup-uniform
    if (nstaged) {
      const uint8_t* __restrict__ nsrc = neye == 0 ? eL : eR;
      const int nix0 = nx0 - 4 - FF_R, niy0 = ny0 - 1 - FF_R;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int t = tid + k * FF_NT;
        if (t < NTASK) {   // lanes past the last group stay inactive: the DMA writes LDS at M0 + 12 * lane for ACTIVE lanes only
          const int ty = t / FF_IS, g = t - ty * FF_IS;
          const uint8_t* gp = nsrc + ((size_t)(niy0 + ty) * W + nix0 + 4 * g) * 3;
#if FF_DMA12
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                           (__attribute__((address_space(3))) void*)(stage + 3 * (k * FF_NT + wv_u * 64)), 12, 0, 0);
#else
#pragma unroll
          for (int j = 0; j < 3; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + 4 * j),
                                             (__attribute__((address_space(3))) void*)(stage + j * STG_P + k * FF_NT + wv_u * 64), 4, 0, 0);
#endif
        }
      }
    }
    ff_epilogue_k<FF_TH, FF_NT>(gb, ai, f_kn, f_kc, eye, x0, y0, gx0, gy0, tid, outp);
    VD_STAMP(ffp_stamps, 5, true);
    if (!more) { leave(); break; }
    vb = nvb_; eye = neye; x0 = nx0; y0 = ny0; staged = nstaged;
    __syncthreads();   // epilogue done with gb / every wave past its levels; the DMA has landed (the compiler drains vmcnt before the barrier)
  }
}
#endif   // VD3D_DEV_KNOBS

// The epilogue alone, for eyes that are graded already (k_dof_grade4's planes: Gaussians beyond the fused kernel's 9 taps): tile of graded dwords
// straight from the u8 planes, then sharpen + fit + mux as above.  Replaces k_sharp_mux (one thread per output pixel, 120 byte loads each: 236 us
// per 4K frame pair) wherever the fused kernel's fit conditions hold.
template <int TH_>
__global__ __launch_bounds__(512) void k_sharp_fit(const uint8_t* __restrict__ gL, const uint8_t* __restrict__ gR, vd_finish_consts fc, vd_ff_args a,
                                                   uint8_t* __restrict__ out) {
  constexpr int FF_TH = TH_, FF_GH = TH_ + 2, FF_NT = 512;
  __shared__ __attribute__((aligned(16))) uint32_t gb[FF_GH][FF_GP];
  int trow, tbx;
  vd_xcd_tile_rows(blockIdx.x, a.ntx, FF_XG, a.xcd, &trow, &tbx);
  if (trow >= 2 * a.nty) return;
  const int eye = trow >= a.nty ? 1 : 0, tby = trow - eye * a.nty;
  const uint8_t* __restrict__ src = eye == 0 ? gL : gR;
  const int H = a.H, W = a.W;
  const int x0 = tbx * FF_TW, y0 = tby * FF_TH, gx0 = x0 - 4, gy0 = y0 - 1;
  const int tid = threadIdx.x;
  {   // all of a thread's pixels in one load batch (round 6; the one-pixel-per-iteration loop was four to five dependent global round trips)
    constexpr int NPX = (FF_GH * FF_GW + FF_NT - 1) / FF_NT;
    uint8_t bb[NPX][3];
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
      const int t = min(tid + j * FF_NT, FF_GH * FF_GW - 1);
      const int row = t / FF_GW, col = t - row * FF_GW;
      const uint8_t* px = src + ((size_t)vd_reflect(gy0 + row, H) * W + vd_reflect(gx0 + col, W)) * 3;
      bb[j][0] = px[0]; bb[j][1] = px[1]; bb[j][2] = px[2];
    }
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
      const int t = tid + j * FF_NT;
      if (t < FF_GH * FF_GW) {
        const int row = t / FF_GW, col = t - row * FF_GW;
        gb[row][col] = (uint32_t)bb[j][0] | ((uint32_t)bb[j][1] << 8) | ((uint32_t)bb[j][2] << 16);   // byte 0 = B, 1 = G, 2 = R (the planes are BGR)
      }
    }
  }
  __syncthreads();
  ff_epilogue<FF_TH, FF_NT>(gb, a, fc, eye, x0, y0, gx0, gy0, tid, out);
}

// returns false when the fast path does not apply (caller runs the unfused kernels)
// fit / mux geometry of the epilogue; false: a format or fit it does not take (VR, fractional / up-scaling INTER_AREA)
static bool ff_geometry(const vd3d_render_params& p, int eh, int ew, vd_ff_args* pa) {
  if (!(p.format == VD3D_FMT_HALF_SBS || p.format == VD3D_FMT_FULL_SBS || p.format == VD3D_FMT_INTERLACED || p.format == VD3D_FMT_ANAGLYPH)) return false;
  vd_ff_args& a = *pa;
  a.H = p.warp_h; a.W = p.warp_w; a.eh = eh; a.ew = ew;
  a.fit_w = p.fit_w; a.fit_h = p.fit_h; a.out_w = p.out_w; a.format = p.format;
  if (p.format == VD3D_FMT_HALF_SBS) { a.in_w = p.fit_w; a.in_h = p.fit_h; a.xo = 0; a.yo = 0; }
  else {
    const double ta = (double)p.fit_w / p.fit_h, ca = (double)p.warp_w / p.warp_h;
    if (ca > ta) { a.in_w = p.fit_w; a.in_h = (int)(p.fit_w / ca); }
    else { a.in_h = p.fit_h; a.in_w = (int)(ca * p.fit_h); }
    a.xo = (p.fit_w - a.in_w) / 2; a.yo = (p.fit_h - a.in_h) / 2;
  }
  if (a.in_w < 1 || a.in_h < 1 || a.in_w > p.warp_w || a.in_h > p.warp_h || p.warp_w % a.in_w || p.warp_h % a.in_h) return false;   // fractional / up-scaling INTER_AREA
  a.fx = p.warp_w / a.in_w; a.fy = p.warp_h / a.in_h;
  if (!((a.fx == 1 || a.fx == 2 || a.fx == 4) && (a.fy == 1 || a.fy == 2 || a.fy == 4))) return false;
  if ((FF_TW / a.fx) % 4) return false;
  a.use_override = 0; a.bar_w = 0; a.bar_s = 0; a.focal = 0.f;
  return true;
}
static int g_ff_xcd = 1;   // vd3d_debug_tune(8, 0): plain row-major tile order (workgroup b on XCD b % 8: neighbouring tiles on different XCDs)
void vd_set_finish_xcd(int on) { g_ff_xcd = on ? 1 : 0; }
static void ff_grid(const vd3d_render_params& p, int th, vd_ff_args* a, dim3* g) {
  a->ntx = (p.warp_w + FF_TW - 1) / FF_TW; a->nty = (p.warp_h + th - 1) / th;
  a->xcd = g_ff_xcd;
  const int ngrp = (2 * a->nty + FF_XG - 1) / FF_XG;
  *g = dim3(a->xcd ? 8 * ((ngrp + 7) / 8) * FF_XG * a->ntx : 2 * a->ntx * a->nty);
}
static void ff_clear_canvas(hipStream_t s, const vd3d_render_params& p, const vd_ff_args& a, uint8_t* out) {
  if (a.xo || a.yo || a.in_w != p.fit_w || a.in_h != p.fit_h)  // pad_to_aspect_ratio canvas (:124): black background
    (void)hipMemsetAsync(out, 0, (size_t)p.out_w * p.out_h * 3, s);
}
// sharpen + fit + mux of two GRADED eyes (k_sharp_fit); false: the fit is not the epilogue's (caller runs k_sharp_mux)
bool vd_launch_sharp_fit(hipStream_t s, const uint8_t* gL, const uint8_t* gR, const vd3d_render_params& p, const vd_finish_consts& fc, uint8_t* out) {
  vd_ff_args a;
  if (!ff_geometry(p, p.warp_h, p.warp_w, &a)) return false;
  constexpr int TH = 28;   // multiple of 4 (fit factor 4), 30 x 76 dwords of LDS
  ff_clear_canvas(s, p, a, out);
  dim3 g;
  ff_grid(p, TH, &a, &g);
  hipLaunchKernelGGL((k_sharp_fit<TH>), g, dim3(512), 0, s, gL, gR, fc, a, out);
  return true;
}

// 0: one tile per workgroup (rounds 2 - 4); k > 0: the persistent kernel with k workgroups per CU (vd3d_debug_tune(6, k))
// Default 0: measured (profiles/r05_e1_persistent.md) the persistent kernel is 7 % SLOWER than the one-tile kernel at three workgroups per CU (299.7 vs 280.3 us
// per 4K frame pair, byte-identical) -- the hardware's own dispatcher already balances the 16 k .. 53 k-cycle tiles dynamically, and keeping the tile loop
// inside 80 VGPRs costs re-materialised constants, per-tile recomputation of the thread mapping and 8 spilled dwords.  Kept as an opt-in for the evidence.
#ifndef FF_PERSIST
#define FF_PERSIST 0
#endif
static int g_ff_persist = FF_PERSIST;
void vd_set_finish_persist(int k) { g_ff_persist = k < 0 ? 0 : (k > 8 ? 8 : k); }
bool vd_launch_finish_fused(hipStream_t s, const uint8_t* L, const uint8_t* R, const float* dn, int eh, int ew,
                            const vd3d_render_params& p, const vd_finish_consts& fc, const vd_dev_work* w, float focal,
                            int use_override, int bar_w, int bar_s, uint8_t* out, int dense, const float* w2_dev) {
  for (int l = 0; l < fc.nlev; ++l) if (fc.ksz[l] > 2 * FF_R + 1 || fc.ksz[l] < 3) return false;
  vd_ff_args a;
  if (!ff_geometry(p, eh, ew, &a)) return false;
  a.use_override = use_override; a.bar_w = bar_w; a.bar_s = bar_s; a.focal = focal;
  ff_clear_canvas(s, p, a, out);
  // wide geometry: 64x30 tiles.  (64x14 tiles -- 5 waves, 4 workgroups per CU -- measured 414 vs 426 us at 4K with 14 % more instructions
  // and a 2.1x instead of 1.7x input halo: not kept.)  Fit factor 4 keeps the 64x16 geometry.
#ifdef FF_AB_GEO16   // A/B build (tools/build_ab.sh): dense levels in the 64x16 geometry everywhere
  const bool wide = false;
#else
  const bool wide = dense && (FF_WIDE_TH % a.fy) == 0;
#endif
  const int th = wide ? FF_WIDE_TH : 16;
  dim3 g;
  ff_grid(p, th, &a, &g);
  if (dense && !w2_dev) return false;
#ifdef VD3D_DEV_KNOBS
  if (wide && g_ff_persist && a.xcd) {
    // persistent route: one workgroup per CU slot; three 51.7 KB workgroups of 512 threads fit a CU (80 VGPRs).  The grid is a multiple of 8, so a
    // workgroup's virtual blocks b, b + grid, ... stay on its XCD's tile rows.
    static int n_cu[64] = {};
    static std::mutex init_mu;   // first use per device is serialised: two host threads with a context each may arrive together
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> init_lock(init_mu);
    if (dev >= 0 && dev < 64 && !n_cu[dev] && hipDeviceGetAttribute(&n_cu[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n_cu[dev] = 256;
    const int ncu = (dev >= 0 && dev < 64 && n_cu[dev] > 0) ? n_cu[dev] : 256;
    const int nvb = (int)g.x;
    int slots = g_ff_persist * ncu;                 // vd3d_debug_tune(6, k): k workgroups per CU (default 3)
    slots -= slots % 8;
    static int* ctr_ring[64] = {};   // per device: 256 counter pairs, all zero; a launch takes the next pair and leaves it zeroed (see the kernel)
    static unsigned ctr_next[64] = {};
    if (dev >= 0 && dev < 64 && !ctr_ring[dev]) {
      if (hipMalloc((void**)&ctr_ring[dev], 256 * 2 * sizeof(int)) != hipSuccess || hipMemset(ctr_ring[dev], 0, 256 * 2 * sizeof(int)) != hipSuccess) ctr_ring[dev] = nullptr;
    }
    if (slots >= 8 && nvb > slots && dev >= 0 && dev < 64 && ctr_ring[dev]) {
      int* ctr = ctr_ring[dev] + 2 * (__atomic_fetch_add(&ctr_next[dev], 1u, __ATOMIC_RELAXED) & 255u);
      hipLaunchKernelGGL((k_finish_fused_p<FF_WIDE_TH>), dim3(slots), dim3(ff_geo<FF_WIDE_TH>::NT), 0, s, L, R, dn, fc, a, w, w2_dev, out, nvb, ctr);
      return true;
    }
  }
#endif
  if (wide) hipLaunchKernelGGL((k_finish_fused<true, FF_WIDE_TH>), g, dim3(ff_geo<FF_WIDE_TH>::NT), 0, s, L, R, dn, fc, a, w, w2_dev, out);
  else if (dense) hipLaunchKernelGGL((k_finish_fused<true, 16>), g, dim3(ff_geo<16>::NT), 0, s, L, R, dn, fc, a, w, w2_dev, out);
  else hipLaunchKernelGGL((k_finish_fused<false, 16>), g, dim3(ff_geo<16>::NT), 0, s, L, R, dn, fc, a, w, w2_dev, out);
  return true;
}
