// vd3d_conv.hip -- the body layers of the up-scale network: 3x3 convolution, 64 -> 64 channels, stride 1, zero
// padding 1, + bias + PReLU, fp16 in / fp16 out / float32 accumulate, on the gfx950 matrix cores.  realesr-general-x4v3 is 32 of these
// layers back to back (animevideov3: 16) at input resolution; they hold > 95 % of the network's flops.
//
// Implicit GEMM per workgroup: D[oc][pixel] = sum over (tap, ic) of Wt[oc][tap, ic] * X[pixel + tap][ic]
//   M = 64 output channels (2 MFMA tiles), N = 512 pixels = a 32 x 16 output tile (16 MFMA tiles), K = 9 taps x 64 channels = 36 steps of 16
//   v_mfma_f32_32x32x16_f16: A = weights (lane&31 = output channel, lane>>5 = which 8 of the step's 16 input channels),
//                            B = pixels  (lane&31 = pixel of a 32-pixel row, lane>>5 = the same 8-channel group),
//                            C/D: col = lane&31 = pixel, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) = output channel of the tile
//   4 waves, wave w owns tile rows 4w .. 4w+3 (4 pixel tiles x 2 channel tiles = 8 accumulators = 128 VGPRs)
// LDS: the (32+2) x (16+2) x 64 fp16 input tile, CHUNK-major: 8 planes (one per 8-channel / 16-byte chunk) of 612 pixels x 16 B, so that
//   the 32 lanes of a B-fragment read (consecutive pixels, one chunk) hit consecutive 16-byte slots -- conflict-free ds_read_b128 --
//   and every element is reused by the 9 taps and both channel tiles from LDS.  78.3 KB -> 2 workgroups per CU (one loads while one
//   multiplies).  Plane pitch padded to 10 016 B so the 8 chunks of one pixel land 2 slots apart on the write side.
// Weights: pre-arranged on the host in fragment order [step 36][channel tile 2][lane 64][8 halves] (73.7 KB per layer, shared by every
//   workgroup, L2-resident); each wave streams its two A fragments per step straight from global memory, one step ahead of use.
// Epilogue: bias + PReLU in float32, convert, transpose through LDS (pixel-major, 144-byte pitch), 16-byte coalesced NHWC stores.
// Bounds (960 x 540, one layer): 38.2 GFLOP -> 15 us at the 2.5 PFLOP/s dense fp16 peak; HBM 66 MB in (x 1.2 halo) + 66 MB out -> 18 us.
// Measured (tools/probe_conv.py): 54 us = 705 TFLOP/s = 28 % of the fp16 peak, 3.3x MIOpen's convolution + PReLU pair.  The three phases
// of a workgroup (tile load at ~11 B/clk/CU, 288 MFMAs, store tail at ~7 B/clk/CU) run back to back and the two workgroups of a CU move
// in lock-step, so the time is their SUM; a 32 x 8 tile with 3 workgroups per CU (-DCV_TH=8) measures the same.  Next lever: a persistent
// workgroup that issues the next tile's global loads before the MFMA loop (registers are there at CV_TH = 8), then two layers per launch.
#include "vd3d_dev.h"
#include "vd3d_kernels.h"
#include <hip/hip_fp16.h>

typedef _Float16 cv_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 cv_h4 __attribute__((ext_vector_type(4)));
typedef float cv_f16 __attribute__((ext_vector_type(16)));

#define CV_TW 32
#ifndef CV_TH
#define CV_TH 16
#endif
#define CV_RW (CV_TH / 4)             // tile rows per wave
#define CV_PW (CV_TW + 2)
#define CV_PH (CV_TH + 2)
#define CV_NPIX (CV_PW * CV_PH)          // 612
#define CV_PLANE ((CV_NPIX * 16 + 255) / 256 * 256 + 32)   // bytes per chunk plane, padded to == 32 (mod 256): 10 016 for 16 rows
#define CV_LDS (8 * CV_PLANE)            // 80 128 B
#define CV_OP 144                        // epilogue: bytes per pixel (128 + 16 pad)

__global__ __launch_bounds__(256, CV_TH == 16 ? 2 : 3) void k_conv3x3_c64(const _Float16* __restrict__ x, int H, int W, const uint4* __restrict__ wfrag,
                                                        const float* __restrict__ bias, const float* __restrict__ slope,
                                                        _Float16* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) uint8_t cv_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, g = lane >> 5;
  const int x0 = blockIdx.x * CV_TW, y0 = blockIdx.y * CV_TH;

  // A (weight) fragments of steps 0 and 1: independent of the tile, issued before the tile load.  Software pipeline of the main loop:
  // weights two steps ahead (L2 latency), pixel fragments one step ahead (LDS latency); sched_barrier keeps the compiler from hoisting
  // every load of the unrolled loop to the top (which spills).
  const uint4* wp = wfrag + lane;
  uint4 aw[3][2];
  aw[0][0] = wp[0]; aw[0][1] = wp[64];
  aw[1][0] = wp[128]; aw[1][1] = wp[192];

  // input tile (+1 halo, zero outside the image) -> LDS; one task = one 16-byte chunk of one pixel.  Thread = (chunk tid&7, pixel
  // (tid>>3) + 32k): ten loads in flight per thread before the first LDS write (two batches cover the 612 pixels).
  {
    const int c = tid & 7, p0 = tid >> 3;
#pragma unroll
    for (int h = 0; h < (CV_NPIX + 319) / 320; ++h) {
      uint4 v[10];
#pragma unroll
      for (int k = 0; k < 10; ++k) {
        const int pix = p0 + 32 * (10 * h + k);
        const int py = pix / CV_PW, px = pix - py * CV_PW;
        const int gy = y0 - 1 + py, gx = x0 - 1 + px;
        v[k] = make_uint4(0u, 0u, 0u, 0u);
        if (pix < CV_NPIX && gy >= 0 && gy < H && gx >= 0 && gx < W) v[k] = *reinterpret_cast<const uint4*>(x + ((size_t)gy * W + gx) * 64 + c * 8);
      }
#pragma unroll
      for (int k = 0; k < 10; ++k) {
        const int pix = p0 + 32 * (10 * h + k);
        if (pix < CV_NPIX) *reinterpret_cast<uint4*>(cv_lds + c * CV_PLANE + pix * 16) = v[k];
      }
    }
  }
  __syncthreads();

  cv_f16 acc[CV_RW][2];
#pragma unroll
  for (int m = 0; m < CV_RW; ++m)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;

  const uint8_t* bbase = cv_lds + g * CV_PLANE + ((wave * CV_RW) * CV_PW + li) * 16;
  cv_h8 bf[2][CV_RW];
#pragma unroll
  for (int m = 0; m < CV_RW; ++m) bf[0][m] = *reinterpret_cast<const cv_h8*>(bbase + m * CV_PW * 16);
#pragma unroll
  for (int s = 0; s < 36; ++s) {
    if (s + 2 < 36) { aw[(s + 2) % 3][0] = wp[(s + 2) * 128]; aw[(s + 2) % 3][1] = wp[(s + 2) * 128 + 64]; }
    if (s + 1 < 36) {
      const int tap = (s + 1) >> 2, kc = (s + 1) & 3, dy = tap / 3, dx = tap - 3 * dy;
      const uint8_t* bp = bbase + (2 * kc) * CV_PLANE + (dy * CV_PW + dx) * 16;
#pragma unroll
      for (int m = 0; m < CV_RW; ++m) bf[(s + 1) & 1][m] = *reinterpret_cast<const cv_h8*>(bp + m * CV_PW * 16);
    }
    const cv_h8 wa0 = __builtin_bit_cast(cv_h8, aw[s % 3][0]), wa1 = __builtin_bit_cast(cv_h8, aw[s % 3][1]);
#pragma unroll
    for (int m = 0; m < CV_RW; ++m) {
      acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa0, bf[s & 1][m], acc[m][0], 0, 0, 0);
      acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa1, bf[s & 1][m], acc[m][1], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();   // every wave is done with the input tile: LDS becomes the output staging buffer

  // bias + PReLU (float32), fp16, pixel-major staging: lane = pixel li of row (wave*4 + m); regs 4q..4q+3 = channels 32t + 8q + 4g + (0..3)
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch = 32 * t + 8 * q + 4 * g;
      const float4 bv = *reinterpret_cast<const float4*>(bias + ch);
      float4 sv = make_float4(1.f, 1.f, 1.f, 1.f);
      if (slope) sv = *reinterpret_cast<const float4*>(slope + ch);
#pragma unroll
      for (int m = 0; m < CV_RW; ++m) {
        float v0 = acc[m][t][4 * q] + bv.x, v1 = acc[m][t][4 * q + 1] + bv.y, v2 = acc[m][t][4 * q + 2] + bv.z, v3 = acc[m][t][4 * q + 3] + bv.w;
        v0 = v0 >= 0.f ? v0 : v0 * sv.x; v1 = v1 >= 0.f ? v1 : v1 * sv.y; v2 = v2 >= 0.f ? v2 : v2 * sv.z; v3 = v3 >= 0.f ? v3 : v3 * sv.w;
        const cv_h4 hv = {(_Float16)v0, (_Float16)v1, (_Float16)v2, (_Float16)v3};
        *reinterpret_cast<cv_h4*>(cv_lds + ((wave * CV_RW + m) * CV_TW + li) * CV_OP + ch * 2) = hv;
      }
    }
  __syncthreads();
  for (int t = tid; t < CV_TW * CV_TH * 8; t += 256) {
    const int c = t & 7, pix = t >> 3;
    const int py = pix / CV_TW, px = pix - py * CV_TW;
    const int gy = y0 + py, gx = x0 + px;
    if (gy < H && gx < W)
      *reinterpret_cast<uint4*>(y + ((size_t)gy * W + gx) * 64 + c * 8) = *reinterpret_cast<const uint4*>(cv_lds + pix * CV_OP + c * 16);
  }
}


bool vd_launch_conv3x3_c64_f16(hipStream_t s, const void* x, int H, int W, const void* wfrag, const float* bias, const float* slope_or_null,
                               void* y) {
  static bool attr_set[64] = {};      // per device: the > 64 KB dynamic-LDS opt-in is a per-device function attribute
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  if (!attr_set[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3x3_c64), hipFuncAttributeMaxDynamicSharedMemorySize, CV_LDS) != hipSuccess)
      return false;
    attr_set[dev] = true;
  }
  dim3 grid((W + CV_TW - 1) / CV_TW, (H + CV_TH - 1) / CV_TH);
  hipLaunchKernelGGL(k_conv3x3_c64, grid, dim3(256), CV_LDS, s, (const _Float16*)x, H, W, (const uint4*)wfrag, bias, slope_or_null, (_Float16*)y);
  return true;
}

// ------------------------------------------------------------------------------------------------
// head and tail of the compact (SRVGG) up-scale networks (round 3): until now the only layers left on MIOpen.
//
// k_conv3x3_head: 3 -> 64 convolution + bias + PReLU, fp16 NHWC [H][W][3] in, fp16 NHWC [H][W][64] out, float32 accumulate in the order
//   (kh, kw, ic).  1.8 GFLOP at 960 x 540: far too little for the matrix cores to matter -- the layer is the 66 MB it writes.  A workgroup
//   = a 64 x 2 pixel tile; thread = (strip of 4 pixels, group of 8 output channels): 32 accumulators, the 27 x 64 float32 weights and the
//   (2 + 2) x 66 x 3 input patch in LDS; the 8 threads of a pixel store its 128 bytes as consecutive 16-byte pieces.
// k_esr_tail: the tail after its 64 -> 48 (x4) / 64 -> 12 (x2) convolution -- which IS the MFMA kernel above, with the weight fragments and
//   the bias zero-padded to 64 output channels and no activation -- pixel_shuffle(r) + the nearest-neighbour up-sampled input + conversion
//   to the float32 planar prediction the reference's post-processing takes (core/merged_pipeline.py:225-229):
//   out[c][h r + i][w r + j] = float(half(float(t[h][w][c r^2 + i r + j]) + float(x[h][w][c])))   (the fp16 addition torch performs).
//   One workgroup = 64 pixels of one input row, staged through LDS so that the loads are 16-byte and the stores whole float32 rows.
// ------------------------------------------------------------------------------------------------
#define CH_TW 64
#define CH_TH 2
__global__ __launch_bounds__(256) void k_conv3x3_head(const _Float16* __restrict__ x, int H, int W, const float* __restrict__ w27, const float* __restrict__ bias,
                                                      const float* __restrict__ slope, _Float16* __restrict__ y) {
  __shared__ __attribute__((aligned(16))) float lw[27][64];
  __shared__ float lp[CH_TH + 2][CH_TW + 2][3];
  const int tid = threadIdx.x, g = tid & 7, strip = tid >> 3;       // strip 0..31: row strip >> 4, columns 4 (strip & 15) ..
  const int x0 = blockIdx.x * CH_TW, y0 = blockIdx.y * CH_TH;
  for (int i = tid; i < 27 * 64; i += 256) (&lw[0][0])[i] = w27[i];
  for (int i = tid; i < (CH_TH + 2) * (CH_TW + 2) * 3; i += 256) {
    const int c = i % 3, px = (i / 3) % (CH_TW + 2), py = i / (3 * (CH_TW + 2));
    const int gy = y0 - 1 + py, gx = x0 - 1 + px;
    (&lp[0][0][0])[i] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? (float)x[((size_t)gy * W + gx) * 3 + c] : 0.f;
  }
  __syncthreads();
  const int ty = strip >> 4, tx = (strip & 15) * 4;
  float acc[4][8];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[q][o] = 0.f;
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    float row[6][3];                     // the 6 x 3 input values of this tap row that the strip's 4 pixels x 3 tap columns touch
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
      for (int c = 0; c < 3; ++c) row[p][c] = lp[ty + kh][tx + p][c];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int t = (kh * 3 + kw) * 3 + c;
        const vd_f4 wa = *reinterpret_cast<const vd_f4*>(&lw[t][8 * g]), wb = *reinterpret_cast<const vd_f4*>(&lw[t][8 * g + 4]);
        const float wv[8] = {wa[0], wa[1], wa[2], wa[3], wb[0], wb[1], wb[2], wb[3]};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int o = 0; o < 8; ++o) acc[q][o] = vd_fma(row[q + kw][c], wv[o], acc[q][o]);
      }
  }
  const int gy = y0 + ty;
  if (gy >= H) return;
  float bv[8], sv[8];
#pragma unroll
  for (int o = 0; o < 8; ++o) { bv[o] = bias[8 * g + o]; sv[o] = slope ? slope[8 * g + o] : 1.f; }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int gx = x0 + tx + q;
    if (gx >= W) break;
    cv_h8 hv;
#pragma unroll
    for (int o = 0; o < 8; ++o) { float v = acc[q][o] + bv[o]; v = v >= 0.f ? v : v * sv[o]; hv[o] = (_Float16)v; }
    *reinterpret_cast<cv_h8*>(y + ((size_t)gy * W + gx) * 64 + 8 * g) = hv;
  }
}

__global__ __launch_bounds__(256) void k_esr_tail(const _Float16* __restrict__ t, const _Float16* __restrict__ x, int H, int W, int r, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) _Float16 lt[64][72];     // 64 pixels x 64 channels (+8 pad: 144-byte pitch)
  __shared__ _Float16 lx[64][4];
  const int tid = threadIdx.x, h = blockIdx.y, x0 = blockIdx.x * 64;
  const int npx = min(64, W - x0);
  for (int i = tid; i < 64 * 8; i += 256) {            // 16-byte pieces: pixel i >> 3, channels 8 (i & 7) ..
    const int p = i >> 3, cg = i & 7;
    if (p < npx) *reinterpret_cast<uint4*>(&lt[p][8 * cg]) = *reinterpret_cast<const uint4*>(t + ((size_t)h * W + x0 + p) * 64 + 8 * cg);
  }
  if (tid < 64 * 3) { const int p = tid / 3, c = tid - 3 * p; if (p < npx) lx[p][c] = x[((size_t)h * W + x0 + p) * 3 + c]; }
  __syncthreads();
  const size_t OW = (size_t)W * r, plane = (size_t)H * r * OW;
  const int p = tid / r, j = tid - p * r;               // output column x0 r + tid of this segment = input pixel p, sub-column j
  if (tid >= 64 * r || p >= npx) return;
  for (int c = 0; c < 3; ++c)
    for (int i = 0; i < r; ++i) {
      const _Float16 s = (_Float16)((float)lt[p][c * r * r + i * r + j] + (float)lx[p][c]);    // the fp16 add of `out + interpolate(x)`
      out[c * plane + ((size_t)h * r + i) * OW + (size_t)x0 * r + tid] = (float)s;
    }
}

void vd_launch_conv3x3_head_f16(hipStream_t s, const void* x, int H, int W, const float* w27, const float* bias, const float* slope_or_null, void* y) {
  hipLaunchKernelGGL(k_conv3x3_head, dim3((W + CH_TW - 1) / CH_TW, (H + CH_TH - 1) / CH_TH), dim3(256), 0, s, (const _Float16*)x, H, W, w27, bias, slope_or_null,
                     (_Float16*)y);
}
void vd_launch_esr_tail_f32(hipStream_t s, const void* t, const void* x, int H, int W, int r, float* out) {
  hipLaunchKernelGGL(k_esr_tail, dim3((W + 63) / 64, H), dim3(256), 0, s, (const _Float16*)t, (const _Float16*)x, H, W, r, out);
}
