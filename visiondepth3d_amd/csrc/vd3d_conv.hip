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
#include <mutex>

typedef _Float16 cv_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 cv_h4 __attribute__((ext_vector_type(4)));
typedef float cv_f16 __attribute__((ext_vector_type(16)));

// development aid (-DVD_PHASE_STAMPS: tools/build_ab.sh stamps): thread 0 of every 16th workgroup (64 slots) records s_memrealtime (100 MHz) at the
// phase boundaries of its first two tiles; tools/gpu_ab.bin conv prints the phase lengths
#ifdef VD_PHASE_STAMPS
static __device__ unsigned long long cv_stamps[64][16];
extern "C" __attribute__((visibility("default"))) int vd3d_debug_stamps_conv(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(cv_stamps), sizeof(cv_stamps)); }
#define CV_STAMP(lin, k) do { if (threadIdx.x == 0 && (lin) % 16 == 5 && (lin) / 16 < 64 && (k) < 16) cv_stamps[(lin) / 16][k] = __builtin_amdgcn_s_memrealtime(); } while (0)
static __device__ unsigned long long cv_where[1024][4];   // persistent kernel, every workgroup: CU id, slot parity, start, end (s_memrealtime)
extern "C" __attribute__((visibility("default"))) int vd3d_debug_where_conv(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(cv_where), sizeof(cv_where)); }
#define CV_WHERE(k, v) do { if (threadIdx.x == 0 && blockIdx.x < 1024) cv_where[blockIdx.x][k] = (v); } while (0)
#else
#define CV_STAMP(lin, k) do { } while (0)
#define CV_WHERE(k, v) do { } while (0)
#endif
// weight (A) fragments are requested CV_WD steps ahead of their MFMAs.
// (Measured with 2, 4, 6 and 8: no difference -- the loop is not waiting for the weights; 2 keeps the registers.)
#ifndef CV_WD
#define CV_WD 2
#endif
#define CV_TW 32
#ifndef CV_TH
#define CV_TH 16
#endif
#define CV_RW (CV_TH / 4)             // tile rows per wave
#define CV_PW (CV_TW + 2)
#define CV_PH (CV_TH + 2)
#define CV_NPIX (CV_PW * CV_PH)          // 612
#define CV_PLANE ((CV_NPIX * 16 + 255) / 256 * 256 + 32)   // bytes per chunk plane, padded to == 32 (mod 256): 10 016 for 16 rows
#define CV_LDS_TILE (8 * CV_PLANE)       // 80 128 B
#define CV_LDS (CV_LDS_TILE + 512)       // + bias[64], slope[64] as float32 (round 5): the epilogue read them from global memory, 16 dependent L2 round
                                         // trips per wave = 4.3 of a workgroup's 21 us (phase stamps, profiles/r05_conv_phases.md); 80 640 B, still two per CU
#define CV_OP 144                        // epilogue: bytes per pixel (128 + 16 pad)

__global__ __launch_bounds__(256, CV_TH == 16 ? 2 : 3) void k_conv3x3_c64(const _Float16* __restrict__ x, int H, int W, const uint4* __restrict__ wfrag,
                                                        const float* __restrict__ bias, const float* __restrict__ slope,
                                                        _Float16* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) uint8_t cv_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, g = lane >> 5;
  const int x0 = blockIdx.x * CV_TW, y0 = blockIdx.y * CV_TH;
  const int cv_lin = blockIdx.y * gridDim.x + blockIdx.x;
  (void)cv_lin;   // (phase-stamp builds only)
  CV_STAMP(cv_lin, 0);

  // A (weight) fragments of steps 0 and 1: independent of the tile, issued before the tile load.  Software pipeline of the main loop:
  // weights two steps ahead (L2 latency), pixel fragments one step ahead (LDS latency); sched_barrier keeps the compiler from hoisting
  // every load of the unrolled loop to the top (which spills).
  const uint4* wp = wfrag + lane;
  uint4 aw[CV_WD + 1][2];
#pragma unroll
  for (int d = 0; d < CV_WD; ++d) { aw[d][0] = wp[d * 128]; aw[d][1] = wp[d * 128 + 64]; }

  // input tile (+1 halo, zero outside the image) -> LDS; one task = one 16-byte chunk of one pixel.  Thread = (chunk tid&7, pixel
  // (tid>>3) + 32k): ten loads in flight per thread before the first LDS write (two batches cover the 612 pixels).  (Round 5, measured: ONE
  // batch of twenty loads -- the accumulators are not live yet -- is SLOWER, 51.2 vs 47.6 us per layer: every CU of the chip requests its
  // tile at the same moment and the phase is bound by the fabric, not by round trips; profiles/r05_conv_phases.md.)
  {
    const int c = tid & 7, p0 = tid >> 3;
    if (tid < 128) reinterpret_cast<float*>(cv_lds + CV_LDS_TILE)[tid] = tid < 64 ? bias[tid] : (slope ? slope[tid - 64] : 1.f);
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
  CV_STAMP(cv_lin, 1);

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
    if (s + CV_WD < 36) { aw[(s + CV_WD) % (CV_WD + 1)][0] = wp[(s + CV_WD) * 128]; aw[(s + CV_WD) % (CV_WD + 1)][1] = wp[(s + CV_WD) * 128 + 64]; }
    if (s + 1 < 36) {
      const int tap = (s + 1) >> 2, kc = (s + 1) & 3, dy = tap / 3, dx = tap - 3 * dy;
      const uint8_t* bp = bbase + (2 * kc) * CV_PLANE + (dy * CV_PW + dx) * 16;
#pragma unroll
      for (int m = 0; m < CV_RW; ++m) bf[(s + 1) & 1][m] = *reinterpret_cast<const cv_h8*>(bp + m * CV_PW * 16);
    }
    const cv_h8 wa0 = __builtin_bit_cast(cv_h8, aw[s % (CV_WD + 1)][0]), wa1 = __builtin_bit_cast(cv_h8, aw[s % (CV_WD + 1)][1]);
#pragma unroll
    for (int m = 0; m < CV_RW; ++m) {
      acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa0, bf[s & 1][m], acc[m][0], 0, 0, 0);
      acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa1, bf[s & 1][m], acc[m][1], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();   // every wave is done with the input tile: LDS becomes the output staging buffer
  CV_STAMP(cv_lin, 2);

  // bias + PReLU (float32), fp16, pixel-major staging: lane = pixel li of row (wave*4 + m); regs 4q..4q+3 = channels 32t + 8q + 4g + (0..3)
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch = 32 * t + 8 * q + 4 * g;
      const float4 bv = *reinterpret_cast<const float4*>(cv_lds + CV_LDS_TILE + ch * 4);        // bias / slope from LDS (see CV_LDS)
      const float4 sv = *reinterpret_cast<const float4*>(cv_lds + CV_LDS_TILE + 256 + ch * 4);  // slope 1 = no activation: v * 1.0f is v
#pragma unroll
      for (int m = 0; m < CV_RW; ++m) {
        float v0 = acc[m][t][4 * q] + bv.x, v1 = acc[m][t][4 * q + 1] + bv.y, v2 = acc[m][t][4 * q + 2] + bv.z, v3 = acc[m][t][4 * q + 3] + bv.w;
        v0 = v0 >= 0.f ? v0 : v0 * sv.x; v1 = v1 >= 0.f ? v1 : v1 * sv.y; v2 = v2 >= 0.f ? v2 : v2 * sv.z; v3 = v3 >= 0.f ? v3 : v3 * sv.w;
        const cv_h4 hv = {(_Float16)v0, (_Float16)v1, (_Float16)v2, (_Float16)v3};
        *reinterpret_cast<cv_h4*>(cv_lds + ((wave * CV_RW + m) * CV_TW + li) * CV_OP + ch * 2) = hv;
      }
    }
  __syncthreads();
  CV_STAMP(cv_lin, 3);
  for (int t = tid; t < CV_TW * CV_TH * 8; t += 256) {
    const int c = t & 7, pix = t >> 3;
    const int py = pix / CV_TW, px = pix - py * CV_TW;
    const int gy = y0 + py, gx = x0 + px;
    if (gy < H && gx < W)
      *reinterpret_cast<uint4*>(y + ((size_t)gy * W + gx) * 64 + c * 8) = *reinterpret_cast<const uint4*>(cv_lds + pix * CV_OP + c * 16);
  }
  CV_STAMP(cv_lin, 4);
}



#ifdef VD3D_DEV_KNOBS   // round 6: the parked persistent variant (no gain, profiles/r05_conv_phases.md) is built only into development libraries
// ================================================================================================================================
// Round 5: the same layer as a PERSISTENT kernel (k_conv3x3_c64_p, the default; k_conv3x3_c64 above stays as the A/B reference and for
// activations too small to fill the chip).  What round 2 - 4 measured: a workgroup's three phases -- tile load (HBM read), 288 MFMAs per wave,
// output store (HBM write) -- run back to back, the two workgroups of a CU start together and have identical phase lengths, so they stay in
// LOCK-STEP through the whole launch: every CU of the chip loads at the same time (HBM-bound: 11 B/clk/CU), then every CU multiplies (HBM
// idle), then every CU stores (7 B/clk/CU): 54 - 59 us = the SUM of the phases.  Here:
//   * one workgroup per CU slot (2 x CUs workgroups) walks its tiles in a loop; an XCD owns one contiguous band of tiles (vd_xcd_tile order),
//     so the halo rows shared by vertically adjacent tiles are found in the XCD's own L2;
//   * PHASE SKEW: a workgroup learns which of its CU's two slots it got (one atomic on a per-CU arrival counter, HW_ID / XCC_ID) and the
//     second arrival starts `skew` later -- about half a tile period -- so that on every CU one workgroup is in its MFMA phase while the other
//     one moves data, for the whole launch (the loops have equal length, the offset persists);
//   * the NEXT tile's global loads are issued at the start of the store phase, when the accumulators are dead (20 x 16 bytes per thread in
//     registers), and land in LDS after it: a workgroup's own load and store phases overlap as well.
// Arithmetic, tile shape, fragment order and the LDS layout are those of k_conv3x3_c64: the two kernels produce identical bytes.
// ================================================================================================================================
#define CV_NLD ((CV_NPIX * 8 + 255) / 256)     // 16-byte chunks of the input tile per thread: 20 (19.1)
// All CV_NLD loads of a thread are requested back to back, branch-free: a position outside the image (zero padding) or beyond the tile reads a
// clamped, valid address and is zeroed by cv_zero_outside when the values are consumed.  `tid` arrives through an opaque copy so that the
// per-position index arithmetic is redone per tile (two dozen integer operations) instead of being hoisted out of the tile loop into 40 live registers.
VD_DEV void cv_issue_tile_loads(const _Float16* __restrict__ x, int H, int W, int x0, int y0, int tid_, uint4 (&v)[CV_NLD], bool live = true) {
  int tid = tid_;
  asm volatile("" : "+v"(tid));
  const int c = tid & 7, p0 = (tid >> 3) & 31;   // (& 31: the opaque copy hides the range of tid; with it the bounds tests below fold for all but the last position)
#pragma unroll
  for (int k = 0; k < CV_NLD; ++k) {
    const int pix = min(p0 + 32 * k, CV_NPIX - 1);
    const int py = pix / CV_PW, px = pix - py * CV_PW;
    const int gy = min(max(y0 - 1 + py, 0), H - 1), gx = min(max(x0 - 1 + px, 0), W - 1);
    // `live` false (workgroup-uniform: no tile left): every lane re-reads the image's first pixel -- one cached line, no HBM traffic -- and the
    // values are never used; a branch around these loads instead costs the register allocator its view of the 80 prefetch registers (24 dwords spilled)
    const size_t pixel = live ? (size_t)gy * W + gx : (size_t)0;
    v[k] = *reinterpret_cast<const uint4*>(x + pixel * 64 + c * 8);
  }
}
// ... and written to the chunk-major LDS planes (zero outside the image)
VD_DEV void cv_store_tile_lds(uint8_t* cv_lds, int H, int W, int x0, int y0, int tid_, const uint4 (&v)[CV_NLD]) {
  int tid = tid_;
  asm volatile("" : "+v"(tid));
  const int c = tid & 7, p0 = (tid >> 3) & 31;   // (& 31: the opaque copy hides the range of tid; with it the bounds tests below fold for all but the last position)
#pragma unroll
  for (int k = 0; k < CV_NLD; ++k) {
    const int pix = p0 + 32 * k;
    const int py = pix / CV_PW, px = pix - py * CV_PW;
    const int gy = y0 - 1 + py, gx = x0 - 1 + px;
    const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
    uint4 u = v[k];
    u.x = in ? u.x : 0u; u.y = in ? u.y : 0u; u.z = in ? u.z : 0u; u.w = in ? u.w : 0u;
    if (pix < CV_NPIX) *reinterpret_cast<uint4*>(cv_lds + c * CV_PLANE + pix * 16) = u;
  }
}

__global__ __launch_bounds__(256, 2) void k_conv3x3_c64_p(const _Float16* __restrict__ x, int H, int W, const uint4* __restrict__ wfrag,
                                                          const float* __restrict__ bias, const float* __restrict__ slope,
                                                          _Float16* __restrict__ y, int ntx, int ntiles, int per, unsigned* __restrict__ cu_cnt,
                                                          unsigned skew_ticks) {
  static_assert(CV_TH == 16, "the persistent kernel is written for the 32 x 16 tile (two workgroups per CU)");
  extern __shared__ __attribute__((aligned(16))) uint8_t cv_lds[];
  const int tid = threadIdx.x;
  const int xcd = blockIdx.x & 7, wpx = gridDim.x >> 3;      // workgroup b runs on XCD b % 8; wpx workgroups per XCD
  int it = blockIdx.x >> 3;                                  // index inside the XCD's band of `per` tiles
  if (it >= per || xcd * per + it >= ntiles) return;         // nothing to do (workgroup-uniform, before any barrier)

  if (skew_ticks) {   // which of the CU's two slots is this?  The second arrival starts half a tile period later (see the header).
    if (tid == 0) {
      const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
      const unsigned cu = ((xcc & 7u) << 8) | ((hw >> 8) & 0xffu);      // HW_ID 15:8 = SE | SH | CU
      const unsigned par = atomicAdd(&cu_cnt[cu], 1u) & 1u;
      *reinterpret_cast<unsigned*>(cv_lds) = par;
      CV_WHERE(0, (unsigned long long)cu); CV_WHERE(1, (unsigned long long)par);
    }
    __syncthreads();
    const unsigned second = *reinterpret_cast<const unsigned*>(cv_lds);
    if (second) {
      if (tid == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
        while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)skew_ticks) __builtin_amdgcn_s_sleep(32);
      }
    }
    __syncthreads();   // also: the LDS word above is dead before the first tile lands
  }

  if (tid < 128) reinterpret_cast<float*>(cv_lds + CV_LDS_TILE)[tid] = tid < 64 ? bias[tid] : (slope ? slope[tid - 64] : 1.f);   // once per workgroup
  int cv_iter = 0;
  CV_STAMP(blockIdx.x, 0);
  CV_WHERE(2, __builtin_amdgcn_s_memrealtime());
  uint4 v[CV_NLD];
  {
    const int tile = xcd * per + it, tby = tile / ntx, tbx = tile - tby * ntx;
    cv_issue_tile_loads(x, H, W, tbx * CV_TW, tby * CV_TH, tid, v);
  }
  while (true) {
    const int tile = xcd * per + it, tby = tile / ntx, tbx = tile - tby * ntx;
    const int x0 = tbx * CV_TW, y0 = tby * CV_TH;
    // per-tile copies of the thread coordinates behind an opaque move: everything derived from them (LDS fragment addresses, staging offsets) is
    // recomputed per tile -- a few dozen integer operations -- instead of being hoisted out of the tile loop into registers the MFMA phase needs
    int tid_i = tid;
    asm volatile("" : "+v"(tid_i));
    const int lane = tid_i & 63, wave = tid_i >> 6, li = lane & 31, g = lane >> 5;
    const uint4* wp = wfrag + lane;
    cv_store_tile_lds(cv_lds, H, W, x0, y0, tid, v);   // the prefetched tile -> LDS (chunk-major planes, see the header of this file)
    __builtin_amdgcn_sched_barrier(0);                 // the 80 prefetch registers die here: nothing of the MFMA phase (accumulator zeroing) moves above
    // the weight / bias / slope streams are the same for every tile: without these opaque copies the optimiser hoists all of their loads out of
    // the tile loop (72 + 16 sixteen-byte registers per lane) and spills
    // (an opaque ZERO OFFSET, not an opaque pointer: a pointer that went through an asm statement loses its address space and every load through
    // it becomes a flat_load, which counts on both wait counters and turns the loop's counted waits into vmcnt(0) lgkmcnt(0))
    int zoff = 0;
    asm volatile("" : "+s"(zoff));
    const uint4* wpi = wp + zoff;
    uint4 aw[CV_WD + 1][2];
#pragma unroll
    for (int d = 0; d < CV_WD; ++d) { aw[d][0] = wpi[d * 128]; aw[d][1] = wpi[d * 128 + 64]; }
    __syncthreads();
    CV_STAMP(blockIdx.x, 8 * cv_iter + 1);

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
      if (s + CV_WD < 36) { aw[(s + CV_WD) % (CV_WD + 1)][0] = wpi[(s + CV_WD) * 128]; aw[(s + CV_WD) % (CV_WD + 1)][1] = wpi[(s + CV_WD) * 128 + 64]; }
      if (s + 1 < 36) {
        const int tap = (s + 1) >> 2, kc = (s + 1) & 3, dy = tap / 3, dx = tap - 3 * dy;
        const uint8_t* bp = bbase + (2 * kc) * CV_PLANE + (dy * CV_PW + dx) * 16;
#pragma unroll
        for (int m = 0; m < CV_RW; ++m) bf[(s + 1) & 1][m] = *reinterpret_cast<const cv_h8*>(bp + m * CV_PW * 16);
      }
      const cv_h8 wa0 = __builtin_bit_cast(cv_h8, aw[s % (CV_WD + 1)][0]), wa1 = __builtin_bit_cast(cv_h8, aw[s % (CV_WD + 1)][1]);
#pragma unroll
      for (int m = 0; m < CV_RW; ++m) {
        acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa0, bf[s & 1][m], acc[m][0], 0, 0, 0);
        acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa1, bf[s & 1][m], acc[m][1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();   // every wave is done with the input tile: LDS becomes the output staging buffer
    CV_STAMP(blockIdx.x, 8 * cv_iter + 2);

#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ch = 32 * t + 8 * q + 4 * g;
        const float4 bv = *reinterpret_cast<const float4*>(cv_lds + CV_LDS_TILE + ch * 4);
        const float4 sv = *reinterpret_cast<const float4*>(cv_lds + CV_LDS_TILE + 256 + ch * 4);
#pragma unroll
        for (int m = 0; m < CV_RW; ++m) {
          float v0 = acc[m][t][4 * q] + bv.x, v1 = acc[m][t][4 * q + 1] + bv.y, v2 = acc[m][t][4 * q + 2] + bv.z, v3 = acc[m][t][4 * q + 3] + bv.w;
          v0 = v0 >= 0.f ? v0 : v0 * sv.x; v1 = v1 >= 0.f ? v1 : v1 * sv.y; v2 = v2 >= 0.f ? v2 : v2 * sv.z; v3 = v3 >= 0.f ? v3 : v3 * sv.w;
          const cv_h4 hv = {(_Float16)v0, (_Float16)v1, (_Float16)v2, (_Float16)v3};
          *reinterpret_cast<cv_h4*>(cv_lds + ((wave * CV_RW + m) * CV_TW + li) * CV_OP + ch * 2) = hv;
        }
      }
    __syncthreads();
    CV_STAMP(blockIdx.x, 8 * cv_iter + 3);
    // the accumulators are dead: request the next tile now, so that its loads travel while this tile's stores are issued
    it += wpx;
    const bool more = it < per && xcd * per + it < ntiles;   // workgroup-uniform
    {   // unconditional, see cv_issue_tile_loads: no control flow around 80 live registers
      const int nt = more ? xcd * per + it : tile, nby = nt / ntx, nbx = nt - nby * ntx;
      cv_issue_tile_loads(x, H, W, nbx * CV_TW, nby * CV_TH, tid, v, more);
    }
    for (int t = tid; t < CV_TW * CV_TH * 8; t += 256) {
      const int c = t & 7, pix = t >> 3;
      const int py = pix / CV_TW, px = pix - py * CV_TW;
      const int gy = y0 + py, gx = x0 + px;
      if (gy < H && gx < W)
        *reinterpret_cast<uint4*>(y + ((size_t)gy * W + gx) * 64 + c * 8) = *reinterpret_cast<const uint4*>(cv_lds + pix * CV_OP + c * 16);
    }
    CV_STAMP(blockIdx.x, 8 * cv_iter + 4);
    CV_WHERE(3, __builtin_amdgcn_s_memrealtime());
    if (!more) break;
    __syncthreads();   // the staging buffer has been read: the next tile may land
    ++cv_iter;
    CV_STAMP(blockIdx.x, 8 * cv_iter);
  }
}
#endif   // VD3D_DEV_KNOBS

// ================================================================================================================================
// k_conv3x3_c64_s (round 5, second step): 32 x 8 tiles, THREE workgroups per CU.  What the phase stamps of the 32 x 16 kernels showed
// (profiles/r05_conv_phases.md): a workgroup multiplies for 5 us of a 13 - 22 us tile (the rest: waiting for its tile, bias + PReLU + staging,
// issuing stores, draining them), and with two workgroups per CU the matrix pipe is busy less than half of the time whatever their relative
// phase -- skewing them (k_conv3x3_c64_p, above) or prefetching the weights further ahead changes nothing.  More workgroups per CU is what
// fills the pipe; the obstacle was the accumulator tile: with 8 tile rows a wave of the old mapping owns 2 rows x 2 channel tiles and uses
// every weight fragment only twice -- 2x the L1 traffic of the weight stream, which is what bounds it (the -DCV_TH=8 build of round 3
// measured no gain).  Here a wave owns 4 rows x ONE channel tile (wave = (row half, channel tile)): every weight fragment still feeds 4 MFMAs,
// every pixel fragment is read by two waves instead of one (LDS has the bandwidth: 128 of 256 B/clk at full MFMA rate), 64 accumulator
// registers instead of 128, a 45 KB tile: three workgroups per CU, 12 waves.  Persistent, next tile prefetched into registers during the store
// phase like k_conv3x3_c64_p.  Same arithmetic (each output = the same 36 MFMA steps in the same order): identical bytes.
// ================================================================================================================================
#define CS_TH 8
#define CS_PH (CS_TH + 2)
#define CS_NPIX (CV_PW * CS_PH)          // 340
#define CS_PLANE ((CS_NPIX * 16 + 255) / 256 * 256 + 32)   // 5 664
#define CS_LDS_TILE (8 * CS_PLANE)       // 45 312
#define CS_LDS (CS_LDS_TILE + 512)       // + bias / slope
#define CS_NLD ((CS_NPIX * 8 + 255) / 256)   // 11
VD_DEV void cs_issue_tile_loads(const _Float16* __restrict__ x, int H, int W, int x0, int y0, int tid_, uint4 (&v)[CS_NLD], bool live = true) {
  int tid = tid_;
  asm volatile("" : "+v"(tid));
  const int c = tid & 7, p0 = (tid >> 3) & 31;
#pragma unroll
  for (int k = 0; k < CS_NLD; ++k) {
    const int pix = min(p0 + 32 * k, CS_NPIX - 1);
    const int py = pix / CV_PW, px = pix - py * CV_PW;
    const int gy = min(max(y0 - 1 + py, 0), H - 1), gx = min(max(x0 - 1 + px, 0), W - 1);
    const size_t pixel = live ? (size_t)gy * W + gx : (size_t)0;
    v[k] = *reinterpret_cast<const uint4*>(x + pixel * 64 + c * 8);
  }
}
VD_DEV void cs_store_tile_lds(uint8_t* cv_lds, int H, int W, int x0, int y0, int tid_, const uint4 (&v)[CS_NLD]) {
  int tid = tid_;
  asm volatile("" : "+v"(tid));
  const int c = tid & 7, p0 = (tid >> 3) & 31;
#pragma unroll
  for (int k = 0; k < CS_NLD; ++k) {
    const int pix = p0 + 32 * k;
    const int py = pix / CV_PW, px = pix - py * CV_PW;
    const int gy = y0 - 1 + py, gx = x0 - 1 + px;
    const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
    uint4 u = v[k];
    u.x = in ? u.x : 0u; u.y = in ? u.y : 0u; u.z = in ? u.z : 0u; u.w = in ? u.w : 0u;
    if (pix < CS_NPIX) *reinterpret_cast<uint4*>(cv_lds + c * CS_PLANE + pix * 16) = u;
  }
}

__global__ __launch_bounds__(256, 3) void k_conv3x3_c64_s(const _Float16* __restrict__ x, int H, int W, const uint4* __restrict__ wfrag,
                                                          const float* __restrict__ bias, const float* __restrict__ slope,
                                                          _Float16* __restrict__ y, int ntx, int ntiles, int per) {
  extern __shared__ __attribute__((aligned(16))) uint8_t cv_lds[];
  const int tid = threadIdx.x;
  const int xcd = blockIdx.x & 7, wpx = gridDim.x >> 3;
  int it = blockIdx.x >> 3;
  if (it >= per || xcd * per + it >= ntiles) return;
  if (tid < 128) reinterpret_cast<float*>(cv_lds + CS_LDS_TILE)[tid] = tid < 64 ? bias[tid] : (slope ? slope[tid - 64] : 1.f);
  uint4 v[CS_NLD];
  {
    const int tile = xcd * per + it, tby = tile / ntx, tbx = tile - tby * ntx;
#ifdef CV_DBG_NOLOAD
    cs_issue_tile_loads(x, H, W, tbx * CV_TW, tby * CS_TH, tid, v, false);
#else
    cs_issue_tile_loads(x, H, W, tbx * CV_TW, tby * CS_TH, tid, v);
#endif
  }
  while (true) {
    const int tile = xcd * per + it, tby = tile / ntx, tbx = tile - tby * ntx;
    const int x0 = tbx * CV_TW, y0 = tby * CS_TH;
    int tid_i = tid;
    asm volatile("" : "+v"(tid_i));
    const int lane = tid_i & 63, wave = (tid_i >> 6) & 3, li = lane & 31, g = lane >> 5;
    const int rh = wave >> 1, ct = wave & 1;                 // row half (tile rows 4 rh ..), channel tile (output channels 32 ct ..)
    cs_store_tile_lds(cv_lds, H, W, x0, y0, tid, v);
    __builtin_amdgcn_sched_barrier(0);
    int zoff = 0;
    asm volatile("" : "+s"(zoff));
    const uint4* wpi = wfrag + lane + 64 * ct + zoff;        // this wave's A fragment of step s: wpi[128 s]
    uint4 aw[CV_WD + 1];
#pragma unroll
    for (int d = 0; d < CV_WD; ++d) aw[d] = wpi[d * 128];
    __syncthreads();

    cv_f16 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    const uint8_t* bbase = cv_lds + g * CS_PLANE + ((rh * 4) * CV_PW + li) * 16;
    cv_h8 bf[2][4];
#pragma unroll
    for (int m = 0; m < 4; ++m) bf[0][m] = *reinterpret_cast<const cv_h8*>(bbase + m * CV_PW * 16);
#ifdef CV_DBG_NOMFMA   // timing experiment only (A/B builds): what do the memory phases cost without the multiply phase?
    constexpr int NSTEP = 0;
#else
    constexpr int NSTEP = 36;
#endif
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      if (s + CV_WD < 36) aw[(s + CV_WD) % (CV_WD + 1)] = wpi[(s + CV_WD) * 128];
      if (s + 1 < 36) {
        const int tap = (s + 1) >> 2, kc = (s + 1) & 3, dy = tap / 3, dx = tap - 3 * dy;
        const uint8_t* bp = bbase + (2 * kc) * CS_PLANE + (dy * CV_PW + dx) * 16;
#pragma unroll
        for (int m = 0; m < 4; ++m) bf[(s + 1) & 1][m] = *reinterpret_cast<const cv_h8*>(bp + m * CV_PW * 16);
      }
      const cv_h8 wa = __builtin_bit_cast(cv_h8, aw[s % (CV_WD + 1)]);
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa, bf[s & 1][m], acc[m], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();   // every wave is done with the input tile: LDS becomes the output staging buffer

    // bias + PReLU (float32), fp16, pixel-major staging: lane = pixel li of tile row 4 rh + m; regs 4q..4q+3 = channels 32 ct + 8q + 4g + (0..3)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch = 32 * ct + 8 * q + 4 * g;
      const float4 bv = *reinterpret_cast<const float4*>(cv_lds + CS_LDS_TILE + ch * 4);
      const float4 sv = *reinterpret_cast<const float4*>(cv_lds + CS_LDS_TILE + 256 + ch * 4);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        float v0 = acc[m][4 * q] + bv.x, v1 = acc[m][4 * q + 1] + bv.y, v2 = acc[m][4 * q + 2] + bv.z, v3 = acc[m][4 * q + 3] + bv.w;
        v0 = v0 >= 0.f ? v0 : v0 * sv.x; v1 = v1 >= 0.f ? v1 : v1 * sv.y; v2 = v2 >= 0.f ? v2 : v2 * sv.z; v3 = v3 >= 0.f ? v3 : v3 * sv.w;
        const cv_h4 hv = {(_Float16)v0, (_Float16)v1, (_Float16)v2, (_Float16)v3};
        *reinterpret_cast<cv_h4*>(cv_lds + ((rh * 4 + m) * CV_TW + li) * CV_OP + ch * 2) = hv;
      }
    }
    __syncthreads();
    it += wpx;
    const bool more = it < per && xcd * per + it < ntiles;
    {
      const int nt = more ? xcd * per + it : tile, nby = nt / ntx, nbx = nt - nby * ntx;
#ifdef CV_DBG_NOLOAD
      cs_issue_tile_loads(x, H, W, nbx * CV_TW, nby * CS_TH, tid, v, false);
#else
      cs_issue_tile_loads(x, H, W, nbx * CV_TW, nby * CS_TH, tid, v, more);
#endif
    }
#ifndef CV_DBG_NOSTORE
    for (int t = tid; t < CV_TW * CS_TH * 8; t += 256) {
      const int c = t & 7, pix = t >> 3;
      const int py = pix / CV_TW, px = pix - py * CV_TW;
      const int gy = y0 + py, gx = x0 + px;
      if (gy < H && gx < W)
        *reinterpret_cast<uint4*>(y + ((size_t)gy * W + gx) * 64 + c * 8) = *reinterpret_cast<const uint4*>(cv_lds + pix * CV_OP + c * 16);
    }
#else
    if (tid == 0 && v[0].x == 0x12345678u && cv_lds[tid * 7] == 99) y[0] = (_Float16)1.f;   // keep the loads / the staging alive
#endif
    if (!more) break;
    __syncthreads();
  }
}

// launch policy (vd3d_debug_tune(5, v)).  v = -2 (default): by size -- the 32 x 8 / three-per-CU kernel up to 4 096 of its tiles (1280 x 720: measured
// 5 - 45 % faster than the 32 x 16 kernels from 240 x 135 to 1280 x 720, profiles/r05_conv_phases.md), the one-tile-per-workgroup 32 x 16 kernel
// above that (1920 x 1080: 3 - 10 % faster there).  v = -1: always the one-tile 32 x 16 kernel (rounds 2 - 4); 0 <= v < 100: the persistent
// 32 x 16 kernel with a phase skew of v microseconds (measured: no gain, kept for the evidence); v >= 100: the 32 x 8 kernel with v - 100
// workgroups per CU.
static int g_cv_mode = -2;
void vd_set_conv_mode(int v) { g_cv_mode = v; }

bool vd_launch_conv3x3_c64_f16(hipStream_t s, const void* x, int H, int W, const void* wfrag, const float* bias, const float* slope_or_null,
                               void* y) {
  static bool attr_set[64] = {};      // per device: the > 64 KB dynamic-LDS opt-in is a per-device function attribute
  static unsigned* cu_cnt[64] = {};   // per device: arrival counters of the CU slots (persistent kernel; never reset: only the parity matters)
  static int n_cu[64] = {};
  static std::mutex init_mu;          // one context per host thread is a supported pattern: first use of a device is serialised
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  std::unique_lock<std::mutex> init_lock(init_mu);
  if (!attr_set[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3x3_c64), hipFuncAttributeMaxDynamicSharedMemorySize, CV_LDS) != hipSuccess)
      return false;
#ifdef VD3D_DEV_KNOBS
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3x3_c64_p), hipFuncAttributeMaxDynamicSharedMemorySize, CV_LDS) != hipSuccess)
      return false;
#endif
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3x3_c64_s), hipFuncAttributeMaxDynamicSharedMemorySize, CS_LDS) != hipSuccess)
      return false;
    if (hipDeviceGetAttribute(&n_cu[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu[dev] < 8) return false;
    if (hipMalloc((void**)&cu_cnt[dev], 2048 * sizeof(unsigned)) != hipSuccess) return false;
    if (hipMemset(cu_cnt[dev], 0, 2048 * sizeof(unsigned)) != hipSuccess) return false;
    attr_set[dev] = true;
  }
  init_lock.unlock();
  const int nt8 = ((W + CV_TW - 1) / CV_TW) * ((H + CS_TH - 1) / CS_TH);
  if (g_cv_mode >= 100 || (g_cv_mode == -2 && nt8 <= 4096)) {   // the 32 x 8 / three-workgroups-per-CU kernel (persistent grid)
    const int ntx = (W + CV_TW - 1) / CV_TW, nty = (H + CS_TH - 1) / CS_TH, ntiles = ntx * nty;
    int slots = (g_cv_mode >= 100 ? g_cv_mode - 100 : 3) * n_cu[dev];
    slots -= slots % 8;
    if (slots >= 8) {
      if (slots > ntiles) slots = (ntiles + 7) / 8 * 8;
      const int per = (ntiles + 7) / 8;
      hipLaunchKernelGGL(k_conv3x3_c64_s, dim3(slots), dim3(256), CS_LDS, s, (const _Float16*)x, H, W, (const uint4*)wfrag, bias, slope_or_null,
                         (_Float16*)y, ntx, ntiles, per);
      return true;
    }
  }
  const int ntx = (W + CV_TW - 1) / CV_TW, nty = (H + CV_TH - 1) / CV_TH;
#ifdef VD3D_DEV_KNOBS
  const int ntiles = ntx * nty;
  const int slots = 2 * (n_cu[dev] & ~3);           // two 78 KB workgroups per CU; a multiple of 8: every XCD gets the same number
  if (!(g_cv_mode < 0 || ntiles <= slots)) {        // the parked persistent kernel (phase skew of g_cv_mode microseconds): development libraries only
    const int per = (ntiles + 7) / 8;
    hipLaunchKernelGGL(k_conv3x3_c64_p, dim3(slots), dim3(256), CV_LDS, s, (const _Float16*)x, H, W, (const uint4*)wfrag, bias, slope_or_null,
                       (_Float16*)y, ntx, ntiles, per, cu_cnt[dev], (unsigned)(g_cv_mode * 100));
    return true;
  }
#endif
  dim3 grid(ntx, nty);
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
