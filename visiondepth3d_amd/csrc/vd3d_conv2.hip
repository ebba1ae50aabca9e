// vd3d_conv2.hip -- the 3 x 3 convolutions of the DPT neck / head (stride 1, zero padding 1, no bias: DepthPipe runs them bias-free with one glue launch
// behind each, vd3d_netops.hip) in the fp16x2 arithmetic of the depth leg's opt-in mode (vd3d_gemm.hip: every float32 operand as two fp16 terms, 22 bits,
// three MFMA products per MAC, float32 accumulation; weights pre-scaled per output channel by an exact power of two).  Float32 NHWC (channels_last) in
// and out, like the MIOpen / CK float32 kernels they stand in for (core/render_depth.py:1106-1119 runs the whole model in float32).
//
// Implicit GEMM per workgroup: D[pixel][oc] = sum over (channel chunk, tap) of X[pixel + tap][chunk] W[oc][tap, chunk]
//   M = 512 pixels = a 16 x 32 output tile (16 MFMA M tiles = the tile's rows), N = C_out (64 or 128), K = 9 taps x C_in in steps of 16 channels.
//   512 threads = 8 waves as 4 (M: four tile rows each) x 2 (N), accumulators 4 x NWN tiles (NWN = C_out / 64), two waves per SIMD; for 32 output channels (the
//   head's 64 -> 32 at the full 518 x 924) 8 (M: two rows each) x 1 (N).
//   A (pixels): the input tile + 1 halo ring, ONE 16-channel chunk at a time, is fetched ONCE (LDS-DMA into a float32 staging buffer, a chunk ahead;
//     pixels outside the image fetch a zero page), split into its two fp16 terms by an LDS -> register -> LDS pass behind the fourth tap of the previous
//     chunk and kept in LDS [term 2][k-half 2][pixel 18 x 34][8 fp16] (39 KB, double-buffered); all nine taps of the chunk read their
//     fragments from it -- a fragment = 32 consecutive pixels of a tile row shifted by the tap = consecutive 16-byte slots: conflict-free ds_read_b128,
//     no swizzle.  That is what a GEMM-shaped kernel cannot have: it would fetch every input pixel nine times (a CU ingests ~12 bytes per cycle).
//   B (weights): packed once per model into the K-step image [chunk][tap][term 2][k-half 2][oc][8 fp16] (8 KB per step for 128 output channels, padded to
//     whole 8 KB DMA rounds) and streamed by LDS-DMA into a ring of four stages, three steps ahead, counted vmcnt + raw s_barrier (vd3d_gemm.hip).
//   Epilogue: acc * colscale[oc] (the power of two the channel's weights were scaled by, exact) -> float32 NHWC stores, 128 bytes per pixel and N tile.
// Per K step and wave: 12 ds_read_b128 feed 4 x NWN x 3 MFMAs.
#include "vd3d_dev.h"
#include "vd3d_kernels.h"

typedef short c2_h8s __attribute__((ext_vector_type(8)));
typedef _Float16 c2_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 c2_h4 __attribute__((ext_vector_type(4)));
typedef float c2_f16 __attribute__((ext_vector_type(16)));

#define C2_TH 16
#define C2_TW 32
#define C2_PH (C2_TH + 2)
#define C2_PW (C2_TW + 2)
#define C2_NPIX (C2_PH * C2_PW)                       // 612
#define C2_NT 512
#define C2_A_BUF (2 * 2 * C2_NPIX * 16)               // one chunk image: 39 168 bytes
#define C2_A_ITEMS (C2_NPIX * 4)                      // (pixel, 4-channel quad) items of a chunk: 2 448
#define C2_A_ITERS ((C2_A_ITEMS + C2_NT - 1) / C2_NT) // 5 per thread
#define C2_A_STG (C2_A_ITERS * C2_NT * 16)            // float32 staging buffer of one chunk, item-linear (what a DMA instruction can write): 40 960 bytes
#define C2_NSTAGE 4
__host__ __device__ constexpr int c2_b_stage(int cout) { return (2 * 2 * cout * 16 + 8191) / 8192 * 8192; }   // 8 192 for 64 and 128 output channels
#define C2_B_OFF (2 * C2_A_BUF + C2_A_STG)
#define C2_LDS(COUT) (C2_B_OFF + C2_NSTAGE * c2_b_stage(COUT))   // 152 064

struct vd_c2_args {
  int B, H, W, Cin, Cout;
  int ntx, nty;          // tiles per frame
  int nchunk;            // Cin / 16
};

typedef __attribute__((address_space(3))) void* c2_lds_vp;
typedef const __attribute__((address_space(1))) void* c2_glb_vp;

template <int WM, int NWN>   // WM waves along M (4: four tile rows each, two waves along N; 8: two rows each, one along N), NWN N tiles per wave: C_out = 32 (8 / WM) NWN
__global__ __launch_bounds__(C2_NT) void k_conv3x3_x2(const float* __restrict__ X, const uint4* __restrict__ Wimg, const float* __restrict__ colscale,
                                                      const float* __restrict__ zero16, float* __restrict__ Y, vd_c2_args a) {
  constexpr int WN = 8 / WM, MR = C2_TH / WM, COUT = 32 * WN * NWN, BST = c2_b_stage(COUT), NBP = BST / (C2_NT * 16);   // B stage bytes, DMA instructions per thread and stage (1)
  extern __shared__ __attribute__((aligned(16))) uint8_t c2_lds[];   // the only LDS object: [A buffer 0][A buffer 1][float32 staging][B ring]
  const int tile = blockIdx.x, b = blockIdx.y;
  const int tyi = tile / a.ntx, txi = tile - tyi * a.ntx;
  const int y0 = tyi * C2_TH, x0 = txi * C2_TW;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave - wm * WN, li = lane & 31, kh = lane >> 5;
  const int wave_base = (tid & ~63) * 16;

  // ---- A staging: item i = it * 512 + tid -> (pixel = i >> 2 of the 18 x 34 halo tile, quad = i & 3 = four of the chunk's 16 channels); the DMA of item i lands
  // in staging slot i.  A pixel outside the image (zero padding) or an item past the tile fetches the 64 zero bytes behind the weight image.
  const float* ap[C2_A_ITERS]; int adst[C2_A_ITERS]; bool azero[C2_A_ITERS];
#pragma unroll
  for (int it = 0; it < C2_A_ITERS; ++it) {
    const int i = it * C2_NT + tid, pix = i >> 2, q = i & 3;
    const int py = pix / C2_PW, px = pix - py * C2_PW;
    const int gy = y0 - 1 + py, gx = x0 - 1 + px;
    const bool in = i < C2_A_ITEMS && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    azero[it] = !in;
    ap[it] = in ? X + (((size_t)b * a.H + gy) * a.W + gx) * a.Cin + q * 4 : zero16;
    adst[it] = i < C2_A_ITEMS ? ((q >> 1) * C2_NPIX + pix) * 16 + (q & 1) * 8 : -1;   // + term * 2 * C2_NPIX * 16
  }
  auto load_a = [&](int chunk) {
#pragma unroll
    for (int it = 0; it < C2_A_ITERS; ++it)
      __builtin_amdgcn_global_load_lds((c2_glb_vp)(azero[it] ? ap[it] : ap[it] + chunk * 16),
                                       (c2_lds_vp)(c2_lds + 2 * C2_A_BUF + it * (C2_NT * 16) + wave_base), 16, 0, 0);
  };
  auto write_a = [&](int buf) {   // staging (float32) -> split (round to nearest) -> two 8-byte LDS stores per item
    uint8_t* dst = c2_lds + buf * C2_A_BUF;
    const uint8_t* stg = c2_lds + 2 * C2_A_BUF + tid * 16;
#pragma unroll
    for (int it = 0; it < C2_A_ITERS; ++it) {
      // read as a short vector and bit-cast: hipcc orders a float4 LDS read behind every LDS-DMA in flight (vmcnt(0)), not this type (vd3d_gemm.hip)
      const c2_h8s raw = *reinterpret_cast<const c2_h8s*>(stg + it * (C2_NT * 16));
      const float4 f = __builtin_bit_cast(float4, raw);
      const float v[4] = {f.x, f.y, f.z, f.w};
      c2_h4 h1, h2;
#pragma unroll
      for (int e = 0; e < 4; ++e) { const _Float16 t = (_Float16)v[e]; h1[e] = t; h2[e] = (_Float16)(v[e] - (float)t); }
      if (adst[it] >= 0) {
        *reinterpret_cast<c2_h4*>(dst + adst[it]) = h1;
        *reinterpret_cast<c2_h4*>(dst + 2 * C2_NPIX * 16 + adst[it]) = h2;
      }
    }
  };
  // ---- B staging: K step ks = chunk * 9 + tap of the packed image, NBP DMA instructions per thread
  const uint4* wb = Wimg + tid;
  auto stage_b = [&](int ks, int slot) {
#pragma unroll
    for (int p = 0; p < NBP; ++p)
      __builtin_amdgcn_global_load_lds((c2_glb_vp)(wb + (size_t)ks * (BST / 16) + p * C2_NT),
                                       (c2_lds_vp)(c2_lds + C2_B_OFF + slot * BST + p * (C2_NT * 16) + wave_base), 16, 0, 0);
  };

  c2_f16 acc[MR][NWN];
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int n = 0; n < NWN; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  const int KS = a.nchunk * 9;
  load_a(0);
  stage_b(0, 0);
  stage_b(KS > 1 ? 1 : 0, 1);
  stage_b(KS > 2 ? 2 : KS - 1, 2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the prologue waits for everything; a thread converts only the staging slots its OWN DMA lanes filled
  write_a(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // fragment base offsets: A: k-half plane, tile row 4 wm + m (+ 1 halo + dy), column li (+ 1 + dx); B: k-half plane, output channel (wn * NWN + n) * 32 + li
  const int fa_base = (kh * C2_NPIX + (MR * wm + 1) * C2_PW + li + 1) * 16;
  const int fb_base = C2_B_OFF + (kh * COUT + wn * NWN * 32 + li) * 16;
  int ks = 0;
  for (int chunk = 0; chunk < a.nchunk; ++chunk) {
    const bool more_a = chunk + 1 < a.nchunk;   // uniform
    if (more_a) load_a(chunk + 1);
    const uint8_t* sa = c2_lds + (chunk & 1) * C2_A_BUF;
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap, ++ks) {
      const int slot = ks & 3, dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
      stage_b(ks + 3 < KS ? ks + 3 : KS - 1, (ks + 3) & 3);   // behind the last step: a harmless re-fetch (straight-line code, one counted wait)
      const uint8_t* sb = c2_lds + slot * BST;
      const uint8_t* sat = sa + fa_base + (dy * C2_PW + dx) * 16;
      c2_h8s bf[NWN][2];
#pragma unroll
      for (int n = 0; n < NWN; ++n)
#pragma unroll
        for (int t = 0; t < 2; ++t) bf[n][t] = *reinterpret_cast<const c2_h8s*>(sb + fb_base + t * (2 * COUT * 16) + n * 512);
#pragma unroll
      for (int m = 0; m < MR; ++m) {
        c2_h8s af[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) af[t] = *reinterpret_cast<const c2_h8s*>(sat + t * (2 * C2_NPIX * 16) + m * (C2_PW * 16));
#pragma unroll
        for (int n = 0; n < NWN; ++n) {   // small products first: x2 w1, x1 w2, x1 w1
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(c2_h8, af[1]), __builtin_bit_cast(c2_h8, bf[n][0]), acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(c2_h8, af[0]), __builtin_bit_cast(c2_h8, bf[n][1]), acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(c2_h8, af[0]), __builtin_bit_cast(c2_h8, bf[n][0]), acc[m][n], 0, 0, 0);
        }
      }
      // Counted wait.  In flight, oldest first: [B (ks + 1)] [B (ks + 2)] [the next chunk's C2_A_ITERS DMAs, issued at tap 0] [B (ks + 3)].  The next step needs
      // B (ks + 1): taps 0 and 1 let the A loads and the two younger DMA rounds stay in flight; from tap 2 on the A loads are older than what must land, so they
      // land too -- two MFMA groups after they were issued.
      if (more_a && tap < 2) {
        if (NBP == 1) asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)" ::: "memory");
      } else {
        if (NBP == 1) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      }
      if (more_a && tap == 3) {   // the next chunk's pixels have landed (tap 2's wait): split them into the other buffer -- last read in the previous chunk
        write_a((chunk + 1) & 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- epilogue: accumulator register r of (m, n) = pixel column (r & 3) + 8 (r >> 2) + 4 kh of tile row 4 wm + m, output channel (wn * NWN + n) * 32 + li
#pragma unroll
  for (int n = 0; n < NWN; ++n) {
    const int oc = (wn * NWN + n) * 32 + li;
    float cs = colscale[oc];
    asm volatile("" : "+v"(cs));   // consumed in front of the masked stores (vd3d_gemm.hip: else one s_waitcnt vmcnt(0) per store)
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      const int y = y0 + MR * wm + m;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int x = x0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (y < a.H && x < a.W) Y[(((size_t)b * a.H + y) * a.W + x) * COUT + oc] = acc[m][n][r] * cs;
      }
    }
  }
}

// ---- weights: float32 [Cout][Cin][3][3] -> colscale[Cout] (2^-e) and the K-step images [chunk][tap][term][k-half][oc][8 fp16], padded per step to c2_b_stage
__global__ __launch_bounds__(256) void k_conv3x3_x2_rowscale(const float* __restrict__ W, int Cout, int per_oc, float* __restrict__ colscale) {
  const int oc = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (oc >= Cout) return;
  float mx = 0.f;
  for (int k = lane; k < per_oc; k += 64) mx = fmaxf(mx, fabsf(W[(size_t)oc * per_oc + k]));
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  if (lane == 0) {
    int e = 0;
    if (mx > 0.f && mx < INFINITY) { e = 13 - (int)floorf(log2f(mx)); e = e < -100 ? -100 : (e > 100 ? 100 : e); }
    colscale[oc] = exp2f((float)-e);
  }
}
__global__ __launch_bounds__(256) void k_conv3x3_x2_pack(const float* __restrict__ W, int Cout, int Cin, const float* __restrict__ colscale, uint8_t* __restrict__ img) {
  // one thread = (chunk, tap, k-half, oc): 8 consecutive input channels
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int nchunk = Cin / 16, total = nchunk * 9 * 2 * Cout;
  if (t >= total) return;
  const int oc = t % Cout, khf = (t / Cout) & 1, tap = (t / (2 * Cout)) % 9, chunk = t / (18 * Cout);
  const float sc = 1.0f / colscale[oc];
  c2_h8 h1, h2;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ic = chunk * 16 + khf * 8 + e;
    const float v = W[((size_t)oc * Cin + ic) * 9 + tap] * sc;
    const _Float16 a1 = (_Float16)v;
    h1[e] = a1; h2[e] = (_Float16)(v - (float)a1);
  }
  const int bst = (2 * 2 * Cout * 16 + 8191) / 8192 * 8192;
  uint8_t* base = img + (size_t)(chunk * 9 + tap) * bst + (khf * Cout + oc) * 16;
  *reinterpret_cast<c2_h8*>(base) = h1;
  *reinterpret_cast<c2_h8*>(base + 2 * Cout * 16) = h2;
}

static bool c2_shape_ok(int Cin, int Cout) { return Cin >= 16 && (Cin & 15) == 0 && (Cout == 32 || Cout == 64 || Cout == 128); }
long long vd_conv3x3_x2_weight_bytes(int Cin, int Cout) {
  if (!c2_shape_ok(Cin, Cout)) return -1;
  const long long bst = (2 * 2 * Cout * 16 + 8191) / 8192 * 8192;
  return (long long)(Cin / 16) * 9 * bst + (long long)Cout * 4 + 64;   // step images, colscale[Cout], 64 zero bytes (the zero page of the padding)
}
bool vd_launch_conv3x3_x2_pack(hipStream_t s, const float* W, int Cin, int Cout, void* img) {
  const long long nb = vd_conv3x3_x2_weight_bytes(Cin, Cout);
  if (nb < 0) return false;
  float* cs = reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(img) + nb - 64 - (long long)Cout * 4);
  if (hipMemsetAsync(img, 0, (size_t)nb, s) != hipSuccess) return false;   // the padding of a step image
  hipLaunchKernelGGL(k_conv3x3_x2_rowscale, dim3((Cout + 3) / 4), dim3(256), 0, s, W, Cout, Cin * 9, cs);
  const int total = (Cin / 16) * 9 * 2 * Cout;
  hipLaunchKernelGGL(k_conv3x3_x2_pack, dim3((total + 255) / 256), dim3(256), 0, s, W, Cout, Cin, (const float*)cs, reinterpret_cast<uint8_t*>(img));
  return true;
}
bool vd_launch_conv3x3_x2(hipStream_t s, const float* X, int B, int H, int W, int Cin, const void* wimg, int Cout, float* Y) {
  const long long nb = vd_conv3x3_x2_weight_bytes(Cin, Cout);
  if (nb < 0 || B < 1 || H < 1 || W < 1 || B > 65535) return false;
  if ((reinterpret_cast<uintptr_t>(X) & 15) || (reinterpret_cast<uintptr_t>(wimg) & 15)) return false;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3x3_x2<4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, C2_LDS(64)) != hipSuccess) return false;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3x3_x2<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, C2_LDS(128)) != hipSuccess) return false;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3x3_x2<8, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, C2_LDS(32)) != hipSuccess) return false;
    attr_set = true;
  }
  vd_c2_args a;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
  a.ntx = (W + C2_TW - 1) / C2_TW; a.nty = (H + C2_TH - 1) / C2_TH; a.nchunk = Cin / 16;
  const float* cs = reinterpret_cast<const float*>(reinterpret_cast<const uint8_t*>(wimg) + nb - 64 - (long long)Cout * 4);
  const float* z16 = reinterpret_cast<const float*>(reinterpret_cast<const uint8_t*>(wimg) + nb - 64);
  const dim3 grid((unsigned)(a.ntx * a.nty), (unsigned)B);
  if (Cout == 32) hipLaunchKernelGGL((k_conv3x3_x2<8, 1>), grid, dim3(C2_NT), C2_LDS(32), s, X, reinterpret_cast<const uint4*>(wimg), cs, z16, Y, a);   // the head's 64 -> 32
  else if (Cout == 64) hipLaunchKernelGGL((k_conv3x3_x2<4, 1>), grid, dim3(C2_NT), C2_LDS(64), s, X, reinterpret_cast<const uint4*>(wimg), cs, z16, Y, a);
  else hipLaunchKernelGGL((k_conv3x3_x2<4, 2>), grid, dim3(C2_NT), C2_LDS(128), s, X, reinterpret_cast<const uint4*>(wimg), cs, z16, Y, a);
  return true;
}
