// vd3d_gemm.hip -- Y[M][N] = X[M][K] . W[N][K]^T + bias (+ exact GELU): the linear layers of the depth network's transformer blocks (boundary B3,
// core/render_depth.py:1106-1119 runs them in float32) as a SPLIT-bf16 GEMM on the gfx950 matrix cores -- an opt-in mode of the depth leg (round 6).
//
// Why.  gfx950 has no TF32 / xf32 MFMA and its float32-input MFMA runs at the float32 VECTOR rate (157 TFLOP/s, 1/16 of bf16): a pure float32 ViT is
// pinned there no matter how good the library is (hipBLASLt reaches 121 TFLOP/s on these shapes).  A float32 number is EXACTLY the sum of three bf16
// numbers (8 significant bits each, 3 x 8 = 24: truncate, subtract, truncate, subtract -- the last remainder has <= 8 bits), so
//   x . w = (x1 + x2 + x3)(w1 + w2 + w3) = x1 w1 + x1 w2 + x2 w1 + x1 w3 + x2 w2 + x3 w1  +  [x2 w3 + x3 w2 + x3 w3]
// with every bf16 x bf16 product exact in float32.  The bracket is <= 2^-23 |x w| (the size of ONE float32 rounding of the product) and is dropped;
// the six kept products per K-step go through v_mfma_f32_32x32x16_bf16 with float32 accumulation.  Result: float32-faithful dot products (error
// model = a float32 GEMM with another summation order plus one extra rounding-sized term per product; tests/test_hip_gemm.py checks it against
// float64) at six bf16 MFMAs per float32 MAC: a ceiling of 2.5 PFLOP/s / 6 = 417 TFLOP/s float32-equivalent instead of 157.
//
// Kernel (k_gemm_bf16x3): 256 x 256 output tile per workgroup, 512 threads = 8 waves as 2 (M) x 4 (N), wave tile 128 x 64 = 4 x 2 MFMA tiles
// = 128 accumulator registers, two waves per SIMD.  K-step 16 (one MFMA K), LDS double-buffered: a stage holds the three bf16 terms of the A tile
// and of the B tile CHUNK-major -- [term 3][k-half 2][row 256][8 bf16] = 24 KB each -- so that the 32 lanes of a fragment read hit consecutive
// 16-byte slots (conflict-free ds_read_b128); 2 x 48 KB = 96 KB.
//   B (weights): split and packed ONCE per model (k_gemm_x3_pack_w) into exactly that stage image, [N tile][K step][term][k-half][n][8]; a stage is
//     24 KB of contiguous global memory and goes to LDS by global_load_lds_dwordx4 (3 per thread), no registers, no VALU.
//   A (activations, float32 row-major as every producer writes them): two 16-byte global loads per thread per stage issued one stage ahead, split
//     in registers (and / sub / and / sub per element, v_perm to pack) and written as 8-byte LDS stores behind the MFMAs of the current stage.
//   Per stage and wave: 18 ds_read_b128 feed 48 MFMAs (a plain bf16 GEMM with this tiling: 6 reads per 8 MFMAs) -- the six-product form is
//   MFMA-bound by construction; small terms are accumulated first, the two N tiles of a row alternate so that dependent MFMAs are 64 cycles apart.
// Tile order: workgroup b runs on XCD b % 8 (speed assumption only): every XCD owns the M tiles mt = x (mod 8) and walks them four at a time across
// all N tiles, so that the ~32 workgroups resident on an XCD share 4 A panels and 8 B panels through its L2.
// Epilogue: + bias[n], optionally exact GELU (0.5 x (1 + erf(x / sqrt 2)), torch.nn.GELU()'s default form, float32 erff), float32 stores (each
// accumulator register = two 128-byte row segments per wave).
#include "vd3d_dev.h"
#include "vd3d_kernels.h"

typedef short gx_bf8 __attribute__((ext_vector_type(8)));     // 8 bf16 = one MFMA A / B fragment (4 VGPRs)
typedef float gx_f16 __attribute__((ext_vector_type(16)));    // one 32 x 32 accumulator tile per wave

#define GX_BM 256
#define GX_BN 256
#define GX_NT 512
#define GX_STAGE_HALF (3 * 2 * 256 * 16)        // bytes of one operand's stage image: 24 576
#define GX_STAGE (2 * GX_STAGE_HALF)             // A image + B image: 49 152
#define GX_LDS (2 * GX_STAGE)                    // double-buffered: 98 304

struct vd_gx_args {
  long long M;
  int K, N, KS;          // KS = K / 16 stages
  int nbm, nbn;          // tiles
  int epilogue;          // 0: bias only, 1: bias + exact GELU
  int has_bias;
};

// exact split of a float32 into bf16 terms by truncation: a == t1 + t2 + t3 (as floats whose low 16 bits are zero)
VD_DEV void gx_split(float a, uint32_t& t1, uint32_t& t2, uint32_t& t3) {
  t1 = __float_as_uint(a) & 0xffff0000u;
  const float r1 = a - __uint_as_float(t1);
  t2 = __float_as_uint(r1) & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(t2);
  t3 = __float_as_uint(r2);   // <= 8 significant bits: its low half is zero
}
// pack the high halves of two words: lo | hi << 16
VD_DEV uint32_t gx_pack(uint32_t lo, uint32_t hi) { return __builtin_amdgcn_perm(hi, lo, 0x07060302u); }

VD_DEV float gx_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

typedef __attribute__((address_space(3))) void* gx_lds_vp;
typedef const __attribute__((address_space(1))) void* gx_glb_vp;

__global__ __launch_bounds__(GX_NT) void k_gemm_bf16x3(const float* __restrict__ X, const uint4* __restrict__ Wimg, const float* __restrict__ bias,
                                                        float* __restrict__ Y, vd_gx_args a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t gx_lds[];
  // ---- tile of this workgroup (XCD-aware order, see the header)
  int mt, nt;
  {
    const int b = blockIdx.x, x = b & 7, idx = b >> 3;
    const int per = 4 * a.nbn, mg = idx / per, rem = idx - mg * per;
    nt = rem >> 2;
    mt = ((mg * 4 + (rem & 3)) << 3) + x;
    if (mt >= a.nbm) return;   // padding workgroup (uniform, before any barrier)
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3, li = lane & 31, kh = lane >> 5;
  const long long m0 = (long long)mt * GX_BM;
  const int n0 = nt * GX_BN;

  // ---- A staging: thread = (k quad q, rows r0 and r0 + 128); 4 lanes cover the 64 bytes a row contributes to a stage
  const int q = tid & 3, r0 = tid >> 2;
  const float* xa[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    long long row = m0 + r0 + 128 * i;
    if (row > a.M - 1) row = a.M - 1;   // rows past the end load a valid row and are never stored
    xa[i] = X + row * (long long)a.K + 4 * q;
  }
  // LDS byte offset of the thread's 8-byte slot inside a term plane pair: (k-half, row, 8-byte half of the chunk)
  const int aw_off = ((q >> 1) * 256 + r0) * 16 + (q & 1) * 8;
  // ---- B staging: the stage image is contiguous in global memory; thread t moves 16-byte pieces t, t + 512, t + 1024
  const uint4* wb = Wimg + (size_t)nt * (size_t)a.KS * (GX_STAGE_HALF / 16) + tid;

  auto stage_b = [&](int ks, int buf) {
    uint8_t* dst = gx_lds + buf * GX_STAGE + GX_STAGE_HALF;   // B image behind the A image
    const uint4* src = wb + (size_t)ks * (GX_STAGE_HALF / 16);
#pragma unroll
    for (int p = 0; p < 3; ++p)   // wave-uniform LDS base + lane * 16: pieces of one wave-instruction are 1 KB contiguous on both sides
      __builtin_amdgcn_global_load_lds((gx_glb_vp)(src + p * GX_NT), (gx_lds_vp)(dst + (p * GX_NT + (tid & ~63)) * 16), 16, 0, 0);
  };
  auto load_a = [&](int ks, float4* ra) {
#pragma unroll
    for (int i = 0; i < 2; ++i) ra[i] = *reinterpret_cast<const float4*>(xa[i] + ks * 16);
  };
  auto write_a = [&](const float4* ra, int buf) {
    uint8_t* dst = gx_lds + buf * GX_STAGE + aw_off;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      uint32_t t1[4], t2[4], t3[4];
      gx_split(ra[i].x, t1[0], t2[0], t3[0]); gx_split(ra[i].y, t1[1], t2[1], t3[1]);
      gx_split(ra[i].z, t1[2], t2[2], t3[2]); gx_split(ra[i].w, t1[3], t2[3], t3[3]);
      uint8_t* d = dst + i * (128 * 16);
      *reinterpret_cast<uint2*>(d + 0 * 8192) = make_uint2(gx_pack(t1[0], t1[1]), gx_pack(t1[2], t1[3]));
      *reinterpret_cast<uint2*>(d + 1 * 8192) = make_uint2(gx_pack(t2[0], t2[1]), gx_pack(t2[2], t2[3]));
      *reinterpret_cast<uint2*>(d + 2 * 8192) = make_uint2(gx_pack(t3[0], t3[1]), gx_pack(t3[2], t3[3]));
    }
  };

  gx_f16 acc[4][2];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int nj = 0; nj < 2; ++nj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][nj][r] = 0.f;

  // fragment addresses inside a stage: A rows wm * 128 + mi * 32 + li, B columns wn * 64 + nj * 32 + li; term plane pairs are 8 192 bytes apart
  const int fa_off = (kh * 256 + wm * 128 + li) * 16;
  const int fb_off = GX_STAGE_HALF + (kh * 256 + wn * 64 + li) * 16;

  float4 ra[2];
  load_a(0, ra);
  stage_b(0, 0);
  write_a(ra, 0);
  __syncthreads();   // (waits for the LDS-DMA of stage 0 as well: vmcnt(0) is part of the barrier's fence while a DMA is in flight)

  for (int ks = 0; ks < a.KS; ++ks) {
    const int cur = ks & 1;
    const bool more = ks + 1 < a.KS;   // uniform
    if (more) { load_a(ks + 1, ra); stage_b(ks + 1, cur ^ 1); }
    const uint8_t* sb = gx_lds + cur * GX_STAGE;
    gx_bf8 bf[2][3];
#pragma unroll
    for (int nj = 0; nj < 2; ++nj)
#pragma unroll
      for (int t = 0; t < 3; ++t) bf[nj][t] = *reinterpret_cast<const gx_bf8*>(sb + fb_off + t * 8192 + nj * 512);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      gx_bf8 af[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) af[t] = *reinterpret_cast<const gx_bf8*>(sb + fa_off + t * 8192 + mi * 512);
      // small products first; (ta, tb): x3 w1, x2 w2, x1 w3, x2 w1, x1 w2, x1 w1
#define GX_MM(ta, tb)                                                                                   \
  acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ta], bf[0][tb], acc[mi][0], 0, 0, 0);        \
  acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ta], bf[1][tb], acc[mi][1], 0, 0, 0);
      GX_MM(2, 0) GX_MM(1, 1) GX_MM(0, 2) GX_MM(1, 0) GX_MM(0, 1) GX_MM(0, 0)
#undef GX_MM
    }
    if (more) write_a(ra, cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: accumulator register r of tile (mi, nj) = row (r & 3) + 8 (r >> 2) + 4 kh, column li
#pragma unroll
  for (int nj = 0; nj < 2; ++nj) {
    const int n = n0 + wn * 64 + nj * 32 + li;
    const bool nok = n < a.N;
    const float bv = (a.has_bias && nok) ? bias[n] : 0.f;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long long m = m0 + wm * 128 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        float v = acc[mi][nj][r] + bv;
        if (a.epilogue == 1) v = gx_gelu(v);
        if (nok && m < a.M) Y[m * (long long)a.N + n] = v;
      }
    }
  }
}

// ---- weights: float32 [N][K] -> the stage images [N tile][K step][term][k-half][n 256][8 bf16]; rows past N are zero.  One thread per (n, 8 k).
__global__ __launch_bounds__(256) void k_gemm_x3_pack_w(const float* __restrict__ W, int N, int K, int nbn, uint4* __restrict__ img) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const int kc = K / 8;
  const long long total = (long long)nbn * GX_BN * kc;
  if (t >= total) return;
  const int n = (int)(t / kc), c = (int)(t - (long long)n * kc);   // chunk c = 8 consecutive k
  uint32_t w1[8], w2[8], w3[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = n < N ? W[(size_t)n * K + c * 8 + j] : 0.f;
    gx_split(v, w1[j], w2[j], w3[j]);
  }
  const int ntile = n / GX_BN, nl = n - ntile * GX_BN, ks = c >> 1, khf = c & 1;
  uint4* base = img + ((size_t)ntile * (K / 16) + ks) * (GX_STAGE_HALF / 16) + khf * 256 + nl;
  base[0 * 512] = make_uint4(gx_pack(w1[0], w1[1]), gx_pack(w1[2], w1[3]), gx_pack(w1[4], w1[5]), gx_pack(w1[6], w1[7]));
  base[1 * 512] = make_uint4(gx_pack(w2[0], w2[1]), gx_pack(w2[2], w2[3]), gx_pack(w2[4], w2[5]), gx_pack(w2[6], w2[7]));
  base[2 * 512] = make_uint4(gx_pack(w3[0], w3[1]), gx_pack(w3[2], w3[3]), gx_pack(w3[4], w3[5]), gx_pack(w3[6], w3[7]));
}

long long vd_gemm_x3_weight_bytes(int N, int K) {
  if (N < 1 || K < 16 || (K & 15)) return -1;
  const long long nbn = (N + GX_BN - 1) / GX_BN;
  return nbn * (K / 16) * (long long)GX_STAGE_HALF;
}

bool vd_launch_gemm_x3_pack_w(hipStream_t s, const float* W, int N, int K, void* img) {
  if (vd_gemm_x3_weight_bytes(N, K) < 0) return false;
  const int nbn = (N + GX_BN - 1) / GX_BN;
  const long long total = (long long)nbn * GX_BN * (K / 8);
  hipLaunchKernelGGL(k_gemm_x3_pack_w, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, W, N, K, nbn, reinterpret_cast<uint4*>(img));
  return true;
}

bool vd_launch_gemm_x3(hipStream_t s, const float* X, long long M, int K, const void* wimg, int N, const float* bias, int epilogue, float* Y) {
  if (M < 1 || vd_gemm_x3_weight_bytes(N, K) < 0 || epilogue < 0 || epilogue > 1) return false;
  if ((reinterpret_cast<uintptr_t>(X) & 15) || (reinterpret_cast<uintptr_t>(wimg) & 15)) return false;
  static bool attr_set = false;   // idempotent: a race between two first calls sets the same value twice
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_bf16x3), hipFuncAttributeMaxDynamicSharedMemorySize, GX_LDS) != hipSuccess) return false;
    attr_set = true;
  }
  vd_gx_args a;
  a.M = M; a.K = K; a.N = N; a.KS = K / 16;
  a.nbm = (int)((M + GX_BM - 1) / GX_BM); a.nbn = (N + GX_BN - 1) / GX_BN;
  a.epilogue = epilogue; a.has_bias = bias ? 1 : 0;
  const int per_xcd = (a.nbm + 7) / 8, groups = (per_xcd + 3) / 4;
  const unsigned grid = 8u * (unsigned)groups * 4u * (unsigned)a.nbn;
  hipLaunchKernelGGL(k_gemm_bf16x3, dim3(grid), dim3(GX_NT), GX_LDS, s, X, reinterpret_cast<const uint4*>(wimg), bias, Y, a);
  return true;
}
