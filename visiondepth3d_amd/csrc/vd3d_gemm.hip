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
// float64, beside hipBLASLt's float32 GEMM) at six bf16 MFMAs per float32 MAC: a ceiling of 2.5 PFLOP/s / 6 = 417 TFLOP/s float32-equivalent
// instead of 157.  That is MODE 0 ("bf16x3").
// MODE 1 ("fp16x2"), same kernel: every operand as TWO fp16 terms, h1 = fp16(x), h2 = fp16(x - h1), both round-to-nearest -- 22 significant bits,
// |x - h1 - h2| <= 2^-22 |x| while h2 is a normal fp16 number -- and the three products x1 w1 + x1 w2 + x2 w1 through v_mfma_f32_32x32x16_f16: HALF the
// matrix work.  Per product up to ~3 x 2^-22 relative instead of one float32 rounding; summed over K >= 64 products that stays below the float32
// accumulation error both modes share (measured: both are MORE accurate against float64 than hipBLASLt's float32 kernel, fp16x2 the most -- three
// accumulations per MAC instead of six).  Range: fp16 stops at 65 504 and its normal numbers at 2^-14, so weight rows are multiplied by the power of two that
// brings their largest magnitude into [2^13, 2^14) before the split (exact; the epilogue multiplies the column's sums by its reciprocal) and activations
// are taken as they are: |x| must stay below 65 504 (else Inf / NaN), and an |x| < 2^-2 is represented with an absolute error <= 2^-25 instead of a relative one.
//
// Kernel (k_gemm_bf16x3): 256 x 256 output tile per workgroup, 512 threads = 8 waves as 4 (M) x 2 (N), wave tile 64 x 128 = 2 x 4 MFMA tiles =
// 128 accumulator registers, two waves per SIMD.  K-step 16 (one MFMA K).  BOTH operands reach LDS by LDS-DMA (global_load_lds_dwordx4: no
// registers, no VALU, no ds_write) into a ring of FOUR stages, three stages ahead of their use, with counted vmcnt and raw s_barrier -- a
// __syncthreads() would drain the DMA queue at every stage.  A stage is visible one whole iteration before its MFMAs, so the split of ITS A
// fragments runs in the shadow of the previous stage's MFMAs instead of in front of its own:
//   A (activations, float32 row-major as every producer writes them) stays FLOAT32 in LDS, [row 256][k quad 4, XOR-swizzled][4 floats] = 16 KB per
//     stage: the source address of a DMA is per lane and the LDS image lane-linear, so the swizzle is applied to WHICH 16 bytes a lane fetches; four
//     neighbouring lanes fetch one row's 64 contiguous bytes.  Each wave splits ITS rows' fragments into the three bf16 terms in
//     registers (and / sub / and / sub per element, v_perm to pack: 44 VALU per 32 x 16 fragment) in the shadow of its MFMAs.
//   B (weights) is split and packed ONCE per model (k_gemm_x3_pack_w) into the stage image [N tile][K step][term 3][k-half 2][n 256][8 bf16]:
//     24 KB of contiguous global memory per stage.
//   Stage = 40 KB, ring = 160 KB = the whole LDS of a CU: one workgroup per CU.
// Measured (round 6, MI355X, M = 39 088 = 16 frames x 2 443 tokens, tools/probe_gemm_x3.py): 162 (K 768, N 2304) .. 204 (K 3072, N 768) TFLOP/s
// float32-equivalent = 0.97 - 1.22 PFLOP/s of bf16 MFMA work, against 114 - 128 TFLOP/s for hipBLASLt's float32 GEMM on the same shapes; one DA-V2-Base
// layer 3.15 ms against 4.59.  Timing ablations of this kernel (tools/ablate_gemm.sh, K 768 / N 2304, 0.855 ms): MFMAs + fragment reads alone 0.596 ms
// (1.39 PFLOP/s: 72 % of the matrix peak at the 1.86 GHz the chip holds under this load, 10 % of it tile quantisation: 1 377 tiles on 256 CUs), + split
// 0.627, + stores 0.682; the DMAs alone 0.347 ms.  The DMA cost that stays exposed (~0.17 ms) did not move with the issue pattern: all DMAs at the top
// of the iteration 0.92, dealt between the MFMA groups 0.85, issued by one wave per SIMD only 0.85, B fragments read one chunk ahead 0.86; without the
// counted wait AND without the barrier 0.84 (nobody waits for a DMA to land); without the A DMAs 0.79, without the B DMAs 0.81, without both 0.67.  The
// cost follows the WORK in flight, not the schedule (the chip holds 1.86 GHz under this load against 2.4 GHz nominal: consistent with a power limit, not
// proven to be one): what moved the kernel was doing less per MAC -- MODE 1: 0.551 ms for the same shape, 228 - 339 TFLOP/s float32-equivalent over the four.  Earlier structures, for the record: A split in
// registers before the LDS write (global loads one stage ahead, 12 ds_write per thread and stage, one __syncthreads() per stage, double buffer)
// 0.86 - 0.92 ms; 256 x 128 tiles with two workgroups per CU 0.88; a three-stage ring with the split in front of its own MFMAs 0.92.
// Tile order: workgroup b runs on XCD b % 8 (speed assumption only): every XCD owns the M tiles mt = x (mod 8) and walks them four at a time across
// all N tiles, so that the ~32 workgroups resident on an XCD share 4 A panels and 8 B panels through its L2.
// Epilogue: + bias[n], optionally exact GELU (0.5 x (1 + erf(x / sqrt 2)), torch.nn.GELU()'s default form, float32 erff), float32 stores (each
// accumulator register = two 128-byte row segments per wave).
#include "vd3d_dev.h"
#include "vd3d_kernels.h"
#include <cstdlib>

typedef short gx_bf8 __attribute__((ext_vector_type(8)));     // 8 bf16 = one MFMA A / B fragment (4 VGPRs)
typedef float gx_f16 __attribute__((ext_vector_type(16)));    // one 32 x 32 accumulator tile per wave

#define GX_BM 256
#define GX_BN 256
#define GX_NT 512
#define GX_MG 4                                  // M tiles of an XCD walked together across the N tiles
#define GX_A_STAGE (4 * GX_BM * 16)              // float32 A image of a stage: 16 384 bytes
// MODE 0 = bf16x3 (three exact bf16 terms per operand, six products); MODE 1 = fp16x2 (two fp16 terms per operand -- 22 bits, round to nearest -- three
// products: HALF the matrix work; see the header)
#define GX_NTERM(MODE) ((MODE) == 0 ? 3 : 2)
#define GX_STAGE_HALF(MODE) (GX_NTERM(MODE) * 2 * GX_BN * 16)   // B image of a stage: 24 576 / 16 384 bytes (the unit of the packed weights)
#define GX_STAGE(MODE) (GX_A_STAGE + GX_STAGE_HALF(MODE))       // 40 960 / 32 768
#define GX_NSTAGE 4
#define GX_LDS(MODE) (GX_NSTAGE * GX_STAGE(MODE))               // 163 840 = all of a CU's LDS / 131 072

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
// eight float32 (k = 8 kh .. 8 kh + 7 of one row) -> the three bf16 fragments of that lane
VD_DEV void gx_split8(const float4& lo, const float4& hi, gx_bf8 out[3]) {
  uint32_t t1[8], t2[8], t3[8];
  gx_split(lo.x, t1[0], t2[0], t3[0]); gx_split(lo.y, t1[1], t2[1], t3[1]); gx_split(lo.z, t1[2], t2[2], t3[2]); gx_split(lo.w, t1[3], t2[3], t3[3]);
  gx_split(hi.x, t1[4], t2[4], t3[4]); gx_split(hi.y, t1[5], t2[5], t3[5]); gx_split(hi.z, t1[6], t2[6], t3[6]); gx_split(hi.w, t1[7], t2[7], t3[7]);
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  const u4 p1 = {gx_pack(t1[0], t1[1]), gx_pack(t1[2], t1[3]), gx_pack(t1[4], t1[5]), gx_pack(t1[6], t1[7])};
  const u4 p2 = {gx_pack(t2[0], t2[1]), gx_pack(t2[2], t2[3]), gx_pack(t2[4], t2[5]), gx_pack(t2[6], t2[7])};
  const u4 p3 = {gx_pack(t3[0], t3[1]), gx_pack(t3[2], t3[3]), gx_pack(t3[4], t3[5]), gx_pack(t3[6], t3[7])};
  out[0] = __builtin_bit_cast(gx_bf8, p1); out[1] = __builtin_bit_cast(gx_bf8, p2); out[2] = __builtin_bit_cast(gx_bf8, p3);
}

// fp16x2: a ~ h1 + h2 with h1 = fp16(a), h2 = fp16(a - h1), both round-to-nearest: |a - h1 - h2| <= 2^-22 |a| while h2 stays a normal fp16 number (|a| >= 2^-2
// for unscaled data; below that the absolute error is <= 2^-25).  |a| must stay below 65 504.
typedef _Float16 gx_h8 __attribute__((ext_vector_type(8)));
VD_DEV void gx_split8_h(const float4& lo, const float4& hi, gx_bf8 out[2]) {
  const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  gx_h8 h1, h2;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const _Float16 a1 = (_Float16)v[e];
    h1[e] = a1;
    h2[e] = (_Float16)(v[e] - (float)a1);
  }
  out[0] = __builtin_bit_cast(gx_bf8, h1); out[1] = __builtin_bit_cast(gx_bf8, h2);
}
template <int MODE> VD_DEV void gx_split8_m(const float4& lo, const float4& hi, gx_bf8* out) {
  if (MODE == 0) gx_split8(lo, hi, out); else gx_split8_h(lo, hi, out);
}
template <int MODE> VD_DEV gx_f16 gx_mfma(const gx_bf8& a, const gx_bf8& b, const gx_f16& c) {
  if (MODE == 0) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gx_h8, a), __builtin_bit_cast(gx_h8, b), c, 0, 0, 0);
}

VD_DEV float gx_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

typedef __attribute__((address_space(3))) void* gx_lds_vp;
typedef const __attribute__((address_space(1))) void* gx_glb_vp;

// DBG (development ablations, VD3D_GEMM_DBG; results are wrong): 1 no DMA, 2 no barrier, 4 no MFMA, 8 no A split, 16 no stores, 32 no A DMA, 64 no B DMA, 128 no vmcnt wait, 256 three of the six products
template <int DBG, int MODE>
__global__ __launch_bounds__(GX_NT) void k_gemm_bf16x3(const float* __restrict__ X, const uint4* __restrict__ Wimg, const float* __restrict__ bias,
                                                        const float* __restrict__ colscale, float* __restrict__ Y, vd_gx_args a) {
  constexpr int NTERM = GX_NTERM(MODE), STAGE_HALF = GX_STAGE_HALF(MODE), STAGE = GX_STAGE(MODE), NB = STAGE_HALF / (GX_NT * 16), NDMA = 2 + NB;
  extern __shared__ __attribute__((aligned(16))) uint8_t gx_lds[];   // the ONLY LDS object of the kernel (a second one makes hipcc drain the DMA queue per stage)
  // ---- tile of this workgroup (XCD-aware order, see the header)
  int mt, nt;
  {
    const int b = blockIdx.x, x = b & 7, idx = b >> 3;
    const int per = GX_MG * a.nbn, mg = idx / per, rem = idx - mg * per;
    nt = rem / GX_MG;
    mt = ((mg * GX_MG + (rem - nt * GX_MG)) << 3) + x;
    if (mt >= a.nbm) return;   // padding workgroup (uniform, before any barrier)
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, kh = lane >> 5;
  const long long m0 = (long long)mt * GX_BM;
  const int n0 = nt * GX_BN;

  // ---- A staging by LDS-DMA: piece p of thread t is 16-byte slot p * 512 + t of the stage image.  Slot s holds (row = s >> 2, quad = (s & 3) ^ ((row >> 2) & 3)):
  // four neighbouring lanes fetch the 64 contiguous bytes a row contributes to the stage (a DMA instruction = 16 rows x 64 B, not 64 rows x 16 B), and
  // the XOR makes the fragment reads -- one quad of 32 consecutive rows, 64 bytes apart -- conflict-free: within any 16-lane group of a ds_read_b128
  // the four 4-row runs have distinct (row >> 2) & 3, so the 16 slots are distinct mod 16
  const float* xa[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int slot = p * GX_NT + tid, r = slot >> 2;
    long long row = m0 + r;
    if (row > a.M - 1) row = a.M - 1;   // rows past the end fetch a valid row and are never stored
    xa[p] = X + row * (long long)a.K + 4 * ((slot & 3) ^ ((r >> 2) & 3));
  }
  // ---- B staging: the stage image is contiguous in global memory; thread t moves 16-byte pieces t, t + 512, t + 1024
  const uint4* wb = Wimg + (size_t)nt * (size_t)a.KS * (STAGE_HALF / 16) + tid;
  const int wave_base = (tid & ~63) * 16;   // wave-uniform LDS base of a DMA instruction's 1 KB

  auto stage = [&](int ks, int buf) {
    if (DBG & 1) return;
    uint8_t* dst = gx_lds + buf * STAGE;
#pragma unroll
    for (int p = 0; p < 2; ++p)
      __builtin_amdgcn_global_load_lds((gx_glb_vp)(xa[p] + ks * 16), (gx_lds_vp)(dst + p * (GX_NT * 16) + wave_base), 16, 0, 0);
    const uint4* src = wb + (size_t)ks * (STAGE_HALF / 16);
#pragma unroll
    for (int p = 0; p < NB; ++p)
      __builtin_amdgcn_global_load_lds((gx_glb_vp)(src + p * GX_NT), (gx_lds_vp)(dst + GX_A_STAGE + p * (GX_NT * 16) + wave_base), 16, 0, 0);
  };

  gx_f16 acc[2][4];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int nj = 0; nj < 4; ++nj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][nj][r] = 0.f;

  // fragment addresses inside a stage: A row wm * 64 + mi * 32 + li, quads 2 kh and 2 kh + 1 at their swizzled slots (mi * 32 rows = 2 048 bytes: the swizzle
  // term (row >> 2) & 3 does not change with mi); B columns wn * 128 + nj * 32 + li, term plane pairs 8 192 bytes apart
  const int a_row = wm * 64 + li, a_sw = (a_row >> 2) & 3;
  const int fa_off0 = (a_row * 4 + ((2 * kh) ^ a_sw)) * 16, fa_off1 = (a_row * 4 + ((2 * kh + 1) ^ a_sw)) * 16;
  const int fb_off = GX_A_STAGE + (kh * GX_BN + wn * 128 + li) * 16;

  auto load_split_a = [&](int buf, gx_bf8 (*af)[NTERM]) {
    const uint8_t* sa = gx_lds + buf * STAGE;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      // read as the fragment type (the LDS reads of this kernel all have ONE type: hipcc orders a float4 LDS read behind every LDS-DMA in flight
      // -- s_waitcnt vmcnt(0) at the top of each stage -- and leaves the short-vector reads alone)
      const gx_bf8 lo8 = *reinterpret_cast<const gx_bf8*>(sa + fa_off0 + mi * 2048);
      const gx_bf8 hi8 = *reinterpret_cast<const gx_bf8*>(sa + fa_off1 + mi * 2048);
      if (DBG & 8) { af[mi][0] = lo8; af[mi][1] = hi8; af[mi][NTERM - 1] = lo8; }
      else gx_split8_m<MODE>(__builtin_bit_cast(float4, lo8), __builtin_bit_cast(float4, hi8), af[mi]);
    }
  };

  // prologue: stages 0 .. 2 in flight (clamped like in the loop), stages 0 and 1 landed, the A fragments of stage 0 split
  stage(0, 0);
  stage(a.KS > 1 ? 1 : a.KS - 1, 1);
  stage(a.KS > 2 ? 2 : a.KS - 1, 2);
  if (NDMA == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  gx_bf8 af[2][NTERM];
  load_split_a(0, af);

  // one DMA instruction of a stage: pieces 0, 1 = A, 2 .. (4 | 3) = B
  auto stage_piece = [&](int ks, int buf, int piece) {
    if (DBG & 1) return;
    uint8_t* dst = gx_lds + buf * STAGE;
    if (piece < 2) { if (!(DBG & 32)) __builtin_amdgcn_global_load_lds((gx_glb_vp)(xa[piece] + ks * 16), (gx_lds_vp)(dst + piece * (GX_NT * 16) + wave_base), 16, 0, 0); }
    else if (!(DBG & 64)) __builtin_amdgcn_global_load_lds((gx_glb_vp)(wb + (size_t)ks * (STAGE_HALF / 16) + (piece - 2) * GX_NT),
                                          (gx_lds_vp)(dst + GX_A_STAGE + (piece - 2) * (GX_NT * 16) + wave_base), 16, 0, 0);
  };

  int cur = 0;
  for (int ks = 0; ks < a.KS; ++ks) {
    // The DMA of stage ks + 3 goes into the buffer stage ks - 1 was read from: every wave passed the barrier that ended iteration ks - 1 after its last
    // read of it.  Unconditional -- behind the last stage it re-fetches stage KS - 1 into a buffer nobody reads again, 3 wasted stages per tile -- and so
    // is the split of the NEXT stage's A fragments (visible since the barrier that ended iteration ks - 1; behind the last stage its result is dropped):
    // the iteration is straight-line code with ONE counted wait.
    // A CU ingests ~12 bytes per cycle through its vector memory path and a wave that issues into a full queue stalls IN ORDER -- its MFMAs wait with it
    // (measured: memory-only 0.47 ms + compute-only 0.66 ms = 0.92 ms with all five DMA instructions at the top of the iteration).  So the five
    // instructions are dealt out between the four groups of 12 MFMAs, pinned there by sched_barrier: while one wave of a SIMD waits at a DMA, the other
    // one has MFMAs to issue.
    const int nxt = (cur + 1) & 3, sk = ks + 3 < a.KS ? ks + 3 : a.KS - 1, sbuf = (cur + 3) & 3;
    const uint8_t* sb = gx_lds + cur * STAGE;
    const uint8_t* sa = gx_lds + nxt * STAGE;
    gx_bf8 an[2][NTERM];
    gx_bf8 raw[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {   // read as the fragment type (see load_split_a)
      raw[mi][0] = *reinterpret_cast<const gx_bf8*>(sa + fa_off0 + mi * 2048);
      raw[mi][1] = *reinterpret_cast<const gx_bf8*>(sa + fa_off1 + mi * 2048);
    }
#pragma unroll
    for (int nj = 0; nj < 4; ++nj) {
      stage_piece(sk, sbuf, nj);
      if (nj == 3 && NDMA == 5) stage_piece(sk, sbuf, 4);
      __builtin_amdgcn_sched_barrier(0);
      gx_bf8 bf[NTERM];
#pragma unroll
      for (int t = 0; t < NTERM; ++t) bf[t] = *reinterpret_cast<const gx_bf8*>(sb + fb_off + t * 8192 + nj * 512);
      if (nj == 0 || nj == 2) {   // the split of one M tile of the next stage rides on this group's MFMAs
        const int mi = nj >> 1;
        if (DBG & 8) { an[mi][0] = raw[mi][0]; an[mi][1] = raw[mi][1]; an[mi][NTERM - 1] = raw[mi][0]; }
        else gx_split8_m<MODE>(__builtin_bit_cast(float4, raw[mi][0]), __builtin_bit_cast(float4, raw[mi][1]), an[mi]);
      }
      // small products first; (ta, tb): x3 w1, x2 w2, x1 w3, x2 w1, x1 w2, x1 w1; the two M tiles alternate (dependent MFMAs 64 cycles apart)
#define GX_MM(ta, tb)                                                                                    \
  if (DBG & 4) { acc[0][nj][0] += (float)af[0][ta][0] * (float)bf[tb][0]; acc[1][nj][0] += (float)af[1][ta][1] * (float)bf[tb][1]; } else {            \
  acc[0][nj] = gx_mfma<MODE>(af[0][ta], bf[tb], acc[0][nj]);         \
  acc[1][nj] = gx_mfma<MODE>(af[1][ta], bf[tb], acc[1][nj]); }
      if (MODE == 0 && !(DBG & 256)) { GX_MM(NTERM - 1, 0) GX_MM(1, 1) GX_MM(0, NTERM - 1) }   // (NTERM - 1 = 2 here; written so that the fp16x2 instantiation compiles)
      GX_MM(1, 0) GX_MM(0, 1) GX_MM(0, 0)
#undef GX_MM
      if (nj == 0 || nj == 2) {   // one MFMA (32 cycles of the SIMD's matrix pipe), four VALU of the split
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // stage ks + 2 must have landed for everyone: my own DMA of it is older than the 5 instructions of stage ks + 3 just issued.  lgkmcnt(0): this
    // wave's LDS reads of the current buffer have RETURNED before it lets the others go on to overwrite it
    if (DBG & 128) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else
    if (NDMA == 5) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    if (!(DBG & 2)) __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int t = 0; t < NTERM; ++t) af[mi][t] = an[mi][t];
    cur = nxt;
  }

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped DMAs of the last iterations
  // ---- epilogue: accumulator register r of tile (mi, nj) = row (r & 3) + 8 (r >> 2) + 4 kh, column li
#pragma unroll
  for (int nj = 0; nj < 4; ++nj) {
    const int n = n0 + wn * 128 + nj * 32 + li;
    const bool nok = n < a.N;
    float bv = (a.has_bias && nok) ? bias[n] : 0.f;
    asm volatile("" : "+v"(bv));
    float cs = (MODE == 1 && nok) ? colscale[n] : 1.f;   // fp16x2: the weight rows were scaled by a power of two (exact) before their split
    // Both loads are consumed HERE, in front of the masked stores: each store sits in its own exec-masked block, hipcc's scoreboard merges "waited" with "skipped"
    // at every join and would put s_waitcnt vmcnt(0) in front of every one of the 128 stores -- and on gfx9 stores count in vmcnt: a serialised epilogue.
    asm volatile("" : "+v"(cs));
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long long m = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        float v = MODE == 1 ? acc[mi][nj][r] * cs + bv : acc[mi][nj][r] + bv;
        if (a.epilogue == 1) v = gx_gelu(v);
        if (nok && m < a.M && (!(DBG & 16) || v == 12345.678f)) Y[m * (long long)a.N + n] = v;
      }
    }
  }
}

// ---- weights: float32 [N][K] -> the stage images [N tile][K step][term][k-half][n 256][8 x 16 bit]; rows past N are zero.  One thread per (n, 8 k).
// MODE 1 (fp16x2): row n is first multiplied by colscale_inv[n] = 2^e(n), the power of two that brings its largest magnitude into [2^13, 2^14) -- exact, and it
// keeps the SECOND fp16 term of every weight that matters a normal number; the kernel's epilogue multiplies the column's sums by 2^-e(n) (exact).  The scales
// live behind the stage images: float colscale[nbn * 256] = 2^-e(n).
__global__ __launch_bounds__(256) void k_gemm_x3_rowscale(const float* __restrict__ W, int N, int K, int nbn, float* __restrict__ colscale) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;   // one wave per row
  if (n >= nbn * GX_BN) return;
  float mx = 0.f;
  if (n < N) for (int k = lane; k < K; k += 64) mx = fmaxf(mx, fabsf(W[(size_t)n * K + k]));
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  if (lane == 0) {
    int e = 0;
    if (mx > 0.f && mx < INFINITY) { e = 13 - (int)floorf(log2f(mx)); e = e < -100 ? -100 : (e > 100 ? 100 : e); }
    colscale[n] = exp2f((float)-e);   // exact power of two
  }
}
template <int MODE>
__global__ __launch_bounds__(256) void k_gemm_x3_pack_w(const float* __restrict__ W, int N, int K, int nbn, uint4* __restrict__ img, const float* __restrict__ colscale) {
  constexpr int NTERM = GX_NTERM(MODE), STAGE_HALF = GX_STAGE_HALF(MODE);
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const int kc = K / 8;
  const long long total = (long long)nbn * GX_BN * kc;
  if (t >= total) return;
  const int n = (int)(t / kc), c = (int)(t - (long long)n * kc);   // chunk c = 8 consecutive k
  const float sc = MODE == 1 ? 1.0f / colscale[n] : 1.0f;           // 2^e(n): the reciprocal of a power of two is exact
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = n < N ? W[(size_t)n * K + c * 8 + j] * sc : 0.f;
  gx_bf8 o[NTERM];
  gx_split8_m<MODE>(float4{v[0], v[1], v[2], v[3]}, float4{v[4], v[5], v[6], v[7]}, o);
  const int ntile = n / GX_BN, nl = n - ntile * GX_BN, ks = c >> 1, khf = c & 1;
  uint4* base = img + ((size_t)ntile * (K / 16) + ks) * (STAGE_HALF / 16) + khf * GX_BN + nl;
#pragma unroll
  for (int t3 = 0; t3 < NTERM; ++t3) base[t3 * 2 * GX_BN] = __builtin_bit_cast(uint4, o[t3]);
}

static bool gx_mode_ok(int mode) { return mode == 0 || mode == 1; }
long long vd_gemm_x3_weight_bytes(int N, int K, int mode) {
  if (N < 1 || K < 16 || (K & 15) || !gx_mode_ok(mode)) return -1;
  const long long nbn = (N + GX_BN - 1) / GX_BN;
  return nbn * (K / 16) * (long long)(mode == 0 ? GX_STAGE_HALF(0) : GX_STAGE_HALF(1)) + (mode == 1 ? nbn * GX_BN * 4 : 0);
}
static const float* gx_colscale(const void* img, int N, int K, int mode) {
  if (mode != 1) return nullptr;
  const long long nbn = (N + GX_BN - 1) / GX_BN;
  return reinterpret_cast<const float*>(reinterpret_cast<const uint8_t*>(img) + nbn * (K / 16) * (long long)GX_STAGE_HALF(1));
}

bool vd_launch_gemm_x3_pack_w(hipStream_t s, const float* W, int N, int K, void* img, int mode) {
  if (vd_gemm_x3_weight_bytes(N, K, mode) < 0) return false;
  const int nbn = (N + GX_BN - 1) / GX_BN;
  const long long total = (long long)nbn * GX_BN * (K / 8);
  float* cs = const_cast<float*>(gx_colscale(img, N, K, mode));
  if (mode == 1) {
    hipLaunchKernelGGL(k_gemm_x3_rowscale, dim3((unsigned)((nbn * GX_BN + 3) / 4)), dim3(256), 0, s, W, N, K, nbn, cs);
    hipLaunchKernelGGL(k_gemm_x3_pack_w<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, W, N, K, nbn, reinterpret_cast<uint4*>(img), (const float*)cs);
  } else {
    hipLaunchKernelGGL(k_gemm_x3_pack_w<0>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, W, N, K, nbn, reinterpret_cast<uint4*>(img), (const float*)nullptr);
  }
  return true;
}

bool vd_launch_gemm_x3(hipStream_t s, const float* X, long long M, int K, const void* wimg, int N, const float* bias, int epilogue, float* Y, int mode) {
  if (M < 1 || vd_gemm_x3_weight_bytes(N, K, mode) < 0 || epilogue < 0 || epilogue > 1) return false;
  if ((reinterpret_cast<uintptr_t>(X) & 15) || (reinterpret_cast<uintptr_t>(wimg) & 15)) return false;
  static bool attr_set = false;   // idempotent: a race between two first calls sets the same value twice
  static int dbg = 0;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_bf16x3<0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, GX_LDS(0)) != hipSuccess) return false;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_bf16x3<0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, GX_LDS(1)) != hipSuccess) return false;
#ifdef VD_GEMM_ABLATE
    for (const void* f : {reinterpret_cast<const void*>(k_gemm_bf16x3<1, 0>), reinterpret_cast<const void*>(k_gemm_bf16x3<2, 0>), reinterpret_cast<const void*>(k_gemm_bf16x3<4, 0>),
                          reinterpret_cast<const void*>(k_gemm_bf16x3<8, 0>), reinterpret_cast<const void*>(k_gemm_bf16x3<5, 0>), reinterpret_cast<const void*>(k_gemm_bf16x3<3, 0>),
                          reinterpret_cast<const void*>(k_gemm_bf16x3<9, 0>), reinterpret_cast<const void*>(k_gemm_bf16x3<11, 0>), reinterpret_cast<const void*>(k_gemm_bf16x3<16, 0>), reinterpret_cast<const void*>(k_gemm_bf16x3<27, 0>),
                          reinterpret_cast<const void*>(k_gemm_bf16x3<32, 0>), reinterpret_cast<const void*>(k_gemm_bf16x3<64, 0>), reinterpret_cast<const void*>(k_gemm_bf16x3<128, 0>), reinterpret_cast<const void*>(k_gemm_bf16x3<130, 0>), reinterpret_cast<const void*>(k_gemm_bf16x3<256, 0>)})
      if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, GX_LDS(0)) != hipSuccess) return false;
    dbg = getenv("VD3D_GEMM_DBG") ? atoi(getenv("VD3D_GEMM_DBG")) : 0;
#endif
    attr_set = true;
  }
  vd_gx_args a;
  a.M = M; a.K = K; a.N = N; a.KS = K / 16;
  a.nbm = (int)((M + GX_BM - 1) / GX_BM); a.nbn = (N + GX_BN - 1) / GX_BN;
  a.epilogue = epilogue; a.has_bias = bias ? 1 : 0;
  const int per_xcd = (a.nbm + 7) / 8, groups = (per_xcd + GX_MG - 1) / GX_MG;
  const unsigned grid = 8u * (unsigned)groups * (unsigned)GX_MG * (unsigned)a.nbn;
  const float* cs = gx_colscale(wimg, N, K, mode);
  if (mode == 1) {
    hipLaunchKernelGGL((k_gemm_bf16x3<0, 1>), dim3(grid), dim3(GX_NT), GX_LDS(1), s, X, reinterpret_cast<const uint4*>(wimg), bias, cs, Y, a);
    return true;
  }
#ifdef VD_GEMM_ABLATE
#define GX_L(D) hipLaunchKernelGGL((k_gemm_bf16x3<D, 0>), dim3(grid), dim3(GX_NT), GX_LDS(0), s, X, reinterpret_cast<const uint4*>(wimg), bias, cs, Y, a)
  switch (dbg) { case 1: GX_L(1); return true; case 2: GX_L(2); return true; case 4: GX_L(4); return true; case 8: GX_L(8); return true; case 5: GX_L(5); return true;
                 case 3: GX_L(3); return true; case 9: GX_L(9); return true; case 11: GX_L(11); return true; case 16: GX_L(16); return true; case 27: GX_L(27); return true; case 32: GX_L(32); return true; case 64: GX_L(64); return true; case 128: GX_L(128); return true; case 130: GX_L(130); return true; case 256: GX_L(256); return true; default: break; }
#endif
  (void)dbg;
  hipLaunchKernelGGL((k_gemm_bf16x3<0, 0>), dim3(grid), dim3(GX_NT), GX_LDS(0), s, X, reinterpret_cast<const uint4*>(wimg), bias, cs, Y, a);
  return true;
}
