// vd3d_attn.hip -- softmax(Q K^T * scale) V of the depth network's transformer blocks (boundary B3, core/render_depth.py:1106-1119: float32 through the
// Hugging Face pipeline) with BOTH matrix products as split-bf16 MFMA work -- the attention half of the opt-in `gemm="bf16x3"` mode (round 6; the
// default mode keeps PyTorch's scaled_dot_product_attention = AOTriton's float32 kernel, 98 TFLOP/s on gfx950, whose float32-input MFMA runs at 1/16 of
// the bf16 rate).  Same arithmetic idea as vd3d_gemm.hip: every float32 operand (Q, K, V and the probabilities P) is split EXACTLY into three bf16
// terms, six of the nine term products go through v_mfma_f32_32x32x16_bf16 with float32 accumulation (the dropped ones are <= 2^-23 of a product);
// the softmax itself is float32 (running maximum / sum, exp2 of the pre-scaled logits by v_exp_f32).
//
// Two kernels per call:
//   k_attn_x3_prep   qkv [B][T][3][H][64] float32 (the fused QKV linear's output) -> three split + packed images, rows past T zero:
//       Q  [b h][q block 32][k-step 4][term 3][k-half 2][q 32][8 bf16]       (B-operand fragments, read once per wave straight into registers)
//       K  [b h][kv tile 64][k-step 4][term 3][k-half 2][kv 64][8 bf16]      (A-operand fragments of S^T = K Q^T; 24 KB per tile, contiguous)
//       V^T[b h][kv tile 64][k-step 4][term 3][k-half 2][d 64][8 bf16]       (A-operand fragments of O^T = V^T P^T; the 16 kv slots of a k-step are
//                                                                             PERMUTED to where the S^T accumulator leaves them, see below)
//   k_attn_bf16x3    one workgroup = 256 queries of one (batch, head): 8 waves x 32 queries, 512 threads, two waves per SIMD.  Per 64-row KV tile
//       (LDS-DMA into a ring of three 48 KB stages, K three and V^T two tiles ahead, counted vmcnt + raw s_barrier; S^T of tile it + 1 is issued beside the softmax of tile it):
//         S^T[kv 64][q 32] = K Q^T          2 M tiles x 4 k-steps x 6 products = 48 MFMAs.  TRANSPOSED on purpose: the accumulator layout (column =
//                                            lane & 31 = the query, rows = kv in registers) makes every softmax reduction an IN-LANE loop over 32
//                                            registers plus one exchange with lane ^ 32, and the per-query scalars (max, sum, rescale) per-lane values;
//         online softmax                     z = s * (scale * log2 e), m' = max(m, max z), p = exp2(z - m'), l = l * exp2(m - m') + sum p, O *= exp2(m - m')
//         O^T[d 64][q 32] += V^T P^T         P^T as the B operand comes straight out of the S^T accumulator registers: k-step j of M tile m uses registers
//                                            8 j .. 8 j + 7, i.e. kv = 32 m + (e & 3) + 8 (2 j + (e >> 2)) + 4 (lane >> 5) for element e -- the order the
//                                            V^T image is packed in.  No LDS round trip, no cross-lane movement.  2 x 4 x 6 = 48 MFMAs.
//       Epilogue: O^T / l through LDS (272-byte rows: conflict-free 16-byte writes per query) to out[b][t][h][64] in 256-byte row segments.
// Work per tile and wave: 96 MFMAs (3 072 cycles of a SIMD's matrix pipe) against ~500 VALU instructions (softmax + the exact split of 32
// probabilities per lane).  A CU ingests 48 KB per tile = 8 bytes per matrix-pipe cycle: under its ~12 B / cycle vector-memory limit (with 128 queries
// per workgroup it would be 16: the reason for the 8-wave workgroup).
#include "vd3d_dev.h"
#include "vd3d_kernels.h"

typedef short at_bf8 __attribute__((ext_vector_type(8)));
typedef float at_f16 __attribute__((ext_vector_type(16)));
typedef uint32_t at_u4 __attribute__((ext_vector_type(4)));

#define AT_D 64
#define AT_BQ 256
#define AT_BK 64
#define AT_NT 512
// MODE 0 = bf16x3 (three exact bf16 terms, six products), MODE 1 = fp16x2 (two fp16 terms = 22 bits, three products: half the matrix work; q, k, v are scaled
// by 2^4 and the probabilities by 2^10 before their split -- exact -- so that the second terms stay normal fp16 numbers; vd3d_gemm.hip, include/vd3d.h)
#define AT_NTERM(MODE) ((MODE) == 0 ? 3 : 2)
#define AT_TILE(MODE) (4 * AT_NTERM(MODE) * 2 * 64 * 16)   // one K or V^T tile image: 24 576 / 16 384 bytes
#define AT_STAGE(MODE) (2 * AT_TILE(MODE))                 // K + V^T: 49 152 / 32 768
#define AT_NSTAGE 3
#define AT_LDS(MODE) (AT_NSTAGE * AT_STAGE(MODE))          // 147 456 / 98 304
#define AT_QBLK(MODE) (4 * AT_NTERM(MODE) * 2 * 32)        // uint4 per 32-query block of the Q image
#define AT_QKV_SCALE 16.0f                                 // fp16x2: q, k, v times 2^4 ...
#define AT_P_SCALE 1024.0f                                 // ... and p times 2^10 before the split
#define AT_OP 272                             // epilogue: bytes per query row in LDS (256 + 16)

struct vd_at_args {
  int B, H, T;
  int nq32, nkv, nqb;        // 32-query blocks, 64-row KV tiles, 256-query workgroups per (b, h)
  float c;                   // scale * log2(e)
};

VD_DEV void at_split(float a, uint32_t& t1, uint32_t& t2, uint32_t& t3) {
  t1 = __float_as_uint(a) & 0xffff0000u;
  const float r1 = a - __uint_as_float(t1);
  t2 = __float_as_uint(r1) & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(t2);
  t3 = __float_as_uint(r2);
}
VD_DEV uint32_t at_pack(uint32_t lo, uint32_t hi) { return __builtin_amdgcn_perm(hi, lo, 0x07060302u); }
VD_DEV void at_split8(const float v[8], at_bf8 out[3]) {
  uint32_t t1[8], t2[8], t3[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) at_split(v[e], t1[e], t2[e], t3[e]);
  const at_u4 p1 = {at_pack(t1[0], t1[1]), at_pack(t1[2], t1[3]), at_pack(t1[4], t1[5]), at_pack(t1[6], t1[7])};
  const at_u4 p2 = {at_pack(t2[0], t2[1]), at_pack(t2[2], t2[3]), at_pack(t2[4], t2[5]), at_pack(t2[6], t2[7])};
  const at_u4 p3 = {at_pack(t3[0], t3[1]), at_pack(t3[2], t3[3]), at_pack(t3[4], t3[5]), at_pack(t3[6], t3[7])};
  out[0] = __builtin_bit_cast(at_bf8, p1); out[1] = __builtin_bit_cast(at_bf8, p2); out[2] = __builtin_bit_cast(at_bf8, p3);
}
typedef _Float16 at_h8 __attribute__((ext_vector_type(8)));
VD_DEV void at_split8_h(const float v[8], float pre, at_bf8 out[2]) {   // h1 = fp16(x pre), h2 = fp16(x pre - h1), round to nearest
  at_h8 h1, h2;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float x = v[e] * pre;
    const _Float16 a1 = (_Float16)x;
    h1[e] = a1;
    h2[e] = (_Float16)(x - (float)a1);
  }
  out[0] = __builtin_bit_cast(at_bf8, h1); out[1] = __builtin_bit_cast(at_bf8, h2);
}
template <int MODE> VD_DEV void at_split8_m(const float v[8], float pre, at_bf8* out) {
  if (MODE == 0) at_split8(v, out); else at_split8_h(v, pre, out);
}
template <int MODE> VD_DEV at_f16 at_mfma(const at_bf8& a, const at_bf8& b, const at_f16& c) {
  if (MODE == 0) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(at_h8, a), __builtin_bit_cast(at_h8, b), c, 0, 0, 0);
}

// ---- prep: one thread = 8 consecutive elements of one fragment chunk.  which 0 / 1 (Q, K): 8 consecutive d of one token; which 2 (V^T): the 8 kv slots
// of one (k-step, k-half) at one d.
template <int MODE>
__global__ __launch_bounds__(256) void k_attn_x3_prep(const float* __restrict__ qkv, vd_at_args a, uint4* __restrict__ Qimg, uint4* __restrict__ Kimg,
                                                      uint4* __restrict__ Vimg) {
  constexpr int NTERM = AT_NTERM(MODE);
  const int which = blockIdx.z, bh = blockIdx.y, b = bh / a.H, h = bh - b * a.H;
  const int t_id = blockIdx.x * 256 + threadIdx.x;
  const size_t tok_stride = (size_t)3 * a.H * AT_D;
  float v[8];
  uint4* dst;
  if (which < 2) {
    const int rows = which == 0 ? a.nq32 * 32 : a.nkv * 64;
    const int t = t_id >> 3, c = t_id & 7;       // token, chunk of 8 d
    if (t >= rows) return;
    if (t < a.T) {
      const float* src = qkv + ((size_t)b * a.T + t) * tok_stride + (size_t)which * a.H * AT_D + h * AT_D + c * 8;
      const float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
      v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
    }
    const int ks = c >> 1, kh = c & 1;
    if (which == 0) dst = Qimg + ((size_t)bh * a.nq32 + (t >> 5)) * AT_QBLK(MODE) + (size_t)((ks * NTERM) * 2 + kh) * 32 + (t & 31);
    else dst = Kimg + ((size_t)bh * a.nkv + (t >> 6)) * (AT_TILE(MODE) / 16) + (size_t)((ks * NTERM) * 2 + kh) * 64 + (t & 63);
    at_bf8 o[NTERM];
    at_split8_m<MODE>(v, AT_QKV_SCALE, o);
    const int ts = which == 0 ? 2 * 32 : 2 * 64;   // term stride in uint4
#pragma unroll
    for (int t3 = 0; t3 < NTERM; ++t3) dst[(size_t)t3 * ts] = __builtin_bit_cast(uint4, o[t3]);
  } else {
    // V^T: thread = (kv tile, k-step, k-half, d); d fastest so that a wave reads 64 consecutive floats of a token row
    const int d = t_id & 63, kh = (t_id >> 6) & 1, ks = (t_id >> 7) & 3, tile = t_id >> 9;
    if (tile >= a.nkv) return;
    const int m = ks >> 1, j = ks & 1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int kv = tile * 64 + 32 * m + (e & 3) + 8 * (2 * j + (e >> 2)) + 4 * kh;
      v[e] = kv < a.T ? qkv[((size_t)b * a.T + kv) * tok_stride + (size_t)2 * a.H * AT_D + h * AT_D + d] : 0.f;
    }
    at_bf8 o[NTERM];
    at_split8_m<MODE>(v, AT_QKV_SCALE, o);
    dst = Vimg + ((size_t)bh * a.nkv + tile) * (AT_TILE(MODE) / 16) + (size_t)((ks * NTERM) * 2 + kh) * 64 + d;
#pragma unroll
    for (int t3 = 0; t3 < NTERM; ++t3) dst[(size_t)t3 * 128] = __builtin_bit_cast(uint4, o[t3]);
  }
}

typedef __attribute__((address_space(3))) void* at_lds_vp;
typedef const __attribute__((address_space(1))) void* at_glb_vp;

// six products (small first: x3 w1, x2 w2, x1 w3, x2 w1, x1 w2, x1 w1) into TWO accumulators that share the B operand, alternating: dependent MFMAs 64 cycles apart
#define AT_MM1(ACC0, AF0, ACC1, AF1, BF, ta, tb)                                               \
  ACC0 = at_mfma<MODE>(AF0[ta], BF[tb], ACC0);                                                 \
  ACC1 = at_mfma<MODE>(AF1[ta], BF[tb], ACC1);
#define AT_MM6(ACC0, AF0, ACC1, AF1, BF)                                                       \
  if (MODE == 0) { AT_MM1(ACC0, AF0, ACC1, AF1, BF, NTERM - 1, 0) AT_MM1(ACC0, AF0, ACC1, AF1, BF, 1, 1) AT_MM1(ACC0, AF0, ACC1, AF1, BF, 0, NTERM - 1) }                               \
  AT_MM1(ACC0, AF0, ACC1, AF1, BF, 1, 0) AT_MM1(ACC0, AF0, ACC1, AF1, BF, 0, 1) AT_MM1(ACC0, AF0, ACC1, AF1, BF, 0, 0)

template <int MODE>
__global__ __launch_bounds__(AT_NT) void k_attn_bf16x3(const uint4* __restrict__ Qimg, const uint4* __restrict__ Kimg, const uint4* __restrict__ Vimg,
                                                       float* __restrict__ out, vd_at_args a) {
  constexpr int NTERM = AT_NTERM(MODE), TILE = AT_TILE(MODE), STAGE = AT_STAGE(MODE), NP = TILE / (AT_NT * 16);   // DMA instructions per thread and operand tile: 3 / 2
  extern __shared__ __attribute__((aligned(16))) uint8_t at_lds[];   // the only LDS object (see vd3d_gemm.hip)
  // workgroups of one (b, h) share its K / V^T images: keep them on one XCD (workgroup id mod 8; speed only)
  int bh, qb;
  {
    const int wg = blockIdx.x, x = wg & 7, idx = wg >> 3;
    const int g = idx / a.nqb;
    qb = idx - g * a.nqb;
    bh = g * 8 + x;
    if (bh >= a.B * a.H) return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;
  const int q0 = qb * AT_BQ + wave * 32;                 // first query of this wave
  const int wave_base = (tid & ~63) * 16;

  // Q fragments of the wave's 32 queries (zero rows past T: the image is padded to whole 32-query blocks; a wave past the last block reads block nq32 - 1 and
  // stores nothing)
  at_bf8 qf[4][NTERM];
  {
    const int blk = min(q0 >> 5, a.nq32 - 1);
    const uint4* qp = Qimg + ((size_t)bh * a.nq32 + blk) * AT_QBLK(MODE) + kh * 32 + li;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int t = 0; t < NTERM; ++t) qf[ks][t] = __builtin_bit_cast(at_bf8, qp[(ks * NTERM + t) * 64]);
  }
  const uint4* kimg = Kimg + (size_t)bh * a.nkv * (TILE / 16) + tid;
  const uint4* vimg = Vimg + (size_t)bh * a.nkv * (TILE / 16) + tid;
  // one DMA instruction of a tile: pieces 0 .. 2 = K, 3 .. 5 = V^T (512 x 16 bytes each; fp16x2 tiles are two pieces: piece 2 / 5 does nothing there)
  auto dma = [&](int tile, int buf, int piece) {
    const int op = piece / 3, pp = piece - 3 * op;
    if (pp >= NP) return;
    const uint4* src = (op == 0 ? kimg : vimg) + (size_t)tile * (TILE / 16) + pp * AT_NT;
    uint8_t* dst = at_lds + buf * STAGE + op * TILE + pp * (AT_NT * 16) + wave_base;
    __builtin_amdgcn_global_load_lds((at_glb_vp)src, (at_lds_vp)dst, 16, 0, 0);
  };

  at_f16 oacc[2];
#pragma unroll
  for (int dm = 0; dm < 2; ++dm)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[dm][r] = 0.f;
  float m_run = -INFINITY, l_half = 0.f;

  // Software pipeline INSIDE a wave: the S^T MFMAs of tile it + 1 are issued in the same straight-line region as the softmax VALU work of tile it (independent
  // data: hipcc interleaves them, sched_group_barrier says how), then P V of tile it with the split of the next k-step's probabilities between its MFMAs.  A
  // wave that alternated MFMA phases and VALU phases left the matrix pipe idle whenever both waves of a SIMD were in a VALU phase -- and the per-tile barrier
  // keeps them in step (measured: 1.89 ms per DA-V2-Base call at 4K = 0.93 PFLOP/s of MFMA work).
  // So K (t) is read in iteration t - 1 and V^T (t) in iteration t.  Stage s of the ring holds tile t with t % 3 == s; iteration it issues the DMA of K (it + 3)
  // [its stage's K was last read in iteration it - 1] and of V^T (it + 2) [its stage's V^T was last read in iteration it - 1]; the wait at the end lets the six
  // instructions just issued stay in flight and so guarantees K (it + 2) and V^T (it + 1), issued one iteration earlier.
  const int last_tile = a.nkv - 1;
  auto tile_c = [&](int t) { return t < last_tile ? t : last_tile; };   // behind the last tile: re-fetch it into a stage nobody reads again (straight-line code)
#pragma unroll
  for (int p = 0; p < 6; ++p) dma(0, 0, p);                 // K (0), V^T (0)
#pragma unroll
  for (int p = 0; p < 3; ++p) dma(tile_c(1), 1, p);         // K (1)
#pragma unroll
  for (int p = 3; p < 6; ++p) dma(tile_c(1), 1, p);         // V^T (1)
#pragma unroll
  for (int p = 0; p < 3; ++p) dma(tile_c(2), 2, p);         // K (2)
  if (NP == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // K (0), V^T (0), K (1) landed; V^T (1), K (2) in flight
  __builtin_amdgcn_s_barrier();

  const int frag_off = (kh * 64 + li) * 16;     // + ((ks * 3 + t) * 2) * 1024 + m * 512
  // one k-step (16 of the 64 head dimensions) of S^T for both 32-row M tiles: 6 fragment reads, 12 MFMAs
  auto st_step = [&](const uint8_t* sk, int ks, at_f16 (&sa)[2]) {
    at_bf8 kf[2][NTERM];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int t = 0; t < NTERM; ++t) kf[m][t] = *reinterpret_cast<const at_bf8*>(sk + frag_off + ((ks * NTERM + t) * 2) * 1024 + m * 512);
    AT_MM6(sa[0], kf[0], sa[1], kf[1], qf[ks])
  };
  // hint for one pinned chunk: 12 x (one MFMA = 32 cycles of the SIMD's matrix pipe, then up to n VALU), the fragment reads first
#define AT_CHUNK_HINT(n)                                           \
  __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);               \
  _Pragma("unroll") for (int i_ = 0; i_ < 12; ++i_) {              \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);             \
    __builtin_amdgcn_sched_group_barrier(0x002, n, 0);             \
  }                                                                \
  __builtin_amdgcn_sched_barrier(0);

  at_f16 sacc[2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[m][r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) st_step(at_lds, ks, sacc);   // S^T (0)
  // the first iteration's DMA of K (3) overwrites K (0): every wave's reads of it have returned first (found the hard way: one shape in twenty, under load)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  int cur = 0;
  for (int it = 0; it < a.nkv; ++it) {
    const int s1 = cur == 2 ? 0 : cur + 1, s2 = cur >= 1 ? cur - 1 : 2;   // stages of tiles it + 1 and it + 2 (= it - 1)
    const uint8_t* sk1 = at_lds + s1 * STAGE;
    const uint8_t* sv = at_lds + cur * STAGE + TILE;
    // ---- tile it: logits -> z (pre-scaled; rows past T masked in the last tile: a uniform branch off the hot path)
    float z[2][16];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) z[m][r] = sacc[m][r] * a.c;
    if ((it + 1) * AT_BK > a.T) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (it * AT_BK + 32 * m + (r & 3) + 8 * (r >> 2) + 4 * kh >= a.T) z[m][r] = -INFINITY;
    }
    // ---- region A: S^T (it + 1) on the matrix pipe || softmax (it) on the VALU, in four pinned chunks of 12 MFMAs
    at_f16 sn[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) sn[m][r] = 0.f;
    // chunk 0: running maximum
    dma(tile_c(it + 3), cur, 0);                       // K (it + 3) -> this tile's stage (its K was consumed in iteration it - 1)
    __builtin_amdgcn_sched_barrier(0);
    st_step(sk1, 0, sn);
    float mx = -INFINITY;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, z[m][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    AT_CHUNK_HINT(2)
    // chunks 1, 2: the exponentials of one M tile each
    float ps = 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      dma(tile_c(it + 3), cur, 1 + m);
      __builtin_amdgcn_sched_barrier(0);
      st_step(sk1, 1 + m, sn);
#pragma unroll
      for (int r = 0; r < 16; ++r) { z[m][r] = __builtin_amdgcn_exp2f(z[m][r] - m_new); ps += z[m][r]; }
      AT_CHUNK_HINT(4)
    }
    // chunk 3: rescale O, split the first k-step's probabilities
    dma(tile_c(it + 2), s2, 3);                        // V^T (it + 2) -> the stage of tile it - 1 (its V^T was consumed in iteration it - 1)
    __builtin_amdgcn_sched_barrier(0);
    st_step(sk1, 3, sn);
    l_half = l_half * alpha + ps;
#pragma unroll
    for (int dm = 0; dm < 2; ++dm)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[dm][r] *= alpha;
    at_bf8 pf[2][NTERM];
    at_split8_m<MODE>(&z[0][0], AT_P_SCALE, pf[0]);
    AT_CHUNK_HINT(7)
    // ---- region B: O^T += V^T P^T (tile it); the split of k-step ks + 1 rides on the MFMAs of k-step ks
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks < 2) { dma(tile_c(it + 2), s2, 4 + ks); __builtin_amdgcn_sched_barrier(0); }
      if (ks < 3) at_split8_m<MODE>(&z[(ks + 1) >> 1][8 * ((ks + 1) & 1)], AT_P_SCALE, pf[(ks + 1) & 1]);
      at_bf8 vf[2][NTERM];
#pragma unroll
      for (int dm = 0; dm < 2; ++dm)
#pragma unroll
        for (int t = 0; t < NTERM; ++t) vf[dm][t] = *reinterpret_cast<const at_bf8*>(sv + frag_off + ((ks * NTERM + t) * 2) * 1024 + dm * 512);
      AT_MM6(oacc[0], vf[0], oacc[1], vf[1], pf[ks & 1])
      AT_CHUNK_HINT(4)
    }
    // K (it + 2) and V^T (it + 1) have landed for everyone (a wave's own DMAs of them are older than the six instructions it just issued), and this wave's LDS
    // reads of the stages it used have returned, before anybody overwrites them
    if (NP == 3) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int m = 0; m < 2; ++m) sacc[m] = sn[m];
    cur = s1;
  }
#undef AT_CHUNK_HINT
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the re-fetches of the last iterations have landed ...
  __builtin_amdgcn_s_barrier();                        // ... everybody's, before the epilogue re-uses the ring

  // ---- epilogue: O^T / l -> LDS [q 32 per wave][d 64] with 272-byte rows -> out[b][t][h][d]
  const float l_tot = l_half + __shfl_xor(l_half, 32, 64);
  const float inv = MODE == 1 ? 1.0f / (l_tot * (AT_P_SCALE * AT_QKV_SCALE)) : 1.0f / l_tot;   // fp16x2: O accumulated (2^10 p)(2^4 v)
  uint8_t* ow = at_lds + wave * (32 * AT_OP);
#pragma unroll
  for (int dm = 0; dm < 2; ++dm)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 v4 = {oacc[dm][4 * g] * inv, oacc[dm][4 * g + 1] * inv, oacc[dm][4 * g + 2] * inv, oacc[dm][4 * g + 3] * inv};
      *reinterpret_cast<float4*>(ow + li * AT_OP + (32 * dm + 8 * g + 4 * kh) * 4) = v4;
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own wave's rows only: no barrier needed (wave-private region; LDS ops of a wave complete in order)
  const int b = bh / a.H, h = bh - b * a.H;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = i * 4 + (lane >> 4), c4 = lane & 15;   // 16 lanes x 16 bytes = one query's 256 bytes
    const int q = q0 + row;
    const float4 v4 = *reinterpret_cast<const float4*>(ow + row * AT_OP + c4 * 16);
    if (q < a.T) *reinterpret_cast<float4*>(out + (((size_t)b * a.T + q) * a.H + h) * AT_D + c4 * 4) = v4;
  }
}

static bool at_mode_ok(int mode) { return mode == 0 || mode == 1; }
long long vd_attn_x3_workspace_bytes(int B, int T, int H, int D, int mode) {
  if (B < 1 || T < 1 || H < 1 || D != AT_D || !at_mode_ok(mode)) return -1;
  const long long nq32 = (T + 31) / 32, nkv = (T + AT_BK - 1) / AT_BK;
  const long long qblk = mode == 0 ? AT_QBLK(0) : AT_QBLK(1), tile = mode == 0 ? AT_TILE(0) : AT_TILE(1);
  return (long long)B * H * (nq32 * qblk * 16 + 2 * nkv * tile);
}

template <int MODE>
static bool at_launch(hipStream_t s, const float* qkv, int B, int T, int H, float scale, void* ws, float* out) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_bf16x3<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, AT_LDS(MODE)) != hipSuccess) return false;
    attr_set = true;
  }
  vd_at_args a;
  a.B = B; a.H = H; a.T = T;
  a.nq32 = (T + 31) / 32; a.nkv = (T + AT_BK - 1) / AT_BK; a.nqb = (T + AT_BQ - 1) / AT_BQ;
  a.c = scale * 1.44269504088896340736f;
  if (MODE == 1) a.c *= 1.0f / (AT_QKV_SCALE * AT_QKV_SCALE);   // the logits were computed from 2^4 q and 2^4 k (exact rescale)
  uint4* Qimg = reinterpret_cast<uint4*>(ws);
  uint4* Kimg = Qimg + (size_t)B * H * a.nq32 * AT_QBLK(MODE);
  uint4* Vimg = Kimg + (size_t)B * H * a.nkv * (AT_TILE(MODE) / 16);
  const int rows_max = a.nkv * 64 > a.nq32 * 32 ? a.nkv * 64 : a.nq32 * 32;   // Q / K: rows * 8 threads; V^T: tiles * 512 threads = the same count
  hipLaunchKernelGGL(k_attn_x3_prep<MODE>, dim3((unsigned)((rows_max * 8 + 255) / 256), (unsigned)(B * H), 3), dim3(256), 0, s, qkv, a, Qimg, Kimg, Vimg);
  const int groups = (B * H + 7) / 8;
  hipLaunchKernelGGL(k_attn_bf16x3<MODE>, dim3((unsigned)(8 * groups * a.nqb)), dim3(AT_NT), AT_LDS(MODE), s, Qimg, Kimg, Vimg, out, a);
  return true;
}

bool vd_launch_attn_x3(hipStream_t s, const float* qkv, int B, int T, int H, int D, float scale, void* ws, float* out, int mode) {
  if (vd_attn_x3_workspace_bytes(B, T, H, D, mode) < 0) return false;
  if ((reinterpret_cast<uintptr_t>(qkv) & 15) || (reinterpret_cast<uintptr_t>(ws) & 15) || (reinterpret_cast<uintptr_t>(out) & 15)) return false;
  if ((long long)B * H > 65535) return false;
  return mode == 0 ? at_launch<0>(s, qkv, B, T, H, scale, ws, out) : at_launch<1>(s, qkv, B, T, H, scale, ws, out);
}
