// vd3d_atensum.hip -- torch.mean's float32 summation ORDER on the device (round 5; vd3d_render_params::aten_sum_threads > 0).
//
// compute_dynamic_parallax_scale (core/render_3d.py:418) and compute_motion_metric (:928) call torch.mean on float32 tensors; its value is
// that of ATen's cascade sum (SumKernel.cpp, cascade_sum<false, float>), divided by float(n) -- NOT the correctly rounded mean: about one
// frame in 300 gets a parallax scale one ULP away, which moves an eye sample by a level.  The algorithm (restated in full in
// oracle/vd3d_oracle.c::vo_sum_aten_2d, pinned against torch for 1 .. 64 threads): the flattened [rows][cols] space is cut into
// min(T, ceil(numel / 32768)) contiguous ranges (one per torch thread), a range is walked row piece by row piece, every piece is reduced by
// the vectorized inner sum -- 8 lanes x 4 interleaved accumulators = 32 independent chains, each with a 4-level cascade whose level-0
// blocks are 16 steps long -- and added to the range's float32 partial in walk order; the T partials are reduced by the same inner sum.
// On the device:
//   k_aten_small  one WAVE per piece shorter than 8 192 elements (every row of the centre crop): lanes 0..31 are the 32 chains;
//   k_aten_big    one WORKGROUP per longer piece (a thread's share of the contiguous |d_t - d_{t-1}| plane): level-0 blocks are independent
//                 (16 adds from zero), so 32 x 32 threads build 256 of them at a time in LDS and 32 chain threads fold them in order;
//   k_aten_final  one thread per frame and sum: the pieces of every range in walk order, then the T partials.
// Same float32 additions in the same order as ATen's: bit-identical to the oracle (tests/test_hip_parity.py::test_aten_sum_order_*).
#include <cstring>
#include <vector>

#include "vd3d_dev.h"
#include "vd3d_kernels.h"

#define AS_BIG 8192   // pieces of at least this many elements take a whole workgroup (>= 256 steps per chain = 16 level-0 blocks)

// value source of the two sums: job 0 = the normalised plane itself (centre crop), job 1 = |dn - previous normalised plane| (the previous plane is
// normalised on the fly in the measure / replay sharding, exactly like k_chain_norm does for its exact sum)
struct as_src {
  const float* dn; const float* prev;
  int job, m3, p_col;
  float p_lo, p_den;
  VD_DEV float at(long long i) const {
    const float v = dn[i];
    if (job == 0) return v;
    float vp = prev[i];
    if (m3) { const float dp = vd_clamp(vp, 0.f, 1.f); vp = p_col ? dp : vd_clamp((dp - p_lo) / p_den, 0.f, 1.f); }
    return fabsf(v - vp);
  }
};

VD_DEV as_src as_make_src(const vd_batch_frame& F, const vd_stage_args& a, int job) {
  as_src s;
  s.dn = F.dn; s.prev = F.dn_prev; s.job = job;
  s.m3 = a.shard == 3 ? 1 : 0; s.p_col = 0; s.p_lo = 0.f; s.p_den = 1.f;
  if (s.m3) { const float* e = a.etab + VD_ETAB * F.shard_idx; s.p_lo = e[0]; s.p_den = e[1]; s.p_col = (int)e[2]; }
  return s;
}
// does frame F have a previous plane (compute_motion_metric returns 0.0 without one: the sum is not needed)
VD_DEV bool as_have_prev(const vd_batch_frame& F, const vd_stage_args& a) {
  return a.shard == 3 ? ((int)a.etab[VD_ETAB * F.shard_idx + 3] != 0) : (F.w->st.prev_depth_valid != 0);
}

// the epilogue every piece shares: chain accumulators a0[k][l] (k = lane >> 3, l = lane & 7 of lanes 0..31) -> leftover vectors into accumulator 0,
// fold 0 += 1, += 2, += 3, tail elements from zero, then the 8 lanes in order.  Called by a full wave; the result is valid in lane 0.
VD_DEV float as_finish_wave(const as_src& s, long long off, int len, float a0, int lane) {
  const int nv = len >> 3, nq = nv >> 2, k = (lane >> 3) & 3, l = lane & 7;
  if (lane < 32 && k == 0)
    for (int v = 4 * nq; v < nv; ++v) a0 += s.at(off + (long long)v * 8 + l);
  float p = a0;                                   // lanes 0..7 (k = 0): accumulator 0 of lane l
  p += __shfl(a0, (lane & 7) + 8, 64);
  p += __shfl(a0, (lane & 7) + 16, 64);
  p += __shfl(a0, (lane & 7) + 24, 64);
  float fin = 0.f;
  for (int t = 8 * nv; t < len; ++t) fin += s.at(off + t);
  for (int j = 0; j < 8; ++j) fin += __shfl(p, j, 64);
  return fin;
}
// a piece of fewer than 8 elements: the scalar row_sum (element 4 i + k -> accumulator k, leftovers -> 0, fold)
VD_DEV float as_tiny(const as_src& s, long long off, int len) {
  float p[4] = {0.f, 0.f, 0.f, 0.f};
  const int q = len >> 2;   // 0 or 1
  for (int i = 0; i < q; ++i)
    for (int kk = 0; kk < 4; ++kk) p[kk] += s.at(off + 4 * i + kk);
  for (int i = 4 * q; i < len; ++i) p[0] += s.at(off + i);
  return ((p[0] + p[1]) + p[2]) + p[3];
}

struct as_plan { const int4* pieces; int n_small, n_big, n_all, T; };   // pieces: {plane offset, length, range (torch thread), job}; small ones first

__global__ __launch_bounds__(1024) void k_aten_small(vd_batch b, vd_stage_args a, as_plan pl, float* __restrict__ scratch) {
  const vd_batch_frame& F = b.f[blockIdx.y];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pi = blockIdx.x * 16 + wave;
  if (pi >= pl.n_small) return;
  const int4 pc = pl.pieces[pi];
  float* out = scratch + (size_t)blockIdx.y * pl.n_all + pi;
  if (pc.w == 1 && !as_have_prev(F, a)) { if (lane == 0) *out = 0.f; return; }
  const as_src s = as_make_src(F, a, pc.w);
  const long long off = pc.x;
  const int len = pc.y;
  if (len < 8) { if (lane == 0) *out = as_tiny(s, off, len); return; }
  const int nq = (len >> 3) >> 2, k = (lane >> 3) & 3, l = lane & 7;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (lane < 32) {   // one chain per lane: element ((4 i + k) * 8 + l) for step i; level-0 blocks of 16 steps (level_power 4: nq < 2^16)
    int i = 0;
    for (; i + 16 <= nq;) {
      float x[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) x[j] = s.at(off + ((long long)(4 * (i + j) + k) * 8 + l));
#pragma unroll
      for (int j = 0; j < 16; ++j) a0 += x[j];
      i += 16;
      a1 += a0; a0 = 0.f;
      if ((i & 0xF0) == 0) { a2 += a1; a1 = 0.f; if ((i & 0xF00) == 0) { a3 += a2; a2 = 0.f; } }
    }
    for (; i < nq; ++i) a0 += s.at(off + ((long long)(4 * i + k) * 8 + l));
    a0 += a1; a0 += a2; a0 += a3;
  }
  const float r = as_finish_wave(s, off, len, a0, lane);
  if (lane == 0) *out = r;
}

__global__ __launch_bounds__(1024) void k_aten_big(vd_batch b, vd_stage_args a, as_plan pl, float* __restrict__ scratch) {
  __shared__ float B0[256][32];       // level-0 block sums of one super-block (256 blocks = 4 096 steps) per chain
  __shared__ float fold[32];
  const vd_batch_frame& F = b.f[blockIdx.y];
  const int tid = threadIdx.x, lane = tid & 63;
  const int pi = pl.n_small + blockIdx.x;
  const int4 pc = pl.pieces[pi];
  float* out = scratch + (size_t)blockIdx.y * pl.n_all + pi;
  if (pc.w == 1 && !as_have_prev(F, a)) { if (tid == 0) *out = 0.f; return; }   // workgroup-uniform
  const as_src s = as_make_src(F, a, pc.w);
  const long long off = pc.x;
  const int len = pc.y;
  const int nv = len >> 3, nq = nv >> 2;
  const int nblk = nq >> 4;                                  // full level-0 blocks
  const int ch = tid & 31, k = ch >> 3, l = ch & 7, bsub = tid >> 5;   // thread = (block of the current group of 32, chain)
  float a1 = 0.f, a2 = 0.f, a3 = 0.f;                       // live in the 32 chain threads (tid < 32)
  for (int sb = 0; sb * 256 < nblk; ++sb) {                 // super-blocks of 256 level-0 blocks
    const int nb = min(256, nblk - sb * 256);
    for (int g = bsub; g < nb; g += 32) {
      const long long i0 = (long long)(sb * 256 + g) * 16;
      float x[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) x[j] = s.at(off + ((4 * (i0 + j) + k) * 8 + l));
      float v = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) v += x[j];
      B0[g][ch] = v;
    }
    __syncthreads();
    if (tid < 32) {
      for (int g = 0; g < nb; ++g) {
        a1 += B0[g][ch];                                    // the flushed accumulator 0 (16 adds from zero)
        const int i = (sb * 256 + g + 1) * 16;              // steps done
        if ((i & 0xF0) == 0) { a2 += a1; a1 = 0.f; if ((i & 0xF00) == 0) { a3 += a2; a2 = 0.f; } }
      }
    }
    __syncthreads();
  }
  float a0 = 0.f;
  if (tid < 32) {
    for (int i = nblk * 16; i < nq; ++i) a0 += s.at(off + ((long long)(4 * i + k) * 8 + l));   // fewer than 16 leftover steps
    a0 += a1; a0 += a2; a0 += a3;
  }
  if (tid < 64) {   // wave 0: lanes 0..31 hold the chains
    const float r = as_finish_wave(s, off, len, a0, lane);
    if (lane == 0) *out = r;
  }
  (void)fold;
}

// serial inner sum of a short float array (the T per-thread partials): the same algorithm, one thread
VD_DEV float as_serial_piece(const float* x, int n) {
  if (n < 8) {
    float p[4] = {0.f, 0.f, 0.f, 0.f};
    const int q = n >> 2;
    for (int i = 0; i < q; ++i)
      for (int kk = 0; kk < 4; ++kk) p[kk] += x[4 * i + kk];
    for (int i = 4 * q; i < n; ++i) p[0] += x[i];
    return ((p[0] + p[1]) + p[2]) + p[3];
  }
  const int nv = n >> 3, nq = nv >> 2;
  float part[4][8];
  for (int kk = 0; kk < 4; ++kk)
    for (int l = 0; l < 8; ++l) {   // one chain at a time (n <= 1 024: nq <= 32, at most two level-0 blocks, no higher level is flushed twice)
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      int i = 0;
      for (; i + 16 <= nq;) {
        for (int j = 0; j < 16; ++j, ++i) a0 += x[(4 * i + kk) * 8 + l];
        a1 += a0; a0 = 0.f;
        if ((i & 0xF0) == 0) { a2 += a1; a1 = 0.f; if ((i & 0xF00) == 0) { a3 += a2; a2 = 0.f; } }
      }
      for (; i < nq; ++i) a0 += x[(4 * i + kk) * 8 + l];
      a0 += a1; a0 += a2; a0 += a3;
      part[kk][l] = a0;
    }
  for (int v = 4 * nq; v < nv; ++v)
    for (int l = 0; l < 8; ++l) part[0][l] += x[v * 8 + l];
  for (int kk = 1; kk < 4; ++kk)
    for (int l = 0; l < 8; ++l) part[0][l] += part[kk][l];
  float fin = 0.f;
  for (int i = 8 * nv; i < n; ++i) fin += x[i];
  for (int l = 0; l < 8; ++l) fin += part[0][l];
  return fin;
}

#define AS_MAX_T 1024   // torch.get_num_threads() of the reference process: 8 KB of LDS for the two partial buffers (256 until round 5)
__global__ __launch_bounds__(64) void k_aten_final(vd_batch b, vd_stage_args a, as_plan pl, const float* __restrict__ scratch, int nr_crop, int nr_mad) {
  __shared__ float buf[2][AS_MAX_T];
  const vd_batch_frame& F = b.f[blockIdx.x];
  const int job = threadIdx.x;          // thread 0: centre-crop sum, thread 1: |difference| sum
  if (job >= 2) return;
  const float* ps = scratch + (size_t)blockIdx.x * pl.n_all;
  const int nr = job == 0 ? nr_crop : nr_mad;   // ranges (torch threads that got work); 1 = the serial path: no buffer pass
  float* bf = buf[job];
  for (int t = 0; t < pl.T; ++t) bf[t] = 0.f;
  float single = 0.f;
  for (int i = 0; i < pl.n_all; ++i) {   // pieces are stored in walk order per job (small and big interleave only across jobs)
    const int4 pc = pl.pieces[i];
    if (pc.w != job) continue;
    if (nr <= 1) single += ps[i]; else bf[pc.z] += ps[i];
  }
  const float sum = nr <= 1 ? single : as_serial_piece(bf, pl.T);
  if (job == 0) F.w->aten_sum_mean = sum; else F.w->aten_sum_mad = sum;
}

// ---- host: the piece plan of one (eye size, thread count) -------------------------------------------------------------------------------
// serial_for_each over the flattened [R][C] space restricted to [b, e): rest of a started row, whole rows, head of the last row
static void as_walk(std::vector<int4>& out, long long base, long long R, long long C, long long row_stride, long long b, long long e, int range, int job) {
  long long off = b;
  while (off < e) {
    const long long r = off / C, c = off - r * C;
    long long step0 = C - c < e - off ? C - c : e - off, step1 = 1;
    if (step0 == C) { step1 = (e - off) / C; if (step1 > R - r) step1 = R - r; }
    for (long long j = 0; j < step1; ++j) out.push_back(make_int4((int)(base + (r + j) * row_stride + c), (int)step0, range, job));
    off += step0 * step1;
  }
}
static int as_ranges(std::vector<int4>& out, long long base, long long R, long long C, long long row_stride, int T, int job) {
  const long long numel = R * C, grain = 32768;
  if (numel < grain || T <= 1) { as_walk(out, base, R, C, row_stride, 0, numel, 0, job); return 1; }
  long long nt = (numel + grain - 1) / grain;
  if (nt > T) nt = T;
  const long long chunk = (numel + nt - 1) / nt;
  int t = 0;
  for (long long b = 0; b < numel; b += chunk, ++t) as_walk(out, base, R, C, row_stride, b, b + chunk < numel ? b + chunk : numel, t, job);
  return t;
}

// builds the plan (host vectors); the caller uploads `pieces` and keeps the counts.  Returns false when a size is outside the kernels' range.
bool vd_aten_plan_build(int eh, int ew, int T, std::vector<int>& pieces_flat, int* n_small, int* n_big, int* nr_crop, int* nr_mad) {
  if (T < 1 || T > AS_MAX_T || (long long)eh * ew >= (1ll << 31)) return false;
  std::vector<int4> crop, mad;
  const int y0 = eh / 4, y1 = eh * 3 / 4, x0 = ew / 4, x1 = ew * 3 / 4;
  *nr_crop = as_ranges(crop, (long long)y0 * ew + x0, y1 - y0, x1 - x0, ew, T, 0);
  *nr_mad = as_ranges(mad, 0, 1, (long long)eh * ew, (long long)eh * ew, T, 1);
  // k_aten_final adds the pieces of a range in the order of this table; the table keeps the small pieces of both sums in front of the big ones, so a sum
  // whose ranges mix small and big pieces would lose its walk order: centre-crop rows are always small (eyes narrower than 16 384), a range of the contiguous
  // difference plane is a single piece
  std::vector<int4> small, big;
  for (const auto* v : {&crop, &mad})
    for (const int4& p : *v) {
      if ((long long)p.y > (1ll << 24)) return false;   // level_power stays 4 up to 2^19 steps per chain = 2^24 elements
      if (p.y >= AS_BIG && p.w == 0) return false;
      (p.y >= AS_BIG ? big : small).push_back(p);
    }
  *n_small = (int)small.size(); *n_big = (int)big.size();
  pieces_flat.clear();
  for (const auto* v : {&small, &big})
    for (const int4& p : *v) { pieces_flat.push_back(p.x); pieces_flat.push_back(p.y); pieces_flat.push_back(p.z); pieces_flat.push_back(p.w); }
  return true;
}

void vd_launch_aten_sums(hipStream_t s, const vd_batch& b, const vd_stage_args& a) {
  as_plan pl;
  pl.pieces = reinterpret_cast<const int4*>(a.aten_plan); pl.n_small = a.aten_n_small; pl.n_big = a.aten_n_big; pl.n_all = a.aten_n_small + a.aten_n_big;
  pl.T = a.aten_threads;
  const unsigned nf = (unsigned)b.n;
  if (pl.n_small) hipLaunchKernelGGL(k_aten_small, dim3((pl.n_small + 15) / 16, nf), dim3(1024), 0, s, b, a, pl, a.aten_scratch);
  if (pl.n_big) hipLaunchKernelGGL(k_aten_big, dim3(pl.n_big, nf), dim3(1024), 0, s, b, a, pl, a.aten_scratch);
  hipLaunchKernelGGL(k_aten_final, dim3(nf), dim3(64), 0, s, b, a, pl, (const float*)a.aten_scratch, a.aten_nr_crop, a.aten_nr_mad);
}
