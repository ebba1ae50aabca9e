// ubench_valu.hip -- issue-rate micro-benchmarks on gfx950 that the E1 / W1 designs depend on (development aid, not product code):
//   * v_fma_f32 vs v_pk_fma_f32 vs v_pk_mul_f32 / v_pk_add_f32 throughput per SIMD at 1 / 2 / 4 / 8 waves per SIMD
//   * f32-input MFMA (32x32x2, 16x16x4) alone and next to VALU-only waves on the same SIMDs (is the matrix pipe free capacity?)
//   * ds_read_b128 streams next to packed FMAs (LDS bytes per clock a VALU-bound stencil can afford)
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/ubench_valu.bin ; run on the GPU box: tools/ubench_valu.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define ITER 2048

// mode 0: scalar fma   1: packed fma   2: packed mul+add alternating   3: scalar fma + packed fma interleaved 1:1
template <int MODE>
__global__ __launch_bounds__(256) void k_valu(float* out, long long* cyc, float s) {
  float a[8];
  f2 p[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = f2{a[i], a[i] + 1.f}; }
  const float b = s, c = 0.5f;
  const f2 pb = {s, s}, pc = {0.5f, 0.25f};
  const long long t0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
      if (MODE == 2) {
        if (i & 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
        else asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
      }
      if (MODE == 3) {
        if (i & 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        else asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
      }
    }
  }
  const long long t1 = clock64();
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) r += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
  if (blockIdx.x == 0 && threadIdx.x == 0) { cyc[256 * 8 * 8] = t1 - t0; cyc[256 * 8 * 8 + 1] = wall_clock64() - w0; }   // shader cycles, 100 MHz ticks
}

// dependent-chain test: NCH independent v_fma_f32 accumulators per wave, each instruction depends on the one NCH places earlier
// (E1's dense DOF loop has 4: the four pixels of a strip; does the VALU pipeline need more to issue back to back?)
template <int NCH>
__global__ __launch_bounds__(256) void k_chain(float* out, float s) {
  float a[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) a[i] = threadIdx.x * 0.001f + i;
  const float b = s, c = 0.5f;
  for (int it = 0; it < ITER * 8 / NCH; ++it) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
  }
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) r += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// MFMA f32 waves and VALU waves in one workgroup: waves [0, nm) run MFMA chains, the rest run v_fma / v_pk_fma chains.
// MF: 0 = 32x32x2 (4 independent accumulators), 1 = 16x16x4 (8 independent accumulators); PK: VALU waves use packed fma
template <int MF, int PK>
__global__ __launch_bounds__(512) void k_mix(float* out, long long* cyc, float s, int nm) {
  const int wv = threadIdx.x >> 6;
  const long long t0 = clock64();
  float r = 0.f;
  if (wv < nm) {
    if (MF == 0) {
      f16v acc[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
      const float a = s + threadIdx.x * 0.01f, b = 0.5f;
      for (int it = 0; it < ITER / 16; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][7];
    } else {
      f4 acc[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
      const float a = s + threadIdx.x * 0.01f, b = 0.5f;
      for (int it = 0; it < ITER / 16; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][3];
    }
  } else {
    float a[8];
    f2 p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = f2{a[i], a[i] + 1.f}; }
    const float b = s, c = 0.5f;
    const f2 pb = {s, s}, pc = {0.5f, 0.25f};
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
        else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) r += a[i] + p[i].x + p[i].y;
  }
  const long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// packed fma with R ds_read_b128 per 8 packed fmas (window loads of a stencil): LDS bandwidth next to a VALU-bound loop
template <int R>
__global__ __launch_bounds__(256) void k_lds(float* out, long long* cyc, float s) {
  __shared__ f4 tile[2048];
  for (int i = threadIdx.x; i < 2048; i += 256) tile[i] = f4{s, s + 1.f, s + 2.f, (float)i};
  __syncthreads();
  f2 p[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) p[i] = f2{threadIdx.x * 0.001f + i, 1.f};
  const f2 pc = {0.5f, 0.25f};
  int idx = threadIdx.x;
  const long long t0 = clock64();
  for (int it = 0; it < ITER; ++it) {
    f4 w[R > 0 ? R : 1];
#pragma unroll
    for (int j = 0; j < R; ++j) w[j] = tile[(idx + 64 * j + it) & 2047];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const f2 m = R > 0 ? f2{w[i % (R > 0 ? R : 1)][i & 3], w[i % (R > 0 ? R : 1)][(i + 1) & 3]} : f2{s, s};
      p[i] = __builtin_elementwise_fma(p[i], m, pc);
    }
  }
  const long long t1 = clock64();
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) r += p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// FETCH_SIZE calibration (MI355X_MICROARCH.md "HBM": the counter is only calibrated for 16 B/lane streams): read NB bytes once with
// 4 / 12 / 16 bytes per lane; run under `rocprofv3 --pmc FETCH_SIZE` and compare the counter with NB.
template <int WIDTH>
__global__ __launch_bounds__(256) void k_read(const uint32_t* __restrict__ src, size_t nwords, uint32_t* out) {
  uint32_t acc = 0;
  const size_t stride = (size_t)gridDim.x * blockDim.x * WIDTH;
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * WIDTH; i + WIDTH <= nwords; i += stride) {
#pragma unroll
    for (int j = 0; j < WIDTH; ++j) acc ^= src[i + j];
  }
  if (acc == 0x12345678u) out[0] = acc;
}

// streaming-copy variants (the yardstick kernel of bench.py): which access pattern reaches the guide's 6.29 TB/s float4 copy?
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
template <int NT>   // A: grid-stride, four loads in flight a whole grid apart (the round-2 kernel)
__global__ __launch_bounds__(256) void k_copy_a(const u4* __restrict__ src, u4* __restrict__ dst, size_t n16) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    u4 a, b, c, d;
    if (NT) { a = __builtin_nontemporal_load(src + i); b = __builtin_nontemporal_load(src + i + stride); c = __builtin_nontemporal_load(src + i + 2 * stride); d = __builtin_nontemporal_load(src + i + 3 * stride); }
    else { a = src[i]; b = src[i + stride]; c = src[i + 2 * stride]; d = src[i + 3 * stride]; }
    if (NT) { __builtin_nontemporal_store(a, dst + i); __builtin_nontemporal_store(b, dst + i + stride); __builtin_nontemporal_store(c, dst + i + 2 * stride); __builtin_nontemporal_store(d, dst + i + 3 * stride); }
    else { dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d; }
  }
  for (; i < n16; i += stride) dst[i] = src[i];
}
template <int NT>   // B: every workgroup owns one contiguous chunk; four loads in flight 4 KB apart
__global__ __launch_bounds__(256) void k_copy_b(const u4* __restrict__ src, u4* __restrict__ dst, size_t n16) {
  const size_t chunk = (n16 + gridDim.x - 1) / gridDim.x;
  const size_t beg = (size_t)blockIdx.x * chunk, end = beg + chunk < n16 ? beg + chunk : n16;
  size_t i = beg + threadIdx.x;
  for (; i + 768 < end; i += 1024) {
    u4 a, b, c, d;
    if (NT) { a = __builtin_nontemporal_load(src + i); b = __builtin_nontemporal_load(src + i + 256); c = __builtin_nontemporal_load(src + i + 512); d = __builtin_nontemporal_load(src + i + 768); }
    else { a = src[i]; b = src[i + 256]; c = src[i + 512]; d = src[i + 768]; }
    if (NT) { __builtin_nontemporal_store(a, dst + i); __builtin_nontemporal_store(b, dst + i + 256); __builtin_nontemporal_store(c, dst + i + 512); __builtin_nontemporal_store(d, dst + i + 768); }
    else { dst[i] = a; dst[i + 256] = b; dst[i + 512] = c; dst[i + 768] = d; }
  }
  for (; i < end; i += 256) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void k_copy_c(const u4* __restrict__ src, u4* __restrict__ dst, size_t n16) {   // C: one element per thread
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n16) dst[i] = src[i];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <typename F>
static int run(const char* name, F launch, int blocks, int threads, double inst_per_wave, double flop_per_wave, float* out, long long* cyc) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const int nw = blocks * threads / 64;
  std::vector<long long> h(nw);
  CK(hipMemcpy(h.data(), cyc, nw * sizeof(long long), hipMemcpyDeviceToHost));
  double avg = 0;
  for (auto v : h) avg += (double)v;
  avg /= nw;
  const double waves_per_simd = (double)nw / (256.0 * 4.0);
  // per SIMD: waves_per_simd waves each issued inst_per_wave instructions in `avg` clock64 ticks (100 MHz on gfx9: report wall-based too)
  const double inst_per_simd = inst_per_wave * waves_per_simd;
  const double clk = ms * 1e-3 * 2.4e9;   // nominal 2.4 GHz cycles of the whole launch
  printf("%-44s waves/SIMD %.1f  %8.3f ms  %7.3f nominal-clk per wave-instr per SIMD  %8.1f TFLOP/s\n", name, waves_per_simd, ms,
         clk / inst_per_simd, flop_per_wave * nw / (ms * 1e-3) / 1e12);
  return 0;
}

int main() {
  float* out; long long* cyc;
  CK(hipMalloc(&out, 256 * 8 * 512 * sizeof(float)));
  CK(hipMalloc(&cyc, (256 * 8 * 8 + 2) * sizeof(long long)));
  const double n = (double)ITER * 8;
  for (int k : {1, 2, 4, 8}) {
    const int blocks = 256 * k;
    {   // the shader clock under this load: s_memtime (cycles) against s_memrealtime (100 MHz) inside one wave of a long v_fma launch
      for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL(k_valu<0>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0001f);
      CK(hipDeviceSynchronize());
      long long h[2];
      CK(hipMemcpy(h, cyc + 256 * 8 * 8, sizeof h, hipMemcpyDeviceToHost));
      printf("shader clock under v_fma_f32 at %d waves/SIMD: %lld cycles in %.2f us = %.0f MHz; %.2f cycles per wave-instruction of one wave\n", k, h[0], h[1] / 100.0,
             h[0] / (h[1] / 100.0), (double)h[0] / (ITER * 8.0));
    }
    run("v_fma_f32", [&] { hipLaunchKernelGGL(k_valu<0>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0001f); }, blocks, 256, n, n * 64 * 2, out, cyc);
    run("v_pk_fma_f32", [&] { hipLaunchKernelGGL(k_valu<1>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0001f); }, blocks, 256, n, n * 64 * 4, out, cyc);
    run("v_pk_mul_f32 / v_pk_add_f32", [&] { hipLaunchKernelGGL(k_valu<2>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0001f); }, blocks, 256, n, n * 64 * 2, out, cyc);
    run("v_fma_f32 : v_pk_fma_f32 1:1", [&] { hipLaunchKernelGGL(k_valu<3>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0001f); }, blocks, 256, n, n * 64 * 3, out, cyc);
  }
  for (int k : {1, 2, 4, 5, 8}) {
    const int blocks = 256 * k;
    char nm[64];
#define CHAIN(N) snprintf(nm, sizeof nm, "v_fma_f32, %d independent chains", N); \
    run(nm, [&] { hipLaunchKernelGGL(k_chain<N>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f); }, blocks, 256, n, n * 64 * 2, out, cyc);
    CHAIN(1) CHAIN(2) CHAIN(4) CHAIN(8) CHAIN(16)
#undef CHAIN
  }
  // MFMA next to VALU: 512-thread workgroups, 1 per CU (2 waves per SIMD) and 2 per CU (4 waves per SIMD)
  const double nmf32 = (double)(ITER / 16) * 4, nmf16 = (double)(ITER / 16) * 8;
  for (int k : {1, 2}) {
    const int blocks = 256 * k;
    for (int nm : {8, 4, 0}) {
      char nmz[96];
      const double fl32 = nm * nmf32 * 2.0 * 32 * 32 * 2, fl16 = nm * nmf16 * 2.0 * 16 * 16 * 4, flv = (8 - nm) * n * 64 * 2;
      snprintf(nmz, sizeof nmz, "mfma32x32x2 x%d waves + v_fma x%d waves", nm, 8 - nm);
      run(nmz, [&] { hipLaunchKernelGGL((k_mix<0, 0>), dim3(blocks), dim3(512), 0, 0, out, cyc, 1.0001f, nm); }, blocks, 512,
          (nm * nmf32 + (8 - nm) * n) / 8, (fl32 + flv) / 8, out, cyc);
      snprintf(nmz, sizeof nmz, "mfma32x32x2 x%d waves + v_pk_fma x%d waves", nm, 8 - nm);
      run(nmz, [&] { hipLaunchKernelGGL((k_mix<0, 1>), dim3(blocks), dim3(512), 0, 0, out, cyc, 1.0001f, nm); }, blocks, 512,
          (nm * nmf32 + (8 - nm) * n) / 8, (fl32 + 2 * flv) / 8, out, cyc);
      snprintf(nmz, sizeof nmz, "mfma16x16x4 x%d waves + v_fma x%d waves", nm, 8 - nm);
      run(nmz, [&] { hipLaunchKernelGGL((k_mix<1, 0>), dim3(blocks), dim3(512), 0, 0, out, cyc, 1.0001f, nm); }, blocks, 512,
          (nm * nmf16 + (8 - nm) * n) / 8, (fl16 + flv) / 8, out, cyc);
    }
  }
  {
    const size_t nb = (size_t)768 << 20;   // past the 256 MiB Infinity Cache
    uint32_t* src;
    CK(hipMalloc(&src, nb));
    CK(hipMemset(src, 1, nb));
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(k_read<1>, dim3(2048), dim3(256), 0, 0, src, nb / 4, (uint32_t*)out);
      hipLaunchKernelGGL(k_read<3>, dim3(2048), dim3(256), 0, 0, src, nb / 4, (uint32_t*)out);
      hipLaunchKernelGGL(k_read<4>, dim3(2048), dim3(256), 0, 0, src, nb / 4, (uint32_t*)out);
    }
    CK(hipDeviceSynchronize());
    printf("k_read<1|3|4>: %zu bytes per launch (compare with FETCH_SIZE under --pmc)\n", nb);
    CK(hipFree(src));
  }
  {
    const size_t nb = (size_t)1 << 30, n16 = nb / 16;
    u4 *a, *b;
    CK(hipMalloc(&a, nb)); CK(hipMalloc(&b, nb));
    CK(hipMemset(a, 1, nb));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#define TIMEIT(NM, LAUNCH)                                                                        \
    do {                                                                                          \
      for (int r_ = 0; r_ < 2; ++r_) { LAUNCH; }                                                  \
      (void)hipEventRecord(e0, 0);                                                                \
      for (int r_ = 0; r_ < 5; ++r_) { LAUNCH; }                                                  \
      (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);                                 \
      float ms_ = 0; (void)hipEventElapsedTime(&ms_, e0, e1);                                     \
      printf("copy %-44s %8.1f GB/s (read + write)\n", NM, 2.0 * nb * 5 / (ms_ * 1e-3) / 1e9);   \
      fflush(stdout);                                                                             \
    } while (0)
    const int grids[4] = {1024, 2048, 4096, 8192};
    for (int gi = 0; gi < 4; ++gi) {
      const int g = grids[gi];
      char nm[64];
      snprintf(nm, sizeof nm, "A grid-stride nt, %d blocks", g); TIMEIT(nm, hipLaunchKernelGGL(k_copy_a<1>, dim3(g), dim3(256), 0, 0, a, b, n16));
      snprintf(nm, sizeof nm, "A grid-stride plain, %d blocks", g); TIMEIT(nm, hipLaunchKernelGGL(k_copy_a<0>, dim3(g), dim3(256), 0, 0, a, b, n16));
      snprintf(nm, sizeof nm, "B chunk nt, %d blocks", g); TIMEIT(nm, hipLaunchKernelGGL(k_copy_b<1>, dim3(g), dim3(256), 0, 0, a, b, n16));
      snprintf(nm, sizeof nm, "B chunk plain, %d blocks", g); TIMEIT(nm, hipLaunchKernelGGL(k_copy_b<0>, dim3(g), dim3(256), 0, 0, a, b, n16));
    }
    TIMEIT("C one element per thread", hipLaunchKernelGGL(k_copy_c, dim3((unsigned)(n16 / 256)), dim3(256), 0, 0, a, b, n16));
    TIMEIT("hipMemcpyAsync D2D", (void)hipMemcpyAsync(b, a, nb, hipMemcpyDeviceToDevice, 0));
    CK(hipFree(a)); CK(hipFree(b));
  }
  for (int k : {2, 4}) {
    const int blocks = 256 * k;
    run("8 pk_fma + 0 ds_read_b128", [&] { hipLaunchKernelGGL(k_lds<0>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0001f); }, blocks, 256, n, n * 64 * 4, out, cyc);
    run("8 pk_fma + 1 ds_read_b128", [&] { hipLaunchKernelGGL(k_lds<1>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0001f); }, blocks, 256, n, n * 64 * 4, out, cyc);
    run("8 pk_fma + 2 ds_read_b128", [&] { hipLaunchKernelGGL(k_lds<2>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0001f); }, blocks, 256, n, n * 64 * 4, out, cyc);
    run("8 pk_fma + 4 ds_read_b128", [&] { hipLaunchKernelGGL(k_lds<4>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0001f); }, blocks, 256, n, n * 64 * 4, out, cyc);
  }
  return 0;
}
