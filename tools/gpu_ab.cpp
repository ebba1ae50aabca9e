// tools/gpu_ab.cpp -- stand-alone A/B harness for the GPU box (no Python, no torch: starts in a second).  Links against the C ABI of
// visiondepth3d_amd/libvd3d_hip.so and compares / times two launch policies of one kernel on the same inputs; the policies are the
// vd3d_debug_tune knobs, the comparison is byte for byte between the two policies (correctness against the ORACLE stays with pytest -m gpu).
//
//   build (here):  /opt/rocm/bin/hipcc -O2 -std=c++17 tools/gpu_ab.cpp -Iinclude -ldl -o tools/gpu_ab.bin
//   run (GPU box): tools/gpu_ab.bin conv [H W]     body convolution: tune(5, -1) (one tile per workgroup) vs tune(5, skew_us) for several skews
//                  tools/gpu_ab.bin finish [H W]   fused finishing kernel through vd3d_finish_frame: tune(6, 0) vs tune(6, 1) (persistent)
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <string>
#include <vector>

extern "C" {
#include "vd3d.h"
}
#include <dlfcn.h>
// the library is dlopen'ed (VD3D_LIB_PATH, default visiondepth3d_amd/libvd3d_hip.so next to this binary's directory) so that A/B builds of it
// (tools/build_ab.sh: visiondepth3d_amd/ab/libvd3d_hip_<name>.so) run through the same harness
static void* g_lib = nullptr;
template <class F> static F sym(const char* name, bool required = true) {
  void* p = dlsym(g_lib, name);
  if (!p && required) { fprintf(stderr, "missing symbol %s\n", name); exit(4); }
  return reinterpret_cast<F>(p);
}
#define DL(name) static auto name##_ = sym<decltype(&name)>(#name)

#define CK(x)                                                                                          \
  do {                                                                                                 \
    hipError_t e_ = (x);                                                                               \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } \
  } while (0)
#define VD(x)                                                                                         \
  do {                                                                                                \
    int r_ = (x);                                                                                     \
    if (r_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #x, r_, vd3d_last_error_()); exit(3); }            \
  } while (0)

static uint32_t rng_state = 12345u;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }
static float rndf() { return (float)(rnd() & 0xffff) / 65536.0f; }

template <class F> static float time_ms(hipStream_t s, int n, F f) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) f(i);
  CK(hipEventRecord(a, s));
  for (int i = 0; i < n; ++i) f(i);
  CK(hipEventRecord(b, s));
  CK(hipEventSynchronize(b));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, a, b));
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
  return ms / n;
}

#define DL_ALL() DL(vd3d_ctx_create); DL(vd3d_ctx_stream); DL(vd3d_debug_tune); DL(vd3d_conv3x3_c64_f16); DL(vd3d_sync); DL(vd3d_ctx_destroy); \
                 DL(vd3d_render_params_default); DL(vd3d_finish_frame)
static const char* (*vd3d_last_error_)(void) = nullptr;

// phase stamps of the conv kernels (stamps build only): s_memrealtime (100 MHz) of thread 0 of every 16th workgroup
static void print_conv_stamps(const char* label, int tiles) {
  auto get = sym<int (*)(unsigned long long*)>("vd3d_debug_stamps_conv", false);
  if (!get) return;
  static unsigned long long t[64][16];
  if (get(&t[0][0]) != 0) return;
  const char* names[4] = {"tile load -> LDS", "MFMA loop", "bias + PReLU -> staging", "prefetch issue + stores"};
  for (int it = 0; it < tiles; ++it) {
    double acc[4] = {0, 0, 0, 0}; int n = 0; double life = 0;
    for (int i = 0; i < 64; ++i) {
      const unsigned long long* r = t[i] + 8 * it;
      if (!r[0] || !r[4] || r[4] < r[0]) continue;
      for (int k = 0; k < 4; ++k) acc[k] += (double)(r[k + 1] - r[k]) / 100.0;
      life += (double)(r[4] - r[0]) / 100.0; ++n;
    }
    if (!n) continue;
    printf("    [%s] tile %d of a workgroup (%d sampled): %.2f us =", label, it, n, life / n);
    for (int k = 0; k < 4; ++k) printf("  %s %.2f", names[k], acc[k] / n);
    printf("\n");
  }
  // spread of the start times: are the workgroups of the launch in lock-step?
  unsigned long long lo = ~0ull, hi = 0, elo = ~0ull, ehi = 0;
  for (int i = 0; i < 64; ++i) if (t[i][0]) { lo = t[i][0] < lo ? t[i][0] : lo; hi = t[i][0] > hi ? t[i][0] : hi; const unsigned long long e = t[i][8 + 4] ? t[i][12] : t[i][4]; elo = e < elo ? e : elo; ehi = e > ehi ? e : ehi; }
  if (hi) printf("    [%s] first stamps within %.2f us, last stamps within %.2f us, launch span seen by the samples %.2f us\n", label, (hi - lo) / 100.0, (ehi - elo) / 100.0, (ehi - lo) / 100.0);
}

// persistent conv kernel, stamps build: where did each workgroup run, did the two workgroups of a CU get different slot parities, and
// how far apart did they start?
static void print_conv_where(int nwg) {
  auto get = sym<int (*)(unsigned long long*)>("vd3d_debug_where_conv", false);
  if (!get) return;
  static unsigned long long w[1024][4];
  if (get(&w[0][0]) != 0) return;
  std::vector<std::vector<int>> by_cu(4096);
  for (int b = 0; b < nwg && b < 1024; ++b) by_cu[w[b][0] & 4095].push_back(b);
  int ncu = 0, pairs = 0, mixed = 0, other = 0; double gap = 0, gap_same = 0; int nsame = 0;
  for (auto& v : by_cu) {
    if (v.empty()) continue;
    ++ncu;
    if (v.size() == 2) {
      ++pairs;
      const double g = fabs((double)w[v[0]][2] - (double)w[v[1]][2]) / 100.0;
      if (w[v[0]][1] != w[v[1]][1]) { ++mixed; gap += g; } else { gap_same += g; ++nsame; }
    } else ++other;
  }
  unsigned long long t0 = ~0ull, t1 = 0;
  for (int b = 0; b < nwg && b < 1024; ++b) { if (w[b][2] && w[b][2] < t0) t0 = w[b][2]; if (w[b][3] > t1) t1 = w[b][3]; }
  printf("    [where] %d workgroups on %d distinct CU ids; %d CUs with exactly two (%d with different parities, start gap %.2f us; %d same parity, gap %.2f us); %d CUs with another count; span %.2f us\n",
         nwg, ncu, pairs, mixed, mixed ? gap / mixed : 0.0, nsame, nsame ? gap_same / nsame : 0.0, other, (t1 - t0) / 100.0);
}

static int run_conv(int H, int W) {
  DL_ALL();
  vd3d_ctx* c = nullptr;
  VD(vd3d_ctx_create_(0, nullptr, &c));
  hipStream_t s = (hipStream_t)vd3d_ctx_stream_(c);
  const size_t n = (size_t)H * W * 64;
  std::vector<__half> hx(n), hw(36 * 2 * 64 * 8);
  for (auto& v : hx) v = __float2half((rndf() - 0.5f));
  for (auto& v : hw) v = __float2half((rndf() - 0.5f) * 0.08f);
  std::vector<float> hb(64), hs(64);
  for (auto& v : hb) v = (rndf() - 0.5f) * 0.2f;
  for (auto& v : hs) v = rndf() * 0.3f;
  __half *x, *ya, *yb, *w; float *b, *sl;
  CK(hipMalloc(&x, n * 2)); CK(hipMalloc(&ya, n * 2)); CK(hipMalloc(&yb, n * 2)); CK(hipMalloc(&w, hw.size() * 2));
  CK(hipMalloc(&b, 256)); CK(hipMalloc(&sl, 256));
  CK(hipMemcpy(x, hx.data(), n * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(b, hb.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(sl, hs.data(), 256, hipMemcpyHostToDevice));
  const double flops = 2.0 * H * W * 64 * 64 * 9;
  std::vector<uint8_t> ref(n * 2), got(n * 2);
  VD(vd3d_debug_tune_(5, -1));
  CK(hipMemset(ya, 0xee, n * 2));
  VD(vd3d_conv3x3_c64_f16_(c, x, H, W, w, b, sl, ya)); VD(vd3d_sync_(c));
  CK(hipMemcpy(ref.data(), ya, n * 2, hipMemcpyDeviceToHost));
  // timing as the network runs it: layer after layer, ping-pong between two activations (the input of a layer was just written)
  auto chain = [&](int i) { if (i & 1) VD(vd3d_conv3x3_c64_f16_(c, yb, H, W, w, b, sl, ya)); else VD(vd3d_conv3x3_c64_f16_(c, ya, H, W, w, b, sl, yb)); };
  float t0 = time_ms(s, 40, chain);
  printf("conv %dx%d  one tile per workgroup (rounds 2-4): %.1f us  %.0f TFLOP/s  %.3f of fp16 peak\n", W, H, t0 * 1e3, flops / t0 / 1e9, flops / t0 / 1e9 / 2500.0);
  print_conv_stamps("one tile", 1);
  int bad_total = 0;
  const int skews[] = {0, 3, 103};   // >= 100: the 32 x 8 kernel, (mode - 100) workgroups per CU
  for (int sk : skews) {
    VD(vd3d_debug_tune_(5, sk));
    for (int rep = 0; rep < 2; ++rep) {   // twice: the CU arrival counters keep their parity between launches
      CK(hipMemset(yb, 0xee, n * 2));
      VD(vd3d_conv3x3_c64_f16_(c, x, H, W, w, b, sl, yb)); VD(vd3d_sync_(c));
      CK(hipMemcpy(got.data(), yb, n * 2, hipMemcpyDeviceToHost));
      size_t bad = 0, first = 0;
      for (size_t i = 0; i < n * 2; ++i) if (got[i] != ref[i]) { if (!bad) first = i; ++bad; }
      if (bad) { printf("  skew %d rep %d: %zu bytes differ (first at byte %zu = pixel %zu ch %zu)\n", sk, rep, bad, first, first / 128, (first % 128) / 2); ++bad_total; }
    }
    CK(hipMemcpy(ya, x, n * 2, hipMemcpyDeviceToDevice));
    float t = time_ms(s, 40, chain);
    printf("conv %dx%d  persistent, mode %3d: %.1f us  %.0f TFLOP/s  %.3f of fp16 peak  (x%.2f)\n", W, H, sk, t * 1e3, flops / t / 1e9, flops / t / 1e9 / 2500.0, t0 / t);
    if (sk == 0 || sk == 5) print_conv_stamps("persistent", 2);
    if (sk == 5) print_conv_where(512);
  }
  printf("conv %dx%d: persistent == one-tile kernel byte for byte: %s\n", W, H, bad_total ? "NO" : "yes");
  vd3d_ctx_destroy_(c);
  return bad_total ? 1 : 0;
}

// synthetic eyes + normalised depth with structure (gradients, edges, noise) so that every DOF level occurs
static void fill_scene(std::vector<uint8_t>& L, std::vector<uint8_t>& R, std::vector<float>& dn, int H, int W, int eh, int ew) {
  L.resize((size_t)H * W * 3); R.resize((size_t)H * W * 3); dn.resize((size_t)eh * ew);
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x)
      for (int ch = 0; ch < 3; ++ch) {
        const int base = ((x * 7 + y * 3 + ch * 50) >> 2) & 255;
        const size_t i = ((size_t)y * W + x) * 3 + ch;
        L[i] = (uint8_t)((base + (rnd() & 31)) & 255);
        R[i] = (uint8_t)((base + 13 + (rnd() & 31)) & 255);
      }
  for (int y = 0; y < eh; ++y)
    for (int x = 0; x < ew; ++x) {
      float d = 0.5f + 0.45f * sinf(x * 0.004f) * cosf(y * 0.006f) + 0.04f * (rndf() - 0.5f);
      if (((x / 200) + (y / 150)) & 1) d = 1.0f - d;      // depth edges: tiles with several DOF levels
      dn[(size_t)y * ew + x] = d < 0.f ? 0.f : (d > 1.f ? 1.f : d);
    }
}

static int run_finish(int H, int W, const char* fmt_name) {
  DL_ALL();
  vd3d_ctx* c = nullptr;
  VD(vd3d_ctx_create_(0, nullptr, &c));
  hipStream_t s = (hipStream_t)vd3d_ctx_stream_(c);
  vd3d_render_params p;
  vd3d_render_params_default_(&p);
  int format = VD3D_FMT_HALF_SBS;
  if (!strcmp(fmt_name, "full")) format = VD3D_FMT_FULL_SBS;
  if (!strcmp(fmt_name, "interlaced")) format = VD3D_FMT_INTERLACED;
  if (!strcmp(fmt_name, "anaglyph")) format = VD3D_FMT_ANAGLYPH;
  p.format = format;
  p.src_w = W; p.src_h = H; p.warp_w = W; p.warp_h = H;
  int eh = H, ew = W;
  if (format == VD3D_FMT_HALF_SBS) { eh = H / 2; ew = W / 2; p.fit_w = W / 2; p.fit_h = H; p.out_w = W; p.out_h = H; }
  else if (format == VD3D_FMT_FULL_SBS) { p.fit_w = W; p.fit_h = H; p.out_w = 2 * W; p.out_h = H; }
  else { p.fit_w = W; p.fit_h = H; p.out_w = W; p.out_h = H; }
  p.eye_w = ew; p.eye_h = eh;
  p.dof_strength = 2.0; p.sharpness_factor = 0.15; p.color_saturation = 1.05; p.color_contrast = 1.02; p.color_brightness = 0.01;
  p.dof_dense_conv = 1;
  std::vector<uint8_t> hL, hR; std::vector<float> hd;
  fill_scene(hL, hR, hd, H, W, eh, ew);
  uint8_t *L, *R, *o0, *o1; float* dn;
  const size_t no = (size_t)p.out_w * p.out_h * 3;
  CK(hipMalloc(&L, hL.size())); CK(hipMalloc(&R, hR.size())); CK(hipMalloc(&dn, hd.size() * 4)); CK(hipMalloc(&o0, no)); CK(hipMalloc(&o1, no));
  CK(hipMemcpy(L, hL.data(), hL.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(R, hR.data(), hR.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(dn, hd.data(), hd.size() * 4, hipMemcpyHostToDevice));
  std::vector<uint8_t> ref(no), got(no);
  int bad_total = 0;
  float t_ref = 0.f;
  const int modes[] = {0, 1, 2, 3};
  for (int m : modes) {
    if (vd3d_debug_tune_(6, m) != 0) { printf("finish: mode %d not built\n", m); continue; }
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipMemset(o1, 0x5a, no));
      VD(vd3d_finish_frame_(c, L, R, dn, eh, ew, &p, 0.42, rep ? 37 : 0, rep ? 2 : 0, o1)); VD(vd3d_sync_(c));
      CK(hipMemcpy(got.data(), o1, no, hipMemcpyDeviceToHost));
      if (m == 0) { if (rep == 0) ref = got; else memcpy(ref.data(), got.data(), 0); }
      if (m == 0 && rep == 1) { CK(hipMemcpy(o0, o1, no, hipMemcpyDeviceToDevice)); }
      if (m != 0) {
        // reference of this rep from mode 0
        VD(vd3d_debug_tune_(6, 0));
        CK(hipMemset(o0, 0x5a, no));
        VD(vd3d_finish_frame_(c, L, R, dn, eh, ew, &p, 0.42, rep ? 37 : 0, rep ? 2 : 0, o0)); VD(vd3d_sync_(c));
        CK(hipMemcpy(ref.data(), o0, no, hipMemcpyDeviceToHost));
        VD(vd3d_debug_tune_(6, m));
        size_t bad = 0, first = 0;
        for (size_t i = 0; i < no; ++i) if (got[i] != ref[i]) { if (!bad) first = i; ++bad; }
        if (bad) { printf("  finish mode %d rep %d: %zu bytes differ (first at byte %zu: row %zu col %zu)\n", m, rep, bad, first, first / ((size_t)p.out_w * 3), (first % ((size_t)p.out_w * 3)) / 3); ++bad_total; }
      }
    }
    float t = time_ms(s, 30, [&](int) { VD(vd3d_finish_frame_(c, L, R, dn, eh, ew, &p, 0.42, 0, 0, o1)); });
    if (m == 0) t_ref = t;
    printf("finish %s %dx%d mode %d: %.1f us per frame pair (x%.3f vs mode 0)\n", fmt_name, W, H, m, t * 1e3, t_ref / t);
  }
  printf("finish %s %dx%d: every mode == mode 0 byte for byte: %s\n", fmt_name, W, H, bad_total ? "NO" : "yes");
  vd3d_ctx_destroy_(c);
  return bad_total ? 1 : 0;
}

int main(int argc, char** argv) {
  {
    std::string path = getenv("VD3D_LIB_PATH") ? getenv("VD3D_LIB_PATH") : "";
    if (path.empty()) {
      std::string self = argv[0];
      const size_t sl = self.rfind('/');
      path = (sl == std::string::npos ? std::string(".") : self.substr(0, sl)) + "/../visiondepth3d_amd/libvd3d_hip.so";
    }
    g_lib = dlopen(path.c_str(), RTLD_NOW | RTLD_GLOBAL);
    if (!g_lib) { fprintf(stderr, "dlopen %s: %s\n", path.c_str(), dlerror()); return 5; }
    vd3d_last_error_ = sym<const char* (*)(void)>("vd3d_last_error");
    printf("library: %s\n", path.c_str());
  }
  if (argc < 2) { fprintf(stderr, "usage: gpu_ab.bin conv|finish [H W] [format]\n"); return 64; }
  const std::string what = argv[1];
  if (what == "conv") return run_conv(argc > 3 ? atoi(argv[2]) : 540, argc > 3 ? atoi(argv[3]) : 960);
  if (what == "finish") return run_finish(argc > 3 ? atoi(argv[2]) : 2160, argc > 3 ? atoi(argv[3]) : 3840, argc > 4 ? argv[4] : "half");
  fprintf(stderr, "unknown test %s\n", what.c_str());
  return 64;
}
