// vd3d_nv12.hip -- optional NV12 wire format at the frame I/O boundary.
// The reference moves bgr24 both ways (cv2.VideoCapture -> ... -> `ffmpeg -f rawvideo -pix_fmt bgr24`, core/render_3d.py:987,1143-1163):
// 3 bytes per pixel over the pipe / PCIe.  A decoder that hands out NV12 and an encoder that takes NV12 (`-pix_fmt nv12`) move 1.5, and
// the colour conversion ffmpeg would do on the host runs here on the frame that is already in HBM.  There is no reference counterpart to
// be bit-exact with (the reference leaves the conversion to swscale); the arithmetic is the widely used BT.601 limited-range 20-bit
// fixed point of OpenCV's cvtColor (COLOR_YUV2BGR_NV12 / COLOR_BGR2YUV_I420 constants), which is also swscale's default matrix for
// untagged RGB <-> YUV:
//   NV12 -> BGR : y' = max(0, Y - 16) * 1220542;  B = (y' + 2116026 (U-128) + 2^19) >> 20;  G = (y' - 409993 (U-128) - 852492 (V-128) + 2^19) >> 20;
//                 R = (y' + 1673527 (V-128) + 2^19) >> 20, saturated; every 2x2 block shares one (U, V)
//   BGR -> NV12 : Y = (269484 R + 528482 G + 102760 B + (16 << 20) + 2^19) >> 20 per pixel;
//                 U = (-155188 R - 305135 G + 460324 B + (128 << 20) + 2^19) >> 20, V = (460324 R - 385875 G - 74448 B + (128 << 20) + 2^19) >> 20
//                 on the rounded 2x2 mean of the block ((sum + 2) >> 2 per channel)
// One thread per 2x2 block; h and w even.
#include "vd3d_dev.h"
#include "vd3d_kernels.h"

VD_DEV uint8_t nv_sat(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

__global__ __launch_bounds__(256) void k_nv12_to_bgr(const uint8_t* __restrict__ yp, const uint8_t* __restrict__ uvp, int h, int w,
                                                     long long y_pitch, long long uv_pitch, uint8_t* __restrict__ out) {
  const int bx = blockIdx.x * 64 + (threadIdx.x & 63), by = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (2 * bx >= w || 2 * by >= h) return;
  const int u = (int)uvp[(long long)by * uv_pitch + 2 * bx] - 128, v = (int)uvp[(long long)by * uv_pitch + 2 * bx + 1] - 128;
  const int ruv = (1 << 19) + 1673527 * v, guv = (1 << 19) - 852492 * v - 409993 * u, buv = (1 << 19) + 2116026 * u;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int yy = 2 * by + j, xx = 2 * bx + i;
      int y = (int)yp[(long long)yy * y_pitch + xx] - 16;
      y = (y < 0 ? 0 : y) * 1220542;
      uint8_t* o = out + ((size_t)yy * w + xx) * 3;
      o[0] = nv_sat((y + buv) >> 20); o[1] = nv_sat((y + guv) >> 20); o[2] = nv_sat((y + ruv) >> 20);
    }
}

__global__ __launch_bounds__(256) void k_bgr_to_nv12(const uint8_t* __restrict__ bgr, int h, int w, uint8_t* __restrict__ yp,
                                                     uint8_t* __restrict__ uvp, long long y_pitch, long long uv_pitch) {
  const int bx = blockIdx.x * 64 + (threadIdx.x & 63), by = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (2 * bx >= w || 2 * by >= h) return;
  int sb = 0, sg = 0, sr = 0;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int yy = 2 * by + j, xx = 2 * bx + i;
      const uint8_t* p = bgr + ((size_t)yy * w + xx) * 3;
      const int b = p[0], g = p[1], r = p[2];
      sb += b; sg += g; sr += r;
      yp[(long long)yy * y_pitch + xx] = nv_sat((269484 * r + 528482 * g + 102760 * b + (16 << 20) + (1 << 19)) >> 20);
    }
  const int b = (sb + 2) >> 2, g = (sg + 2) >> 2, r = (sr + 2) >> 2;
  uvp[(long long)by * uv_pitch + 2 * bx] = nv_sat((-155188 * r - 305135 * g + 460324 * b + (128 << 20) + (1 << 19)) >> 20);
  uvp[(long long)by * uv_pitch + 2 * bx + 1] = nv_sat((460324 * r - 385875 * g - 74448 * b + (128 << 20) + (1 << 19)) >> 20);
}

void vd_launch_nv12_to_bgr(hipStream_t s, const uint8_t* y, const uint8_t* uv, int h, int w, long long y_pitch, long long uv_pitch, uint8_t* out) {
  hipLaunchKernelGGL(k_nv12_to_bgr, dim3((w / 2 + 63) / 64, (h / 2 + 3) / 4), dim3(256), 0, s, y, uv, h, w, y_pitch, uv_pitch, out);
}
void vd_launch_bgr_to_nv12(hipStream_t s, const uint8_t* bgr, int h, int w, uint8_t* y, uint8_t* uv, long long y_pitch, long long uv_pitch) {
  hipLaunchKernelGGL(k_bgr_to_nv12, dim3((w / 2 + 63) / 64, (h / 2 + 3) / 4), dim3(256), 0, s, bgr, h, w, y, uv, y_pitch, uv_pitch);
}
