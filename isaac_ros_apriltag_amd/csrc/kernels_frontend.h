// kernels_frontend.h -- front steps that usually feed the detector: resize and rectify of mono8 frames
// (reference README.md:16-29 recommends resizing 4K input; launch/isaac_ros_apriltag_usb_cam.launch.py:43-63
// puts a RectifyNode in front of the AprilTag node).  HBM-streaming kernels; all sampling arithmetic is
// integer fixed point so that the CPU oracle reproduces the bytes exactly.
#pragma once
#include "common.h"

// dst(x,y) = bilinear sample of src at ((x+0.5)*sw/dw - 0.5, (y+0.5)*sh/dh - 0.5), coordinates in 1/2048
__global__ __launch_bounds__(256) void k_resize_mono8(const uint8_t* __restrict__ src, size_t spitch, int sw, int sh,
                                                      uint8_t* __restrict__ dst, size_t dpitch, int dw, int dh) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= dw || y >= dh) return;
  // fixed-point source position: ((2x+1)*sw*1024/dw - 1024), exact in 64-bit integers
  long long fx = ((long long)(2 * x + 1) * sw * 1024) / dw - 1024;
  long long fy = ((long long)(2 * y + 1) * sh * 1024) / dh - 1024;
  if (fx < 0) fx = 0;
  if (fy < 0) fy = 0;
  int x0 = (int)(fx >> 11), y0 = (int)(fy >> 11);
  int wx = (int)(fx & 2047), wy = (int)(fy & 2047);
  if (x0 >= sw - 1) { x0 = sw - 1; wx = 0; }
  if (y0 >= sh - 1) { y0 = sh - 1; wy = 0; }
  const int x1 = min(x0 + 1, sw - 1), y1 = min(y0 + 1, sh - 1);
  const uint32_t p00 = src[(size_t)y0 * spitch + x0], p01 = src[(size_t)y0 * spitch + x1];
  const uint32_t p10 = src[(size_t)y1 * spitch + x0], p11 = src[(size_t)y1 * spitch + x1];
  const uint32_t top = p00 * (2048 - wx) + p01 * wx, bot = p10 * (2048 - wx) + p11 * wx;
  const uint64_t v = (uint64_t)top * (2048 - wy) + (uint64_t)bot * wy;
  dst[(size_t)y * dpitch + x] = (uint8_t)((v + (1ull << 21)) >> 22);
}

struct RectifyParams {
  double fx, fy, cx, cy;        // source camera K
  double k1, k2, p1, p2, k3;    // plumb_bob
  double nfx, nfy, ncx, ncy;    // destination (pinhole) camera
};

__global__ __launch_bounds__(256) void k_rectify_mono8(const uint8_t* __restrict__ src, size_t spitch, uint8_t* __restrict__ dst,
                                                       size_t dpitch, int w, int h, RectifyParams R) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w || y >= h) return;
  // normalised pinhole ray of the destination pixel, then the plumb_bob model, then source pixels
  const double xn = ((double)x - R.ncx) / R.nfx, yn = ((double)y - R.ncy) / R.nfy;
  const double r2 = xn * xn + yn * yn;
  const double radial = 1.0 + r2 * (R.k1 + r2 * (R.k2 + r2 * R.k3));
  const double xd = xn * radial + (2.0 * R.p1 * xn * yn + R.p2 * (r2 + 2.0 * xn * xn));
  const double yd = yn * radial + (R.p1 * (r2 + 2.0 * yn * yn) + 2.0 * R.p2 * xn * yn);
  const double u = R.fx * xd + R.cx, v = R.fy * yd + R.cy;
  uint8_t out = 0;
  if (u >= 0.0 && v >= 0.0 && u <= (double)(w - 1) && v <= (double)(h - 1)) {
    const int fu = (int)(u * 32.0 + 0.5), fv = (int)(v * 32.0 + 0.5);  // 1/32 pixel
    int x0 = fu >> 5, y0 = fv >> 5, wx = fu & 31, wy = fv & 31;
    if (x0 >= w - 1) { x0 = w - 1; wx = 0; }
    if (y0 >= h - 1) { y0 = h - 1; wy = 0; }
    const int x1 = min(x0 + 1, w - 1), y1 = min(y0 + 1, h - 1);
    const uint32_t p00 = src[(size_t)y0 * spitch + x0], p01 = src[(size_t)y0 * spitch + x1];
    const uint32_t p10 = src[(size_t)y1 * spitch + x0], p11 = src[(size_t)y1 * spitch + x1];
    const uint32_t top = p00 * (32 - wx) + p01 * wx, bot = p10 * (32 - wx) + p11 * wx;
    out = (uint8_t)((top * (32 - wy) + bot * wy + 512) >> 10);
  }
  dst[(size_t)y * dpitch + x] = out;
}
