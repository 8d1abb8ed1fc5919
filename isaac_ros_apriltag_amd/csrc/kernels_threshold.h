// kernels_threshold.h -- S1 (point-sample decimation) fused into S2 (adaptive 4x4-tile min/max
// threshold with 3x3 tile dilation/erosion).  Stands in for the first stages inside the closed
// cuAprilTagsDetect call (reference src/apriltag_node.cpp:491-493); semantics per SURVEY.md A.1/A.2.
//
// HBM-bound streaming kernel, 2 B/pixel algorithmic (1 read + 1 written):
//   * a 256-thread block owns a 1024 x 32 pixel region (256 x 8 tiles); every thread owns two
//     16 x 4 pixel units = 8 tiles and keeps its 128 pixels in registers (8 x 16-byte loads);
//   * per-tile min/max go to LDS (u8, 10 x 260 incl. a one-tile halo ring), the 3x3 dilation reads
//     two aligned dwords per halo row, the thresholded units leave as 8 x 16-byte stores;
//   * the halo ring (2 tile rows + 2 tile columns) is recomputed from the image (L2 hits).
#pragma once
#include "common.h"

#define TH_LDS_STRIDE 260  // bytes per LDS tile row (65 dwords): 256 tiles + one halo tile each side

// The caller's image pointer comes out of the frame descriptor, so the compiler cannot know its address space and
// would emit flat loads; the pointer is global by contract (include/apriltag_amd.h: device memory).
typedef const __attribute__((address_space(1))) uint8_t* th_gimg_t;
typedef uint32_t th_u32x4 __attribute__((ext_vector_type(4)));   // (HIP's uint4 is a class: no copy out of an address space)
typedef const __attribute__((address_space(1))) th_u32x4* th_gimg4_t;
// ---- colour frames as the reference's default path feeds them (rgb8 / bgr8 of its cuAprilTags branch, src/apriltag_node.cpp:469-486;
// rgba8 / bgra8 of its encoding table, :76-82).  FMT = amdAprilTagsEncoding: 0 mono8, 1 rgb8, 2 bgr8, 3 rgba8, 4 bgra8.  The loader
// of the one-pass kernel reads the interleaved frame itself and forms the gray value on the fly, the kernel writes the gray plane
// once (as it does for a decimated image) and thresholds in the same pass: 3 (4) bytes read + 2 written per pixel, instead of a
// conversion launch (3 + 1) and the mono8 pass (1 + 1) behind it.  Decimation 1 only (the reference's cuAprilTags path has none).
// Gray value: the fixed-point BT.601 weights of cv_bridge / OpenCV, Y = (4899 R + 9617 G + 1868 B + 8192) >> 14 -- the same
// statement as k_to_mono8 (detector.hip) -- evaluated with two 4 x u8 dot products per pixel: 4899 = 19 * 256 + 35,
// 9617 = 37 * 256 + 145, 1868 = 7 * 256 + 76, so Y = ((dot(px, hi) << 8) + dot(px, lo) + 8192) >> 14 exactly; the weight of a
// fourth byte (alpha, or the next pixel's first byte of a 3-byte format) is 0.
template <int FMT> struct th_fmt {
  static constexpr int nch = FMT == 0 ? 1 : (FMT <= 2 ? 3 : 4);
  static constexpr bool red_first = FMT == 1 || FMT == 3;
  static constexpr uint32_t w_hi = red_first ? 0x00072513u : 0x00132507u;   // bytes {19, 37, 7} in memory order
  static constexpr uint32_t w_lo = red_first ? 0x004C9123u : 0x0023914Cu;   // bytes {35, 145, 76}
};
template <int FMT>
__device__ __forceinline__ uint32_t th_gray_of(uint32_t px) {   // px: the pixel's bytes in memory order in bits 0..23
  const uint32_t lo = __builtin_amdgcn_udot4(px, th_fmt<FMT>::w_lo, 8192u, false);
  const uint32_t hi = __builtin_amdgcn_udot4(px, th_fmt<FMT>::w_hi, 0u, false);
  return ((hi << 8) + lo) >> 14;
}

// one working-image pixel through the decimating gather (slow path: halos, edges, unaligned input)
template <int DEC, int FMT = 0>
__device__ __forceinline__ uint32_t th_px(th_gimg_t img, uint32_t pitch, int W0, int H0, int x, int y) {
  int sx = x * DEC, sy = y * DEC;
  if (sx >= W0 || sy >= H0) return 0;
  if (FMT == 0) return img[(size_t)sy * pitch + sx];
  th_gimg_t p = img + (size_t)sy * pitch + (size_t)sx * th_fmt<FMT>::nch;
  return th_gray_of<FMT>((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16));
}

// loads 16 consecutive working pixels of row y starting at x0 (multiple of 16) into 4 dwords
template <int DEC, int FMT = 0>
__device__ __forceinline__ void th_load16(th_gimg_t img, uint32_t pitch, int W0, int H0, bool aligned,
                                          int x0, int y, uint32_t out[4]) {
  int sy = y * DEC;
  if (sy >= H0 || x0 * DEC >= W0) { out[0] = out[1] = out[2] = out[3] = 0; return; }
  th_gimg_t row = img + (size_t)sy * pitch;
  if (FMT != 0) {   // (DEC == 1) 16 pixels = 48 or 64 consecutive bytes: three or four 16-byte loads, then the dot products
    constexpr int NCH = th_fmt<FMT>::nch;
    if (aligned && x0 + 16 <= W0) {
      uint32_t d[4 * NCH + 1];
#pragma unroll
      for (int q = 0; q < NCH; q++) {
        const th_u32x4 v = *(th_gimg4_t)(row + (size_t)x0 * NCH + 16 * q);
        d[4 * q] = v.x; d[4 * q + 1] = v.y; d[4 * q + 2] = v.z; d[4 * q + 3] = v.w;
      }
      d[4 * NCH] = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        uint32_t w = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
          const int k = 4 * j + b;   // pixel k starts at byte NCH * k of the 16-pixel run
          const uint32_t px = NCH == 4 ? d[k] : __builtin_amdgcn_alignbyte(d[(3 * k) / 4 + 1], d[(3 * k) / 4], (3 * k) & 3);
          w |= th_gray_of<FMT>(px) << (8 * b);
        }
        out[j] = w;
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      uint32_t w = 0;
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const int sx = x0 + 4 * j + b;
        uint32_t v = 0;
        if (sx < W0) { th_gimg_t p = row + (size_t)sx * NCH; v = th_gray_of<FMT>((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16)); }
        w |= v << (8 * b);
      }
      out[j] = w;
    }
    return;
  }
  if (DEC == 1) {
    if (aligned && x0 + 16 <= W0) {
      const th_u32x4 v = *(th_gimg4_t)(row + x0);
      out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
      return;
    }
  } else if (DEC == 2) {
    if (aligned && 2 * x0 + 32 <= W0) {
      const th_u32x4 a = *(th_gimg4_t)(row + 2 * x0);
      const th_u32x4 b = *(th_gimg4_t)(row + 2 * x0 + 16);
      // keep the even bytes of each dword pair
      out[0] = (a.x & 0xFF) | ((a.x >> 8) & 0xFF00) | ((a.y & 0xFF) << 16) | ((a.y << 8) & 0xFF000000u);
      out[1] = (a.z & 0xFF) | ((a.z >> 8) & 0xFF00) | ((a.w & 0xFF) << 16) | ((a.w << 8) & 0xFF000000u);
      out[2] = (b.x & 0xFF) | ((b.x >> 8) & 0xFF00) | ((b.y & 0xFF) << 16) | ((b.y << 8) & 0xFF000000u);
      out[3] = (b.z & 0xFF) | ((b.z >> 8) & 0xFF00) | ((b.w & 0xFF) << 16) | ((b.w << 8) & 0xFF000000u);
      return;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    uint32_t w = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) {
      int sx = (x0 + 4 * j + b) * DEC;
      uint32_t v = (sx < W0) ? row[sx] : 0;
      w |= v << (8 * b);
    }
    out[j] = w;
  }
}

__device__ __forceinline__ void th_minmax_word(uint32_t w, uint32_t& mn, uint32_t& mx) {
#pragma unroll
  for (int b = 0; b < 4; b++) {
    uint32_t v = (w >> (8 * b)) & 0xFF;
    mn = min(mn, v);
    mx = max(mx, v);
  }
}

// Launch: 1-D grid of 8*ceil(T/8) blocks, T = gx*gy*frames tiles.  Workgroups are dispatched to the 8
// XCDs round-robin, so block L runs on XCD L%8; the remap below hands every XCD one contiguous run of
// tiles (x fastest, then y, then frame), which keeps the halo rows a block re-reads from its vertical
// neighbours in that XCD's own L2 instead of fetching them again over the fabric.  Placement is a speed
// matter only: any block->tile bijection gives the same result.
//
// Block geometry: 256 threads as 64 (x) by 4 (y); a thread owns two vertically stacked 16x4-pixel units,
// so a block covers 1024 x 32 pixels (256 x 8 tiles) and every wave-wide load is one contiguous 1 KiB
// row segment.
#define TH_BTX 256  // tiles per block in x
#define TH_BTY 8    // tiles per block in y
// (the colour instances take 117 registers -- all loads of both units in flight -- and run at four waves per SIMD against the mono8
// instance's seven; a budget of 96 or 80 registers makes the compiler spill 12 / 28 of them instead of issuing the loads later, and
// a scheduling fence between the two units changes nothing: measured 63 - 66 % of 8 TB/s on 5 N bytes as it stands)
template <int DEC, int FMT = 0>
__global__ __launch_bounds__(256) void k_threshold(const FrameDesc* __restrict__ frames, uint8_t* __restrict__ gray_all,
                                                   uint8_t* __restrict__ thr_all, int gx, int gy, int nframes, DetParams P) {
  static_assert(FMT == 0 || DEC == 1, "the colour loader does not decimate");
  __shared__ __attribute__((aligned(16))) uint8_t smin[10 * TH_LDS_STRIDE];
  __shared__ __attribute__((aligned(16))) uint8_t smax[10 * TH_LDS_STRIDE];

  const int ntiles = gx * gy * nframes;
  const int per_xcd = (int)(gridDim.x >> 3);
  const int tile = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  if (tile >= ntiles) return;
  const int lframe = tile / (gx * gy);
  const int trem = tile - lframe * (gx * gy);
  const int frame = lframe + P.frame0;
  const int bx = trem % gx, by = trem / gx;
  const FrameDesc fd = frames[frame];
  // (a colour submission: the loader reads the caller's interleaved frame, fd.img is the handle's gray plane of pitch WS)
  const th_gimg_t simg = (th_gimg_t)(FMT ? fd.src : fd.img);
  const uint32_t spitch = FMT ? fd.src_pitch : fd.pitch;
  const bool aligned = ((((uintptr_t)simg) | (uintptr_t)spitch) & 15) == 0;
  const int tid = threadIdx.x;
  const int tx64 = tid & 63, ty4 = tid >> 6;
  const int TX0 = bx * TH_BTX, TY0 = by * TH_BTY;  // first tile of the block
  uint8_t* thr = thr_all + (size_t)frame * P.H * P.WS;
  uint8_t* gray = (DEC > 1) ? gray_all + (size_t)frame * P.H * P.WS : (FMT ? const_cast<uint8_t*>(fd.img) : nullptr);

  // ---- own units: 2 x (16 x 4 pixels) ---------------------------------------------------------
  const int ux = (TX0 + 4 * tx64) * 4;
  uint32_t u[2][4][4];
#pragma unroll
  for (int v = 0; v < 2; v++) {
    const int uy = (TY0 + 2 * ty4 + v) * 4;
#pragma unroll
    for (int r = 0; r < 4; r++) th_load16<DEC, FMT>(simg, spitch, P.W0, P.H0, aligned, ux, uy + r, u[v][r]);
  }
#pragma unroll
  for (int v = 0; v < 2; v++) {
    const int lty = 2 * ty4 + v;
    const int tY = TY0 + lty, uy = tY * 4;
    if (DEC > 1 || FMT) {
#pragma unroll
      for (int r = 0; r < 4; r++)
        if (uy + r < P.H && ux < P.WS)
          *reinterpret_cast<uint4*>(gray + (size_t)(uy + r) * P.WS + ux) = make_uint4(u[v][r][0], u[v][r][1], u[v][r][2], u[v][r][3]);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int tX = TX0 + 4 * tx64 + j;
      uint32_t mn = 255, mx = 0;
      if (tX < P.tw && tY < P.th) {
#pragma unroll
        for (int r = 0; r < 4; r++) th_minmax_word(u[v][r][j], mn, mx);
      }
      smin[(lty + 1) * TH_LDS_STRIDE + 4 * tx64 + j + 1] = (uint8_t)mn;
      smax[(lty + 1) * TH_LDS_STRIDE + 4 * tx64 + j + 1] = (uint8_t)mx;
    }
  }
  // ---- halo ring ----------------------------------------------------------------------------
  if (tid < 128) {  // tile rows TY0-1 and TY0+8, 64 units each
    const int side = tid >> 6, c = tid & 63;
    const int tY = side ? TY0 + TH_BTY : TY0 - 1;
    const int lrow = side ? 9 : 0;
    uint32_t h[4][4];
    const bool rowok = tY >= 0 && tY < P.th;
    if (rowok) {
#pragma unroll
      for (int r = 0; r < 4; r++) th_load16<DEC, FMT>(simg, spitch, P.W0, P.H0, aligned, (TX0 + 4 * c) * 4, tY * 4 + r, h[r]);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int tX = TX0 + 4 * c + j;
      uint32_t mn = 255, mx = 0;
      if (rowok && tX < P.tw) {
#pragma unroll
        for (int r = 0; r < 4; r++) th_minmax_word(h[r][j], mn, mx);
      }
      smin[lrow * TH_LDS_STRIDE + 4 * c + j + 1] = (uint8_t)mn;
      smax[lrow * TH_LDS_STRIDE + 4 * c + j + 1] = (uint8_t)mx;
    }
  } else if (tid < 148) {  // tile columns TX0-1 and TX0+256, tile rows TY0-1 .. TY0+8
    const int k = tid - 128;
    const int side = k / 10, lrow = k % 10;
    const int tX = side ? TX0 + TH_BTX : TX0 - 1;
    const int tY = TY0 - 1 + lrow;
    uint32_t mn = 255, mx = 0;
    if (tX >= 0 && tX < P.tw && tY >= 0 && tY < P.th) {
      for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
          uint32_t v = th_px<DEC, FMT>(simg, spitch, P.W0, P.H0, tX * 4 + c, tY * 4 + r);
          mn = min(mn, v);
          mx = max(mx, v);
        }
    }
    const int lcol = side ? TH_BTX + 1 : 0;
    smin[lrow * TH_LDS_STRIDE + lcol] = (uint8_t)mn;
    smax[lrow * TH_LDS_STRIDE + lcol] = (uint8_t)mx;
  }
  __syncthreads();

  // ---- 3x3 dilation / erosion over tiles, then binarize --------------------------------------
  // LDS columns 4*tx64 .. 4*tx64+5 hold tiles (first-1) .. (first+4): two aligned dwords per row.
#pragma unroll
  for (int v = 0; v < 2; v++) {
    const int lty = 2 * ty4 + v;
    const int tY = TY0 + lty, uy = tY * 4;
    if (tY >= P.th) continue;
    uint32_t cmin[6], cmax[6];
#pragma unroll
    for (int c = 0; c < 6; c++) { cmin[c] = 255; cmax[c] = 0; }
#pragma unroll
    for (int dr = 0; dr < 3; dr++) {
      const uint32_t* rmin = reinterpret_cast<const uint32_t*>(smin + (lty + dr) * TH_LDS_STRIDE + 4 * tx64);
      const uint32_t* rmax = reinterpret_cast<const uint32_t*>(smax + (lty + dr) * TH_LDS_STRIDE + 4 * tx64);
      const uint32_t a0 = rmin[0], a1 = rmin[1], b0 = rmax[0], b1 = rmax[1];
#pragma unroll
      for (int c = 0; c < 4; c++) {
        cmin[c] = min(cmin[c], (a0 >> (8 * c)) & 0xFF);
        cmax[c] = max(cmax[c], (b0 >> (8 * c)) & 0xFF);
      }
#pragma unroll
      for (int c = 0; c < 2; c++) {
        cmin[4 + c] = min(cmin[4 + c], (a1 >> (8 * c)) & 0xFF);
        cmax[4 + c] = max(cmax[4 + c], (b1 >> (8 * c)) & 0xFF);
      }
    }
    uint32_t o[4][4];
    bool valid[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int tX = TX0 + 4 * tx64 + j;
      valid[j] = tX < P.tw;
      const uint32_t mn = min(min(cmin[j], cmin[j + 1]), cmin[j + 2]);
      const uint32_t mx = max(max(cmax[j], cmax[j + 1]), cmax[j + 2]);
      if ((int)(mx - mn) < P.min_white_black_diff) {
#pragma unroll
        for (int r = 0; r < 4; r++) o[r][j] = 0x7F7F7F7Fu;
      } else {
        const uint32_t thresh = mn + (mx - mn) / 2;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const uint32_t w = u[v][r][j];
          uint32_t ow = 0;
#pragma unroll
          for (int b = 0; b < 4; b++) ow |= (((w >> (8 * b)) & 0xFF) > thresh ? 0xFFu : 0u) << (8 * b);
          o[r][j] = ow;
        }
      }
    }
    if (valid[3]) {
#pragma unroll
      for (int r = 0; r < 4; r++)
        {
          // streaming output, larger than every cache level at batch sizes that matter and next read by another kernel:
          // a non-temporal store (0.1255 -> 0.1165 ms per 160-frame launch, 5.3 -> 5.7 TB/s; the pipeline's next stage
          // reads from HBM either way)
          th_u32x4 ov; ov.x = o[r][0]; ov.y = o[r][1]; ov.z = o[r][2]; ov.w = o[r][3];
          __builtin_nontemporal_store(ov, reinterpret_cast<th_u32x4*>(thr + (size_t)(uy + r) * P.WS + ux));
        }
    } else {
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (valid[j]) {
#pragma unroll
          for (int r = 0; r < 4; r++) *reinterpret_cast<uint32_t*>(thr + (size_t)(uy + r) * P.WS + ux + 4 * j) = o[r][j];
        }
    }
  }
}

// Pixels right of / below the last full tile (only when W or H is not a multiple of 4): threshold
// against the nearest tile's dilated min/max, recomputed from its 12x12 neighbourhood.  No
// low-contrast rule there.
template <int DEC, int FMT = 0>
__global__ __launch_bounds__(256) void k_threshold_leftover(const FrameDesc* __restrict__ frames, uint8_t* __restrict__ gray_all,
                                                            uint8_t* __restrict__ thr_all, DetParams P) {
  const int frame = (int)blockIdx.z + P.frame0;
  const FrameDesc fd = frames[frame];
  const th_gimg_t simg = (th_gimg_t)(FMT ? fd.src : fd.img);
  const uint32_t spitch = FMT ? fd.src_pitch : fd.pitch;
  const int nright = P.W - P.tw * 4;  // columns per row in the right strip
  const int nbot = P.H - P.th * 4;    // rows in the bottom strip
  const int right_cnt = nright * (P.th * 4);
  const int total = right_cnt + nbot * P.W;
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  int x, y;
  if (i < right_cnt) { y = i / nright; x = P.tw * 4 + i % nright; }
  else { int k = i - right_cnt; y = P.th * 4 + k / P.W; x = k % P.W; }
  int tX = min(x / 4, P.tw - 1), tY = min(y / 4, P.th - 1);
  uint32_t mn = 255, mx = 0;
  for (int ty = max(tY - 1, 0); ty <= min(tY + 1, P.th - 1); ty++)
    for (int tx = max(tX - 1, 0); tx <= min(tX + 1, P.tw - 1); tx++)
      for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
          uint32_t v = th_px<DEC, FMT>(simg, spitch, P.W0, P.H0, tx * 4 + c, ty * 4 + r);
          mn = min(mn, v);
          mx = max(mx, v);
        }
  uint32_t thresh = mn + (mx - mn) / 2;
  uint32_t v = th_px<DEC, FMT>(simg, spitch, P.W0, P.H0, x, y);
  thr_all[(size_t)frame * P.H * P.WS + (size_t)y * P.WS + x] = v > thresh ? 255 : 0;
  if (DEC > 1) gray_all[(size_t)frame * P.H * P.WS + (size_t)y * P.WS + x] = (uint8_t)v;
  // (FMT: the one-pass kernel's units cover these pixels too and have written their gray values)
}

// ---- tile sizes other than 4 (the reference's settable `tile_size`, src/apriltag_node.cpp:566, handed to the library at
// :451).  4 is the reference's default and the path every measurement is quoted on: it keeps the one-pass kernel above.  Any other
// accepted size takes the plain statement of SURVEY.md A.2 in two passes -- per-tile min / max to two small global arrays
// (tw x th bytes per frame), then per pixel the 3 x 3 tile neighbourhood, the low-contrast rule inside the full tiles and the
// nearest tile's threshold right of / below them.  Same bytes as the oracle's ato_threshold(.., tile, ..); not a tuned kernel.
template <int DEC>
__global__ __launch_bounds__(256) void k_tile_minmax(const FrameDesc* __restrict__ frames, uint8_t* __restrict__ tmin_all,
                                                     uint8_t* __restrict__ tmax_all, int ts, DetParams P) {
  const int frame = (int)blockIdx.z + P.frame0;
  const FrameDesc fd = frames[frame];
  const int t = (int)(blockIdx.x * 256 + threadIdx.x);
  if (t >= P.tw * P.th) return;
  const int tx = t % P.tw, ty = t / P.tw;
  uint32_t mn = 255, mx = 0;
  for (int r = 0; r < ts; r++)
    for (int c = 0; c < ts; c++) {
      const uint32_t v = th_px<DEC>((th_gimg_t)fd.img, fd.pitch, P.W0, P.H0, tx * ts + c, ty * ts + r);
      mn = min(mn, v);
      mx = max(mx, v);
    }
  tmin_all[(size_t)frame * P.tw * P.th + t] = (uint8_t)mn;
  tmax_all[(size_t)frame * P.tw * P.th + t] = (uint8_t)mx;
}

template <int DEC>
__global__ __launch_bounds__(256) void k_threshold_any_tile(const FrameDesc* __restrict__ frames, uint8_t* __restrict__ gray_all,
                                                            uint8_t* __restrict__ thr_all, const uint8_t* __restrict__ tmin_all,
                                                            const uint8_t* __restrict__ tmax_all, int ts, DetParams P) {
  const int frame = (int)blockIdx.z + P.frame0;
  const FrameDesc fd = frames[frame];
  const int x = (int)(blockIdx.x * 256 + threadIdx.x), y = (int)blockIdx.y;
  if (x >= P.W) return;
  const uint8_t* tmin = tmin_all + (size_t)frame * P.tw * P.th;
  const uint8_t* tmax = tmax_all + (size_t)frame * P.tw * P.th;
  const int tX = min(x / ts, P.tw - 1), tY = min(y / ts, P.th - 1);
  uint32_t mn = 255, mx = 0;
  for (int ty = max(tY - 1, 0); ty <= min(tY + 1, P.th - 1); ty++)
    for (int tx = max(tX - 1, 0); tx <= min(tX + 1, P.tw - 1); tx++) {
      mn = min(mn, (uint32_t)tmin[ty * P.tw + tx]);
      mx = max(mx, (uint32_t)tmax[ty * P.tw + tx]);
    }
  const uint32_t v = th_px<DEC>((th_gimg_t)fd.img, fd.pitch, P.W0, P.H0, x, y);
  const bool in_full_tile = x < P.tw * ts && y < P.th * ts;   // (no low-contrast rule right of / below the last full tile)
  uint8_t o;
  if (in_full_tile && (int)(mx - mn) < P.min_white_black_diff) o = 127;
  else o = v > mn + (mx - mn) / 2 ? 255 : 0;
  thr_all[(size_t)frame * P.H * P.WS + (size_t)y * P.WS + x] = o;
  if (DEC > 1) gray_all[(size_t)frame * P.H * P.WS + (size_t)y * P.WS + x] = (uint8_t)v;
}
