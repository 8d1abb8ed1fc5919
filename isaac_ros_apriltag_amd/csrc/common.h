// common.h -- shared device-side types and helpers of the MI355X AprilTag detector.
// gfx950 only: wave = 64 lanes, __ballot is 64-bit.  Compiled with -ffp-contract=off; every
// floating-point statement below is meant to be exactly one IEEE operation per operator.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tools_hooks.h"

#define AT_NO_LABEL 0xFFFFFFFFu
#define AT_LABEL_BIG 0x80000000u    // on a root's own label entry: the component has at least min_component_size pixels
#define AT_LABEL_MASK 0x7FFFFFFFu
#define AT_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull
#define AT_INVALID_SLOT 0xFFFFFFFFu
#define AT_MAX_FAMILIES 4
#define AT_FLAG_CANDS 0x20u   // internal frame flag: quad-candidate list full (never reported; see run_batch)

// Per-frame descriptor (device copy of amdAprilTagsImageInput_t + intrinsics), one per batch slot.
struct FrameDesc {
  const uint8_t* img;  // mono8, full resolution -- of a colour submission: the handle's gray plane, which the threshold pass (or the
                       // conversion launch ahead of it) fills from `src`; every later stage reads `img` and never sees the colour frame
  uint32_t pitch;
  uint32_t seq;        // number of the launch this descriptor was written for (the handle's counter): comes back in FrameCounters::seq
  double fx, fy, cx, cy;
  double skew;
  const uint8_t* src;  // colour submissions (amdAprilTagsEncoding != mono8): the caller's interleaved frame; null otherwise
  uint32_t src_pitch;
  uint32_t fmt;        // amdAprilTagsEncoding of `src`
};

// Per-frame counters (one struct per batch slot), zeroed at the start of every submission.
struct FrameCounters {
  uint32_t npoints_raw;   // staged boundary points
  uint32_t nclusters;     // kept clusters
  uint32_t npoints_kept;  // points in kept clusters (allocation cursor)
  uint32_t nquads;
  uint32_t ndets;         // raw detections before reconcile
  uint32_t flags;         // AMDAT_FLAG_*
  uint32_t nout;          // detections after reconcile
  uint32_t nroots;        // tile-local component roots (CC root list)
  uint32_t ncand;         // quad candidates (four fitted lines) awaiting k_quad_finish
  uint32_t nlong;         // long staging records (k_points -> k_scatter)
  uint32_t seq;           // FrameDesc::seq of the launch that produced these counters, written last (k_reconcile): the host checks it
                          // after its stream wait, so results of an EARLIER launch can never be taken for this one's
};

struct ClusterRec {
  uint64_t key;
  uint32_t start;
  uint32_t count;
};

struct QuadRec {
  float p[4][2];
  int32_t reversed_border;
  uint32_t pad;
  uint64_t key;
};

// A corner choice that passed the line-fit tests: the four lines {Ex, Ey, nx, ny} of the segments q0->q1, q1->q2, q2->q3,
// q3->q0, handed from k_fit_quads to k_quad_finish.
struct FitCand {
  double line[4][4];
  double wm[2];               // wrap_is_moments: line[3] = {Mx, My, Mxx, Mxy} and wm = {Myy, W} of the segment q3->q0
  uint64_t key;
  int32_t reversed_border;
  uint32_t wrap_is_moments;
};

// Same layout as amdAprilTagsDetectionEx_t.
struct DetRec {
  int32_t family, id, hamming;
  float decision_margin;
  double H[9];
  double c[2];
  double p[4][2];
  double R[9];
  double t[3];
};

struct FamilyDev {
  uint32_t nbits, d, width_at_border, total_width;
  int32_t reversed_border;
  uint32_t ncodes;
  const uint64_t* codes;  // device pointer
  // AprilTag-3 style layout: cell of data bit i in border coordinates, and the bit that the 90-degree pattern rotation
  // moves onto bit i (new(x, y) = old(width_at_border - 1 - y, x)); classic families: (1 + i % d, 1 + i / d)
  int8_t bit_x[64], bit_y[64];
  uint8_t rot_src[64];
};

// Geometry + algorithm parameters shared by all kernels of a handle.
struct DetParams {
  int frame0;        // first batch slot of this launch (a submission may be split into concurrent halves)
  int W0, H0;        // input size
  int W, H;          // working (decimated) size
  int WS;            // pitch of the working u8 images (multiple of 16)
  int decimate;
  int tw, th;        // full tiles of the threshold (tile x tile pixels)
  int tile;          // threshold tile edge: 4 (the one-pass kernel) or 8 (the two-pass statement)
  int min_white_black_diff;
  int min_component_size;
  int min_cluster_points;
  int max_cluster_points;
  int max_nmaxima;
  int min_tag_width;
  int normal_border, reversed_border;
  int refine_edges;
  int max_hamming;
  int nfam;
  int split_moments;  // moment sums carried as two doubles (exact while terms < 2^31 and clusters < 2^15 points)
  double cos_critical_rad;
  double max_line_fit_mse;
  double decode_sharpening;
  double tag_size;
  // capacities per frame
  uint32_t pcap, hcap, hshift, ccap, qcap, dcap;
  uint32_t rcap;     // tile-local roots per frame (CC root list)
  uint32_t cand_cap; // quad candidates per frame (k_fit_quads -> k_quad_finish); grows on demand up to ccap
  uint32_t lcap;     // long staging records per frame (emissions without a block-table entry, kernels_cluster.h): pcap / 8
  uint32_t wshift;   // a work item of the quad fit is (frame << wshift) | cluster index: 32 - bits of the handle's frame count, at most 24
  FamilyDev fam[AT_MAX_FAMILIES];
};

__device__ __forceinline__ uint32_t pack_point(int x, int y, int gx, int gy) {
  return ((uint32_t)x << 18) | ((uint32_t)y << 4) | ((uint32_t)(gx / 255 + 1) << 2) | (uint32_t)(gy / 255 + 1);
}

__device__ __forceinline__ uint32_t float_sortable(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// Correctly rounded float sqrt.  HIP's __fsqrt_rn lowers to the native (approximate) instruction on
// gfx950 (measured: not bit-exact against IEEE), while plain sqrtf under this build's flags (no fast
// math; correctly-rounded f32 divide/sqrt is the HIP default) expands to v_sqrt_f32 plus the fix-up steps
// and is exact (amdAprilTagsDebugMath op 2 checks it on the device against IEEE on 200 000 operands).
__device__ __forceinline__ float at_sqrtf_rn(float x) { return sqrtf(x); }

// Correctly rounded double square root of an integer below 2^18 (the squared gradient magnitudes of the line-fit
// weights, at most 2 * 255^2): v_rsq_f32 seed (about 23 bits), one coupled Goldschmidt step (46 bits) and one
// residual correction, which lands within 2^-88 of the root before the final rounding.  Less than half the
// instructions of the generic double-precision expansion (no v_rsq_f64, no range scaling).  Equality with IEEE sqrt
// holds for every one of the 2^18 arguments even when the seed is off by +-8 ulp (checked exhaustively on the host),
// and tests/test_gpu_parity.py checks all of them on the device (amdAprilTagsDebugMath op 5).
__device__ __forceinline__ double sqrt_u18(uint32_t G) {
  const float gf = (float)G;
  const float r0 = __builtin_amdgcn_rsqf(fmaxf(gf, 1e-30f));   // G = 0: s stays 0 through every step
  const double g = (double)G, r = (double)r0;
  double s = g * r, h = 0.5 * r;
  const double e = __fma_rn(-h, s, 0.5);
  s = __fma_rn(s, e, s);
  h = __fma_rn(h, e, h);
  const double d = __fma_rn(-s, s, g);
  return __fma_rn(d, h, s);
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// Block -> (frame, block of the frame) for kernels whose blocks of ONE frame meet on a few hot addresses (the roots of a
// giant component, the pair-table slot of the two big components of a textured background): a 1-D grid of bpf * n blocks in
// which the frames of a group of G take turns, so that blocks running at the same time belong to G different frames.  (G = 1
// is the plain frame-major order; G bounds the number of per-frame tables the blocks in flight touch.)
__device__ __forceinline__ void at_frame_block(uint32_t lin, uint32_t bpf, uint32_t n, uint32_t G, uint32_t* frame, uint32_t* blk) {
  if (G <= 1) { *frame = lin / bpf; *blk = lin - *frame * bpf; return; }
  const uint32_t ngroups = (n + G - 1) / G;
  uint32_t group = lin / (bpf * G);
  if (group >= ngroups) group = ngroups - 1;
  const uint32_t rem = lin - group * bpf * G;
  const uint32_t gsize = (group == ngroups - 1) ? n - group * G : G;
  *blk = rem / gsize;
  *frame = group * G + (rem - *blk * gsize);
}

// wave-level inclusive scan (wave64)
// (on the DPP network -- row_shr 1/2/4/8 inside the rows of 16 lanes, then the row broadcasts 15 and 31; lanes without a
// source add 0 -- not with __shfl_up: that is six ds_bpermute_b32 through the LDS crossbar per scan)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, true);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, true);
  return v;
}

// Exclusive scan of one uint32 per thread over a 256-thread block; returns exclusive prefix and the
// block total through *total.  scratch: 4 uint32 in LDS.
__device__ __forceinline__ uint32_t block_excl_scan256(uint32_t v, uint32_t* scratch, uint32_t* total) {
  uint32_t inc = wave_incl_scan(v);
  int w = threadIdx.x >> 6;
  if (lane_id() == 63) scratch[w] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    uint32_t s = scratch[i];
    if (i < w) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}
