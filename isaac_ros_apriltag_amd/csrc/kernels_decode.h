// kernels_decode.h -- S6 edge refinement + homography, S7 decode, S8 reconcile/order, S9 planar pose
// (the tail of the closed cuAprilTagsDetect call, reference
// src/apriltag_node.cpp:491-493, whose outputs the node reads at :503-546).
//
// This file holds the scalar building blocks (projection, gray model, bilinear sample, code rotation)
// and the per-frame reconcile/pose kernel; the quad stage itself is the one-wave-per-quad kernel in
// kernels_decode_wave.h.  Every floating-point statement is evaluated in the same order as the
// sequential CPU definition so that ids, corners and poses come out bit-identical.  The code lookup is
// a brute-force popcount scan over the family table (<= 587 codes x 4 rotations), not a hash table.
#pragma once
#include "common.h"

__device__ __forceinline__ void homography_project_dev(const double* H, double x, double y, double* ox, double* oy) {
  const double xx = H[0] * x + H[1] * y + H[2];
  const double yy = H[3] * x + H[4] * y + H[5];
  const double zz = H[6] * x + H[7] * y + H[8];
  *ox = xx / zz;
  *oy = yy / zz;
}

struct GrayModel { double A00, A01, A02, A11, A12, A22, B0, B1, B2, C0, C1, C2; };

__device__ __forceinline__ void graymodel_add_dev(GrayModel& g, double x, double y, double gray) {
  g.A00 += x * x; g.A01 += x * y; g.A02 += x; g.A11 += y * y; g.A12 += y; g.A22 += 1;
  g.B0 += x * gray; g.B1 += y * gray; g.B2 += gray;
}
__device__ __forceinline__ void graymodel_solve_dev(GrayModel& g) {
  const double L0 = __dsqrt_rn(g.A00);
  const double L3 = g.A01 / L0;
  const double L6 = g.A02 / L0;
  const double L4 = __dsqrt_rn(g.A11 - L3 * L3);
  const double L7 = (g.A12 - L3 * L6) / L4;
  const double L8 = __dsqrt_rn(g.A22 - L6 * L6 - L7 * L7);
  const double M0 = 1 / L0;
  const double M3 = -L3 * M0 / L4;
  const double M4 = 1 / L4;
  const double M6 = (-L6 * M0 - L7 * M3) / L8;
  const double M7 = -L7 * M4 / L8;
  const double M8 = 1 / L8;
  const double t0 = M0 * g.B0;
  const double t1 = M3 * g.B0 + M4 * g.B1;
  const double t2 = M6 * g.B0 + M7 * g.B1 + M8 * g.B2;
  g.C0 = M0 * t0 + M3 * t1 + M6 * t2;
  g.C1 = M4 * t1 + M7 * t2;
  g.C2 = M8 * t2;
}
__device__ __forceinline__ double graymodel_interp_dev(const GrayModel& g, double x, double y) {
  return g.C0 * x + g.C1 * y + g.C2;
}

template <typename ImgPtr>
__device__ __forceinline__ double value_for_pixel_dev(ImgPtr im, int w, int h, int pitch, double px, double py) {
  const int x1 = (int)floor(px - 0.5), x2 = (int)ceil(px - 0.5);
  const double x = px - 0.5 - x1;
  const int y1 = (int)floor(py - 0.5), y2 = (int)ceil(py - 0.5);
  const double y = py - 0.5 - y1;
  if (x1 < 0 || x2 >= w || y1 < 0 || y2 >= h) return -1;
  return im[(size_t)y1 * pitch + x1] * (1 - x) * (1 - y) + im[(size_t)y1 * pitch + x2] * x * (1 - y) +
         im[(size_t)y2 * pitch + x1] * (1 - x) * y + im[(size_t)y2 * pitch + x2] * x * y;
}

// ---- S8 reconcile + S9 pose ---------------------------------------------------------------------
__device__ __forceinline__ double orient2d_dev(const double* a, const double* b, const double* c) {
  return (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0]);
}
__device__ __forceinline__ bool on_segment_dev(const double* a, const double* b, const double* c) {
  const double lox = a[0] < b[0] ? a[0] : b[0], hix = a[0] < b[0] ? b[0] : a[0];
  const double loy = a[1] < b[1] ? a[1] : b[1], hiy = a[1] < b[1] ? b[1] : a[1];
  return c[0] >= lox && c[0] <= hix && c[1] >= loy && c[1] <= hiy;
}
__device__ bool segments_intersect_dev(const double* p1, const double* p2, const double* q1, const double* q2) {
  const double d1 = orient2d_dev(q1, q2, p1), d2 = orient2d_dev(q1, q2, p2);
  const double d3 = orient2d_dev(p1, p2, q1), d4 = orient2d_dev(p1, p2, q2);
  if (((d1 > 0 && d2 < 0) || (d1 < 0 && d2 > 0)) && ((d3 > 0 && d4 < 0) || (d3 < 0 && d4 > 0))) return true;
  if (d1 == 0 && on_segment_dev(q1, q2, p1)) return true;
  if (d2 == 0 && on_segment_dev(q1, q2, p2)) return true;
  if (d3 == 0 && on_segment_dev(p1, p2, q1)) return true;
  if (d4 == 0 && on_segment_dev(p1, p2, q2)) return true;
  return false;
}
__device__ bool point_in_quad_dev(const double (*poly)[2], const double* q) {
  bool inside = false;
  for (int i = 0, j = 3; i < 4; j = i++) {
    if (((poly[i][1] > q[1]) != (poly[j][1] > q[1])) &&
        (q[0] < (poly[j][0] - poly[i][0]) * (q[1] - poly[i][1]) / (poly[j][1] - poly[i][1]) + poly[i][0]))
      inside = !inside;
  }
  return inside;
}
__device__ bool quads_overlap_dev(const double (*a)[2], const double (*b)[2]) {
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++)
      if (segments_intersect_dev(a[i], a[(i + 1) & 3], b[j], b[(j + 1) & 3])) return true;
  if (point_in_quad_dev(a, b[0])) return true;
  if (point_in_quad_dev(b, a[0])) return true;
  return false;
}
// strict "a precedes b" in the canonical preference order
__device__ bool det_before_dev(const DetRec* a, const DetRec* b) {
  if (a->family != b->family) return a->family < b->family;
  if (a->id != b->id) return a->id < b->id;
  if (a->hamming != b->hamming) return a->hamming < b->hamming;
  if (a->decision_margin != b->decision_margin) return a->decision_margin > b->decision_margin;
  for (int i = 0; i < 4; i++)
    for (int k = 0; k < 2; k++)
      if (a->p[i][k] != b->p[i][k]) return a->p[i][k] < b->p[i][k];
  return false;
}

__device__ void mat33_inv_transpose_dev(const double* M, double* O) {
  const double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
  const double c10 = M[2] * M[7] - M[1] * M[8], c11 = M[0] * M[8] - M[2] * M[6], c12 = M[1] * M[6] - M[0] * M[7];
  const double c20 = M[1] * M[5] - M[2] * M[4], c21 = M[2] * M[3] - M[0] * M[5], c22 = M[0] * M[4] - M[1] * M[3];
  const double det = M[0] * c00 + M[1] * c01 + M[2] * c02;
  O[0] = c00 / det; O[1] = c01 / det; O[2] = c02 / det;
  O[3] = c10 / det; O[4] = c11 / det; O[5] = c12 / det;
  O[6] = c20 / det; O[7] = c21 / det; O[8] = c22 / det;
}

__device__ void pose_from_homography_dev(const double* H, double fx_in, double fy, double cx, double cy, double skew,
                                         double tag_size, double* R, double* t) {
  const double fx = -fx_in;
  double R20 = H[6], R21 = H[7], TZ = H[8];
  double R10 = (H[3] - cy * R20) / fy, R11 = (H[4] - cy * R21) / fy, TY = (H[5] - cy * TZ) / fy;
  double R00, R01, TX;
  if (skew == 0.0) {
    R00 = (H[0] - cx * R20) / fx; R01 = (H[1] - cx * R21) / fx; TX = (H[2] - cx * TZ) / fx;
  } else {  // first row of K = (fx, skew, cx): remove the skew share of the second camera axis as well
    R00 = (H[0] - cx * R20 - skew * R10) / fx; R01 = (H[1] - cx * R21 - skew * R11) / fx; TX = (H[2] - cx * TZ - skew * TY) / fx;
  }
  const double length1 = (double)at_sqrtf_rn((float)(R00 * R00 + R10 * R10 + R20 * R20));
  const double length2 = (double)at_sqrtf_rn((float)(R01 * R01 + R11 * R11 + R21 * R21));
  double s = 1.0 / (double)at_sqrtf_rn((float)(length1 * length2));
  if (TZ > 0) s *= -1;
  R20 *= s; R21 *= s; TZ *= s; R00 *= s; R01 *= s; TX *= s; R10 *= s; R11 *= s; TY *= s;
  const double R02 = R10 * R21 - R20 * R11, R12 = R20 * R01 - R00 * R21, R22 = R00 * R11 - R10 * R01;
  double X[9] = {R00, R01, R02, R10, R11, R12, R20, R21, R22};
  for (int it = 0; it < 12; it++) {
    double Y[9];
    mat33_inv_transpose_dev(X, Y);
    for (int i = 0; i < 9; i++) X[i] = 0.5 * (X[i] + Y[i]);
  }
  const double scale = tag_size / 2.0;
  TX *= scale; TY *= scale; TZ *= scale;
  R[0] = X[0]; R[1] = X[1]; R[2] = X[2];
  R[3] = -X[3]; R[4] = -X[4]; R[5] = -X[5];
  R[6] = -X[6]; R[7] = -X[7]; R[8] = -X[8];
  t[0] = TX; t[1] = -TY; t[2] = -TZ;
}

// one 64-thread block per frame; lane 0 orders and reconciles (tens of records), all lanes copy
__global__ __launch_bounds__(64) void k_reconcile(const FrameDesc* __restrict__ frames, const DetRec* __restrict__ dets_all,
                                                  FrameCounters* __restrict__ counters,
                                                  uint16_t* __restrict__ order_all, DetRec* __restrict__ host_out,
                                                  uint32_t host_stride, FrameCounters* __restrict__ host_counters, DetParams P) {
  // the kept records and the frame's counters go straight to the pinned host buffers the API call reads -- `host_stride` records
  // per frame -- instead of to device buffers that two copy commands would move afterwards
  const int frame = (int)blockIdx.x + P.frame0;
  uint32_t nd = counters[frame].ndets;
  if (nd > P.dcap) nd = P.dcap;
  const DetRec* dets = dets_all + (size_t)frame * P.dcap;
  uint16_t* order = order_all + (size_t)frame * P.dcap;
  __shared__ uint32_t s_nout;
  // rank sort by the canonical preference order: every lane counts the records that precede its own
  // (ties cannot occur between distinct records except exact duplicates, broken by index)
  for (uint32_t a = threadIdx.x; a < nd; a += 64) {
    uint32_t rank = 0;
    for (uint32_t b = 0; b < nd; b++)
      if (b != a && (det_before_dev(&dets[b], &dets[a]) || (!det_before_dev(&dets[a], &dets[b]) && b < a))) rank++;
    order[rank] = (uint16_t)a;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // keep a detection unless an already-kept one with the same family+id overlaps it; records of one
    // (family, id) are contiguous in the order, so only that run is scanned
    uint32_t nk = 0, run_start = 0;
    for (uint32_t i = 0; i < nd; i++) {
      const DetRec* di = &dets[order[i]];
      if (nk > 0) {
        const DetRec* dl = &dets[order[nk - 1]];
        if (dl->family != di->family || dl->id != di->id) run_start = nk;
      }
      bool dead = false;
      for (uint32_t j = run_start; j < nk && !dead; j++) {
        const DetRec* dj = &dets[order[j]];
        if (dj->family == di->family && dj->id == di->id && quads_overlap_dev(dj->p, di->p)) dead = true;
      }
      if (!dead) order[nk++] = order[i];
    }
    s_nout = nk;
    counters[frame].nout = nk;
  }
  __syncthreads();
  const uint32_t nk = s_nout;
  const FrameDesc fd = frames[frame];
  for (uint32_t i = threadIdx.x; i < nk; i += 64) {
    DetRec d = dets[order[i]];
    pose_from_homography_dev(d.H, fd.fx, fd.fy, fd.cx, fd.cy, fd.skew, P.tag_size, d.R, d.t);
    if (i < host_stride) host_out[(size_t)frame * host_stride + i] = d;
  }
  // The stamp is the LAST thing the host can see of this launch: every lane's records are out (barrier) and ordered ahead of
  // it system-wide (fence), then the counters without the stamp, another fence, and the stamp as one store of its own -- a host
  // that reads the stamp of this launch reads this launch's records and counters.
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    static_assert(offsetof(FrameCounters, seq) == sizeof(FrameCounters) - 4, "the stamp is the struct's last word");
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&counters[frame]);   // (this block wrote the last field, nout, itself)
    uint32_t* dst = reinterpret_cast<uint32_t*>(&host_counters[frame]);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(FrameCounters) / 4) - 1; i++) dst[i] = src[i];   // (the previous launch's stamp stays until the store below)
    __threadfence_system();
    __hip_atomic_store(&host_counters[frame].seq, fd.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    counters[frame].seq = fd.seq;
  }
}
