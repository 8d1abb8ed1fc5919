// kernels_decode.h -- S6 edge refinement + homography, S7 decode, S8 reconcile/order, S9 planar pose
// (the tail of the closed cuAprilTagsDetect call, reference
// src/apriltag_node.cpp:491-493, whose outputs the node reads at :503-546).
//
// This file holds the scalar building blocks (projection, gray model, bilinear sample, code rotation),
// a one-lane-per-quad kernel k_decode kept as the simple form of the stage, and the per-frame
// reconcile/pose kernel.  The production launch is the one-wave-per-quad kernel in
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

__device__ void refine_edges_dev(const DetParams& P, const uint8_t* im, int w, int h, int pitch, QuadRec* quad) {
  double lines[4][4];
  for (int edge = 0; edge < 4; edge++) {
    const int a = edge, b = (edge + 1) & 3;
    double nx = (double)quad->p[b][1] - (double)quad->p[a][1];
    double ny = -(double)quad->p[b][0] + (double)quad->p[a][0];
    const double mag = __dsqrt_rn(nx * nx + ny * ny);
    nx /= mag; ny /= mag;
    if (quad->reversed_border) { nx = -nx; ny = -ny; }
    int nsamples = (int)(mag / 8);
    if (nsamples < 16) nsamples = 16;
    double Mx = 0, My = 0, Mxx = 0, Mxy = 0, Myy = 0, N = 0;
    const double range = P.decimate + 1;
    const int steps = (int)(2 * range * 4) + 1;
    for (int s = 0; s < nsamples; s++) {
      const double alpha = (1.0 + s) / (nsamples + 1);
      const double x0 = alpha * (double)quad->p[a][0] + (1 - alpha) * (double)quad->p[b][0];
      const double y0 = alpha * (double)quad->p[a][1] + (1 - alpha) * (double)quad->p[b][1];
      double Mn = 0, Mcount = 0;
      for (int k = 0; k < steps; k++) {
        const double n = -range + 0.25 * k;
        const double grange = 1;
        const int x1 = (int)(x0 + (n + grange) * nx);
        const int y1 = (int)(y0 + (n + grange) * ny);
        if (x1 < 0 || x1 >= w || y1 < 0 || y1 >= h) continue;
        const int x2 = (int)(x0 + (n - grange) * nx);
        const int y2 = (int)(y0 + (n - grange) * ny);
        if (x2 < 0 || x2 >= w || y2 < 0 || y2 >= h) continue;
        const int g1 = im[(size_t)y1 * pitch + x1];
        const int g2 = im[(size_t)y2 * pitch + x2];
        if (g1 < g2) continue;
        const double weight = (double)((g2 - g1) * (g2 - g1));
        Mn += weight * n;
        Mcount += weight;
      }
      if (Mcount == 0) continue;
      const double n0 = Mn / Mcount;
      const double bestx = x0 + n0 * nx, besty = y0 + n0 * ny;
      Mx += bestx; My += besty; Mxx += bestx * bestx; Mxy += bestx * besty; Myy += besty * besty; N++;
    }
    const double Ex = Mx / N, Ey = My / N;
    const double Cxx = Mxx / N - Ex * Ex, Cxy = Mxy / N - Ex * Ey, Cyy = Myy / N - Ey * Ey;
    const double disc = (Cxx - Cyy) * (Cxx - Cyy) + 4 * Cxy * Cxy;
    const double eig = 0.5 * (Cxx + Cyy + (double)at_sqrtf_rn((float)disc));
    const double nx1 = Cxx - eig, ny1 = Cxy, M1 = nx1 * nx1 + ny1 * ny1;
    const double nx2 = Cxy, ny2 = Cyy - eig, M2 = nx2 * nx2 + ny2 * ny2;
    double M;
    if (M1 > M2) { nx = nx1; ny = ny1; M = M1; } else { nx = nx2; ny = ny2; M = M2; }
    const double length = (double)at_sqrtf_rn((float)M);
    if (fabs(length) < 1e-12) { nx = 0; ny = 0; } else { nx = nx / length; ny = ny / length; }
    lines[edge][0] = Ex; lines[edge][1] = Ey; lines[edge][2] = nx; lines[edge][3] = ny;
  }
  for (int i = 0; i < 4; i++) {
    const double A00 = lines[i][3], A01 = -lines[(i + 1) & 3][3];
    const double A10 = -lines[i][2], A11 = lines[(i + 1) & 3][2];
    const double B0 = -lines[i][0] + lines[(i + 1) & 3][0];
    const double B1 = -lines[i][1] + lines[(i + 1) & 3][1];
    const double det = A00 * A11 - A10 * A01;
    if (fabs(det) > 0.001) {
      const double W00 = A11 / det, W01 = -A01 / det;
      const double L0 = W00 * B0 + W01 * B1;
      quad->p[i][0] = (float)(lines[i][0] + L0 * A00);
      quad->p[i][1] = (float)(lines[i][1] + L0 * A10);
    }
  }
}

__device__ int homography_compute_dev(const QuadRec* q, double* H) {
  double A[72];
  for (int i = 0; i < 4; i++) {
    const double x = (i == 0 || i == 3) ? -1 : 1, y = (i == 0 || i == 1) ? -1 : 1;
    const double u = (double)q->p[i][0], v = (double)q->p[i][1];
    double* r0 = &A[(2 * i) * 9];
    double* r1 = &A[(2 * i + 1) * 9];
    r0[0] = x; r0[1] = y; r0[2] = 1; r0[3] = 0; r0[4] = 0; r0[5] = 0; r0[6] = -x * u; r0[7] = -y * u; r0[8] = u;
    r1[0] = 0; r1[1] = 0; r1[2] = 0; r1[3] = x; r1[4] = y; r1[5] = 1; r1[6] = -x * v; r1[7] = -y * v; r1[8] = v;
  }
  for (int col = 0; col < 8; col++) {
    double max_val = 0;
    int max_idx = -1;
    for (int row = col; row < 8; row++) {
      const double val = fabs(A[row * 9 + col]);
      if (val > max_val) { max_val = val; max_idx = row; }
    }
    if (max_val < 1e-10) return -1;
    if (max_idx != col)
      for (int i = col; i < 9; i++) { const double t = A[col * 9 + i]; A[col * 9 + i] = A[max_idx * 9 + i]; A[max_idx * 9 + i] = t; }
    for (int i = col + 1; i < 8; i++) {
      const double f = A[i * 9 + col] / A[col * 9 + col];
      A[i * 9 + col] = 0;
      for (int j = col + 1; j < 9; j++) A[i * 9 + j] -= f * A[col * 9 + j];
    }
  }
  for (int col = 7; col >= 0; col--) {
    double sum = 0;
    for (int i = col + 1; i < 8; i++) sum += A[col * 9 + i] * A[i * 9 + 8];
    A[col * 9 + 8] = (A[col * 9 + 8] - sum) / A[col * 9 + col];
  }
  for (int i = 0; i < 8; i++) H[i] = A[i * 9 + 8];
  H[8] = 1;
  return 0;
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

__device__ __forceinline__ double value_for_pixel_dev(const uint8_t* im, int w, int h, int pitch, double px, double py) {
  const int x1 = (int)floor(px - 0.5), x2 = (int)ceil(px - 0.5);
  const double x = px - 0.5 - x1;
  const int y1 = (int)floor(py - 0.5), y2 = (int)ceil(py - 0.5);
  const double y = py - 0.5 - y1;
  if (x1 < 0 || x2 >= w || y1 < 0 || y2 >= h) return -1;
  return im[(size_t)y1 * pitch + x1] * (1 - x) * (1 - y) + im[(size_t)y1 * pitch + x2] * x * (1 - y) +
         im[(size_t)y2 * pitch + x1] * (1 - x) * y + im[(size_t)y2 * pitch + x2] * x * y;
}

__device__ __forceinline__ uint64_t rotate90_dev(uint64_t w, int d) {
  uint64_t o = 0;
  const int nb = d * d;
  for (int r = 0; r < d; r++)
    for (int c = 0; c < d; c++) {
      const int sr = c, sc = d - 1 - r;
      if ((w >> (nb - 1 - (sr * d + sc))) & 1) o |= 1ull << (nb - 1 - (r * d + c));
    }
  return o;
}

__device__ float quad_decode_dev(const DetParams& P, const FamilyDev& fam, const uint8_t* im, int w, int h, int pitch,
                                 const double* H, int* id, int* hamming, int* rotation, int* found) {
  const int wb = (int)fam.width_at_border, tw = (int)fam.total_width;
  GrayModel whitemodel = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, blackmodel = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  *found = 0;
  for (int pi = 0; pi < 8; pi++) {
    // {x0, y0, dx, dy, is_white} of the 8 border sample lines
    float p0, p1, p2, p3;
    int is_white;
    switch (pi) {
      case 0: p0 = -0.5f; p1 = 0.5f; p2 = 0; p3 = 1; is_white = 1; break;
      case 1: p0 = 0.5f; p1 = 0.5f; p2 = 0; p3 = 1; is_white = 0; break;
      case 2: p0 = (float)wb + 0.5f; p1 = 0.5f; p2 = 0; p3 = 1; is_white = 1; break;
      case 3: p0 = (float)wb - 0.5f; p1 = 0.5f; p2 = 0; p3 = 1; is_white = 0; break;
      case 4: p0 = 0.5f; p1 = -0.5f; p2 = 1; p3 = 0; is_white = 1; break;
      case 5: p0 = 0.5f; p1 = 0.5f; p2 = 1; p3 = 0; is_white = 0; break;
      case 6: p0 = 0.5f; p1 = (float)wb + 0.5f; p2 = 1; p3 = 0; is_white = 1; break;
      default: p0 = 0.5f; p1 = (float)wb - 0.5f; p2 = 1; p3 = 0; is_white = 0; break;
    }
    for (int i = 0; i < wb; i++) {
      const double tagx01 = ((double)p0 + i * (double)p2) / wb;
      const double tagy01 = ((double)p1 + i * (double)p3) / wb;
      const double tagx = 2 * (tagx01 - 0.5), tagy = 2 * (tagy01 - 0.5);
      double px, py;
      homography_project_dev(H, tagx, tagy, &px, &py);
      const int ix = (int)px, iy = (int)py;
      if (ix < 0 || iy < 0 || ix >= w || iy >= h) continue;
      const int v = im[(size_t)iy * pitch + ix];
      if (is_white) graymodel_add_dev(whitemodel, tagx, tagy, v);
      else graymodel_add_dev(blackmodel, tagx, tagy, v);
    }
  }
  graymodel_solve_dev(whitemodel);
  graymodel_solve_dev(blackmodel);
  if ((graymodel_interp_dev(whitemodel, 0, 0) - graymodel_interp_dev(blackmodel, 0, 0) < 0) != (fam.reversed_border != 0))
    return -1;

  double values[12 * 12];
  for (int i = 0; i < tw * tw; i++) values[i] = 0;
  const int min_coord = (wb - tw) / 2;
  const int d = (int)fam.d;
  for (int i = 0; i < (int)fam.nbits; i++) {
    const int bitx = 1 + i % d, bity = 1 + i / d;
    const double tagx01 = (bitx + 0.5) / wb, tagy01 = (bity + 0.5) / wb;
    const double tagx = 2 * (tagx01 - 0.5), tagy = 2 * (tagy01 - 0.5);
    double px, py;
    homography_project_dev(H, tagx, tagy, &px, &py);
    const double v = value_for_pixel_dev(im, w, h, pitch, px, py);
    if (v == -1) continue;
    const double thresh = (graymodel_interp_dev(blackmodel, tagx, tagy) + graymodel_interp_dev(whitemodel, tagx, tagy)) / 2.0;
    values[tw * (bity - min_coord) + bitx - min_coord] = v - thresh;
  }
  float black_score = 0, white_score = 0, black_count = 1, white_count = 1;
  uint64_t rcode = 0;
  for (int i = 0; i < (int)fam.nbits; i++) {
    const int bitx = 1 + i % d, bity = 1 + i / d;
    const int y = bity - min_coord, x = bitx - min_coord;
    // sharpened value at (x,y): v + s * (4v - up - left - right - down), accumulated in kernel order
    double s = 0;
    if (y - 1 >= 0) s += values[(y - 1) * tw + x] * -1.0;
    if (x - 1 >= 0) s += values[y * tw + x - 1] * -1.0;
    s += values[y * tw + x] * 4.0;
    if (x + 1 <= tw - 1) s += values[y * tw + x + 1] * -1.0;
    if (y + 1 <= tw - 1) s += values[(y + 1) * tw + x] * -1.0;
    const double v = values[y * tw + x] + P.decode_sharpening * s;
    rcode <<= 1;
    if (v > 0) { white_score = (float)((double)white_score + v); white_count++; rcode |= 1; }
    else { black_score = (float)((double)black_score - v); black_count++; }
  }
  for (int r = 0; r < 4 && !*found; r++) {
    int best = 1 << 30, bid = -1;
    for (uint32_t i = 0; i < fam.ncodes; i++) {
      const int hd = __popcll(rcode ^ fam.codes[i]);
      if (hd < best) { best = hd; bid = (int)i; }
    }
    if (best <= P.max_hamming) { *id = bid; *hamming = best; *rotation = r; *found = 1; }
    else rcode = rotate90_dev(rcode, d);
  }
  const float a = white_score / white_count, b = black_score / black_count;
  return a < b ? a : b;
}

// one thread per quad
__global__ __launch_bounds__(64) void k_decode(const FrameDesc* __restrict__ frames, const QuadRec* __restrict__ quads_all,
                                               DetRec* __restrict__ dets_all, FrameCounters* __restrict__ counters, DetParams P) {
  const int frame = (int)blockIdx.y + P.frame0;
  uint32_t nq = counters[frame].nquads;
  if (nq > P.qcap) nq = P.qcap;
  const FrameDesc fd = frames[frame];
  for (uint32_t qi = blockIdx.x * 64 + threadIdx.x; qi < nq; qi += gridDim.x * 64) {
    QuadRec q = quads_all[(size_t)frame * P.qcap + qi];
    if (P.refine_edges) refine_edges_dev(P, fd.img, P.W0, P.H0, (int)fd.pitch, &q);
    double H[9];
    if (homography_compute_dev(&q, H) != 0) continue;
    for (int fi = 0; fi < P.nfam; fi++) {
      if ((P.fam[fi].reversed_border != 0) != (q.reversed_border != 0)) continue;
      int id = 0, hamming = 0, rotation = 0, found = 0;
      const float margin = quad_decode_dev(P, P.fam[fi], fd.img, P.W0, P.H0, (int)fd.pitch, H, &id, &hamming, &rotation, &found);
      if (!(margin >= 0 && found)) continue;
      const uint32_t di = atomicAdd(&counters[frame].ndets, 1u);
      if (di >= P.dcap) { atomicOr(&counters[frame].flags, 0x10u); continue; }
      DetRec det;
      det.family = fi; det.id = id; det.hamming = hamming; det.decision_margin = margin;
      const double c = (rotation == 0) ? 1.0 : (rotation == 2) ? -1.0 : 0.0;
      const double s = (rotation == 1) ? 1.0 : (rotation == 3) ? -1.0 : 0.0;
      for (int r = 0; r < 3; r++) {
        det.H[r * 3 + 0] = H[r * 3 + 0] * c + H[r * 3 + 1] * s;
        det.H[r * 3 + 1] = H[r * 3 + 0] * -s + H[r * 3 + 1] * c;
        det.H[r * 3 + 2] = H[r * 3 + 2];
      }
      homography_project_dev(det.H, 0, 0, &det.c[0], &det.c[1]);
      for (int i = 0; i < 4; i++) {
        const double tcx = (i == 1 || i == 2) ? 1 : -1, tcy = (i < 2) ? 1 : -1;
        homography_project_dev(det.H, tcx, tcy, &det.p[i][0], &det.p[i][1]);
      }
      for (int i = 0; i < 9; i++) det.R[i] = 0;
      det.t[0] = det.t[1] = det.t[2] = 0;
      dets_all[(size_t)frame * P.dcap + di] = det;
    }
  }
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

__device__ void pose_from_homography_dev(const double* H, double fx_in, double fy, double cx, double cy, double tag_size,
                                         double* R, double* t) {
  const double fx = -fx_in;
  double R20 = H[6], R21 = H[7], TZ = H[8];
  double R00 = (H[0] - cx * R20) / fx, R01 = (H[1] - cx * R21) / fx, TX = (H[2] - cx * TZ) / fx;
  double R10 = (H[3] - cy * R20) / fy, R11 = (H[4] - cy * R21) / fy, TY = (H[5] - cy * TZ) / fy;
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
                                                  DetRec* __restrict__ out_all, FrameCounters* __restrict__ counters,
                                                  uint16_t* __restrict__ order_all, DetParams P) {
  const int frame = (int)blockIdx.x + P.frame0;
  uint32_t nd = counters[frame].ndets;
  if (nd > P.dcap) nd = P.dcap;
  const DetRec* dets = dets_all + (size_t)frame * P.dcap;
  DetRec* out = out_all + (size_t)frame * P.dcap;
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
    pose_from_homography_dev(d.H, fd.fx, fd.fy, fd.cx, fd.cy, P.tag_size, d.R, d.t);
    out[i] = d;
  }
}
