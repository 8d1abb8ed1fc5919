// kernels_quad.h -- S5 quad fitting (inside cuAprilTagsDetect, reference
// src/apriltag_node.cpp:491-493).  One 256-thread workgroup per cluster:
//   bbox / gradient-dot by integer block reductions -> slope keys -> in-place bitonic sort (LDS up
//   to 4096 points, global scratch above) -> per-point weighted moment terms in parallel -> the
//   cumulative moments as ONE sequential double chain per moment (6 lanes), which is what makes the
//   result bit-identical to the sequential CPU definition -> windowed line-fit errors, 7-tap
//   smoothing, local maxima, top-10 selection, all C(10,4) corner choices evaluated from a
//   precomputed table of pairwise segment fits -> 4 line fits, intersections, area/angle checks.
#pragma once
#include "common.h"

#define FQ_SORT_LDS 4096

__device__ __forceinline__ int block_reduce_min_i(int v, int* scratch) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
  if (lane_id() == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  int r = min(min(scratch[0], scratch[1]), min(scratch[2], scratch[3]));
  __syncthreads();
  return r;
}
__device__ __forceinline__ int block_reduce_max_i(int v, int* scratch) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
  if (lane_id() == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  int r = max(max(scratch[0], scratch[1]), max(scratch[2], scratch[3]));
  __syncthreads();
  return r;
}
__device__ __forceinline__ long long block_reduce_sum_ll(long long v, long long* scratch) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  if (lane_id() == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  long long r = scratch[0] + scratch[1] + scratch[2] + scratch[3];
  __syncthreads();
  return r;
}

// Line fit over the cumulative moments lf[i*6 + {Mx,My,Mxx,Mxy,Myy,W}] of points i0..i1 (circular).
__device__ __forceinline__ void fit_line_dev(const double* lf, int sz, int i0, int i1, double* lineparm, double* err,
                                             double* mse) {
  double Mx, My, Mxx, Mxy, Myy, W;
  int N;
  const double* b = lf + (size_t)i1 * 6;
  if (i0 < i1) {
    N = i1 - i0 + 1;
    Mx = b[0]; My = b[1]; Mxx = b[2]; Mxy = b[3]; Myy = b[4]; W = b[5];
    if (i0 > 0) {
      const double* a = lf + (size_t)(i0 - 1) * 6;
      Mx -= a[0]; My -= a[1]; Mxx -= a[2]; Mxy -= a[3]; Myy -= a[4]; W -= a[5];
    }
  } else {
    const double* e = lf + (size_t)(sz - 1) * 6;
    const double* a = lf + (size_t)(i0 - 1) * 6;
    Mx = e[0] - a[0]; My = e[1] - a[1]; Mxx = e[2] - a[2]; Mxy = e[3] - a[3]; Myy = e[4] - a[4]; W = e[5] - a[5];
    Mx += b[0]; My += b[1]; Mxx += b[2]; Mxy += b[3]; Myy += b[4]; W += b[5];
    N = sz - i0 + i1 + 1;
  }
  const double Ex = Mx / W, Ey = My / W;
  const double Cxx = Mxx / W - Ex * Ex, Cxy = Mxy / W - Ex * Ey, Cyy = Myy / W - Ey * Ey;
  const double disc = (Cxx - Cyy) * (Cxx - Cyy) + 4 * Cxy * Cxy;
  const float rootf = at_sqrtf_rn((float)disc);
  const double eig_small = 0.5 * (Cxx + Cyy - (double)rootf);
  if (lineparm) {
    lineparm[0] = Ex; lineparm[1] = Ey;
    const double eig = 0.5 * (Cxx + Cyy + (double)rootf);
    const double nx1 = Cxx - eig, ny1 = Cxy, M1 = nx1 * nx1 + ny1 * ny1;
    const double nx2 = Cxy, ny2 = Cyy - eig, M2 = nx2 * nx2 + ny2 * ny2;
    double nx, ny, M;
    if (M1 > M2) { nx = nx1; ny = ny1; M = M1; } else { nx = nx2; ny = ny2; M = M2; }
    const double length = (double)at_sqrtf_rn((float)M);
    if (fabs(length) < 1e-12) { lineparm[2] = 0; lineparm[3] = 0; }
    else { lineparm[2] = nx / length; lineparm[3] = ny / length; }
  }
  if (err) *err = N * eig_small;
  if (mse) *mse = eig_small;
}

template <typename KeyPtr>
__device__ __forceinline__ void bitonic_sort_block(KeyPtr A, int n) {
  // all-ascending bitonic network; indices >= n act as +infinity and are never touched
  int npow = 1;
  while (npow < n) npow <<= 1;
  const int half = npow >> 1;
  for (int k = 2; k <= npow; k <<= 1) {
    const int hk = k >> 1;
    for (int i = threadIdx.x; i < half; i += 256) {
      const int blk = i / hk, off = i % hk;
      const int a = blk * k + off, b = blk * k + k - 1 - off;
      if (b < n) {
        unsigned long long x = A[a], y = A[b];
        if (x > y) { A[a] = y; A[b] = x; }
      }
    }
    __syncthreads();
    for (int j = k >> 2; j >= 1; j >>= 1) {
      for (int i = threadIdx.x; i < half; i += 256) {
        const int a = (i / j) * 2 * j + (i % j), b = a + j;
        if (b < n) {
          unsigned long long x = A[a], y = A[b];
          if (x > y) { A[a] = y; A[b] = x; }
        }
      }
      __syncthreads();
    }
  }
}

__global__ __launch_bounds__(256) void k_fit_quads(const FrameDesc* __restrict__ frames, const uint8_t* __restrict__ gray_all,
                                                   const uint32_t* __restrict__ pts_all, const ClusterRec* __restrict__ clusters_all,
                                                   unsigned long long* __restrict__ keys_all, double* __restrict__ lf_all,
                                                   double* __restrict__ errs_a_all, double* __restrict__ errs_b_all,
                                                   QuadRec* __restrict__ quads_all, FrameCounters* __restrict__ counters,
                                                   DetParams P) {
  __shared__ unsigned long long skeys[FQ_SORT_LDS];
  __shared__ long long sred_ll[4];
  __shared__ int sred_i[4];
  __shared__ double sred_d[4];
  __shared__ int sred_di[4];
  __shared__ int s_removed[12];
  __shared__ int s_nrem;
  __shared__ double s_thresh;
  __shared__ int s_maxidx[16];
  __shared__ int s_nmaxkept;
  __shared__ double s_ferr[100], s_fmse[100], s_fnx[100], s_fny[100], s_werr[100], s_wmse[100];

  const int frame = blockIdx.y;
  const int tid = threadIdx.x;
  const FrameDesc fd = frames[frame];
  const uint8_t* gray = (P.decimate > 1) ? gray_all + (size_t)frame * P.H * P.WS : fd.img;
  const int gpitch = (P.decimate > 1) ? P.WS : (int)fd.pitch;
  const int W = P.W, H = P.H;
  uint32_t ncl = counters[frame].nclusters;
  if (ncl > P.ccap) ncl = P.ccap;

  for (uint32_t ci = blockIdx.x; ci < ncl; ci += gridDim.x) {
    __syncthreads();
    const ClusterRec cl = clusters_all[(size_t)frame * P.ccap + ci];
    const int sz = (int)cl.count;
    const uint32_t* pts = pts_all + (size_t)frame * P.pcap + cl.start;
    if (sz < 24) continue;

    // ---- bbox and exact gradient dot -----------------------------------------------------------
    int xmin = 1 << 30, xmax = -1, ymin = 1 << 30, ymax = -1;
    long long sxg = 0, sgx = 0, sgy = 0;
    for (int i = tid; i < sz; i += 256) {
      const uint32_t p = pts[i];
      const int x = (int)(p >> 18), y = (int)((p >> 4) & 0x3FFF);
      const int gx = ((int)((p >> 2) & 3) - 1) * 255, gy = ((int)(p & 3) - 1) * 255;
      xmin = min(xmin, x); xmax = max(xmax, x); ymin = min(ymin, y); ymax = max(ymax, y);
      sxg += (long long)x * gx + (long long)y * gy;
      sgx += gx; sgy += gy;
    }
    xmin = block_reduce_min_i(xmin, sred_i); xmax = block_reduce_max_i(xmax, sred_i);
    ymin = block_reduce_min_i(ymin, sred_i); ymax = block_reduce_max_i(ymax, sred_i);
    sxg = block_reduce_sum_ll(sxg, sred_ll); sgx = block_reduce_sum_ll(sgx, sred_ll); sgy = block_reduce_sum_ll(sgy, sred_ll);
    if ((xmax - xmin) * (ymax - ymin) < P.min_tag_width) continue;
    const double cxd = (xmin + xmax) * 0.5 + 0.05118, cyd = (ymin + ymax) * 0.5 + -0.028581;
    const double dot = (double)sxg - cxd * (double)sgx - cyd * (double)sgy;
    const int q_reversed = dot < 0;
    if (!P.reversed_border && q_reversed) continue;
    if (!P.normal_border && !q_reversed) continue;

    // ---- slope keys + sort -----------------------------------------------------------------------
    const float cx = (float)cxd, cy = (float)cyd;
    const bool in_lds = sz <= FQ_SORT_LDS;
    unsigned long long* gkeys = keys_all + (size_t)frame * P.pcap + cl.start;
    for (int i = tid; i < sz; i += 256) {
      const uint32_t p = pts[i];
      const int x = (int)(p >> 18), y = (int)((p >> 4) & 0x3FFF);
      float dx = (float)x - cx, dy = (float)y - cy;
      float quadrant;
      if (dy > 0) quadrant = (dx > 0) ? 65536.0f : 131072.0f;
      else quadrant = (dx > 0) ? 0.0f : -65536.0f;
      if (dy < 0) { dy = -dy; dx = -dx; }
      if (dx < 0) { float tmp = dx; dx = dy; dy = -tmp; }
      const float slope = quadrant + __fdiv_rn(dy, dx);
      const unsigned long long key = ((unsigned long long)float_sortable(slope) << 32) | ((unsigned long long)y << 18) |
                                     ((unsigned long long)x << 4) | (unsigned long long)(p & 15u);
      if (in_lds) skeys[i] = key; else gkeys[i] = key;
    }
    __syncthreads();
    if (in_lds) bitonic_sort_block(skeys, sz); else bitonic_sort_block(gkeys, sz);

    // ---- per-point weighted moment terms (parallel), then the sequential cumulative sums ---------
    double* lf = lf_all + ((size_t)frame * P.pcap + cl.start) * 6;
    for (int i = tid; i < sz; i += 256) {
      const unsigned long long key = in_lds ? skeys[i] : gkeys[i];
      const int px = (int)((key >> 4) & 0x3FFF), py = (int)((key >> 18) & 0x3FFF);
      const double x = px * .5 + 0.5, y = py * .5 + 0.5;
      const int ix = (int)x, iy = (int)y;
      double Wt = 1;
      if (ix > 0 && ix + 1 < W && iy > 0 && iy + 1 < H) {
        const int grad_x = (int)gray[(size_t)iy * gpitch + ix + 1] - (int)gray[(size_t)iy * gpitch + ix - 1];
        const int grad_y = (int)gray[(size_t)(iy + 1) * gpitch + ix] - (int)gray[(size_t)(iy - 1) * gpitch + ix];
        Wt = __dsqrt_rn((double)(grad_x * grad_x + grad_y * grad_y)) + 1;
      }
      double* o = lf + (size_t)i * 6;
      o[0] = Wt * x; o[1] = Wt * y; o[2] = Wt * x * x; o[3] = Wt * x * y; o[4] = Wt * y * y; o[5] = Wt;
    }
    __syncthreads();
    if (tid < 6) {
      double acc = 0;
      int i = 0;
      for (; i + 8 <= sz; i += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = lf[(size_t)(i + u) * 6 + tid];
#pragma unroll
        for (int u = 0; u < 8; u++) { acc += v[u]; lf[(size_t)(i + u) * 6 + tid] = acc; }
      }
      for (; i < sz; i++) { acc += lf[(size_t)i * 6 + tid]; lf[(size_t)i * 6 + tid] = acc; }
    }
    __syncthreads();

    // ---- windowed line-fit error, smoothing ------------------------------------------------------
    const int ksz = min(20, sz / 12);
    double* ea = errs_a_all + (size_t)frame * P.pcap + cl.start;
    double* eb = errs_b_all + (size_t)frame * P.pcap + cl.start;
    for (int i = tid; i < sz; i += 256) {
      double e;
      fit_line_dev(lf, sz, (i + sz - ksz) % sz, (i + ksz) % sz, nullptr, &e, nullptr);
      ea[i] = e;
    }
    __syncthreads();
    {
      const float f0 = 0x1.6c0504p-7f, f1 = 0x1.152aaap-3f, f2 = 0x1.368b3p-1f;
      const double F[7] = {(double)f0, (double)f1, (double)f2, 1.0, (double)f2, (double)f1, (double)f0};
      for (int i = tid; i < sz; i += 256) {
        double acc = 0;
#pragma unroll
        for (int k = 0; k < 7; k++) acc += ea[(i + k - 3 + sz) % sz] * F[k];
        eb[i] = acc;
      }
    }
    __syncthreads();

    // ---- local maxima ----------------------------------------------------------------------------
    int mycount = 0;
    for (int i = tid; i < sz; i += 256) {
      const double e = eb[i];
      if (e > eb[(i + 1) % sz] && e > eb[(i + sz - 1) % sz]) mycount++;
    }
    const int nmaxima = (int)block_reduce_sum_ll(mycount, sred_ll);
    if (nmaxima < 4) continue;
    if (tid == 0) { s_nrem = 0; s_nmaxkept = 0; s_thresh = 0; }
    __syncthreads();
    const bool select = nmaxima > P.max_nmaxima;
    if (select) {
      // value of the (max_nmaxima+1)-th largest maximum: remove the current largest max_nmaxima+1 times
      for (int round = 0; round <= P.max_nmaxima; round++) {
        double bv = 0; int bi = -1;
        const int nrem = s_nrem;
        for (int i = tid; i < sz; i += 256) {
          const double e = eb[i];
          if (!(e > eb[(i + 1) % sz] && e > eb[(i + sz - 1) % sz])) continue;
          bool removed = false;
          for (int r = 0; r < nrem; r++) removed |= (s_removed[r] == i);
          if (removed) continue;
          if (bi < 0 || e > bv) { bv = e; bi = i; }
        }
        // block argmax (value, then lowest index)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
          const double ov = __shfl_xor(bv, off, 64);
          const int oi = __shfl_xor(bi, off, 64);
          if (oi >= 0 && (bi < 0 || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
        }
        if (lane_id() == 0) { sred_d[tid >> 6] = bv; sred_di[tid >> 6] = bi; }
        __syncthreads();
        if (tid == 0) {
          double v = 0; int ix = -1;
          for (int w = 0; w < 4; w++) {
            const double ov = sred_d[w]; const int oi = sred_di[w];
            if (oi >= 0 && (ix < 0 || ov > v || (ov == v && oi < ix))) { v = ov; ix = oi; }
          }
          s_removed[s_nrem++] = ix;
          s_thresh = v;
        }
        __syncthreads();
      }
    }
    const double thresh = s_thresh;
    for (int i = tid; i < sz; i += 256) {
      const double e = eb[i];
      if (!(e > eb[(i + 1) % sz] && e > eb[(i + sz - 1) % sz])) continue;
      if (select && e <= thresh) continue;
      const int k = atomicAdd(&s_nmaxkept, 1);
      if (k < 16) s_maxidx[k] = i;
    }
    __syncthreads();
    const int m = min(s_nmaxkept, 10);
    if (tid == 0) {  // ascending index order (<= 10 entries)
      for (int a = 1; a < m; a++) {
        const int v = s_maxidx[a];
        int b = a - 1;
        while (b >= 0 && s_maxidx[b] > v) { s_maxidx[b + 1] = s_maxidx[b]; b--; }
        s_maxidx[b + 1] = v;
      }
    }
    __syncthreads();
    if (m < 4) continue;

    // ---- pairwise segment fits, then all corner quadruples ---------------------------------------
    if (tid < 200) {
      const int t = tid % 100, a = t / 10, b = t % 10;
      if (a < b && b < m) {
        if (tid < 100) {
          double prm[4], e, ms;
          fit_line_dev(lf, sz, s_maxidx[a], s_maxidx[b], prm, &e, &ms);
          s_ferr[t] = e; s_fmse[t] = ms; s_fnx[t] = prm[2]; s_fny[t] = prm[3];
        } else {
          double e, ms;
          fit_line_dev(lf, sz, s_maxidx[b], s_maxidx[a], nullptr, &e, &ms);
          s_werr[t] = e; s_wmse[t] = ms;
        }
      }
    }
    __syncthreads();
    double best_err = (double)HUGE_VALF;
    int best_t = 1 << 30;
    if (tid < 210) {
      int c = 0, q0 = 0, q1 = 0, q2 = 0, q3 = 0;
      bool found = false;
      for (int m0 = 0; m0 < 7 && !found; m0++)
        for (int m1 = m0 + 1; m1 < 8 && !found; m1++)
          for (int m2 = m1 + 1; m2 < 9 && !found; m2++)
            for (int m3 = m2 + 1; m3 < 10; m3++) {
              if (c == tid) { q0 = m0; q1 = m1; q2 = m2; q3 = m3; found = true; break; }
              c++;
            }
      if (found && q3 < m) {
        const double mse01 = s_fmse[q0 * 10 + q1], mse12 = s_fmse[q1 * 10 + q2], mse23 = s_fmse[q2 * 10 + q3];
        const double mse30 = s_wmse[q0 * 10 + q3];
        const double dotn = s_fnx[q0 * 10 + q1] * s_fnx[q1 * 10 + q2] + s_fny[q0 * 10 + q1] * s_fny[q1 * 10 + q2];
        if (!(mse01 > P.max_line_fit_mse) && !(mse12 > P.max_line_fit_mse) && !(fabs(dotn) > P.cos_critical_rad) &&
            !(mse23 > P.max_line_fit_mse) && !(mse30 > P.max_line_fit_mse)) {
          const double e = s_ferr[q0 * 10 + q1] + s_ferr[q1 * 10 + q2] + s_ferr[q2 * 10 + q3] + s_werr[q0 * 10 + q3];
          if (e < best_err) { best_err = e; best_t = tid; }
        }
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const double ov = __shfl_xor(best_err, off, 64);
      const int ot = __shfl_xor(best_t, off, 64);
      if (ov < best_err || (ov == best_err && ot < best_t)) { best_err = ov; best_t = ot; }
    }
    if (lane_id() == 0) { sred_d[tid >> 6] = best_err; sred_di[tid >> 6] = best_t; }
    __syncthreads();
    if (tid == 0) {
      double be = sred_d[0]; int bt = sred_di[0];
      for (int w = 1; w < 4; w++)
        if (sred_d[w] < be || (sred_d[w] == be && sred_di[w] < bt)) { be = sred_d[w]; bt = sred_di[w]; }
      bool ok = (be != (double)HUGE_VALF) && (be / sz < P.max_line_fit_mse);
      int indices[4] = {0, 0, 0, 0};
      if (ok) {
        int c = 0;
        bool found = false;
        for (int m0 = 0; m0 < 7 && !found; m0++)
          for (int m1 = m0 + 1; m1 < 8 && !found; m1++)
            for (int m2 = m1 + 1; m2 < 9 && !found; m2++)
              for (int m3 = m2 + 1; m3 < 10; m3++) {
                if (c == bt) {
                  indices[0] = s_maxidx[m0]; indices[1] = s_maxidx[m1]; indices[2] = s_maxidx[m2]; indices[3] = s_maxidx[m3];
                  found = true;
                  break;
                }
                c++;
              }
      }
      QuadRec q;
      double lines[4][4];
      for (int i = 0; i < 4 && ok; i++) {
        double ms;
        fit_line_dev(lf, sz, indices[i], indices[(i + 1) & 3], lines[i], nullptr, &ms);
        if (ms > P.max_line_fit_mse) ok = false;
      }
      for (int i = 0; i < 4 && ok; i++) {
        const double A00 = lines[i][3], A01 = -lines[(i + 1) & 3][3];
        const double A10 = -lines[i][2], A11 = lines[(i + 1) & 3][2];
        const double B0 = -lines[i][0] + lines[(i + 1) & 3][0];
        const double B1 = -lines[i][1] + lines[(i + 1) & 3][1];
        const double det = A00 * A11 - A10 * A01;
        if (fabs(det) < 0.001) { ok = false; break; }
        const double W00 = A11 / det, W01 = -A01 / det;
        const double L0 = W00 * B0 + W01 * B1;
        q.p[i][0] = (float)(lines[i][0] + L0 * A00);
        q.p[i][1] = (float)(lines[i][1] + L0 * A10);
      }
      if (ok) {
        double area = 0, length[3], pp;
        for (int i = 0; i < 3; i++) {
          const int a = i, b = (i + 1) % 3;
          const double ddx = (double)q.p[b][0] - (double)q.p[a][0], ddy = (double)q.p[b][1] - (double)q.p[a][1];
          length[i] = __dsqrt_rn(ddx * ddx + ddy * ddy);
        }
        pp = (length[0] + length[1] + length[2]) / 2;
        area += __dsqrt_rn(pp * (pp - length[0]) * (pp - length[1]) * (pp - length[2]));
        const int idxs[4] = {2, 3, 0, 2};
        for (int i = 0; i < 3; i++) {
          const int a = idxs[i], b = idxs[i + 1];
          const double ddx = (double)q.p[b][0] - (double)q.p[a][0], ddy = (double)q.p[b][1] - (double)q.p[a][1];
          length[i] = __dsqrt_rn(ddx * ddx + ddy * ddy);
        }
        pp = (length[0] + length[1] + length[2]) / 2;
        area += __dsqrt_rn(pp * (pp - length[0]) * (pp - length[1]) * (pp - length[2]));
        if (area < 0.95 * P.min_tag_width * P.min_tag_width) ok = false;
      }
      for (int i = 0; i < 4 && ok; i++) {
        const int i0 = i, i1 = (i + 1) & 3, i2 = (i + 2) & 3;
        const double dx1 = (double)q.p[i1][0] - (double)q.p[i0][0], dy1 = (double)q.p[i1][1] - (double)q.p[i0][1];
        const double dx2 = (double)q.p[i2][0] - (double)q.p[i1][0], dy2 = (double)q.p[i2][1] - (double)q.p[i1][1];
        const double cos_dtheta = (dx1 * dx2 + dy1 * dy2) / __dsqrt_rn((dx1 * dx1 + dy1 * dy1) * (dx2 * dx2 + dy2 * dy2));
        if ((cos_dtheta > P.cos_critical_rad || cos_dtheta < -P.cos_critical_rad) || dx1 * dy2 < dy1 * dx2) ok = false;
      }
      if (ok) {
        if (P.decimate > 1) {
          const double f = (double)(float)P.decimate;
          for (int c = 0; c < 4; c++) {
            q.p[c][0] = (float)(((double)q.p[c][0] - 0.5) * f + 0.5);
            q.p[c][1] = (float)(((double)q.p[c][1] - 0.5) * f + 0.5);
          }
        }
        q.reversed_border = q_reversed;
        q.pad = 0;
        q.key = cl.key;
        const uint32_t qi = atomicAdd(&counters[frame].nquads, 1u);
        if (qi < P.qcap) quads_all[(size_t)frame * P.qcap + qi] = q;
        else atomicOr(&counters[frame].flags, 0x8u);
      }
    }
  }
}
