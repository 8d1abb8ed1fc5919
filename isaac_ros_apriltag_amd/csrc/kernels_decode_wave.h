// kernels_decode_wave.h -- S6 edge refinement + homography and S7 decode with ONE WAVE PER QUAD
// (inside cuAprilTagsDetect, reference src/apriltag_node.cpp:491-493).
//
// Lanes take the data-parallel axes -- edge samples (each lane walks the normal search of one sample),
// the (row, column) pairs of a Gaussian-elimination step, the border samples of the gray model, the data
// bits, the code table -- while every floating-point reduction whose order matters (line-fit moments,
// gray-model normal equations, decision-margin scores, back substitution) is evaluated in the CPU
// definition's order from LDS, so the results stay bit-identical.  The per-sample search sums
// (weight*n with integer weights and quarter-pixel n) are exact in double and need no ordering.
#pragma once
#include "common.h"
#include "kernels_decode.h"

#define DW_MAXS 64  // samples per chunk
#ifndef DW_KB
#define DW_KB 9     // steps of the normal search whose image gathers are in flight together (17 steps at full resolution)
#endif

// 6 waves per SIMD (80 registers, a few spilled): the per-quad work is chains of dependent double-precision operations in
// the CPU definition's order, so the stage's rate is set by how many quads are in flight (1.16 -> 0.80 ms per 256 noisy
// frames together with the larger grid; 5 and 7 waves measured slower)
#ifndef DW_WPE
#define DW_WPE 6
#endif
__global__ __launch_bounds__(64, DW_WPE) void k_decode_wave(const FrameDesc* __restrict__ frames, const QuadRec* __restrict__ quads_all,
                                                    DetRec* __restrict__ dets_all, FrameCounters* __restrict__ counters,
                                                    DetParams P) {
  __shared__ double s_bx[DW_MAXS], s_by[DW_MAXS];
  __shared__ int s_bok[DW_MAXS];
  __shared__ double s_lines[16];
  __shared__ float s_p[4][2], s_p0[4][2];   // corners: refined (or as fitted), and as fitted
  __shared__ double s_A[72];
  __shared__ double s_gx[80], s_gy[80], s_gv[80];
  __shared__ int s_gflag[80];  // bit0: valid, bit1: white
  __shared__ double s_C[2][3];
  __shared__ double s_values[144], s_sharp[64];

  const int frame = (int)blockIdx.y + P.frame0;
  const int lane = threadIdx.x;
  uint32_t nq = counters[frame].nquads;
  if (nq > P.qcap) nq = P.qcap;
  const FrameDesc fd = frames[frame];
  // the caller's image pointer is global by contract; out of the descriptor the compiler would have to assume generic
  const __attribute__((address_space(1))) uint8_t* im = (const __attribute__((address_space(1))) uint8_t*)fd.img;
  const int w = P.W0, h = P.H0, pitch = (int)fd.pitch;

  for (uint32_t qi = blockIdx.x; qi < nq; qi += gridDim.x) {
    __syncthreads();
    const QuadRec q = quads_all[(size_t)frame * P.qcap + qi];
    if (lane < 4) { s_p[lane][0] = s_p0[lane][0] = q.p[lane][0]; s_p[lane][1] = s_p0[lane][1] = q.p[lane][1]; }
    __syncthreads();

    // ---- S6 edge refinement -------------------------------------------------------------------
    // The four edges are refined TOGETHER: the lanes take the (edge, sample) pairs of all four edges, flattened edge by
    // edge (a tag side below 136 px has the minimum of 16 samples: one pass of the wave covers the quad), and lane e
    // accumulates edge e's moments in the CPU definition's sample order and fits its line.  (One edge after the other,
    // with every lane repeating the ordered sums, the stage's one-frame latency was four dependent rounds of image
    // gathers and line fits: 71 us for a noisy 1080p frame.)
    if (P.refine_edges) {
      // per-edge geometry, computed by every lane for the edge its sample belongs to (and by lane e for edge e)
      int ns[4], base[5];
      base[0] = 0;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int b = (e + 1) & 3;
        // (upstream: double nx = quad->p[b][1] - quad->p[a][1] with float p[][]: FLOAT operations, then widened)
        const double ex = (double)(q.p[b][1] - q.p[e][1]), ey = (double)(-q.p[b][0] + q.p[e][0]);
        const double mag = __dsqrt_rn(ex * ex + ey * ey);
        int n = (int)(mag / 8);
        if (n < 16) n = 16;
        ns[e] = n;
        base[e + 1] = base[e] + n;
      }
      const int total = base[4];
      const double range = P.decimate + 1;
      const int steps = (int)(2 * range * 4) + 1;
      double Mx = 0, My = 0, Mxx = 0, Mxy = 0, Myy = 0, N = 0;   // lane e < 4: moments of edge e
      for (int c0 = 0; c0 < total; c0 += DW_MAXS) {
        const int idx = c0 + lane;
        int ok = 0;
        double bestx = 0, besty = 0;
        if (idx < total) {
          const int edge = (idx >= base[1]) + (idx >= base[2]) + (idx >= base[3]);
          const int sidx = idx - (edge == 0 ? base[0] : edge == 1 ? base[1] : edge == 2 ? base[2] : base[3]);
          const int nsamples = edge == 0 ? ns[0] : edge == 1 ? ns[1] : edge == 2 ? ns[2] : ns[3];
          const int b = (edge + 1) & 3;
          const double pax = (double)s_p0[edge][0], pay = (double)s_p0[edge][1], pbx = (double)s_p0[b][0], pby = (double)s_p0[b][1];
          double nx = (double)(s_p0[b][1] - s_p0[edge][1]);    // float differences, as upstream
          double ny = (double)(-s_p0[b][0] + s_p0[edge][0]);
          const double mag = __dsqrt_rn(nx * nx + ny * ny);
          nx /= mag; ny /= mag;
          if (q.reversed_border) { nx = -nx; ny = -ny; }
          const double alpha = (1.0 + sidx) / (nsamples + 1);
          const double x0 = alpha * pax + (1 - alpha) * pbx;
          const double y0 = alpha * pay + (1 - alpha) * pby;
          double Mn = 0, Mcount = 0;  // exact sums: integer weights times multiples of 0.25
          // the image gathers of DW_KB steps are issued together (their addresses do not depend on each other); a step
          // outside the image reads pixel 0 and is not counted
          for (int k0 = 0; k0 < steps; k0 += DW_KB) {
            int g1[DW_KB], g2[DW_KB];
            bool in[DW_KB];
#pragma unroll
            for (int u = 0; u < DW_KB; u++) {
              const double n = -range + 0.25 * (k0 + u);
              const int x1 = (int)(x0 + (n + 1.0) * nx);
              const int y1 = (int)(y0 + (n + 1.0) * ny);
              const int x2 = (int)(x0 + (n - 1.0) * nx);
              const int y2 = (int)(y0 + (n - 1.0) * ny);
              in[u] = (k0 + u < steps) && !(x1 < 0 || x1 >= w || y1 < 0 || y1 >= h) && !(x2 < 0 || x2 >= w || y2 < 0 || y2 >= h);
              g1[u] = im[in[u] ? (size_t)y1 * pitch + x1 : (size_t)0];
              g2[u] = im[in[u] ? (size_t)y2 * pitch + x2 : (size_t)0];
            }
#pragma unroll
            for (int u = 0; u < DW_KB; u++) {
              if (!in[u] || g1[u] < g2[u]) continue;
              const double n = -range + 0.25 * (k0 + u);
              const double weight = (double)((g2[u] - g1[u]) * (g2[u] - g1[u]));
              Mn += weight * n;
              Mcount += weight;
            }
          }
          if (Mcount != 0) {
            const double n0 = Mn / Mcount;
            bestx = x0 + n0 * nx;
            besty = y0 + n0 * ny;
            ok = 1;
          }
        }
        s_bx[lane] = bestx; s_by[lane] = besty; s_bok[lane] = ok;
        __syncthreads();
        if (lane < 4) {   // this chunk's samples of edge `lane`, in sample order
          const int lo = (lane == 0 ? base[0] : lane == 1 ? base[1] : lane == 2 ? base[2] : base[3]);
          const int hi = (lane == 0 ? base[1] : lane == 1 ? base[2] : lane == 2 ? base[3] : base[4]);
          const int j0 = max(lo, c0) - c0, j1 = min(hi, c0 + DW_MAXS) - c0;
          for (int j = j0; j < j1; j++) {
            if (!s_bok[j]) continue;
            const double bx = s_bx[j], by = s_by[j];
            Mx += bx; My += by; Mxx += bx * bx; Mxy += bx * by; Myy += by * by; N++;
          }
        }
        __syncthreads();
      }
      if (lane < 4) {
        const double Ex = Mx / N, Ey = My / N;
        const double Cxx = Mxx / N - Ex * Ex, Cxy = Mxy / N - Ex * Ey, Cyy = Myy / N - Ey * Ey;
        const double disc = (Cxx - Cyy) * (Cxx - Cyy) + 4 * Cxy * Cxy;
        const double eig = 0.5 * (Cxx + Cyy + (double)at_sqrtf_rn((float)disc));
        const double nx1 = Cxx - eig, ny1 = Cxy, M1 = nx1 * nx1 + ny1 * ny1;
        const double nx2 = Cxy, ny2 = Cyy - eig, M2 = nx2 * nx2 + ny2 * ny2;
        double nx, ny, M;
        if (M1 > M2) { nx = nx1; ny = ny1; M = M1; } else { nx = nx2; ny = ny2; M = M2; }
        const double length = (double)at_sqrtf_rn((float)M);
        if (fabs(length) < 1e-12) { nx = 0; ny = 0; } else { nx = nx / length; ny = ny / length; }
        s_lines[lane * 4 + 0] = Ex; s_lines[lane * 4 + 1] = Ey; s_lines[lane * 4 + 2] = nx; s_lines[lane * 4 + 3] = ny;
      }
      __syncthreads();
      if (lane < 4) {
        const int i = lane, j = (lane + 1) & 3;
        const double A00 = s_lines[i * 4 + 3], A01 = -s_lines[j * 4 + 3];
        const double A10 = -s_lines[i * 4 + 2], A11 = s_lines[j * 4 + 2];
        const double B0 = -s_lines[i * 4 + 0] + s_lines[j * 4 + 0];
        const double B1 = -s_lines[i * 4 + 1] + s_lines[j * 4 + 1];
        const double det = A00 * A11 - A10 * A01;
        if (fabs(det) > 0.001) {
          const double W00 = A11 / det, W01 = -A01 / det;
          const double L0 = W00 * B0 + W01 * B1;
          s_p[i][0] = (float)(s_lines[i * 4 + 0] + L0 * A00);
          s_p[i][1] = (float)(s_lines[i * 4 + 1] + L0 * A10);
        }
      }
      __syncthreads();
    }

    // ---- homography: 8x9 Gaussian elimination with partial pivoting, lanes = matrix entries --------
    if (lane < 8) {
      const int i = lane >> 1;
      const double x = (i == 0 || i == 3) ? -1 : 1, y = (i == 0 || i == 1) ? -1 : 1;
      const double u = (double)s_p[i][0], v = (double)s_p[i][1];
      double* r = &s_A[lane * 9];
      if ((lane & 1) == 0) { r[0] = x; r[1] = y; r[2] = 1; r[3] = 0; r[4] = 0; r[5] = 0; r[6] = -x * u; r[7] = -y * u; r[8] = u; }
      else { r[0] = 0; r[1] = 0; r[2] = 0; r[3] = x; r[4] = y; r[5] = 1; r[6] = -x * v; r[7] = -y * v; r[8] = v; }
    }
    __syncthreads();
    bool singular = false;
    for (int col = 0; col < 8; col++) {
      double max_val = 0;
      int max_idx = -1;
      for (int row = col; row < 8; row++) {
        const double val = fabs(s_A[row * 9 + col]);
        if (val > max_val) { max_val = val; max_idx = row; }
      }
      if (max_val < 1e-10) { singular = true; break; }
      __syncthreads();
      if (max_idx != col && lane >= col && lane < 9) {
        const double t = s_A[col * 9 + lane];
        s_A[col * 9 + lane] = s_A[max_idx * 9 + lane];
        s_A[max_idx * 9 + lane] = t;
      }
      __syncthreads();
      const int ncols = 8 - col;            // columns col+1 .. 8
      const int npairs = (7 - col) * ncols;  // rows col+1 .. 7
      double nv = 0;
      int ti = 0, tj = 0;
      if (lane < npairs) {
        ti = col + 1 + lane / ncols;
        tj = col + 1 + lane % ncols;
        const double f = s_A[ti * 9 + col] / s_A[col * 9 + col];
        nv = s_A[ti * 9 + tj] - f * s_A[col * 9 + tj];
      }
      __syncthreads();
      if (lane < npairs) s_A[ti * 9 + tj] = nv;
      __syncthreads();
    }
    if (singular) continue;
    double H[9];
    {
      double xs[8];
#pragma unroll
      for (int col = 7; col >= 0; col--) {
        double sum = 0;
#pragma unroll
        for (int i = col + 1; i < 8; i++) sum += s_A[col * 9 + i] * xs[i];
        xs[col] = (s_A[col * 9 + 8] - sum) / s_A[col * 9 + col];
      }
#pragma unroll
      for (int i = 0; i < 8; i++) H[i] = xs[i];
      H[8] = 1;
    }

    // ---- S7 decode, once per enabled family ----------------------------------------------------
    for (int fi = 0; fi < P.nfam; fi++) {
      const FamilyDev& fam = P.fam[fi];   // (kernel argument: uniform loads)
      if ((fam.reversed_border != 0) != (q.reversed_border != 0)) continue;
      __syncthreads();
      const int wb = (int)fam.width_at_border, tw = (int)fam.total_width, nbits = (int)fam.nbits;
      // border samples of the two gray models: 8 lines x wb samples
      const int nsamp = 8 * wb;
      for (int sidx = lane; sidx < nsamp; sidx += 64) {
        const int pi = sidx / wb, i = sidx % wb;
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
        // upstream: (pattern[0] + i*pattern[2]) / (family->width_at_border) on float pattern[]: float multiply, add and DIVIDE
        const double tagx01 = (double)__fdiv_rn(p0 + (float)i * p2, (float)wb);
        const double tagy01 = (double)__fdiv_rn(p1 + (float)i * p3, (float)wb);
        const double tagx = 2 * (tagx01 - 0.5), tagy = 2 * (tagy01 - 0.5);
        double px, py;
        homography_project_dev(H, tagx, tagy, &px, &py);
        const int ix = (int)px, iy = (int)py;
        int flag = is_white ? 2 : 0;
        double v = 0;
        if (!(ix < 0 || iy < 0 || ix >= w || iy >= h)) { flag |= 1; v = (double)im[(size_t)iy * pitch + ix]; }
        s_gx[sidx] = tagx; s_gy[sidx] = tagy; s_gv[sidx] = v; s_gflag[sidx] = flag;
      }
      for (int i = lane; i < tw * tw; i += 64) s_values[i] = 0;
      __syncthreads();
      if (lane < 2) {  // lane 0: white model, lane 1: black model; sample order of the CPU definition
        GrayModel g = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const int want = (lane == 0) ? 3 : 1;
        for (int sidx = 0; sidx < nsamp; sidx++)
          if (s_gflag[sidx] == want) graymodel_add_dev(g, s_gx[sidx], s_gy[sidx], s_gv[sidx]);
        graymodel_solve_dev(g);
        s_C[lane][0] = g.C0; s_C[lane][1] = g.C1; s_C[lane][2] = g.C2;
      }
      __syncthreads();
      GrayModel whitemodel, blackmodel;
      whitemodel.C0 = s_C[0][0]; whitemodel.C1 = s_C[0][1]; whitemodel.C2 = s_C[0][2];
      blackmodel.C0 = s_C[1][0]; blackmodel.C1 = s_C[1][1]; blackmodel.C2 = s_C[1][2];
      if ((graymodel_interp_dev(whitemodel, 0, 0) - graymodel_interp_dev(blackmodel, 0, 0) < 0) != (fam.reversed_border != 0)) continue;

      // data bits: one lane per bit (nbits <= 64), cells from the family's layout
      const int min_coord = (wb - tw) / 2;
      int gx = 0, gy = 0;
      // (the layout arrays are indexed per lane: packed four to a dword in the kernel arguments)
      const int bitx = (int)(int8_t)(reinterpret_cast<const uint32_t*>(fam.bit_x)[(lane & 63) >> 2] >> (8 * (lane & 3)));
      const int bity = (int)(int8_t)(reinterpret_cast<const uint32_t*>(fam.bit_y)[(lane & 63) >> 2] >> (8 * (lane & 3)));
      const int rsrc = (int)(uint8_t)(reinterpret_cast<const uint32_t*>(fam.rot_src)[(lane & 63) >> 2] >> (8 * (lane & 3)));
      if (lane < nbits) {
        gx = bitx - min_coord; gy = bity - min_coord;
        const double tagx01 = (bitx + 0.5) / wb, tagy01 = (bity + 0.5) / wb;
        const double tagx = 2 * (tagx01 - 0.5), tagy = 2 * (tagy01 - 0.5);
        double px, py;
        homography_project_dev(H, tagx, tagy, &px, &py);
        const double v = value_for_pixel_dev(im, w, h, pitch, px, py);
        if (v != -1) {
          const double thresh = (graymodel_interp_dev(blackmodel, tagx, tagy) + graymodel_interp_dev(whitemodel, tagx, tagy)) / 2.0;
          s_values[tw * gy + gx] = v - thresh;
        }
      }
      __syncthreads();
      bool bit = false;
      if (lane < nbits) {
        double s = 0;
        if (gy - 1 >= 0) s += s_values[(gy - 1) * tw + gx] * -1.0;
        if (gx - 1 >= 0) s += s_values[gy * tw + gx - 1] * -1.0;
        s += s_values[gy * tw + gx] * 4.0;
        if (gx + 1 <= tw - 1) s += s_values[gy * tw + gx + 1] * -1.0;
        if (gy + 1 <= tw - 1) s += s_values[(gy + 1) * tw + gx] * -1.0;
        const double v = s_values[gy * tw + gx] + P.decode_sharpening * s;
        s_sharp[lane] = v;
        bit = v > 0;
      }
      const unsigned long long mask = __ballot(bit);
      // bit i of the code is data cell i counted from the MSB
      uint64_t rcode = __brevll(mask) >> (64 - nbits);
      __syncthreads();
      float black_score = 0, white_score = 0, black_count = 1, white_count = 1;
      for (int i = 0; i < nbits; i++) {
        const double v = s_sharp[i];
        if (v > 0) { white_score = (float)((double)white_score + v); white_count++; }
        else { black_score = (float)((double)black_score - v); black_count++; }
      }
      int id = 0, hamming = 0, rotation = 0;
      bool found = false;
      for (int r = 0; r < 4 && !found; r++) {
        int best = 1 << 30, bid = 1 << 30;
        for (uint32_t c = lane; c < fam.ncodes; c += 64) {
          const int hd = __popcll(rcode ^ fam.codes[c]);
          if (hd < best) { best = hd; bid = (int)c; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
          const int ob = __shfl_xor(best, off, 64), oi = __shfl_xor(bid, off, 64);
          if (ob < best || (ob == best && oi < bid)) { best = ob; bid = oi; }
        }
        if (best <= P.max_hamming) { id = bid; hamming = best; rotation = r; found = true; }
        else {   // pattern rotated by 90 degrees: bit i takes the value of bit rot_src[i]
          const bool rb = lane < nbits && ((rcode >> (nbits - 1 - rsrc)) & 1ull);
          rcode = __brevll(__ballot(rb)) >> (64 - nbits);
        }
      }
      const float ma = white_score / white_count, mb = black_score / black_count;
      const float margin = ma < mb ? ma : mb;
      if (!(margin >= 0 && found)) continue;
      if (lane == 0) {
        const uint32_t di = atomicAdd(&counters[frame].ndets, 1u);
        if (di >= P.dcap) {
          atomicOr(&counters[frame].flags, 0x10u);
        } else {
          DetRec det;
          det.family = fi; det.id = id; det.hamming = hamming; det.decision_margin = margin;
          // H' = H * Rz(rotation * 90 deg) as upstream forms it: libm's cos / sin of rotation * M_PI / 2 (their correctly
          // rounded values as literals: cos(pi/2) is 6.1e-17, not 0) and the full product of matd_op("M*M"), acc = 0, k = 0, 1, 2
          const double c = (rotation == 0) ? 1.0 : (rotation == 1) ? 6.123233995736766e-17 : (rotation == 2) ? -1.0 : -1.8369701987210297e-16;
          const double sn = (rotation == 0) ? 0.0 : (rotation == 1) ? 1.0 : (rotation == 2) ? 1.2246467991473532e-16 : -1.0;
          const double Rz[9] = {c, -sn, 0.0, sn, c, 0.0, 0.0, 0.0, 1.0};
#pragma unroll
          for (int r = 0; r < 3; r++) {
#pragma unroll
            for (int cc = 0; cc < 3; cc++) {
              double acc = 0;
#pragma unroll
              for (int k = 0; k < 3; k++) acc += H[r * 3 + k] * Rz[k * 3 + cc];
              det.H[r * 3 + cc] = acc;
            }
          }
          homography_project_dev(det.H, 0, 0, &det.c[0], &det.c[1]);
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const double tcx = (i == 1 || i == 2) ? 1 : -1, tcy = (i < 2) ? 1 : -1;
            homography_project_dev(det.H, tcx, tcy, &det.p[i][0], &det.p[i][1]);
          }
#pragma unroll
          for (int i = 0; i < 9; i++) det.R[i] = 0;
          det.t[0] = det.t[1] = det.t[2] = 0;
          dets_all[(size_t)frame * P.dcap + di] = det;
        }
      }
    }
  }
}
