// kernels_quad.h -- S5 quad fitting (inside cuAprilTagsDetect, reference
// src/apriltag_node.cpp:491-493).  One workgroup per cluster, five launch classes by cluster size
// (one wave for <= 768 points ... 1024 threads above 8192) so that small clusters do not pay for idle
// waves and the slope sort always runs in LDS (6 KB ... 145 KB of keys).  Every class is one launch of
// PERSISTENT workgroups: k_cluster_select has bucketed the clusters of all frames of the submission into one
// compact work list per class, and a workgroup pops the next cluster with one atomic until its list is
// empty -- no workgroup walks clusters of another class, and the cumulative-moment array of a cluster lives
// in a scratch slot owned by the workgroup (reused cluster after cluster, so it stays cache-resident) instead
// of 48 bytes per boundary point of every frame.  Per cluster:
//   bbox / gradient-dot: seven DPP wave reductions, one barrier -> slope keys, stored so that their order
//   as IEEE doubles is the wanted order -> bitonic sort in LDS on an array padded to a power of two with
//   +infinity keys (first three levels in registers, then up to three network steps per pass, no bounds
//   tests, compare-exchange = v_min_f64 + v_max_f64) -> moment sweep: duplicate points dropped, weighted
//   moment terms summed EXACTLY and every prefix rounded once (bit-identical to the CPU definition in any
//   order).  Fast path (images up to 2048 x 2048): sums carried as two doubles, two walks over
//   lane-contiguous points with one scan per cluster; general path: 128-bit fixed point, one DPP scan per
//   chunk.  -> windowed line-fit errors, 7-tap smoothing -> local maxima compacted into LDS -> wave 0
//   alone: top-10 selection (rank among <= 64 candidates, else 11 arg-max rounds) -> table of the 45
//   pairwise segment fits (all threads) -> wave 0 alone: best of the C(10,4) corner choices, 4 line fits,
//   intersections, lane-parallel area / angle checks.
// Tools-only builds: -DAMDAT_FQ_PROFILE (per-phase shader-cycle counters), -DAMDAT_FQ_STOP=n (drop every
// cluster after phase n, for per-phase instruction counts); isaac_ros_apriltag_amd.build.build_amd_variant.
// The product build carries neither.
#pragma once
#include "common.h"
#include "tools_hooks.h"


// ---- wave reductions on the DPP network (row_shr 1/2/4/8 inside each row of 16 lanes, then the row
// broadcasts 15 and 31): after the six steps lane 63 holds the reduction of the whole wave.  For the
// idempotent operations a lane without a source keeps its own value (old = v, bound_ctrl off).
#define AT_DPP_STEPS(OP)          \
  OP(0x111, 0xF) OP(0x112, 0xF) OP(0x114, 0xF) OP(0x118, 0xF) OP(0x142, 0xA) OP(0x143, 0xC)
__device__ __forceinline__ int wave_min_i(int v) {
#define OP(C, M) v = min(v, __builtin_amdgcn_update_dpp(v, v, C, M, 0xF, false));
  AT_DPP_STEPS(OP)
#undef OP
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_max_i(int v) {
#define OP(C, M) v = max(v, __builtin_amdgcn_update_dpp(v, v, C, M, 0xF, false));
  AT_DPP_STEPS(OP)
#undef OP
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ long long wave_sum_ll(long long v) {
#define OP(C, M)                                                                                         \
  {                                                                                                      \
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, C, M, 0xF, true);      \
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)((unsigned long long)v >> 32), C, M, 0xF, true); \
    v += (long long)((unsigned long long)lo | ((unsigned long long)hi << 32));                           \
  }
  AT_DPP_STEPS(OP)
#undef OP
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((unsigned long long)v >> 32), 63);
  return (long long)((unsigned long long)lo | ((unsigned long long)hi << 32));
}
// max of an unsigned 64-bit key over the wave (used for arg-max with an order-preserving key)
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#define OP(C, M)                                                                                         \
  {                                                                                                      \
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)v, (int)(uint32_t)v, C, M, 0xF, false);       \
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(v >> 32), (int)(uint32_t)(v >> 32), C, M, 0xF, false); \
    const unsigned long long o = (unsigned long long)lo | ((unsigned long long)hi << 32);                \
    v = o > v ? o : v;                                                                                   \
  }
  AT_DPP_STEPS(OP)
#undef OP
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63);
  return (unsigned long long)lo | ((unsigned long long)hi << 32);
}
// order-preserving map double -> u64 (total order of the finite values, -0 < +0)
__device__ __forceinline__ unsigned long long double_sortable(double d) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(d);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

// ---- exact 128-bit fixed-point sums of doubles (52 fractional bits) --------------------------------
struct U128 { unsigned long long lo, hi; };
__device__ __forceinline__ U128 u128_zero() { U128 r; r.lo = 0; r.hi = 0; return r; }
__device__ __forceinline__ U128 u128_sub(U128 a, U128 b) {
  U128 r;
  r.lo = a.lo - b.lo;
  r.hi = a.hi - b.hi - (a.lo < b.lo ? 1ull : 0ull);
  return r;
}
__device__ __forceinline__ U128 u128_add(U128 a, U128 b) {
  U128 r;
  r.lo = a.lo + b.lo;
  r.hi = a.hi + b.hi + (r.lo < a.lo ? 1ull : 0ull);
  return r;
}
// DPP move of a 128-bit value: CTRL 0x110+n = row_shr:n, 0x142 = row_bcast:15, 0x143 = row_bcast:31;
// lanes without a source (or masked rows) receive zero
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ U128 u128_dpp(U128 v) {
  uint32_t a0 = (uint32_t)v.lo, a1 = (uint32_t)(v.lo >> 32), a2 = (uint32_t)v.hi, a3 = (uint32_t)(v.hi >> 32);
  a0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a0, CTRL, ROW_MASK, 0xF, true);
  a1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a1, CTRL, ROW_MASK, 0xF, true);
  a2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a2, CTRL, ROW_MASK, 0xF, true);
  a3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a3, CTRL, ROW_MASK, 0xF, true);
  U128 r;
  r.lo = (unsigned long long)a0 | ((unsigned long long)a1 << 32);
  r.hi = (unsigned long long)a2 | ((unsigned long long)a3 << 32);
  return r;
}
// 96-bit variant for the scan inside one wave: a term is < 2^88 (W < 2^9, x*y < 2^27, 52 fractional
// bits), so the sum of 64 of them is < 2^94 and the top dword of the 128-bit form stays zero
struct U96 { unsigned long long lo; uint32_t hi; };
__device__ __forceinline__ U96 u96_add(U96 a, U96 b) {
  U96 r;
  r.lo = a.lo + b.lo;
  r.hi = a.hi + b.hi + (r.lo < a.lo ? 1u : 0u);
  return r;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ U96 u96_dpp(U96 v) {
  uint32_t a0 = (uint32_t)v.lo, a1 = (uint32_t)(v.lo >> 32), a2 = v.hi;
  a0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a0, CTRL, ROW_MASK, 0xF, true);
  a1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a1, CTRL, ROW_MASK, 0xF, true);
  a2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a2, CTRL, ROW_MASK, 0xF, true);
  U96 r;
  r.lo = (unsigned long long)a0 | ((unsigned long long)a1 << 32);
  r.hi = a2;
  return r;
}
__device__ __forceinline__ U96 u96_of(U128 v) { U96 r; r.lo = v.lo; r.hi = (uint32_t)v.hi; return r; }
__device__ __forceinline__ U128 u128_of(U96 v) { U128 r; r.lo = v.lo; r.hi = (unsigned long long)v.hi; return r; }
// value * 2^52 of a double >= 1 (every moment term is: W >= 1, x,y >= 1) and < 2^64
__device__ __forceinline__ U128 exact_to_fixed(double t) {
  const unsigned long long bits = (unsigned long long)__double_as_longlong(t);
  const int exp = (int)((bits >> 52) & 0x7FF);
  const unsigned long long m = (bits & ((1ull << 52) - 1)) | (1ull << 52);
  const int shift = exp - 1023;
  U128 r;
  if (shift <= 0) { r.lo = shift < -63 ? 0 : (m >> (-shift)); r.hi = 0; return r; }  // t < 2 (shift 0) or never
  r.lo = m << shift;
  r.hi = m >> (64 - shift);
  return r;
}
// nearest-even rounding of v / 2^52 to double, v < 2^127.  Branch-free: normalise so that the leading
// one sits at bit 127, take the top 53 bits as the mantissa, the next bit as the half bit, the rest as
// sticky.  (v == 0 maps to 0.)
__device__ __forceinline__ double exact_from_fixed(U128 v) {
  const bool hi0 = v.hi == 0;
  const unsigned long long top = hi0 ? v.lo : v.hi;
  const int lz = (int)__clzll((long long)top);          // 0..63 (64 when top == 0)
  // n = v << (lz + (hi0 ? 64 : 0)), leading one at bit 127
  const unsigned long long nh = hi0 ? (v.lo << lz) : ((v.hi << lz) | (lz ? (v.lo >> (64 - lz)) : 0ull));
  const unsigned long long nl = hi0 ? 0ull : (v.lo << lz);
  const int p = (hi0 ? 63 : 127) - lz;                  // position of the leading one
  unsigned long long mant = nh >> 11;                   // 53 bits
  const bool halfbit = (nh >> 10) & 1ull;
  const bool sticky = ((nh & 0x3FFull) | nl) != 0ull;
  mant += (halfbit && (sticky || (mant & 1ull))) ? 1ull : 0ull;
  // a carry out of the mantissa (mant == 2^53) is absorbed by the exponent field arithmetic below
  const unsigned long long bits = ((unsigned long long)(1023 - 52 + p - 1) << 52) + mant;
  return top == 0 ? 0.0 : __longlong_as_double((long long)bits);
}

// ---- the same exact sums in two doubles (fast path of the moment sweep) ---------------------------------
// For images up to 2048 x 2048 working pixels every moment term t is a double with 1 <= t < 2^31 and a cluster
// holds fewer than 2^15 points, so every partial sum S is a multiple of 2^-52 below 2^46.  Such a sum is kept as
// (hi, lo): hi a multiple of 2^-6 (exact in a double below 2^47), lo the remainder.  A term splits exactly with
// hi = (t + C) - C, lo = t - hi, C = 1.5 * 2^46 (adding C rounds t to the 2^-6 grid; |lo| <= 2^-7).  Sums of up to
// 128 such terms stay exact component-wise (|sum lo| <= 1 is a multiple of 2^-52 below 2^53 ulps); across chunks and
// waves the running lo part is renormalised onto the grid after every addition (split_renorm).  The rounded prefix
// is the single IEEE addition hi + lo -- the correctly rounded value of the exact sum, i.e. bit for bit what
// exact_from_fixed produces from the 128-bit form.  Three instructions per term instead of a dozen, one instead
// of twenty-five per rounding.
#define AT_SPLIT_C 0x1.8p+46
struct D2 { double hi, lo; };
__device__ __forceinline__ D2 split_term(double t) {
  D2 r;
  r.hi = (t + AT_SPLIT_C) - AT_SPLIT_C;
  r.lo = t - r.hi;
  return r;
}
// (a.hi + b.hi, a.lo + b.lo) with the lo part brought back to |lo| <= 2^-7; requires |a.lo| <= 2^-7, |b.lo| <= 1
__device__ __forceinline__ D2 split_add_renorm(D2 a, D2 b) {
  const double l = a.lo + b.lo;
  const double c = (l + AT_SPLIT_C) - AT_SPLIT_C;
  D2 r;
  r.hi = (a.hi + b.hi) + c;
  r.lo = l - c;
  return r;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double f64_dpp(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, true);
  return __hiloint2double(hi, lo);
}
// inclusive wave scan of a double whose partial sums are exact (lanes without a source add +0.0)
__device__ __forceinline__ double wave_scan_f64(double v) {
  v += f64_dpp<0x111, 0xF>(v);
  v += f64_dpp<0x112, 0xF>(v);
  v += f64_dpp<0x114, 0xF>(v);
  v += f64_dpp<0x118, 0xF>(v);
  v += f64_dpp<0x142, 0xA>(v);
  v += f64_dpp<0x143, 0xC>(v);
  return v;
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// n / d with the reciprocal refinement hoisted: shared_recip is v_rcp_f64 + two Newton steps (the
// operations the compiler's f64 division expands to), div_by is quotient, exact residual, correction.
// Correctly rounded -- bit-identical to n / d -- for operands that need no v_div_scale rescaling; the
// line fit uses it with 1 <= d < 2^40 and |n| < 2^70 (checked on the device by the IEEE test, op 4).
__device__ __forceinline__ double shared_recip(double d) {
  double r0;
  asm("v_rcp_f64 %0, %1" : "=v"(r0) : "v"(d));
  const double e0 = __builtin_fma(-d, r0, 1.0);
  const double r1 = __builtin_fma(e0, r0, r0);
  const double e1 = __builtin_fma(-d, r1, 1.0);
  return __builtin_fma(e1, r1, r1);
}
__device__ __forceinline__ double div_by(double n, double d, double r) {
  const double q = n * r;
  const double e = __builtin_fma(-d, q, n);
  return __builtin_fma(e, r, q);
}

// Line parameters and errors from the moments of a run of points (the second half of every line fit).
__device__ __forceinline__ void fit_line_moments(double Mx, double My, double Mxx, double Mxy, double Myy, double W, int N,
                                                 double* lineparm, double* err, double* mse) {
  // Five IEEE divisions by the same W share one reciprocal refinement (shared_recip / div_by)
  const double rW = shared_recip(W);
  auto divW = [W, rW](double n) { return div_by(n, W, rW); };
  const double Ex = divW(Mx), Ey = divW(My);
  const double Cxx = divW(Mxx) - Ex * Ex, Cxy = divW(Mxy) - Ex * Ey, Cyy = divW(Myy) - Ey * Ey;
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
    // (the two divisions through one shared reciprocal refinement -- bit-identical, checked -- measured no gain)
    else { lineparm[2] = nx / length; lineparm[3] = ny / length; }
  }
  if (err) *err = N * eig_small;
  if (mse) *mse = eig_small;
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
  fit_line_moments(Mx, My, Mxx, Mxy, Myy, W, N, lineparm, err, mse);
}


// ---- sound early exit before the second walk of the moment sweep ----------------------------------------------------
// A quad needs four corner indices i0 < i1 < i2 < i3 of the sorted, duplicate-free point sequence such that each of
// the four arcs between them fits a line with mse <= max_line_fit_mse.  mse = lambda_min(S) / W with S the weighted
// scatter matrix of the arc and W its weight.  For point sets A within B: lambda_min(S_A) <= lambda_min(S_B) (the
// scatter of a union is the sum of the parts' scatters plus a positive semidefinite between-part term) and W_A <= W_B.
// Cut the sequence into FQ_XG groups of consecutive points; with the corners in groups a <= b <= c <= d, an arc between
// two cut groups contains every group strictly between them and lies within the groups from cut to cut, so
//   lambda_min(S of the groups strictly between) <= mse_limit * (weight of the groups from cut to cut)
// is NECESSARY for the arc.  If no a <= b <= c <= d passes that test on all four arcs (the fourth wraps around the end
// of the sequence), no corner choice can be admissible and the cluster ends at "no admissible corner choice" whatever
// the maxima turn out to be: the second walk, the windowed errors, the maxima and the corner search are skipped.
// The test never changes a result.  mse_limit carries an allowance for the rounding of the evaluation it stands in
// for (differences of once-rounded prefixes: a few ulps of the largest prefix over an arc weight >= 2; the float square
// root: 2^-23.5 of the arc's total variance, which the image diagonal bounds) and for its own rounding (2 %, plus 2^-44 of
// the cluster's total second moments for the order-dependent rounding of the sector sums and their prefix scan).
// On sigma-2 1080p frames 72 % of the points of clusters without an admissible corner choice sit in clusters this
// test rejects with 64 groups, 68 % with 32 (tools/early_exit_power.py, CPU restatement); the one-wave class (up to
// 768 points) gains nothing from it and does not run it.
#define FQ_XG 32
// The pre-sort sector test runs in k_fit_prefilter for the clusters above 2048 points.  (Inside k_fit_quads --
// kept for comparison builds, -DFQ_PRESORT_MIN_NT=256 -- the large classes paid for it with one or two workgroups per CU;
// the 128-thread class gains nothing from it either way: 7 % of its points rejected for 27 % of its cycles, its clusters
// are small against the 64 x 16 tiles the points arrive in, so the sector changes at almost every point.)
#ifndef FQ_PRESORT_MIN_NT
#define FQ_PRESORT_MIN_NT (1 << 20)
#endif
// Moments m = hi - lo (+ add) of the groups strictly between two cuts; returns false only if lambda_min of their scatter
// certainly exceeds thr * Wc.  Division- and root-free: with W = m5, A = W m2 - m0^2, B = W m3 - m0 m1, C = W m4 - m1^2 (W times
// the scatter matrix) and c = thr * Wc, both eigenvalues of the scatter exceed c exactly when (A - cW) + (C - cW) > 0 and
// (A - cW)(C - cW) - B^2 > 0.  Rounding: coordinates are below 2048 on this path (split_moments), so m2, m4 <= 2^22 W and
// each of A, B, C carries an absolute error below e = 4e-9 W^2 (eight roundings of terms <= W^2 2^22); the trace and the
// determinant are compared against the error bounds that follow from e, so a "false" is a proof.
__device__ __forceinline__ bool fq_arc_possible(const double* lo, const double* hi, const double* add, double Wc, double thr) {
  double m[6];
#pragma unroll
  for (int j = 0; j < 6; j++) m[j] = hi[j] - lo[j];
  if (add) {
#pragma unroll
    for (int j = 0; j < 6; j++) m[j] += add[j];
  }
  const double W = m[5];
  if (!(W > 0.25)) return true;   // no points in between (every weight is >= 1/2)
  const double cW = thr * Wc * W;
  const double a = (W * m[2] - m[0] * m[0]) - cW, d = (W * m[4] - m[1] * m[1]) - cW, B = W * m[3] - m[0] * m[1];
  const double e = 4.1e-9 * W * W;
  const double fa = fabs(a), fd = fabs(d), fB = fabs(B);
  const double tr = a + d, det = a * d - B * B;
  const double e1 = 2.0 * e + 0x1p-50 * (fa + fd);
  const double e2 = e * (fa + fd + 2.0 * fB) + 3.0 * e * e + 0x1p-50 * (fa * fd + fB * fB);
  return !(tr > e1 && det > e2);
}

// The whole test on a prefix array sP[(FQ_XG + 1)][ST] in LDS (entry g = sums over the groups before g; components 0..5 =
// the six moments with the scatter weights, component WI = the weight bound): true if some a <= b <= c <= d passes on all
// four arcs.  Contains workgroup barriers; every thread returns the same value.
template <int NT, int ST, int WI>
__device__ __forceinline__ bool fq_feasible(const double* sP, double mse_limit, int W, int H, uint32_t* s_okf, uint32_t* s_okw,
                                            int* s_feasible) {
  const int tid = threadIdx.x;
  // allowance for the rounding of the evaluation this test stands in for (see above)
  const double thr = mse_limit * 1.02 + 0.1 + 0x1p-44 * (sP[FQ_XG * ST + 2] + sP[FQ_XG * ST + 4]) +
                     1e-7 * 0.25 * ((double)W * (double)W + (double)H * (double)H);
#pragma unroll 1
  for (int pbase = 0; pbase < FQ_XG * FQ_XG; pbase += NT) {
    const int pi = pbase + tid, a = pi / FQ_XG, b = pi & (FQ_XG - 1);
    bool okf = true, okw = true;
    if (b >= a) {
      if (b >= a + 2) okf = fq_arc_possible(sP + (a + 1) * ST, sP + b * ST, nullptr, sP[(b + 1) * ST + WI] - sP[a * ST + WI], thr);
      __builtin_amdgcn_sched_barrier(0);   // (one evaluation's prefix loads at a time)
      okw = fq_arc_possible(sP + (b + 1) * ST, sP + FQ_XG * ST, sP + a * ST,
                            (sP[FQ_XG * ST + WI] - sP[b * ST + WI]) + sP[(a + 1) * ST + WI], thr);
    }
    // lanes 0..31 of a wave share a, lanes 32..63 the next one (FQ_XG == 32)
    const unsigned long long mf = __ballot(okf && b >= a), mw = __ballot(okw && b >= a);
    if ((tid & 31) == 0) {
      s_okf[a] = (uint32_t)(mf >> (tid & 32));
      s_okw[a] = (uint32_t)(mw >> (tid & 32));
    }
  }
  __syncthreads();
  if (tid < 64) {
    // cut groups a <= b <= c <= d: forward arcs a->b, b->c, c->d and the wrap-around arc d->a must all be possible
    const uint32_t rowf = tid < FQ_XG ? s_okf[tid] : 0u, roww = tid < FQ_XG ? s_okw[tid] : 0u;
    uint32_t r2 = 0, r3 = 0;
#pragma unroll 8
    for (int b = 0; b < FQ_XG; b++) {
      const uint32_t rb = (uint32_t)__builtin_amdgcn_readlane((int)rowf, b);
      r2 |= ((rowf >> b) & 1u) ? rb : 0u;
    }
#pragma unroll 8
    for (int c = 0; c < FQ_XG; c++) {
      const uint32_t rc = (uint32_t)__builtin_amdgcn_readlane((int)rowf, c);
      r3 |= ((r2 >> c) & 1u) ? rc : 0u;
    }
    const unsigned long long any = __ballot((r3 & roww) != 0u);
    if (tid == 0) *s_feasible = any != 0ull;
  }
  __syncthreads();
  return *s_feasible != 0;
}
// The same test BEFORE the sort, on angular sectors instead of groups of sorted points: a sector is an interval of the
// slope key, so its points are consecutive in the sorted order whatever that order turns out to be.  Duplicates are
// still present here (a half-pixel location can occur twice in a cluster, and only when both coordinates are odd: the
// (1,1) point of one pixel and the (-1,1) point of its right neighbour); both copies carry the same weight and fall
// into the same sector, so entering every odd-odd point with HALF its weight keeps the scatter of any union of
// sectors below that of its duplicate-free points, line by line; the weight bound takes every point in full.
// 32 sectors, eight per quadrant band of the slope key, cut at multiples of 11.25 degrees.
__device__ __forceinline__ int fq_sector(float slope) {
  const int qi = (slope >= 0.0f ? 1 : 0) + (slope >= 65536.0f ? 1 : 0) + (slope >= 131072.0f ? 1 : 0);
  const float r = slope - (float)(qi - 1) * 65536.0f;   // monotone in slope inside a band
  const int sub = (r >= 0.19891237f ? 1 : 0) + (r >= 0.41421357f ? 1 : 0) + (r >= 0.66817864f ? 1 : 0) + (r >= 1.0f ? 1 : 0) +
                  (r >= 1.4966058f ? 1 : 0) + (r >= 2.4142137f ? 1 : 0) + (r >= 5.0273395f ? 1 : 0);
  return qi * 8 + sub;
}
// 64 sectors (sixteen per band, cut at multiples of 5.625 degrees); sector s of fq_sector is the union of 2s and 2s + 1
__device__ __forceinline__ int fq_sector64(float slope) {
  const int qi = (slope >= 0.0f ? 1 : 0) + (slope >= 65536.0f ? 1 : 0) + (slope >= 131072.0f ? 1 : 0);
  const float r = slope - (float)(qi - 1) * 65536.0f;
  const int sub = (r >= 0.09849140f ? 1 : 0) + (r >= 0.19891237f ? 1 : 0) + (r >= 0.30334668f ? 1 : 0) + (r >= 0.41421357f ? 1 : 0) +
                  (r >= 0.53451114f ? 1 : 0) + (r >= 0.66817864f ? 1 : 0) + (r >= 0.82067879f ? 1 : 0) + (r >= 1.0f ? 1 : 0) +
                  (r >= 1.2185035f ? 1 : 0) + (r >= 1.4966058f ? 1 : 0) + (r >= 1.8708684f ? 1 : 0) + (r >= 2.4142137f ? 1 : 0) +
                  (r >= 3.2965582f ? 1 : 0) + (r >= 5.0273395f ? 1 : 0) + (r >= 10.153170f ? 1 : 0);
  return qi * 16 + sub;
}
// fq_feasible over 64 groups for a one-wave workgroup: lane b evaluates the arcs (a, b) of one cut group a per trip, so a
// ballot is row a of the relation; the path search keeps one 64-bit row per lane.
template <int ST, int WI, int MUT_EXTRA = 0>   // (MUT_EXTRA: tools_hooks.h, 0 in the product build)
__device__ __forceinline__ bool fq_feasible64(const double* sP, double mse_limit, int W, int H) {
  const int b = (int)threadIdx.x;   // 0..63
  const double thr = mse_limit * 1.02 + 0.1 + 0x1p-44 * (sP[64 * ST + 2] + sP[64 * ST + 4]) +
                     1e-7 * 0.25 * ((double)W * (double)W + (double)H * (double)H);
  unsigned long long rowf = 0, roww = 0;   // lane a keeps row a
#pragma unroll 1
  for (int a = 0; a < 64; a++) {
    bool okf = true, okw = true;
    if (b >= a) {
      if (b >= a + 2) okf = fq_arc_possible(sP + (a + 1 - MUT_EXTRA) * ST, sP + (b + MUT_EXTRA) * ST, nullptr, sP[(b + 1) * ST + WI] - sP[a * ST + WI], thr);
      __builtin_amdgcn_sched_barrier(0);
      okw = fq_arc_possible(sP + (b + 1) * ST, sP + 64 * ST, sP + a * ST, (sP[64 * ST + WI] - sP[b * ST + WI]) + sP[(a + 1) * ST + WI], thr);
    }
    const unsigned long long mf = __ballot(okf && b >= a), mw = __ballot(okw && b >= a);
    if (b == a) { rowf = mf; roww = mw; }
  }
  auto row_of = [&](int l) {
    return (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)rowf, l) |
           ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(rowf >> 32), l) << 32);
  };
  unsigned long long r2 = 0, r3 = 0;
#pragma unroll 8
  for (int l = 0; l < 64; l++) r2 |= ((rowf >> l) & 1ull) ? row_of(l) : 0ull;
#pragma unroll 8
  for (int l = 0; l < 64; l++) r3 |= ((r2 >> l) & 1ull) ? row_of(l) : 0ull;
  return __ballot((r3 & roww) != 0ull) != 0ull;
}

// Sort keys are stored so that their order as IEEE doubles equals the wanted unsigned order: a
// compare-exchange is then v_min_f64 + v_max_f64 (two instructions instead of a 64-bit compare and four
// selects).  u >= 2^63 -> positive double with the same lower 63 bits; u < 2^63 -> ~u, a negative double
// whose magnitude falls as u grows.  The upper word of u is float_sortable(slope) <= 0xFF800000 and
// >= 0x387FFFFF (+ bias), so no encoded key has an all-ones exponent (no NaN, no infinity); +infinity is
// the pad.
// A bias of 2^20 on the upper word keeps the slope +0.0 (upper word 0x80000000) away from the denormal
// doubles, so the result does not depend on the denormal mode of the min/max instructions.
#define AT_KEY_BIAS 0x0010000000000000ull
__device__ __forceinline__ unsigned long long key_enc(unsigned long long u) {
  u += AT_KEY_BIAS;
  return (u >> 63) ? (u ^ 0x8000000000000000ull) : ~u;
}
__device__ __forceinline__ unsigned long long key_dec(unsigned long long k) {
  return ((k >> 63) ? ~k : (k | 0x8000000000000000ull)) - AT_KEY_BIAS;
}
#define AT_KEY_PAD 0x7FF0000000000000ull

// Up to three network steps per pass: a thread owns a group of 2^r elements that is closed under r
// consecutive steps (a flip followed by half-cleaners, or half-cleaners only), so every element is
// loaded and stored once per r steps and the workgroup synchronises a third as often.  Indices >= n
// behave as +infinity and are never written.
template <int R>
__device__ __forceinline__ void bitonic_group(unsigned long long (&v)[8], bool flip_first) {
  constexpr int M = 1 << R;
  auto ce = [](unsigned long long& lo, unsigned long long& hi) {
    const double a = __longlong_as_double((long long)lo), b = __longlong_as_double((long long)hi);
    double mn, mx;
    asm("v_min_f64 %0, %1, %2" : "=v"(mn) : "v"(a), "v"(b));
    asm("v_max_f64 %0, %1, %2" : "=v"(mx) : "v"(a), "v"(b));
    lo = (unsigned long long)__double_as_longlong(mn);
    hi = (unsigned long long)__double_as_longlong(mx);
  };
  int first_stride = M >> 1;
  if (flip_first) {
#pragma unroll
    for (int e = 0; e < M / 2; e++) ce(v[e], v[M - 1 - e]);
    first_stride = M >> 2;
  }
#pragma unroll
  for (int st = M >> 1; st >= 1; st >>= 1) {
    if (st > first_stride) continue;
#pragma unroll
    for (int e = 0; e < M; e++)
      if ((e & st) == 0) ce(v[e], v[e + st]);
  }
}

// PADDED: the array physically holds +infinity keys in [n, 2^lpow), so loads and stores need no bounds tests (a
// group that lies entirely in the pad region is skipped with one comparison: the pads never move, every exchange
// puts the smaller key at the lower index).
// SKEW: the array is the LDS key array, stored with one unused slot after every 32 keys (FQ_KP): a thread's 2^R keys
// lie 2^lsp apart, so without the skew the lanes of a wave -- consecutive groups -- would meet in 2^lsp of the 32
// eight-byte bank pairs whenever lsp < 5 (the last two passes of every merge level: 8- to 32-way conflicts, measured
// as 60 % of the kernel's LDS cycles).  The physical index of a group's e-th key is FQ_KP(first key) plus a uniform
// offset, because the keys of a group either share a 32-key row or lie whole rows apart.
#define FQ_KP(i) ((i) + ((i) >> 5))
template <int NT, int R, bool PADDED, bool SKEW, typename KeyPtr>
__device__ __forceinline__ void bitonic_pass_r(KeyPtr A, int n, int lpow, int lk, int s) {
  // steps s .. s+R-1 of merge level lk (step 0 = flip of 2^lk blocks, step t = half-cleaner of stride 2^(lk-1-t))
  constexpr int M = 1 << R;
  const unsigned long long INF = AT_KEY_PAD;
  const int ngroups = (1 << lpow) >> R;
  // spacing of the group's elements: 2^lsp, where the last step's stride is 2^(lk-1-(s+R-1)) = 2^lsp
  const int lsp = lk - s - R;
  const int spm = (1 << lsp) - 1;
  int soff[M];   // uniform: physical offset of the e-th key of a run that starts in a row's first 2^lsp slots
#pragma unroll
  for (int e = 0; e < M; e++) soff[e] = SKEW ? FQ_KP(e << lsp) : (e << lsp);
  for (int g = threadIdx.x; g < ngroups; g += NT) {
    int idx[M];    // physical indices
    int lidx[M];   // logical indices (bounds tests of the unpadded variant; dead code otherwise)
    if (s == 0) {
      // lower side ascending, upper side mirrored: positions 0..M/2-1 are x_e, M/2..M-1 are y_(M/2-1-e')
      const int blk = g >> lsp, off = g & spm;
      const int lo = (blk << lk) + off, hi = (blk << lk) + (1 << lk) - 1 - off;
      const int hi0 = hi - ((M / 2 - 1) << lsp);   // lowest key of the upper side
      const int plo = SKEW ? FQ_KP(lo) : lo, phi0 = SKEW ? FQ_KP(hi0) : hi0;
#pragma unroll
      for (int e = 0; e < M / 2; e++) {
        idx[e] = plo + soff[e]; idx[M - 1 - e] = phi0 + soff[M / 2 - 1 - e];
        lidx[e] = lo + (e << lsp); lidx[M - 1 - e] = hi - (e << lsp);
      }
    } else {
      const int base = ((g >> lsp) << (lsp + R)) + (g & spm);
      const int pbase = SKEW ? FQ_KP(base) : base;
#pragma unroll
      for (int e = 0; e < M; e++) { idx[e] = pbase + soff[e]; lidx[e] = base + (e << lsp); }
    }
    if (lidx[1] >= n) continue;  // at most one real element: nothing to exchange
    unsigned long long v[8];
    if (PADDED) {
#pragma unroll
      for (int e = 0; e < M; e++) v[e] = A[idx[e]];
      bitonic_group<R>(v, s == 0);
#pragma unroll
      for (int e = 0; e < M; e++) A[idx[e]] = v[e];
    } else {
#pragma unroll
      for (int e = 0; e < M; e++) v[e] = lidx[e] < n ? A[idx[e]] : INF;
      bitonic_group<R>(v, s == 0);
#pragma unroll
      for (int e = 0; e < M; e++)
        if (lidx[e] < n) A[idx[e]] = v[e];
    }
  }
  __syncthreads();
}

// merge levels 1..3 in one pass: every thread sorts 8 consecutive elements in registers
template <int NT, bool PADDED, bool SKEW, typename KeyPtr>
__device__ __forceinline__ void bitonic_first8(KeyPtr A, int n, int lpow) {
  auto ce = [](unsigned long long& lo, unsigned long long& hi) {
    const double a = __longlong_as_double((long long)lo), b = __longlong_as_double((long long)hi);
    double mn, mx;
    asm("v_min_f64 %0, %1, %2" : "=v"(mn) : "v"(a), "v"(b));
    asm("v_max_f64 %0, %1, %2" : "=v"(mx) : "v"(a), "v"(b));
    lo = (unsigned long long)__double_as_longlong(mn);
    hi = (unsigned long long)__double_as_longlong(mx);
  };
  const int ngroups = (1 << lpow) >> 3;
  for (int g = threadIdx.x; g < ngroups; g += NT) {
    const int base = g << 3;
    if (base + 1 >= n) continue;
    const int pb = SKEW ? FQ_KP(base) : base;   // eight keys of one 32-key row
    unsigned long long v[8];
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = (PADDED || base + e < n) ? A[pb + e] : AT_KEY_PAD;
#pragma unroll
    for (int lk = 1; lk <= 3; lk++) {
      const int k = 1 << lk;
#pragma unroll
      for (int e = 0; e < 8; e++) {   // flip inside blocks of k
        const int off = e & (k - 1);
        if (off < k / 2) ce(v[e], v[(e & ~(k - 1)) + k - 1 - off]);
      }
#pragma unroll
      for (int st = k >> 2; st >= 1; st >>= 1) {
#pragma unroll
        for (int e = 0; e < 8; e++)
          if ((e & st) == 0) ce(v[e], v[e + st]);
      }
    }
#pragma unroll
    for (int e = 0; e < 8; e++)
      if (PADDED || base + e < n) A[pb + e] = v[e];
  }
  __syncthreads();
}

template <int NT, bool PADDED, bool SKEW, typename KeyPtr>
__device__ __forceinline__ void bitonic_sort_block2(KeyPtr A, int n, int lpow) {
  int lk0 = 1;
  if (lpow >= 3) { bitonic_first8<NT, PADDED, SKEW>(A, n, lpow); lk0 = 4; }
  for (int lk = lk0; lk <= lpow; lk++) {
    int s = 0;
    while (s < lk) {
      const int left = lk - s;
      if (left >= 3) { bitonic_pass_r<NT, 3, PADDED, SKEW>(A, n, lpow, lk, s); s += 3; }
      else if (left == 2) { bitonic_pass_r<NT, 2, PADDED, SKEW>(A, n, lpow, lk, s); s += 2; }
      else { bitonic_pass_r<NT, 1, PADDED, SKEW>(A, n, lpow, lk, s); s += 1; }
    }
  }
}

// ---- the same network for a cluster that fits the REGISTERS of one wave (64 K keys, K = 1, 2, 4) ---------------------------
// The LDS passes above give every thread a group of eight keys, so a 128-key sort keeps 16 lanes of the wave busy and costs
// as many instructions as a 512-key sort: the sort was a fixed ~780 wave-instructions per cluster of the one-wave class,
// whose clusters mostly have 65 ... 256 points.  Here lane l holds keys l K ... l K + K - 1 for the whole sort: steps of
// stride < K exchange registers, the others fetch the partner lane's key over the LDS crossbar (ds_bpermute, no memory)
// and keep the minimum or the maximum by the lane's side of the exchange.  Same network, same +infinity pads, same result.
__device__ __forceinline__ unsigned long long fq_fetch64(unsigned long long v, int byte_addr) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(byte_addr, (int)(uint32_t)v);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(byte_addr, (int)(uint32_t)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long fq_minmax64(unsigned long long mine, unsigned long long other, bool keep_min) {
  const double a = __longlong_as_double((long long)mine), b = __longlong_as_double((long long)other);
  double mn, mx;
  asm("v_min_f64 %0, %1, %2" : "=v"(mn) : "v"(a), "v"(b));
  asm("v_max_f64 %0, %1, %2" : "=v"(mx) : "v"(a), "v"(b));
  return (unsigned long long)__double_as_longlong(keep_min ? mn : mx);
}
template <int K>
__device__ __forceinline__ void fq_wave_sort_regs(unsigned long long (&v)[K]) {   // lane l holds keys l K .. l K + K - 1 of 64 K (pads included)
  constexpr int LK = K == 1 ? 0 : K == 2 ? 1 : K == 4 ? 2 : 3;
  const int lane = lane_id();
  auto ce = [](unsigned long long& lo, unsigned long long& hi) {
    const double a = __longlong_as_double((long long)lo), b = __longlong_as_double((long long)hi);
    double mn, mx;
    asm("v_min_f64 %0, %1, %2" : "=v"(mn) : "v"(a), "v"(b));
    asm("v_max_f64 %0, %1, %2" : "=v"(mx) : "v"(a), "v"(b));
    lo = (unsigned long long)__double_as_longlong(mn);
    hi = (unsigned long long)__double_as_longlong(mx);
  };
#pragma unroll
  for (int lk = 1; lk <= LK + 6; lk++) {
    // flip inside blocks of 2^lk: index i meets i ^ (2^lk - 1)
    if ((1 << lk) <= K) {
#pragma unroll
      for (int j = 0; j < K; j++) {
        const int j2 = j ^ ((1 << lk) - 1);
        if (j < j2) ce(v[j], v[j2]);
      }
    } else {
      const int lx = (1 << (lk - LK)) - 1;                      // partner lane = lane ^ lx, partner register = K - 1 - j
      const bool keep_min = (lane & (1 << (lk - LK - 1))) == 0;  // the lower of the two lanes
      const int addr = (lane ^ lx) << 2;
      unsigned long long o[K];
#pragma unroll
      for (int j = 0; j < K; j++) o[j] = fq_fetch64(v[K - 1 - j], addr);
#pragma unroll
      for (int j = 0; j < K; j++) v[j] = fq_minmax64(v[j], o[j], keep_min);
    }
    // half-cleaners of stride 2^(lk-2) ... 1
#pragma unroll
    for (int S = (1 << lk) >> 2; S >= 1; S >>= 1) {
      if (S < K) {
#pragma unroll
        for (int j = 0; j < K; j++)
          if ((j & S) == 0) ce(v[j], v[j + S]);
      } else {
        const int lx = S / K;
        const bool keep_min = (lane & lx) == 0;
        const int addr = (lane ^ lx) << 2;
#pragma unroll
        for (int j = 0; j < K; j++) v[j] = fq_minmax64(v[j], fq_fetch64(v[j], addr), keep_min);
      }
    }
  }
}
template <int K>
__device__ __forceinline__ void fq_wave_sort(unsigned long long* A) {   // A: skewed LDS key array holding 64 K keys (pads included)
  const int lane = lane_id();
  unsigned long long v[K];
#pragma unroll
  for (int j = 0; j < K; j++) v[j] = A[FQ_KP(lane * K + j)];
  fq_wave_sort_regs<K>(v);
#pragma unroll
  for (int j = 0; j < K; j++) A[FQ_KP(lane * K + j)] = v[j];
}

// lexicographic list of the 4-subsets of {0..9}: entry t = {m0,m1,m2,m3} packed 4 bits each, built at
// compile time (computing it in the kernel prologue cost ~2600 instructions per workgroup)
struct ComboTable { uint16_t v[210]; };
constexpr ComboTable make_combo_table() {
  ComboTable t{};
  int c = 0;
  for (int m0 = 0; m0 < 7; m0++)
    for (int m1 = m0 + 1; m1 < 8; m1++)
      for (int m2 = m1 + 1; m2 < 9; m2++)
        for (int m3 = m2 + 1; m3 < 10; m3++) t.v[c++] = (uint16_t)(m0 | (m1 << 4) | (m2 << 8) | (m3 << 12));
  return t;
}
__device__ const ComboTable g_combo_table = make_combo_table();
// the 45 index pairs a < b < 10 in triangular order, packed (a << 4) | b
struct PairTable { uint8_t v[45]; };
constexpr PairTable make_pair_table() {
  PairTable t{};
  int c = 0;
  for (int a = 0; a < 9; a++)
    for (int b = a + 1; b < 10; b++) t.v[c++] = (uint8_t)((a << 4) | b);
  return t;
}
__device__ const PairTable g_pair_table = make_pair_table();
// the same pair from its triangular index in registers (a table load from global memory sat in every cluster's
// critical path: one more round trip of latency before the segment fits could start)
__device__ __forceinline__ int fq_pair_of(int t) {
  const int a = (t >= 9) + (t >= 17) + (t >= 24) + (t >= 30) + (t >= 35) + (t >= 39) + (t >= 42) + (t >= 44);
  const int b = t - ((a * (19 - a)) >> 1) + a + 1;
  return (a << 4) | b;
}

// ---- the tail of the fit, shared by k_fit_quads and k_fit_small: segment fits between the selected maxima and the search over
// the corner choices ---------------------------------------------------------------------------------------------------
// Table layout (doubles) inside the dead key / moment region: for the 45 index pairs a < b < 10 (triangular index FQ_PIDX) the
// error, mse and four line parameters of the forward segment a -> b; error and mse of the wrap-around segment b -> a and its
// six moments (its line is needed for the one chosen corner set only: k_quad_finish fits it from the moments the candidate
// record carries); behind them the 2 m + 1 staged rows of cumulative moments the fits read.
#define FQ_PIDX(a, b) ((((a) * (19 - (a))) >> 1) + (b) - (a) - 1)   // a < b < 10 -> 0..44
#define FQT_FERR 0
#define FQT_FMSE 45
#define FQT_FEX 90
#define FQT_FEY 135
#define FQT_FNX 180
#define FQT_FNY 225
#define FQT_WERR 270
#define FQT_WMSE 315
#define FQT_WMOM 360     // [45][4]: Mx, My, Mxx, Mxy
#define FQT_WMOM2 540    // [45][2]: Myy, W
#define FQT_ROWS 630     // [21][6]
#define FQT_DOUBLES 756

// the 210 corner choices m0 < m1 < m2 < m3 < 10 in lexicographic order: pair-table indices of the segments m0->m1, m1->m2,
// m2->m3 (forward) and m0..m3 (wrap-around), 6 bits each, and m3 in bits 24..27
struct ComboPairs { uint32_t v[210]; };
constexpr ComboPairs make_combo_pairs() {
  ComboPairs t{};
  int c = 0;
  for (int m0 = 0; m0 < 7; m0++)
    for (int m1 = m0 + 1; m1 < 8; m1++)
      for (int m2 = m1 + 1; m2 < 9; m2++)
        for (int m3 = m2 + 1; m3 < 10; m3++)
          t.v[c++] = (uint32_t)FQ_PIDX(m0, m1) | ((uint32_t)FQ_PIDX(m1, m2) << 6) | ((uint32_t)FQ_PIDX(m2, m3) << 12) |
                     ((uint32_t)FQ_PIDX(m0, m3) << 18) | ((uint32_t)m3 << 24);
  return t;
}
__device__ const ComboPairs g_combo_pairs = make_combo_pairs();


// One segment fit: pair-table entry t (a < b), forward (dir 0: a -> b, with line parameters) or around the end (dir 1: b -> a,
// error and mse only).  s_maxidx: the m selected maxima in ascending order; rows: FQT_ROWS region.
__device__ __forceinline__ void fq_segment_fit(double* s_tab, const int* s_maxidx, int m, int szd, int t, int dir) {
  const int pr = fq_pair_of(t), a = pr >> 4, b = pr & 15;   // a < b
  if (b >= m) return;
  const double* s_rows = s_tab + FQT_ROWS;
  if (dir == 0) {
    // fit_line_dev(lf, szd, i_a, i_b) with i_a < i_b -- row at b, minus the row before a unless a is point 0
    double e, ms, lp[4];
    const double* rb = s_rows + b * 6;
    double Mx = rb[0], My = rb[1], Mxx = rb[2], Mxy = rb[3], Myy = rb[4], W = rb[5];
    if (s_maxidx[a] > 0) {
      const double* ra = s_rows + (m + a) * 6;
      Mx -= ra[0]; My -= ra[1]; Mxx -= ra[2]; Mxy -= ra[3]; Myy -= ra[4]; W -= ra[5];
    }
    fit_line_moments(Mx, My, Mxx, Mxy, Myy, W, s_maxidx[b] - s_maxidx[a] + 1, lp, &e, &ms);
    s_tab[FQT_FERR + t] = e; s_tab[FQT_FMSE + t] = ms; s_tab[FQT_FEX + t] = lp[0]; s_tab[FQT_FEY + t] = lp[1];
    s_tab[FQT_FNX + t] = lp[2]; s_tab[FQT_FNY + t] = lp[3];
  } else {
    // fit_line_dev(lf, szd, i_b, i_a) with i_b > i_a -- (last row - row before b) + row at a
    double e, ms;
    const double* re = s_rows + 2 * m * 6;
    const double* rp = s_rows + (m + b) * 6;
    const double* ra = s_rows + a * 6;
    double Mx = re[0] - rp[0], My = re[1] - rp[1], Mxx = re[2] - rp[2], Mxy = re[3] - rp[3], Myy = re[4] - rp[4], W = re[5] - rp[5];
    Mx += ra[0]; My += ra[1]; Mxx += ra[2]; Mxy += ra[3]; Myy += ra[4]; W += ra[5];
    fit_line_moments(Mx, My, Mxx, Mxy, Myy, W, szd - s_maxidx[b] + s_maxidx[a] + 1, nullptr, &e, &ms);
    s_tab[FQT_WERR + t] = e; s_tab[FQT_WMSE + t] = ms;
    s_tab[FQT_WMOM + t * 4 + 0] = Mx; s_tab[FQT_WMOM + t * 4 + 1] = My; s_tab[FQT_WMOM + t * 4 + 2] = Mxx; s_tab[FQT_WMOM + t * 4 + 3] = Mxy;
    s_tab[FQT_WMOM2 + t * 2 + 0] = Myy; s_tab[FQT_WMOM2 + t * 2 + 1] = W;
  }
}

// Best of the C(m, 4) corner choices and the candidate record, ONE wave (lane = 0..63), no workgroup barriers inside.  A
// choice is looked at in two steps: the four mse bits of its segments (one bit per pair-table entry, from two ballots) and only
// then the normals' dot product and the error sum.  Equal errors resolve to the smaller combination index (the CPU loop order).
__device__ __forceinline__ void fq_corner_search(const double* s_tab, const uint32_t* s_cpairs, int m, int szd, int lane, const DetParams& P,
                                                 FitCand* __restrict__ cands_all, FrameCounters* __restrict__ counters, int frame,
                                                 unsigned long long cl_key, int q_reversed) {
  // segments whose mse passes (a NaN passes, as in the serial comparison)
  unsigned long long fok, wok;
  {
    bool f = false, w = false;
    if (lane < 45 && (fq_pair_of(lane) & 15) < m) {
      f = !(s_tab[FQT_FMSE + lane] > P.max_line_fit_mse);
      w = !(s_tab[FQT_WMSE + lane] > P.max_line_fit_mse);
    }
    fok = __ballot(f); wok = __ballot(w);
  }
  double best_err = (double)HUGE_VALF;
  int best_t = 1 << 30;
  for (int t = lane; t < 210; t += 64) {
    const uint32_t cp = s_cpairs[t];
    const int p01 = cp & 63, p12 = (cp >> 6) & 63, p23 = (cp >> 12) & 63, p03 = (cp >> 18) & 63, q3 = (int)(cp >> 24);
    const bool pass = q3 < m && ((fok >> p01) & 1ull) && ((fok >> p12) & 1ull) && ((fok >> p23) & 1ull) && ((wok >> p03) & 1ull);
    if (pass) {
      const double dotn = s_tab[FQT_FNX + p01] * s_tab[FQT_FNX + p12] + s_tab[FQT_FNY + p01] * s_tab[FQT_FNY + p12];
      if (!(fabs(dotn) > P.cos_critical_rad)) {
        const double e = s_tab[FQT_FERR + p01] + s_tab[FQT_FERR + p12] + s_tab[FQT_FERR + p23] + s_tab[FQT_WERR + p03];
        if (e < best_err) { best_err = e; best_t = t; }
      }
    }
  }
  const unsigned long long mykey = ~double_sortable(best_err + 0.0);
  const unsigned long long topkey = wave_max_u64(mykey);
  const int bt = wave_min_i(mykey == topkey ? best_t : (1 << 30));
  if (bt == (1 << 30)) return;
  {
    // the winning error is the decoded top key (double_sortable is invertible)
    const unsigned long long sk = ~topkey;
    const unsigned long long bits = (sk >> 63) ? (sk & 0x7FFFFFFFFFFFFFFFull) : ~sk;
    const double bev = __longlong_as_double((long long)bits);
    if (!((bev != (double)HUGE_VALF) && (bev / szd < P.max_line_fit_mse))) return;
  }
  // The lines of the best choice are table entries (the same fits, bit for bit): segments q0->q1, q1->q2, q2->q3 forward; of
  // q3->q0 around the end the record takes the six moments.  Intersections, area and angle checks run in k_quad_finish, one
  // thread per candidate, instead of here on a handful of lanes.
  const uint32_t cp = s_cpairs[bt];
  const int p01 = cp & 63, p12 = (cp >> 6) & 63, p23 = (cp >> 12) & 63, p03 = (cp >> 18) & 63;
  uint32_t ci = 0;
  if (lane == 0) ci = atomicAdd(&counters[frame].ncand, 1u);
  ci = (uint32_t)__builtin_amdgcn_readfirstlane((int)ci);
  if (ci < P.cand_cap) {
    FitCand* const cd = cands_all + (size_t)frame * P.cand_cap + ci;
    if (lane < 3) {
      const int pi = lane == 0 ? p01 : lane == 1 ? p12 : p23;
      cd->line[lane][0] = s_tab[FQT_FEX + pi]; cd->line[lane][1] = s_tab[FQT_FEY + pi];
      cd->line[lane][2] = s_tab[FQT_FNX + pi]; cd->line[lane][3] = s_tab[FQT_FNY + pi];
    } else if (lane == 3) {
      cd->line[3][0] = s_tab[FQT_WMOM + p03 * 4 + 0]; cd->line[3][1] = s_tab[FQT_WMOM + p03 * 4 + 1];
      cd->line[3][2] = s_tab[FQT_WMOM + p03 * 4 + 2]; cd->line[3][3] = s_tab[FQT_WMOM + p03 * 4 + 3];
      cd->wm[0] = s_tab[FQT_WMOM2 + p03 * 2 + 0]; cd->wm[1] = s_tab[FQT_WMOM2 + p03 * 2 + 1];
    } else if (lane == 4) {
      cd->key = cl_key;
      cd->reversed_border = q_reversed;
      cd->wrap_is_moments = 1;
    }
  } else if (lane == 0) {
    atomicOr(&counters[frame].flags, AT_FLAG_CANDS);   // (internal: the host grows the list and repeats, or reports 0x8)
  }
}

// Dynamic LDS layout: [0, 8*sort_cap) slope keys (later: raw/smoothed errors, then maxima candidates) |
// FQ_TABLE_DOUBLES doubles for the group prefixes of the early-exit test (none in the one-wave class).  Clusters with size in (size_lo, size_hi] are processed by this
// launch; those above sort_cap (only possible in the last class) sort in global scratch.
FQ_TIMELINE_GLOBALS   // (tools_hooks.h: nothing in the product build)
template <int NT, bool SPLIT>
#ifndef FQ_EPT
#define FQ_EPT(NT) ((NT) >= 256 ? 2 : 1)   // elements per lane in the moment sweep
#endif
#define FQ_SEL_REGS 8          // maxima candidates per lane held in registers during the top-10 selection
#define FQ_SMOOTH_REGS_OF(NT) ((NT) >= 1024 ? 8 : 16)   // smoothed errors per thread kept in registers (clusters up to that many x threads)
#define FQ_TABLE_DOUBLES ((FQ_XG + 1) * 7)   // prefixes over FQ_XG groups, up to seven sums each
// bytes of the key array region: the skewed keys, and at least the twelve pair tables + the 21 staged prefix rows that take the region over later
#define FQ_KEY_BYTES(sort_cap) ((size_t)FQ_KP(sort_cap) * 8 > (size_t)(756 * 8) ? (size_t)FQ_KP(sort_cap) * 8 : (size_t)(756 * 8))   /* 756 = FQT_DOUBLES */
#ifndef FQ_WPE_64
#define FQ_WPE_64 4
#define FQ_WPE_128 4
#define FQ_WPE_256 4
#endif
// persistent workgroups per CU of the two small classes (they fill FQ_WPE waves per SIMD when alone on a CU)
#define FQ_NT_BIG 1024   // threads of the largest class: one workgroup fills a CU (16 waves, 4 per SIMD)
#ifndef FQ_GRID_64
#define FQ_GRID_64 16
#define FQ_GRID_128 8
#endif
// (the body of k_fit_quads)
__device__ __forceinline__ void fit_quads_body(const FrameDesc* __restrict__ frames, const uint8_t* __restrict__ gray_all,
                                                   const uint32_t* __restrict__ pts_all, const ClusterRec* __restrict__ clusters_all,
                                                   const uint32_t* __restrict__ work, const uint32_t* __restrict__ work_n, uint32_t work_cap,
                                                   uint32_t* __restrict__ work_cursor, double* __restrict__ lf_scratch,
                                                   unsigned long long* __restrict__ keys_scratch, double* __restrict__ errs_scratch,
                                                   FitCand* __restrict__ cands_all, FrameCounters* __restrict__ counters,
                                                   unsigned long long* __restrict__ prof, int sort_cap, int slot_cap,
                                                   int pop, DetParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fq_smem[];
  unsigned long long* skeys = reinterpret_cast<unsigned long long*>(fq_smem);
  double* chunk = reinterpret_cast<double*>(fq_smem + FQ_KEY_BYTES(sort_cap));   // (the key array is skewed); prefixes of the early-exit tests
  // Twelve tables over the 45 index pairs a < b < 10 (triangular index FQ_PIDX): error, mse and the four line parameters
  // of the forward segment a -> b and of the wrap-around segment b -> a.  They live in the key array, which is dead once
  // the maxima are selected (4320 bytes: FQ_KEY_BYTES keeps the region at least that large).
  double* const s_tab = reinterpret_cast<double*>(fq_smem);   // FQT_* layout
  constexpr int NW = NT / 64;
  __shared__ long long s_dot[NW][3];
  __shared__ int s_box[NW][4];
  __shared__ double s_remval[12];
  __shared__ int s_remidx[12];
  __shared__ int s_ncand;
  __shared__ int s_maxidx[16];
  __shared__ int s_nkept;
  __shared__ uint32_t s_cpairs[210];
  __shared__ U128 s_wtot[NW * 6];   // [wave][moment]: wave totals of the current chunk
  __shared__ U128 s_woff[NW * 6];   // [wave][moment]: offset every lane of the wave adds
  __shared__ int s_wcnt[NW];
  __shared__ int s_coff[NW + 1];    // kept-point offsets per wave, [NW] = chunk total
  __shared__ U128 s_carry[12];   // [chunk parity][moment]: running totals up to the chunk
  __shared__ uint32_t s_item;
  __shared__ uint32_t s_okf[FQ_XG], s_okw[FQ_XG];   // early exit: admissible forward / wrap-around arcs per cut group
  __shared__ int s_feasible;

  const int tid = threadIdx.x;
  const int W = P.W, H = P.H;
  const uint32_t nwork = min(*work_n, work_cap);
  // scratch slot of this workgroup: cumulative moments of the cluster in flight (and, for clusters that do
  // not fit the LDS key array, their sort keys and two error arrays)
  // (keeping the moments of the small classes in LDS instead was measured slower: 22.0 vs 19.7 ms, the LDS footprint
  // halves the resident waves)
  double* const lf = lf_scratch + (size_t)blockIdx.x * slot_cap * 6;
  unsigned long long* const gkeys = keys_scratch ? keys_scratch + (size_t)blockIdx.x * slot_cap : nullptr;
  double* const gerrs_a = errs_scratch ? errs_scratch + (size_t)blockIdx.x * slot_cap * 2 : nullptr;
  double* const gerrs_b = errs_scratch ? gerrs_a + slot_cap : nullptr;

  for (int t = tid; t < 210; t += NT) s_cpairs[t] = g_combo_pairs.v[t];

  FQ_HOOKS_DECL   // measurement hooks (FQ_TICK / FQ_COUNT / FQ_STOP_AT / FQ_TL_*): tools_hooks.h, nothing in the product build

  // Work is popped `pop` clusters at a time: one device-scope atomic on a single word saturates near 90
  // returns per microsecond (MI355X_MICROARCH.md, "dequeue"), which a one-cluster pop of the small classes
  // (0.7 M clusters per 256-frame submission) would hit.
  // The FIRST item of a workgroup is its own index -- no atomic: a one-frame submission has about as many clusters as the
  // persistent grid has workgroups, and 4096 workgroups popping their first item from one cursor word (about 90 returns per
  // microsecond) put the last cluster's start some 40 us behind the first's.  The cursor hands out the items from gridDim.x on.
  uint32_t next_item = blockIdx.x, chunk_left = 1;   // uniform
  for (;;) {
    __syncthreads();   // the previous cluster's LDS use (and s_item) is finished in every wave
    if (chunk_left == 0) {
      if (tid == 0) s_item = gridDim.x + atomicAdd(work_cursor, (uint32_t)pop);
      __syncthreads();
      next_item = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_item);   // uniform: keeps everything derived from it in SGPRs
      chunk_left = (uint32_t)pop;
    }
    const uint32_t item = next_item++;
    chunk_left--;
    FQ_TL_NEXT_CLUSTER()
    if (item >= nwork) break;
    const uint32_t wi = (uint32_t)__builtin_amdgcn_readfirstlane((int)work[item]);
    const int frame = (int)(wi >> P.wshift);
    const FrameDesc fd = frames[frame];
    const uint8_t* gray = (P.decimate > 1) ? gray_all + (size_t)frame * P.H * P.WS : fd.img;
    const int gpitch = (P.decimate > 1) ? P.WS : (int)fd.pitch;
    const __attribute__((address_space(1))) uint8_t* const ggray = (const __attribute__((address_space(1))) uint8_t*)gray;   // global, not generic
    const ClusterRec cl = clusters_all[(size_t)frame * P.ccap + (wi & ((1u << P.wshift) - 1u))];
    const int sz = (int)cl.count;
    FQ_TL_SIZE(sz)
    if (sz < 24 || sz > slot_cap) continue;   // (the work list only holds clusters of this class)
    FQ_TICK(0)
    const uint32_t* pts = pts_all + (size_t)frame * P.pcap + cl.start;

    // ---- bbox and exact gradient dot -----------------------------------------------------------
    int xmin = 1 << 30, xmax = -1, ymin = 1 << 30, ymax = -1;
    long long sxg = 0, sgx = 0, sgy = 0;
    // (four independent loads per trip: one load per trip left the loop waiting a full memory latency per 64 points.
    // A lane without a point in some slot repeats its first point with a zero gradient, which changes neither the box
    // nor the sums.)
    for (int i = tid; i < sz; i += 4 * NT) {
      uint32_t pq[4];
#pragma unroll
      for (int u = 0; u < 4; u++) pq[u] = pts[min(i + u * NT, sz - 1)];
      // gradient signs (-1, 0, 1) summed in 32 bits per trip (4 x 2^15 at most) with 24-bit multiplies, scaled by 255 once per
      // trip: as 64-bit multiply-adds per point this loop spent a third of its cycles in two quarter-rate instructions a point
      int t_xg = 0, t_gx = 0, t_gy = 0;
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint32_t p = (i + u * NT < sz) ? pq[u] : ((pq[0] & ~15u) | 5u);
        const int x = (int)(p >> 18), y = (int)((p >> 4) & 0x3FFF);
        const int gx = (int)((p >> 2) & 3) - 1, gy = (int)(p & 3) - 1;
        xmin = min(xmin, x); xmax = max(xmax, x); ymin = min(ymin, y); ymax = max(ymax, y);
        t_xg += __mul24(x, gx) + __mul24(y, gy);
        t_gx += gx; t_gy += gy;
      }
      sxg += (long long)(t_xg * 255); sgx += t_gx * 255; sgy += t_gy * 255;
    }
    // seven reductions, one barrier: every wave reduces on the DPP network and parks its results
    xmin = wave_min_i(xmin); xmax = wave_max_i(xmax); ymin = wave_min_i(ymin); ymax = wave_max_i(ymax);
    sxg = wave_sum_ll(sxg); sgx = wave_sum_ll(sgx); sgy = wave_sum_ll(sgy);
    if (NW > 1) {
      const int wv = tid >> 6;
      if (lane_id() == 0) {
        s_box[wv][0] = xmin; s_box[wv][1] = xmax; s_box[wv][2] = ymin; s_box[wv][3] = ymax;
        s_dot[wv][0] = sxg; s_dot[wv][1] = sgx; s_dot[wv][2] = sgy;
      }
      __syncthreads();
      xmin = s_box[0][0]; xmax = s_box[0][1]; ymin = s_box[0][2]; ymax = s_box[0][3];
      sxg = s_dot[0][0]; sgx = s_dot[0][1]; sgy = s_dot[0][2];
      // (not unrolled for the 16-wave instance: the unrolled loads cost it 34 spilled registers, and a kernel that uses
      // scratch started 0.13 ms late)
      constexpr int kUnrollWaves = NW > 8 ? 1 : NW;
#pragma unroll kUnrollWaves
      for (int w = 1; w < NW; w++) {
        xmin = min(xmin, s_box[w][0]); xmax = max(xmax, s_box[w][1]); ymin = min(ymin, s_box[w][2]); ymax = max(ymax, s_box[w][3]);
        sxg += s_dot[w][0]; sgx += s_dot[w][1]; sgy += s_dot[w][2];
      }
    }
    if ((xmax - xmin) * (ymax - ymin) < P.min_tag_width) continue;
    const double cxd = (xmin + xmax) * 0.5 + 0.05118, cyd = (ymin + ymax) * 0.5 + -0.028581;
    const double dot = (double)sxg - cxd * (double)sgx - cyd * (double)sgy;
    const int q_reversed = dot < 0;
    if (!P.reversed_border && q_reversed) continue;
    if (!P.normal_border && !q_reversed) continue;
    FQ_TICK(1)
    FQ_STOP_AT(1)

    // ---- slope keys + sort -----------------------------------------------------------------------
    const float cx = (float)cxd, cy = (float)cyd;
    // only the largest class (FQ_NT_BIG threads) can meet clusters beyond its LDS key array: every smaller instance
    // addresses LDS unconditionally (no generic-address loads)
    const bool in_lds = NT < FQ_NT_BIG || sz <= sort_cap;
    for (int i4 = tid; i4 < sz; i4 += 4 * NT) {
      uint32_t pq[4];
#pragma unroll
      for (int u = 0; u < 4; u++) pq[u] = pts[min(i4 + u * NT, sz - 1)];
#pragma unroll
      for (int u = 0; u < 4; u++) {
      const int i = i4 + u * NT;
      if (i >= sz) break;
      const uint32_t p = pq[u];
      const int x = (int)(p >> 18), y = (int)((p >> 4) & 0x3FFF);
      float dx = (float)x - cx, dy = (float)y - cy;
      float quadrant;
      if (dy > 0) quadrant = (dx > 0) ? 65536.0f : 131072.0f;
      else quadrant = (dx > 0) ? 0.0f : -65536.0f;
      if (dy < 0) { dy = -dy; dx = -dx; }
      if (dx < 0) { float tmp = dx; dx = dy; dy = -tmp; }
      const float slope = quadrant + __fdiv_rn(dy, dx);
      const unsigned long long key = key_enc(((unsigned long long)float_sortable(slope) << 32) | ((unsigned long long)y << 18) |
                                             ((unsigned long long)x << 4) | (unsigned long long)(p & 15u));
      if (in_lds) skeys[FQ_KP(i)] = key; else gkeys[i] = key;
      }
    }
    int lpow = NT == 64 ? 6 : 0;   // (the one-wave class sorts at least 64 keys: one per lane of the register sort)
    while ((1 << lpow) < sz) lpow++;
    // when the LDS array holds the next power of two, the tail is filled with +infinity keys and the network runs
    // without per-element bounds tests
    const bool padded = in_lds && (1 << lpow) <= sort_cap;
    if (padded)
      for (int i = sz + tid; i < (1 << lpow); i += NT) skeys[FQ_KP(i)] = AT_KEY_PAD;
    constexpr bool kPresort = FQ_SOUND_EXIT_PRESORT && SPLIT && NT >= FQ_PRESORT_MIN_NT;   // (tools_hooks.h: 1 in the product build)
    if constexpr (kPresort)
      for (int t = tid; t < (FQ_XG + 1) * 7; t += NT) chunk[t] = 0.0;   // sector sums (pair-table region: free until the maxima exist)
    __syncthreads();
    if constexpr (kPresort) {
      // ---- sound early exit before the sort (see fq_sector above) -------------------------------------------------
      // Every lane walks a run of consecutive keys of the (still unsorted) list -- points arrive tile by tile, so a run
      // mostly stays inside one sector -- and keeps the sector's seven sums in registers; they go to the sector's LDS
      // entry (seven f64 atomics) only when the sector changes.
      double* const sB = chunk;   // [(FQ_XG + 1)][7]: entry s + 1 = sums of sector s, then prefix sums
      {
        int E = (sz + NT - 1) / NT;
        if (!in_lds && E >= 8) E |= 1;
        if (in_lds && (E & 31) == 31) E++;
        const int i0 = tid * E, i1 = min(sz, i0 + E);
        double a[7];
        int cur = -1;
        auto flush = [&]() {
#pragma unroll
          for (int j = 0; j < 7; j++) __hip_atomic_fetch_add(&sB[(cur + 1) * 7 + j], a[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        for (int i = i0; i < i1; i++) {
          const unsigned long long key = key_dec(in_lds ? skeys[FQ_KP(i)] : gkeys[i]);
          const uint32_t fs = (uint32_t)(key >> 32);
          const float slope = __uint_as_float((fs & 0x80000000u) ? (fs & 0x7FFFFFFFu) : ~fs);   // inverse of float_sortable
          const int sec = fq_sector(slope);
          const uint32_t px = (uint32_t)((key >> 4) & 0x3FFF), py = (uint32_t)((key >> 18) & 0x3FFF);
          const double x = (int)(px + 1) * .5, y = (int)(py + 1) * .5;
          const int ix = (int)((px + 1) >> 1), iy = (int)((py + 1) >> 1);
          uint32_t G = 0;
          if (((unsigned)(ix - 1) < (unsigned)(W - 2)) & ((unsigned)(iy - 1) < (unsigned)(H - 2))) {
            const uint32_t o = __umul24((uint32_t)iy, (uint32_t)gpitch) + (uint32_t)ix;   // (rows below 2^14, pitches below 2^24: check_images)
            const int g_r = ggray[o + 1], g_l = ggray[o - 1], g_d = ggray[o + (uint32_t)gpitch], g_u = ggray[o - (uint32_t)gpitch];
            const int grad_x = g_r - g_l, grad_y = g_d - g_u;
            G = (uint32_t)(grad_x * grad_x + grad_y * grad_y);
          }
          const double Wt = sqrt_u18(G) + 1;
          const double wl = (px & py & 1u) ? 0.5 * Wt : Wt;
          if (sec != cur) {
            if (cur >= 0) flush();
            cur = sec;
#pragma unroll
            for (int j = 0; j < 7; j++) a[j] = 0.0;
          }
          const double wx = wl * x, wy = wl * y;
          a[0] += wx; a[1] += wy; a[2] += wx * x; a[3] += wx * y; a[4] += wy * y; a[5] += wl; a[6] += Wt;
        }
        if (cur >= 0) flush();
      }
      __syncthreads();
      if (tid < 7) {
        double run = 0.0;
        for (int sct = 1; sct <= FQ_XG; sct++) { run += sB[sct * 7 + tid]; sB[sct * 7 + tid] = run; }
      }
      __syncthreads();
      const bool feas_pre = fq_feasible<NT, 7, 6>(sB, P.max_line_fit_mse, W, H, s_okf, s_okw, &s_feasible);
      FQ_TICK(3)
      FQ_COUNT(0, sz)
      if (!feas_pre) { FQ_COUNT(1, sz) continue; }
    }
    // (a wave's own LDS accesses are ordered: the one-wave class needs no barrier around the register sort)
#ifndef FQ_REGSORT_MAX_LPOW
#define FQ_REGSORT_MAX_LPOW 8
#endif
    if (NT == 64 && padded && lpow <= FQ_REGSORT_MAX_LPOW) {
      if (lpow <= 6) fq_wave_sort<1>(skeys); else if (lpow == 7) fq_wave_sort<2>(skeys); else fq_wave_sort<4>(skeys);
    }
    else if (padded) bitonic_sort_block2<NT, true, true>(skeys, sz, lpow);
    else if (in_lds) bitonic_sort_block2<NT, false, true>(skeys, sz, lpow);
    else bitonic_sort_block2<NT, false, false>(gkeys, sz, lpow);
    FQ_TICK(2)
    FQ_STOP_AT(2)

    // ---- duplicate removal + weighted moment terms + exact cumulative sums, one sweep ---------------
    // Duplicate points (same half-pixel location, adjacent after the sort) contribute nothing and get no
    // output slot: the slot of a kept point is the running count of kept points, scanned together with
    // the moments.  Each term (a double >= 1) is widened to value*2^52 in a 128-bit integer, where
    // addition is exact and associative; the inclusive scan runs over the whole workgroup and every
    // prefix is rounded to nearest-even double once -- the same definition the CPU oracle uses,
    // independent of order.  Wave totals are double-buffered by chunk parity, so a chunk costs one
    // barrier; the running carries are double-buffered in LDS the same way (one wave: registers).
    int szd;
    if constexpr (SPLIT) {
      // Fast path (see split_term above): the same exact prefix sums carried as two doubles per moment, and ONE scan
      // per cluster instead of one per 64-point chunk.  Every lane owns E consecutive points of the sorted order:
      //   walk 1  terms of the lane's points, summed into lane totals (exact); the key slot of a point is rewritten
      //           with what walk 2 needs (kept flag, coordinates, squared gradient);
      //   scan    lane totals over the wave (DPP) and the waves (LDS), kept-point counts alongside;
      //   walk 2  every lane replays its points from its exclusive offset and rounds each prefix once.
      // (The lanes' key slots lie E keys apart; the skew of the key array -- one slot per 32 keys -- spreads any stride
      // over the banks, so E needs no rounding to an odd number: in LDS key k of lane t sits at E*t + k + ((E*t + k) >> 5).
      // Only the global-scratch path of the largest class keeps the odd stride, for its 64-byte segments.)
      const int lane = lane_id(), wv = tid >> 6;
      int E = (sz + NT - 1) / NT;
      if (!in_lds && E >= 8) E |= 1;
      if (in_lds && (E & 31) == 31) E++;   // the one stride the skew folds back onto a single bank pair (31 + 31/32 keys)
      const int i0 = tid * E, i1 = min(sz, i0 + E);
      D2* const sd_wtot = reinterpret_cast<D2*>(s_wtot);
      D2* const sd_woff = reinterpret_cast<D2*>(s_woff);
      unsigned long long prev = 0;
      if (i0 > 0 && i0 < sz) prev = key_dec(in_lds ? skeys[FQ_KP(i0 - 1)] : gkeys[i0 - 1]);
      if (NW > 1) __syncthreads();   // every lane holds its predecessor key before any slot is rewritten
      D2 acc[6];
#pragma unroll
      for (int j = 0; j < 6; j++) { acc[j].hi = 0; acc[j].lo = 0; }
      int kept = 0;
      for (int i = i0; i < i1; i++) {
        const unsigned long long key = key_dec(in_lds ? skeys[FQ_KP(i)] : gkeys[i]);
        const bool keep = (i == 0) || ((key >> 4) != (prev >> 4));
        prev = key;
        const uint32_t px = (uint32_t)((key >> 4) & 0x3FFF), py = (uint32_t)((key >> 18) & 0x3FFF);
        uint32_t G = 0;   // squared gradient magnitude; 0 also stands for "no gradient taken" (weight 1 either way)
        if (keep) {
          // x = px / 2 + 1/2 exactly; its integer part and the image offset stay in 32-bit integers (a frame spans
          // less than 2^31 bytes, checked at submission), the four neighbours are loaded before the first is used
          const double x = (int)(px + 1) * .5, y = (int)(py + 1) * .5;
          const int ix = (int)((px + 1) >> 1), iy = (int)((py + 1) >> 1);
          if (((unsigned)(ix - 1) < (unsigned)(W - 2)) & ((unsigned)(iy - 1) < (unsigned)(H - 2))) {
            const uint32_t o = __umul24((uint32_t)iy, (uint32_t)gpitch) + (uint32_t)ix;   // (rows below 2^14, pitches below 2^24: check_images)
            const int g_r = ggray[o + 1], g_l = ggray[o - 1], g_d = ggray[o + (uint32_t)gpitch], g_u = ggray[o - (uint32_t)gpitch];
            const int grad_x = g_r - g_l, grad_y = g_d - g_u;
            G = (uint32_t)(grad_x * grad_x + grad_y * grad_y);
          }
          const double Wt = sqrt_u18(G) + 1;
          const double tt[6] = {Wt * x, Wt * y, Wt * x * x, Wt * x * y, Wt * y * y, Wt};
#pragma unroll
          for (int j = 0; j < 6; j++) {
            const D2 t = split_term(tt[j]);
            acc[j].hi += t.hi; acc[j].lo += t.lo;
          }
          kept++;
        }
        const unsigned long long stash = ((unsigned long long)(keep ? 1u : 0u) << 63) | ((unsigned long long)G << 28) |
                                         ((unsigned long long)py << 14) | (unsigned long long)px;
        if (in_lds) skeys[FQ_KP(i)] = stash; else gkeys[i] = stash;
      }
      // lane totals onto the grid (|lo| <= 2^-7), then the inclusive scans
      D2 incl[6];
#pragma unroll
      for (int j = 0; j < 6; j++) {
        const double c = (acc[j].lo + AT_SPLIT_C) - AT_SPLIT_C;
        acc[j].hi += c; acc[j].lo -= c;
        incl[j].hi = wave_scan_f64(acc[j].hi);
        incl[j].lo = wave_scan_f64(acc[j].lo);
      }
      int kincl = kept;
#define OP(C, M) kincl += __builtin_amdgcn_update_dpp(0, kincl, C, M, 0xF, true);
      AT_DPP_STEPS(OP)
#undef OP
      D2 off[6];
      int pos = kincl - kept;
      if (NW > 1) {
        if (lane == 63) {
#pragma unroll
          for (int j = 0; j < 6; j++) sd_wtot[wv * 6 + j] = incl[j];
          s_wcnt[wv] = kincl;
        }
        __syncthreads();
        if (tid < NW * 6) {
          const int ww = tid / 6, j = tid - ww * 6;
          D2 run; run.hi = 0; run.lo = 0;
          for (int w2 = 0; w2 < ww; w2++) run = split_add_renorm(run, sd_wtot[w2 * 6 + j]);
          sd_woff[tid] = run;
        } else if (tid < NW * 7) {
          const int ww = tid - NW * 6;
          int run = 0;
          for (int w2 = 0; w2 < ww; w2++) run += s_wcnt[w2];
          s_coff[ww] = run;
          if (ww == NW - 1) s_coff[NW] = run + s_wcnt[ww];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 6; j++) {
          const D2 o = sd_woff[wv * 6 + j];
          off[j].hi = (incl[j].hi - acc[j].hi) + o.hi;
          off[j].lo = (incl[j].lo - acc[j].lo) + o.lo;
        }
        pos += s_coff[wv];
        szd = s_coff[NW];
        // ---- sound early exit (see fq_arc_possible above): no four cut points can give four admissible arcs ------
        // The kept points of the sorted order are cut into FQ_XG groups (runs of NT / FQ_XG lanes); the inclusive
        // moment prefix at the end of every group goes to LDS (the pair-table region is free until the maxima exist).
        if constexpr (FQ_SOUND_EXIT_AFTER_WALK1) {   // (tools_hooks.h: 1 in the product build)
          constexpr int LPG = NT / FQ_XG;
          double* const sP = chunk;   // [(FQ_XG + 1)][6]
          if (tid < 6) sP[tid] = 0.0;
          if ((tid & (LPG - 1)) == LPG - 1) {
            const int g = tid / LPG;
#pragma unroll
            for (int j = 0; j < 6; j++) sP[(g + 1) * 6 + j] = (off[j].hi + acc[j].hi) + (off[j].lo + acc[j].lo);
          }
          __syncthreads();
          if (!fq_feasible<NT, 6, 5>(sP, P.max_line_fit_mse, W, H, s_okf, s_okw, &s_feasible)) { FQ_COUNT(2, sz) continue; }
        }
      } else {
#pragma unroll
        for (int j = 0; j < 6; j++) { off[j].hi = incl[j].hi - acc[j].hi; off[j].lo = incl[j].lo - acc[j].lo; }
        szd = __builtin_amdgcn_readlane(kincl, 63);
      }
      // walk 2: replay the lane's points from its exclusive offset
      // (the row pointer advances with the kept points: formed from the position per point it was a 64-bit multiply-add a trip)
      double* o = lf + (size_t)pos * 6;
      for (int i = i0; i < i1; i++) {
        const unsigned long long st = in_lds ? skeys[FQ_KP(i)] : gkeys[i];
        if (st >> 63) {
          const double x = (int)(((uint32_t)st & 0x3FFFu) + 1u) * .5, y = (int)(((uint32_t)(st >> 14) & 0x3FFFu) + 1u) * .5;
          const double Wt = sqrt_u18((uint32_t)(st >> 28) & 0x3FFFFu) + 1;
          const double tt[6] = {Wt * x, Wt * y, Wt * x * x, Wt * x * y, Wt * y * y, Wt};
#pragma unroll
          for (int j = 0; j < 6; j++) {
            const D2 t = split_term(tt[j]);
            off[j].hi += t.hi; off[j].lo += t.lo;
            o[j] = off[j].hi + off[j].lo;
          }
          o += 6;
        }
      }
    } else {
      // EPT consecutive elements per lane: the DPP scan, the barrier and the carry traffic are paid once
      // per EPT elements; the lane's own elements are separated again after the scan by subtraction
      constexpr int EPT = FQ_EPT(NT);
      U128 carry[6];
#pragma unroll
      for (int j = 0; j < 6; j++) carry[j] = u128_zero();
      int cnt_carry = 0;
      int par = 0;
      const int lane = lane_id(), wv = tid >> 6;
      const unsigned long long lt_mask = (1ull << lane) - 1ull;
      if (NW > 1 && tid < 6) s_carry[tid] = u128_zero();   // visible after the first chunk's barrier
      for (int base = 0; base < sz; base += NT * EPT, par ^= 1) {
        U96 v[6];          // sum of the lane's elements, then the in-wave inclusive prefix (96 bits suffice)
        U96 t1[6];         // terms of the lane's second element (EPT == 2)
#pragma unroll
        for (int j = 0; j < 6; j++) { v[j].lo = 0; v[j].hi = 0; t1[j].lo = 0; t1[j].hi = 0; }
        bool keep[EPT];
        unsigned long long prev_key = 0;
#pragma unroll
        for (int e = 0; e < EPT; e++) {
          const int i = base + tid * EPT + e;
          keep[e] = false;
          if (i < sz) {
            const unsigned long long key = key_dec(in_lds ? skeys[FQ_KP(i)] : gkeys[i]);
            const unsigned long long prev = (e > 0) ? prev_key : ((i > 0) ? key_dec(in_lds ? skeys[FQ_KP(i - 1)] : gkeys[i - 1]) : ~key);
            prev_key = key;
            keep[e] = (i == 0) || ((key >> 4) != (prev >> 4));
            if (keep[e]) {
              const int px = (int)((key >> 4) & 0x3FFF), py = (int)((key >> 18) & 0x3FFF);
              const double x = px * .5 + 0.5, y = py * .5 + 0.5;
              const int ix = (int)x, iy = (int)y;
              double Wt = 1;
              if (ix > 0 && ix + 1 < W && iy > 0 && iy + 1 < H) {
                const int grad_x = (int)gray[(size_t)iy * gpitch + ix + 1] - (int)gray[(size_t)iy * gpitch + ix - 1];
                const int grad_y = (int)gray[(size_t)(iy + 1) * gpitch + ix] - (int)gray[(size_t)(iy - 1) * gpitch + ix];
                Wt = __dsqrt_rn((double)(grad_x * grad_x + grad_y * grad_y)) + 1;
              }
              U96 t[6];
              t[0] = u96_of(exact_to_fixed(Wt * x));
              t[1] = u96_of(exact_to_fixed(Wt * y));
              t[2] = u96_of(exact_to_fixed(Wt * x * x));
              t[3] = u96_of(exact_to_fixed(Wt * x * y));
              t[4] = u96_of(exact_to_fixed(Wt * y * y));
              t[5] = u96_of(exact_to_fixed(Wt));
#pragma unroll
              for (int j = 0; j < 6; j++) {
                v[j] = u96_add(v[j], t[j]);
                if (e == 1) t1[j] = t[j];
              }
            }
          }
        }
        unsigned long long kmask[EPT];
        int before = 0, wcount = 0;
#pragma unroll
        for (int e = 0; e < EPT; e++) {
          kmask[e] = __ballot(keep[e]);
          before += (int)__popcll(kmask[e] & lt_mask);
          wcount += (int)__popcll(kmask[e]);
        }
        // wave-inclusive scan with DPP lane shifts: out-of-range sources read as zero, so no per-step
        // select is needed
#pragma unroll
        for (int j = 0; j < 6; j++) {
          v[j] = u96_add(v[j], u96_dpp<0x111, 0xF>(v[j]));
          v[j] = u96_add(v[j], u96_dpp<0x112, 0xF>(v[j]));
          v[j] = u96_add(v[j], u96_dpp<0x114, 0xF>(v[j]));
          v[j] = u96_add(v[j], u96_dpp<0x118, 0xF>(v[j]));
          v[j] = u96_add(v[j], u96_dpp<0x142, 0xA>(v[j]));
          v[j] = u96_add(v[j], u96_dpp<0x143, 0xC>(v[j]));
        }
        int pos = cnt_carry + before;   // slot of the lane's first kept element
        U128 w[6];   // workgroup-wide prefix including the lane's last element
        if (NW > 1) {
          // two-level combine: the wave totals go to LDS, NW*6 threads turn them into per-wave offsets
          // (running total of the previous chunks + the waves before), everybody adds its wave's offset.
          // Per-thread work and register use do not grow with the number of waves.
          if (lane == 63) {
#pragma unroll
            for (int j = 0; j < 6; j++) s_wtot[wv * 6 + j] = u128_of(v[j]);
            s_wcnt[wv] = wcount;
          }
          __syncthreads();
          if (tid < NW * 6) {
            const int ww = tid / 6, j = tid - ww * 6;
            U128 run = s_carry[par * 6 + j];
            for (int w2 = 0; w2 < ww; w2++) run = u128_add(run, s_wtot[w2 * 6 + j]);
            s_woff[tid] = run;
            if (ww == NW - 1) s_carry[(par ^ 1) * 6 + j] = u128_add(run, s_wtot[ww * 6 + j]);
          } else if (tid < NW * 7) {
            const int ww = tid - NW * 6;
            int run = 0;
            for (int w2 = 0; w2 < ww; w2++) run += s_wcnt[w2];
            s_coff[ww] = run;
            if (ww == NW - 1) s_coff[NW] = run + s_wcnt[ww];
          }
          __syncthreads();
#pragma unroll
          for (int j = 0; j < 6; j++) w[j] = u128_add(u128_of(v[j]), s_woff[wv * 6 + j]);
          pos += s_coff[wv];
          cnt_carry += s_coff[NW];
        } else {
#pragma unroll
          for (int j = 0; j < 6; j++) {
            w[j] = u128_add(u128_of(v[j]), carry[j]);
            U128 t;
            t.lo = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)w[j].lo, 63) |
                   ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(w[j].lo >> 32), 63) << 32);
            t.hi = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)w[j].hi, 63) |
                   ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(w[j].hi >> 32), 63) << 32);
            carry[j] = t;
          }
          cnt_carry += wcount;
        }
        if (EPT == 2) {
          // second element: prefix w; first element: w minus the second element's terms
          if (keep[1]) {
            double* o = lf + (size_t)(pos + (keep[0] ? 1 : 0)) * 6;
#pragma unroll
            for (int j = 0; j < 6; j++) o[j] = exact_from_fixed(w[j]);
          }
          if (keep[0]) {
            double* o = lf + (size_t)pos * 6;
#pragma unroll
            for (int j = 0; j < 6; j++) o[j] = exact_from_fixed(u128_sub(w[j], u128_of(t1[j])));
          }
        } else if (keep[0]) {
          double* o = lf + (size_t)pos * 6;
#pragma unroll
          for (int j = 0; j < 6; j++) o[j] = exact_from_fixed(w[j]);
        }
      }
      szd = cnt_carry;
    }
    __syncthreads();   // lf complete (read by other threads below); key array free
    if (szd < 24) continue;
    FQ_TICK(4)
    FQ_STOP_AT(4)

    // ---- windowed line-fit error, smoothing ------------------------------------------------------
    const int ksz = min(20, szd / 12);
    // raw errors: the key array is dead after the moment terms, so they live there (LDS); global scratch
    // for clusters that do not fit it
    double* ea = in_lds ? reinterpret_cast<double*>(skeys) : gerrs_a;
    // (indices wrap with compare/subtract: integer division by a run-time value costs ~40 instructions)
    // (two interleaved fits per trip for instruction-level parallelism measured no gain)
    {
      // The window of point i is rows (i - ksz - 1, i + ksz] of the cumulative moments, around the end for the first and last ksz
      // points.  fit_line_dev's three cases are one statement once the row before the window reads as zero where the window starts at
      // point 0 (b - 0.0 is b, bit for bit: every moment is positive) -- inside: b - a; around the end: (total - a) + b -- so a trip
      // is two row loads whose addresses depend on nothing but i, and the total row is loaded once per cluster instead of once per
      // wrap-around trip (9.71 -> 9.66 ms for the stage per 256 frames).  Holding the rows of trip k + 1 in registers under the fit of
      // trip k was measured too: 24 more live doubles took the 64- and 128-thread instances from 17 / 26 to 41 / 58 spilled registers
      // and the stage to 10.28 ms.
      double tot[6];
#pragma unroll
      for (int j = 0; j < 6; j++) tot[j] = lf[(size_t)(szd - 1) * 6 + j];
      for (int i = tid; i < szd; i += NT) {
        const int i0 = (i >= ksz) ? i - ksz : i - ksz + szd;
        const int i1 = (i + ksz < szd) ? i + ksz : i + ksz - szd;
        const double2* pb = reinterpret_cast<const double2*>(lf + (size_t)i1 * 6);
        const double2 b0 = pb[0], b1 = pb[1], b2 = pb[2];
        const double b[6] = {b0.x, b0.y, b1.x, b1.y, b2.x, b2.y};
        double a[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (i0 > 0) {
          const double2* pa = reinterpret_cast<const double2*>(lf + (size_t)(i0 - 1) * 6);
          const double2 a0 = pa[0], a1 = pa[1], a2 = pa[2];
          a[0] = a0.x; a[1] = a0.y; a[2] = a1.x; a[3] = a1.y; a[4] = a2.x; a[5] = a2.y;
        }
        const bool around = (i < ksz) || (i + ksz >= szd);
        double m[6];
#pragma unroll
        for (int j = 0; j < 6; j++) m[j] = around ? (tot[j] - a[j]) + b[j] : b[j] - a[j];
        double e;
        fit_line_moments(m[0], m[1], m[2], m[3], m[4], m[5], 2 * ksz + 1, nullptr, &e, nullptr);
        ea[i] = e;
      }
    }
    if (tid == 0) { s_ncand = 0; s_nkept = 0; }
    __syncthreads();
    const float f0 = 0x1.6c0504p-7f, f1 = 0x1.152aaap-3f, f2 = 0x1.368b3p-1f;
    const double F0 = (double)f0, F1 = (double)f1, F2 = (double)f2;
    auto wrap = [szd](int k) { return k < 0 ? k + szd : (k >= szd ? k - szd : k); };
    double* cand_val = in_lds ? reinterpret_cast<double*>(skeys) : ea;
    int* cand_idx = in_lds ? reinterpret_cast<int*>(skeys + (sort_cap >> 1)) : reinterpret_cast<int*>(ea + (szd >> 1) + 1);
    constexpr int FQ_SMOOTH_REGS = FQ_SMOOTH_REGS_OF(NT);
    if (in_lds && szd <= FQ_SMOOTH_REGS * NT) {
      // Up to FQ_SMOOTH_REGS points per thread: the smoothed errors are formed in registers, written back
      // over the raw ones, compared with their neighbours, and only then does the same LDS array take the
      // candidate list -- no second error array (LDS or global) at all.  The pointers below are derived
      // from the LDS array directly (not from the LDS-or-global selects above), so the compiler emits
      // ds_ instructions with LDS alignment rules instead of flat accesses it may widen to 16 bytes.
      double* const le = reinterpret_cast<double*>(skeys);
      double* const lcv = reinterpret_cast<double*>(skeys);
      int* const lci = reinterpret_cast<int*>(skeys + (sort_cap >> 1));
      double acc[FQ_SMOOTH_REGS];
#pragma unroll
      for (int it = 0; it < FQ_SMOOTH_REGS; it++) {
        const int i = tid + it * NT;
        acc[it] = 0;
        if (i < szd) {
          double a2 = 0;
          a2 += le[wrap(i - 3)] * F0;
          a2 += le[wrap(i - 2)] * F1;
          a2 += le[wrap(i - 1)] * F2;
          a2 += le[i] * 1.0;
          a2 += le[wrap(i + 1)] * F2;
          a2 += le[wrap(i + 2)] * F1;
          a2 += le[wrap(i + 3)] * F0;
          acc[it] = a2;
        }
      }
      __syncthreads();
#pragma unroll
      for (int it = 0; it < FQ_SMOOTH_REGS; it++) {
        const int i = tid + it * NT;
        if (i < szd) le[i] = acc[it];
      }
      __syncthreads();
      uint32_t mx = 0;
#pragma unroll
      for (int it = 0; it < FQ_SMOOTH_REGS; it++) {
        const int i = tid + it * NT;
        if (i < szd) {
          const double e = acc[it];
          if (e > le[i + 1 < szd ? i + 1 : 0] && e > le[i > 0 ? i - 1 : szd - 1]) mx |= 1u << it;
        }
      }
      __syncthreads();
#pragma unroll
      for (int it = 0; it < FQ_SMOOTH_REGS; it++) {
        if (it * NT >= szd) break;
        const bool is_max = (mx >> it) & 1u;
        const unsigned long long mm = __ballot(is_max);
        if (mm) {
          int kbase = 0;
          if (lane_id() == 0) kbase = atomicAdd(&s_ncand, (int)__popcll(mm));
          kbase = __builtin_amdgcn_readfirstlane(kbase);
          if (is_max) {
            const int k = kbase + (int)__popcll(mm & ((1ull << lane_id()) - 1ull));
            lcv[k] = acc[it];
            lci[k] = tid + it * NT;
          }
        }
      }
    } else {
      double* eb = gerrs_b;
      for (int i = tid; i < szd; i += NT) {
        double acc = 0;
        acc += ea[wrap(i - 3)] * F0;
        acc += ea[wrap(i - 2)] * F1;
        acc += ea[wrap(i - 1)] * F2;
        acc += ea[i] * 1.0;
        acc += ea[wrap(i + 1)] * F2;
        acc += ea[wrap(i + 2)] * F1;
        acc += ea[wrap(i + 3)] * F0;
        eb[i] = acc;
      }
      __syncthreads();
      for (int base = 0; base < szd; base += NT) {
        const int i = base + tid;
        double e = 0;
        bool is_max = false;
        if (i < szd) {
          e = eb[i];
          is_max = e > eb[i + 1 < szd ? i + 1 : 0] && e > eb[i > 0 ? i - 1 : szd - 1];
        }
        const unsigned long long mm = __ballot(is_max);
        if (mm) {
          int kbase = 0;
          if (lane_id() == 0) kbase = atomicAdd(&s_ncand, (int)__popcll(mm));
          kbase = __builtin_amdgcn_readfirstlane(kbase);
          if (is_max) {
            const int k = kbase + (int)__popcll(mm & ((1ull << lane_id()) - 1ull));
            cand_val[k] = e;
            cand_idx[k] = i;
          }
        }
      }
    }
    FQ_TICK(5)
    FQ_STOP_AT(5)
    __syncthreads();
    const int nmaxima = s_ncand;
    if (nmaxima < 4) continue;
    // Selection and ordering of at most max_nmaxima corners: wave 0 alone, no workgroup barriers inside.
    // When there are more candidates, the current largest is removed (max_nmaxima + 1) times; the last
    // removed value is the threshold (only the values matter, so ties may be broken arbitrarily).
    if (tid < 64) {
      const int lane = tid;
      if (nmaxima > P.max_nmaxima && nmaxima <= 64) {
        // one candidate per lane: rank every candidate among all of them (value descending, lane
        // ascending for equal values); rank max_nmaxima holds the threshold value
        const bool have = lane < nmaxima;
        const unsigned long long myk = have ? double_sortable(cand_val[lane] + 0.0) : 0ull;
        const int myi = have ? cand_idx[lane] : -1;
        int rank = 0;
        for (int l = 0; l < nmaxima; l++) {
          const unsigned long long ok =
              (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)myk, l) |
              ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(myk >> 32), l) << 32);
          rank += (ok > myk || (ok == myk && l < lane)) ? 1 : 0;
        }
        const unsigned long long tmask = __ballot(have && rank == P.max_nmaxima);
        const int tl = (int)__ffsll((long long)tmask) - 1;
        const unsigned long long tk = (unsigned long long)(uint32_t)__shfl((int)(uint32_t)myk, tl, 64) |
                                      ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(myk >> 32), tl, 64) << 32);
        const bool kept = have && rank < P.max_nmaxima && myk > tk;
        const unsigned long long kmask = __ballot(kept);
        if (kept) s_maxidx[__popcll(kmask & ((1ull << lane) - 1ull))] = myi;
        if (lane == 0) s_nkept = (int)__popcll(kmask);
        __threadfence_block();
      } else if (nmaxima > P.max_nmaxima && nmaxima <= 64 * FQ_SEL_REGS && P.max_nmaxima < 12) {
        // up to FQ_SEL_REGS candidates per lane, held in registers as order-preserving keys (0 = empty:
        // no finite double maps to 0): the arg-max rounds touch no memory until the winners are known
        unsigned long long key[FQ_SEL_REGS];
#pragma unroll
        for (int r = 0; r < FQ_SEL_REGS; r++) {
          const int k = lane + 64 * r;
          key[r] = (k < nmaxima) ? double_sortable(cand_val[k] + 0.0) : 0ull;
        }
        unsigned long long rem_key = 0;   // lane r (< 12) keeps the key and candidate index of round r
        int rem_k = 0;
        for (int round = 0; round <= P.max_nmaxima; round++) {
          unsigned long long bk = 0; int br = 0;
#pragma unroll
          for (int r = 0; r < FQ_SEL_REGS; r++)
            if (key[r] > bk) { bk = key[r]; br = r; }
          const unsigned long long mk = wave_max_u64(bk);
          const unsigned long long who = __ballot(bk == mk && bk != 0ull);
          const int src = (int)__ffsll((long long)who) - 1;
          const int wr = __shfl(br, src, 64);
          if (lane == src) {
#pragma unroll
            for (int r = 0; r < FQ_SEL_REGS; r++)
              if (r == wr) key[r] = 0ull;   // removed
          }
          if (lane == round) { rem_key = mk; rem_k = src + 64 * wr; }
        }
        // threshold = key of the last round; keep the earlier winners that are strictly larger
        const unsigned long long tk = (unsigned long long)(uint32_t)__shfl((int)(uint32_t)rem_key, P.max_nmaxima, 64) |
                                      ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(rem_key >> 32), P.max_nmaxima, 64) << 32);
        const bool kept = lane < P.max_nmaxima && rem_key > tk;
        const unsigned long long kmask = __ballot(kept);
        if (kept) s_maxidx[__popcll(kmask & ((1ull << lane) - 1ull))] = cand_idx[rem_k];
        if (lane == 0) s_nkept = (int)__popcll(kmask);
        __threadfence_block();
      } else if (nmaxima > P.max_nmaxima) {
        for (int round = 0; round <= P.max_nmaxima; round++) {
          unsigned long long bk = 0; int bi = -1;
          for (int k = lane; k < nmaxima; k += 64) {
            if (cand_idx[k] >= 0) {
              const unsigned long long kk = double_sortable(cand_val[k] + 0.0);
              if (bi < 0 || kk > bk) { bk = kk; bi = k; }
            }
          }
          const unsigned long long mk = wave_max_u64(bi < 0 ? 0ull : bk);
          const unsigned long long who = __ballot(bi >= 0 && bk == mk);
          const int ix = __shfl(bi, (int)__ffsll((long long)who) - 1, 64);
          if (lane == 0) {
            s_remval[round] = cand_val[ix];
            s_remidx[round] = cand_idx[ix];
            cand_idx[ix] = -1;  // removed
          }
          __threadfence_block();
        }
        if (lane == 0) {
          const double thresh = s_remval[P.max_nmaxima];
          int mm = 0;
          for (int r = 0; r < P.max_nmaxima; r++)
            if (s_remval[r] > thresh) s_maxidx[mm++] = s_remidx[r];
          s_nkept = mm;
        }
      } else if (lane == 0) {
        for (int k = 0; k < nmaxima; k++) s_maxidx[k] = cand_idx[k];
        s_nkept = nmaxima;
      }
      if (lane == 0) {  // ascending index order (<= 10 entries)
        const int mm = s_nkept;
        for (int a2 = 1; a2 < mm; a2++) {
          const int v = s_maxidx[a2];
          int b2 = a2 - 1;
          while (b2 >= 0 && s_maxidx[b2] > v) { s_maxidx[b2 + 1] = s_maxidx[b2]; b2--; }
          s_maxidx[b2 + 1] = v;
        }
      }
    }
    __syncthreads();
    const int m = s_nkept;
    if (m < 4) continue;
    FQ_TICK(6)
    FQ_STOP_AT(6)

    // ---- pairwise segment fits, then all corner quadruples ---------------------------------------
    // The 90 fits read only 2 m + 1 rows of the cumulative moments -- at every selected maximum, just before it, and the
    // last row -- so those rows are fetched ONCE (one lane per row, all loads in flight together) and parked in the key
    // region behind the pair tables; the fits then take them from LDS.  (Every fit used to start with its own three global
    // loads, round after round: the phase is ~15 us of the ~40 us every cluster of the one-wave class costs, mostly waiting.)
    double* const s_rows = s_tab + FQT_ROWS;   // [2 m + 1][6]: rows 0 .. m-1 at the maxima, m .. 2m-1 before them, 2m the last row
    for (int r = tid; r < 2 * m + 1; r += NT) {
      const int src = r < m ? s_maxidx[r] : r < 2 * m ? s_maxidx[r - m] - 1 : szd - 1;
      double row[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};   // (before point 0: nothing; the fit does not subtract it)
      if (src >= 0) {
        const double* g = lf + (size_t)src * 6;
#pragma unroll
        for (int j = 0; j < 6; j++) row[j] = g[j];
      }
#pragma unroll
      for (int j = 0; j < 6; j++) s_rows[r * 6 + j] = row[j];
    }
    __syncthreads();
    // 45 forward + 45 wrap-around segment fits (tools_hooks-free shared tail: fq_segment_fit / fq_corner_search above)
    if (NT == 64) {
      if (tid < 45) { fq_segment_fit(s_tab, s_maxidx, m, szd, tid, 0); fq_segment_fit(s_tab, s_maxidx, m, szd, tid, 1); }
    } else {
      // wave 0: the forward fits, wave 1: the wrap-around ones (uniform paths per wave)
      if (tid < 45) fq_segment_fit(s_tab, s_maxidx, m, szd, tid, 0);
      else if (tid >= 64 && tid < 64 + 45) fq_segment_fit(s_tab, s_maxidx, m, szd, tid - 64, 1);
    }
    __syncthreads();
    // wave 0 alone, without workgroup barriers (the other waves go on to the barrier at the top of the cluster loop)
    if (tid < 64) fq_corner_search(s_tab, s_cpairs, m, szd, tid, P, cands_all, counters, frame, cl.key, q_reversed);
    FQ_TICK(7)
  }
}

// second launch-bound argument = minimum waves per SIMD the register allocation must allow
template <int NT, bool SPLIT>
__global__ __launch_bounds__(NT, (NT == 64 ? FQ_WPE_64 : NT == 128 ? FQ_WPE_128 : NT == 256 ? FQ_WPE_256 : 4)) void k_fit_quads(const FrameDesc* __restrict__ frames, const uint8_t* __restrict__ gray_all,
                                                   const uint32_t* __restrict__ pts_all, const ClusterRec* __restrict__ clusters_all,
                                                   const uint32_t* __restrict__ work, const uint32_t* __restrict__ work_n, uint32_t work_cap,
                                                   uint32_t* __restrict__ work_cursor, double* __restrict__ lf_scratch,
                                                   unsigned long long* __restrict__ keys_scratch, double* __restrict__ errs_scratch,
                                                   FitCand* __restrict__ cands_all, FrameCounters* __restrict__ counters,
                                                   unsigned long long* __restrict__ prof, int sort_cap, int slot_cap,
                                                   int pop, DetParams P) {
  fit_quads_body<NT, SPLIT>(frames, gray_all, pts_all, clusters_all, work, work_n, work_cap, work_cursor, lf_scratch, keys_scratch, errs_scratch,
                            cands_all, counters, prof, sort_cap, slot_cap, pop, P);
}

// ---- k_fit_prefilter: the cheap exits of the quad fit for the clusters of the large classes, ahead of k_fit_quads ---------
// Clusters above FQ_PREFILTER_LO points (the 256-, 512- and 1024-thread classes) need most of a CU's LDS for their sort
// keys, run as one or two persistent workgroups per CU and, on frames with textured background, almost all end at "no
// admissible corner choice" (config 2: every one of them).  This kernel takes exactly those work items and applies the
// exits that need neither the sort nor the key array -- bounding box, border direction (the same statements as
// k_fit_quads) and the sound sector test (fq_sector / fq_feasible above) -- at full occupancy: 256 threads and about
// 10 KB of LDS per cluster.  The surviving items are appended to a second, compact work list per class, which is what the
// persistent workgroups of these classes pop from (marking the rejected items in place left them popping tens of thousands
// of dead items through one cursor word: 0.25 ms per class at the ~90 atomics per microsecond one word sustains).  Nothing
// here decides anything k_fit_quads would decide differently: bounding box and border direction are its own tests, and the
// sector test only fires when no corner choice can be admissible.  Items are taken in a fixed stride (largest class
// first, so that neighbouring items are of similar size): a shared cursor would saturate here as well.
// Two instances: 64 threads -- one wave per cluster, no workgroup barriers, as many clusters in flight as the chip holds
// waves (throughput-sized submissions: tens of thousands of clusters) -- and 1024 threads, where one cluster's points are
// spread over a whole CU (small submissions: a single frame has ~120 such clusters and 256 CUs to put them on; with one
// wave each the largest cluster alone took 0.2 ms).
template <int FQ_PF_NT>
__global__ __launch_bounds__(FQ_PF_NT) void k_fit_prefilter(const FrameDesc* __restrict__ frames, const uint8_t* __restrict__ gray_all,
                                                       const uint32_t* __restrict__ pts_all, const ClusterRec* __restrict__ clusters_all,
                                                       const uint32_t* __restrict__ work, const uint32_t* __restrict__ work_n,
                                                       uint32_t* __restrict__ work_out, uint32_t* __restrict__ work_n_out,
                                                       uint32_t* __restrict__ cursor, FqWorkLayout L, int first_class,
                                                       unsigned long long* __restrict__ prof, DetParams P) {
  constexpr int FQ_PF_CHUNK = 8 * FQ_PF_NT;     // points staged in LDS per round (eight per thread)
  __shared__ __attribute__((aligned(16))) uint32_t spts[FQ_PF_CHUNK];
  __shared__ double sB[(64 + 1) * 7];   // sums of the 64 sectors, then their prefixes
  __shared__ long long s_dot[FQ_PF_NT / 64][3];
  __shared__ int s_box[FQ_PF_NT / 64][4];
  __shared__ uint32_t s_okf[FQ_XG], s_okw[FQ_XG];
  __shared__ int s_feasible;
  const int tid = threadIdx.x;
  const int W = P.W, H = P.H;
  PF_HOOKS_DECL   // (tools_hooks.h)
  TL_MARK_MIN(0)
  // items of the classes first_class .. FQ_NCLS - 1, largest class first
  uint32_t cnt[FQ_NCLS], total = 0;
#pragma unroll
  for (int c = FQ_NCLS - 1; c >= 0; c--) {
    cnt[c] = c >= first_class ? min(work_n[c], L.cap[c]) : 0u;
    total += cnt[c];
  }
  // One wave per cluster (FQ_PF_NT == 64): a wave's FIRST item is its own index, the rest come from a cursor -- largest class
  // first, so the clusters of tens of thousands of points start at t = 0 and the many clusters of 2 000 .. 4 000 points fill in
  // behind them (in a fixed stride a wave that began with a 30 000-point cluster still had its three other items to do when the
  // rest of the chip had drained: the kernel ran alone for 1.2 ms at 62 % of its issue time).  The CU-wide instance (one-frame
  // submissions: about as many items as workgroups) keeps the fixed stride.
  uint32_t it0 = blockIdx.x;
  for (; it0 < total; it0 = (FQ_PF_NT == 64 ? gridDim.x + (uint32_t)__builtin_amdgcn_readfirstlane((int)(lane_id() == 0 ? atomicAdd(cursor, 1u) : 0u)) : it0 + gridDim.x)) {
    __syncthreads();   // the previous cluster's LDS use is finished in every wave
    uint32_t it = it0, widx = 0;
    int cls = 0;
    {
      bool found = false;
#pragma unroll
      for (int c = FQ_NCLS - 1; c >= 0; c--) {
        if (!found) {
          if (it < cnt[c]) { widx = L.off[c] + it; cls = c; found = true; }
          else it -= cnt[c];
        }
      }
    }
    const uint32_t wi = (uint32_t)__builtin_amdgcn_readfirstlane((int)work[widx]);
    const int frame = (int)(wi >> P.wshift);
    const FrameDesc fd = frames[frame];
    const uint8_t* gray = (P.decimate > 1) ? gray_all + (size_t)frame * P.H * P.WS : fd.img;
    const int gpitch = (P.decimate > 1) ? P.WS : (int)fd.pitch;
    const __attribute__((address_space(1))) uint8_t* const ggray = (const __attribute__((address_space(1))) uint8_t*)gray;
    const ClusterRec cl = clusters_all[(size_t)frame * P.ccap + (wi & ((1u << P.wshift) - 1u))];
    const int sz = (int)cl.count;
    const uint32_t* pts = pts_all + (size_t)frame * P.pcap + cl.start;

    // bounding box and exact gradient dot: the statements of k_fit_quads
    int xmin = 1 << 30, xmax = -1, ymin = 1 << 30, ymax = -1;
    long long sxg = 0, sgx = 0, sgy = 0;
    for (int i = tid; i < sz; i += 8 * FQ_PF_NT) {   // (eight independent loads per trip)
      uint32_t pq[8];
#pragma unroll
      for (int u = 0; u < 8; u++) pq[u] = pts[min(i + u * FQ_PF_NT, sz - 1)];
      // gradient signs (-1, 0, 1) summed in 32 bits per trip (8 x 2^14 at most), scaled by 255 once per trip
      int t_xg = 0, t_gx = 0, t_gy = 0;
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const uint32_t p = (i + u * FQ_PF_NT < sz) ? pq[u] : ((pq[0] & ~15u) | 5u);
        const int x = (int)(p >> 18), y = (int)((p >> 4) & 0x3FFF);
        const int gx = (int)((p >> 2) & 3) - 1, gy = (int)(p & 3) - 1;
        xmin = min(xmin, x); xmax = max(xmax, x); ymin = min(ymin, y); ymax = max(ymax, y);
        t_xg += __mul24(x, gx) + __mul24(y, gy);
        t_gx += gx; t_gy += gy;
      }
      sxg += (long long)(t_xg * 255); sgx += t_gx * 255; sgy += t_gy * 255;
    }
    xmin = wave_min_i(xmin); xmax = wave_max_i(xmax); ymin = wave_min_i(ymin); ymax = wave_max_i(ymax);
    sxg = wave_sum_ll(sxg); sgx = wave_sum_ll(sgx); sgy = wave_sum_ll(sgy);
    {
      const int wv = tid >> 6;
      if (lane_id() == 0) {
        s_box[wv][0] = xmin; s_box[wv][1] = xmax; s_box[wv][2] = ymin; s_box[wv][3] = ymax;
        s_dot[wv][0] = sxg; s_dot[wv][1] = sgx; s_dot[wv][2] = sgy;
      }
      for (int t = tid; t < (64 + 1) * 7; t += FQ_PF_NT) sB[t] = 0.0;
      __syncthreads();
      xmin = s_box[0][0]; xmax = s_box[0][1]; ymin = s_box[0][2]; ymax = s_box[0][3];
      sxg = s_dot[0][0]; sgx = s_dot[0][1]; sgy = s_dot[0][2];
#pragma unroll
      for (int w = 1; w < FQ_PF_NT / 64; w++) {
        xmin = min(xmin, s_box[w][0]); xmax = max(xmax, s_box[w][1]); ymin = min(ymin, s_box[w][2]); ymax = max(ymax, s_box[w][3]);
        sxg += s_dot[w][0]; sgx += s_dot[w][1]; sgy += s_dot[w][2];
      }
    }
    PF_TICK(60)
    bool reject = (xmax - xmin) * (ymax - ymin) < P.min_tag_width;
    const double cxd = (xmin + xmax) * 0.5 + 0.05118, cyd = (ymin + ymax) * 0.5 + -0.028581;
    const double dot = (double)sxg - cxd * (double)sgx - cyd * (double)sgy;
    const int q_reversed = dot < 0;
    if (!P.reversed_border && q_reversed) reject = true;
    if (!P.normal_border && !q_reversed) reject = true;
    if (!reject && P.split_moments) {   // (the sector sums assume the fast path's coordinate range; larger images skip the test)
      const float cx = (float)cxd, cy = (float)cyd;
      double a[7];
      int cur = -1;
      auto flush = [&]() {
#pragma unroll
        for (int j = 0; j < 7; j++) __hip_atomic_fetch_add(&sB[(cur + 1) * 7 + j], a[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      };
      for (int base = 0; base < sz; base += FQ_PF_CHUNK) {
        if (base) __syncthreads();   // the previous round's points have been consumed
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const int i = base + k * FQ_PF_NT + tid;
          if (i < sz) spts[k * FQ_PF_NT + tid] = pts[i];
        }
        __syncthreads();
        uint32_t pp[8];
        {
          const uint4 v0 = *reinterpret_cast<const uint4*>(spts + tid * 8), v1 = *reinterpret_cast<const uint4*>(spts + tid * 8 + 4);
          pp[0] = v0.x; pp[1] = v0.y; pp[2] = v0.z; pp[3] = v0.w; pp[4] = v1.x; pp[5] = v1.y; pp[6] = v1.z; pp[7] = v1.w;
        }
        // squared gradients of the eight points first (32 independent byte gathers in flight)
        uint32_t GG[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const uint32_t p = pp[e];
          const uint32_t px = p >> 18, py = (p >> 4) & 0x3FFFu;
          const int ix = (int)((px + 1) >> 1), iy = (int)((py + 1) >> 1);
          GG[e] = 0;
          if ((base + tid * 8 + e < sz) & ((unsigned)(ix - 1) < (unsigned)(W - 2)) & ((unsigned)(iy - 1) < (unsigned)(H - 2))) {
            const uint32_t o = __umul24((uint32_t)iy, (uint32_t)gpitch) + (uint32_t)ix;   // (rows below 2^14, pitches below 2^24: check_images)
            const int g_r = ggray[o + 1], g_l = ggray[o - 1], g_d = ggray[o + (uint32_t)gpitch], g_u = ggray[o - (uint32_t)gpitch];
            const int grad_x = g_r - g_l, grad_y = g_d - g_u;
            GG[e] = (uint32_t)(grad_x * grad_x + grad_y * grad_y);
          }
        }
#pragma unroll
        for (int e = 0; e < 8; e++) {
          if (base + tid * 8 + e >= sz) break;
          const uint32_t p = pp[e];
          const int xi = (int)(p >> 18), yi = (int)((p >> 4) & 0x3FFF);
          // slope key of k_fit_quads (same float statements)
          float dx = (float)xi - cx, dy = (float)yi - cy;
          float quadrant;
          if (dy > 0) quadrant = (dx > 0) ? 65536.0f : 131072.0f;
          else quadrant = (dx > 0) ? 0.0f : -65536.0f;
          if (dy < 0) { dy = -dy; dx = -dx; }
          if (dx < 0) { float tmp = dx; dx = dy; dy = -tmp; }
          const float slope = quadrant + __fdiv_rn(dy, dx);
          const int sec = fq_sector64(slope);
          const uint32_t px = (uint32_t)xi, py = (uint32_t)yi;
          const double x = (int)(px + 1) * .5, y = (int)(py + 1) * .5;
          const double Wt = sqrt_u18(GG[e]) + 1;
          const double wl = (px & py & 1u) ? 0.5 * Wt : Wt;
          if (sec != cur) {
            if (cur >= 0) flush();
            cur = sec;
#pragma unroll
            for (int j = 0; j < 7; j++) a[j] = 0.0;
          }
          const double wx = wl * x, wy = wl * y;
          a[0] += wx; a[1] += wy; a[2] += wx * x; a[3] += wx * y; a[4] += wy * y; a[5] += wl; a[6] += Wt;
        }
      }
      if (cur >= 0) flush();
      __syncthreads();
      PF_TICK(61)
      if (tid < 7) {
        double run = 0.0;
        for (int sct = 1; sct <= 64; sct++) { run += sB[sct * 7 + tid]; sB[sct * 7 + tid] = run; }
      }
      __syncthreads();
      // 32 sectors first (every second prefix: a quarter of the arc evaluations); a cluster that passes is looked at again
      // with all 64 (config 2: 84 % of the points above 2048 per cluster fail the first test, 96 % the second)
      reject = !fq_feasible<FQ_PF_NT, 14, 6>(sB, P.max_line_fit_mse, W, H, s_okf, s_okw, &s_feasible);
      PF_TICK(62)
      if (!reject) {   // (one wave evaluates the 64 x 64 relation; a larger workgroup waits for its verdict)
        if (tid < 64) { const bool f64 = fq_feasible64<7, 6, PF_MUT_EXTRA_SECTOR(FQ_PF_NT)>(sB, P.max_line_fit_mse, W, H); if (tid == 0) s_feasible = f64 ? 1 : 0; }
        __syncthreads();
        reject = s_feasible == 0;
      }
      PF_TICK(63)
    }
    if (!reject && tid == 0) {
      // (one list per class: sending all survivors to the largest class's 1024-thread workgroups, one per CU, was measured
      // slower -- 0.53 against 0.35 ms of tail: they are few, but each is a chain of ~100 us)
      const uint32_t pos = atomicAdd(&work_n_out[cls], 1u);   // (pos < cap: the list holds at most the items of the input list)
      work_out[L.off[cls] + pos] = wi;
    }
  }
  TL_MARK_MAX(1)
}

// ---- k_quad_finish: corners, area and angle checks of the candidates k_fit_quads found (one thread per candidate) -------
// The statements are the serial form of upstream's fit_quad tail (and of the CPU restatement): four line intersections,
// corners rounded to float, two Heron triangles, four corner angles with the winding test.
__global__ __launch_bounds__(256) void k_quad_finish(const FitCand* __restrict__ cands_all, QuadRec* __restrict__ quads_all,
                                                     FrameCounters* __restrict__ counters, DetParams P) {
  TL_MARK_MIN(3)
  const int frame = (int)blockIdx.y + P.frame0;
  uint32_t ncand = counters[frame].ncand;
  if (ncand > P.cand_cap) ncand = P.cand_cap;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < ncand; i += gridDim.x * 256) {
    FitCand cd = cands_all[(size_t)frame * P.cand_cap + i];
    if (cd.wrap_is_moments) {
      // k_fit_small hands over the six moments of the wrap-around segment instead of its line: the same fit, here
      double lp[4];
      fit_line_moments(cd.line[3][0], cd.line[3][1], cd.line[3][2], cd.line[3][3], cd.wm[0], cd.wm[1], 0, lp, nullptr, nullptr);
      cd.line[3][0] = lp[0]; cd.line[3][1] = lp[1]; cd.line[3][2] = lp[2]; cd.line[3][3] = lp[3];
    }
    float corner[4][2];
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int j = (k + 1) & 3;
      const double A00 = cd.line[k][3], A01 = -cd.line[j][3];
      const double A10 = -cd.line[k][2], A11 = cd.line[j][2];
      const double B0 = -cd.line[k][0] + cd.line[j][0];
      const double B1 = -cd.line[k][1] + cd.line[j][1];
      const double det = A00 * A11 - A10 * A01;
      if (fabs(det) < 0.001) ok = false;
      const double W00 = A11 / det, W01 = -A01 / det;
      const double L0 = W00 * B0 + W01 * B1;
      corner[k][0] = (float)(cd.line[k][0] + L0 * A00);
      corner[k][1] = (float)(cd.line[k][1] + L0 * A10);
    }
    if (!ok) continue;
    double len[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int i1 = (k + 1) & 3, i2 = (k + 2) & 3;
      // upstream: double dx1 = quad->p[i1][0] - quad->p[i0][0] and sq(quad->p[b][0] - quad->p[a][0]) with float p[][]:
      // the corner differences are FLOAT operations, widened afterwards
      const double dx1 = (double)(corner[i1][0] - corner[k][0]), dy1 = (double)(corner[i1][1] - corner[k][1]);
      const double dx2 = (double)(corner[i2][0] - corner[i1][0]), dy2 = (double)(corner[i2][1] - corner[i1][1]);
      const double q1 = dx1 * dx1 + dy1 * dy1;
      len[k] = __dsqrt_rn(q1);
      const double cos_dtheta = (dx1 * dx2 + dy1 * dy2) / __dsqrt_rn(q1 * (dx2 * dx2 + dy2 * dy2));
      if ((cos_dtheta > P.cos_critical_rad || cos_dtheta < -P.cos_critical_rad) || dx1 * dy2 < dy1 * dx2) ok = false;
    }
    double area = 0;
    {
      // diagonal: triangle (0,1,2) takes it as p[0] - p[2], triangle (2,3,0) as p[2] - p[0]; float differences negate exactly
      const double ddx = (double)(corner[0][0] - corner[2][0]), ddy = (double)(corner[0][1] - corner[2][1]);
      const double ed = __dsqrt_rn(ddx * ddx + ddy * ddy);
#pragma unroll
      for (int t = 0; t < 2; t++) {   // triangle (0,1,2): sides len0, len1, diagonal; triangle (2,3,0): len2, len3, diagonal
        const double sa = len[2 * t], sb = len[2 * t + 1];
        const double hp = (sa + sb + ed) / 2;
        area += __dsqrt_rn(hp * (hp - sa) * (hp - sb) * (hp - ed));
      }
    }
    if (area < 0.95 * P.min_tag_width * P.min_tag_width) ok = false;
    if (!ok) continue;
    QuadRec q;
#pragma unroll
    for (int c = 0; c < 4; c++) {
      float fx = corner[c][0], fy = corner[c][1];
      if (P.decimate > 1) {
        const double f = (double)(float)P.decimate;
        fx = (float)(((double)fx - 0.5) * f + 0.5);
        fy = (float)(((double)fy - 0.5) * f + 0.5);
      }
      q.p[c][0] = fx; q.p[c][1] = fy;
    }
    q.reversed_border = cd.reversed_border;
    q.pad = 0;
    q.key = cd.key;
    const uint32_t qi = atomicAdd(&counters[frame].nquads, 1u);
    if (qi < P.qcap) quads_all[(size_t)frame * P.qcap + qi] = q;
    else atomicOr(&counters[frame].flags, 0x8u);
  }
}
