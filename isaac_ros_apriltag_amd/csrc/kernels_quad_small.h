// kernels_quad_small.h -- S5 quad fitting for the SMALL clusters (at most 64 K boundary points, K = 2 or 4), the bulk of a
// noisy frame's clusters (inside cuAprilTagsDetect, reference src/apriltag_node.cpp:491-493).
// Same statements and the same results, bit for bit, as k_fit_quads (kernels_quad.h); what differs is where the data lives:
//   * a lane loads its K points ONCE and keeps them in registers through bounding box, border direction and slope keys;
//   * the keys never touch LDS: the bitonic network runs on the registers (partner keys over ds_bpermute), and after it
//     lane l holds the sorted positions l K .. l K + K - 1 -- exactly the lane-contiguous runs the moment sweep walks;
//   * the duplicate test takes the previous key from the neighbour lane over DPP, the sweep's per-point state (position,
//     squared gradient, kept flag) stays in registers between its two walks;
//   * the cumulative moments (48 bytes per kept point) live in LDS, not in a global scratch slot: the windowed errors, the
//     rows of the pair fits and the fits themselves never leave the CU (k_fit_quads sends them to HBM and back);
//   * box and border direction are reduced in 32-bit integers (gradient signs, scaled by 255 once);
//   * the pair tables keep (error, mse) for all 90 segments but line parameters only for the 45 forward ones: the
//     wrap-around segment's line is needed for the one chosen corner set only, and k_quad_finish computes it from the
//     six moments the candidate record carries; the corner search tests the four mse bits of a choice before it touches
//     normals or errors, with the pair indices of the 210 choices from a compile-time table.
// One wave per cluster, no workgroup barriers that synchronise anything (a one-wave workgroup's barrier is only a
// compiler fence).
// K = 2 (clusters up to 128 points, moments in LDS) is the instance the library launches.  K = 4 (129 .. 256 points, GROWS: moments
// in a global scratch slot, because 12 KB of them per workgroup would cut the resident workgroups to 10 per CU) compiles and is
// bit-exact too, but measured no gain over k_fit_quads<64> for its sizes and is off: FS_B1 == FS_B0 in
// detector.hip's class table leaves its class empty.
#pragma once
#include "kernels_quad.h"

__device__ __forceinline__ int wave_sum_i(int v) {
#define OP(C, M) v += __builtin_amdgcn_update_dpp(0, v, C, M, 0xF, true);
  AT_DPP_STEPS(OP)
#undef OP
  return __builtin_amdgcn_readlane(v, 63);
}

#ifndef FS_WPE
#define FS_WPE 4   // waves per SIMD the register allocation must allow
#endif
#ifndef FS_GRID_K2
#define FS_GRID_K2 16   // persistent workgroups per CU of the K = 2 class
#endif
#ifndef FS_GRID_K4
#define FS_GRID_K4 16   // persistent workgroups per CU of the K = 4 class
#endif
#define FS_TAB_BYTES (FQT_DOUBLES * 8)   // pair tables + staged moment rows
// LDS of a workgroup: GROWS = false -- moments [64 K][6] | errors [64 K], the tables over the moments once they are dead;
//                     GROWS = true  -- tables | errors [64 K]; the moments live in the workgroup's global scratch slot
#define FS_LDS_BYTES(K, GROWS) ((GROWS) ? (FS_TAB_BYTES + 64 * (K) * 8) : ((64 * (K) * 56) > FS_TAB_BYTES ? (64 * (K) * 56) : FS_TAB_BYTES))

template <int K, bool GROWS>
__device__ __forceinline__ void fit_small_body(const FrameDesc* __restrict__ frames, const uint8_t* __restrict__ gray_all,
                                                          const uint32_t* __restrict__ pts_all, const ClusterRec* __restrict__ clusters_all,
                                                          const uint32_t* __restrict__ work, const uint32_t* __restrict__ work_n, uint32_t work_cap,
                                                          uint32_t* __restrict__ work_cursor, double* __restrict__ lf_scratch, FitCand* __restrict__ cands_all,
                                                          FrameCounters* __restrict__ counters, int pop, DetParams P) {
  constexpr int CAP = 64 * K;
  extern __shared__ __attribute__((aligned(16))) unsigned char fs_smem[];
  // [CAP][6] cumulative moments of the kept points: in LDS, or (GROWS: the K = 4 class, whose 12 KB of moments would cut
  // the resident workgroups to 10 per CU) in the workgroup's slot of a global scratch array, which stays in L2
  double* const lds = reinterpret_cast<double*>(fs_smem);
  double* const rows = GROWS ? lf_scratch + (size_t)blockIdx.x * CAP * 6 : lds;
  // [CAP] windowed errors, then the smoothed ones, then the maxima list
  double* const errs = GROWS ? lds + FQT_DOUBLES : lds + CAP * 6;
  // the pair tables and the staged rows (FQT_* layout, kernels_quad.h): in the moments' place once the maxima are selected
  double* const s_tab = lds;
  double* const s_rows = s_tab + FQT_ROWS;
  static_assert(GROWS || FQT_DOUBLES * 8 <= 64 * K * 56, "tables fit the moment region");
  __shared__ uint32_t s_cpairs[210];
  __shared__ int s_maxidx[16];

  const int lane = (int)threadIdx.x;
  const int W = P.W, H = P.H;
  const uint32_t nwork = min(*work_n, work_cap);
  for (int t = lane; t < 210; t += 64) s_cpairs[t] = g_combo_pairs.v[t];

  // Software pipeline over the work list: while cluster c is being fitted, the loads of cluster c + 1 are in flight -- its
  // work item (stage 1, issued before c's box reductions), its cluster record and frame (stage 2, before c's sort) and its
  // points (stage 3, before c's error pass).  A cluster used to start with four dependent global round trips (list cursor /
  // item / record / points) in front of its first instruction, a fifth of its time when the kernel runs alone.
  typedef const __attribute__((address_space(1))) uint32_t* gptr32;
  uint32_t next_item = blockIdx.x, chunk_left = 1;   // uniform; the first item is the workgroup's own index (no atomic), the cursor
                                                      // hands out the items from gridDim.x on (see k_fit_quads)
  auto next_index = [&]() -> uint32_t {
    if (chunk_left == 0) {
      uint32_t got = 0;
      if (lane == 0) got = gridDim.x + atomicAdd(work_cursor, (uint32_t)pop);
      next_item = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
      chunk_left = (uint32_t)pop;
    }
    chunk_left--;
    return next_item++;
  };
  struct Pending {      // cluster whose loads are in flight
    uint32_t item;      // uniform; >= nwork: none
    uint32_t wi_v;      // stage 1: work item (the same word in every lane)
    uint32_t rec_v[4];  // stage 2: cluster record {key lo, key hi, start, count}
    uint32_t img_v[3];  //          frame: image pointer lo / hi, pitch
    uint32_t pp[K];     // stage 3: the lane's points
  };
  auto stage1 = [&](Pending& n) {
    n.item = next_index();
    n.wi_v = 0;
    if (n.item < nwork) n.wi_v = ((gptr32)work)[n.item];
  };
  auto stage2 = [&](Pending& n) {
    if (n.item >= nwork) return;
    const uint32_t wi = (uint32_t)__builtin_amdgcn_readfirstlane((int)n.wi_v);
    const int frame = (int)(wi >> P.wshift);
    const gptr32 rec = (gptr32)(clusters_all + (size_t)frame * P.ccap + (wi & ((1u << P.wshift) - 1u)));
    const gptr32 fdp = (gptr32)(frames + frame);
#pragma unroll
    for (int j = 0; j < 4; j++) n.rec_v[j] = rec[j];
    n.img_v[0] = fdp[0]; n.img_v[1] = fdp[1]; n.img_v[2] = fdp[2];   // FrameDesc: img (8 bytes), pitch
  };
  auto stage3 = [&](Pending& n) {
    if (n.item >= nwork) return;
    const uint32_t wi = (uint32_t)__builtin_amdgcn_readfirstlane((int)n.wi_v);
    const int frame = (int)(wi >> P.wshift);
    const uint32_t start = (uint32_t)__builtin_amdgcn_readfirstlane((int)n.rec_v[2]);
    const int sz = __builtin_amdgcn_readfirstlane((int)n.rec_v[3]);
    const gptr32 pts = (gptr32)(pts_all + (size_t)frame * P.pcap + start);
#pragma unroll
    for (int j = 0; j < K; j++) n.pp[j] = pts[min(lane + 64 * j, max(sz, 1) - 1)];
  };
  static_assert(offsetof(FrameDesc, img) == 0 && offsetof(FrameDesc, pitch) == 8 && sizeof(ClusterRec) == 16, "stage 2 reads raw words");
  Pending cur, nxt;
  stage1(cur); stage2(cur); stage3(cur);
  while (cur.item < nwork) {
    __syncthreads();   // (compiler fence: the previous cluster's LDS reads are done)
    int pf = 0;        // prefetch stages of the next cluster issued so far
    do {
    const uint32_t wi = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur.wi_v);
    const int frame = (int)(wi >> P.wshift);
    const uint8_t* gray = (P.decimate > 1) ? gray_all + (size_t)frame * P.H * P.WS
                                           : reinterpret_cast<const uint8_t*>(((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)cur.img_v[1]) << 32) |
                                                                              (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)cur.img_v[0]));
    const int gpitch = (P.decimate > 1) ? P.WS : __builtin_amdgcn_readfirstlane((int)cur.img_v[2]);
    const __attribute__((address_space(1))) uint8_t* const ggray = (const __attribute__((address_space(1))) uint8_t*)gray;
    const unsigned long long cl_key = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)cur.rec_v[1]) << 32) |
                                      (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)cur.rec_v[0]);
    const int sz = __builtin_amdgcn_readfirstlane((int)cur.rec_v[3]);
    stage1(nxt); pf = 1;
    if (sz < 24 || sz > CAP) break;   // (the work list only holds clusters of this class)

    AT_MARK("box")
    // ---- the lane's K points, bounding box and exact gradient dot -------------------------------------------------
    uint32_t pp[K];
#pragma unroll
    for (int j = 0; j < K; j++) pp[j] = cur.pp[j];
    int xmin = 1 << 30, xmax = -1, ymin = 1 << 30, ymax = -1;
    int sxg = 0, sg = 0;   // sums of x sgn(gx) + y sgn(gy) and of the packed signs (sgn(gx) << 16) + sgn(gy): |sums| < 2^24, 2^9
#pragma unroll
    for (int j = 0; j < K; j++) {
      // a lane without a point in this slot repeats a point of the cluster with a zero gradient (neither box nor sums change)
      const uint32_t p = (lane + 64 * j < sz) ? pp[j] : ((pp[0] & ~15u) | 5u);
      const int x = (int)(p >> 18), y = (int)((p >> 4) & 0x3FFF);
      const int gx = (int)((p >> 2) & 3) - 1, gy = (int)(p & 3) - 1;
      xmin = min(xmin, x); xmax = max(xmax, x); ymin = min(ymin, y); ymax = max(ymax, y);
      sxg += __mul24(x, gx) + __mul24(y, gy);
      sg += gx * 65536 + gy;
    }
    xmin = wave_min_i(xmin); xmax = wave_max_i(xmax); ymin = wave_min_i(ymin); ymax = wave_max_i(ymax);
    sxg = wave_sum_i(sxg); sg = wave_sum_i(sg);
    if ((xmax - xmin) * (ymax - ymin) < P.min_tag_width) break;
    const int sgy_s = (int)(short)(sg & 0xFFFF), sgx_s = (sg - sgy_s) >> 16;
    const double cxd = (xmin + xmax) * 0.5 + 0.05118, cyd = (ymin + ymax) * 0.5 + -0.028581;
    // (the gradients are +-255: the integer sums of k_fit_quads are 255 times these, exactly)
    const double dot = (double)(255LL * sxg) - cxd * (double)(255 * sgx_s) - cyd * (double)(255 * sgy_s);
    const int q_reversed = dot < 0;
    if (!P.reversed_border && q_reversed) break;
    if (!P.normal_border && !q_reversed) break;

    stage2(nxt); pf = 2;
    AT_MARK("keys")
    // ---- slope keys (the statements of k_fit_quads) and the sort, in registers ---------------------------------------
    const float cx = (float)cxd, cy = (float)cyd;
    unsigned long long v[K];
#pragma unroll
    for (int j = 0; j < K; j++) {
      const uint32_t p = pp[j];
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
      v[j] = (lane + 64 * j < sz) ? key : AT_KEY_PAD;
    }
    AT_MARK("sort")
    fq_wave_sort_regs<K>(v);

    AT_MARK("walk1")
    // ---- moment sweep: walk 1 (terms of the lane's kept points), one scan, walk 2 (rounded prefixes to LDS) ------------
    // lane l owns the sorted positions l K + j; the key before its first one comes from lane l - 1 (wave_shr:1)
    unsigned long long prev;
    {
      const unsigned long long last = v[K - 1];
      const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)last, 0x138, 0xF, 0xF, false);
      const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(last >> 32), 0x138, 0xF, 0xF, false);
      prev = key_dec(((unsigned long long)hi << 32) | lo);
    }
    uint32_t st_xy[K], st_g[K];   // per point: kept flag << 31 | py << 14 | px; squared gradient magnitude
    D2 acc[6];
#pragma unroll
    for (int j = 0; j < 6; j++) { acc[j].hi = 0; acc[j].lo = 0; }
    int kept = 0;
#pragma unroll
    for (int j = 0; j < K; j++) {
      const int i = lane * K + j;
      const unsigned long long key = key_dec(v[j]);
      const bool keep = (i < sz) && ((i == 0) || ((key >> 4) != (prev >> 4)));
      prev = key;
      const uint32_t px = (uint32_t)((key >> 4) & 0x3FFF), py = (uint32_t)((key >> 18) & 0x3FFF);
      uint32_t G = 0;
      if (keep) {
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
        for (int m = 0; m < 6; m++) {
          const D2 t = split_term(tt[m]);
          acc[m].hi += t.hi; acc[m].lo += t.lo;
        }
        kept++;
      }
      st_xy[j] = ((keep ? 1u : 0u) << 31) | (py << 14) | px;
      st_g[j] = G;
    }
    AT_MARK("scan")
    D2 off[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
      const double c = (acc[j].lo + AT_SPLIT_C) - AT_SPLIT_C;
      acc[j].hi += c; acc[j].lo -= c;
      off[j].hi = wave_scan_f64(acc[j].hi) - acc[j].hi;
      off[j].lo = wave_scan_f64(acc[j].lo) - acc[j].lo;
    }
    int kincl = kept;
#define OP(C, M) kincl += __builtin_amdgcn_update_dpp(0, kincl, C, M, 0xF, true);
    AT_DPP_STEPS(OP)
#undef OP
    AT_MARK("walk2")
    int pos = kincl - kept;
    const int szd = __builtin_amdgcn_readlane(kincl, 63);
    if (szd < 24) break;
#pragma unroll
    for (int j = 0; j < K; j++) {
      if (st_xy[j] >> 31) {
        const double x = (int)((st_xy[j] & 0x3FFFu) + 1u) * .5, y = (int)(((st_xy[j] >> 14) & 0x3FFFu) + 1u) * .5;
        const double Wt = sqrt_u18(st_g[j]) + 1;
        const double tt[6] = {Wt * x, Wt * y, Wt * x * x, Wt * x * y, Wt * y * y, Wt};
        double r[6];
#pragma unroll
        for (int m = 0; m < 6; m++) {
          const D2 t = split_term(tt[m]);
          off[m].hi += t.hi; off[m].lo += t.lo;
          r[m] = off[m].hi + off[m].lo;
        }
        double2* const o = reinterpret_cast<double2*>(rows + pos * 6);
        o[0] = make_double2(r[0], r[1]); o[1] = make_double2(r[2], r[3]); o[2] = make_double2(r[4], r[5]);
        pos++;
      }
    }
    __syncthreads();   // (fence: the rows are read by other lanes below)

    stage3(nxt); pf = 3;
    AT_MARK("errors")
    // ---- windowed line-fit error, smoothing, local maxima ------------------------------------------------------------
    const int ksz = min(20, szd / 12);
    double sm[K];
#pragma unroll
    for (int it = 0; it < K; it++) {
      const int i = lane + it * 64;
      if (i < szd) {
        double e;
        const int i0 = (i >= ksz) ? i - ksz : i - ksz + szd;
        const int i1 = (i + ksz < szd) ? i + ksz : i + ksz - szd;
        fit_line_dev(rows, szd, i0, i1, nullptr, &e, nullptr);
        errs[i] = e;
      }
    }
    __syncthreads();
    AT_MARK("smooth")
    const float f0 = 0x1.6c0504p-7f, f1 = 0x1.152aaap-3f, f2 = 0x1.368b3p-1f;
    const double F0 = (double)f0, F1 = (double)f1, F2 = (double)f2;
    auto wrap = [szd](int k) { return k < 0 ? k + szd : (k >= szd ? k - szd : k); };
#pragma unroll
    for (int it = 0; it < K; it++) {
      const int i = lane + it * 64;
      sm[it] = 0;
      if (i < szd) {
        double a2 = 0;
        a2 += errs[wrap(i - 3)] * F0;
        a2 += errs[wrap(i - 2)] * F1;
        a2 += errs[wrap(i - 1)] * F2;
        a2 += errs[i] * 1.0;
        a2 += errs[wrap(i + 1)] * F2;
        a2 += errs[wrap(i + 2)] * F1;
        a2 += errs[wrap(i + 3)] * F0;
        sm[it] = a2;
      }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < K; it++) {
      const int i = lane + it * 64;
      if (i < szd) errs[i] = sm[it];
    }
    __syncthreads();
    uint32_t mx = 0;
#pragma unroll
    for (int it = 0; it < K; it++) {
      const int i = lane + it * 64;
      if (i < szd) {
        const double e = sm[it];
        if (e > errs[i + 1 < szd ? i + 1 : 0] && e > errs[i > 0 ? i - 1 : szd - 1]) mx |= 1u << it;
      }
    }
    __syncthreads();
    // the maxima, in ascending index order: values in errs[0 ..), indices behind them
    double* const cand_val = errs;
    int* const cand_idx = reinterpret_cast<int*>(errs + CAP / 2);   // (maxima are never neighbours: at most CAP / 2 of them)
    int nmaxima = 0;
#pragma unroll
    for (int it = 0; it < K; it++) {
      const bool is_max = (mx >> it) & 1u;
      const unsigned long long mm = __ballot(is_max);
      if (is_max) {
        const int k = nmaxima + (int)__popcll(mm & ((1ull << lane) - 1ull));
        cand_val[k] = sm[it];
        cand_idx[k] = lane + it * 64;
      }
      nmaxima += (int)__popcll(mm);
    }
    __syncthreads();
    if (nmaxima < 4) break;

    AT_MARK("top10")
    // ---- at most max_nmaxima corners: those whose value exceeds the (max_nmaxima + 1)-th largest --------------------------
    int m;
    if (nmaxima > P.max_nmaxima) {
      // KR candidates per lane as order-preserving keys (0 = none); rank of a key = keys above it (value descending,
      // position ascending among equals); the key of rank max_nmaxima is the threshold
      constexpr int KR = K <= 2 ? 1 : 2;
      unsigned long long myk[KR];
#pragma unroll
      for (int r = 0; r < KR; r++) {
        const int k = lane + 64 * r;
        myk[r] = (k < nmaxima) ? double_sortable(cand_val[k] + 0.0) : 0ull;
      }
      int rank[KR];
#pragma unroll
      for (int r = 0; r < KR; r++) rank[r] = 0;
#pragma unroll
      for (int r2 = 0; r2 < KR; r2++) {
        const int n2 = min(64, nmaxima - 64 * r2);
        for (int l = 0; l < n2; l++) {
          const unsigned long long ok =
              (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)myk[r2], l) |
              ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(myk[r2] >> 32), l) << 32);
#pragma unroll
          for (int r = 0; r < KR; r++) rank[r] += (ok > myk[r] || (ok == myk[r] && (r2 < r || (r2 == r && l < lane)))) ? 1 : 0;
        }
      }
      // the threshold key: the one of rank max_nmaxima (exactly one candidate has it)
      unsigned long long tk = 0;
#pragma unroll
      for (int r = 0; r < KR; r++) {
        const unsigned long long tmask = __ballot(myk[r] != 0ull && rank[r] == P.max_nmaxima);
        if (tmask) {
          const int tl = (int)__ffsll((long long)tmask) - 1;
          tk = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)myk[r], tl) |
               ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(myk[r] >> 32), tl) << 32);
        }
      }
      m = 0;
#pragma unroll
      for (int r = 0; r < KR; r++) {
        const bool keepc = myk[r] != 0ull && rank[r] < P.max_nmaxima && myk[r] > tk;
        const unsigned long long kmask = __ballot(keepc);
        if (keepc) s_maxidx[m + (int)__popcll(kmask & ((1ull << lane) - 1ull))] = cand_idx[lane + 64 * r];
        m += (int)__popcll(kmask);
      }
    } else {
      if (lane < nmaxima) s_maxidx[lane] = cand_idx[lane];
      m = nmaxima;
    }
    __syncthreads();
    if (m < 4) break;

    AT_MARK("rows")
    // ---- the 2 m + 1 moment rows the segment fits read, then the 45 + 45 fits ---------------------------------------------
    {
      double row[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      if (lane < 2 * m + 1) {
        const int src = lane < m ? s_maxidx[lane] : lane < 2 * m ? s_maxidx[lane - m] - 1 : szd - 1;
        if (src >= 0) {
#pragma unroll
          for (int j = 0; j < 6; j++) row[j] = rows[src * 6 + j];
        }
      }
      __syncthreads();   // (every lane holds its row before the region is rewritten)
      if (lane < 2 * m + 1) {
#pragma unroll
        for (int j = 0; j < 6; j++) s_rows[lane * 6 + j] = row[j];
      }
    }
    __syncthreads();
    AT_MARK("pairfits")
    if (lane < 45) { fq_segment_fit(s_tab, s_maxidx, m, szd, lane, 0); fq_segment_fit(s_tab, s_maxidx, m, szd, lane, 1); }
    __syncthreads();
    AT_MARK("combos")
    fq_corner_search(s_tab, s_cpairs, m, szd, lane, P, cands_all, counters, frame, cl_key, q_reversed);
    AT_MARK("end")
    } while (0);
    if (pf < 1) stage1(nxt);
    if (pf < 2) stage2(nxt);
    if (pf < 3) stage3(nxt);
    cur = nxt;
  }
}

template <int K, bool GROWS>
__global__ __launch_bounds__(64, FS_WPE) void k_fit_small(const FrameDesc* __restrict__ frames, const uint8_t* __restrict__ gray_all,
                                                          const uint32_t* __restrict__ pts_all, const ClusterRec* __restrict__ clusters_all,
                                                          const uint32_t* __restrict__ work, const uint32_t* __restrict__ work_n, uint32_t work_cap,
                                                          uint32_t* __restrict__ work_cursor, double* __restrict__ lf_scratch, FitCand* __restrict__ cands_all,
                                                          FrameCounters* __restrict__ counters, int pop, DetParams P) {
  fit_small_body<K, GROWS>(frames, gray_all, pts_all, clusters_all, work, work_n, work_cap, work_cursor, lf_scratch, cands_all, counters, pop, P);
}
