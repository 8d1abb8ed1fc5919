// kernels_cc.h -- S3 union-find connected components for the closed
// cuAprilTagsDetect call (reference src/apriltag_node.cpp:491-493).
//
// Link set (the partition is all that matters downstream): for a source pixel (x,y), x in [1,W-2],
// value v != 127:  left (x-1,y);  up (x,y-1);  and for v == 255 also up-left / up-right.
// Representative = smallest pixel index of the component (y*W + x).
//
//   k_cc_local   one 64x64 tile per 256-thread block, entirely in LDS: wave-ballot run labelling
//                of each 64-pixel row (no atomics), lock-free atomicMin union of rows, flatten,
//                per-run size counting; writes global labels (index of the tile-local root), the
//                local size at root pixels -- with the size bit (AT_LABEL_BIG) on every root that has min_component_size
//                pixels inside the tile alone -- and the tile's perimeter arrays.
//   k_cc_border  unions across tile borders with global atomicMin (only first-overlap pixels).  It reads neither the threshold
//                image nor the label array: k_cc_local leaves, per frame, the PERIMETER of every tile -- class and tile-local
//                root of the pixels of its first / last row and first / last column -- in four compact arrays laid out along the
//                borders (CcPerim below), so a border pixel's whole neighbourhood is a handful of coalesced words (the column
//                borders used to gather bytes and labels at a stride of one image row: a cache line per lane and load).
//                ★ round 6: it LISTS the roots its unions turn into non-roots (each exactly once, by the thread whose atomicMin
//                did it).  That list -- a third to a half of the perimeter roots k_cc_local used to list: a speck that touches
//                a tile border without a partner across it is final where it stands -- is all the two passes below visit.
//   k_cc_sizes   over the list: every listed root is pointed at its final representative
//                and its pixel count is added there.  Pixels keep the index of their tile-local root, so
//                a consumer reaches the representative with two loads (label[label[p]]) and the image-wide
//                flatten pass (12 B/pixel of traffic) is not needed.
//   k_cc_resolve second pass over the list: where the component has reached min_component_size pixels, the listed root's entry
//                becomes representative | AT_LABEL_BIG and the representative's own entry gets the bit (plain stores).  Roots
//                that were large enough inside their tile carry it from k_cc_local.  A consumer then needs ONE dependent load per pixel --
//                e = label[label[p] & AT_LABEL_MASK]: representative e & AT_LABEL_MASK, "large enough" e >> 31 -- and no
//                size gather at all (k_points used to fetch label[l] and csize[l]: two cache lines per root, a quarter of
//                its HBM traffic).
//   k_cc_flatten only used on demand by the stage-inspection call (writes representatives per pixel).
#pragma once
#include "common.h"

#define CC_T 64  // tile edge
#define CC_UROWS 4   // (2 rows: 3.38 ms, 4: 3.34, 8: 4.6 -- the list costs a workgroup slot per CU)

// (relaxed workgroup-scope atomic loads, not volatile ones: a volatile access keeps the generic address space and
// compiles to flat_load ... sc0 sc1 through the shared aperture instead of ds_read_b32)
// A root maps to itself, so hops past the root are harmless: the first CC_FIND_HOPS hops of a chase are taken unconditionally
// (a load each -- no compare, no exec-mask bookkeeping; the divergent loop's scalar instructions were as many as the kernel's
// vector instructions), the loop only finishes the rare longer chains.
#ifndef CC_FIND_HOPS
#define CC_FIND_HOPS 2
#endif
#ifndef CC_FLATTEN_HOPS
#define CC_FLATTEN_HOPS 4
#endif
// The tile's parent array holds BYTE offsets into itself (pixel index << 2) while the unions run: a hop is then one LDS load
// whose result is the next hop's address -- no shift per hop (the finds are a fifth of the kernel's vector instructions).
__device__ __forceinline__ uint32_t* lds_at(uint32_t* L, uint32_t byte_off) {
  return reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(L) + byte_off);
}
__device__ __forceinline__ void lds_union(uint32_t* L, uint32_t a, uint32_t b) {   // byte offsets
  for (;;) {
    // (the two chases advance together: two independent loads in flight per step)
#pragma unroll
    for (int h = 0; h < CC_FIND_HOPS; h++) {
      a = __hip_atomic_load(lds_at(L, a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      b = __hip_atomic_load(lds_at(L, b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    for (;;) {
      const uint32_t pa = __hip_atomic_load(lds_at(L, a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const uint32_t pb = __hip_atomic_load(lds_at(L, b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (pa == a && pb == b) break;
      a = pa; b = pb;
    }
    if (a == b) return;
    if (a < b) { uint32_t t = a; a = b; b = t; }
    uint32_t old = atomicMin(lds_at(L, a), b);
    if (old == a) return;
    a = old;
  }
}

// (a ROOT's own entry may carry AT_LABEL_BIG -- k_cc_local sets it on every root with enough pixels inside its tile --, a
// non-root's entry never does: the finds mask the bit)
__device__ __forceinline__ uint32_t glb_load(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Lock-free union (the larger root is pointed at the smaller one; entries only ever decrease towards the root, which is
// what keeps concurrent unions correct -- a root's entry with the size bit is larger than every index, so the atomicMin that
// makes it a non-root drops the bit with it).  Returns the root this call turned into a non-root (every root is turned exactly
// once: the border pass lists it for k_cc_sizes / k_cc_resolve), or AT_NO_LABEL when the two were joined already.
// Trees are never rebalanced, so a giant component's chain of tile-local roots
// grows with the number of tiles it has crossed.  Two shortcuts were measured and dropped: path halving inside the find
// (grandparent links at every hop: 1.27-1.36 vs 1.35-1.39 ms, noise; round 4 again: 0.71 vs 0.68 ms, one frame 0.41 vs 0.41 ms) and pointing both START entries at the final root
// with an atomicMin after the union (1.54 ms: two more atomics on entries that are already contended).  What does help
// is launching the frames of a submission interleaved (at_frame_block): the unions of one frame then contend with the
// unions of other frames for the atomic units instead of with each other for the same few roots.
__device__ __forceinline__ uint32_t glb_union(uint32_t* L, uint32_t a, uint32_t b) {
  for (;;) {
    // (the two chases advance together: two independent L2 round trips in flight per step, not one after the other)
    for (;;) {
      const uint32_t pa = glb_load(&L[a]) & AT_LABEL_MASK, pb = glb_load(&L[b]) & AT_LABEL_MASK;
      if (pa == a && pb == b) break;
      a = pa; b = pb;
    }
    if (a == b) return AT_NO_LABEL;
    if (a < b) { uint32_t t = a; a = b; b = t; }
    const uint32_t old = atomicMin(&L[a], b);
    if ((old & AT_LABEL_MASK) == a) return a;
    a = old;   // (a non-root's entry: an index without the bit)
  }
}

// Perimeter arrays of one frame (32-bit entries: tile-local root of the pixel = a pixel index, bit 31 set for a white pixel;
// AT_NO_LABEL for a pixel without a class, which includes everything outside the image):
//   top[ty][x], bot[ty][x]     first / last row of tile row ty, x = 0 .. WP - 1 (WP = tiles per row x 64)
//   left[tx][y], right[tx][y]  first / last column of tile column tx, y = 0 .. HP - 1
// i.e. a border between two tile rows is bot[ty - 1] over top[ty], entry for entry -- neighbours x - 1, x + 1 are the adjacent
// entries, tile corners need no case of their own -- and a border between tile columns is right[tx - 1] beside left[tx].
struct CcPerim {
  int WP, HP;
  uint32_t top, bot, left, right;   // word offsets of the four arrays inside a frame's block
  uint32_t words;                   // words per frame
};
__host__ __device__ inline CcPerim cc_perim_layout(int W, int H) {
  CcPerim L;
  const int ntx = (W + CC_T - 1) / CC_T, nty = (H + CC_T - 1) / CC_T;
  L.WP = ntx * CC_T; L.HP = nty * CC_T;
  L.top = 0; L.bot = (uint32_t)(nty * L.WP); L.left = 2u * (uint32_t)(nty * L.WP); L.right = L.left + (uint32_t)(ntx * L.HP);
  L.words = L.right + (uint32_t)(ntx * L.HP);
  return L;
}

// NW waves per tile: 4 (throughput: a lane walks 16 rows, 7 workgroups per CU) or 16 (small submissions: 4 rows per lane --
// a one-frame call has two tiles per CU and is over when the slowest tile is, so the chain per wave is what counts).
// (register budget of k_cc_local: at least this many waves per SIMD)
#ifndef CC_LOCAL_MIN_WAVES
#define CC_LOCAL_MIN_WAVES 8
#endif
template <int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(CC_LOCAL_MIN_WAVES, 8))) void k_cc_local(const uint8_t* __restrict__ thr_all, uint32_t* __restrict__ label_all,
                                                  uint32_t* __restrict__ csize_all, uint32_t* __restrict__ roots_all,
                                                  uint32_t* __restrict__ perim_all, FrameCounters* __restrict__ counters, DetParams P) {
  // (the threshold tile is only read into registers right after the load; the link-request lists of the union pass take
  // over its space -- a barrier lies between)
  constexpr int ROWS = CC_T / NW;                                   // rows of a wave's strip
  constexpr int UROWS = ROWS < CC_UROWS ? ROWS : CC_UROWS;          // rows per batch of link requests
  constexpr int UREQ = UROWS * 2 * 64;                              // link requests of UROWS rows of one wave (at most two entries per pixel)
  __shared__ __attribute__((aligned(16))) uint8_t s_tile_or_requests[(CC_T * CC_T > NW * UREQ * 2) ? CC_T * CC_T : NW * UREQ * 2];
  uint8_t* const st = s_tile_or_requests;
  uint16_t* const s_ureq = reinterpret_cast<uint16_t*>(s_tile_or_requests);
  __shared__ uint32_t sl[CC_T * CC_T];
  // (with 4 waves the block is 20 KB of LDS to the byte -- eight blocks per CU, and eight waves per SIMD)
  const int frame = (int)blockIdx.z + P.frame0;
  const int X0 = blockIdx.x * CC_T, Y0 = blockIdx.y * CC_T;
  const int W = P.W, H = P.H;
  const uint8_t* thr = thr_all + (size_t)frame * H * P.WS;
  const int tid = threadIdx.x;

  if (tid < 256) {  // load the tile: 16 bytes per thread, out-of-image pixels become 127
    const int row = tid >> 2, seg = tid & 3;
    const int gy = Y0 + row, gx = X0 + seg * 16;
    uint32_t w[4] = {0x7F7F7F7Fu, 0x7F7F7F7Fu, 0x7F7F7F7Fu, 0x7F7F7F7Fu};
    if (gy < H && gx < W) {
      uint4 v = *reinterpret_cast<const uint4*>(thr + (size_t)gy * P.WS + gx);
      w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
      if (gx + 16 > W) {
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
          for (int b = 0; b < 4; b++)
            if (gx + 4 * j + b >= W) w[j] = (w[j] & ~(0xFFu << (8 * b))) | (0x7Fu << (8 * b));
      }
    }
    *reinterpret_cast<uint4*>(st + row * CC_T + seg * 16) = make_uint4(w[0], w[1], w[2], w[3]);
  }
  __syncthreads();

  const int lane = tid & 63, wv = tid >> 6;
  const int gx = X0 + lane;
  // Byte offset of the lane's pixel in the first row of the wave's strip: row k of the strip is this plus the CONSTANT k * 256, in
  // the parent array (an immediate offset of the LDS instruction) and as a label (one OR).  Every pass takes an opaque copy:
  // formed from the row index per row, the compiler kept the ROWS offsets in a register each from the first pass to the last.
  const uint32_t me4w = (uint32_t)(wv * ROWS * CC_T + lane) << 2;
  auto fresh = [](uint32_t x) { asm volatile("" : "+v"(x)); return x; };
  const bool src_ok = gx >= 1 && gx <= W - 2;  // this column may be a link source

  // The lane's column of the wave's ROWS rows stays in registers from here on, and so does the row above the strip; left and
  // right neighbours come over the DPP network (wave_shr:1 / wave_shl:1), not from the byte array again -- the three passes
  // below used to re-read it eight times per pixel.
  // (four rows to a register, read back with byte selects: sixteen separate registers cost the kernel a wave of occupancy)
  uint32_t vv4[(ROWS + 3) / 4];
#pragma unroll
  for (int j = 0; j < (ROWS + 3) / 4; j++) {
    uint32_t w = 0;
#pragma unroll
    for (int b = 0; b < 4 && 4 * j + b < ROWS; b++) w |= (uint32_t)st[(wv * ROWS + 4 * j + b) * CC_T + lane] << (8 * b);
    // (the bytes become class codes -- black 0, no class 2, white 3: bit 7 and bit 0 of the values 0 / 127 / 255 -- so that the
    // row passes compare against inline constants; 255 had to be put into a scalar register again for every row)
    w = ((w >> 7) & 0x01010101u) | ((w << 1) & 0x02020202u);
    asm volatile("" : "+v"(w));   // (opaque: keeps the compiler from carrying the bytes unpacked)
    vv4[j] = w;
  }
  constexpr uint32_t CLS_BLACK = 0u, CLS_NONE = 2u, CLS_WHITE = 3u;
  auto px = [&](int k) { return (vv4[k >> 2] >> (8 * (k & 3))) & 0xFFu; };   // class code of the lane's pixel in row k
  const uint32_t vtop = wv > 0 ? (uint32_t)st[(wv * ROWS - 1) * CC_T + lane] : 127u;
  // Class masks of the wave's rows (bit i = lane i) and of the row above the strip.  The link rules below are stated on these
  // 64-bit masks with SCALAR shifts and logic -- a row of 64 pixels per instruction -- instead of per pixel on the vector unit
  // (DPP moves of four neighbours and a dozen compares per pixel and row: the kernel is VALU-issue bound).
  // (the masks of a row are formed where they are used -- two compares per row and pass -- so that only two rows' worth of them
  // are live in scalar registers at a time)
  const unsigned long long SRC = __ballot(src_ok);              // columns that may be a link source (1 <= x <= W - 2)
  const unsigned long long LSRC = __ballot(gx - 1 >= 1);        // ... whose left neighbour is one
  const unsigned long long RSRC = __ballot(gx + 1 <= W - 2);    // ... whose right neighbour is one
  // (a lane's own bit of a uniform mask is the mask used as the execution mask -- inverse ballot, no vector instruction -- and
  // the number of set bits below the lane is v_mbcnt on the scalar mask)
  auto mine = [](unsigned long long m) { return __builtin_amdgcn_inverse_ballot_w64(m); };
  // (plus `base`: v_mbcnt adds its count to an accumulator operand, so a list position base + count costs no extra add)
  auto below_me = [](unsigned long long m, uint32_t base) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, base)); };
  // ---- 1. run labelling per row (wave masks, no atomics) --------------------------------------
  // The LAST pixel of a black or white run keeps the run's length (one byte per row, four rows to a register): pass 3 adds it
  // to the component's count from there, with no run geometry to work out again.
  static_assert(CC_T == 64, "a tile row is one wave wide; byte offsets split into row (>> 8) and column");
  uint32_t runlen[(ROWS + 3) / 4];
#pragma unroll
  for (int j = 0; j < (ROWS + 3) / 4; j++) runlen[j] = 0;
  const unsigned long long below = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
  const uint32_t strip4 = fresh((uint32_t)(wv * ROWS * CC_T) << 2);   // byte offset of the strip's first pixel
  uint32_t* const sl1 = lds_at(sl, fresh(me4w));
#pragma unroll
  for (int k = 0; k < ROWS; k++) {
    // link: same class as the left neighbour (inside the tile: the shift leaves bit 0 clear) and a valid source column.
    // Pixels without a class form runs like the others here: no rule below makes them a link source or partner, their entries are
    // chased by nobody, and the flatten pass overrides their label -- so the label store needs no case for them.
    const unsigned long long W_ = __ballot(px(k) == CLS_WHITE), B_ = __ballot(px(k) == CLS_BLACK), N_ = ~(W_ | B_);
    const unsigned long long L = ((W_ & (W_ << 1)) | (B_ & (B_ << 1)) | (N_ & (N_ << 1))) & SRC;
    const unsigned long long m = ~L & below;  // run starts at or below this lane (lane 0 always set)
    const int s = 63 - __clzll((long long)m);
    sl1[k * CC_T] = strip4 + (uint32_t)(k * CC_T * 4) + ((uint32_t)s << 2);   // byte offset of the run's first pixel
    const unsigned long long ends = (~(L >> 1) | (1ull << 63)) & ~N_;   // last pixels of the black and white runs
    runlen[k >> 2] |= (mine(ends) ? (uint32_t)(lane + 1 - s) : 0u) << (8 * (k & 3));
  }
  // (opaque: the compiler otherwise sees through the packing and carries the ROWS lengths in a register each)
#pragma unroll
  for (int j = 0; j < (ROWS + 3) / 4; j++) asm volatile("" : "+v"(runlen[j]));
  __syncthreads();

  CC_STOP_AT(1)   // (tools_hooks.h: nothing in the product build)
  // ---- 2. unions with the row above (only the first pixel of every overlap) -------------------
  // A lane has at most three links to the row above (up, up-left, up-right) and most lanes have none, so the link
  // requests of four rows at a time are compacted into a wave-private list (mask + popcount, no atomics) and the
  // unions -- root chases and atomicMin retries -- run DENSE over it, every lane on a real link.  (Calling the union
  // under the three conditions directly ran each call on the few lanes that had that link and as long as the
  // longest chase among them.)  Entry = pixel << 2 | partner (0: up, 1: up-left, 2: up-right).
  // The rules, per source pixel (x, y) of class c (x a source column, y > 0), on masks C = class c in this row, Cu = in the
  // row above:  up: Cu, unless the left neighbour is a source of the class with its own up link to the same run (C << 1, Cu << 1);
  // white only -- up-left: Wu << 1 and not Wu, unless the left neighbour is a white source; up-right: Wu >> 1, unless the
  // right neighbour is a source and the pixel above or the right neighbour is white (they carry the link).
  {
    uint16_t* ureq = s_ureq + wv * UREQ;
    const uint32_t me4 = fresh(me4w);
    unsigned long long Wu = __ballot(vtop == 255u), Bu = __ballot(vtop == 0u);   // the row above the one in hand
#pragma unroll
    for (int k0 = 0; k0 < ROWS; k0 += UROWS) {
      uint32_t nreq = 0;   // uniform
#pragma unroll
      for (int kk = 0; kk < UROWS; kk++) {
        const int k = k0 + kk;   // (compile-time after unrolling)
        const uint32_t e4 = me4 | (uint32_t)(k * CC_T * 4);   // the pixel's byte offset: the entry's upper 14 bits
        uint32_t v2 = px(k);
        asm volatile("" : "+v"(v2));   // (opaque copy: the compiler otherwise keeps pass 1's 2 x ROWS masks alive for this pass and spills them)
        const unsigned long long W_ = __ballot(v2 == CLS_WHITE), B_ = __ballot(v2 == CLS_BLACK);
        const unsigned long long rows_ok = SRC;   // (the tile's first row has no row above: vtop is 127 there, Wu = Bu = 0 and every rule below comes out empty)
        const unsigned long long m0 = ((W_ & Wu & ~((W_ << 1) & (Wu << 1) & LSRC)) | (B_ & Bu & ~((B_ << 1) & (Bu << 1) & LSRC))) & rows_ok;
        const unsigned long long m1 = W_ & (Wu << 1) & ~Wu & ~((W_ << 1) & LSRC) & rows_ok;
        const unsigned long long m2 = W_ & (Wu >> 1) & ~((Wu | (W_ >> 1)) & RSRC) & rows_ok;
        // One entry per pixel with a link -- its up link, else its up-left link, else its up-right one (up and up-left exclude
        // each other) -- and a second entry for the few pixels that have an up-right link besides: one compaction per row, and
        // the second one behind a scalar branch.
        const unsigned long long mp = m0 | m1 | m2, ms = m2 & (m0 | m1);
        uint32_t t = mine(m1) ? 1u : 2u;   // (two selects on the scalar masks; the nested conditional came out as branches)
        t = mine(m0) ? 0u : t;
        if (mine(mp)) ureq[below_me(mp, nreq)] = (uint16_t)(e4 | t);
        nreq += (uint32_t)__popcll(mp);
        if (ms) {
          if (mine(ms)) ureq[below_me(ms, nreq)] = (uint16_t)(e4 | 2u);
          nreq += (uint32_t)__popcll(ms);
        }
        Wu = W_; Bu = B_;
      }
      // (wave-private list: the wave's own LDS writes are visible to its later reads in program order)
      for (uint32_t i = (uint32_t)lane; i < nreq; i += 64) {
        const uint32_t q = ureq[i];
        const uint32_t p4 = q & ~3u, t = q & 3u;
        lds_union(sl, p4, p4 - 4u * CC_T - ((t & 1u) << 2) + ((t & 2u) << 1));
      }
    }
  }
  __syncthreads();

  CC_STOP_AT(2)
  // ---- 3. flatten into registers, then count pixels per root (one LDS atomic per run) ----------
  // (the chases of the wave's rows advance together, one hop of every row per step: ROWS independent loads in flight instead of
  // one chain after the other; hops past a root are harmless, it maps to itself)
  uint32_t root[ROWS];
  {
    uint32_t* const sl3 = lds_at(sl, fresh(me4w));
#pragma unroll
    for (int k = 0; k < ROWS; k++) root[k] = __hip_atomic_load(&sl3[k * CC_T], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (the pixel's own entry)
  }
#pragma unroll
  for (int h = 0; h < CC_FLATTEN_HOPS; h++) {
#pragma unroll
    for (int k = 0; k < ROWS; k++) root[k] = __hip_atomic_load(lds_at(sl, root[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  for (;;) {
    bool moved = false;
#pragma unroll
    for (int k = 0; k < ROWS; k++) {
      const uint32_t p = __hip_atomic_load(lds_at(sl, root[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      moved |= p != root[k];
      root[k] = p;
    }
    if (!__any(moved)) break;
  }
  // (root[] stays in byte offsets: the count pass addresses with it, the write pass splits it into row and column)
#pragma unroll
  for (int k = 0; k < ROWS; k++) root[k] = (px(k) == CLS_NONE) ? AT_NO_LABEL : root[k];
  __syncthreads();
  {
    uint32_t* const sl3 = lds_at(sl, fresh(me4w));
#pragma unroll
    for (int k = 0; k < ROWS; k++) sl3[k * CC_T] = 0;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < ROWS; k++) {
    const uint32_t len = (runlen[k >> 2] >> (8 * (k & 3))) & 0xFFu;   // (nonzero on the last pixel of a black or white run)
    if (len) atomicAdd(lds_at(sl, root[k]), len);
  }
  __syncthreads();

  CC_STOP_AT(3)
  // ---- 4. write global labels (index of the local root), local sizes at the roots, root list -------
  uint32_t* label = label_all + (size_t)frame * W * H;
  uint32_t* csize = csize_all + (size_t)frame * W * H;
  // No root list any more (round 6): a root keeps the size bit of its LOCAL pixel count whether or not its component touches the
  // tile's perimeter -- the count only grows when the border pass joins it to another tile's component, so the bit is final when
  // set -- and the passes behind the border pass (k_cc_sizes, k_cc_resolve) run over the roots that LOST a union there, which
  // k_cc_border lists as it makes them: the specks that touch a tile border without crossing it, half of a noisy frame's
  // perimeter roots, are final here and are never visited again.
  // (straight-line per row: every lane forms its label -- the select of "no label" included -- and the row goes out as one
  // predicated store; only the size and the list entry of a root, one lane per component, sit under a branch.  An early exit
  // from the unrolled row loop had the compiler build a state machine around every row.)
  {
    const bool col_ok = gx < W;
    const int gyw = Y0 + wv * ROWS;   // image row of the strip's first row
    uint32_t gi = (uint32_t)(gyw * W + gx);   // (W * H pixels of a frame index 32 bits: the labels are such indices)
    const uint32_t me4 = fresh(me4w);
    const uint32_t* const sl4 = lds_at(sl, me4);
    // the tile's perimeter for k_cc_border (CcPerim): the first / last row go out as one 256-byte store each, the first / last
    // column as one two-lane store per row (lane 0 to `left`, lane 63 to `right`)
    const CcPerim PL = cc_perim_layout(W, H);
    uint32_t* const perim = perim_all + (size_t)frame * PL.words;
    uint32_t* const pcol = perim + (lane == 0 ? PL.left + blockIdx.x * (uint32_t)PL.HP : PL.right + blockIdx.x * (uint32_t)PL.HP) + (uint32_t)gyw;
    const bool col_lane = lane == 0 || lane == 63;
#pragma unroll
    for (int k = 0; k < ROWS; k++, gi += (uint32_t)W) {
      const uint32_t rt = root[k];    // byte offset of the root pixel in the tile
      const uint32_t cs = sl4[k * CC_T];
      const bool isroot = rt == (me4 | (uint32_t)(k * CC_T * 4));   // (AT_NO_LABEL is no pixel's byte offset)
      // (24-bit multiply: one full-rate instruction; rows and the width are far below 2^24)
      uint32_t lab = __umul24((uint32_t)Y0 + (rt >> 8), (uint32_t)W) + (uint32_t)X0 + ((rt >> 2) & (CC_T - 1));
      {
        // (formed before the size bit goes into `lab`: bit 31 of a perimeter entry is the pixel's class)
        const uint32_t pe = rt == AT_NO_LABEL ? AT_NO_LABEL : (lab | (px(k) == CLS_WHITE ? 0x80000000u : 0u));
        if (k == 0 && wv == 0) perim[PL.top + blockIdx.y * (uint32_t)PL.WP + (uint32_t)gx] = pe;
        if (k == ROWS - CC_LAST_ROW_OFFSET(NW) && wv == NW - 1) perim[PL.bot + blockIdx.y * (uint32_t)PL.WP + (uint32_t)gx] = pe;   // (tools_hooks.h: 1)
        if (col_lane) pcol[k] = pe;
      }
      // (enough pixels inside this tile alone: the size bit is final whatever the border pass adds)
      if (isroot && (int)cs >= P.min_component_size) lab |= AT_LABEL_BIG;
      if (rt == AT_NO_LABEL) lab = AT_NO_LABEL;
      if (col_ok && gyw < H - k) {
        label[gi] = lab;
        if (isroot) csize[gi] = cs;
      }
    }
  }
}

// Link requests of the border pass: up to three (root, partner root) pairs per thread; a slot without a link holds
// AT_NO_LABEL.  Perimeter entries (CcPerim): AT_NO_LABEL = no class; otherwise bit 31 = white, the rest the pixel's tile-local root.
__device__ __forceinline__ bool pe_white(uint32_t e) { return e != AT_NO_LABEL && (e >> 31); }
__device__ __forceinline__ bool pe_same(uint32_t a, uint32_t b) {   // both have a class, and the same one
  return a != AT_NO_LABEL && b != AT_NO_LABEL && ((a ^ b) >> 31) == 0;
}
// Pixel (gx, gy) on a tile-top row: its up, up-left and up-right links, with the same first-overlap rules as the tile kernel.
// t* = entries of the row itself (left, own, right), b* = of the row above it (the last row of the tile row above).
__device__ __forceinline__ void cc_row_requests(int W, int gx, uint32_t tl, uint32_t tc, uint32_t tr, uint32_t bl, uint32_t bc, uint32_t br,
                                                uint32_t (&ra)[3], uint32_t (&rb)[3]) {
  if (gx < 1 || gx > W - 2 || tc == AT_NO_LABEL) return;
  const uint32_t me = tc & AT_LABEL_MASK;
  const bool left_src = gx - 1 >= 1;
  if (pe_same(bc, tc) && !(left_src && pe_same(tl, tc) && pe_same(bl, tc))) { ra[0] = me; rb[0] = bc & AT_LABEL_MASK; }
  if (pe_white(tc)) {
    if (pe_white(bl) && !pe_white(bc) && !(left_src && pe_white(tl))) { ra[1] = me; rb[1] = bl & AT_LABEL_MASK; }
    const bool right_src = gx + 1 <= W - 2;
    if (pe_white(br) && !(right_src && (pe_white(bc) || pe_white(tr)))) { ra[2] = me; rb[2] = br & AT_LABEL_MASK; }
  }
}

// Links across the vertical border between column gx - 1 (last of the left tile) and gx (first of the right tile) in
// row gy: the left link and the up-left link of pixel (gx, gy), and the up-right link of pixel (gx - 1, gy).  eL / eR = entries
// of the two pixels, eLu / eRu = of the pixels above them (tile-top rows take only the left link: the row pass owns every upward
// link of those rows).
__device__ __forceinline__ void cc_column_requests(int W, int gx, bool upward, uint32_t eL, uint32_t eR, uint32_t eLu, uint32_t eRu,
                                                   uint32_t (&ra)[3], uint32_t (&rb)[3]) {
  if (!upward) { eLu = AT_NO_LABEL; eRu = AT_NO_LABEL; }
  if (gx <= W - 2 && eR != AT_NO_LABEL) {               // (gx, gy) is a link source
    // (the same link one row up, with both pixels joined to their upper neighbours inside their tiles, already implies it)
    if (pe_same(eL, eR) && !(upward && pe_same(eLu, eR) && pe_same(eRu, eR))) { ra[0] = eR & AT_LABEL_MASK; rb[0] = eL & AT_LABEL_MASK; }
    if (upward && pe_white(eR) && pe_white(eLu) && !pe_white(eRu) && !pe_white(eL)) { ra[1] = eR & AT_LABEL_MASK; rb[1] = eLu & AT_LABEL_MASK; }
  }
  if (upward && pe_white(eL)) {                         // (gx - 1, gy) is a source (1 <= gx - 1 <= W - 2 always)
    const bool right_src = gx <= W - 2;
    if (pe_white(eRu) && !(right_src && (pe_white(eLu) || pe_white(eR)))) { ra[2] = eL & AT_LABEL_MASK; rb[2] = eRu & AT_LABEL_MASK; }
  }
}

// grid.x covers [border rows: nrows*W pixels][tile columns: ncols*H]
#ifndef CC_ILEAVE
#define CC_ILEAVE 256
#endif
template <bool PER_WAVE>
__global__ __launch_bounds__(256) void k_cc_border(const uint32_t* __restrict__ perim_all, uint32_t* __restrict__ label_all,
                                                   uint32_t* __restrict__ roots_all, FrameCounters* __restrict__ counters,
                                                   uint32_t bpf, uint32_t nframes, DetParams P) {
  uint32_t fr_, blk_;
  at_frame_block(blockIdx.x, bpf, nframes, CC_ILEAVE, &fr_, &blk_);
  const int frame = (int)fr_ + P.frame0;
  const int W = P.W, H = P.H;
  const CcPerim PL = cc_perim_layout(W, H);
  const uint32_t* perim = perim_all + (size_t)frame * PL.words;
  uint32_t* label = label_all + (size_t)frame * W * H;
  const int nrows = (H - 1) / CC_T;  // tile-top rows at y = 64, 128, ...
  const int ncols = (W - 1) / CC_T;  // tile-left columns at x = 64, 128, ...
  int i = (int)blk_ * 256 + threadIdx.x;
  __shared__ uint32_t s_lost, s_base;
  if (!PER_WAVE) {
    if (threadIdx.x == 0) s_lost = 0;
    __syncthreads();
  }
  uint32_t lost[3];
  uint32_t ra[3] = {AT_NO_LABEL, AT_NO_LABEL, AT_NO_LABEL}, rb[3] = {AT_NO_LABEL, AT_NO_LABEL, AT_NO_LABEL};
  if (i < nrows * W) {
    const int gx = i % W, ty = i / W + 1;
    const uint32_t* t = perim + PL.top + (uint32_t)ty * (uint32_t)PL.WP + (uint32_t)gx;
    const uint32_t* b = perim + PL.bot + (uint32_t)(ty - 1) * (uint32_t)PL.WP + (uint32_t)gx;
    // (x - 1 at gx = 0 and x + 1 at the array's end are never used by the rules: such a pixel is no link source)
    const uint32_t tc = t[0], bc = b[0];
    const uint32_t tl = gx > 0 ? t[-1] : AT_NO_LABEL, bl = gx > 0 ? b[-1] : AT_NO_LABEL;
    const uint32_t tr = gx + 1 < PL.WP ? t[1] : AT_NO_LABEL, br = gx + 1 < PL.WP ? b[1] : AT_NO_LABEL;
    cc_row_requests(W, gx, tl, tc, tr, bl, bc, br, ra, rb);
  } else {
    i -= nrows * W;
    if (i < ncols * H) {
      const int tx = i / H + 1, gy = i % H;
      const uint32_t* l = perim + PL.right + (uint32_t)(tx - 1) * (uint32_t)PL.HP + (uint32_t)gy;
      const uint32_t* r = perim + PL.left + (uint32_t)tx * (uint32_t)PL.HP + (uint32_t)gy;
      const bool upward = (gy % CC_T) != 0;
      cc_column_requests(W, tx * CC_T, upward, l[0], r[0], upward ? l[-1] : AT_NO_LABEL, upward ? r[-1] : AT_NO_LABEL, ra, rb);
    }
  }
  // The requests name tile-local ROOTS (the perimeter entries carry them), so the device-scope loads of the find start at the
  // roots.  Along a tile border most links join the SAME two tile-local roots again and again (the big components of a textured
  // background cross it dozens of times): a wave keeps one request per distinct pair of roots (a few leader rounds; what they do
  // not cover is simply linked twice).
#pragma unroll
  for (int k = 0; k < 3; k++) {
    uint32_t a = ra[k], b2 = rb[k];
    if (a != AT_NO_LABEL) {
      if (a > b2) { const uint32_t t = a; a = b2; b2 = t; }
      if (a == b2) a = b2 = AT_NO_LABEL;   // already the same tile-local root
    }
    unsigned long long todo = __ballot(a != AT_NO_LABEL);
    bool mine = a != AT_NO_LABEL;
    for (int round = 0; round < 6 && todo; round++) {
      const int leader = __ffsll((long long)todo) - 1;
      const uint32_t la = (uint32_t)__builtin_amdgcn_readlane((int)a, leader), lb = (uint32_t)__builtin_amdgcn_readlane((int)b2, leader);
      const bool same = a == la && b2 == lb;
      if (same && (int)lane_id() != leader) mine = false;
      todo &= ~__ballot(same);
    }
    lost[k] = mine ? glb_union(label, a, b2) : AT_NO_LABEL;
  }
  // The roots this block's unions turned into non-roots go to the frame's list -- these, and only these, are the tile-local roots
  // whose entry, size and size bit k_cc_sizes / k_cc_resolve still have to settle.  Small submissions append once per BLOCK: the
  // list's counter is one word per frame, and a one-frame submission has a thousand waves here (one add per wave and request slot
  // cost such a call 15 us).  Large ones -- their blocks in flight belong to different frames -- append once per WAVE and spare the
  // two barriers, behind which a block's waves wait for its slowest chase.
  const uint32_t mycnt = (lost[0] != AT_NO_LABEL ? 1u : 0u) + (lost[1] != AT_NO_LABEL ? 1u : 0u) + (lost[2] != AT_NO_LABEL ? 1u : 0u);
  uint32_t pos;
  if (PER_WAVE) {
    const uint32_t incl = wave_incl_scan(mycnt);
    const uint32_t wtotal = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    if (wtotal == 0) return;   // (uniform)
    uint32_t base = 0;
    if (lane_id() == 63) base = atomicAdd(&counters[frame].nroots, wtotal);
    pos = (uint32_t)__builtin_amdgcn_readlane((int)base, 63) + incl - mycnt;
  } else {
    pos = mycnt ? atomicAdd(&s_lost, mycnt) : 0u;
    __syncthreads();
    if (threadIdx.x == 0 && s_lost) s_base = atomicAdd(&counters[frame].nroots, s_lost);
    __syncthreads();
    pos += s_base;
  }
  if (mycnt) {
#pragma unroll
    for (int k = 0; k < 3; k++)
      if (lost[k] != AT_NO_LABEL) {
        if (pos < P.rcap) roots_all[(size_t)frame * P.rcap + pos] = lost[k];   // (at most one loser per perimeter root: rcap is their bound)
        pos++;
      }
  }
}

// one thread per listed root of the frame (the roots that lost a union in k_cc_border)
__global__ __launch_bounds__(256) void k_cc_sizes(uint32_t* __restrict__ label_all, uint32_t* __restrict__ csize_all,
                                                  const uint32_t* __restrict__ roots_all, const FrameCounters* __restrict__ counters,
                                                  DetParams P) {
  const int frame = (int)blockIdx.z + P.frame0;
  const size_t n = (size_t)P.W * P.H;
  uint32_t* label = label_all + (size_t)frame * n;
  uint32_t* csize = csize_all + (size_t)frame * n;
  const uint32_t* roots = roots_all + (size_t)frame * P.rcap;
  const uint32_t nroots = min(counters[frame].nroots, P.rcap);
  // A frame with textured background has a few giant components with tens of thousands of tile-local roots each: one
  // atomicAdd per root on the representative's size serialised on a single address (most of this kernel's time, 80 us
  // of a single frame's latency).  The adds of a wave are combined per representative first: leader rounds over the
  // distinct representatives of the wave (bounded; the rest add directly), so a hot representative sees one add per wave.
  const uint32_t stride = gridDim.x * 256;
  for (uint32_t i0 = blockIdx.x * 256; i0 < nroots; i0 += stride) {   // (uniform trip count per wave: ballots below)
    const uint32_t i = i0 + threadIdx.x;
    uint32_t r = AT_NO_LABEL, add = 0;
    if (i < nroots) {
      const uint32_t p = roots[i];   // (a root that lost a union: never its component's representative)
      uint32_t q;
      r = p;
      while ((q = glb_load(&label[r]) & AT_LABEL_MASK) != r) r = q;
      label[p] = r;  // every chain through p now ends in one more hop
      add = csize[p];
    }
    unsigned long long todo = __ballot(r != AT_NO_LABEL);
    for (int round = 0; round < 4 && todo; round++) {
      const int leader = __ffsll((long long)todo) - 1;
      const uint32_t rl = (uint32_t)__builtin_amdgcn_readlane((int)r, leader);
      const bool mine = r == rl;
      const unsigned long long grp = __ballot(mine);
      uint32_t v = mine ? add : 0u;
#define OP(Cc, Mm) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, Cc, Mm, 0xF, true);
      OP(0x111, 0xF) OP(0x112, 0xF) OP(0x114, 0xF) OP(0x118, 0xF) OP(0x142, 0xA) OP(0x143, 0xC)
#undef OP
      const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
      if ((int)lane_id() == leader) atomicAdd(&csize[rl], total);
      if (mine) r = AT_NO_LABEL;
      todo &= ~grp;
    }
    if (r != AT_NO_LABEL) atomicAdd(&csize[r], add);
  }
}

// Second pass over the list, after every k_cc_sizes thread has finished: where the representative's count has reached
// min_component_size, the listed root's entry becomes representative | AT_LABEL_BIG and the representative's own entry gets the
// bit (see the header of this file).  Plain stores: a listed root is visited by one thread, a representative's entry is only ever
// written with the one value r | AT_LABEL_BIG, and no thread of this kernel reads an entry another one writes (the listed roots
// and the representatives are disjoint sets).  Entries that need no bit stay as k_cc_sizes and k_cc_local left them.
__global__ __launch_bounds__(256) void k_cc_resolve(uint32_t* __restrict__ label_all, const uint32_t* __restrict__ csize_all,
                                                    const uint32_t* __restrict__ roots_all,
                                                    const FrameCounters* __restrict__ counters, DetParams P) {
  const int frame = (int)blockIdx.z + P.frame0;
  const size_t n = (size_t)P.W * P.H;
  uint32_t* label = label_all + (size_t)frame * n;
  const uint32_t* csize = csize_all + (size_t)frame * n;
  const uint32_t* roots = roots_all + (size_t)frame * P.rcap;
  const uint32_t nroots = min(counters[frame].nroots, P.rcap);
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < nroots; i += gridDim.x * 256) {
    const uint32_t p = roots[i];
    const uint32_t r = label[p];   // (k_cc_sizes left the representative there, without the bit)
    if ((int)csize[r] >= P.min_component_size) {
      label[p] = r | AT_LABEL_BIG;
      label[r] = r | AT_LABEL_BIG;
    }
  }
}

// Stage inspection only: representative per pixel (label[label[p]] after k_cc_sizes; idempotent).
__global__ __launch_bounds__(256) void k_cc_flatten(uint32_t* __restrict__ label_all, DetParams P) {
  const int frame = (int)blockIdx.z + P.frame0;
  const size_t n = (size_t)P.W * P.H;
  uint32_t* label = label_all + (size_t)frame * n;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t l = label[i];
  if (l == AT_NO_LABEL) return;
  // (idempotent under concurrent execution: every entry on a chain keeps pointing at an ancestor, with or without the bit)
  uint32_t r = l & AT_LABEL_MASK, q;
  while ((q = glb_load(&label[r]) & AT_LABEL_MASK) != r) r = q;
  if (r != l) label[i] = r;   // representative, without the size bit
}
