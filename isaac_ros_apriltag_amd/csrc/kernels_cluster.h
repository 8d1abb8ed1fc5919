// kernels_cluster.h -- S4 boundary-point extraction and clustering by component pair
// (inside cuAprilTagsDetect, reference src/apriltag_node.cpp:491-493).
//
//   k_points          every pixel of a 64x16 tile looks at its 4 forward neighbours (LDS halo tile of
//                     {value, label-if-component>=25}) and only notes which of them emit (a 16-bit mask per
//                     thread); the emitters are compacted into a block-wide LDS list, and everything that costs
//                     -- the per-block component-pair table in LDS (insert, count, rank from the counting
//                     atomic's return value), then one insert + one add per (block, pair) in the per-frame
//                     open-addressing hash table (64-bit CAS on the key, 32-bit add on the count, whose return
//                     value is the block's base rank in the cluster), then the staging stores -- runs DENSE over
//                     that list, every lane on a real emission.
//   k_cluster_select  keeps pairs with min_cluster_points <= count <= 3*(2W+2H), allocates their point
//                     ranges with one atomic per 1024-slot block (ballot/scan), emits the cluster list.
//   k_scatter         moves the staged points to range start + rank -- no atomics (order inside a
//                     cluster is fixed later by the slope sort, so arrival order never shows).
#pragma once
#include "common.h"

#define PT_TW 64
#define PT_TH 16
#define PT_LW (PT_TW + 2)
#define PT_LH (PT_TH + 1)

__device__ __forceinline__ uint32_t hash_slot(uint64_t key, uint32_t shift) {
  return (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> shift);
}

// returns slot or AT_INVALID_SLOT when the table is full
__device__ __forceinline__ uint32_t hash_insert(unsigned long long* hkeys, uint32_t hcap, uint32_t hshift, uint64_t key) {
  uint32_t h = hash_slot(key, hshift);
  for (uint32_t probe = 0; probe < hcap; probe++) {
    unsigned long long cur = __hip_atomic_load(&hkeys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == key) return h;
    if (cur == AT_EMPTY_KEY) {
      unsigned long long old = atomicCAS(&hkeys[h], AT_EMPTY_KEY, (unsigned long long)key);
      if (old == AT_EMPTY_KEY || old == key) return h;
    }
    h = (h + 1) & (hcap - 1);
  }
  return AT_INVALID_SLOT;
}

// the same, with the load of the home slot already made (`cur0`: what it held then -- the table only ever turns empty slots into
// keys, so a key seen there is final and an empty slot is settled by the CAS)
__device__ __forceinline__ uint32_t hash_insert_probed(unsigned long long* hkeys, uint32_t hcap, uint32_t hshift, uint64_t key, unsigned long long cur0) {
  uint32_t h = hash_slot(key, hshift);
  if (cur0 == key) return h;
  if (cur0 == AT_EMPTY_KEY) {
    unsigned long long old = atomicCAS(&hkeys[h], AT_EMPTY_KEY, (unsigned long long)key);
    if (old == AT_EMPTY_KEY || old == key) return h;
  }
  h = (h + 1) & (hcap - 1);
  for (uint32_t probe = 1; probe < hcap; probe++) {
    unsigned long long cur = __hip_atomic_load(&hkeys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == key) return h;
    if (cur == AT_EMPTY_KEY) {
      unsigned long long old = atomicCAS(&hkeys[h], AT_EMPTY_KEY, (unsigned long long)key);
      if (old == AT_EMPTY_KEY || old == key) return h;
    }
    h = (h + 1) & (hcap - 1);
  }
  return AT_INVALID_SLOT;
}

#define PT_WHITE 0x80000000u   // slab word of k_points: class bits (white / black; neither: not in a counted component) ...
#define PT_BLACK 0x40000000u
#define PT_REP_MASK 0x3FFFFFFFu   // ... above the component's representative (a pixel index: images below 2^30 pixels)
#define PT_TB 256  // entries of the per-block component-pair table
#ifndef PT_ELIST
#define PT_ELIST 2048   // emissions of one tile kept in the LDS list of pass 2 (the tile has 1024 pixels; tools builds shrink it
                        // to drive the long-record path of the emissions beyond the list)
#endif

// per-block table in LDS: returns entry index or -1 when full
// (slot = top byte of a 32-bit multiplicative hash of the two labels: two quarter-rate multiplies instead of the four
// of the frame table's 64-bit golden-ratio multiply, paid here once per emitted point.  Two full-rate 24-bit multiplies with
// 24-bit constants spread the keys so much worse that the kernel went from 3.3 to 4.5 ms.)
__device__ __forceinline__ uint32_t ltab_hash(uint64_t key) {
  const uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
  return (lo * 0x9E3779B1u + hi * 0x85EBCA77u) >> 24;
}
__device__ __forceinline__ int ltab_insert(unsigned long long* tkey, uint64_t key) {
  uint32_t h = ltab_hash(key);
  for (int probe = 0; probe < PT_TB; probe++) {
    const unsigned long long cur = __hip_atomic_load(&tkey[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // ds_read_b64 (volatile: a flat load)
    if (cur == key) return (int)h;
    if (cur == AT_EMPTY_KEY) {
      const unsigned long long old = atomicCAS(&tkey[h], AT_EMPTY_KEY, (unsigned long long)key);
      if (old == AT_EMPTY_KEY || old == key) return (int)h;
    }
    h = (h + 1) & (PT_TB - 1);
  }
  return -1;
}
__device__ __forceinline__ int ltab_find(const unsigned long long* tkey, uint64_t key) {
  uint32_t h = ltab_hash(key);
  for (int probe = 0; probe < PT_TB; probe++) {
    const unsigned long long cur = tkey[h];
    if (cur == key) return (int)h;
    if (cur == AT_EMPTY_KEY) return -1;
    h = (h + 1) & (PT_TB - 1);
  }
  return -1;
}

// Where the kernel's time goes (256 sigma-2 1080p frames, 3.97 ms; builds that stop after a phase, and builds with one
// operation taken out): tile load 0.85 ms, emission tests + scan 0.43, list 0.37, pass 2 1.0 (of which the returning
// same-address LDS add 0.5, the table insert 0.23), frame table 0.8 (two or three dependent L2 round trips per block, with
// the block's other waves at the barrier), emit + staging stores 0.6.  Grouping a wave's lanes by 64-bit KEY with ballots, BEFORE
// the insert, so that one lane per group inserts and adds (2 / 4 / 6 leader rounds) measured 4.00 / 4.09 / 4.05 ms: the rounds
// cost what the serialised atomic does.  ONE round on the table ENTRY after the insert (pass 2 below) is cheap enough:
// 3.62 -> 3.49 ms.  (The cycle shares of the profiling build put 60 % on the last two phases: its per-phase global atomics are
// waited for there.)
// Boundary points of one 64x16 tile.  Points are counted per component pair in a per-block LDS table
// first, so that the per-frame hash table sees ONE insert and ONE atomicAdd per (block, pair); the
// value the add returns is the block's base rank inside the cluster, so every staged point carries its
// final position (hoff[slot] + rank) and the scatter pass needs no atomics at all.
__global__ __launch_bounds__(256) void k_points(const uint8_t* __restrict__ thr_all, const uint32_t* __restrict__ label_all,
                                                unsigned long long* __restrict__ hkeys_all,
                                                uint32_t* __restrict__ hcnt_all, uint32_t* __restrict__ stage_all,
                                                uint2* __restrict__ bhdr_all, uint2* __restrict__ btab_all, uint4* __restrict__ long_all,
                                                FrameCounters* __restrict__ counters,
                                                unsigned long long* __restrict__ prof, uint32_t gx_tiles, uint32_t gy_tiles, uint32_t nframes,
                                                DetParams P) {
  PT_HOOKS_DECL   // (tools_hooks.h: measurement hooks, nothing in the product build)
  // tile + halo, one word per pixel: representative of the pixel's component if it is large enough, with the pixel's class in
  // the two top bits -- white 10, black 01 -- and 0 = no component that counts (value 127, or too small).  Two pixels lie on
  // opposite sides of an edge between counted components exactly when the XOR of their words has BOTH top bits set, so an
  // emission test is one XOR and one compare on this one array (a separate byte array of the values cost a second LDS read
  // per test, and its four-pixels-per-bank byte reads were most of the kernel's LDS bank conflicts; "labelled?" as compares
  // of their own, under branches, were a third of the test pass's instructions).
  __shared__ uint32_t slab[PT_LH * PT_LW];
  __shared__ uint32_t sscan[4];
  __shared__ uint32_t sbase;
  __shared__ unsigned long long tkey[PT_TB];
  __shared__ uint32_t tcnt[PT_TB];
  __shared__ uint32_t elist[PT_ELIST];
#ifndef PT_ILEAVE
#define PT_ILEAVE 256
#endif
  uint32_t fr_, blk_;
  at_frame_block(blockIdx.x, gx_tiles * gy_tiles, nframes, PT_ILEAVE, &fr_, &blk_);
  const int frame = (int)fr_ + P.frame0;
  const int bx_ = (int)(blk_ % gx_tiles), by_ = (int)(blk_ / gx_tiles);
  const int W = P.W, H = P.H;
  const size_t npx = (size_t)W * H;
  const uint8_t* thr = thr_all + (size_t)frame * H * P.WS;
  const uint32_t* label = label_all + (size_t)frame * npx;
  const int X0 = bx_ * PT_TW, Y0 = by_ * PT_TH;
  const int tid = threadIdx.x;

  // An INTERIOR tile -- all of its 17 x 66 entries inside the image, every pixel of it a valid emission source whose left
  // neighbour is one too (nine tiles in ten at 1080p) -- takes the straight-line forms of the tile load and of the emission tests
  // below: no per-entry bounds tests, row addresses as a uniform base plus the lane's offset (the general form pays two 64-bit
  // multiply-adds per entry), the halo column with wave 1 and row 16 with wave 0 (five entries per thread instead of six in the
  // wave the barrier waits for).
#ifndef PT_INTERIOR
#define PT_INTERIOR 1
#endif
  const bool interior = PT_INTERIOR && bx_ >= 1 && by_ >= 1 && X0 + PT_TW + 1 <= W && Y0 + PT_TH + 1 <= H;
  if (interior) {
    const int lane_ = tid & 63;
    const int wvu = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint8_t* const t0 = thr + ((uint32_t)Y0 * (uint32_t)P.WS + (uint32_t)X0);                       // (uniform)
    const char* const l0 = reinterpret_cast<const char*>(label + ((uint32_t)Y0 * (uint32_t)W + (uint32_t)X0));
    const char* const lab0 = reinterpret_cast<const char*>(label);
    const uint32_t WS_ = (uint32_t)P.WS, W4 = (uint32_t)W * 4u;
    uint32_t v[5], l[5], r[5];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const uint32_t row = (uint32_t)(wvu + 4 * e);
      v[e] = (t0 + row * WS_)[lane_];
      l[e] = *reinterpret_cast<const uint32_t*>(l0 + row * W4 + (uint32_t)lane_ * 4u);
    }
    int si4 = -1;
    v[4] = 127; l[4] = AT_NO_LABEL;
    if (wvu == 0) {
      v[4] = (t0 + (uint32_t)PT_TH * WS_)[lane_];
      l[4] = *reinterpret_cast<const uint32_t*>(l0 + (uint32_t)PT_TH * W4 + (uint32_t)lane_ * 4u);
      si4 = PT_TH * PT_LW + lane_ + 1;
    } else if (wvu == 1 && lane_ < 2 * PT_LH) {
      const uint32_t ly = (uint32_t)lane_ >> 1;
      const int col = (lane_ & 1) ? PT_TW : -1;
      v[4] = t0[(int)(ly * WS_) + col];
      l[4] = *reinterpret_cast<const uint32_t*>(l0 + (int)(ly * W4) + col * 4);
      si4 = (int)ly * PT_LW + ((lane_ & 1) ? PT_LW - 1 : 0);
    }
    // (inside the image a pixel has no label exactly when its value is 127; (l << 2) drops the flag bit of a root's own entry)
#pragma unroll
    for (int e = 0; e < 5; e++) {
      r[e] = 0;
      if (l[e] != AT_NO_LABEL) r[e] = *reinterpret_cast<const uint32_t*>(lab0 + (l[e] << 2));
    }
#pragma unroll
    for (int e = 0; e < 4; e++)
      slab[(wvu + 4 * e) * PT_LW + lane_ + 1] = (r[e] >> 31) ? ((r[e] & PT_REP_MASK) | (v[e] == 255u ? PT_WHITE : PT_BLACK)) : 0u;
    if (si4 >= 0) slab[si4] = (r[4] >> 31) ? ((r[4] & PT_REP_MASK) | (v[4] == 255u ? PT_WHITE : PT_BLACK)) : 0u;
  } else {
    // tile + halo: 17 rows x 66 columns.  A wave takes rows wv, wv + 4, ... with lane = column (64 coalesced entries per
    // row, no division to find an entry's place), the 2 x 17 halo entries go to the first 34 threads.  Two dependent loads
    // per entry (pixel -> tile-local root l, then label[l] = representative | size bit, kernels_cc.h), issued level by level
    // for all of a thread's entries so that their latencies overlap instead of adding up.
    constexpr int NR = (PT_LH + 3) / 4;   // rows per wave (5; the last one exists for wave 0 only)
    const int lane_ = tid & 63, wv_ = tid >> 6;
    uint32_t v[NR + 1], l[NR + 1], r[NR + 1];
    int si[NR + 1];                        // slab index of the entry, -1: none
    const int gxm = X0 + lane_;
#pragma unroll
    for (int e = 0; e <= NR; e++) {
      int ly, gx;
      if (e < NR) { ly = wv_ + 4 * e; gx = gxm; si[e] = ly < PT_LH ? ly * PT_LW + lane_ + 1 : -1; }
      else { ly = tid >> 1; gx = (tid & 1) ? X0 + PT_TW : X0 - 1; si[e] = tid < 2 * PT_LH ? ly * PT_LW + ((tid & 1) ? PT_LW - 1 : 0) : -1; }
      const int gy = Y0 + ly;
      v[e] = 127; l[e] = AT_NO_LABEL; r[e] = 0;
      if (si[e] >= 0 && gx >= 0 && gx < W && gy < H) {
        v[e] = thr[(uint32_t)gy * (uint32_t)P.WS + (uint32_t)gx];
        l[e] = label[(uint32_t)gy * (uint32_t)W + (uint32_t)gx];
      }
    }
#pragma unroll
    for (int e = 0; e <= NR; e++)
      if (v[e] != 127 && l[e] != AT_NO_LABEL) r[e] = label[l[e] & AT_LABEL_MASK];
#pragma unroll
    for (int e = 0; e <= NR; e++) {
      const uint32_t lab = (r[e] >> 31) ? ((r[e] & PT_REP_MASK) | (v[e] == 255u ? PT_WHITE : PT_BLACK)) : 0u;
      if (si[e] >= 0) slab[si[e]] = lab;
    }
  }
  tkey[tid] = AT_EMPTY_KEY;
  tcnt[tid] = 0;
  __syncthreads();
  PT_TICK(0)
  PT_STOP_AT(1, (void)0)

  const int lx = tid & 63;
  const int gx = X0 + lx;
  // pass 1: which (pixel, direction) of this thread emits -- a 16-bit mask and a count, nothing else.  Directions 0..3 =
  // (1,0), (0,1), (-1,1), (1,1).  The component-pair
  // table is NOT touched here: about one test in five emits, so table work inside this loop runs on a fifth of the
  // lanes and as long as the busiest lane; it is done on the compacted list instead (pass 2), where every lane has an
  // emission.
  // (History of this pass: aggregating the table inserts and counter atomics over the wave -- one leader per distinct
  // key -- was measured slower, 9.6 vs 6.1 ms: a 64-pixel row step meets too many distinct component pairs.  Taking
  // the rank from the counting atomic's return value INSIDE this sparse loop was slower too, 7.8 ms.)
  // (straight-line: every lane reads its pixel's six words and forms the four tests; validity of the source pixel is two masks)
  // (bit d * 4 + k of the mask = direction d of the thread's k-th pixel, tile pixel tid + 256 k: the list record of an emission --
  // pixel | direction << 10 -- is tid + (bit << 8))
  uint32_t emask = 0;
  {
    auto opposite = [](uint32_t a, uint32_t b) { return (a ^ b) > 0xBFFFFFFFu; };   // one white, one black, both counted
    if (interior) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int c = ((tid >> 6) + 4 * k) * PT_LW + lx + 1;
        const uint32_t s0 = slab[c], s_l = slab[c - 1], s_r = slab[c + 1];
        const uint32_t s_dl = slab[c + PT_LW - 1], s_d = slab[c + PT_LW], s_dr = slab[c + PT_LW + 1];
        const uint32_t e0 = opposite(s0, s_r) ? 1u : 0u, e1 = opposite(s0, s_d) ? 16u : 0u;
        const uint32_t e2 = (!opposite(s_l, s_d) && opposite(s0, s_dl)) ? 256u : 0u, e3 = opposite(s0, s_dr) ? 4096u : 0u;
        emask |= (e0 | e1 | e2 | e3) << k;
      }
    } else {
    const bool gx_ok = gx >= 1 && gx <= W - 2;
    const bool left_is_source = gx - 1 >= 1;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int ly = (tid >> 6) + 4 * k;
      const int gy = Y0 + ly;
      const bool ok = gx_ok && gy >= 1 && gy <= H - 2;
      const int c = ly * PT_LW + lx + 1;
      const uint32_t s0 = slab[c], s_l = slab[c - 1], s_r = slab[c + 1];
      const uint32_t s_dl = slab[c + PT_LW - 1], s_d = slab[c + PT_LW], s_dr = slab[c + PT_LW + 1];
      // upstream's connected_last: the left neighbour (a valid source itself) emitted its (1,1) point,
      // which is this pixel's (-1,1) half-pixel location
      const bool left_emits = left_is_source && opposite(s_l, s_d);
      const uint32_t e0 = (ok && opposite(s0, s_r)) ? 1u : 0u, e1 = (ok && opposite(s0, s_d)) ? 16u : 0u;
      const uint32_t e2 = (ok && !left_emits && opposite(s0, s_dl)) ? 256u : 0u, e3 = (ok && opposite(s0, s_dr)) ? 4096u : 0u;
      emask |= (e0 | e1 | e2 | e3) << k;
    }
    }
  }
  const uint32_t cnt = (uint32_t)__popc(emask);
  uint32_t total;
  const uint32_t off = block_excl_scan256(cnt, sscan, &total);  // contains __syncthreads
  PT_TICK(1)
  const uint32_t bpf = gx_tiles * gy_tiles;   // blocks (tiles) per frame
  if (total == 0) {
    if (tid == 0) bhdr_all[(size_t)frame * bpf + blk_] = make_uint2(0u, 0u);
    return;
  }
  PT_STOP_AT(2, stage_all[(size_t)frame * P.pcap + (blk_ % 64u) * 256 + tid] = emask ^ off)
  unsigned long long* hkeys = hkeys_all + (size_t)frame * P.hcap;
  uint32_t* hcnt = hcnt_all + (size_t)frame * P.hcap;
  if (tid == 0) sbase = atomicAdd(&counters[frame].npoints_raw, total);
  // the emitters go to a block-wide list in LDS at the thread's scan offset: record = pixel (10 bits) | direction (2);
  // emissions beyond the list capacity (more than two per pixel of the tile on average) are kept by their thread
  {
    // (the number of records this thread writes is known up front -- its emissions, cut where the list ends --, so a trip of the
    // loop is find-first-set, record, clear, store: no test of the mask or of the list's end per trip)
    const uint32_t room = off < PT_ELIST ? PT_ELIST - off : 0u;
    uint32_t m = emask;
    uint32_t* dst = elist + off;
    for (uint32_t n = cnt < room ? cnt : room; n; n--) {
      const uint32_t sidx = (uint32_t)__builtin_ctz(m);   // (m != 0: n counts its set bits)
      m &= m - 1;
      *dst++ = (uint32_t)tid | (sidx << 8);
    }
  }
  __syncthreads();
  PT_TICK(2)
  PT_STOP_AT(4, stage_all[(size_t)frame * P.pcap + (blk_ % 64u) * 256 + tid] = emask ^ off ^ elist[tid])
  // pass 2, DENSE over the list (entry q belongs to thread q mod 256): pair key -> block table entry e (one insert), the
  // emission's rank inside its (block, pair) group from the value the counting atomic returns -- on a full wave of real
  // emissions the returning LDS atomic costs what a leader loop over the wave's distinct entries does, and the insert is
  // paid once per emission instead of once per (pixel, direction) slot -- and, with both in hand, the staging record itself:
  // the stores of a wave are 64 consecutive words.  (The records used to go back to the list for a third pass behind a
  // barrier, which read the tile again for the sign bit: 0.7 ms of the kernel for a copy.)
  // A staging record is ONE word -- block-table entry (8 bits) | rank inside the (block, pair) group (11) | pixel of the tile
  // (10) | direction (2) | sign of the value step (1) -- because everything else k_scatter needs is per block: the tile origin
  // follows from the block index, the pair-table slot and the block's base rank inside the cluster from the table entry (btab),
  // the range of the block's records from its header (bhdr).  (8-byte records {slot | rank, packed point} were 40 % more bytes
  // through the staging buffer and back.)  The few emissions without a table entry -- block table full, or beyond the list --
  // take their slot and rank from the frame table one by one and go to a side list of long records.
  const uint32_t base = sbase;
  if (base + total > P.pcap) {
    if (tid == 0) atomicOr(&counters[frame].flags, 0x1u);
  }
  uint32_t* stage = stage_all + (size_t)frame * P.pcap;
  if (tid == 0) bhdr_all[(size_t)frame * bpf + blk_] = make_uint2(base, base + total > P.pcap ? (base < P.pcap ? P.pcap - base : 0u) : total);
  // an emission that is not counted in the block table: straight to the frame table and to the long records; its staging word
  // says "not here" (entry 255: k_scatter skips it)
  auto emit_long = [&](uint32_t pixd, uint32_t pos) {
    const int ly = (int)(pixd & 1023u) >> 6, plx = (int)(pixd & 63u), d = (int)((pixd >> 10) & 3u);
    const int c = ly * PT_LW + plx + 1;
    const int ddx = (d == 2) ? -1 : (d == 1 ? 0 : 1), ddy = (d == 0) ? 0 : 1;
    const uint32_t s0 = slab[c], s1 = slab[c + ddy * PT_LW + ddx];
    const uint32_t r0 = s0 & PT_REP_MASK, r1 = s1 & PT_REP_MASK;
    const uint64_t key = r0 < r1 ? ((uint64_t)r0 << 32) | r1 : ((uint64_t)r1 << 32) | r0;
    const uint32_t slot = hash_insert(hkeys, P.hcap, P.hshift, key);
    if (slot == AT_INVALID_SLOT) atomicOr(&counters[frame].flags, 0x2u);
    else {
      const uint32_t rk = atomicAdd(&hcnt[slot], 1u);
      const uint32_t li = atomicAdd(&counters[frame].nlong, 1u);
      const int v0 = (s0 >> 31) ? 255 : 0, v1 = (s1 >> 31) ? 255 : 0;
      if (li < P.lcap) long_all[(size_t)frame * P.lcap + li] = make_uint4(slot, rk, pack_point(2 * (X0 + plx) + ddx, 2 * (Y0 + ly) + ddy, ddx * (v1 - v0), ddy * (v1 - v0)), 0u);
      else atomicOr(&counters[frame].flags, 0x1u);   // (the point buffers grow together)
    }
    if (pos < P.pcap) __builtin_nontemporal_store(0xFFFFFFFFu, stage + pos);
  };
  const uint32_t nlist = total < PT_ELIST ? total : PT_ELIST;
  for (uint32_t q = tid; q < nlist; q += 256) {
    const uint32_t rec = elist[q];
    // slab index of the pixel: row * 66 + column + 1 = pixel + 2 * row + 1; of its neighbour: + 1, 66, 65, 67 by direction
    const uint32_t pix = rec & 1023u;
    const uint32_t c = pix + 2u * (pix >> 6) + 1u;
    static_assert(PT_LW == 66, "neighbour offsets of pass 2");
    const uint32_t dn = (0x43414201u >> ((rec >> 7) & 24u)) & 0xFFu;
    const uint32_t s0 = slab[c];
    const uint32_t r0 = s0 & PT_REP_MASK, r1 = slab[c + dn] & PT_REP_MASK;
    const uint64_t key = ((uint64_t)min(r0, r1) << 32) | max(r0, r1);
    const int e = ltab_insert(tkey, key);
    uint32_t ee = 255u, rk = 0u;
#ifndef PT_NO_LEADER_ADD
    // Most emissions of a wave's trip belong to ONE pair (the background's two big components meet in almost every tile):
    // 64 lanes adding 1 to the same counter serialise in the LDS pipeline (the kernel's "bank conflicts" are these
    // same-address atomics).  The lanes that share the first active lane's entry take their ranks from ONE add of their
    // number; the others add for themselves.
    const int e_first = __builtin_amdgcn_readfirstlane(e);
    const bool with_first = e == e_first && e >= 0 && e < 255;
    const unsigned long long gm = __ballot(with_first);
    if (with_first) {
      const int leader = (int)__ffsll((long long)gm) - 1;
      uint32_t gbase = 0;
      if ((int)(tid & 63) == leader) gbase = atomicAdd(&tcnt[e], (uint32_t)__popcll(gm));
      gbase = (uint32_t)__builtin_amdgcn_readlane((int)gbase, leader);
      ee = (uint32_t)e;
      rk = gbase + (uint32_t)__popcll(gm & ((1ull << (tid & 63)) - 1ull));
    } else
#endif
    if (e >= 0 && e < 255) { ee = (uint32_t)e; rk = atomicAdd(&tcnt[e], 1u); }
    if (ee != 255u) {
      // (pixel | direction << 10 of the list record are the word's bits 19..30 as they stand; value step v1 - v0 = +-255: bit 31
      // set when it is negative, v0 white.  Written once, read once by k_scatter two kernels later: non-temporal stores keep it
      // out of the caches' way.)
      const uint32_t w = ee | (rk << 8) | ((rec & 0xFFFu) << 19) | (s0 & 0x80000000u);
      if (base + q < P.pcap) __builtin_nontemporal_store(w, stage + base + q);
    } else emit_long(rec, base + q);
  }
  PT_TICK(3)
  PT_STOP_AT(3, stage_all[(size_t)frame * P.pcap + (blk_ % 64u) * 256 + tid] = emask ^ off ^ elist[tid] ^ sbase)
#ifndef PT_EARLY_PROBE
#define PT_EARLY_PROBE 1
#endif
  // The frame-table phase at the end of the block is a chain of two or three dependent round trips to the memory side per table
  // entry (0.37 ms of the kernel: a build that stops before it).  Its first one -- the load of the key's home slot -- is issued
  // HERE, as soon as this wave is through its share of the list, for the entry the thread will own (other waves may still add
  // entries: those are probed at the end as before), and is in flight while the wave waits at the barrier below.
  unsigned long long pf_key = AT_EMPTY_KEY, pf_cur = 0;
  if (PT_EARLY_PROBE) {
    pf_key = __hip_atomic_load(&tkey[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (pf_key != AT_EMPTY_KEY) pf_cur = __hip_atomic_load(&hkeys[hash_slot(pf_key, P.hshift)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (off + cnt > PT_ELIST) {   // this thread's emissions beyond the list
    uint32_t q = off, m = emask;
    while (m) {
      const int sidx = __ffs((int)m) - 1;
      m &= m - 1;
      if (q >= PT_ELIST) emit_long((uint32_t)tid + ((uint32_t)sidx << 8), base + q);
      q++;
    }
  }
  // the block table's counts are complete.  (An LDS-only barrier: nothing global is handed from thread to thread across it, and
  // __syncthreads' fence would wait for the probe above -- every outstanding global access -- in front of the barrier, in the
  // wave that arrives last as in the others.)
  if (PT_EARLY_PROBE) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
  } else {
    __syncthreads();
  }
  PT_TICK(4)
  PT_STOP_AT(5, (void)0)
  {
    // LAST, with nothing waiting for it: one global insert + one global add per distinct pair of this block; the add's
    // return value is the base rank of the block's points inside the cluster.  Slot and base go straight to the block's
    // table for k_scatter -- the staging words above do not depend on them, so the two or three dependent L2 round trips
    // of this phase no longer stand between the block's other waves and their stores (it used to sit before pass 3,
    // behind a barrier: 0.8 ms of the kernel).
    const unsigned long long key = tkey[tid];
    if (key != AT_EMPTY_KEY) {
      const uint32_t slot = (PT_EARLY_PROBE && key == pf_key) ? hash_insert_probed(hkeys, P.hcap, P.hshift, key, pf_cur)
                                                              : hash_insert(hkeys, P.hcap, P.hshift, key);
      uint32_t kbase = 0;
      if (slot != AT_INVALID_SLOT) kbase = atomicAdd(&hcnt[slot], tcnt[tid]);
      else atomicOr(&counters[frame].flags, 0x2u);
      btab_all[((size_t)frame * bpf + blk_) * PT_TB + tid] = make_uint2(slot, kbase);   // (only used entries are ever written or read)
    }
  }
  PT_TICK(5)
#undef PT_TICK
}

// Work lists of the quad fit: the kept clusters of all frames of the submission, bucketed by size class
// (class c holds lo[c] < count <= hi[c]).  An item is (frame << P.wshift) | cluster index (wshift >= 16: DetParams).  k_cluster_select appends them as it
// creates the cluster records; appends are aggregated per block in LDS, so every class counter sees one global atomic per
// block.  (A separate k_worklist pass over the records cost a launch and a round trip through them.)
#define FQ_NCLS 7
#define FQ_C0 2     // classes 0 .. FQ_C0 - 1: k_fit_small (K = 2, 4); FQ_C0 ..: k_fit_quads (64 ... 1024 threads)
struct FqWorkLayout {
  int lo[FQ_NCLS], hi[FQ_NCLS];
  uint32_t off[FQ_NCLS], cap[FQ_NCLS];   // item range of class c inside the work array
};

// One thread per four pair-table slots, SEL_CHUNKS chunks of 1024 slots per block (`nchunks` of them used: a large submission
// takes all four, so that the shared class counters of the work lists see a quarter of the blocks' atomics -- every
// block of every frame adds to the same five words; a small one takes one chunk per block and keeps the frame's chunks
// side by side).
#define SEL_CHUNKS 4
__global__ __launch_bounds__(256) void k_cluster_select(unsigned long long* __restrict__ hkeys_all,
                                                        uint32_t* __restrict__ hcnt_all, uint32_t* __restrict__ hoff_all,
                                                        ClusterRec* __restrict__ clusters_all,
                                                        FrameCounters* __restrict__ counters, uint32_t* __restrict__ work,
                                                        uint32_t* __restrict__ work_n, FqWorkLayout L, int nchunks, DetParams P) {
  // 1024 table slots per chunk, four consecutive ones per thread (16-byte loads and stores; hcap is a power of two >= 256)
  __shared__ uint32_t wsum[4], wcnt[4];
  __shared__ uint32_t s_pbase, s_cbase;
  __shared__ uint32_t s_wcnt[FQ_NCLS], s_wbase[FQ_NCLS];
  if (threadIdx.x < FQ_NCLS) s_wcnt[threadIdx.x] = 0;   // (visible after the first scan's barrier below)
  const int frame = (int)blockIdx.z + P.frame0;
  // a frame whose point staging overflowed has pair counts that exceed what was staged: it yields no
  // clusters at all (the overflow bit is reported), never an out-of-range range
  const bool frame_ok = (counters[frame].flags & 0x1u) == 0;
  const int lane = lane_id(), wv = threadIdx.x >> 6;
  // work item of a kept cluster: size class, rank among the block's items of the class, cluster index
  int wcls[SEL_CHUNKS][4];
  uint32_t wrank[SEL_CHUNKS][4], wci[SEL_CHUNKS][4];
#pragma unroll
  for (int ch = 0; ch < SEL_CHUNKS; ch++) {
#pragma unroll
    for (int j = 0; j < 4; j++) { wcls[ch][j] = -1; wrank[ch][j] = 0; wci[ch][j] = 0; }
    if (ch >= nchunks) continue;   // (uniform)
    if (ch) __syncthreads();       // the previous chunk's scan scratch has been read
    const uint32_t slot0 = ((uint32_t)blockIdx.x * (uint32_t)nchunks + (uint32_t)ch) * 1024u + threadIdx.x * 4;
    const bool in_range = slot0 < P.hcap;
    const size_t hi = (size_t)frame * P.hcap + (in_range ? slot0 : 0);
    // a slot holds a key exactly when its count is non-zero (every insert is followed by an add of at least one point),
    // so the 8-byte keys of the ~95 % empty slots are never read
    uint4 c4 = make_uint4(0, 0, 0, 0);
    if (in_range) c4 = *reinterpret_cast<const uint4*>(hcnt_all + hi);
    const uint32_t c[4] = {c4.x, c4.y, c4.z, c4.w};
    if (c4.x | c4.y | c4.z | c4.w) *reinterpret_cast<uint4*>(hcnt_all + hi) = make_uint4(0, 0, 0, 0);
    unsigned long long key[4];
    bool keep[4];
    uint32_t tsum = 0, tcnt = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      key[j] = c[j] ? hkeys_all[hi + j] : AT_EMPTY_KEY;
      if (c[j]) hkeys_all[hi + j] = AT_EMPTY_KEY;   // this kernel is the table's last reader: it leaves it empty for the next submission
      keep[j] = frame_ok && key[j] != AT_EMPTY_KEY && (int)c[j] >= P.min_cluster_points && (int)c[j] <= P.max_cluster_points;
      if (keep[j]) { tsum += c[j]; tcnt++; }
    }
    const uint32_t inc = wave_incl_scan(tsum), cinc = wave_incl_scan(tcnt);
    if (lane == 63) { wsum[wv] = inc; wcnt[wv] = cinc; }
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t ps = 0, cs = 0;
      for (int w = 0; w < 4; w++) { const uint32_t a = wsum[w], b = wcnt[w]; wsum[w] = ps; wcnt[w] = cs; ps += a; cs += b; }
      s_pbase = ps ? atomicAdd(&counters[frame].npoints_kept, ps) : 0;
      s_cbase = cs ? atomicAdd(&counters[frame].nclusters, cs) : 0;
    }
    __syncthreads();
    uint32_t off[4] = {AT_INVALID_SLOT, AT_INVALID_SLOT, AT_INVALID_SLOT, AT_INVALID_SLOT};
    uint32_t po = s_pbase + wsum[wv] + inc - tsum, ci = s_cbase + wcnt[wv] + cinc - tcnt;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (!keep[j]) continue;
      if (ci < P.ccap) {
        off[j] = po;
        ClusterRec rec;
        rec.key = key[j]; rec.start = po; rec.count = c[j];
        clusters_all[(size_t)frame * P.ccap + ci] = rec;
#pragma unroll
        for (int k = 0; k < FQ_NCLS; k++)
          if ((int)c[j] > L.lo[k] && (int)c[j] <= L.hi[k]) wcls[ch][j] = k;
        if (wcls[ch][j] >= 0) { wrank[ch][j] = atomicAdd(&s_wcnt[wcls[ch][j]], 1u); wci[ch][j] = ci; }
      } else {
        atomicOr(&counters[frame].flags, 0x4u);
      }
      po += c[j];
      ci++;
    }
    if (in_range) *reinterpret_cast<uint4*>(hoff_all + hi) = make_uint4(off[0], off[1], off[2], off[3]);
  }
  __syncthreads();
  if (threadIdx.x < FQ_NCLS) s_wbase[threadIdx.x] = s_wcnt[threadIdx.x] ? atomicAdd(&work_n[threadIdx.x], s_wcnt[threadIdx.x]) : 0u;
  __syncthreads();
#pragma unroll
  for (int ch = 0; ch < SEL_CHUNKS; ch++) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (wcls[ch][j] < 0) continue;
      const uint32_t pos = s_wbase[wcls[ch][j]] + wrank[ch][j];
      if (pos < L.cap[wcls[ch][j]]) work[L.off[wcls[ch][j]] + pos] = ((uint32_t)frame << P.wshift) | wci[ch][j];
      else atomicOr(&counters[frame].flags, 0x4u);
    }
  }
}


#ifndef SC_U
#define SC_U 8   // records of a lane in flight together (block per tile: 2: 1.41 ms, 4: 1.20, 8: 1.13 per 256 frames)
#endif
// One block per block of k_points (same tile): final position of a staged point = cluster range start (hoff of its pair's
// slot) + the block's base rank inside the cluster + the point's rank inside the block's group; no atomics.  The packed
// point is rebuilt from the tile origin, the pixel and the direction.  The blocks also share out the frame's long records.
__global__ __launch_bounds__(256) void k_scatter(const uint32_t* __restrict__ stage_all, const uint2* __restrict__ bhdr_all,
                                                 const uint2* __restrict__ btab_all, const uint4* __restrict__ long_all,
                                                 const uint32_t* __restrict__ hoff_all, uint32_t* __restrict__ pts_all,
                                                 const FrameCounters* __restrict__ counters, uint32_t gx_tiles, uint32_t gy_tiles,
                                                 DetParams P) {
  // One WAVE per tile of k_points (four tiles per block).  With a block per tile a thread had three records and the block was
  // gone after four dependent round trips -- header, staging word, table entry, range start -- so the kernel ran at the pace of
  // 8 blocks per CU times that latency; a wave per tile puts four times the records behind every wave slot's chain.
  const int frame = (int)blockIdx.z + P.frame0;
  const uint32_t bpf = gx_tiles * gy_tiles, blk = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
  const uint32_t* hoff = hoff_all + (size_t)frame * P.hcap;
  uint32_t* pts = pts_all + (size_t)frame * P.pcap;
  const uint2 hdr = blk < bpf ? bhdr_all[(size_t)frame * bpf + blk] : make_uint2(0u, 0u);
  const uint2* btab = btab_all + ((size_t)frame * bpf + blk) * PT_TB;
  const uint32_t* stage = stage_all + (size_t)frame * P.pcap + hdr.x;
  const int X0 = (int)(blk % gx_tiles) * PT_TW, Y0 = (int)(blk / gx_tiles) * PT_TH;
#ifndef SC_LDS_TABLE
#define SC_LDS_TABLE 1
#endif
#if SC_LDS_TABLE
  // The tile's pair table is resolved ONCE per wave into LDS -- entry -> first position of the tile's points of that pair in the
  // cluster's range, hoff[slot] + base rank -- so a record is one load (its staging word), one LDS read and one store; with the
  // table entry and the range start fetched per record it was three dependent loads.  All 256 entries are resolved: which of them
  // the tile used is known only to its records, and an unused one holds whatever an earlier submission left (a slot of the pair
  // table, or -- first use of the buffer -- anything: slots beyond the table read as "none"), never looked at.
  __shared__ uint32_t s_abs[4][PT_TB];
  {
    uint32_t* const mine = s_abs[threadIdx.x >> 6];
    uint2 te[PT_TB / 64];
    uint32_t ho[PT_TB / 64];
#pragma unroll
    for (int j = 0; j < PT_TB / 64; j++) te[j] = hdr.y ? btab[lane + 64u * (uint32_t)j] : make_uint2(AT_INVALID_SLOT, 0u);
#pragma unroll
    for (int j = 0; j < PT_TB / 64; j++) ho[j] = te[j].x < P.hcap ? hoff[te[j].x] : AT_INVALID_SLOT;
#pragma unroll
    for (int j = 0; j < PT_TB / 64; j++) mine[lane + 64u * (uint32_t)j] = ho[j] != AT_INVALID_SLOT ? ho[j] + te[j].y : AT_INVALID_SLOT;
  }
  __syncthreads();
  const uint32_t* const tabs = s_abs[threadIdx.x >> 6];
  for (uint32_t i0 = lane; i0 < hdr.y; i0 += SC_U * 64) {
    uint32_t w[SC_U];
#pragma unroll
    for (int u = 0; u < SC_U; u++) {
      const uint32_t i = i0 + (uint32_t)u * 64u;
      w[u] = i < hdr.y ? __builtin_nontemporal_load(stage + i) : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int u = 0; u < SC_U; u++) {
      if ((w[u] & 255u) == 255u) continue;   // (a long record's place holder, or beyond the tile's records)
      const uint32_t base = tabs[w[u] & 255u];
      if (base == AT_INVALID_SLOT) continue;
      const uint32_t pix = (w[u] >> 19) & 1023u;
      const int ly = (int)(pix >> 6), plx = (int)(pix & 63u), d = (int)((w[u] >> 29) & 3u);
      const int ddx = (d == 2) ? -1 : (d == 1 ? 0 : 1), ddy = (d == 0) ? 0 : 1;
      // packed point = x << 18 | y << 4 | (sign of gx + 1) << 2 | (sign of gy + 1), the gradient (dx, dy) * (+-255): pack_point
      // without its divisions by 255
      const int sgn = (w[u] >> 31) ? -1 : 1;
      pts[base + ((w[u] >> 8) & 2047u)] = ((uint32_t)(2 * (X0 + plx) + ddx) << 18) | ((uint32_t)(2 * (Y0 + ly) + ddy) << 4) |
                                          ((uint32_t)(ddx * sgn + 1) << 2) | (uint32_t)(ddy * sgn + 1);
    }
  }
#else
  // Three dependent loads per record (staging word -> table entry -> range start).  A tile has a few records per thread:
  // they are taken SC_U at a time, level by level, so that the latencies of a thread's records overlap instead of adding up.
  for (uint32_t i0 = lane; i0 < hdr.y; i0 += SC_U * 64) {
    uint32_t w[SC_U], off[SC_U];
    uint2 tb[SC_U];
#pragma unroll
    for (int u = 0; u < SC_U; u++) {
      const uint32_t i = i0 + (uint32_t)u * 64u;
      w[u] = i < hdr.y ? __builtin_nontemporal_load(stage + i) : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int u = 0; u < SC_U; u++) tb[u] = (w[u] & 255u) != 255u ? btab[w[u] & 255u] : make_uint2(AT_INVALID_SLOT, 0u);
#pragma unroll
    for (int u = 0; u < SC_U; u++) off[u] = tb[u].x != AT_INVALID_SLOT ? hoff[tb[u].x] : AT_INVALID_SLOT;
#pragma unroll
    for (int u = 0; u < SC_U; u++) {
      if (off[u] == AT_INVALID_SLOT) continue;
      const uint32_t pix = (w[u] >> 19) & 1023u;
      const int ly = (int)(pix >> 6), plx = (int)(pix & 63u), d = (int)((w[u] >> 29) & 3u);
      const int ddx = (d == 2) ? -1 : (d == 1 ? 0 : 1), ddy = (d == 0) ? 0 : 1;
      // packed point = x << 18 | y << 4 | (sign of gx + 1) << 2 | (sign of gy + 1), the gradient (dx, dy) * (+-255): pack_point
      // without its divisions by 255
      const int sgn = (w[u] >> 31) ? -1 : 1;
      pts[off[u] + tb[u].y + ((w[u] >> 8) & 2047u)] = ((uint32_t)(2 * (X0 + plx) + ddx) << 18) | ((uint32_t)(2 * (Y0 + ly) + ddy) << 4) |
                                                       ((uint32_t)(ddx * sgn + 1) << 2) | (uint32_t)(ddy * sgn + 1);
    }
  }
#endif
  uint32_t nl = counters[frame].nlong;
  if (nl > P.lcap) nl = P.lcap;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < nl; i += gridDim.x * 256) {
    const uint4 r = long_all[(size_t)frame * P.lcap + i];
    const uint32_t off = hoff[r.x];
    if (off != AT_INVALID_SLOT) pts[off + r.y] = r.z;
  }
}
