/* gen_tag_family.c -- regenerates a classic AprilTag lexicode family (tag16h5, tag25h9, tag36h11, ...)
 * by running the published generator's procedure (Olson, "AprilTag: A robust and flexible visual fiducial
 * system", ICRA 2011, section on tag families; april.tag.TagFamilyGenerator):
 *
 *   V0 = java.util.Random(nbits*10000 + minhamming*100 + mincomplexity).nextLong()
 *   for k = 1, 2, ...:  v = (V0 + k * 982451653) mod 2^nbits; accept v when
 *     - its complexity (greedy count of the rectangles needed to paint the d x d pattern) >= mincomplexity
 *     - the four rotations of v are pairwise >= minhamming apart
 *     - v is >= minhamming away from every rotation of every accepted code
 *
 * Nothing here comes from /root/reference (it holds no codebook).  The procedure was pinned against the
 * tables that are independently known: it reproduces tag16h5 (30 codes) and tag25h9 (35 codes) exactly over
 * their complete 2^16 / 2^25 sweeps, and the 27 tag36h11 codes round 1 had validated by the stride property
 * (tests/test_families_cpu.py).  The seed formula was found by searching java.util.Random seeds for the one
 * whose first output precedes code[0] by a small multiple of the stride: 361110, 250908 and 160505 hit for
 * the three families with k = 2, 3, 2 -- i.e. mincomplexity 10 / 8 / 5, the values the published tables'
 * headers state.
 *
 * Build: gcc -O3 -march=native -fopenmp -o gen_tag_family tools/gen_tag_family.c
 * Use:   gen_tag_family nbits minhamming mincomplexity [max_codes] [log2_sweep]   (prints one code per line)
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PRIME 982451653ULL

static int g_d, g_nbits, g_minham, g_mincomplex;
static uint64_t g_mask;

static int64_t java_next_long(int64_t seed) {
  uint64_t s = ((uint64_t)seed ^ 0x5DEECE66DULL) & ((1ULL << 48) - 1);
  s = (s * 0x5DEECE66DULL + 0xBULL) & ((1ULL << 48) - 1);
  int32_t hi = (int32_t)(s >> 16);
  s = (s * 0x5DEECE66DULL + 0xBULL) & ((1ULL << 48) - 1);
  int32_t lo = (int32_t)(s >> 16);
  return (int64_t)((uint64_t)(int64_t)hi << 32) + (int64_t)lo;
}

/* bit (nbits-1-i) = cell i, cells row-major; rotation as in the classic tables */
static uint64_t rotate90(uint64_t w, int d) {
  uint64_t wr = 0;
  for (int r = d - 1; r >= 0; r--)
    for (int c = 0; c < d; c++) {
      int b = r + d * c;
      wr = (wr << 1) | ((w >> b) & 1);
    }
  return wr;
}

/* Candidate rectangles in evaluation order: bottom row, top row, left column, right column ascending, white
 * before black; among equal gains the LAST candidate wins.  The published description only says "greedy";
 * this tie rule is the one (of 230 400 loop-order / tie / initial-canvas variants searched) that reproduces the
 * complete accept/reject pattern of the tag16h5 and tag25h9 sweeps and of the validated tag36h11 prefix; the
 * twelve orderings that do differ only in the priority of the column and colour keys and generate identical
 * tables. */
static uint64_t g_rect[2048];
static int g_rcol[2048], g_nrect;

static void make_rects(void) {
  int d = g_d;
  g_nrect = 0;
  for (int r1 = 0; r1 < d; r1++)
    for (int r0 = 0; r0 <= r1; r0++)
      for (int c0 = 0; c0 < d; c0++)
        for (int c1 = c0; c1 < d; c1++)
          for (int col = 1; col >= 0; col--) {
            uint64_t m = 0;
            for (int r = r0; r <= r1; r++)
              for (int c = c0; c <= c1; c++) m |= 1ULL << (g_nbits - 1 - (r * d + c));
            g_rect[g_nrect] = m;
            g_rcol[g_nrect] = col;
            g_nrect++;
          }
}

/* Greedy painter: starting from a blank canvas, repeatedly paint the rectangle (any colour, overdraw allowed)
 * that most reduces the number of wrong cells, until the canvas equals the pattern.  Returns
 * min(count, limit). */
static int complexity(uint64_t T, int limit) {
  uint64_t wrong = g_mask, canvas = 0;
  int n = 0;
  while (wrong) {
    if (n + 1 >= limit) return limit;   /* at least one more rectangle is needed */
    int best = -1000, bi = -1;
    for (int i = 0; i < g_nrect; i++) {
      const uint64_t m = g_rect[i];
      const uint64_t tm = g_rcol[i] ? T : ~T;
      const int gain = __builtin_popcountll(m & wrong & tm) - __builtin_popcountll(m & ~wrong & ~tm & g_mask);
      if (gain >= best) { best = gain; bi = i; }
    }
    canvas = (canvas & ~g_rect[bi]) | (g_rcol[bi] ? g_rect[bi] : 0);
    wrong = (wrong & ~g_rect[bi]) | ((canvas ^ T) & g_rect[bi]);
    n++;
  }
  return n;
}

static inline int ham_ge(uint64_t a, uint64_t b, int m) { return __builtin_popcountll(a ^ b) >= m; }

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s nbits minhamming mincomplexity [max_codes] [log2_sweep]\n", argv[0]); return 2; }
  g_nbits = atoi(argv[1]); g_minham = atoi(argv[2]); g_mincomplex = atoi(argv[3]);
  long max_codes = argc > 4 ? atol(argv[4]) : 1L << 30;
  int lg = argc > 5 ? atoi(argv[5]) : g_nbits;
  for (g_d = 1; g_d * g_d < g_nbits; g_d++) {}
  if (g_d * g_d != g_nbits || g_d > 7) { fprintf(stderr, "nbits must be a square <= 49\n"); return 2; }
  g_mask = (g_nbits == 64) ? ~0ULL : ((1ULL << g_nbits) - 1);
  make_rects();
  const uint64_t V0 = (uint64_t)java_next_long((int64_t)g_nbits * 10000 + g_minham * 100 + g_mincomplex) & g_mask;
  fprintf(stderr, "# V0 = 0x%llx, %d candidate rectangles\n", (unsigned long long)V0, g_nrect);

  size_t rcap = 1 << 20, nrot = 0;
  uint64_t* rot = (uint64_t*)malloc(rcap * 8);
  long ncodes = 0;
  const uint64_t total = 1ULL << lg;
  uint64_t chunk = 256;
  uint64_t* surv = (uint64_t*)malloc(sizeof(uint64_t) * (1 << 24));
  for (uint64_t k0 = 1; k0 <= total && ncodes < max_codes;) {
    const uint64_t kn = (k0 + chunk - 1 <= total) ? chunk : total - k0 + 1;
    size_t ns = 0;
    const size_t nfrozen = nrot;
    /* parallel filter against the codes known at the start of the chunk */
#pragma omp parallel
    {
      uint64_t local[4096];
      size_t nl = 0;
#pragma omp for schedule(dynamic, 4096) nowait
      for (uint64_t j = 0; j < kn; j++) {
        const uint64_t v = (V0 + (k0 + j) * PRIME) & g_mask;
        int ok = 1;
        for (size_t i = 0; i < nfrozen; i++)
          if (!ham_ge(v, rot[i], g_minham)) { ok = 0; break; }
        if (!ok) continue;
        local[nl++] = k0 + j;
        if (nl == 4096) {
#pragma omp critical
          { memcpy(surv + ns, local, nl * 8); ns += nl; }
          nl = 0;
        }
      }
#pragma omp critical
      { memcpy(surv + ns, local, nl * 8); ns += nl; }
    }
    /* survivors in k order, sequentially */
    for (size_t a = 1; a < ns; a++) {  /* insertion sort of nearly sorted blocks would do; ns is small */
      uint64_t x = surv[a]; size_t b = a;
      while (b > 0 && surv[b - 1] > x) { surv[b] = surv[b - 1]; b--; }
      surv[b] = x;
    }
    for (size_t a = 0; a < ns && ncodes < max_codes; a++) {
      const uint64_t k = surv[a];
      const uint64_t v = (V0 + k * PRIME) & g_mask;
      int ok = 1;
      for (size_t i = nfrozen; i < nrot; i++)
        if (!ham_ge(v, rot[i], g_minham)) { ok = 0; break; }
      if (!ok) continue;
      const uint64_t r1 = rotate90(v, g_d), r2 = rotate90(r1, g_d), r3 = rotate90(r2, g_d);
      if (!ham_ge(v, r1, g_minham) || !ham_ge(v, r2, g_minham) || !ham_ge(v, r3, g_minham) || !ham_ge(r1, r2, g_minham) ||
          !ham_ge(r1, r3, g_minham) || !ham_ge(r2, r3, g_minham))
        continue;
      if (complexity(v, g_mincomplex) < g_mincomplex) continue;
      if (nrot + 4 > rcap) { rcap *= 2; rot = (uint64_t*)realloc(rot, rcap * 8); }
      rot[nrot++] = v; rot[nrot++] = r1; rot[nrot++] = r2; rot[nrot++] = r3;
      printf("0x%0*llx %llu\n", (g_nbits + 3) / 4, (unsigned long long)v, (unsigned long long)k);
      fflush(stdout);
      ncodes++;
    }
    k0 += kn;
    if (ns < 64 && chunk < (1ULL << 24)) chunk *= 2;
    if ((k0 & ((1ULL << 30) - 1)) < chunk)
      fprintf(stderr, "# k = %llu (%.1f%%), %ld codes\n", (unsigned long long)k0, 100.0 * (double)k0 / (double)total, ncodes);
  }
  fprintf(stderr, "# done: %ld codes\n", ncodes);
  return 0;
}
