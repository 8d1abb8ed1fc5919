/* Generates the stand-in family "synth36h11": the 27 stride-validated tag36h11 codes (ids 0-26)
 * followed by a deterministic lexicode continuation (same 36-bit / 6x6 / min-Hamming-11 shape).
 * NOT the published tag36h11 table beyond id 26 -- see DESIGN.md "Tag codebooks".
 * Build: gcc -O2 -o /tmp/gen tools/gen_synth36h11.c && /tmp/gen > /tmp/synth36h11.inc
 */
#include <stdint.h>
#include <stdio.h>
#define D 6
#define NB 36
static uint64_t rot90(uint64_t w) {
  /* row-major code, MSB = top-left; rotate pattern by 90 degrees */
  uint64_t o = 0;
  for (int r = 0; r < D; r++) for (int c = 0; c < D; c++) {
    /* destination (r,c) takes source (D-1-c, r) */
    int sr = D-1-c, sc = r;
    int sbit = NB-1-(sr*D+sc), dbit = NB-1-(r*D+c);
    if ((w >> sbit) & 1) o |= 1ULL << dbit;
  }
  return o;
}
static int hd(uint64_t a, uint64_t b) { return __builtin_popcountll(a^b); }
static int energy(uint64_t w) {
  int e = 0;
  for (int r = 0; r < D; r++) for (int c = 0; c < D; c++) {
    int b = (w >> (NB-1-(r*D+c))) & 1;
    if (c+1 < D) e += b != (int)((w >> (NB-1-(r*D+c+1))) & 1);
    if (r+1 < D) e += b != (int)((w >> (NB-1-((r+1)*D+c))) & 1);
  }
  return e;
}
int main(int argc, char** argv) {
  const uint64_t P = 982451653ULL, M = (1ULL<<NB)-1;
  static uint64_t codes[587]; static uint64_t rots[587*4]; int n = 0;
  /* validated prefix: k offsets of the real table */
  static const int ks[27] = {0,1,2,4,6,8,13,16,21,22,26,29,30,31,33,34,36,40,41,46,47,48,50,55,56,59,60};
  uint64_t v0 = 0xd5d628584ULL;
  for (int i = 0; i < 27; i++) {
    uint64_t v = (v0 + (uint64_t)ks[i]*P) & M;
    codes[n] = v; rots[4*n] = v; rots[4*n+1] = rot90(v); rots[4*n+2] = rot90(rots[4*n+1]); rots[4*n+3] = rot90(rots[4*n+2]); n++;
  }
  uint64_t v = (v0 + 60ULL*P) & M;
  long iter = 0;
  while (n < 587 && iter < 400000000L) {
    iter++; v = (v + P) & M;
    if (energy(v) < 20) continue;   /* 1/3 of max 60 */
    uint64_t r1 = rot90(v), r2 = rot90(r1), r3 = rot90(r2);
    if (hd(v,r1) < 11 || hd(v,r2) < 11 || hd(v,r3) < 11 || hd(r1,r2) < 11 || hd(r1,r3) < 11 || hd(r2,r3) < 11) continue;
    int ok = 1;
    for (int i = 0; i < 4*n; i++) if (hd(v, rots[i]) < 11) { ok = 0; break; }
    if (!ok) continue;
    codes[n] = v; rots[4*n] = v; rots[4*n+1] = r1; rots[4*n+2] = r2; rots[4*n+3] = r3; n++;
  }
  fprintf(stderr, "n=%d iter=%ld\n", n, iter);
  for (int i = 0; i < n; i++) printf("0x%09llxULL,%s", (unsigned long long)codes[i], (i%6==5)?"\n":" ");
  printf("\n");
  return 0;
}
