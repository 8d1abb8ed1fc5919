/* Host model of sqrt_u18 (isaac_ros_apriltag_amd/csrc/common.h): f32 reciprocal-square-root seed, one coupled
 * Goldschmidt step, one residual correction, all in fused multiply-adds.  The device seed (v_rsq_f32) is accurate to
 * about 1 ulp; the model perturbs a correctly rounded seed by -8..+8 ulp and requires the result to equal IEEE sqrt
 * for every integer argument below 2^18.  Prints the number of mismatches. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static double sqrt_u18_model(uint32_t G, int seed_ulps) {
  float gf = (float)G;
  float in = gf < 1e-30f ? 1e-30f : gf;
  float r0 = 1.0f / sqrtf(in);
  uint32_t bits;
  memcpy(&bits, &r0, 4);
  bits += (uint32_t)seed_ulps;
  memcpy(&r0, &bits, 4);
  double r = (double)r0, g = (double)G;
  double s = g * r, h = 0.5 * r;
  double e = fma(-h, s, 0.5);
  s = fma(s, e, s);
  h = fma(h, e, h);
  double d = fma(-s, s, g);
  return fma(d, h, s);
}

int main(void) {
  long bad = 0;
  for (int k = -8; k <= 8; k++)
    for (uint32_t G = 0; G < (1u << 18); G++) {
      double a = sqrt_u18_model(G, k), b = sqrt((double)G);
      if (memcmp(&a, &b, 8)) bad++;
    }
  printf("%ld\n", bad);
  return 0;
}
