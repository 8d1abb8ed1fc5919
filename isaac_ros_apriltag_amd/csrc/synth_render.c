/* synth_render.c -- deterministic synthetic frame renderer (host C, no dependencies).
 *
 * The reference's only image fixture (isaac_ros_apriltag/test/test_cases/apriltag0/image.png) is a
 * Git-LFS pointer, so every test/bench frame of this repo is rendered here: tags are warped by a
 * homography with ss x ss supersampling onto a flat background, then counter-based integer noise is
 * added.  All arithmetic that decides a byte is integer or a single IEEE double operation, and the
 * PRNG is a counter-based splitmix64, so the container and the GPU box render identical bytes.
 *
 * Tag coordinates follow AprilRobotics: the black border's outer edge is the square [-1,1]^2,
 * x right, y down; the white quiet ring is one cell wide outside it.  Pixel (ix,iy) covers
 * [ix,ix+1) x [iy,iy+1) (centre at +0.5), matching value_for_pixel().
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  uint64_t code;  /* row-major, MSB = top-left data cell, 1 = white */
  int32_t d;      /* data cells per side (width_at_border = d + 2) */
  int32_t pad;
  double H[9];    /* tag [-1,1]^2 -> image pixels, row-major */
} synth_tag_t;

static inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ULL;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
  return x ^ (x >> 31);
}

static int inv3(const double* m, double* o) {
  double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
  double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
  if (fabs(det) < 1e-300) return -1;
  o[0] = c00 / det; o[1] = (m[2] * m[7] - m[1] * m[8]) / det; o[2] = (m[1] * m[5] - m[2] * m[4]) / det;
  o[3] = c01 / det; o[4] = (m[0] * m[8] - m[2] * m[6]) / det; o[5] = (m[2] * m[3] - m[0] * m[5]) / det;
  o[6] = c02 / det; o[7] = (m[1] * m[6] - m[0] * m[7]) / det; o[8] = (m[0] * m[4] - m[1] * m[3]) / det;
  return 0;
}

/* Renders into img (w x h, pitch bytes per row).
 *   background : flat gray level before noise
 *   sigma_q8   : noise standard deviation in 1/256 gray levels (0 = none)
 *   seed       : noise seed
 *   ss         : supersampling factor per axis (1..8)
 *   black/white: tag gray levels
 * Returns 0, or -1 on a singular homography. */
int synth_render(uint8_t* img, int w, int h, int pitch, int background, int sigma_q8, uint64_t seed,
                 const synth_tag_t* tags, int ntags, int ss, int black, int white) {
  for (int y = 0; y < h; y++) memset(img + (size_t)y * pitch, background, (size_t)w);
  for (int t = 0; t < ntags; t++) {
    const synth_tag_t* tg = &tags[t];
    double Hi[9];
    if (inv3(tg->H, Hi) != 0) return -1;
    int wb = tg->d + 2;
    double cell = 2.0 / wb, ext = 1.0 + cell;
    /* bounding box of the quiet ring's outer corners */
    double xmin = 1e30, xmax = -1e30, ymin = 1e30, ymax = -1e30;
    for (int k = 0; k < 4; k++) {
      double tx = (k == 1 || k == 2) ? ext : -ext, ty = (k >= 2) ? ext : -ext;
      double X = tg->H[0] * tx + tg->H[1] * ty + tg->H[2];
      double Y = tg->H[3] * tx + tg->H[4] * ty + tg->H[5];
      double Z = tg->H[6] * tx + tg->H[7] * ty + tg->H[8];
      double u = X / Z, v = Y / Z;
      if (u < xmin) xmin = u;
      if (u > xmax) xmax = u;
      if (v < ymin) ymin = v;
      if (v > ymax) ymax = v;
    }
    int x0 = (int)floor(xmin) - 1, x1 = (int)ceil(xmax) + 1, y0 = (int)floor(ymin) - 1, y1 = (int)ceil(ymax) + 1;
    if (x0 < 0) x0 = 0;
    if (y0 < 0) y0 = 0;
    if (x1 > w - 1) x1 = w - 1;
    if (y1 > h - 1) y1 = h - 1;
    for (int iy = y0; iy <= y1; iy++)
      for (int ix = x0; ix <= x1; ix++) {
        int acc = 0, hit = 0;
        for (int sy = 0; sy < ss; sy++)
          for (int sx = 0; sx < ss; sx++) {
            double u = ix + (sx + 0.5) / ss, v = iy + (sy + 0.5) / ss;
            double X = Hi[0] * u + Hi[1] * v + Hi[2];
            double Y = Hi[3] * u + Hi[4] * v + Hi[5];
            double Z = Hi[6] * u + Hi[7] * v + Hi[8];
            double tx = X / Z, ty = Y / Z;
            int val = img[(size_t)iy * pitch + ix];
            if (tx >= -ext && tx < ext && ty >= -ext && ty < ext) {
              hit = 1;
              int cx = (int)floor((tx + 1.0) / cell), cy = (int)floor((ty + 1.0) / cell); /* -1 .. wb */
              if (cx < 0 || cy < 0 || cx >= wb || cy >= wb) val = white;                 /* quiet ring */
              else if (cx == 0 || cy == 0 || cx == wb - 1 || cy == wb - 1) val = black;   /* border */
              else {
                int bit = (cy - 1) * tg->d + (cx - 1);
                val = ((tg->code >> (tg->d * tg->d - 1 - bit)) & 1) ? white : black;
              }
            }
            acc += val;
          }
        if (hit) img[(size_t)iy * pitch + ix] = (uint8_t)((2 * acc + ss * ss) / (2 * ss * ss));
      }
  }
  if (sigma_q8 > 0) {
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) {
        uint64_t ctr = seed * 0xD1342543DE82EF95ULL + ((uint64_t)y * (uint64_t)w + (uint64_t)x) * 3ULL;
        int64_t sum = 0;
        for (int j = 0; j < 3; j++) {
          uint64_t r = splitmix64(ctr + (uint64_t)j);
          sum += (int64_t)(r & 0xFFFF) + (int64_t)((r >> 16) & 0xFFFF) + (int64_t)((r >> 32) & 0xFFFF) + (int64_t)((r >> 48) & 0xFFFF);
        }
        /* 12 uniforms on [0,65536): mean 6*65535, variance 12 * 65536^2/12 -> unit sigma = 65536 */
        int64_t nq = (sum - 6 * 65535) * (int64_t)sigma_q8; /* noise in 2^-24 gray levels */
        int64_t v = ((int64_t)img[(size_t)y * pitch + x] << 24) + nq + (1LL << 23);
        int64_t r = v >> 24; /* floor for negatives as well (arithmetic shift) */
        if (r < 0) r = 0;
        if (r > 255) r = 255;
        img[(size_t)y * pitch + x] = (uint8_t)r;
      }
  }
  return 0;
}

/* Family table access for the scene generator (data from include/apriltag_amd_families.h). */
#include "../../include/apriltag_amd_families.h"
const uint64_t* synth_family_codes(const char* name, int* ncodes, int* d) {
  if (!strcmp(name, "tag36h11")) { *ncodes = APRILTAG_AMD_TAG36H11_NCODES; *d = 6; return apriltag_amd_tag36h11_codes; }
#ifdef APRILTAG_AMD_TAG36H10_NCODES
  if (!strcmp(name, "tag36h10")) { *ncodes = APRILTAG_AMD_TAG36H10_NCODES; *d = 6; return apriltag_amd_tag36h10_codes; }
#endif
  if (!strcmp(name, "tag25h9")) { *ncodes = APRILTAG_AMD_TAG25H9_NCODES; *d = 5; return apriltag_amd_tag25h9_codes; }
  if (!strcmp(name, "tag16h5")) { *ncodes = APRILTAG_AMD_TAG16H5_NCODES; *d = 4; return apriltag_amd_tag16h5_codes; }
  *ncodes = 0; *d = 0;
  return 0;
}
