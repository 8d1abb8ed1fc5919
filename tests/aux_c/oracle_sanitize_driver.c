/* Drives the CPU restatement (oracle/apriltag_oracle.c) under -fsanitize=address,undefined: a frame with one axis-aligned
 * tag36h11 (id 0 drawn from the code table), a frame of noise, a frame of stripes (many boundary points) and degenerate
 * sizes, each through the full pipeline with a stage dump.  SURVEY.md section 5 (auxiliary subsystems): the reference
 * builds with -Wall -Wextra -Wpedantic only; this is the memory / UB check of this repository's own host code. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../oracle/apriltag_oracle.h"

static uint32_t lcg(uint32_t* s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }

static int run(const uint8_t* img, int w, int h, int pitch, int decimate, int expect_id) {
  ato_params_t prm;
  ato_default_params(&prm);
  prm.decimate = decimate;
  prm.fx = prm.fy = 500; prm.cx = w / 2.0; prm.cy = h / 2.0;
  ato_family_t fams[2];
  if (ato_builtin_family("tag36h11", &fams[0]) || ato_builtin_family("tag25h9", &fams[1])) return -100;
  ato_detection_t out[32];
  ato_dump_t dump;
  memset(&dump, 0, sizeof(dump));
  int n = ato_detect(&prm, fams, 2, img, w, h, pitch, out, 32, &dump);
  ato_dump_free(&dump);
  if (n < 0) return n;
  if (expect_id >= 0 && !(n == 1 && out[0].id == expect_id && out[0].hamming == 0)) return -200 - n;
  return n;
}

int main(void) {
  const int W = 320, H = 240, pitch = 336;
  setvbuf(stdout, NULL, _IONBF, 0);
  uint8_t* img = (uint8_t*)malloc((size_t)pitch * H);
  ato_family_t f;
  if (ato_builtin_family("tag36h11", &f)) return 2;
  /* tag id 0, cells of 12 px: white quiet ring, black border, 6 x 6 data */
  memset(img, 150, (size_t)pitch * H);
  const int cell = 12, x0 = 100, y0 = 60;
  for (int cy = 0; cy < 10; cy++)
    for (int cx = 0; cx < 10; cx++) {
      int v = 230;
      if (cx >= 1 && cx <= 8 && cy >= 1 && cy <= 8) {
        v = 25;
        if (cx >= 2 && cx <= 7 && cy >= 2 && cy <= 7) {
          int bit = (cy - 2) * 6 + (cx - 2);
          v = ((f.codes[0] >> (35 - bit)) & 1) ? 230 : 25;
        }
      }
      for (int y = 0; y < cell; y++) memset(img + (size_t)(y0 + cy * cell + y) * pitch + x0 + cx * cell, v, cell);
    }
  int rc = run(img, W, H, pitch, 1, 0);
  printf("tag frame: %d\n", rc);
  if (rc != 1) return 3;
  rc = run(img, W, H, pitch, 2, 0);
  printf("tag frame, decimate 2: %d\n", rc);
  if (rc != 1) return 4;
  uint32_t s = 12345;
  for (int i = 0; i < pitch * H; i++) img[i] = (uint8_t)lcg(&s);
  rc = run(img, W, H, pitch, 1, -1);
  printf("noise frame: %d\n", rc);
  if (rc < 0) return 5;
  for (int y = 0; y < H; y++) memset(img + (size_t)y * pitch, (y & 1) ? 215 : 40, W);
  rc = run(img, W, H, pitch, 1, -1);
  printf("stripes: %d\n", rc);
  if (rc < 0) return 6;
  static const int sizes[][2] = {{1, 1}, {3, 7}, {4, 4}, {9, 5}, {17, 33}};
  for (unsigned k = 0; k < sizeof(sizes) / sizeof(sizes[0]); k++) {
    rc = run(img, sizes[k][0], sizes[k][1], pitch, 1, -1);
    /* (frames smaller than a few tiles are refused with an error code, never processed out of bounds) */
    if (rc < 0 && !(rc == -2 && (sizes[k][0] < 8 || sizes[k][1] < 8))) { printf("size %dx%d: %d\n", sizes[k][0], sizes[k][1], rc); return 7; }
  }
  /* a custom AprilTag-3 style layout: rejected when not closed under the quarter turn, accepted otherwise */
  {
    int8_t bx[8] = {-1, 0, 1, 2, 3, 3, 3, 3}, by[8] = {-1, -1, -1, -1, -1, 0, 1, 2};
    uint64_t codes[1] = {0xA5};
    ato_family_t g;
    if (ato_custom_family("bad", 8, bx, by, 3, 5, 0, codes, 1, &g) == 0) return 8;
  }
  free(img);
  printf("ok\n");
  return 0;
}
