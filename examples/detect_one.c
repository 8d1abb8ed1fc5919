/* detect_one.c -- minimal C99 host of libapriltag_amd.so, shaped like CUAprilTagImpl in the reference
 * node (src/apriltag_node.cpp:434-461 create, :463-550 per frame, :552-558 destroy).
 *
 *   gcc -std=c99 -Iinclude examples/detect_one.c -Lisaac_ros_apriltag_amd -lapriltag_amd \
 *       -Wl,-rpath,$PWD/isaac_ros_apriltag_amd -o detect_one
 *   ./detect_one frame.pgm            (binary P5 PGM, 8 bit)
 *
 * The frame goes to the device through the library's own allocation helpers, so the example needs no
 * HIP headers.  Exit code 0 = ran; the detections are printed one per line.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "apriltag_amd.h"

static unsigned char* read_pgm(const char* path, unsigned* w, unsigned* h) {
  FILE* f = fopen(path, "rb");
  if (!f) return NULL;
  char magic[3] = {0};
  unsigned maxv = 0;
  if (fscanf(f, "%2s", magic) != 1 || strcmp(magic, "P5") != 0) { fclose(f); return NULL; }
  int c;
  /* skip comment lines between the header fields */
  for (int field = 0; field < 3; field++) {
    do { c = fgetc(f); } while (c == ' ' || c == '\n' || c == '\r' || c == '\t');
    while (c == '#') { while (c != '\n' && c != EOF) c = fgetc(f); do { c = fgetc(f); } while (c == ' ' || c == '\n'); }
    ungetc(c, f);
    unsigned v = 0;
    if (fscanf(f, "%u", &v) != 1) { fclose(f); return NULL; }
    if (field == 0) *w = v; else if (field == 1) *h = v; else maxv = v;
  }
  fgetc(f); /* the single whitespace byte after maxval */
  if (maxv != 255 || *w == 0 || *h == 0) { fclose(f); return NULL; }
  unsigned char* px = (unsigned char*)malloc((size_t)*w * *h);
  if (px && fread(px, 1, (size_t)*w * *h, f) != (size_t)*w * *h) { free(px); px = NULL; }
  fclose(f);
  return px;
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s frame.pgm [tag_size_m] [fx fy cx cy]\n", argv[0]); return 2; }
  unsigned w = 0, h = 0;
  unsigned char* host = read_pgm(argv[1], &w, &h);
  if (!host) { fprintf(stderr, "cannot read %s as a binary 8-bit PGM\n", argv[1]); return 2; }
  const float tag_size = argc > 2 ? (float)atof(argv[2]) : 0.22f;           /* node default, apriltag_node.cpp:565 */
  amdAprilTagsCameraIntrinsics_t cam;
  cam.fx = argc > 6 ? (float)atof(argv[3]) : (float)w;                        /* K[0], K[4], K[2], K[5] (:442-446) */
  cam.fy = argc > 6 ? (float)atof(argv[4]) : (float)w;
  cam.cx = argc > 6 ? (float)atof(argv[5]) : 0.5f * (float)w;
  cam.cy = argc > 6 ? (float)atof(argv[6]) : 0.5f * (float)h;

  amdAprilTagsHandle det = NULL;
  int rc = amdCreateAprilTagsDetector(&det, w, h, 4 /* tile_size, :566 */, AMDAT_TAG36H11, &cam, tag_size);
  if (rc != 0) { fprintf(stderr, "Failed to create AprilTags detector (error code %d)\n", rc); return 1; }

  void* dev = NULL;
  const size_t pitch = w;
  rc = amdAprilTagsDeviceAlloc(&dev, pitch * h);
  if (rc == 0) rc = amdAprilTagsCopyToDevice(dev, host, pitch * h, NULL);
  if (rc != 0) { fprintf(stderr, "device buffer: error code %d\n", rc); return 1; }

  amdAprilTagsImageInput_t image;
  image.width = w; image.height = h; image.dev_ptr = (const uint8_t*)dev; image.pitch = pitch;
  enum { MAX_TAGS = 64 };                                                     /* node default, :564 */
  amdAprilTagsID_t tags[MAX_TAGS];
  uint32_t n = 0;
  rc = amdAprilTagsDetect(det, &image, tags, &n, MAX_TAGS, NULL /* the handle's own stream */);
  if (rc != 0) { fprintf(stderr, "Failed to run AprilTags detector (error code %d)\n", rc); return 1; }

  for (uint32_t i = 0; i < n; i++) {
    const amdAprilTagsID_t* t = &tags[i];
    printf("id %d  centre (%.2f, %.2f)  corners", (int)t->id, t->center.x, t->center.y);
    for (int c = 0; c < 4; c++) printf(" (%.2f, %.2f)", t->corners[c].x, t->corners[c].y);
    printf("  t = (%.4f, %.4f, %.4f) m\n", t->translation[0], t->translation[1], t->translation[2]);
  }
  printf("%u detection(s)\n", (unsigned)n);

  amdAprilTagsDeviceFree(dev);
  amdAprilTagsDestroy(det);
  free(host);
  return 0;
}
