/* apriltag_oracle.h -- CPU restatement of the AprilRobotics apriltag_detect() pipeline.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (isaac_ros_apriltag_amd/) links, imports
 * or calls this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * PARITY STATUS: "parity unpinned" beyond the reference's own golden numbers.  The detector
 * arithmetic of the reference lives in closed binaries (libcuapriltags.a, called at
 * isaac_ros_apriltag/src/apriltag_node.cpp:450-452,491-493; NVIDIA VPI, called at :228-231,:290-301)
 * and AprilRobotics' apriltag is not vendored/installed.  This file restates
 * the published AprilTag-3 algorithm (Olson 2011; Wang & Olson 2016; AprilRobotics/apriltag 3.x,
 * BSD-2) from its public description; it is pinned against the reference's golden vector
 * (isaac_ros_apriltag/test/isaac_ros_apriltag_pol_test.py:113-175), against the analytic ground
 * truth of the in-repo renderer, and its tag tables against the published generator's procedure
 * (tools/gen_tag_family.c reproduces tag16h5, tag25h9 and the 587 codes of tag36h11).  Where the public
 * algorithm accumulates in a data-dependent order (hash iteration), this restatement fixes a canonical
 * order; those places are marked CANONICAL.  The steps it formulates differently from upstream can be
 * switched to upstream's formulation (ato_params_t.variant) so that the distance is measured, not
 * assumed (tests/test_oracle_variants_cpu.py).  An optional cross-check against a real libapriltag.so
 * exists (oracle/aprilrobotics_xcheck.py); none was available, so parity with AprilRobotics' binary
 * remains unpinned.
 */
#ifndef APRILTAG_ORACLE_H_
#define APRILTAG_ORACLE_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ATO_MAX_FAMILIES 4
#define ATO_NO_LABEL 0xFFFFFFFFu

typedef struct {
  char name[32];
  uint32_t nbits;            /* data bits = d*d */
  uint32_t d;                /* data cells per side */
  uint32_t width_at_border;  /* cells border-to-border (d + 2 for the classic families) */
  uint32_t total_width;      /* incl. white quiet ring (d + 4) */
  int32_t reversed_border;   /* 0 for the classic families */
  uint32_t ncodes;
  const uint64_t* codes;     /* bit (nbits-1-i) of a code is data cell i (classic: row-major), 1 = white */
  /* AprilTag-3 layout: cell of data bit i in border coordinates ((0,0) = top-left cell of the border square, cells of the
   * outer rings are negative or >= width_at_border).  Classic families: (1 + i % d, 1 + i / d); d = 0 for layouts that are
   * not a d x d square.  A layout must map onto itself under (x, y) -> (width_at_border - 1 - y, x). */
  int8_t bit_x[64], bit_y[64];
} ato_family_t;

typedef struct {
  int32_t decimate;              /* quad_decimate, integer >= 1 */
  int32_t tile_size;             /* 4 (apriltag_node.cpp:566 default) */
  int32_t min_white_black_diff;  /* 5 */
  int32_t min_component_size;    /* 25 */
  int32_t min_cluster_points;    /* 24 */
  int32_t max_nmaxima;           /* 10 */
  double cos_critical_rad;       /* cos(10 deg) */
  double max_line_fit_mse;       /* 10 */
  int32_t refine_edges;          /* 1 */
  double decode_sharpening;      /* 0.25 */
  int32_t max_hamming;           /* 2 */
  double fx, fy, cx, cy;         /* intrinsics K[0],K[4],K[2],K[5] (apriltag_node.cpp:442-446) */
  double tag_size;               /* metres, black-border edge (apriltag_node.cpp:565) */
  double skew;                   /* K[0][1]; the reference's VPI path passes it (apriltag_node.cpp:215-225) */
  int32_t variant;               /* 0 = canonical definitions (what the HIP path is checked against bit for bit);
                                  * ATO_VAR_* bits switch single steps to AprilRobotics' own formulation, used only
                                  * to BOUND the distance between the two (tests/test_oracle_variants_cpu.py) */
} ato_params_t;

/* upstream formulations of the steps this restatement defines canonically */
#define ATO_VAR_SEQ_MOMENTS 1  /* compute_lfps: the six running sums are sequential double additions */
#define ATO_VAR_ATAN_NORMAL 2  /* refine_edges: normal = (cosf, sinf) of 0.5*atan2f(-2Cxy, Cyy-Cxx) */
#define ATO_VAR_SVD_POLAR 4    /* homography_to_pose: R = U V^T from the SVD instead of Newton steps */
#define ATO_VAR_FLOAT_DOT 8    /* fit_quad: border-direction dot accumulated in float, point by point */
/* (16 was FLOAT_COS: upstream's float cos_critical_rad is the definition now) */
#define ATO_VAR_AT3_BIT_ORDER 32 /* quad_decode: white/black scores (floats) accumulated in AprilTag 3's bit order (four
                                * rotated quadrant triangles, centre bit last) instead of row-major */
#define ATO_VAR_FAST_PATHS 128   /* NOT a change of definition: the same statements through cheaper code, for bench.py's second CPU row
                                  * ("a CPU baseline that is trying").  With ATO_VAR_SEQ_MOMENTS the 128-bit exact sums are not
                                  * formed at all (without this flag both are, so that the checker can compare them); the slope keys
                                  * are sorted by an LSD radix sort instead of qsort with a comparator (same total order); codes are
                                  * looked up in a hash table of every code word within two bit errors -- AprilRobotics' quick_decode --
                                  * instead of a scan over the family per rotation (same result for max_hamming <= 2). */
/* (64 was TRIG_RZ: H * Rz with libm's cos / sin values and the full 3x3 product is the definition now) */

typedef struct {
  int32_t family;   /* index into the family list */
  int32_t id;
  int32_t hamming;
  float decision_margin;
  double H[9];      /* row-major homography tag[-1,1]^2 -> image */
  double c[2];      /* centre = H(0,0) */
  double p[4][2];   /* AprilRobotics order: H(-1,1), H(1,1), H(1,-1), H(-1,-1) */
  double R[9];      /* row-major rotation, tag frame in camera optical frame */
  double t[3];      /* metres */
} ato_detection_t;

typedef struct {
  float p[4][2];
  int32_t reversed_border;
  uint64_t key;     /* component-pair key of the cluster the quad came from */
} ato_quad_t;

typedef struct {
  uint64_t key;     /* (min(rep0,rep1) << 32) | max(rep0,rep1) */
  uint32_t start;   /* offset into points[] */
  uint32_t count;
} ato_cluster_t;

/* Stage dump of one frame (all arrays malloc'ed by ato_detect_dump, freed by ato_dump_free). */
typedef struct {
  int32_t w, h;                 /* working-image size */
  uint8_t* gray;                /* decimated gray image, w*h */
  uint8_t* thr;                 /* threshold image, w*h */
  uint32_t* label;              /* w*h, canonical representative (min pixel index) or ATO_NO_LABEL */
  uint32_t* csize;              /* w*h, component size stored at the representative's index */
  uint32_t nclusters;
  ato_cluster_t* clusters;      /* kept clusters (min_cluster_points <= count <= 3*(2w+2h)), sorted by key */
  uint32_t npoints;
  uint32_t* points;             /* packed (x<<18)|(y<<4)|(gxc<<2)|gyc, per cluster sorted ascending */
  uint32_t nquads;
  ato_quad_t* quads;            /* after fit (working-image coords scaled back to full-res), sorted by key */
  uint32_t ndet;
  ato_detection_t* dets;        /* final detections (after reconcile + sort) */
} ato_dump_t;

void ato_default_params(ato_params_t* p);
/* Built-in family tables (include/apriltag_amd_families.h). Returns 0 on success. */
int ato_builtin_family(const char* name, ato_family_t* out);
/* A family given as data (AprilTag-3 style layout); codes must stay valid while the family is used.  Returns 0, or -1 if
 * the layout is not closed under the 90-degree rotation or does not fit (nbits <= 64, total_width <= 12). */
int ato_custom_family(const char* name, uint32_t nbits, const int8_t* bit_x, const int8_t* bit_y, uint32_t width_at_border,
                      uint32_t total_width, int reversed_border, const uint64_t* codes, uint32_t ncodes, ato_family_t* out);

/* stage entry points (each follows the cited public algorithm step) */
void ato_decimate(const uint8_t* in, int w, int h, int pitch, int f, uint8_t* out, int* sw, int* sh);
void ato_threshold(const uint8_t* im, int w, int h, int tile, int min_diff, uint8_t* out);
void ato_connected_components(const uint8_t* thr, int w, int h, uint32_t* label, uint32_t* csize);

/* Full pipeline. image is mono8 pitch-linear. Returns number of detections written (<= max_det),
 * or a negative error code. dump may be NULL. */
int ato_detect(const ato_params_t* prm, const ato_family_t* fams, int nfam,
               const uint8_t* image, int width, int height, int pitch,
               ato_detection_t* out, int max_det, ato_dump_t* dump);
void ato_dump_free(ato_dump_t* d);

/* Front steps (resize / rectify of mono8 frames), same fixed-point definitions as the HIP kernels in
 * isaac_ros_apriltag_amd/csrc/kernels_frontend.h. */
void ato_resize_mono8(const uint8_t* src, int spitch, int sw, int sh, uint8_t* dst, int dpitch, int dw, int dh);
void ato_rectify_mono8(const uint8_t* src, int spitch, uint8_t* dst, int dpitch, int w, int h, const double K[9],
                       const double D[5], const double Knew[9]);

/* Pose from a detection homography ("reference homography solve", AprilRobotics
 * estimate_pose_for_tag_homography). */
void ato_pose_from_homography(const double H[9], double fx, double fy, double cx, double cy,
                              double tag_size, double R[9], double t[3]);
/* same with a skew term in K and a choice of formulation (ATO_VAR_SVD_POLAR) */
void ato_pose_from_homography_ex(const double H[9], double fx, double fy, double cx, double cy, double skew,
                                 double tag_size, int variant, double R[9], double t[3]);

#ifdef __cplusplus
}
#endif
#endif
