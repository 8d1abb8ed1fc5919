/* apriltag_amd.h -- C ABI of the MI355X-native AprilTag detector (libapriltag_amd.so).
 *
 * Drop-in boundary for the detector calls the reference node makes into NVIDIA's closed
 * cuAprilTags library (paths relative to /root/reference/isaac_ros_apriltag/):
 *
 *   amdCreateAprilTagsDetector   replaces nvCreateAprilTagsDetector   src/apriltag_node.cpp:450-452
 *   amdAprilTagsDetect           replaces cuAprilTagsDetect           src/apriltag_node.cpp:491-493
 *   amdAprilTagsDestroy          replaces cuAprilTagsDestroy          src/apriltag_node.cpp:556
 *   amdAprilTagsImageInput_t     replaces cuAprilTagsImageInput_t     src/apriltag_node.cpp:481-486
 *   amdAprilTagsID_t             replaces cuAprilTagsID_t             src/apriltag_node.cpp:412-419,509-516
 *   amdAprilTagsCameraIntrinsics_t replaces cuAprilTagsCameraIntrinsics_t  src/apriltag_node.cpp:447
 *   amdAprilTagsFamily           replaces cuAprilTagsFamily           src/apriltag_node.cpp:401-407
 *
 * Differences that are part of the contract (north_star of BASELINE.json):
 *   - the image is mono8 (1 byte/pixel, pitch-linear, DEVICE memory) for the calls above; colour frames as the reference
 *     feeds them -- rgb8 / bgr8 `uchar3` on its cuAprilTags branch (src/apriltag_node.cpp:469-486), the encoding table of
 *     :76-82 on its VPI branch -- go through amdAprilTagsDetectColor / amdAprilTagsDetectBatchColor /
 *     amdAprilTagsSubmitBatchColor, whose threshold pass reads the interleaved frame itself (no separate conversion launch
 *     at tile_size 4, decimate 1); amdAprilTagsConvertToMono8 remains as the stand-alone conversion
 *     (vpiSubmitConvertImageFormat, src/apriltag_node.cpp:275-282);
 *   - a batched entry point (amdAprilTagsDetectBatch) processes independent frames in one
 *     submission -- also in two halves, amdAprilTagsSubmitBatch / amdAprilTagsWaitBatch, so that a host
 *     overlaps its next host-to-device copy with the detection; a HIP stream replaces the CUDA stream;
 *   - more than one tag family can be enabled (amdCreateAprilTagsDetectorEx); tile_size is the
 *     reference's parameter (src/apriltag_node.cpp:566, handed over at :451): 4 or 8;
 *   - the skew K[0][1] of the reference's VPI path (src/apriltag_node.cpp:215-225) is a field of the
 *     configuration and, per frame of a batch, amdAprilTagsSetFrameSkews.
 * Ownership and threading follow the reference's use: the caller owns the input buffers and the
 * output arrays (host memory); the library owns everything behind the handle; one thread per handle;
 * every Detect call is host-synchronous (results valid on return).  All functions return 0 on
 * success and a non-zero amdAprilTagsStatus otherwise (the node drops the frame on a non-zero
 * detect status and throws on a non-zero create status, src/apriltag_node.cpp:453-457,494-497).
 *
 * Plain C, no C++ or torch types.  hipStream_t is passed as void* so that this header needs no HIP
 * include; pass NULL for the detector's own stream.
 */
#ifndef APRILTAG_AMD_H_
#define APRILTAG_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct amdAprilTagsDetector_st* amdAprilTagsHandle;
typedef void* amdAprilTagsStream; /* hipStream_t */

typedef enum {
  AMDAT_SUCCESS = 0,
  AMDAT_INVALID_ARGUMENT = 1,
  AMDAT_UNSUPPORTED = 2,      /* tile size / family / encoding not supported */
  AMDAT_HIP_ERROR = 3,
  AMDAT_SIZE_MISMATCH = 4,    /* image size differs from the size given at creation */
  AMDAT_OUT_OF_MEMORY = 5,
  AMDAT_BATCH_TOO_LARGE = 6
} amdAprilTagsStatus;

/* Per-frame status bits reported by amdAprilTagsGetFrameFlags (capacity overflows are reported,
 * never undefined behaviour). */
#define AMDAT_FLAG_POINTS_OVERFLOW 0x1u    /* boundary points dropped */
#define AMDAT_FLAG_HASH_OVERFLOW 0x2u      /* cluster table full */
#define AMDAT_FLAG_CLUSTERS_OVERFLOW 0x4u  /* cluster list full */
#define AMDAT_FLAG_QUADS_OVERFLOW 0x8u     /* quad list full */
#define AMDAT_FLAG_DETS_OVERFLOW 0x10u     /* detection list full */

typedef enum {
  AMDAT_TAG36H11 = 0,   /* 587 codes (include/apriltag_amd_families.h states the provenance of every table) */
  AMDAT_TAG25H9 = 1,    /* 35 codes */
  AMDAT_TAG16H5 = 2,    /* 30 codes */
  AMDAT_TAG36H10 = 3,   /* no built-in table: the offline regeneration does not reproduce the published 2320 codes
                         * (include/apriltag_amd_families.h); the slot accepts a table through amdAprilTagsRegisterFamily */
  AMDAT_CUSTOM0 = 4,    /* slots filled by amdAprilTagsRegisterFamily[Ex] */
  AMDAT_CUSTOM1 = 5,
  AMDAT_CUSTOM2 = 6,
  AMDAT_CUSTOM3 = 7,
  AMDAT_CUSTOM4 = 8,
  AMDAT_ENUM_SIZE = 9
} amdAprilTagsFamily;

typedef struct {
  float fx, fy, cx, cy;
} amdAprilTagsCameraIntrinsics_t;

/* Pixel layout of the frames of a colour submission: the ROS encoding strings the reference accepts
 * (src/apriltag_node.cpp:76-82; its cuAprilTags branch takes rgb8 / bgr8 only, :469-476). */
typedef enum {
  AMDAT_ENC_MONO8 = 0,
  AMDAT_ENC_RGB8 = 1,
  AMDAT_ENC_BGR8 = 2,
  AMDAT_ENC_RGBA8 = 3,
  AMDAT_ENC_BGRA8 = 4
} amdAprilTagsEncoding;
/* "mono8", "rgb8", "bgr8", "rgba8", "bgra8" -> amdAprilTagsEncoding; -1 for any other string. */
int amdAprilTagsEncodingFromName(const char* name);

/* Field NAMES follow cuAprilTagsImageInput_t as the reference assigns them (width, height, dev_ptr, pitch:
 * src/apriltag_node.cpp:481-486), so its binding compiles unchanged; the LAYOUT is this library's own (the closed header's
 * is not known), i.e. name-compatible, not binary-compatible, with cuAprilTags.  In the *Color calls dev_ptr is the
 * interleaved frame (the reference's uchar3*) and pitch its row stride in bytes. */
typedef struct {
  uint32_t width;
  uint32_t height;
  const uint8_t* dev_ptr; /* mono8 (or, in the *Color calls, interleaved colour), device memory */
  size_t pitch;           /* bytes per row; below 2^24, and pitch * height below 2^31 (AMDAT_INVALID_ARGUMENT otherwise) */
} amdAprilTagsImageInput_t;

typedef struct {
  float x, y;
} amdFloat2;

/* One detection.  The leading fields have the meaning the reference reads from cuAprilTagsID_t:
 * corners in the library's native order (= message order, src/apriltag_node.cpp:512-517; for an
 * upright tag: top-left, top-right, bottom-right, bottom-left), orientation COLUMN-major 3x3
 * (src/apriltag_node.cpp:416-419), translation in metres. */
typedef struct {
  uint16_t id;
  amdFloat2 corners[4];
  uint16_t hamming_error;
  float orientation[9];
  float translation[3];
  /* extensions */
  uint16_t family;        /* amdAprilTagsFamily */
  uint16_t reserved;
  float decision_margin;
  amdFloat2 center;       /* H(0,0); the reference recomputes it from the diagonals (:519-530) */
} amdAprilTagsID_t;

/* Full-precision record (parity tests, pose consumers): AprilRobotics conventions. */
typedef struct {
  int32_t family; /* index into the detector's family list */
  int32_t id;
  int32_t hamming;
  float decision_margin;
  double H[9];    /* row-major homography, tag [-1,1]^2 -> pixels */
  double c[2];    /* centre */
  double p[4][2]; /* H(-1,1), H(1,1), H(1,-1), H(-1,-1) */
  double R[9];    /* row-major rotation, tag frame in the camera optical frame */
  double t[3];    /* metres */
} amdAprilTagsDetectionEx_t;

typedef struct {
  uint32_t struct_size;        /* sizeof(amdAprilTagsConfig_t) of the header the caller was built against: set by
                                * amdAprilTagsDefaultConfig -- every configuration must start from that call.  Round 5's header is the
                                * FIRST versioned layout (struct_size went in at offset 0 then: a one-time ABI break against the
                                * rounds before it, which had no size field -- binaries built against those headers must be rebuilt;
                                * amdAprilTagsConfigLayoutVersion() says which layout a library speaks).  From that layout on the
                                * struct only grows at its end: a caller built against an older, shorter versioned header passes its
                                * smaller size and the fields it does not know keep their defaults (round 6 appended
                                * no_graph_replay and no_stream_priorities); a size below the first versioned layout's, or beyond the library's own, is
                                * AMDAT_INVALID_ARGUMENT (a struct that did not come from amdAprilTagsDefaultConfig). */
  uint32_t width, height;      /* input image size (fixed for the handle, as in the reference) */
  uint32_t tile_size;          /* 4 (src/apriltag_node.cpp:566) or 8; other values: AMDAT_UNSUPPORTED */
  uint32_t decimate;           /* quad_decimate, integer >= 1 (1 = cuAprilTags behaviour) */
  uint32_t num_families;       /* 1..4 */
  amdAprilTagsFamily families[4];
  amdAprilTagsCameraIntrinsics_t intrinsics;
  float tag_size;              /* metres (src/apriltag_node.cpp:565) */
  uint32_t max_batch;          /* frames per submission the handle is sized for (>= 1) */
  uint32_t refine_edges;       /* 1 */
  uint32_t max_hamming;        /* 2 */
  float decode_sharpening;     /* 0.25 */
  /* capacities per frame; 0 = defaults derived from the image size */
  uint32_t max_points;         /* boundary points; default 2 per working pixel */
  uint32_t hash_slots;         /* power of two */
  uint32_t max_clusters;       /* 0: starts at 65 536 and grows to the fullest frame's count when a frame fills it (up to one cluster per slot
                                * of the largest pair table); an explicit value is never grown and reports AMDAT_FLAG_CLUSTERS_OVERFLOW */
  uint32_t max_quads;          /* 0: starts at min(cluster capacity, 16 384) and doubles when a frame fills it; an explicit value is
                                * never grown and reports AMDAT_FLAG_QUADS_OVERFLOW */
  uint32_t max_detections;     /* 0: 1024 decoded candidates per frame (before the same-id overlap test), at most 65 535; not grown: a frame
                                * with more reports AMDAT_FLAG_DETS_OVERFLOW */
  int32_t device;              /* HIP device ordinal, -1 = current */
  float skew;                  /* K[0][1] of the pinhole matrix; 0 on the cuAprilTags-shaped path, the VPI path of
                                * the reference passes it with its 2x3 intrinsics (src/apriltag_node.cpp:215-225) */
  uint32_t corner_convention;  /* amdAprilTagsID_t only (amdAprilTagsDetectionEx_t always carries AprilRobotics' own):
                                * AMDAT_CORNERS_DEFAULT: corners[i] = p[3 - i], R as solved -- the reading of the reference's
                                * golden frame this library was built to (test/isaac_ros_apriltag_pol_test.py:132-175);
                                * AMDAT_CORNERS_ROTATED_180: the other reading of that frame -- corner index turned by two
                                * (corners[i] = p[(5 - i) & 3]) and the tag frame turned about its normal, R * Rz(pi) -- kept
                                * selectable until output of the closed library itself is available */
  uint32_t no_graph_replay;    /* != 0: small submissions are never stream-captured into launch graphs (plain enqueues, ~0.1 ms more per
                                * one-frame call).  For hosts whose OTHER threads make legacy-stream HIP calls (hipMemcpy, hipMemset on
                                * stream 0) on the same device while this handle detects: on ROCm 7 such a call fails -- in the host's
                                * thread -- whenever it meets a capture in progress, whatever the capture mode.  The
                                * library itself survives the collision either way (the submission goes out uncaptured). */
  uint32_t no_stream_priorities; /* != 0: a throughput-sized handle (max_batch > 8) creates its side streams without priorities (about
                                * 2 % fewer frames per second on 256-frame submissions).  For processes that ALSO hold handles of up to
                                * eight frames and create them AFTER the throughput-sized one: on ROCm 7 the launch graphs such a handle
                                * replays find their branches on the prioritised handle's hardware queues and run 30 % slower
                                * ("stream priorities").  Creating the small handles first, or one process per
                                * handle -- a node's shape -- needs nothing. */
} amdAprilTagsConfig_t;
#define AMDAT_CORNERS_DEFAULT 0u
#define AMDAT_CORNERS_ROTATED_180 1u

void amdAprilTagsDefaultConfig(amdAprilTagsConfig_t* cfg, uint32_t width, uint32_t height);
/* Layout generation of amdAprilTagsConfig_t this library was built with: 1 = the first versioned layout (struct_size at offset 0,
 * fields up to corner_convention), 2 = + no_graph_replay, 3 = + no_stream_priorities.  Layouts before 1 (no size field) are not
 * accepted. */
#define AMDAT_CONFIG_LAYOUT_VERSION 3
uint32_t amdAprilTagsConfigLayoutVersion(void);

/* nvCreateAprilTagsDetector-shaped constructor (one family, batch 1, decimate 1). */
int amdCreateAprilTagsDetector(amdAprilTagsHandle* handle, uint32_t img_width, uint32_t img_height,
                               uint32_t tile_size, amdAprilTagsFamily tag_family,
                               const amdAprilTagsCameraIntrinsics_t* cam, float tag_dim);
int amdCreateAprilTagsDetectorEx(amdAprilTagsHandle* handle, const amdAprilTagsConfig_t* cfg);
int amdAprilTagsDestroy(amdAprilTagsHandle handle);

/* cuAprilTagsDetect-shaped call: one frame, blocking. */
int amdAprilTagsDetect(amdAprilTagsHandle handle, const amdAprilTagsImageInput_t* img_input,
                       amdAprilTagsID_t* tags_out, uint32_t* num_tags, uint32_t max_tags,
                       amdAprilTagsStream stream);

/* The same call on a colour frame, as the reference's cuAprilTags branch makes it (uchar3 rgb8 / bgr8 device image,
 * src/apriltag_node.cpp:469-493): the gray value is the fixed-point BT.601 statement of amdAprilTagsConvertToMono8, formed by the
 * threshold pass's loader; results equal those of the mono8 call on the converted frame bit for bit.  encoding AMDAT_ENC_MONO8
 * is amdAprilTagsDetect. */
int amdAprilTagsDetectColor(amdAprilTagsHandle handle, const amdAprilTagsImageInput_t* img_input, amdAprilTagsEncoding encoding,
                            amdAprilTagsID_t* tags_out, uint32_t* num_tags, uint32_t max_tags, amdAprilTagsStream stream);

/* Batched call: n independent frames (n <= max_batch).  tags_out holds n*max_tags records,
 * num_tags n counts.  per_frame_intrinsics may be NULL (handle intrinsics used for all frames). */
int amdAprilTagsDetectBatch(amdAprilTagsHandle handle, uint32_t n, const amdAprilTagsImageInput_t* images,
                            const amdAprilTagsCameraIntrinsics_t* per_frame_intrinsics,
                            amdAprilTagsID_t* tags_out, uint32_t* num_tags, uint32_t max_tags,
                            amdAprilTagsStream stream);
/* Same, full-precision records. */
int amdAprilTagsDetectBatchEx(amdAprilTagsHandle handle, uint32_t n, const amdAprilTagsImageInput_t* images,
                              const amdAprilTagsCameraIntrinsics_t* per_frame_intrinsics,
                              amdAprilTagsDetectionEx_t* dets_out, uint32_t* num_dets, uint32_t max_dets,
                              amdAprilTagsStream stream);

/* The batched calls on colour frames (all frames of a submission share one encoding). */
int amdAprilTagsDetectBatchColor(amdAprilTagsHandle handle, uint32_t n, const amdAprilTagsImageInput_t* images,
                                 amdAprilTagsEncoding encoding, const amdAprilTagsCameraIntrinsics_t* per_frame_intrinsics,
                                 amdAprilTagsID_t* tags_out, uint32_t* num_tags, uint32_t max_tags, amdAprilTagsStream stream);
int amdAprilTagsDetectBatchColorEx(amdAprilTagsHandle handle, uint32_t n, const amdAprilTagsImageInput_t* images,
                                   amdAprilTagsEncoding encoding, const amdAprilTagsCameraIntrinsics_t* per_frame_intrinsics,
                                   amdAprilTagsDetectionEx_t* dets_out, uint32_t* num_dets, uint32_t max_dets,
                                   amdAprilTagsStream stream);

/* The batched call in two halves, for hosts that overlap the NEXT batch's host-to-device copy (on a stream of their own) with
 * this batch's detection: amdAprilTagsSubmitBatch enqueues the submission and returns; amdAprilTagsWaitBatch[Ex] blocks until
 * it is done and hands out the records (layout as amdAprilTagsDetectBatch[Ex] with the max_tags given at submit).  One
 * submission per handle at a time: a second Submit, or any other call that runs or inspects a submission, returns
 * AMDAT_INVALID_ARGUMENT until the wait has returned, and so does a wait with nothing in flight.  Images and their device
 * buffers must stay valid until the wait returns.  The blocking calls are exactly Submit followed by Wait. */
int amdAprilTagsSubmitBatch(amdAprilTagsHandle handle, uint32_t n, const amdAprilTagsImageInput_t* images,
                            const amdAprilTagsCameraIntrinsics_t* per_frame_intrinsics, uint32_t max_tags,
                            amdAprilTagsStream stream);
int amdAprilTagsSubmitBatchColor(amdAprilTagsHandle handle, uint32_t n, const amdAprilTagsImageInput_t* images,
                                 amdAprilTagsEncoding encoding, const amdAprilTagsCameraIntrinsics_t* per_frame_intrinsics,
                                 uint32_t max_tags, amdAprilTagsStream stream);
int amdAprilTagsWaitBatch(amdAprilTagsHandle handle, amdAprilTagsID_t* tags_out, uint32_t* num_tags);
int amdAprilTagsWaitBatchEx(amdAprilTagsHandle handle, amdAprilTagsDetectionEx_t* dets_out, uint32_t* num_dets);

/* Per-frame skew K[0][1] for the batch slots 0 .. n-1 of the submissions that follow (frames beyond n, and every frame after
 * n = 0, take the handle's amdAprilTagsConfig_t.skew).  The VPI path of the reference passes every camera's own skew with its
 * 2x3 intrinsics (src/apriltag_node.cpp:215-225); a batching front end that serves several cameras with one handle sets them
 * here next to the per-frame fx, fy, cx, cy of amdAprilTagsDetectBatch.  Only the pose depends on it. */
int amdAprilTagsSetFrameSkews(amdAprilTagsHandle handle, uint32_t n, const float* skews);

/* Device memory the handle owns, in bytes. */
int amdAprilTagsGetDeviceBytes(amdAprilTagsHandle handle, size_t* bytes);

/* Status bits of the frames of the last submission (n values). */
int amdAprilTagsGetFrameFlags(amdAprilTagsHandle handle, uint32_t* flags, uint32_t n);

/* Colour -> mono8 on the device.  encoding: "mono8","rgb8","bgr8","rgba8","bgra8"
 * (src/apriltag_node.cpp:76-82). */
int amdAprilTagsConvertToMono8(const void* src_dev, size_t src_pitch, const char* encoding, uint32_t width,
                               uint32_t height, uint8_t* dst_dev, size_t dst_pitch, amdAprilTagsStream stream);

/* ---- front steps of the usual graph (camera -> rectify -> resize -> apriltag; reference README.md:16-29,
 * launch/isaac_ros_apriltag_usb_cam.launch.py:43-63) on mono8 device images ------------------------- */
/* Bilinear resize, pixel-centre aligned; source coordinates and weights are 1/2048 fixed point, so the
 * result is exactly reproducible (oracle: ato_resize_mono8). */
int amdAprilTagsResizeMono8(const uint8_t* src_dev, size_t src_pitch, uint32_t src_width, uint32_t src_height,
                            uint8_t* dst_dev, size_t dst_pitch, uint32_t dst_width, uint32_t dst_height,
                            amdAprilTagsStream stream);
/* Undistortion of a plumb_bob image (sensor_msgs/CameraInfo K and D = k1,k2,p1,p2,k3) onto the pinhole
 * camera K_new (row-major 3x3 each): every destination pixel is projected through the distortion model
 * in double precision, the source position is quantised to 1/32 pixel and sampled bilinearly in integer
 * arithmetic; pixels that map outside the source are 0 (oracle: ato_rectify_mono8). */
int amdAprilTagsRectifyMono8(const uint8_t* src_dev, size_t src_pitch, uint8_t* dst_dev, size_t dst_pitch, uint32_t width,
                             uint32_t height, const double* K9, const double* D5, const double* Knew9,
                             amdAprilTagsStream stream);

/* Device-memory helpers for hosts that do not link the HIP runtime themselves (the node shell copies
 * sensor_msgs/Image payloads with these).  Plain hipMalloc / hipFree / hipMemcpyAsync + sync. */
int amdAprilTagsDeviceAlloc(void** dev_ptr, size_t bytes);
int amdAprilTagsDeviceFree(void* dev_ptr);
int amdAprilTagsCopyToDevice(void* dst_dev, const void* src_host, size_t bytes, amdAprilTagsStream stream);
/* Enqueue-only form on a stream of the host's (amdAprilTagsStreamCreate): the copy is ordered ahead of a detection submitted on the
 * same stream, whose wait then covers both; the source must stay valid until that detection has returned. */
int amdAprilTagsCopyToDeviceAsync(void* dst_dev, const void* src_host, size_t bytes, amdAprilTagsStream stream);
int amdAprilTagsStreamCreate(amdAprilTagsStream* stream);    /* a non-blocking HIP stream */
int amdAprilTagsStreamDestroy(amdAprilTagsStream stream);   /* waits for it first */

/* Registers a tag family as data (row-major codes, MSB = top-left data cell) in a custom slot. */
int amdAprilTagsRegisterFamily(amdAprilTagsFamily slot, const char* name, uint32_t data_bits_per_side,
                               const uint64_t* codes, uint32_t ncodes);
/* The same for an AprilTag-3 style layout -- what the circle / standard / custom families of the reference's table
 * (src/apriltag_node.cpp:47-58: circle21h7, circle49h12, custom48h12, standard41h12, standard52h13) need: nbits data bits
 * (<= 64), bit i at cell (bit_x[i], bit_y[i]) in border coordinates ((0, 0) = top-left cell of the border square of
 * width_at_border cells; cells of outer rings are negative or >= width_at_border), total_width cells across everything
 * (<= 12, same parity as width_at_border), reversed_border != 0 when the border square is white inside a black ring.
 * Bit (nbits - 1 - i) of a code is data bit i (AprilTag 3's own convention), 1 = white.  The layout must map onto itself
 * under the quarter turn (x, y) -> (width_at_border - 1 - y, x); AMDAT_INVALID_ARGUMENT otherwise.  This library ships
 * no code tables for those five families (none can be verified offline); a host that has them registers them here under
 * the reference's names and amdAprilTagsFamilyFromName / the node shell find them. */
int amdAprilTagsRegisterFamilyEx(amdAprilTagsFamily slot, const char* name, uint32_t nbits, const int8_t* bit_x,
                                 const int8_t* bit_y, uint32_t width_at_border, uint32_t total_width, int reversed_border,
                                 const uint64_t* codes, uint32_t ncodes);
/* Registered names are at most 31 characters and code words carry no bits above the family's width (AMDAT_INVALID_ARGUMENT
 * otherwise).  amdAprilTagsUnregisterFamily empties a registrable slot again (handles created earlier keep their copy of the
 * table). */
int amdAprilTagsUnregisterFamily(amdAprilTagsFamily slot);
/* Family metadata: returns 0 and fills the outputs if the family is known.  The pointers stay valid until the slot is
 * registered again or emptied. */
int amdAprilTagsFamilyInfo(amdAprilTagsFamily family, const char** name, uint32_t* data_bits_per_side,
                           uint32_t* ncodes, const uint64_t** codes);
/* Family lookup by the reference's parameter string (src/apriltag_node.cpp:47-58); -1 if unknown
 * or without an offline codebook. */
int amdAprilTagsFamilyFromName(const char* name);

/* Measurement and stage-inspection entry points (profiling, the threshold-only launch of the roofline measurement,
 * intermediate buffers, the device arithmetic self check) are declared in apriltag_amd_debug.h: they are exported by
 * the same library but are not part of the detector boundary a node binds. */

#ifdef __cplusplus
}
#endif
#endif /* APRILTAG_AMD_H_ */
