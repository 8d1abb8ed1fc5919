/* apriltag_amd_debug.h -- measurement and stage-inspection entry points of libapriltag_amd.so.
 *
 * NOT part of the drop-in boundary (include/apriltag_amd.h mirrors the three cuAprilTags calls of
 * isaac_ros_apriltag/src/apriltag_node.cpp:450-452,491-493,556 and nothing a node does not need): these are what
 * bench.py (per-stage HIP-event times, the threshold-only launch behind the roofline figure) and the parity tests
 * (intermediate buffers of every stage, the device arithmetic self check) call.
 */
#ifndef APRILTAG_AMD_DEBUG_H_
#define APRILTAG_AMD_DEBUG_H_
#include "apriltag_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- measurement ------------------------------------------------------------------------- */
#define AMDAT_NUM_STAGES 12
/* Stage names, index-aligned with amdAprilTagsGetStageMs. */
const char* amdAprilTagsStageName(uint32_t stage);
/* enable != 0: bracket every stage of subsequent submissions with HIP events on the submission
 * stream.  enable == 2 additionally accumulates per-phase shader-cycle counters inside the quad-fit
 * kernel (AMDAT_DBG_FQPROF; perturbs its timing). */
int amdAprilTagsSetProfiling(amdAprilTagsHandle handle, int enable);
/* Milliseconds per stage of the last submission (AMDAT_NUM_STAGES floats). */
int amdAprilTagsGetStageMs(amdAprilTagsHandle handle, float* ms);
/* Runs only the threshold pass (S1+S2) on n frames; used by the roofline measurement. */
int amdAprilTagsThresholdOnly(amdAprilTagsHandle handle, uint32_t n, const amdAprilTagsImageInput_t* images,
                              amdAprilTagsStream stream);
/* The same on colour frames: at tile_size 4, decimate 1 the one-pass kernel with the colour loader (3 or 4 bytes read, the gray
 * plane and the threshold image written: 5 or 6 bytes per pixel), otherwise the conversion launch and the mono8 pass. */
int amdAprilTagsThresholdOnlyColor(amdAprilTagsHandle handle, uint32_t n, const amdAprilTagsImageInput_t* images,
                                   amdAprilTagsEncoding encoding, amdAprilTagsStream stream);

/* Which launch set a submission gets.  The library picks it from the submission's size: up to eight 1080p frames' worth of
 * working pixels take the LATENCY set (k_cc_local<16>, every small cluster in the one-wave class of k_fit_quads, no
 * k_fit_small, CU-wide prefilter, one select chunk, captured-graph replay); above that the THROUGHPUT set (k_cc_local<4>,
 * k_fit_small<2> for clusters up to 128 points, per-wave prefilter, chunked select).  Results never depend on it.  The parity
 * tests pin it so that BOTH sets are compared with the oracle stage by stage at any frame count (a one-frame submission on the
 * THROUGHPUT set; a 12-frame one on the LATENCY set). */
#define AMDAT_PATH_AUTO 0
#define AMDAT_PATH_LATENCY 1
#define AMDAT_PATH_THROUGHPUT 2
int amdAprilTagsDebugSetSubmissionPath(amdAprilTagsHandle handle, int path);
/* AMDAT_PATH_LATENCY or AMDAT_PATH_THROUGHPUT: the set the handle's last submission ran (AMDAT_PATH_AUTO before the first). */
int amdAprilTagsDebugLastSubmissionPath(amdAprilTagsHandle handle);
/* Launches of this handle whose stream wait returned before their results were there (every launch's counters carry its sequence
 * number; the library then waits for the whole device and checks again): 0 on a healthy runtime. */
int amdAprilTagsDebugLateWaits(amdAprilTagsHandle handle);
/* Captured-graph replay of small submissions: returns 1 while the handle still captures launch graphs for new submission shapes,
 * 0 once it has stopped (a failed capture, eight cache evictions in a row, or too many retired graphs: such a handle replays the
 * graphs it has and enqueues everything else plainly, ~0.1 ms more per one-frame call), -1 for a null handle; the counts of live
 * cache entries and of retired graphs (kept until the handle is destroyed, see csrc/detector.hip: retire_graph) through the outputs. */
int amdAprilTagsDebugGraphReplay(amdAprilTagsHandle handle, uint32_t* live_graphs, uint32_t* retired_graphs);

/* ---- stage inspection (parity tests) ------------------------------------------------------ */
typedef enum {
  AMDAT_DBG_GRAY = 0,      /* u8  w*h working gray image */
  AMDAT_DBG_THRESH = 1,    /* u8  w*h */
  AMDAT_DBG_LABEL = 2,     /* u32 w*h canonical representative or 0xFFFFFFFF */
  AMDAT_DBG_CSIZE = 3,     /* u32 w*h, valid at representatives */
  AMDAT_DBG_CLUSTERS = 4,  /* {u64 key; u32 start; u32 count} x nclusters */
  AMDAT_DBG_POINTS = 5,    /* u32 packed points, grouped by cluster */
  AMDAT_DBG_QUADS = 6,     /* {float p[4][2]; i32 reversed_border; u32 pad; u64 key} x nquads */
  AMDAT_DBG_COUNTS = 7,    /* u32[8]: npoints_raw, nclusters, npoints_kept, nquads, ndets, flags, w, h */
  AMDAT_DBG_FQPROF = 8     /* u64[64]: shader-cycle totals, 8 phases x up to 8 size classes of the quad-fit kernel (profiling on) */
} amdAprilTagsDebugBuffer;
/* Copies an intermediate buffer of frame `frame` of the last submission to host memory.
 * Returns the number of bytes the buffer holds through *bytes (copy truncated to capacity). */
int amdAprilTagsDebugCopy(amdAprilTagsHandle handle, uint32_t frame, amdAprilTagsDebugBuffer what,
                          void* host_dst, size_t capacity, size_t* bytes);
/* Device-arithmetic self check: op 0 = sqrt(f64), 1 = a/b (f64), 2 = sqrtf(f32 bits in low word),
 * 3 = a/b (f32), 4 = a/b (f64) through the shared-reciprocal sequence the line fit uses, 5 = square root of the
 * integer a < 2^18 through the line-fit weights' f32-seeded sequence.  n pairs in, n results out (host pointers). */
int amdAprilTagsDebugMath(int op, uint32_t n, const double* a, const double* b, double* out);

#ifdef __cplusplus
}
#endif
#endif /* APRILTAG_AMD_DEBUG_H_ */
