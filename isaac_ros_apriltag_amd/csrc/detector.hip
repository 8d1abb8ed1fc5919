// detector.hip -- host side of libapriltag_amd.so: the C ABI declared in include/apriltag_amd.h,
// buffer management, and the launch sequence of one batched submission.
//
// Replaces the closed cuAprilTags calls of the reference node
// (src/apriltag_node.cpp:450-452 create, :491-493 detect, :556 destroy).  gfx950 only.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/apriltag_amd.h"
#include "../../include/apriltag_amd_debug.h"
#include "../../include/apriltag_amd_families.h"
#include "common.h"
#include "kernels_cc.h"
#include "kernels_cluster.h"
#include "kernels_decode.h"
#include "kernels_decode_wave.h"
#include "kernels_frontend.h"
#include "kernels_quad.h"
#include "kernels_quad_small.h"
#include "kernels_threshold.h"

static_assert(sizeof(DetRec) == sizeof(amdAprilTagsDetectionEx_t), "DetRec must match the public record");
static_assert(sizeof(QuadRec) == 48, "QuadRec layout");
static_assert(sizeof(ClusterRec) == 16, "ClusterRec layout");

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) {                                                                        \
      fprintf(stderr, "[apriltag_amd] %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return AMDAT_HIP_ERROR;                                                                      \
    }                                                                                              \
  } while (0)

// ------------------------------------------------------------------------------------------------
// family registry
// ------------------------------------------------------------------------------------------------
namespace {
struct FamilyHost {
  const char* name = nullptr;
  uint32_t d = 0;            // data cells per side of a classic family, 0 for other layouts
  uint32_t nbits = 0, width_at_border = 0, total_width = 0;
  int reversed_border = 0;
  int8_t bit_x[64] = {0}, bit_y[64] = {0};
  uint8_t rot_src[64] = {0};
  uint32_t ncodes = 0;
  const uint64_t* codes = nullptr;
  std::vector<uint64_t> owned;
  char name_buf[32] = {0};
};
// Fills the rotation table; false if the layout does not map onto itself under (x, y) -> (wb - 1 - y, x) or repeats a cell.
bool family_finish_layout(FamilyHost& f) {
  for (uint32_t i = 0; i < f.nbits; i++) {
    const int sx = (int)f.width_at_border - 1 - f.bit_y[i], sy = f.bit_x[i];
    int src = -1;
    for (uint32_t j = 0; j < f.nbits; j++) {
      if (f.bit_x[j] == sx && f.bit_y[j] == sy) src = (int)j;
      if (j < i && f.bit_x[j] == f.bit_x[i] && f.bit_y[j] == f.bit_y[i]) return false;
    }
    if (src < 0) return false;
    f.rot_src[i] = (uint8_t)src;
  }
  return true;
}
void family_set_classic(FamilyHost& f, uint32_t d) {
  f.d = d; f.nbits = d * d; f.width_at_border = d + 2; f.total_width = d + 4; f.reversed_border = 0;
  for (uint32_t i = 0; i < f.nbits; i++) { f.bit_x[i] = (int8_t)(1 + i % d); f.bit_y[i] = (int8_t)(1 + i / d); }
  family_finish_layout(f);
}
FamilyHost g_families[AMDAT_ENUM_SIZE];
std::once_flag g_fam_once;
std::mutex g_fam_mutex;

void init_families() {
  g_families[AMDAT_TAG36H11].name = "tag36h11";
  family_set_classic(g_families[AMDAT_TAG36H11], 6);
  g_families[AMDAT_TAG36H11].ncodes = APRILTAG_AMD_TAG36H11_NCODES;
  g_families[AMDAT_TAG36H11].codes = apriltag_amd_tag36h11_codes;
  g_families[AMDAT_TAG25H9].name = "tag25h9";
  family_set_classic(g_families[AMDAT_TAG25H9], 5);
  g_families[AMDAT_TAG25H9].ncodes = APRILTAG_AMD_TAG25H9_NCODES;
  g_families[AMDAT_TAG25H9].codes = apriltag_amd_tag25h9_codes;
  g_families[AMDAT_TAG16H5].name = "tag16h5";
  family_set_classic(g_families[AMDAT_TAG16H5], 4);
  g_families[AMDAT_TAG16H5].ncodes = APRILTAG_AMD_TAG16H5_NCODES;
  g_families[AMDAT_TAG16H5].codes = apriltag_amd_tag16h5_codes;
  // (AMDAT_TAG36H10 starts empty: see the header)
}

bool is_registrable_slot(int slot) { return slot == AMDAT_TAG36H10 || (slot >= AMDAT_CUSTOM0 && slot < AMDAT_ENUM_SIZE); }

// roctx ranges around the stages, behind the profiling switch: the marker library is looked up at run
// time the first time profiling is on (libroctx64.so ships with ROCm; a host without it simply gets no ranges), so the
// product library has no link-time dependency on the tracing stack and an unprofiled call never touches it.
struct RoctxApi {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  RoctxApi() {
    void* h = dlopen("libroctx64.so", RTLD_LAZY | RTLD_LOCAL);
    if (!h) h = dlopen("libroctx64.so.4", RTLD_LAZY | RTLD_LOCAL);
    if (!h) return;
    push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
    pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
    if (!push || !pop) { push = nullptr; pop = nullptr; }
  }
};
static const RoctxApi& roctx() { static const RoctxApi api; return api; }

const char* kStageNames[AMDAT_NUM_STAGES] = {"upload_clear", "threshold", "cc_local",  "cc_border",
                                             "cc_sizes",   "points",    "cluster_select", "scatter",
                                             "fit_quads",    "decode",    "reconcile", "download"};

// Makes `device` current for the scope of one C-ABI call and restores the caller's device afterwards, so
// that a multi-GPU host (one handle per device in one process) never finds its current device changed.
struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device) ok = hipSetDevice(device) == hipSuccess; else prev = -1;
  }
  ~DeviceGuard() { if (prev >= 0) hipSetDevice(prev); }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};

uint32_t next_pow2(uint32_t v) {
  uint32_t p = 1;
  while (p < v) p <<= 1;
  return p;
}
}  // namespace

// One size class of the quad fit: workgroup size, LDS key capacity, cluster sizes (lo, hi], persistent grid,
// scratch slot size (points) and its slice of the work array.
// Side streams of the quad fit.  Three, as in every earlier round: a captured submission with six parallel branches (five side
// streams + the submission stream) crashed inside hipGraphLaunch (hip::Graph::UpdateStreams, ROCm 7.2) in about one of seven
// 300-case fuzz runs; with four branches it never has.  Classes that launch share the streams round-robin.
#define FQ_NAUX 3
struct FqClass {
  int nt, sort_cap, lo, hi;
  unsigned grid;
  int slot_cap;
  int pop;                                // clusters taken from the work list per atomic
  int small_k = 0;                        // > 0: k_fit_small<small_k> (keys and sweep state in registers, moments in LDS), no scratch
  double* d_lf = nullptr;                 // grid x slot_cap x 6 doubles
  double* d_errs = nullptr;               // grid x slot_cap x 2 doubles: error arrays of clusters that exceed what the
                                          // kernel keeps in LDS / registers (slot_cap > 16 x nt, or > sort_cap)
};

struct amdAprilTagsDetector_st {
  amdAprilTagsConfig_t cfg;
  DetParams P;
  int device = 0;
  int num_cus = 256;
  size_t device_bytes = 0;
  hipStream_t own_stream = nullptr;
  // the size classes of the quad fit fork to auxiliary streams and join before decode
  hipStream_t aux_stream[FQ_NAUX] = {};
  bool aux_prioritised = false;      // the side streams carry priorities (handles above eight frames per submission; see creation)
  hipEvent_t ev_fork = nullptr, ev_join[FQ_NAUX] = {};
  // device buffers
  uint8_t* d_gray = nullptr;          // working-size gray plane: decimated handles, and (allocated on first use) colour submissions at decimate 1
  uint8_t* d_conv = nullptr;          // full-size mono8 plane of colour submissions that take the conversion launch (decimate > 1, tile_size 8)
  size_t conv_pitch = 0;
  uint8_t* d_thr = nullptr;
  uint8_t* d_tmin = nullptr;         // per-tile min / max of the two-pass threshold (tile_size != 4 only)
  uint8_t* d_tmax = nullptr;
  uint32_t* d_label = nullptr;
  uint32_t* d_csize = nullptr;
  uint32_t* d_roots = nullptr;
  uint32_t* d_perim = nullptr;       // tile perimeters (class + tile-local root per pixel), k_cc_local -> k_cc_border (kernels_cc.h: CcPerim)
  unsigned long long* d_hkeys = nullptr;
  uint32_t* d_hcnt = nullptr;
  uint32_t* d_hoff = nullptr;
  uint32_t* d_stage = nullptr;       // one word per staged boundary point (kernels_cluster.h, pass 3 of k_points)
  uint2* d_bhdr = nullptr;           // per block (tile) of k_points: {first staging word, words}
  uint2* d_btab = nullptr;           // per block: its component-pair table, {pair-table slot, base rank inside the cluster} per entry
  uint4* d_long = nullptr;           // {slot, rank, packed point}: emissions without a block-table entry
  uint32_t* d_pts = nullptr;
  ClusterRec* d_clusters = nullptr;
  uint32_t* d_work = nullptr;        // work lists of the quad fit (all classes, FqWorkLayout)
  bool tables_dirty = false;         // a submission was cut short after k_points: the pair table is not empty
  uint32_t* d_workctl = nullptr;     // [0..7] items per class, [8..15] pop cursors, [16..23] items per class after k_fit_prefilter, [24] the prefilter's own cursor
  uint32_t* d_work2 = nullptr;       // compact work lists of the prefiltered classes (same layout as d_work)
  unsigned long long* d_keys_scr = nullptr;  // only when a cluster can exceed the LDS key array (large images)
  QuadRec* d_quads = nullptr;
  DetRec* d_dets = nullptr;
  uint16_t* d_order = nullptr;
  FitCand* d_cands = nullptr;        // quad candidates of k_fit_quads (four lines each), consumed by k_quad_finish
  FrameCounters* d_counters = nullptr;
  FrameDesc* d_frames = nullptr;
  uint64_t* d_codes[AT_MAX_FAMILIES] = {nullptr, nullptr, nullptr, nullptr};
  unsigned long long* d_ptprof = nullptr;  // per-phase cycle counters of k_points (same builds), inside d_fqprof's allocation
  unsigned long long* d_fqprof = nullptr;  // per-phase cycle counters of k_fit_quads (-DAMDAT_FQ_PROFILE builds only)
  FqClass cls[FQ_NCLS];
  FqWorkLayout work_layout;
  FqWorkLayout work_layout_small;    // small submissions: the k_fit_small classes are empty, the one-wave class starts at 0 (issue_pipeline)
  int prefilter_class = FQ_C0 + 2;           // first size class whose clusters go through k_fit_prefilter (those above 2048 points)
  bool grow_points = false;          // point capacity follows the content (no explicit max_points)
  bool grow_hash = false;            // the same for the component-pair table (no explicit hash_slots)
  bool grow_quads = false;           // the same for the quad list (no explicit max_quads): doubles up to the cluster capacity
  bool grow_clusters = false;        // and for the cluster list (no explicit max_clusters): up to ccap_hard
  uint32_t ccap_hard = 0;            // what the pair table and a work item's index bits admit
  size_t clusters_bytes = 0;
  size_t quads_bytes = 0;
  bool pending_hash_grow = false;
  uint32_t lcap_div = 0;             // long-record capacity = point capacity / lcap_div (alloc_point_buffers; halves when the long records overflow)
  bool unusable = false;             // a capacity change failed twice (grown and original size): buffers are gone, every later call reports it
  size_t cands_bytes = 0;
  uint32_t hcap_hard = 0;
  size_t hash_buffer_bytes[3] = {0, 0, 0};
  uint32_t pcap_hard = 0;            // 2 points per working pixel: what any content stays below
  size_t point_buffer_bytes[5] = {0, 0, 0, 0, 0};
  uint32_t grown = 0;                // number of times the point buffers grew (amdAprilTagsGetDeviceBytes reports the result)
  // pinned host buffers
  FrameDesc* h_frames = nullptr;
  FrameCounters* h_counters = nullptr;
  DetRec* h_out = nullptr;
  // profiling
  bool profiling = false;
  bool fq_counters = false;  // per-phase cycle counters inside k_fit_quads (profiling level 2; perturbs timing)
  bool fq_attr_set = false;
  // captured enqueue sequence of small submissions (see run_batch)
  uint32_t graph_max_frames = 8;
  struct GraphEntry { hipGraphExec_t exec = nullptr; uint32_t n = 0, ostride = 0, fmt = 0; hipStream_t stream = nullptr; uint64_t last_use = 0; };
  GraphEntry graphs[6];
  std::vector<hipGraphExec_t> retired_graphs;   // see drop_graphs
  uint64_t graph_clock = 0;
  uint32_t graph_misses = 0;         // consecutive captures that had to evict an entry
  uint32_t capture_failures = 0;     // captures that did not end in a graph (recover_from_failed_capture)
  hipEvent_t ev[AMDAT_NUM_STAGES + 1] = {};
  // which launch set a submission gets: by its size (AMDAT_PATH_AUTO) or pinned by amdAprilTagsDebugSetSubmissionPath, so that
  // the parity tests can put BOTH launch sets under the oracle at any frame count
  bool events_recorded = false;      // the submission in flight recorded the stage events (profiling on, no graph replay)
  struct { bool active = false; uint32_t n = 0, ostride = 0, max_out = 0; hipStream_t stream = nullptr; } inflight;   // amdAprilTagsSubmitBatch .. WaitBatch
  std::vector<float> frame_skew;     // per batch slot, amdAprilTagsSetFrameSkews; empty: cfg.skew for every frame
  uint32_t launched_n = 0;
  uint32_t launched_fmt = 0;         // amdAprilTagsEncoding of the submission in flight (a regrowth relaunch repeats it)
  uint32_t seq = 0;                  // launch counter: travels through the descriptor block and comes back with the counters
  uint32_t late_waits = 0;           // launches whose stream wait returned before their results (finish_once); amdAprilTagsDebugLateWaits
  int path_mode = AMDAT_PATH_AUTO;
  int last_path = AMDAT_PATH_AUTO;   // the set the last submission ran (amdAprilTagsDebugLastSubmissionPath)
  float stage_ms[AMDAT_NUM_STAGES] = {};
  uint32_t last_n = 0;
};

// ------------------------------------------------------------------------------------------------
// small kernels that live with the host code
// ------------------------------------------------------------------------------------------------
__global__ void k_debug_math(int op, uint32_t n, const double* a, const double* b, double* out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (op == 0) out[i] = __dsqrt_rn(a[i]);
  else if (op == 1) out[i] = a[i] / b[i];
  else if (op == 2) out[i] = (double)at_sqrtf_rn((float)a[i]);
  else if (op == 3) out[i] = (double)__fdiv_rn((float)a[i], (float)b[i]);
  else if (op == 5) out[i] = sqrt_u18((uint32_t)a[i]);     // integer arguments below 2^18
  else out[i] = div_by(a[i], b[i], shared_recip(b[i]));   // the line fit's shared-reciprocal division
}

// colour -> mono8 with the fixed-point BT.601 weights cv_bridge/OpenCV use for the reference's mono8
// test input (test/isaac_ros_apriltag_mono8_test.py): Y = (4899 R + 9617 G + 1868 B + 8192) >> 14
template <int NCH, int RIDX, int BIDX>
__global__ __launch_bounds__(256) void k_to_mono8(const uint8_t* __restrict__ src, size_t spitch, uint8_t* __restrict__ dst,
                                                  size_t dpitch, uint32_t w, uint32_t h) {
  const uint32_t x4 = (blockIdx.x * 256 + threadIdx.x) * 4;
  const uint32_t y = blockIdx.y;
  if (x4 >= w || y >= h) return;
  const uint8_t* s = src + (size_t)y * spitch + (size_t)x4 * NCH;
  uint8_t* d = dst + (size_t)y * dpitch + x4;
  for (uint32_t k = 0; k < 4 && x4 + k < w; k++) {
    const uint32_t R = s[k * NCH + RIDX], G = s[k * NCH + 1], B = s[k * NCH + BIDX];
    d[k] = (uint8_t)((4899u * R + 9617u * G + 1868u * B + 8192u) >> 14);
  }
}

// The same conversion for the frames of a colour submission that does not take the fused loader of k_threshold (decimate > 1,
// tile_size 8): source and destination of frame blockIdx.z come from its descriptor (src -> img), one launch per submission.
template <int NCH, int RIDX, int BIDX>
__global__ __launch_bounds__(256) void k_to_mono8_frames(const FrameDesc* __restrict__ frames, uint32_t w, uint32_t h) {
  const FrameDesc fd = frames[blockIdx.z];
  const uint32_t x4 = (blockIdx.x * 256 + threadIdx.x) * 4;
  const uint32_t y = blockIdx.y;
  if (x4 >= w || y >= h) return;
  const uint8_t* s = fd.src + (size_t)y * fd.src_pitch + (size_t)x4 * NCH;
  uint8_t* d = const_cast<uint8_t*>(fd.img) + (size_t)y * fd.pitch + x4;
  for (uint32_t k = 0; k < 4 && x4 + k < w; k++) {
    const uint32_t R = s[k * NCH + RIDX], G = s[k * NCH + 1], B = s[k * NCH + BIDX];
    d[k] = (uint8_t)((4899u * R + 9617u * G + 1868u * B + 8192u) >> 14);
  }
}

static inline uint32_t enc_channels(uint32_t fmt) { return fmt == AMDAT_ENC_MONO8 ? 1u : (fmt <= AMDAT_ENC_BGR8 ? 3u : 4u); }

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

int amdAprilTagsEncodingFromName(const char* name) {
  if (!name) return -1;
  static const char* const kNames[5] = {"mono8", "rgb8", "bgr8", "rgba8", "bgra8"};   // src/apriltag_node.cpp:76-82
  for (int i = 0; i < 5; i++) if (!strcmp(name, kNames[i])) return i;
  return -1;
}

void amdAprilTagsDefaultConfig(amdAprilTagsConfig_t* cfg, uint32_t width, uint32_t height) {
  memset(cfg, 0, sizeof(*cfg));
  cfg->struct_size = (uint32_t)sizeof(*cfg);
  cfg->width = width;
  cfg->height = height;
  cfg->tile_size = 4;
  cfg->decimate = 1;
  cfg->num_families = 1;
  cfg->families[0] = AMDAT_TAG36H11;
  cfg->intrinsics = {1000.0f, 1000.0f, width / 2.0f, height / 2.0f};
  cfg->tag_size = 0.22f;
  cfg->max_batch = 1;
  cfg->refine_edges = 1;
  cfg->max_hamming = 2;
  cfg->decode_sharpening = 0.25f;
  cfg->device = -1;
  cfg->no_graph_replay = 0;
  cfg->no_stream_priorities = 0;
}

uint32_t amdAprilTagsConfigLayoutVersion(void) { return AMDAT_CONFIG_LAYOUT_VERSION; }

// a registered name must fit the slot's buffer whole (a truncated name could never be found again), and a code word must not
// carry bits above the family's width (it could never match, and the Hamming search would count the stray bits)
static bool family_args_ok(const char* name, uint32_t nbits, const uint64_t* codes, uint32_t ncodes) {
  if (strlen(name) >= sizeof(FamilyHost::name_buf)) return false;
  if (nbits < 64)
    for (uint32_t i = 0; i < ncodes; i++)
      if (codes[i] >> nbits) return false;
  return true;
}

int amdAprilTagsRegisterFamily(amdAprilTagsFamily slot, const char* name, uint32_t d, const uint64_t* codes, uint32_t ncodes) {
  std::call_once(g_fam_once, init_families);
  if (!is_registrable_slot((int)slot)) return AMDAT_INVALID_ARGUMENT;
  if (!name || !codes || ncodes == 0 || d < 3 || d > 7) return AMDAT_INVALID_ARGUMENT;
  if (!family_args_ok(name, d * d, codes, ncodes)) return AMDAT_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lk(g_fam_mutex);
  FamilyHost& f = g_families[slot];
  f.owned.assign(codes, codes + ncodes);
  strncpy(f.name_buf, name, sizeof(f.name_buf) - 1);
  f.name_buf[sizeof(f.name_buf) - 1] = 0;
  f.name = f.name_buf;
  family_set_classic(f, d);
  f.ncodes = ncodes;
  f.codes = f.owned.data();
  return AMDAT_SUCCESS;
}

int amdAprilTagsRegisterFamilyEx(amdAprilTagsFamily slot, const char* name, uint32_t nbits, const int8_t* bit_x, const int8_t* bit_y,
                                 uint32_t width_at_border, uint32_t total_width, int reversed_border, const uint64_t* codes,
                                 uint32_t ncodes) {
  std::call_once(g_fam_once, init_families);
  if (!is_registrable_slot((int)slot)) return AMDAT_INVALID_ARGUMENT;
  if (!name || !bit_x || !bit_y || !codes || ncodes == 0 || nbits == 0 || nbits > 64) return AMDAT_INVALID_ARGUMENT;
  if (!family_args_ok(name, nbits, codes, ncodes)) return AMDAT_INVALID_ARGUMENT;
  if (width_at_border < 3 || total_width < width_at_border || total_width > 12 || ((total_width - width_at_border) & 1u))
    return AMDAT_INVALID_ARGUMENT;
  FamilyHost f;
  f.d = 0; f.nbits = nbits; f.width_at_border = width_at_border; f.total_width = total_width; f.reversed_border = reversed_border ? 1 : 0;
  const int min_coord = ((int)width_at_border - (int)total_width) / 2;
  for (uint32_t i = 0; i < nbits; i++) {
    if (bit_x[i] < min_coord || bit_x[i] >= min_coord + (int)total_width || bit_y[i] < min_coord || bit_y[i] >= min_coord + (int)total_width)
      return AMDAT_INVALID_ARGUMENT;
    f.bit_x[i] = bit_x[i]; f.bit_y[i] = bit_y[i];
  }
  if (!family_finish_layout(f)) return AMDAT_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lk(g_fam_mutex);
  FamilyHost& g = g_families[slot];
  g = f;
  g.owned.assign(codes, codes + ncodes);
  strncpy(g.name_buf, name, sizeof(g.name_buf) - 1);
  g.name_buf[sizeof(g.name_buf) - 1] = 0;
  g.name = g.name_buf;
  g.ncodes = ncodes;
  g.codes = g.owned.data();
  return AMDAT_SUCCESS;
}

int amdAprilTagsUnregisterFamily(amdAprilTagsFamily slot) {
  std::call_once(g_fam_once, init_families);
  if (!is_registrable_slot((int)slot)) return AMDAT_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lk(g_fam_mutex);
  FamilyHost& f = g_families[slot];
  f.codes = nullptr; f.ncodes = 0; f.name = nullptr; f.name_buf[0] = 0;
  f.owned.clear();
  return AMDAT_SUCCESS;
}

int amdAprilTagsFamilyInfo(amdAprilTagsFamily family, const char** name, uint32_t* d, uint32_t* ncodes, const uint64_t** codes) {
  std::call_once(g_fam_once, init_families);
  if ((int)family < 0 || family >= AMDAT_ENUM_SIZE) return AMDAT_UNSUPPORTED;
  std::lock_guard<std::mutex> lk(g_fam_mutex);   // (the pointers handed out stay valid until the slot is registered again)
  if (g_families[family].codes == nullptr) return AMDAT_UNSUPPORTED;
  if (name) *name = g_families[family].name;
  if (d) *d = g_families[family].d;
  if (ncodes) *ncodes = g_families[family].ncodes;
  if (codes) *codes = g_families[family].codes;
  return AMDAT_SUCCESS;
}

int amdAprilTagsFamilyFromName(const char* name) {
  std::call_once(g_fam_once, init_families);
  if (!name) return -1;
  // registered tables take precedence over a built-in of the same name
  static const int scan[AMDAT_ENUM_SIZE] = {AMDAT_CUSTOM0, AMDAT_CUSTOM1, AMDAT_CUSTOM2, AMDAT_CUSTOM3, AMDAT_CUSTOM4,
                                            AMDAT_TAG36H10, AMDAT_TAG36H11, AMDAT_TAG25H9, AMDAT_TAG16H5};
  std::lock_guard<std::mutex> lk(g_fam_mutex);
  for (int k = 0; k < AMDAT_ENUM_SIZE; k++) {
    const int i = scan[k];
    if (g_families[i].codes && g_families[i].name && !strcmp(g_families[i].name, name)) return i;
  }
  return -1;
}

const char* amdAprilTagsStageName(uint32_t stage) { return stage < AMDAT_NUM_STAGES ? kStageNames[stage] : ""; }

static void free_all(amdAprilTagsDetector_st* D) {
  for (auto& g : D->graphs) if (g.exec) hipGraphExecDestroy(g.exec);
  for (hipGraphExec_t e : D->retired_graphs) hipGraphExecDestroy(e);
  hipFree(D->d_gray); hipFree(D->d_conv); hipFree(D->d_thr); hipFree(D->d_tmin); hipFree(D->d_tmax); hipFree(D->d_label); hipFree(D->d_csize); hipFree(D->d_roots); hipFree(D->d_perim); hipFree(D->d_hkeys);
  hipFree(D->d_hcnt); hipFree(D->d_hoff); hipFree(D->d_stage); hipFree(D->d_bhdr); hipFree(D->d_btab); hipFree(D->d_long); hipFree(D->d_pts); hipFree(D->d_clusters);
  hipFree(D->d_work); hipFree(D->d_work2); hipFree(D->d_workctl); hipFree(D->d_keys_scr); hipFree(D->d_quads);
  for (auto& c : D->cls) { hipFree(c.d_lf); hipFree(c.d_errs); }
  hipFree(D->d_fqprof);
  hipFree(D->d_cands); hipFree(D->d_dets); hipFree(D->d_order); hipFree(D->d_frames);   // (d_counters lives behind d_workctl)
  for (int i = 0; i < AT_MAX_FAMILIES; i++) hipFree(D->d_codes[i]);
  if (D->h_frames) hipHostFree(D->h_frames);
  if (D->h_counters) hipHostFree(D->h_counters);
  if (D->h_out) hipHostFree(D->h_out);
  for (auto& e : D->ev) if (e) hipEventDestroy(e);
  if (D->own_stream) hipStreamDestroy(D->own_stream);
  for (auto& a : D->aux_stream) if (a) hipStreamDestroy(a);
  if (D->ev_fork) hipEventDestroy(D->ev_fork);
  for (auto& e : D->ev_join) if (e) hipEventDestroy(e);
}

// The pair table is empty between submissions: k_cluster_select, its last reader, empties every slot it finds used (a few
// per cent of the table), so a submission starts without the two table-sized fills it used to open with (201 MB per 256
// frames; on a one-frame call two of five 4-5 us fill launches ahead of the first kernel).  The whole table is only
// written here: after (re)allocation, and after a submission that did not run to its end (tables_dirty).
static int clear_hash_tables(amdAprilTagsDetector_st* D) {
  const size_t B = D->cfg.max_batch;
  // (this runs at creation and when a capacity changes, never in a steady-state call.  On the handle's OWN stream, and a wait for
  // that stream only: a legacy-stream fill would fail -- here, in this thread -- whenever another host thread's handle on the
  // same device is capturing its launch graph at that moment, and a device-wide wait would stall that thread's streams.  Round 4
  // had tried this and taken it out again over a crash "inside the regrowth test" that round 6 traced to hipGraphExecDestroy.)
  if (hipMemsetAsync(D->d_hkeys, 0xFF, B * (size_t)D->P.hcap * 8, D->own_stream) != hipSuccess) return AMDAT_HIP_ERROR;
  if (hipMemsetAsync(D->d_hcnt, 0, B * (size_t)D->P.hcap * 4, D->own_stream) != hipSuccess) return AMDAT_HIP_ERROR;
  if (hipStreamSynchronize(D->own_stream) != hipSuccess) return AMDAT_HIP_ERROR;
  D->tables_dirty = false;
  return AMDAT_SUCCESS;
}

// The component-pair table of P.hcap slots per frame.
static int alloc_hash_buffers(amdAprilTagsDetector_st* D) {
  DetParams& P = D->P;
  const size_t B = D->cfg.max_batch;
  { uint32_t lg = 0; while ((1u << lg) < P.hcap) lg++; P.hshift = 64 - lg; }
  void** bufs[3] = {(void**)&D->d_hkeys, (void**)&D->d_hcnt, (void**)&D->d_hoff};
  const size_t bytes[3] = {B * (size_t)P.hcap * 8, B * (size_t)P.hcap * 4, B * (size_t)P.hcap * 4};
  for (int i = 0; i < 3; i++)
    if (*bufs[i]) { hipFree(*bufs[i]); *bufs[i] = nullptr; D->device_bytes -= D->hash_buffer_bytes[i]; D->hash_buffer_bytes[i] = 0; }
  for (int i = 0; i < 3; i++) {
    if (hipMalloc(bufs[i], bytes[i]) != hipSuccess) return AMDAT_OUT_OF_MEMORY;
    D->hash_buffer_bytes[i] = bytes[i];
    D->device_bytes += bytes[i];
  }
  return clear_hash_tables(D);
}

// Buffers whose size follows the point capacity P.pcap: staging words, long staging records, points and the quad fit's work lists (their
// capacities are bounded by points / smallest cluster of the class).  Called at creation and again when the capacity grows.
static int alloc_point_buffers(amdAprilTagsDetector_st* D) {
  DetParams& P = D->P;
  const size_t B = D->cfg.max_batch;
  uint64_t off = 0;
  for (int k = 0; k < FQ_NCLS; k++) {
    const FqClass& c = D->cls[k];
    D->work_layout.lo[k] = c.lo < 23 ? 23 : c.lo;
    D->work_layout.hi[k] = c.hi;
    // (the one-wave class of k_fit_quads also takes the k_fit_small classes' clusters on small submissions: sized from 24 points)
    const uint32_t per_frame = P.pcap / (uint32_t)((k == FQ_C0 ? 23 : D->work_layout.lo[k]) + 1) + 1;
    const uint64_t cap = c.hi <= c.lo ? 16 : (uint64_t)B * (per_frame < P.ccap ? per_frame : P.ccap);   // (an empty class keeps a token range)
    if (cap > 0x7FFFFFFFull || off + cap > 0xFFFFFFFFull) return AMDAT_BATCH_TOO_LARGE;   // offsets and cursors are 32-bit
    D->work_layout.off[k] = (uint32_t)off;
    D->work_layout.cap[k] = (uint32_t)cap;
    off += cap;
  }
  D->work_layout_small = D->work_layout;
  for (int k = 0; k < FQ_C0; k++) { D->work_layout_small.lo[k] = 23; D->work_layout_small.hi[k] = 0; }
  D->work_layout_small.lo[FQ_C0] = 23;
  // Long staging records (kernels_cluster.h) only occur where a 64 x 16 tile has more than 2048 emissions -- above two per pixel --
  // or more than 255 component pairs; an eighth of the point capacity is room for them on ordinary content.  An overflow reports
  // like a point overflow (0x1); the host tells the two apart by the counters and grows the list by itself -- a quarter, half, all
  // of the point capacity (end_batch): two-level noise near the percolation threshold puts more than a quarter of its points there
  // (a fuzz case of round 6: such a frame used to keep its overflow flag at the largest point capacity).  Tools builds that shrink
  // the tile's list start with the whole capacity.
#ifndef AMDAT_LCAP_DIV
#define AMDAT_LCAP_DIV 8
#endif
  if (D->lcap_div == 0) D->lcap_div = AMDAT_LCAP_DIV;
  P.lcap = P.pcap / D->lcap_div > 4096u ? P.pcap / D->lcap_div : 4096u;
  void** bufs[5] = {(void**)&D->d_stage, (void**)&D->d_long, (void**)&D->d_pts, (void**)&D->d_work, (void**)&D->d_work2};
  const size_t bytes[5] = {B * (size_t)P.pcap * 4, B * (size_t)P.lcap * 16, B * (size_t)P.pcap * 4, (size_t)off * 4,
                           ((size_t)off - D->work_layout.off[D->prefilter_class]) * 4};
  for (int i = 0; i < 5; i++) {
    if (*bufs[i]) { hipFree(*bufs[i]); *bufs[i] = nullptr; D->device_bytes -= D->point_buffer_bytes[i]; D->point_buffer_bytes[i] = 0; }
  }
  for (int i = 0; i < 5; i++) {
    const size_t nb = bytes[i] ? bytes[i] : 16;
    if (hipMalloc(bufs[i], nb) != hipSuccess) return AMDAT_OUT_OF_MEMORY;
    D->point_buffer_bytes[i] = nb;
    D->device_bytes += nb;
  }
  return AMDAT_SUCCESS;
}

int amdCreateAprilTagsDetectorEx(amdAprilTagsHandle* handle, const amdAprilTagsConfig_t* cfg_in) {
  std::call_once(g_fam_once, init_families);
  if (!handle || !cfg_in) return AMDAT_INVALID_ARGUMENT;
  *handle = nullptr;
  // the caller's struct may be an older, shorter one (include/apriltag_amd.h: struct_size): only its own bytes are read, the
  // fields beyond them keep the defaults
  // layout 1 (round 5, the first with a size field) ended behind corner_convention; nothing shorter was ever published with a size
  constexpr uint32_t kMinConfig = (uint32_t)offsetof(amdAprilTagsConfig_t, no_graph_replay);
  if (cfg_in->struct_size < kMinConfig || cfg_in->struct_size > sizeof(amdAprilTagsConfig_t)) return AMDAT_INVALID_ARGUMENT;
  amdAprilTagsConfig_t cfg;
  amdAprilTagsDefaultConfig(&cfg, 0, 0);
  memcpy(&cfg, cfg_in, cfg_in->struct_size);
  cfg.struct_size = (uint32_t)sizeof(cfg);
  if (cfg.width == 0 || cfg.height == 0 || cfg.max_batch == 0 || cfg.decimate == 0) return AMDAT_INVALID_ARGUMENT;
  if (cfg.max_batch > 65535) return AMDAT_BATCH_TOO_LARGE;   // a work item carries the batch slot in at most 16 bits
  if (cfg.tile_size != 4 && cfg.tile_size != 8) return AMDAT_UNSUPPORTED;   // (the reference's default and twice it)
  if (cfg.decimate > 4) return AMDAT_UNSUPPORTED;     // the threshold loader is instantiated for 1..4
  if (cfg.max_hamming > 3) return AMDAT_INVALID_ARGUMENT;  // AprilRobotics' own limit for the code search
  if (cfg.corner_convention > AMDAT_CORNERS_ROTATED_180) return AMDAT_INVALID_ARGUMENT;
  if (cfg.num_families < 1 || cfg.num_families > AT_MAX_FAMILIES) return AMDAT_INVALID_ARGUMENT;
  // the family tables are copied under the registry's lock: a concurrent amdAprilTagsRegisterFamily[Ex] cannot swap a table
  // out from under the copy
  // (only for the copy: creation itself -- allocations, uploads, a device-wide wait -- runs without the registry's lock, so hosts
  // that create their per-GPU handles in parallel, or register families meanwhile, are not serialised behind it)
  struct FamilyCopy { FamilyHost layout; std::vector<uint64_t> codes; };
  std::vector<FamilyCopy> fams(cfg.num_families);
  {
    std::lock_guard<std::mutex> fam_lock(g_fam_mutex);
    for (uint32_t i = 0; i < cfg.num_families; i++) {
      if ((int)cfg.families[i] < 0 || cfg.families[i] >= AMDAT_ENUM_SIZE || !g_families[cfg.families[i]].codes)
        return AMDAT_UNSUPPORTED;
      const FamilyHost& g = g_families[cfg.families[i]];
      fams[i].codes.assign(g.codes, g.codes + g.ncodes);
      fams[i].layout = g;
      fams[i].layout.owned.clear(); fams[i].layout.codes = nullptr; fams[i].layout.name = nullptr;
    }
  }
  const int W = 1 + ((int)cfg.width - 1) / (int)cfg.decimate, H = 1 + ((int)cfg.height - 1) / (int)cfg.decimate;
  if (W / (int)cfg.tile_size < 1 || H / (int)cfg.tile_size < 1 || 2 * W + 1 >= (1 << 14) || 2 * H + 1 >= (1 << 14)) return AMDAT_UNSUPPORTED;

  auto* D = new (std::nothrow) amdAprilTagsDetector_st();
  if (!D) return AMDAT_OUT_OF_MEMORY;
  D->cfg = cfg;
  int caller_device = -1;
  if (hipGetDevice(&caller_device) != hipSuccess) { delete D; return AMDAT_HIP_ERROR; }
  D->device = cfg.device >= 0 ? cfg.device : caller_device;
  DeviceGuard guard(D->device);
  if (!guard.ok) { delete D; return AMDAT_HIP_ERROR; }
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, D->device) == hipSuccess && prop.multiProcessorCount > 0) D->num_cus = prop.multiProcessorCount;
  }

  // (first: every creation-time fill and copy below runs on it -- no legacy-stream call in this library, see clear_hash_tables)
  if (hipStreamCreateWithFlags(&D->own_stream, hipStreamNonBlocking) != hipSuccess) { delete D; return AMDAT_HIP_ERROR; }
  DetParams& P = D->P;
  memset(&P, 0, sizeof(P));
  P.W0 = (int)cfg.width; P.H0 = (int)cfg.height; P.W = W; P.H = H;
  P.WS = (W + 15) & ~15;
  P.decimate = (int)cfg.decimate;
  P.tile = (int)cfg.tile_size;
  P.tw = W / P.tile; P.th = H / P.tile;
  P.min_white_black_diff = 5;
  P.min_component_size = 25;
  P.min_cluster_points = 24;
  P.max_cluster_points = 3 * (2 * W + 2 * H);
  P.max_nmaxima = 10;
  // exactness bounds of the two-double moment sums (kernels_quad.h, split_term): W * x * x < 2^31, < 2^15 points
  P.split_moments = (W <= 2048 && H <= 2048 && P.max_cluster_points < 32768) ? 1 : 0;
  P.refine_edges = cfg.refine_edges ? 1 : 0;
  P.max_hamming = (int)cfg.max_hamming;
  P.nfam = (int)cfg.num_families;
  P.cos_critical_rad = (double)(float)0x1.f838b8c811c17p-1;   // cos(10 deg) as upstream's FLOAT parameter field holds it
  P.max_line_fit_mse = 10.0;
  P.decode_sharpening = (double)cfg.decode_sharpening;
  P.tag_size = (double)cfg.tag_size;
  int min_tag_width = 1000000;
  for (int i = 0; i < P.nfam; i++) {
    const FamilyHost& f = fams[i].layout;
    P.fam[i].d = f.d; P.fam[i].nbits = f.nbits; P.fam[i].width_at_border = f.width_at_border; P.fam[i].total_width = f.total_width;
    P.fam[i].reversed_border = f.reversed_border; P.fam[i].ncodes = f.ncodes;
    memcpy(P.fam[i].bit_x, f.bit_x, 64); memcpy(P.fam[i].bit_y, f.bit_y, 64); memcpy(P.fam[i].rot_src, f.rot_src, 64);
    if ((int)P.fam[i].width_at_border < min_tag_width) min_tag_width = (int)P.fam[i].width_at_border;
    if (f.reversed_border) P.reversed_border |= 1; else P.normal_border |= 1;
  }
  min_tag_width = (int)((float)min_tag_width / (float)P.decimate);
  if (min_tag_width < 3) min_tag_width = 3;
  P.min_tag_width = min_tag_width;
  const uint32_t npx = (uint32_t)W * (uint32_t)H;
  // Boundary points per frame: 2 per pixel covers every content the fuzzer produces (thin diagonal lines come
  // closest); frames that binarise completely (noise on every 4x4 tile) measure ~0.85 per pixel.
  // Default: 1 per pixel, and the handle GROWS the point buffers (up to the hard 2 per pixel) and repeats the submission
  // when a frame reports AMDAT_FLAG_POINTS_OVERFLOW -- results never depend on the capacity, memory follows the content
  // (15 GB instead of 27 GB for the 256-frame 1080p handle).  An explicit max_points is taken as given and never grown.
  D->grow_points = cfg.max_points == 0;
  D->pcap_hard = 2u * npx;
  P.pcap = cfg.max_points ? cfg.max_points : npx;
  // Component-pair table: one slot per N/8 pixels is what no content overflowed; a sigma-2 1080p frame has ~4 000 pairs, so
  // the table starts at N/32 slots (1 MB instead of 4 MB per frame to clear, probe and scan) and, like the point buffers,
  // doubles when a frame reports AMDAT_FLAG_HASH_OVERFLOW or fills beyond a quarter (an explicit hash_slots is never grown).
  D->grow_hash = cfg.hash_slots == 0;
  D->hcap_hard = cfg.hash_slots ? next_pow2(cfg.hash_slots) : next_pow2(npx / 8 > 4096 ? npx / 8 : 4096);
  if (D->hcap_hard < 256) D->hcap_hard = 256;
  P.hcap = cfg.hash_slots ? D->hcap_hard : next_pow2(npx / 32 > 4096 ? npx / 32 : 4096);
  if (P.hcap > D->hcap_hard) P.hcap = D->hcap_hard;
  // A work item of the quad fit is one word: (frame << wshift) | cluster index.  The frame takes the bits the handle's frame count
  // needs, the index the rest (at most 24): 256 frames per submission leave room for 2^24 clusters per frame, 65 536 frames for
  // 65 536.  A frame has at most one cluster per used slot of the pair table, so min(2^wshift, hcap_hard) bounds the list; it starts
  // at 65 536 (a sigma-2 1080p frame has 4 000 clusters) and, like the point buffers, GROWS when a frame reports
  // AMDAT_FLAG_CLUSTERS_OVERFLOW -- an eight-megapixel checkerboard of six-pixel cells has 116 000 (end_batch).  An explicit
  // max_clusters is taken as given (clamped to the bound) and never grown.
  {
    uint32_t fbits = 0;
    while ((1ull << fbits) < (uint64_t)cfg.max_batch) fbits++;
    // (max_batch <= 65 535 was checked above: fbits <= 16)
    P.wshift = 32u - fbits > 24u ? 24u : 32u - fbits;
  }
#ifndef AMDAT_CCAP0
#define AMDAT_CCAP0 65536u   // (tools builds start lower, so that ordinary content exercises the growth)
#endif
  D->ccap_hard = D->hcap_hard < (1u << P.wshift) ? D->hcap_hard : (1u << P.wshift);
  D->grow_clusters = cfg.max_clusters == 0;
  P.ccap = cfg.max_clusters ? cfg.max_clusters : (D->ccap_hard < AMDAT_CCAP0 ? D->ccap_hard : AMDAT_CCAP0);
  if (P.ccap > D->ccap_hard) P.ccap = D->ccap_hard;
  P.qcap = cfg.max_quads ? cfg.max_quads : (P.ccap < 16384 ? P.ccap : 16384);
  D->grow_quads = cfg.max_quads == 0;   // (a two-megapixel checkerboard of two-pixel cells has 29 000 quads: the list doubles and the
                                        // submission is repeated, end_batch; an explicit max_quads reports AMDAT_FLAG_QUADS_OVERFLOW)
  P.dcap = cfg.max_detections ? cfg.max_detections : 1024;
  if (P.dcap > 65535) P.dcap = 65535;

  const size_t B = cfg.max_batch;
  // ---- size classes of the quad fit ------------------------------------------------------------------
  // one wave per small cluster, bigger workgroups and LDS key arrays above; persistent grids sized to the
  // chip (CUs x workgroups that fit one CU) but not beyond what a submission of B frames can feed.
  // The two large classes need most of a CU's LDS for their key arrays, so they cannot share a CU with the other
  // classes' persistent workgroups: they run first and alone, at the 128-register budget of the other classes --
  // clusters above 8192 points in 1024-thread workgroups (16 waves, one workgroup per CU), then 4096..8192 points in
  // 512-thread workgroups, two per CU.  (With 512 threads at twice the registers and no room for a neighbour the
  // largest clusters used to run at a quarter of the chip's occupancy, mostly at the end of the stage: 17 ms of wall
  // time for 4 % of the stage's instructions.)
  {
    const unsigned cus = (unsigned)D->num_cus;
    auto minu = [](unsigned a, unsigned b) { return a < b ? a : b; };
    FqClass* c = D->cls;
    // Class boundaries: the one-wave class has no workgroup barriers at all and runs closest to the VALU issue rate
    // (82 % against 53-57 % for the 128- and 256-thread classes), so it takes clusters up to 768 points, the most its
    // 16 workgroups per CU can hold in LDS (9 KB each); measured 16.1 ms (256 / 1024) -> 15.5 (512 / 1024) -> 15.35
    // (768 / 2048); 896 and 1024 are slower again (fewer resident workgroups).
#ifndef FQ_B01
#define FQ_B01 768
#define FQ_B12 2048
#endif
    // Clusters per pop: one atomic per cluster saturates the list cursor (one-wave class: 18.7 ms), chunks of 16 / 8 / 2
    // leave workgroups with up to 16 clusters of work while others have drained the list (15.4 ms); 4 / 2 / 1: 15.0.
#ifndef FQ_POP0
#define FQ_POP0 4
#define FQ_POP1 2
#define FQ_POP2 1
#endif
    // The two smallest classes run k_fit_small (kernels_quad_small.h): clusters up to 128 (and 256) points, whose keys and
    // sweep state fit a wave's registers and whose cumulative moments fit LDS.  They exist on the two-double path only
    // (working images up to 2048 x 2048); otherwise their ranges are empty and the one-wave class starts at 0.
#ifndef FS_B0
#define FS_B0 128
#endif
#ifndef FS_B1
#define FS_B1 FS_B0   // (= FS_B0: no K = 4 class)
#endif
#ifndef FS_K4_GROWS
#define FS_K4_GROWS true   // the K = 4 class's moments: true -- global scratch slot; false -- LDS (14 KB per workgroup)
#endif
    const int sb0 = P.split_moments ? FS_B0 : 0, sb1 = P.split_moments ? FS_B1 : 0;
    c[0] = {64, 0, 0, sb0, minu((unsigned)FS_GRID_K2 * cus, 4096u * (unsigned)B), sb0, FQ_POP0, 2};
    c[1] = {64, 0, sb0, sb1, minu((unsigned)FS_GRID_K4 * cus, 4096u * (unsigned)B), sb1 > 256 ? sb1 : 256, FQ_POP0, 4};   // (slot: 256 rows of moments)
    FqClass* const q = c + FQ_C0;   // the classes of k_fit_quads
    q[0] = {64, FQ_B01, sb1, FQ_B01, minu((unsigned)FQ_GRID_64 * cus, 4096u * (unsigned)B), FQ_B01, FQ_POP0};
    q[1] = {128, FQ_B12, FQ_B01, FQ_B12, minu((unsigned)FQ_GRID_128 * cus, 1024u * (unsigned)B), FQ_B12, FQ_POP1};
    q[2] = {256, 4096, FQ_B12, 4096, minu(4u * cus, 256u * (unsigned)B), 4096, FQ_POP2};
    q[3] = {512, 8192, 4096, 8192, minu(2u * cus, 64u * (unsigned)B), 8192, 1};
    // (a "latency layout" for small-batch handles -- about three times the threads per cluster: 64 up to 256 points, 128 up
    // to 768, 256 up to 2048, 512 up to 8192 -- measured slower on one-frame submissions, 0.36 against 0.28 ms for the
    // stage: the larger workgroups' barriers cost more than the shorter per-lane runs save)
    D->prefilter_class = FQ_C0 + 2;
    q[4] = {FQ_NT_BIG, 16384, 8192, 0x7FFFFFFF, minu(cus, 16u * (unsigned)B), P.max_cluster_points, 1};
    static_assert(FQ_C0 + 5 == FQ_NCLS, "class table");
    if (P.max_cluster_points > 16384 && P.max_cluster_points <= 18432) q[4].sort_cap = (P.max_cluster_points + 63) & ~63;
    if (q[4].slot_cap < 8193) q[4].slot_cap = 8193;
  }

  bool ok = true;
  auto alloc = [&](void** p, size_t bytes) {
    if (!bytes) bytes = 16;
    if (ok && hipMalloc(p, bytes) != hipSuccess) ok = false;
    if (ok) D->device_bytes += bytes;
  };
  if (P.decimate > 1) alloc((void**)&D->d_gray, B * (size_t)H * P.WS);
  alloc((void**)&D->d_thr, B * (size_t)H * P.WS);
  if (P.tile != 4) { alloc((void**)&D->d_tmin, B * (size_t)P.tw * P.th); alloc((void**)&D->d_tmax, B * (size_t)P.tw * P.th); }
  alloc((void**)&D->d_label, B * (size_t)npx * 4);
  alloc((void**)&D->d_csize, B * (size_t)npx * 4);
  // tile-local roots that go to the list touch their 64 x 64 tile's perimeter, and components are disjoint: at most
  // 252 (perimeter pixels) per tile
  P.rcap = (uint32_t)(((W + CC_T - 1) / CC_T) * ((H + CC_T - 1) / CC_T)) * (4u * CC_T - 4u);
  alloc((void**)&D->d_roots, B * (size_t)P.rcap * 4);
  alloc((void**)&D->d_perim, B * (size_t)cc_perim_layout(W, H).words * 4);
  if (ok) ok = alloc_hash_buffers(D) == AMDAT_SUCCESS;   // (before the point buffers: it decides the staging format)
  {   // per block (64 x 16 tile) of k_points: header and component-pair table for k_scatter
    const size_t tiles = (size_t)((W + PT_TW - 1) / PT_TW) * (size_t)((H + PT_TH - 1) / PT_TH);
    alloc((void**)&D->d_bhdr, B * tiles * sizeof(uint2));
    alloc((void**)&D->d_btab, B * tiles * PT_TB * sizeof(uint2));
  }
  alloc((void**)&D->d_clusters, B * (size_t)P.ccap * sizeof(ClusterRec));
  D->clusters_bytes = B * (size_t)P.ccap * sizeof(ClusterRec);
  if (ok) { const int rc = alloc_point_buffers(D); if (rc == AMDAT_BATCH_TOO_LARGE) { free_all(D); delete D; return rc; } ok = rc == AMDAT_SUCCESS; }
  for (int k = 0; k < FQ_NCLS; k++) {
    FqClass& c = D->cls[k];
    // (the one-wave class of k_fit_quads takes every cluster from 24 points on when a submission runs the latency set: its lower
    // bound for "can this class ever see a cluster" is 23, whatever the k_fit_small classes below it would take on the other set)
    const int lo_eff = k == FQ_C0 ? 23 : c.lo;
    if (P.max_cluster_points <= lo_eff || c.small_k == 2 || c.hi <= c.lo) continue;   // (k_fit_small<2> keeps its moments in LDS)
    alloc((void**)&c.d_lf, (size_t)c.grid * c.slot_cap * 48);
    // smoothed errors stay in registers up to FQ_SMOOTH_REGS_OF(threads) points per thread; larger clusters need a second array
    if (!c.small_k && (c.slot_cap > FQ_SMOOTH_REGS_OF(c.nt) * c.nt || c.slot_cap > c.sort_cap)) alloc((void**)&c.d_errs, (size_t)c.grid * c.slot_cap * 16);
  }
  if (D->cls[FQ_NCLS - 1].slot_cap > D->cls[FQ_NCLS - 1].sort_cap) {   // clusters beyond the LDS key array exist
    const FqClass& c = D->cls[FQ_NCLS - 1];
    alloc((void**)&D->d_keys_scr, (size_t)c.grid * c.slot_cap * 8);
  }
  alloc((void**)&D->d_quads, B * (size_t)P.qcap * sizeof(QuadRec));
  D->quads_bytes = B * (size_t)P.qcap * sizeof(QuadRec);
  // every kept cluster can become a candidate (ccap); the list starts at the quad capacity and grows when a frame fills it
  P.cand_cap = P.qcap;
  alloc((void**)&D->d_cands, B * (size_t)P.cand_cap * sizeof(FitCand));
  D->cands_bytes = B * (size_t)P.cand_cap * sizeof(FitCand);
  alloc((void**)&D->d_dets, B * (size_t)P.dcap * sizeof(DetRec));
  alloc((void**)&D->d_order, B * (size_t)P.dcap * 2);
  // work-list control words and frame counters share one allocation, cleared by ONE fill per submission
  alloc((void**)&D->d_workctl, 32 * 4 + B * sizeof(FrameCounters));
  if (ok) D->d_counters = reinterpret_cast<FrameCounters*>(D->d_workctl + 32);
  alloc((void**)&D->d_frames, B * sizeof(FrameDesc));
  alloc((void**)&D->d_fqprof, (64 + 8) * 8);
  D->d_ptprof = D->d_fqprof + 64;   // k_points' phase counters follow the quad fit's
  for (int i = 0; ok && i < P.nfam; i++) {
    const std::vector<uint64_t>& codes = fams[i].codes;
    alloc((void**)&D->d_codes[i], codes.size() * 8);
    if (ok && (hipMemcpyAsync(D->d_codes[i], codes.data(), codes.size() * 8, hipMemcpyHostToDevice, D->own_stream) != hipSuccess ||
               hipStreamSynchronize(D->own_stream) != hipSuccess)) ok = false;   // (`fams` is pageable: waited for before it goes)
    P.fam[i].codes = D->d_codes[i];
  }
  // pinned blocks the kernels read (descriptors) and write (records, counters + stamp) over the bus: COHERENT (fine-grained) host
  // memory, stated rather than left to the runtime's default -- the host reads them right after a stream wait, not after a
  // device-wide one
  const unsigned host_flags = hipHostMallocCoherent | hipHostMallocMapped;
  if (ok && hipHostMalloc((void**)&D->h_frames, B * sizeof(FrameDesc), host_flags) != hipSuccess) ok = false;
  if (ok && hipHostMalloc((void**)&D->h_counters, B * sizeof(FrameCounters), host_flags) != hipSuccess) ok = false;
  if (ok && hipHostMalloc((void**)&D->h_out, B * (size_t)P.dcap * sizeof(DetRec), host_flags) != hipSuccess) ok = false;
  if (ok) memset(D->h_counters, 0, B * sizeof(FrameCounters));
  for (auto& e : D->ev) if (ok && hipEventCreate(&e) != hipSuccess) ok = false;
  // Side streams.  The persistent grids of the fit's classes each hold the whole chip's wave slots, so which class's workgroups
  // are placed first decides who runs beside whom.  A handle sized for throughput (more than eight frames per submission) gets
  // PRIORITISED side streams -- greatest, default, least: the class with the longest chains goes on the first and is placed
  // first, the one-wave and small-cluster kernels fill what it leaves (AMDAT_AUX_PRIO: measured below) -- and never captures
  // launch graphs: replaying a graph whose branches were captured on prioritised streams costs 0.25 ms per launch on this runtime
  // (one frame: 0.39 -> 0.65 ms).  A handle of up to eight frames keeps plain side streams and graph replay.  (Both sets on one
  // handle -- seven streams -- slowed every stage: 17.6 -> 18.1 ms per 256 frames; the runtime multiplexes its streams onto a
  // few hardware queues.)
#ifndef AMDAT_AUX_PRIO
#define AMDAT_AUX_PRIO 1
#endif
  D->aux_prioritised = AMDAT_AUX_PRIO && cfg.max_batch > 8 && !cfg.no_stream_priorities;
  {
    int lo = 0, hi = 0;   // (numerically hi <= lo: hi is the greatest priority)
    if (D->aux_prioritised && hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { D->aux_prioritised = false; (void)hipGetLastError(); }
    // (What the runtime does with the streams' hardware queues is sensitive to the ORDER in which plain and prioritised streams
    // come into being in a process -- measured on ROCm 7.0 / 7.2, bench.py's step and one- / eight-frame calls of a small handle,
    // tools/stream_order.py, tools/bench_variant.py:
    //   throughput-sized handle first, small handle after it      17.30 ms per 256 frames | 0.49 ms one frame, 1.32 ms eight
    //   small handle first (its four plain streams alive)         17.5                    | 0.38, 1.05
    //   four plain streams created and destroyed first            18.0 (every stage)      | 0.38, 1.05
    //   ... created and destroyed after the prioritised ones      17.5                    | 0.47, 1.34
    //   no priorities at all (cfg.no_stream_priorities)           17.7                    | 0.38, 1.05
    // rocprofv3's kernel trace shows the graph branches of a small handle created after a prioritised one on that handle's
    // hardware queues.  The library does not try to steer this: the first row is what a process with one throughput-sized handle
    // gets, a node's process -- one small handle -- never meets a prioritised stream, and a process that mixes both creates the
    // small handles first or sets no_stream_priorities: include/apriltag_amd.h, INTEGRATION.md.)
    auto prime_plain_queues = [&]() {
      hipStream_t plain[4] = {};
      for (auto& q : plain) if (hipStreamCreateWithFlags(&q, hipStreamNonBlocking) != hipSuccess) { q = nullptr; (void)hipGetLastError(); }
      for (auto& q : plain) if (q) hipStreamDestroy(q);
    };
#ifndef AMDAT_PRIME_PLAIN_QUEUES
#define AMDAT_PRIME_PLAIN_QUEUES 0   // 0: never, 1: before the prioritised streams are created, 2: after (measurement builds)
#endif
    if (D->aux_prioritised && AMDAT_PRIME_PLAIN_QUEUES == 1) prime_plain_queues();
    for (int k = 0; k < FQ_NAUX; k++) {
      const int pr = k == 0 ? hi : (k == 1 ? (lo + hi) / 2 : lo);
      if (ok && (D->aux_prioritised ? hipStreamCreateWithPriority(&D->aux_stream[k], hipStreamNonBlocking, pr)
                                    : hipStreamCreateWithFlags(&D->aux_stream[k], hipStreamNonBlocking)) != hipSuccess) ok = false;
    }
    if (D->aux_prioritised && AMDAT_PRIME_PLAIN_QUEUES == 2) prime_plain_queues();
    if (D->aux_prioritised) D->graph_max_frames = 0;
  }
  if (ok && hipEventCreateWithFlags(&D->ev_fork, hipEventDisableTiming) != hipSuccess) ok = false;
  for (auto& e : D->ev_join) if (ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) ok = false;
  // dynamic LDS beyond 64 KB has to be allowed per kernel (once per device; not allowed while a stream is being captured)
  if (ok) {
      const void* fns[10] = {reinterpret_cast<const void*>(k_fit_quads<512, false>), reinterpret_cast<const void*>(k_fit_quads<512, true>),
                            reinterpret_cast<const void*>(k_fit_quads<FQ_NT_BIG, false>), reinterpret_cast<const void*>(k_fit_quads<256, false>),
                            reinterpret_cast<const void*>(k_fit_quads<128, false>), reinterpret_cast<const void*>(k_fit_quads<64, false>),
                            reinterpret_cast<const void*>(k_fit_quads<FQ_NT_BIG, true>), reinterpret_cast<const void*>(k_fit_quads<256, true>),
                            reinterpret_cast<const void*>(k_fit_quads<128, true>), reinterpret_cast<const void*>(k_fit_quads<64, true>)};
      for (const void* fn : fns) if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 157000) != hipSuccess) ok = false;
  }
  if (ok && D->d_thr) {
    // the padding columns of the working images are read by vector loads; define them once
    if (hipMemsetAsync(D->d_thr, 127, B * (size_t)H * P.WS, D->own_stream) != hipSuccess) ok = false;
    if (ok && D->d_gray && hipMemsetAsync(D->d_gray, 0, B * (size_t)H * P.WS, D->own_stream) != hipSuccess) ok = false;
    if (ok && hipStreamSynchronize(D->own_stream) != hipSuccess) ok = false;
  }
  if (!ok) {
    free_all(D);
    delete D;
    return AMDAT_OUT_OF_MEMORY;
  }
  if (cfg.no_graph_replay) D->graph_max_frames = 0;
  *handle = D;
  return AMDAT_SUCCESS;
}

int amdCreateAprilTagsDetector(amdAprilTagsHandle* handle, uint32_t img_width, uint32_t img_height, uint32_t tile_size,
                               amdAprilTagsFamily tag_family, const amdAprilTagsCameraIntrinsics_t* cam, float tag_dim) {
  if (!cam) return AMDAT_INVALID_ARGUMENT;
  amdAprilTagsConfig_t cfg;
  amdAprilTagsDefaultConfig(&cfg, img_width, img_height);
  cfg.tile_size = tile_size;
  cfg.families[0] = tag_family;
  cfg.intrinsics = *cam;
  cfg.tag_size = tag_dim;
  return amdCreateAprilTagsDetectorEx(handle, &cfg);
}

int amdAprilTagsDestroy(amdAprilTagsHandle handle) {
  if (!handle) return AMDAT_INVALID_ARGUMENT;
  DeviceGuard guard(handle->device);
  if (handle->own_stream) (void)hipStreamSynchronize(handle->own_stream);
  for (auto& a : handle->aux_stream) if (a) (void)hipStreamSynchronize(a);
  (void)hipDeviceSynchronize();   // (a caller's stream may have carried the last submission; fails harmlessly under another thread's capture)
  (void)hipGetLastError();
  const bool had_graphs = !handle->retired_graphs.empty() || [&]() { for (auto& g : handle->graphs) if (g.exec) return true; return false; }();
  free_all(handle);
  // (hipGraphExecDestroy followed, with no device-wide wait in between, by the capture, instantiation and launch of another graph
  // is the sequence that crashes inside PyTorch's bundled HIP 7.0 runtime -- retire_graph below -- and the next graph may be
  // another handle's: one run of the regrowth stress loop in the GPU suite died that way, handle after handle in one process)
  if (had_graphs) { (void)hipDeviceSynchronize(); (void)hipGetLastError(); }
  delete handle;
  return AMDAT_SUCCESS;
}

int amdAprilTagsGetDeviceBytes(amdAprilTagsHandle handle, size_t* bytes) {
  if (!handle || !bytes) return AMDAT_INVALID_ARGUMENT;
  *bytes = handle->device_bytes;
  return AMDAT_SUCCESS;
}

int amdAprilTagsSetProfiling(amdAprilTagsHandle handle, int enable) {
  if (!handle) return AMDAT_INVALID_ARGUMENT;
  handle->profiling = enable != 0;
  handle->fq_counters = enable >= 2;
  return AMDAT_SUCCESS;
}

static void drop_graphs(amdAprilTagsDetector_st* D);   // (defined with the submission code below)

int amdAprilTagsDebugSetSubmissionPath(amdAprilTagsHandle handle, int path) {
  if (!handle || path < AMDAT_PATH_AUTO || path > AMDAT_PATH_THROUGHPUT) return AMDAT_INVALID_ARGUMENT;
  if (handle->inflight.active) return AMDAT_INVALID_ARGUMENT;   // (a regrowth relaunch inside the wait must run the set the submit reported)
  if (path == handle->path_mode) return AMDAT_SUCCESS;
  DeviceGuard guard(handle->device);
  if (!guard.ok) return AMDAT_HIP_ERROR;
  drop_graphs(handle);   // captured under the other path
  handle->path_mode = path;
  return AMDAT_SUCCESS;
}

int amdAprilTagsDebugLateWaits(amdAprilTagsHandle handle) { return handle ? (int)handle->late_waits : -1; }

int amdAprilTagsDebugGraphReplay(amdAprilTagsHandle handle, uint32_t* live_graphs, uint32_t* retired_graphs) {
  if (!handle) return -1;
  uint32_t live = 0;
  for (const auto& g : handle->graphs) live += g.exec ? 1u : 0u;
  if (live_graphs) *live_graphs = live;
  if (retired_graphs) *retired_graphs = (uint32_t)handle->retired_graphs.size();
  return handle->graph_max_frames ? 1 : 0;
}

int amdAprilTagsDebugLastSubmissionPath(amdAprilTagsHandle handle) {
  return handle ? handle->last_path : -1;
}

int amdAprilTagsGetStageMs(amdAprilTagsHandle handle, float* ms) {
  if (!handle || !ms) return AMDAT_INVALID_ARGUMENT;
  memcpy(ms, handle->stage_ms, sizeof(handle->stage_ms));
  return AMDAT_SUCCESS;
}

static int check_images(amdAprilTagsDetector_st* D, uint32_t n, const amdAprilTagsImageInput_t* images, uint32_t fmt = AMDAT_ENC_MONO8) {
  if (n == 0 || !images) return AMDAT_INVALID_ARGUMENT;
  if (fmt > AMDAT_ENC_BGRA8) return AMDAT_UNSUPPORTED;
  if (n > D->cfg.max_batch) return AMDAT_BATCH_TOO_LARGE;
  for (uint32_t i = 0; i < n; i++) {
    if (!images[i].dev_ptr) return AMDAT_INVALID_ARGUMENT;
    if (images[i].width != D->cfg.width || images[i].height != D->cfg.height) return AMDAT_SIZE_MISMATCH;
    if (images[i].pitch < (size_t)images[i].width * enc_channels(fmt)) return AMDAT_INVALID_ARGUMENT;
    if ((uint64_t)images[i].pitch * images[i].height > 0x7FFFFFFFull) return AMDAT_INVALID_ARGUMENT;   // 32-bit pixel offsets on the device
    if (images[i].pitch >= (1u << 24)) return AMDAT_INVALID_ARGUMENT;   // (row offsets are formed with 24-bit multiplies: a 16 MB row is no image)
  }
  return AMDAT_SUCCESS;
}

// A colour submission takes the fused loader of the one-pass threshold kernel where that kernel runs undecimated (tile_size 4,
// decimate 1: the reference's cuAprilTags configuration); otherwise one conversion launch ahead of the mono8 pipeline.
static inline bool colour_fused(const amdAprilTagsDetector_st* D, uint32_t fmt) {
  return fmt != AMDAT_ENC_MONO8 && D->P.decimate == 1 && D->P.tile == 4;
}
// The gray plane the colour frames of a submission become (allocated on the first colour submission; the pointers travel
// through the descriptor block, so captured launch sequences of earlier mono8 submissions stay valid).
static int ensure_colour_plane(amdAprilTagsDetector_st* D, uint32_t fmt) {
  if (fmt == AMDAT_ENC_MONO8) return AMDAT_SUCCESS;
  const size_t B = D->cfg.max_batch;
  if (colour_fused(D, fmt)) {
    if (D->d_gray) return AMDAT_SUCCESS;
    const size_t bytes = B * (size_t)D->P.H * D->P.WS;
    if (hipMalloc((void**)&D->d_gray, bytes) != hipSuccess) { D->d_gray = nullptr; return AMDAT_OUT_OF_MEMORY; }
    D->device_bytes += bytes;
    if (hipMemsetAsync(D->d_gray, 0, bytes, D->own_stream) != hipSuccess || hipStreamSynchronize(D->own_stream) != hipSuccess) return AMDAT_HIP_ERROR;
    return AMDAT_SUCCESS;
  }
  if (D->d_conv) return AMDAT_SUCCESS;
  D->conv_pitch = ((size_t)D->cfg.width + 63) & ~(size_t)63;
  const size_t bytes = B * D->conv_pitch * D->cfg.height;
  if (hipMalloc((void**)&D->d_conv, bytes) != hipSuccess) { D->d_conv = nullptr; return AMDAT_OUT_OF_MEMORY; }
  D->device_bytes += bytes;
  return AMDAT_SUCCESS;
}

static void fill_frames(amdAprilTagsDetector_st* D, uint32_t n, const amdAprilTagsImageInput_t* images,
                        const amdAprilTagsCameraIntrinsics_t* intr, uint32_t fmt = AMDAT_ENC_MONO8) {
  for (uint32_t i = 0; i < n; i++) {
    const amdAprilTagsCameraIntrinsics_t& k = intr ? intr[i] : D->cfg.intrinsics;
    if (fmt == AMDAT_ENC_MONO8) {
      D->h_frames[i].img = images[i].dev_ptr;
      D->h_frames[i].pitch = (uint32_t)images[i].pitch;
      D->h_frames[i].src = nullptr; D->h_frames[i].src_pitch = 0;
    } else {   // every stage behind the threshold pass (or the conversion launch) reads the handle's gray plane
      const bool fused = colour_fused(D, fmt);
      D->h_frames[i].img = fused ? D->d_gray + (size_t)i * D->P.H * D->P.WS : D->d_conv + (size_t)i * D->conv_pitch * D->cfg.height;
      D->h_frames[i].pitch = fused ? (uint32_t)D->P.WS : (uint32_t)D->conv_pitch;
      D->h_frames[i].src = images[i].dev_ptr;
      D->h_frames[i].src_pitch = (uint32_t)images[i].pitch;
    }
    D->h_frames[i].fmt = fmt;
    D->h_frames[i].seq = D->seq;
    D->h_frames[i].fx = (double)k.fx; D->h_frames[i].fy = (double)k.fy;
    D->h_frames[i].cx = (double)k.cx; D->h_frames[i].cy = (double)k.cy;
    D->h_frames[i].skew = (double)(i < D->frame_skew.size() ? D->frame_skew[i] : D->cfg.skew);
  }
}

static void launch_threshold(amdAprilTagsDetector_st* D, const DetParams& P, uint32_t n, hipStream_t s, uint32_t fmt = AMDAT_ENC_MONO8) {
  if (fmt != AMDAT_ENC_MONO8 && !colour_fused(D, fmt)) {   // conversion launch: src -> the full-size mono8 plane the descriptors name
    const dim3 g((P.W0 + 1023) / 1024, (unsigned)P.H0, n);
    switch (fmt) {
      case AMDAT_ENC_RGB8: hipLaunchKernelGGL((k_to_mono8_frames<3, 0, 2>), g, dim3(256), 0, s, D->d_frames, (uint32_t)P.W0, (uint32_t)P.H0); break;
      case AMDAT_ENC_BGR8: hipLaunchKernelGGL((k_to_mono8_frames<3, 2, 0>), g, dim3(256), 0, s, D->d_frames, (uint32_t)P.W0, (uint32_t)P.H0); break;
      case AMDAT_ENC_RGBA8: hipLaunchKernelGGL((k_to_mono8_frames<4, 0, 2>), g, dim3(256), 0, s, D->d_frames, (uint32_t)P.W0, (uint32_t)P.H0); break;
      default: hipLaunchKernelGGL((k_to_mono8_frames<4, 2, 0>), g, dim3(256), 0, s, D->d_frames, (uint32_t)P.W0, (uint32_t)P.H0); break;
    }
    fmt = AMDAT_ENC_MONO8;
  }
  if (P.tile != 4) {   // the two-pass statement (kernels_threshold.h); 4 keeps the one-pass kernel below
    const dim3 g1((unsigned)((P.tw * P.th + 255) / 256), 1, n), g2((unsigned)((P.W + 255) / 256), (unsigned)P.H, n);
#define TH_ANY(DEC)                                                                                                      \
    hipLaunchKernelGGL(k_tile_minmax<DEC>, g1, dim3(256), 0, s, D->d_frames, D->d_tmin, D->d_tmax, P.tile, P);           \
    hipLaunchKernelGGL(k_threshold_any_tile<DEC>, g2, dim3(256), 0, s, D->d_frames, D->d_gray, D->d_thr, D->d_tmin, D->d_tmax, P.tile, P);
    switch (P.decimate) {
      case 1: TH_ANY(1) break;
      case 2: TH_ANY(2) break;
      case 3: TH_ANY(3) break;
      default: TH_ANY(4) break;
    }
#undef TH_ANY
    return;
  }
  const int gx = ((P.W + 3) / 4 + TH_BTX - 1) / TH_BTX, gy = ((P.H + 3) / 4 + TH_BTY - 1) / TH_BTY;
  const unsigned ntiles = (unsigned)gx * gy * n;
  dim3 grid(8u * ((ntiles + 7u) / 8u));
  const bool leftover = (P.W % 4) || (P.H % 4);
  const int nleft = (P.W - P.tw * 4) * (P.th * 4) + (P.H - P.th * 4) * P.W;
  dim3 lgrid((unsigned)((nleft + 255) / 256), 1, n);
#define TH_LAUNCH(DEC)                                                                                                   \
  hipLaunchKernelGGL(k_threshold<DEC>, grid, dim3(256), 0, s, D->d_frames, D->d_gray, D->d_thr, gx, gy, (int)n, P);     \
  if (leftover) hipLaunchKernelGGL(k_threshold_leftover<DEC>, lgrid, dim3(256), 0, s, D->d_frames, D->d_gray, D->d_thr, P);
#define TH_LAUNCH_FMT(FMT)                                                                                                 \
  hipLaunchKernelGGL((k_threshold<1, FMT>), grid, dim3(256), 0, s, D->d_frames, D->d_gray, D->d_thr, gx, gy, (int)n, P);   \
  if (leftover) hipLaunchKernelGGL((k_threshold_leftover<1, FMT>), lgrid, dim3(256), 0, s, D->d_frames, D->d_gray, D->d_thr, P);
  if (fmt != AMDAT_ENC_MONO8) {   // (decimate 1: colour_fused)
    switch (fmt) {
      case AMDAT_ENC_RGB8: TH_LAUNCH_FMT(1) break;
      case AMDAT_ENC_BGR8: TH_LAUNCH_FMT(2) break;
      case AMDAT_ENC_RGBA8: TH_LAUNCH_FMT(3) break;
      default: TH_LAUNCH_FMT(4) break;
    }
    return;
  }
#undef TH_LAUNCH_FMT
  switch (P.decimate) {
    case 1: TH_LAUNCH(1) break;
    case 2: TH_LAUNCH(2) break;
    case 3: TH_LAUNCH(3) break;
    default: TH_LAUNCH(4) break;
  }
#undef TH_LAUNCH
}

// Issues the whole stage sequence for batch slots [0, n) on stream s.  mark() is called between stages
// (event timing when profiling).
// first launch of a submission: frame descriptors from the pinned host block to device memory, work-list control words
// and frame counters to zero (one block per frame)
__global__ __launch_bounds__(64) void k_prologue(const uint32_t* __restrict__ host_frames, uint32_t* __restrict__ frames,
                                                 uint32_t* __restrict__ workctl, uint32_t* __restrict__ counters, int fd_words, int fc_words) {
  const int frame = (int)blockIdx.x, t = (int)threadIdx.x;
  for (int i = t; i < fd_words; i += 64) frames[frame * fd_words + i] = host_frames[frame * fd_words + i];
  for (int i = t; i < fc_words; i += 64) counters[frame * fc_words + i] = 0u;
  if (frame == 0 && t < 32) workctl[t] = 0u;
}

// k_reconcile writes every frame's records and counters straight into the pinned host buffers the API call reads: no copy commands
// after the last kernel (about 10 us of a one-frame call; at 256 frames the strided 3.8 MB copy of mostly empty record slots cost
// 0.08 ms per step against the ~0.6 MB of real records the kernel writes: 17.85 -> 17.77 ms).

// A small submission (the node's one-frame calls, up to eight 1080p frames) is about latency, not throughput: every cluster
// is a workgroup's only one, the stage ends with its longest chain, and a launch more costs more than k_fit_small's shorter
// chain per small cluster saves (measured: 0.54 against 0.47 ms per one-frame call).  Such a submission buckets all clusters
// up to the one-wave class's bound into that class (work_layout_small) and launches no k_fit_small.
#ifndef AMDAT_SMALL_PX
#define AMDAT_SMALL_PX (16ull << 20)
#endif
#ifndef AMDAT_SMALL_PX_CC      // k_cc_local<16> below this many working pixels per submission (eight 1080p frames: 0.089 ms with
#define AMDAT_SMALL_PX_CC (8ull << 20)   // four waves per tile against 0.123 with sixteen -- the chip is full by then; four frames: 0.068 either way)
#endif
#ifndef AMDAT_SMALL_PX_PF      // CU-wide prefilter workgroups below this
#define AMDAT_SMALL_PX_PF AMDAT_SMALL_PX
#endif
static inline bool small_submission(const amdAprilTagsDetector_st* D, const DetParams& P, uint32_t n, uint64_t limit = AMDAT_SMALL_PX) {
  if (D->path_mode != AMDAT_PATH_AUTO) return D->path_mode == AMDAT_PATH_LATENCY;
  return (uint64_t)n * (uint64_t)P.W * (uint64_t)P.H < limit;
}
// the frame count the launch heuristics see (chunks of k_cluster_select, clusters per pop): a pinned path takes the values of
// the submissions that path is for, whatever the real count
static inline uint32_t heuristic_frames(const amdAprilTagsDetector_st* D, uint32_t n) {
  if (D->path_mode == AMDAT_PATH_THROUGHPUT) return n < 64u ? 64u : n;
  if (D->path_mode == AMDAT_PATH_LATENCY) return n > 8u ? 8u : n;
  return n;
}

static int issue_pipeline(amdAprilTagsDetector_st* D, uint32_t n, uint32_t ostride, hipStream_t s, uint32_t fmt, const std::function<void()>& mark) {
  DetParams P = D->P;
  P.frame0 = 0;
  launch_threshold(D, P, n, s, fmt);
  mark();
#ifndef AMDAT_CC_WIDE
#define AMDAT_CC_WIDE 1
#endif
  if (AMDAT_CC_WIDE && small_submission(D, P, n, AMDAT_SMALL_PX_CC))   // sixteen waves per tile: a quarter of the rows per lane (latency, not throughput)
    hipLaunchKernelGGL((k_cc_local<16>), dim3((P.W + CC_T - 1) / CC_T, (P.H + CC_T - 1) / CC_T, n), dim3(1024), 0, s, D->d_thr,
                     D->d_label, D->d_csize, D->d_roots, D->d_perim, D->d_counters, P);
  else
    hipLaunchKernelGGL((k_cc_local<4>), dim3((P.W + CC_T - 1) / CC_T, (P.H + CC_T - 1) / CC_T, n), dim3(256), 0, s, D->d_thr,
                     D->d_label, D->d_csize, D->d_roots, D->d_perim, D->d_counters, P);
  mark();
  {
    const int nrows = (P.H - 1) / CC_T, ncols = (P.W - 1) / CC_T;
    const long total = (long)nrows * P.W + (long)ncols * P.H;
    if (total > 0) {
#define BORDER_ARGS dim3((unsigned)((total + 255) / 256) * n), dim3(256), 0, s, D->d_perim, D->d_label, D->d_roots, D->d_counters,   \
                    (uint32_t)((total + 255) / 256), n, P
#ifndef AMDAT_BORDER_PER_WAVE
#define AMDAT_BORDER_PER_WAVE 1
#endif
      if (!AMDAT_BORDER_PER_WAVE || small_submission(D, P, n)) hipLaunchKernelGGL((k_cc_border<false>), BORDER_ARGS);   // (one list append per block)
      else hipLaunchKernelGGL((k_cc_border<true>), BORDER_ARGS);                               // (per wave, no barriers)
#undef BORDER_ARGS
    }
  }
  mark();
  {
    unsigned gr = (unsigned)(((size_t)P.W * P.H / 16 + 255) / 256);
    if (gr < 1) gr = 1;
#ifndef AMDAT_CC_ROOT_GRID
#define AMDAT_CC_ROOT_GRID 1024
#endif
    if (gr > AMDAT_CC_ROOT_GRID) gr = AMDAT_CC_ROOT_GRID;
    hipLaunchKernelGGL(k_cc_sizes, dim3(gr, 1, n), dim3(256), 0, s, D->d_label, D->d_csize, D->d_roots, D->d_counters, P);
    hipLaunchKernelGGL(k_cc_resolve, dim3(gr, 1, n), dim3(256), 0, s, D->d_label, D->d_csize, D->d_roots, D->d_counters, P);
  }
  mark();
  {
    const uint32_t gxt = (uint32_t)((P.W + PT_TW - 1) / PT_TW), gyt = (uint32_t)((P.H + PT_TH - 1) / PT_TH);
    hipLaunchKernelGGL(k_points, dim3(gxt * gyt * n), dim3(256), 0, s, D->d_thr, D->d_label, D->d_hkeys, D->d_hcnt,
                       D->d_stage, D->d_bhdr, D->d_btab, D->d_long, D->d_counters, (D->fq_counters ? D->d_ptprof : nullptr), gxt, gyt, n, P);
  }
  mark();
  {
    const int nchunks = heuristic_frames(D, n) >= 16 ? SEL_CHUNKS : 1;   // (see the kernel)
    hipLaunchKernelGGL(k_cluster_select, dim3((P.hcap + 1024 * nchunks - 1) / (1024 * nchunks), 1, n), dim3(256), 0, s, D->d_hkeys,
                       D->d_hcnt, D->d_hoff, D->d_clusters, D->d_counters, D->d_work, D->d_workctl,
                       small_submission(D, P, n) ? D->work_layout_small : D->work_layout, nchunks, P);
  }
  mark();
  {
    const uint32_t gxt = (uint32_t)((P.W + PT_TW - 1) / PT_TW), gyt = (uint32_t)((P.H + PT_TH - 1) / PT_TH);
    hipLaunchKernelGGL(k_scatter, dim3((gxt * gyt + 3) / 4, 1, n), dim3(256), 0, s, D->d_stage, D->d_bhdr, D->d_btab, D->d_long, D->d_hoff, D->d_pts,
                       D->d_counters, gxt, gyt, P);
  }
  mark();
  {
    // keys | pair-table region
    // skewed key array (later: errors, candidates, pair tables) | group prefixes of the early-exit test (not in the one-wave class)
    auto lds_bytes = [](const FqClass& c) { return FQ_KEY_BYTES(c.sort_cap) + (c.nt > 64 ? (size_t)FQ_TABLE_DOUBLES * 8 : 0); };

    // The classes are independent (they only append to the quad list).  Every class's persistent grid can fill the
    // chip's register file by itself, so whichever workgroups are placed first stay until their list is empty, and the
    // stage is work-conserving whatever the order of the three small classes (17.97 - 18.34 ms over five stream
    // assignments).  The two large classes are different: their workgroups only find room on (half-)empty CUs.  Next
    // to the small classes they were placed last and ran at the end of the stage at a quarter of the chip's
    // occupancy, so a throughput-sized submission runs them first, one after the other on the submission stream, and
    // the small classes start when both are done: 16.9 -> 15.8 ms.  (Queuing the largest class on a stream of its own
    // next to the second one cost 0.9 ms: its queue sat stalled until the scheduler looked at it again.)
    // A small submission is about latency: all classes start together.
    // cheap exits of the large classes (bounding box, border direction, sector test) at full occupancy, ahead of their
    // persistent workgroups, which pop the survivors from the compact lists it writes (d_work2, counts at d_workctl + 16)
    const bool prefilter = FQ_SOUND_EXIT_PREFILTER && P.max_cluster_points > D->cls[D->prefilter_class].lo && D->d_work2;   // (tools_hooks.h: 1)
    uint32_t* const work2 = D->d_work2 ? D->d_work2 - D->work_layout.off[D->prefilter_class] : nullptr;   // (indexed with the common layout)
    // A small submission (the node's one-frame calls) is over when its slowest chain is: there the 256-thread class starts
    // at once beside the small classes (its in-kernel test after the first walk still drops most of its clusters) and
    // only the two largest classes wait for the prefilter -- prefilter, then the survivors' sort, was the longest chain.
    const bool small = small_submission(D, P, n);
    const int pf_first = small ? D->prefilter_class + 1 : D->prefilter_class;
    auto launch_prefilter = [&](hipStream_t sp) {
      if (!prefilter) return;
      if (FQ_SKIP_PREFILTER()) return;   // (tools_hooks.h: always 0 in the product build)
      // small submissions: one cluster per CU-wide workgroup (latency); otherwise one per wave (throughput)
      const bool wide = small_submission(D, P, n, AMDAT_SMALL_PX_PF);
#define PF_ARGS D->d_frames, D->d_gray, D->d_pts, D->d_clusters, D->d_work, D->d_workctl, work2, D->d_workctl + 16, D->d_workctl + 24,   \
                D->work_layout, pf_first, (D->fq_counters ? D->d_fqprof : nullptr), P
      if (wide) {
        unsigned gp = 2u * (unsigned)D->num_cus;
        if (gp > 256u * n) gp = 256u * n;
        hipLaunchKernelGGL(k_fit_prefilter<1024>, dim3(gp), dim3(1024), 0, sp, PF_ARGS);
      } else {
        unsigned gp = 32u * (unsigned)D->num_cus;
        if (gp > 256u * n) gp = 256u * n;
        hipLaunchKernelGGL(k_fit_prefilter<64>, dim3(gp), dim3(64), 0, sp, PF_ARGS);
      }
#undef PF_ARGS
    };
    auto launch_class = [&](int c, hipStream_t sc) -> bool {   // false: the class has no clusters on this handle, nothing was launched
      const FqClass& cl = D->cls[c];
      // (working images so small that no cluster can exceed the k_fit_small classes' bound -- 3 (2 W + 2 H) <= 128 -- still need the
      // one-wave class on the latency set, where it takes everything from 24 points on; found by the fuzzer in round 5: a 9 x 8
      // working image whose only cluster, 34 points, was bucketed into a list no kernel was launched for)
      const int lo_eff = (small && c == FQ_C0) ? 23 : cl.lo;
      if (P.max_cluster_points <= lo_eff || cl.hi <= cl.lo || (!cl.small_k && !cl.d_lf)) return false;
      if (small && cl.small_k) return false;   // (their clusters are in the one-wave class's list: work_layout_small)
      if (FQ_SKIP_CLASS(c)) return false;   // (tools_hooks.h: always 0 in the product build)
#ifndef AMDAT_SMALL_GRID64
#define AMDAT_SMALL_GRID64 8   // workgroups per CU of the one-wave class on the latency set (0: the handle's grid)
#endif
#ifndef AMDAT_SMALL_GRID128
#define AMDAT_SMALL_GRID128 2
#endif
      // (latency set: the one-wave class's sixteen workgroups per CU -- the throughput grid -- leave the 128- and 256-thread classes'
      // workgroups waiting for slots; eight per CU: one 1080p frame 0.395 -> 0.383 ms, four 0.682 -> 0.650, eight 1.028 -> 1.00;
      // six: the same; four: 0.397 / 0.723 / 1.12)
      unsigned gsz = cl.grid;
      if (AMDAT_SMALL_GRID64 && small && c == FQ_C0 && gsz > (unsigned)AMDAT_SMALL_GRID64 * (unsigned)D->num_cus) gsz = (unsigned)AMDAT_SMALL_GRID64 * (unsigned)D->num_cus;
#ifndef AMDAT_SMALL_GRID256
#define AMDAT_SMALL_GRID256 1
#endif
      if (AMDAT_SMALL_GRID256 && small && c == FQ_C0 + 2 && gsz > (unsigned)AMDAT_SMALL_GRID256 * (unsigned)D->num_cus) gsz = (unsigned)AMDAT_SMALL_GRID256 * (unsigned)D->num_cus;
      if (AMDAT_SMALL_GRID128 && small && c == FQ_C0 + 1 && gsz > (unsigned)AMDAT_SMALL_GRID128 * (unsigned)D->num_cus) gsz = (unsigned)AMDAT_SMALL_GRID128 * (unsigned)D->num_cus;
      const dim3 grid(gsz);   // (a submission of n < max_batch frames still gets the handle's persistent grid)
      const size_t lds = lds_bytes(cl);
      const bool big = c == FQ_NCLS - 1;
      // a small submission spreads its clusters over the workgroups one by one (latency); large ones pop in chunks
      const int nh16 = (int)(heuristic_frames(D, n) / 16u);
      const int pop = cl.pop < nh16 ? cl.pop : (nh16 < 1 ? 1 : nh16);
      const bool filtered = prefilter && c >= pf_first;
      if (cl.small_k) {
#define FS_ARGS D->d_frames, D->d_gray, D->d_pts, D->d_clusters, D->d_work + D->work_layout.off[c], D->d_workctl + c,               \
                D->work_layout.cap[c], D->d_workctl + 8 + c, cl.d_lf, D->d_cands, D->d_counters, pop, P
        if (cl.small_k == 2) hipLaunchKernelGGL((k_fit_small<2, false>), grid, dim3(64), FS_LDS_BYTES(2, false), sc, FS_ARGS);
        else hipLaunchKernelGGL((k_fit_small<4, FS_K4_GROWS>), grid, dim3(64), FS_LDS_BYTES(4, FS_K4_GROWS), sc, FS_ARGS);
#undef FS_ARGS
        return true;
      }
#define FQ_ARGS D->d_frames, D->d_gray, D->d_pts, D->d_clusters, (filtered ? work2 : D->d_work) + D->work_layout.off[c],                 \
                D->d_workctl + (filtered ? 16 : 0) + c,                                                                                \
                D->work_layout.cap[c], D->d_workctl + 8 + c, cl.d_lf, (big ? D->d_keys_scr : nullptr), cl.d_errs,                        \
                D->d_cands, D->d_counters, (D->fq_counters ? D->d_fqprof + 8 * (c - FQ_C0) : nullptr), cl.sort_cap, cl.slot_cap, pop, P
#define FQ_LAUNCH(NTV)                                                                                          \
  if (P.split_moments) hipLaunchKernelGGL((k_fit_quads<NTV, true>), grid, dim3(NTV), lds, sc, FQ_ARGS);          \
  else hipLaunchKernelGGL((k_fit_quads<NTV, false>), grid, dim3(NTV), lds, sc, FQ_ARGS);
      if (cl.nt == 64) { FQ_LAUNCH(64) }
      else if (cl.nt == 128) { FQ_LAUNCH(128) }
      else if (cl.nt == 256) { FQ_LAUNCH(256) }
      else if (cl.nt == 512) { FQ_LAUNCH(512) }
      else { FQ_LAUNCH(FQ_NT_BIG) }
#undef FQ_LAUNCH
#undef FQ_ARGS
      return true;
    };
    hipStream_t* aux = D->aux_stream;
    // (throughput-sized: from about 32 1080p working images on; below that the two extra dependent launches cost more
    // than the placement gains -- 64 half-resolution frames measured 3.00 vs 2.88 ms)
#ifndef AMDAT_FQ_ORDER
#define AMDAT_FQ_ORDER 0
#endif
    const bool large_first = AMDAT_FQ_ORDER != 2 && (uint64_t)n * (uint64_t)P.W * (uint64_t)P.H >= (64ull << 20);
    HIP_TRY(hipEventRecord(D->ev_fork, s));
    if (prefilter) {
      // The prefilter runs first and alone (a fraction of a millisecond at full occupancy; started beside the small classes
      // it starved behind their persistent workgroups and finished last).  Then everything starts together: the large
      // classes first on the submission stream -- they find only the prefilter's few survivors in their lists and are gone
      // before the small classes have filled the chip -- and the two small classes on side streams, the 128-thread class
      // (the longest chain) first.
      // A small submission leaves most of the chip empty either way: its classes below the prefilter start at once on the
      // side streams, beside the prefilter.
      if (!small) launch_prefilter(s);
      // The two largest classes need (half) a CU's LDS per workgroup: queued beside the small classes' persistent grids they found no
      // room until those drained -- the 1024-thread class, 15 us of work, sat behind k_fit_small for 1.4 ms whenever it lost that race
      // (profiles/r05_v4_fit_timeline.txt) -- so a throughput-sized submission runs them right behind the prefilter, on the empty
      // chip, and everything else starts when they are through (their lists hold the prefilter's few survivors).
#ifndef AMDAT_BIG_FIRST
#define AMDAT_BIG_FIRST 1
#endif
      const int big_first = (!small && AMDAT_BIG_FIRST) ? pf_first + 1 : FQ_NCLS;
      for (int c = big_first; c < FQ_NCLS; c++) launch_class(c, s);
      HIP_TRY(hipEventRecord(D->ev_fork, s));
      for (int a = 0; a < FQ_NAUX; a++) HIP_TRY(hipStreamWaitEvent(aux[a], D->ev_fork, 0));
      if (small) launch_prefilter(s);
      for (int c = pf_first; c < big_first; c++) launch_class(c, s);   // (nearly all survivors are in the first of them)
      // the longest chains first; classes that launch nothing take no stream.  (The one-wave class on the submission stream itself,
      // so that it starts without the 60 .. 90 us the fork event takes to reach a side stream -- measured with the wall clock inside
      // the kernels -- cost 0.5 ms: its persistent grid then holds the chip before the 128-thread class is placed, which ends up
      // running last and alone.)
      // (holding the shorter-chained classes of a small submission back a few microseconds with a one-wave wait kernel ahead of
      // them, to get on plain streams -- a captured graph -- the placement order stream priorities give, measured nothing: 0.395 vs
      // 0.395 ms for one frame, 1.055 vs 1.065 for eight.  Priorities act on every slot that frees up, not on the first placement.)
      int a = 0;
      for (int c = pf_first - 1; c >= 0; c--)
        if (launch_class(c, aux[a < FQ_NAUX ? a : FQ_NAUX - 1])) a++;   // (a fourth class queues behind the third: the two k_fit_small classes)
    } else {
    if (large_first) {
      launch_class(FQ_C0 + 3, s);
      launch_class(FQ_C0 + 4, s);
      if (AMDAT_FQ_ORDER == 1) launch_class(FQ_C0 + 2, s);
      HIP_TRY(hipEventRecord(D->ev_fork, s));   // both large classes are done
    }
    for (int a = 0; a < FQ_NAUX; a++) HIP_TRY(hipStreamWaitEvent(aux[a], D->ev_fork, 0));
    // (which small class shares the chip with which was measured over seven assignments: 16.0 - 16.9 ms; best when
    // the 256-thread class is the one that ends up running last)
    if (large_first) {
      for (int c = 0, a2 = 0; c < FQ_C0 + (AMDAT_FQ_ORDER == 1 ? 2 : 3); c++) if (launch_class(c, aux[a2 % FQ_NAUX])) a2++;
    } else {   // the longest chains side by side: the largest clusters | 4096..8192 then the one-wave class | the other two
      launch_class(FQ_C0 + 4, s);
      launch_class(FQ_C0 + 3, aux[0]);
      launch_class(FQ_C0 + 2, aux[1]);
      launch_class(FQ_C0 + 1, aux[2]);
      launch_class(FQ_C0 + 0, aux[0]);
      launch_class(1, aux[1]);
      launch_class(0, aux[2]);
    }
    }
    for (int a = 0; a < FQ_NAUX; a++) {
      HIP_TRY(hipEventRecord(D->ev_join[a], aux[a]));
      HIP_TRY(hipStreamWaitEvent(s, D->ev_join[a], 0));
    }
    // corners + area / angle checks of the candidates, one thread each
    hipLaunchKernelGGL(k_quad_finish, dim3(16, n), dim3(256), 0, s, D->d_cands, D->d_quads, D->d_counters, P);
  }
  mark();
  {
    unsigned gq = 2048u / n;
    if (gq < 96u) gq = 96u;   // about one wave per candidate quad of a noisy frame (16 per frame measured 0.36 ms slower)
    if (gq > 256u) gq = 256u;
    hipLaunchKernelGGL(k_decode_wave, dim3(gq, n), dim3(64), 0, s, D->d_frames, D->d_quads, D->d_dets, D->d_counters, P);
  }
  mark();
  {
    hipLaunchKernelGGL(k_reconcile, dim3(n), dim3(64), 0, s, D->d_frames, D->d_dets, D->d_counters, D->d_order, D->h_out, ostride,
                       D->h_counters, P);
  }
  mark();
  return AMDAT_SUCCESS;
}

// Everything one submission enqueues on stream s (and the auxiliary streams forked from it): descriptor upload, clears,
// the stage sequence, result download.  No host synchronisation inside, so the sequence can be stream-captured.
static int enqueue_submission(amdAprilTagsDetector_st* D, uint32_t n, uint32_t ostride, hipStream_t s, uint32_t fmt, const std::function<void()>& mark) {
  mark();
  // descriptor upload + clears in one small kernel (it reads the pinned descriptor block over the bus itself): a copy
  // command and a fill command ahead of the first kernel cost a one-frame call about 15 us, this launch 4
  static_assert(sizeof(FrameDesc) % 4 == 0 && sizeof(FrameDesc) <= 256 && sizeof(FrameCounters) % 4 == 0 && sizeof(FrameCounters) <= 256, "k_prologue: one word per thread");
  hipLaunchKernelGGL(k_prologue, dim3(n), dim3(64), 0, s, reinterpret_cast<const uint32_t*>(D->h_frames),
                     reinterpret_cast<uint32_t*>(D->d_frames), D->d_workctl, reinterpret_cast<uint32_t*>(D->d_counters),
                     (int)(sizeof(FrameDesc) / 4), (int)(sizeof(FrameCounters) / 4));
  if (D->fq_counters) HIP_TRY(hipMemsetAsync(D->d_fqprof, 0, (64 + 8) * 8, s));
  mark();
  {
    const int rc = issue_pipeline(D, n, ostride, s, fmt, mark);
    if (rc) return rc;
  }
  mark();
  return AMDAT_SUCCESS;
}

// An instantiated graph that is no longer wanted -- a capacity grew (its launches carry the old pointers), the cache evicted it,
// the submission path was pinned -- is RETIRED, not destroyed.  Root cause (round 6): hipGraphExecDestroy
// followed, with no device-wide wait in between, by the capture, instantiation and launch of the next graph crashes inside
// hipGraphLaunch -- a null node pointer in the runtime's walk over the new graph's nodes -- on the HIP runtime 7.0.51831 that
// PyTorch 2.10.0+rocm7.0 bundles (torch/lib/libamdhip64.so: the runtime every Python process of this repository binds, tests and
// bench included), and NOT on the system's ROCm 7.2.0 runtime the C and C++ hosts link.  tools/repro_graph_regrow.hip is the
// reproducer without library code: `repro_graph_regrow 400 destroy_nosync 3 16` dies with the same runtime frames on PyTorch's
// runtime and completes on ROCm 7.2's.  With the destroy taken out of the way -- retired graphs die with the handle, after its
// device-wide wait and with no capture behind them -- re-capturing on regrown buffers is clean (tools/stress_regrow.py with
// replay kept: 15 of 15 runs of 150 handles; with hipGraphExecDestroy at the regrowth: 11 of 12 runs die), so a regrown handle
// keeps graph replay.  The list is bounded: beyond AMDAT_MAX_RETIRED_GRAPHS the handle stops capturing new graphs (plain
// enqueues for the submission shapes it has no graph for; it says so once on stderr and through amdAprilTagsDebugGraphReplay).
#ifndef AMDAT_MAX_RETIRED_GRAPHS
#define AMDAT_MAX_RETIRED_GRAPHS 24
#endif
static void retire_graph(amdAprilTagsDetector_st* D, amdAprilTagsDetector_st::GraphEntry& g) {
  if (!g.exec) return;
#ifdef AMDAT_GRAPH_DESTROY_NOW   // (tools build of the crash hunt: what the library did until round 4)
  hipGraphExecDestroy(g.exec);
#else
  D->retired_graphs.push_back(g.exec);
  if (D->retired_graphs.size() > AMDAT_MAX_RETIRED_GRAPHS && D->graph_max_frames) {
    D->graph_max_frames = 0;   // (the live cache entries keep replaying; nothing new is captured)
    fprintf(stderr, "[apriltag_amd] handle %p: %zu retired launch graphs -- no new graphs are captured from here on (plain enqueues "
                    "for submission shapes without one)\n", (void*)D, D->retired_graphs.size());
  }
#endif
  g.exec = nullptr;
}
static void drop_graphs(amdAprilTagsDetector_st* D) {
  for (auto& g : D->graphs) retire_graph(D, g);
}
// A capacity grew: the captured launches carry the old pointers and capacities.  The next submission of each shape is captured
// again on the new buffers (rounds 4 and 5 gave replay up for good here, at ~0.1 ms per later one-frame call; see retire_graph).
static void drop_graphs_for_regrowth(amdAprilTagsDetector_st* D) {
  drop_graphs(D);
#ifdef AMDAT_REGROW_DROPS_REPLAY   // (tools build: round 5's behaviour)
  D->graph_max_frames = 0;
#endif
}

// After a capture that did not end in a graph: clear the error state and make sure no stream of the handle is left inside the
// dead capture (a side stream that joined it through the fork event and was never joined back stays "capturing": every later
// launch on it would fail with "operation failed due to a previous error during capture") -- such a stream is replaced.
static int recover_from_failed_capture(amdAprilTagsDetector_st* D, hipStream_t s) {
  for (int i = 0; i < 4 && hipGetLastError() != hipSuccess; i++) {}
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone) {
    hipGraph_t g = nullptr;
    (void)hipStreamEndCapture(s, &g);
    if (g) hipGraphDestroy(g);
  }
  for (auto& a : D->aux_stream) {
    st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(a, &st) != hipSuccess || st != hipStreamCaptureStatusNone) {
      (void)hipGetLastError();
      (void)hipStreamDestroy(a);
      a = nullptr;
      if (hipStreamCreateWithFlags(&a, hipStreamNonBlocking) != hipSuccess) return AMDAT_HIP_ERROR;
      // (graphs captured on the old stream object are still valid: a graph holds nodes and edges, not the capture's streams)
    }
  }
  for (int i = 0; i < 4 && hipGetLastError() != hipSuccess; i++) {}
  return AMDAT_SUCCESS;
}

// One pass of a submission over the device: captured-graph replay for small submissions, plain enqueues otherwise.
// launch_once enqueues it and returns; finish_once waits for it (and reads the stage events when profiling is on).
static int launch_once(amdAprilTagsDetector_st* D, uint32_t n, uint32_t ostride, hipStream_t s, uint32_t fmt) {
  D->launched_fmt = fmt;
  D->seq++;
  for (uint32_t i = 0; i < n; i++) D->h_frames[i].seq = D->seq;   // (k_prologue reads the pinned block when the launch executes)
  D->launched_n = n;
  const bool prof = D->profiling;
  D->events_recorded = false;
  int evi = 0;
  // profiling: one HIP event per stage boundary, and a roctx range per stage (it spans the stage's enqueues; rocprofv3
  // --marker-trace shows them beside the kernels they launched)
  const std::function<void()> mark = [&]() {
    if (!prof) return;
    if (roctx().push) {
      if (evi > 0) roctx().pop();
      if (evi < AMDAT_NUM_STAGES) roctx().push(kStageNames[evi]);
    }
    hipEventRecord(D->ev[evi++], s);
  };
  const std::function<void()> nomark = []() {};

  // Small submissions (the node's one-frame calls) are launch-bound: ~20 enqueues for well under a millisecond of
  // device work.  Their enqueue sequence is captured once per (frames, output stride, stream) into a hipGraph and
  // replayed; everything that changes between calls lives in the descriptor block the graph's first node uploads.  A
  // handful of instantiated graphs is kept (a host that alternates batch sizes or streams would otherwise re-capture on
  // every call, far slower than the plain enqueues the graph replaces); after a few consecutive misses with the cache
  // full, or one failed capture, the handle falls back to plain enqueues for good.
  if (n <= 8 && !prof && D->path_mode != AMDAT_PATH_THROUGHPUT) {
    amdAprilTagsDetector_st::GraphEntry* hit = nullptr;
    for (auto& g : D->graphs)
      if (g.exec && g.n == n && g.ostride == ostride && g.stream == s && g.fmt == fmt) hit = &g;
    if (hit) {
      D->graph_misses = 0;
      hit->last_use = ++D->graph_clock;
    } else if (D->graph_max_frames && n <= D->graph_max_frames && D->graph_misses < 8) {
      amdAprilTagsDetector_st::GraphEntry* slot = nullptr;
      for (auto& g : D->graphs) if (!g.exec) { slot = &g; break; }
      if (!slot) {   // evict the least recently used entry
        D->graph_misses++;
        slot = &D->graphs[0];
        for (auto& g : D->graphs) if (g.last_use < slot->last_use) slot = &g;
        retire_graph(D, *slot);   // (not destroyed here: see drop_graphs)
      }
      hipGraph_t graph = nullptr;
      bool ok = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess;
      if (ok) {
        const int rc = enqueue_submission(D, n, ostride, s, fmt, nomark);
        const hipError_t e = hipStreamEndCapture(s, &graph);
        ok = rc == AMDAT_SUCCESS && e == hipSuccess && graph != nullptr;
      }
      if (ok) ok = hipGraphInstantiate(&slot->exec, graph, nullptr, nullptr, 0) == hipSuccess;
      if (graph) hipGraphDestroy(graph);
      if (ok) { slot->n = n; slot->ostride = ostride; slot->fmt = fmt; slot->stream = s; slot->last_use = ++D->graph_clock; hit = slot; }
      else {
        // A capture can be invalidated from OUTSIDE the library: on this runtime a legacy-stream call of any other host thread on
        // the same device (a plain hipMemcpy) while this thread captures fails that call and poisons the capture, in every capture
        // mode (examples/multi_stream_host --shared-gpu met it with eight threads on one device).  The submission
        // then goes out as plain enqueues -- after the side streams have been taken out of the dead capture -- and the handle
        // tries a capture again on later submissions; three failures end graph replay for the handle.
        slot->exec = nullptr;
        const int crc = recover_from_failed_capture(D, s);
        if (crc) return crc;
        if (++D->capture_failures >= 3) D->graph_max_frames = 0;
      }
    }
    if (hit) {
      HIP_TRY(hipGraphLaunch(hit->exec, s));
      return AMDAT_SUCCESS;
    }
  }
  {
    const int rc = enqueue_submission(D, n, ostride, s, fmt, mark);
    if (rc) return rc;
  }
  D->events_recorded = prof;
  HIP_TRY(hipGetLastError());
  return AMDAT_SUCCESS;
}

static int finish_once(amdAprilTagsDetector_st* D, hipStream_t s) {
  HIP_TRY(hipStreamSynchronize(s));
  // The counters the host is about to act on must be THIS launch's: k_reconcile stamps them with the descriptor's sequence number
  // as its last store.  (The regrowth stress loop once ended with a handle that had not grown -- counters read before they were
  // final, under graph replay on ROCm 7.2.)  A mismatch waits for the whole device and looks again; if the results still are not
  // there the call fails instead of handing out an earlier launch's records.
  auto stamped = [&]() {
    for (uint32_t f = 0; f < D->launched_n; f++)
      if (reinterpret_cast<volatile FrameCounters*>(D->h_counters)[f].seq != D->seq) return false;
    return true;
  };
  if (!stamped()) {
    D->late_waits++;
    HIP_TRY(hipDeviceSynchronize());
    if (!stamped()) {
      fprintf(stderr, "[apriltag_amd] launch %u: results missing after a device-wide wait\n", D->seq);
      return AMDAT_HIP_ERROR;
    }
  }
  if (D->events_recorded) {
    for (int i = 0; i < AMDAT_NUM_STAGES; i++) {
      float ms = 0;
      hipEventElapsedTime(&ms, D->ev[i], D->ev[i + 1]);
      D->stage_ms[i] = ms;
    }
  }
  return AMDAT_SUCCESS;
}


// One batched submission; results land in h_out / h_counters with `ostride` records per frame.
// begin_batch fills the descriptor block and enqueues the submission; end_batch waits for it, and where a frame overflowed a
// capacity the handle may grow, grows it and runs the submission again (the descriptors are still in the pinned block).
static int begin_batch(amdAprilTagsDetector_st* D, uint32_t n, const amdAprilTagsImageInput_t* images,
                       const amdAprilTagsCameraIntrinsics_t* intr, uint32_t ostride, hipStream_t s, uint32_t fmt = AMDAT_ENC_MONO8) {
  if (D->inflight.active) return AMDAT_INVALID_ARGUMENT;   // one submission per handle at a time (amdAprilTagsWaitBatch first)
  DeviceGuard guard(D->device);
  if (!guard.ok) return AMDAT_HIP_ERROR;
  if (D->unusable) return AMDAT_OUT_OF_MEMORY;   // (never launch on the half-allocated buffers of a failed regrowth)
  { const int crc = ensure_colour_plane(D, fmt); if (crc) return crc; }
  fill_frames(D, n, images, intr, fmt);   // image pointers, pitches and intrinsics travel through the pinned descriptor block
  D->last_n = n;
  D->last_path = small_submission(D, D->P, n) ? AMDAT_PATH_LATENCY : AMDAT_PATH_THROUGHPUT;
  if (ostride > D->P.dcap) ostride = D->P.dcap;
  if (D->pending_hash_grow) {   // the pair table of the previous submission was crowded: grow it now (its buffers are dead)
    D->pending_hash_grow = false;
    if (D->grow_hash && D->P.hcap < D->hcap_hard) {
      drop_graphs_for_regrowth(D);
      const uint32_t before = D->P.hcap;
      D->P.hcap = D->P.hcap * 2 > D->hcap_hard ? D->hcap_hard : D->P.hcap * 2;
      if (alloc_hash_buffers(D) != AMDAT_SUCCESS || alloc_point_buffers(D) != AMDAT_SUCCESS) {
        D->P.hcap = before; D->grow_hash = false;
        if (alloc_hash_buffers(D) != AMDAT_SUCCESS || alloc_point_buffers(D) != AMDAT_SUCCESS) { D->unusable = true; return AMDAT_OUT_OF_MEMORY; }
      } else {
        D->grown++;
      }
    }
  }
  if (D->tables_dirty) { const int crc = clear_hash_tables(D); if (crc) return crc; }
  D->tables_dirty = true;
  const int rc = launch_once(D, n, ostride, s, fmt);
  if (rc) return rc;
  D->inflight.active = true; D->inflight.n = n; D->inflight.ostride = ostride; D->inflight.stream = s;
  return AMDAT_SUCCESS;
}

static int end_batch(amdAprilTagsDetector_st* D) {
  if (!D->inflight.active) return AMDAT_INVALID_ARGUMENT;
  DeviceGuard guard(D->device);
  if (!guard.ok) return AMDAT_HIP_ERROR;
  const uint32_t n = D->inflight.n, ostride = D->inflight.ostride;
  const hipStream_t s = D->inflight.stream;
  D->inflight.active = false;
  for (;;) {
    const int rc = finish_once(D, s);
    if (rc) return rc;
    D->tables_dirty = false;   // ran to its end: k_cluster_select left the pair table empty
    // A frame whose boundary points did not fit yields no clusters at all (flag 0x1), one whose component pairs did not fit
    // loses clusters (0x2).  Unless the host fixed the capacities, the buffers grow -- doubling, up to what no content
    // exceeds -- and the submission runs again; a pair table filled beyond a quarter grows for the next submission.
    bool pts_over = false, hash_over = false, hash_crowded = false, long_over = false;
    uint32_t nlong_max = 0;
    for (uint32_t f = 0; f < n; f++) {
      // (0x1 covers both the staging words and the long records: the counters say which list it was -- both count every attempt)
      if (D->h_counters[f].flags & 0x1u) {
        if (D->h_counters[f].nlong > D->P.lcap) { long_over = true; if (D->h_counters[f].nlong > nlong_max) nlong_max = D->h_counters[f].nlong; }
        if (D->h_counters[f].npoints_raw > D->P.pcap || D->h_counters[f].nlong <= D->P.lcap) pts_over = true;
      }
      hash_over |= (D->h_counters[f].flags & 0x2u) != 0;
      hash_crowded |= D->h_counters[f].nclusters > D->P.hcap / 4;
    }
    bool cands_over = false;
    for (uint32_t f = 0; f < n; f++) cands_over |= (D->h_counters[f].flags & AT_FLAG_CANDS) != 0;
    bool again = false;
    {   // the cluster list of a handle without an explicit max_clusters follows the content: to the next power of two that holds the
        // fullest frame (the counter counts every kept cluster, listed or not); the work lists of the quad fit follow it
      uint32_t ncl_max = 0;
      for (uint32_t f = 0; f < n; f++)
        if ((D->h_counters[f].flags & 0x4u) && D->h_counters[f].nclusters > D->P.ccap && D->h_counters[f].nclusters > ncl_max) ncl_max = D->h_counters[f].nclusters;
      if (ncl_max && D->grow_clusters && D->P.ccap < D->ccap_hard) {
        uint32_t ncap = D->P.ccap;
        while (ncap < ncl_max && ncap < D->ccap_hard) ncap *= 2;
        if (ncap > D->ccap_hard) ncap = D->ccap_hard;
        ClusterRec* nb = nullptr;
        const size_t nbytes = (size_t)D->cfg.max_batch * ncap * sizeof(ClusterRec);
        const uint32_t before = D->P.ccap;
        if (hipMalloc((void**)&nb, nbytes) == hipSuccess) {
          drop_graphs_for_regrowth(D);
          D->P.ccap = ncap;
          if (alloc_point_buffers(D) == AMDAT_SUCCESS) {
            hipFree(D->d_clusters);
            D->d_clusters = nb;
            D->device_bytes += nbytes - D->clusters_bytes;
            D->clusters_bytes = nbytes;
            D->grown++;
            again = true;
          } else {   // not enough memory for the longer work lists: the old capacities stay and the overflow is reported
            (void)hipGetLastError();
            hipFree(nb);
            D->P.ccap = before; D->grow_clusters = false;
            if (alloc_point_buffers(D) != AMDAT_SUCCESS) { D->unusable = true; return AMDAT_OUT_OF_MEMORY; }
          }
        } else {
          (void)hipGetLastError();
          D->grow_clusters = false;
        }
      }
    }
    if (!again) {   // the quad list of a handle without an explicit max_quads follows the content like the candidate list
      bool quads_over = false;
      for (uint32_t f = 0; f < n; f++) quads_over |= (D->h_counters[f].flags & 0x8u) != 0 && D->h_counters[f].nquads > D->P.qcap;
      if (quads_over && D->grow_quads && D->P.qcap < D->P.ccap) {
        const uint64_t want = (uint64_t)D->P.qcap * 2;
        const uint32_t ncap = want > D->P.ccap ? D->P.ccap : (uint32_t)want;
        QuadRec* nb = nullptr;
        const size_t nbytes = (size_t)D->cfg.max_batch * ncap * sizeof(QuadRec);
        if (hipMalloc((void**)&nb, nbytes) == hipSuccess) {
          drop_graphs_for_regrowth(D);
          hipFree(D->d_quads);
          D->d_quads = nb;
          D->device_bytes += nbytes - D->quads_bytes;
          D->quads_bytes = nbytes;
          D->P.qcap = ncap;
          D->grown++;
          again = true;
        } else {
          (void)hipGetLastError();
          D->grow_quads = false;   // (not enough memory: the overflow is reported from here on)
        }
      }
    }
    if (!again && cands_over) {
      if (D->P.cand_cap < D->P.ccap) {   // grow the candidate list and repeat
        drop_graphs_for_regrowth(D);
        const uint64_t want = (uint64_t)D->P.cand_cap * 2;
        const uint32_t ncap = want > D->P.ccap ? D->P.ccap : (uint32_t)want;
        FitCand* nb = nullptr;
        const size_t nbytes = (size_t)D->cfg.max_batch * ncap * sizeof(FitCand);
        if (hipMalloc((void**)&nb, nbytes) == hipSuccess) {
          hipFree(D->d_cands);
          D->d_cands = nb;
          D->device_bytes += nbytes - D->cands_bytes;
          D->cands_bytes = nbytes;
          D->P.cand_cap = ncap;
          D->grown++;
          again = true;
        }
      }
      // cannot grow: report it as what it is for the caller, a quad-list overflow
      if (!again)
        for (uint32_t f = 0; f < n; f++)
          if (D->h_counters[f].flags & AT_FLAG_CANDS) D->h_counters[f].flags = (D->h_counters[f].flags & ~AT_FLAG_CANDS) | 0x8u;
    }
    if (!again) {
      const bool can_pts = D->grow_points && D->P.pcap < D->pcap_hard;
      const bool can_long = D->grow_points && D->lcap_div > 1;
      const bool can_hash = D->grow_hash && D->P.hcap < D->hcap_hard;
      const bool redo = (pts_over && can_pts) || (long_over && can_long) || (hash_over && can_hash);
      if (!redo) {   // (a crowded table grows before the next submission: this one's buffers may still be inspected)
        if (hash_crowded && can_hash) D->pending_hash_grow = true;
        return AMDAT_SUCCESS;
      }
      drop_graphs_for_regrowth(D);   // captured launches carry the old pointers and capacities
      const uint32_t pcap_before = D->P.pcap, hcap_before = D->P.hcap, ldiv_before = D->lcap_div;
      if (pts_over && can_pts) { const uint64_t want = (uint64_t)D->P.pcap * 2; D->P.pcap = want > D->pcap_hard ? D->pcap_hard : (uint32_t)want; }
      if (long_over && can_long) {   // the smallest share of the (new) point capacity that holds what this submission asked for
        do D->lcap_div >>= 1; while (D->lcap_div > 1 && D->P.pcap / D->lcap_div < nlong_max);
      }
      if (hash_over && can_hash) D->P.hcap = D->P.hcap * 2 > D->hcap_hard ? D->hcap_hard : D->P.hcap * 2;
      int grc = alloc_hash_buffers(D);
      if (grc == AMDAT_SUCCESS) grc = alloc_point_buffers(D);   // (also when only the staging format changed)
      if (grc != AMDAT_SUCCESS) {   // not enough memory to grow: keep reporting the overflow with the old capacities
        D->P.pcap = pcap_before; D->P.hcap = hcap_before; D->lcap_div = ldiv_before;
        D->grow_points = false; D->grow_hash = false;
        if (alloc_hash_buffers(D) != AMDAT_SUCCESS || alloc_point_buffers(D) != AMDAT_SUCCESS) { D->unusable = true; return AMDAT_OUT_OF_MEMORY; }
        return AMDAT_SUCCESS;
      }
      D->grown++;
    }
    // run the submission again on the grown buffers
    if (D->tables_dirty) { const int crc = clear_hash_tables(D); if (crc) return crc; }
    D->tables_dirty = true;
    const int lrc = launch_once(D, n, ostride, s, D->launched_fmt);
    if (lrc) return lrc;
  }
}

static int run_batch(amdAprilTagsDetector_st* D, uint32_t n, const amdAprilTagsImageInput_t* images,
                     const amdAprilTagsCameraIntrinsics_t* intr, uint32_t ostride, hipStream_t s, uint32_t fmt = AMDAT_ENC_MONO8) {
  const int rc = begin_batch(D, n, images, intr, ostride, s, fmt);
  return rc ? rc : end_batch(D);
}

// copies the finished submission's records out of the pinned buffers
static void copy_out_ex(amdAprilTagsDetector_st* D, uint32_t n, uint32_t ostride, uint32_t max_dets, amdAprilTagsDetectionEx_t* dets_out, uint32_t* num_dets) {
  for (uint32_t f = 0; f < n; f++) {
    uint32_t k = D->h_counters[f].nout;
    if (k > ostride) k = ostride;
    num_dets[f] = k;
    memcpy(dets_out + (size_t)f * max_dets, D->h_out + (size_t)f * ostride, (size_t)k * sizeof(DetRec));
  }
}

int amdAprilTagsDetectBatchColorEx(amdAprilTagsHandle handle, uint32_t n, const amdAprilTagsImageInput_t* images, amdAprilTagsEncoding encoding,
                                   const amdAprilTagsCameraIntrinsics_t* per_frame_intrinsics, amdAprilTagsDetectionEx_t* dets_out,
                                   uint32_t* num_dets, uint32_t max_dets, amdAprilTagsStream stream) {
  if (!handle || !dets_out || !num_dets || max_dets == 0) return AMDAT_INVALID_ARGUMENT;
  int rc = check_images(handle, n, images, (uint32_t)encoding);
  if (rc) return rc;
  hipStream_t s = stream ? (hipStream_t)stream : handle->own_stream;
  uint32_t ostride = max_dets < handle->P.dcap ? max_dets : handle->P.dcap;
  rc = run_batch(handle, n, images, per_frame_intrinsics, ostride, s, (uint32_t)encoding);
  if (rc) return rc;
  copy_out_ex(handle, n, ostride, max_dets, dets_out, num_dets);
  return AMDAT_SUCCESS;
}

int amdAprilTagsDetectBatchEx(amdAprilTagsHandle handle, uint32_t n, const amdAprilTagsImageInput_t* images,
                              const amdAprilTagsCameraIntrinsics_t* per_frame_intrinsics, amdAprilTagsDetectionEx_t* dets_out,
                              uint32_t* num_dets, uint32_t max_dets, amdAprilTagsStream stream) {
  return amdAprilTagsDetectBatchColorEx(handle, n, images, AMDAT_ENC_MONO8, per_frame_intrinsics, dets_out, num_dets, max_dets, stream);
}

static void to_public(const DetRec& d, uint16_t family_enum, uint32_t corner_convention, amdAprilTagsID_t* o) {
  memset(o, 0, sizeof(*o));
  o->id = (uint16_t)d.id;
  // library-native corner order = message order = AprilRobotics p[3-i]; AMDAT_CORNERS_ROTATED_180 starts two corners later
  // and turns the tag frame about its normal: R * Rz(pi) = R with its first two columns negated
  const int turn = corner_convention == AMDAT_CORNERS_ROTATED_180 ? 2 : 0;
  const double sgn = turn ? -1.0 : 1.0;
  for (int i = 0; i < 4; i++) { o->corners[i].x = (float)d.p[(3 - i + turn) & 3][0]; o->corners[i].y = (float)d.p[(3 - i + turn) & 3][1]; }
  o->hamming_error = (uint16_t)d.hamming;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) o->orientation[c * 3 + r] = (float)(c < 2 ? sgn * d.R[r * 3 + c] : d.R[r * 3 + c]);  // column-major
  for (int i = 0; i < 3; i++) o->translation[i] = (float)d.t[i];
  o->family = family_enum;
  o->decision_margin = d.decision_margin;
  o->center.x = (float)d.c[0];
  o->center.y = (float)d.c[1];
}

static void copy_out_public(amdAprilTagsDetector_st* D, uint32_t n, uint32_t ostride, uint32_t max_tags, amdAprilTagsID_t* tags_out, uint32_t* num_tags) {
  for (uint32_t f = 0; f < n; f++) {
    uint32_t k = D->h_counters[f].nout;
    if (k > ostride) k = ostride;
    num_tags[f] = k;
    for (uint32_t i = 0; i < k; i++) {
      const DetRec& d = D->h_out[(size_t)f * ostride + i];
      to_public(d, (uint16_t)D->cfg.families[d.family], D->cfg.corner_convention, &tags_out[(size_t)f * max_tags + i]);
    }
  }
}

int amdAprilTagsDetectBatchColor(amdAprilTagsHandle handle, uint32_t n, const amdAprilTagsImageInput_t* images, amdAprilTagsEncoding encoding,
                                 const amdAprilTagsCameraIntrinsics_t* per_frame_intrinsics, amdAprilTagsID_t* tags_out,
                                 uint32_t* num_tags, uint32_t max_tags, amdAprilTagsStream stream) {
  if (!handle || !tags_out || !num_tags || max_tags == 0) return AMDAT_INVALID_ARGUMENT;
  int rc = check_images(handle, n, images, (uint32_t)encoding);
  if (rc) return rc;
  hipStream_t s = stream ? (hipStream_t)stream : handle->own_stream;
  uint32_t ostride = max_tags < handle->P.dcap ? max_tags : handle->P.dcap;
  rc = run_batch(handle, n, images, per_frame_intrinsics, ostride, s, (uint32_t)encoding);
  if (rc) return rc;
  copy_out_public(handle, n, ostride, max_tags, tags_out, num_tags);
  return AMDAT_SUCCESS;
}

int amdAprilTagsDetectBatch(amdAprilTagsHandle handle, uint32_t n, const amdAprilTagsImageInput_t* images,
                            const amdAprilTagsCameraIntrinsics_t* per_frame_intrinsics, amdAprilTagsID_t* tags_out,
                            uint32_t* num_tags, uint32_t max_tags, amdAprilTagsStream stream) {
  return amdAprilTagsDetectBatchColor(handle, n, images, AMDAT_ENC_MONO8, per_frame_intrinsics, tags_out, num_tags, max_tags, stream);
}

int amdAprilTagsDetectColor(amdAprilTagsHandle handle, const amdAprilTagsImageInput_t* img_input, amdAprilTagsEncoding encoding,
                            amdAprilTagsID_t* tags_out, uint32_t* num_tags, uint32_t max_tags, amdAprilTagsStream stream) {
  return amdAprilTagsDetectBatchColor(handle, 1, img_input, encoding, nullptr, tags_out, num_tags, max_tags, stream);
}

// ---- the same submission in two halves: enqueue, then wait -------------------------------------------------------------
// amdAprilTagsSubmitBatch returns as soon as the submission is enqueued on the stream; the host is free -- to copy the NEXT
// batch's frames to the device on a stream of its own, to serve other handles -- until amdAprilTagsWaitBatch[Ex] blocks for
// the results.  One submission per handle may be in flight; the images (and their device buffers) must stay valid until the
// wait returns.  Submit + Wait gives exactly what the blocking call gives (the blocking call IS the two, back to back).
int amdAprilTagsSubmitBatchColor(amdAprilTagsHandle handle, uint32_t n, const amdAprilTagsImageInput_t* images, amdAprilTagsEncoding encoding,
                                 const amdAprilTagsCameraIntrinsics_t* per_frame_intrinsics, uint32_t max_tags, amdAprilTagsStream stream) {
  if (!handle || max_tags == 0) return AMDAT_INVALID_ARGUMENT;
  int rc = check_images(handle, n, images, (uint32_t)encoding);
  if (rc) return rc;
  hipStream_t s = stream ? (hipStream_t)stream : handle->own_stream;
  const uint32_t ostride = max_tags < handle->P.dcap ? max_tags : handle->P.dcap;
  rc = begin_batch(handle, n, images, per_frame_intrinsics, ostride, s, (uint32_t)encoding);
  if (rc) return rc;
  handle->inflight.max_out = max_tags;
  return AMDAT_SUCCESS;
}

int amdAprilTagsSubmitBatch(amdAprilTagsHandle handle, uint32_t n, const amdAprilTagsImageInput_t* images,
                            const amdAprilTagsCameraIntrinsics_t* per_frame_intrinsics, uint32_t max_tags, amdAprilTagsStream stream) {
  return amdAprilTagsSubmitBatchColor(handle, n, images, AMDAT_ENC_MONO8, per_frame_intrinsics, max_tags, stream);
}

int amdAprilTagsWaitBatch(amdAprilTagsHandle handle, amdAprilTagsID_t* tags_out, uint32_t* num_tags) {
  if (!handle || !tags_out || !num_tags || !handle->inflight.active) return AMDAT_INVALID_ARGUMENT;
  const uint32_t n = handle->inflight.n, ostride = handle->inflight.ostride, max_tags = handle->inflight.max_out;
  const int rc = end_batch(handle);
  if (rc) return rc;
  copy_out_public(handle, n, ostride, max_tags, tags_out, num_tags);
  return AMDAT_SUCCESS;
}

int amdAprilTagsWaitBatchEx(amdAprilTagsHandle handle, amdAprilTagsDetectionEx_t* dets_out, uint32_t* num_dets) {
  if (!handle || !dets_out || !num_dets || !handle->inflight.active) return AMDAT_INVALID_ARGUMENT;
  const uint32_t n = handle->inflight.n, ostride = handle->inflight.ostride, max_dets = handle->inflight.max_out;
  const int rc = end_batch(handle);
  if (rc) return rc;
  copy_out_ex(handle, n, ostride, max_dets, dets_out, num_dets);
  return AMDAT_SUCCESS;
}

int amdAprilTagsDetect(amdAprilTagsHandle handle, const amdAprilTagsImageInput_t* img_input, amdAprilTagsID_t* tags_out,
                       uint32_t* num_tags, uint32_t max_tags, amdAprilTagsStream stream) {
  return amdAprilTagsDetectBatch(handle, 1, img_input, nullptr, tags_out, num_tags, max_tags, stream);
}

int amdAprilTagsSetFrameSkews(amdAprilTagsHandle handle, uint32_t n, const float* skews) {
  if (!handle || n > handle->cfg.max_batch || (n && !skews) || handle->inflight.active) return AMDAT_INVALID_ARGUMENT;
  handle->frame_skew.assign(skews, skews + n);
  return AMDAT_SUCCESS;
}

int amdAprilTagsGetFrameFlags(amdAprilTagsHandle handle, uint32_t* flags, uint32_t n) {
  if (!handle || !flags || n > handle->last_n || handle->inflight.active) return AMDAT_INVALID_ARGUMENT;
  for (uint32_t i = 0; i < n; i++) flags[i] = handle->h_counters[i].flags;
  return AMDAT_SUCCESS;
}

int amdAprilTagsThresholdOnly(amdAprilTagsHandle handle, uint32_t n, const amdAprilTagsImageInput_t* images,
                              amdAprilTagsStream stream) {
  return amdAprilTagsThresholdOnlyColor(handle, n, images, AMDAT_ENC_MONO8, stream);
}

int amdAprilTagsThresholdOnlyColor(amdAprilTagsHandle handle, uint32_t n, const amdAprilTagsImageInput_t* images, amdAprilTagsEncoding encoding,
                                   amdAprilTagsStream stream) {
  if (!handle || handle->inflight.active) return AMDAT_INVALID_ARGUMENT;
  const uint32_t fmt = (uint32_t)encoding;
  int rc = check_images(handle, n, images, fmt);
  if (rc) return rc;
  hipStream_t s = stream ? (hipStream_t)stream : handle->own_stream;
  DeviceGuard guard(handle->device);
  if (!guard.ok) return AMDAT_HIP_ERROR;
  rc = ensure_colour_plane(handle, fmt);
  if (rc) return rc;
  fill_frames(handle, n, images, nullptr, fmt);
  handle->last_n = n;
  HIP_TRY(hipMemcpyAsync(handle->d_frames, handle->h_frames, n * sizeof(FrameDesc), hipMemcpyHostToDevice, s));
  if (handle->profiling) hipEventRecord(handle->ev[1], s);
  { DetParams P0 = handle->P; P0.frame0 = 0; launch_threshold(handle, P0, n, s, fmt); }
  if (handle->profiling) hipEventRecord(handle->ev[2], s);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(s));
  if (handle->profiling) {
    memset(handle->stage_ms, 0, sizeof(handle->stage_ms));
    hipEventElapsedTime(&handle->stage_ms[1], handle->ev[1], handle->ev[2]);
  }
  return AMDAT_SUCCESS;
}

int amdAprilTagsConvertToMono8(const void* src_dev, size_t src_pitch, const char* encoding, uint32_t width, uint32_t height,
                               uint8_t* dst_dev, size_t dst_pitch, amdAprilTagsStream stream) {
  if (!src_dev || !dst_dev || !encoding || width == 0 || height == 0) return AMDAT_INVALID_ARGUMENT;
  hipStream_t s = (hipStream_t)stream;
  const uint8_t* src = (const uint8_t*)src_dev;
  dim3 grid((width + 1023) / 1024, height), block(256);
  if (!strcmp(encoding, "mono8")) {
    HIP_TRY(hipMemcpy2DAsync(dst_dev, dst_pitch, src_dev, src_pitch, width, height, hipMemcpyDeviceToDevice, s));
  } else if (!strcmp(encoding, "rgb8")) {
    hipLaunchKernelGGL((k_to_mono8<3, 0, 2>), grid, block, 0, s, src, src_pitch, dst_dev, dst_pitch, width, height);
  } else if (!strcmp(encoding, "bgr8")) {
    hipLaunchKernelGGL((k_to_mono8<3, 2, 0>), grid, block, 0, s, src, src_pitch, dst_dev, dst_pitch, width, height);
  } else if (!strcmp(encoding, "rgba8")) {
    hipLaunchKernelGGL((k_to_mono8<4, 0, 2>), grid, block, 0, s, src, src_pitch, dst_dev, dst_pitch, width, height);
  } else if (!strcmp(encoding, "bgra8")) {
    hipLaunchKernelGGL((k_to_mono8<4, 2, 0>), grid, block, 0, s, src, src_pitch, dst_dev, dst_pitch, width, height);
  } else {
    return AMDAT_UNSUPPORTED;
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(s));
  return AMDAT_SUCCESS;
}

int amdAprilTagsResizeMono8(const uint8_t* src_dev, size_t src_pitch, uint32_t sw, uint32_t sh, uint8_t* dst_dev, size_t dst_pitch,
                            uint32_t dw, uint32_t dh, amdAprilTagsStream stream) {
  if (!src_dev || !dst_dev || sw == 0 || sh == 0 || dw == 0 || dh == 0 || src_pitch < sw || dst_pitch < dw) return AMDAT_INVALID_ARGUMENT;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_resize_mono8, dim3((dw + 63) / 64, (dh + 3) / 4), dim3(256), 0, s, src_dev, src_pitch, (int)sw, (int)sh, dst_dev,
                     dst_pitch, (int)dw, (int)dh);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(s));
  return AMDAT_SUCCESS;
}

int amdAprilTagsRectifyMono8(const uint8_t* src_dev, size_t src_pitch, uint8_t* dst_dev, size_t dst_pitch, uint32_t width,
                             uint32_t height, const double* K9, const double* D5, const double* Knew9, amdAprilTagsStream stream) {
  if (!src_dev || !dst_dev || !K9 || !D5 || !Knew9 || width == 0 || height == 0 || src_pitch < width || dst_pitch < width)
    return AMDAT_INVALID_ARGUMENT;
  RectifyParams R = {K9[0], K9[4], K9[2], K9[5], D5[0], D5[1], D5[2], D5[3], D5[4], Knew9[0], Knew9[4], Knew9[2], Knew9[5]};
  if (R.nfx == 0.0 || R.nfy == 0.0) return AMDAT_INVALID_ARGUMENT;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_rectify_mono8, dim3((width + 63) / 64, (height + 3) / 4), dim3(256), 0, s, src_dev, src_pitch, dst_dev, dst_pitch,
                     (int)width, (int)height, R);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(s));
  return AMDAT_SUCCESS;
}

int amdAprilTagsDeviceAlloc(void** dev_ptr, size_t bytes) {
  if (!dev_ptr || bytes == 0) return AMDAT_INVALID_ARGUMENT;
  return hipMalloc(dev_ptr, bytes) == hipSuccess ? AMDAT_SUCCESS : AMDAT_OUT_OF_MEMORY;
}
int amdAprilTagsDeviceFree(void* dev_ptr) {
  if (!dev_ptr) return AMDAT_INVALID_ARGUMENT;
  return hipFree(dev_ptr) == hipSuccess ? AMDAT_SUCCESS : AMDAT_HIP_ERROR;
}
int amdAprilTagsCopyToDevice(void* dst_dev, const void* src_host, size_t bytes, amdAprilTagsStream stream) {
  if (!dst_dev || !src_host) return AMDAT_INVALID_ARGUMENT;
  HIP_TRY(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return AMDAT_SUCCESS;
}

// enqueue-only copy + stream helpers for hosts that do not link the HIP runtime (the node shell): the copy is ordered on `stream`
// ahead of the detection the host enqueues on the same stream, and the detection's own wait covers it -- no second host wait
int amdAprilTagsCopyToDeviceAsync(void* dst_dev, const void* src_host, size_t bytes, amdAprilTagsStream stream) {
  if (!dst_dev || !src_host || !stream) return AMDAT_INVALID_ARGUMENT;
  HIP_TRY(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  return AMDAT_SUCCESS;
}
int amdAprilTagsStreamCreate(amdAprilTagsStream* stream) {
  if (!stream) return AMDAT_INVALID_ARGUMENT;
  hipStream_t s = nullptr;
  HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream = (amdAprilTagsStream)s;
  return AMDAT_SUCCESS;
}
int amdAprilTagsStreamDestroy(amdAprilTagsStream stream) {
  if (!stream) return AMDAT_INVALID_ARGUMENT;
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  HIP_TRY(hipStreamDestroy((hipStream_t)stream));
  return AMDAT_SUCCESS;
}

int amdAprilTagsDebugCopy(amdAprilTagsHandle handle, uint32_t frame, amdAprilTagsDebugBuffer what, void* host_dst,
                          size_t capacity, size_t* bytes) {
  if (!handle || !bytes || frame >= handle->last_n || handle->inflight.active) return AMDAT_INVALID_ARGUMENT;
  const DetParams& P = handle->P;
  DeviceGuard guard(handle->device);
  if (!guard.ok) return AMDAT_HIP_ERROR;
  HIP_TRY(hipDeviceSynchronize());
  const FrameCounters& fc = handle->h_counters[frame];
  const size_t npx = (size_t)P.W * P.H;
  const void* src = nullptr;
  size_t sz = 0;
  std::vector<uint8_t> tmp;
  switch (what) {
    case AMDAT_DBG_GRAY:
    case AMDAT_DBG_THRESH: {
      // de-pitch rows on the host
      const uint8_t* base = nullptr;
      size_t pitch = P.WS;
      if (what == AMDAT_DBG_THRESH) base = handle->d_thr + (size_t)frame * P.H * P.WS;
      else if (P.decimate > 1) base = handle->d_gray + (size_t)frame * P.H * P.WS;
      else { base = handle->h_frames[frame].img; pitch = handle->h_frames[frame].pitch; }
      tmp.resize(npx);
      HIP_TRY(hipMemcpy2D(tmp.data(), P.W, base, pitch, P.W, P.H, hipMemcpyDeviceToHost));
      *bytes = npx;
      if (host_dst) memcpy(host_dst, tmp.data(), npx < capacity ? npx : capacity);
      return AMDAT_SUCCESS;
    }
    case AMDAT_DBG_LABEL: {
      DetParams Pf = P;
      Pf.frame0 = (int)frame;
      hipLaunchKernelGGL(k_cc_flatten, dim3((unsigned)((npx + 255) / 256), 1, 1), dim3(256), 0, 0, handle->d_label, Pf);
      HIP_TRY(hipDeviceSynchronize());
      src = handle->d_label + (size_t)frame * npx; sz = npx * 4;
      break;
    }
    case AMDAT_DBG_CSIZE: src = handle->d_csize + (size_t)frame * npx; sz = npx * 4; break;
    case AMDAT_DBG_CLUSTERS:
      src = handle->d_clusters + (size_t)frame * P.ccap;
      sz = (size_t)(fc.nclusters < P.ccap ? fc.nclusters : P.ccap) * sizeof(ClusterRec);
      break;
    case AMDAT_DBG_POINTS:
      src = handle->d_pts + (size_t)frame * P.pcap;
      sz = (size_t)(fc.npoints_kept < P.pcap ? fc.npoints_kept : P.pcap) * 4;
      break;
    case AMDAT_DBG_QUADS:
      src = handle->d_quads + (size_t)frame * P.qcap;
      sz = (size_t)(fc.nquads < P.qcap ? fc.nquads : P.qcap) * sizeof(QuadRec);
      break;
    case AMDAT_DBG_FQPROF:
      src = handle->d_fqprof; sz = (64 + 8) * 8;
      break;
    case AMDAT_DBG_COUNTS: {
      uint32_t c[8] = {fc.npoints_raw, fc.nclusters, fc.npoints_kept, fc.nquads, fc.ndets, fc.flags, (uint32_t)P.W, (uint32_t)P.H};
      *bytes = sizeof(c);
      if (host_dst) memcpy(host_dst, c, sizeof(c) < capacity ? sizeof(c) : capacity);
      return AMDAT_SUCCESS;
    }
    default: return AMDAT_INVALID_ARGUMENT;
  }
  *bytes = sz;
  if (host_dst && sz) HIP_TRY(hipMemcpy(host_dst, src, sz < capacity ? sz : capacity, hipMemcpyDeviceToHost));
  return AMDAT_SUCCESS;
}

#include "tools_timeline.h"   // debug entry points of the -DAMDAT_FQ_TIMELINE tools build; empty in the product build

int amdAprilTagsDebugMath(int op, uint32_t n, const double* a, const double* b, double* out) {
  if (!a || !b || !out || n == 0) return AMDAT_INVALID_ARGUMENT;
  double *da = nullptr, *db = nullptr, *dout = nullptr;
  int rc = AMDAT_HIP_ERROR;
  if (hipMalloc((void**)&da, n * 8) == hipSuccess && hipMalloc((void**)&db, n * 8) == hipSuccess &&
      hipMalloc((void**)&dout, n * 8) == hipSuccess && hipMemcpy(da, a, n * 8, hipMemcpyHostToDevice) == hipSuccess &&
      hipMemcpy(db, b, n * 8, hipMemcpyHostToDevice) == hipSuccess) {
    hipLaunchKernelGGL(k_debug_math, dim3((n + 255) / 256), dim3(256), 0, 0, op, n, da, db, dout);
    if (hipGetLastError() == hipSuccess && hipMemcpy(out, dout, n * 8, hipMemcpyDeviceToHost) == hipSuccess) rc = AMDAT_SUCCESS;
  }
  hipFree(da); hipFree(db); hipFree(dout);
  return rc;
}

}  // extern "C"
