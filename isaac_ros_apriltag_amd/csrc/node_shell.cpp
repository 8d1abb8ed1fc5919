// node_shell.cpp -- implementation of include/apriltag_node_shell.hpp: the reference node's host logic
// (src/apriltag_node.cpp:389-623) over the C ABI of libapriltag_amd.so.  Built into libapriltag_node.so;
// the detector library is bound at run time (dlopen next to this file's .so), so this translation unit
// needs neither HIP headers nor ROS.
#include "../../include/apriltag_node_shell.hpp"

#include <dlfcn.h>

#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <set>
#include <sstream>
#include <stdexcept>

#include "../../include/apriltag_amd.h"

namespace amd {
namespace isaac_ros {
namespace apriltag {

namespace {

// Entry points of libapriltag_amd.so used by the shell.
struct DetectorApi {
  void* lib = nullptr;
  decltype(&amdCreateAprilTagsDetectorEx) create_ex = nullptr;
  decltype(&amdAprilTagsDefaultConfig) default_config = nullptr;
  decltype(&amdAprilTagsDetect) detect = nullptr;
  decltype(&amdAprilTagsDetectBatch) detect_batch = nullptr;
  decltype(&amdAprilTagsDestroy) destroy = nullptr;
  decltype(&amdAprilTagsFamilyFromName) family_from_name = nullptr;
  decltype(&amdAprilTagsConvertToMono8) to_mono8 = nullptr;
  decltype(&amdAprilTagsDeviceAlloc) dev_alloc = nullptr;
  decltype(&amdAprilTagsDeviceFree) dev_free = nullptr;
  decltype(&amdAprilTagsCopyToDevice) copy_to_device = nullptr;
  decltype(&amdAprilTagsSetFrameSkews) set_frame_skews = nullptr;
  decltype(&amdAprilTagsDetectColor) detect_color = nullptr;
  decltype(&amdAprilTagsDetectBatchColor) detect_batch_color = nullptr;
  decltype(&amdAprilTagsEncodingFromName) encoding_from_name = nullptr;
  decltype(&amdAprilTagsCopyToDeviceAsync) copy_to_device_async = nullptr;
  decltype(&amdAprilTagsStreamCreate) stream_create = nullptr;
  decltype(&amdAprilTagsStreamDestroy) stream_destroy = nullptr;
};

DetectorApi& api() {
  static DetectorApi a;
  if (a.lib) return a;
  Dl_info info;
  std::string path = "libapriltag_amd.so";
  if (dladdr(reinterpret_cast<void*>(&api), &info) && info.dli_fname) {
    std::string self(info.dli_fname);
    const size_t slash = self.find_last_of('/');
    if (slash != std::string::npos) path = self.substr(0, slash + 1) + "libapriltag_amd.so";
  }
  a.lib = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
  if (!a.lib) throw std::runtime_error(std::string("cannot load libapriltag_amd.so: ") + dlerror());
#define BIND(field, name)                                                        \
  a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.lib, name));             \
  if (!a.field) throw std::runtime_error(std::string("missing symbol ") + name);
  BIND(create_ex, "amdCreateAprilTagsDetectorEx")
  BIND(default_config, "amdAprilTagsDefaultConfig")
  BIND(detect, "amdAprilTagsDetect")
  BIND(detect_batch, "amdAprilTagsDetectBatch")
  BIND(destroy, "amdAprilTagsDestroy")
  BIND(family_from_name, "amdAprilTagsFamilyFromName")
  BIND(to_mono8, "amdAprilTagsConvertToMono8")
  BIND(dev_alloc, "amdAprilTagsDeviceAlloc")
  BIND(dev_free, "amdAprilTagsDeviceFree")
  BIND(copy_to_device, "amdAprilTagsCopyToDevice")
  BIND(set_frame_skews, "amdAprilTagsSetFrameSkews")
  BIND(detect_color, "amdAprilTagsDetectColor")
  BIND(detect_batch_color, "amdAprilTagsDetectBatchColor")
  BIND(encoding_from_name, "amdAprilTagsEncodingFromName")
  BIND(copy_to_device_async, "amdAprilTagsCopyToDeviceAsync")
  BIND(stream_create, "amdAprilTagsStreamCreate")
  BIND(stream_destroy, "amdAprilTagsStreamDestroy")
#undef BIND
  return a;
}

// Family strings the reference accepts as parameter values (src/apriltag_node.cpp:47-58).
const char* const kKnownFamilyStrings[] = {"tag36h11", "tag16h5", "tag25h9", "tag36h10", "circle21h7",
                                           "circle49h12", "custom48h12", "standard41h12", "standard52h13"};

// Encodings the node's input conversion accepts (src/apriltag_node.cpp:76-82).
int bytes_per_pixel(const std::string& enc) {
  if (enc == "mono8") return 1;
  if (enc == "rgb8" || enc == "bgr8") return 3;
  if (enc == "rgba8" || enc == "bgra8") return 4;
  return 0;
}

// `backends` is the reference's VPI backend flag string: one name or a comma-separated list
// (test/isaac_ros_apriltag_backends_compare_test.py:33-37 uses 'CPU', 'CUDA', 'PVA').  Names the reference's
// VPI builds know plus this library's own.  Throws on an unknown name.
std::set<std::string> parse_backends(const std::string& backends) {
  static const char* const kKnown[] = {"CPU", "CUDA", "PVA", "VIC", "NVENC", "OFA", "ALL", "HIP", "GPU"};
  std::set<std::string> out;
  size_t pos = 0;
  while (pos <= backends.size()) {
    size_t comma = backends.find(',', pos);
    if (comma == std::string::npos) comma = backends.size();
    std::string tok = backends.substr(pos, comma - pos);
    const size_t a = tok.find_first_not_of(" \t"), b = tok.find_last_not_of(" \t");
    tok = (a == std::string::npos) ? std::string() : tok.substr(a, b - a + 1);
    for (auto& ch : tok) ch = static_cast<char>(std::toupper(static_cast<unsigned char>(ch)));
    bool known = false;
    for (const char* k : kKnown) known |= tok == k;
    if (!known) throw std::runtime_error("Unrecognized backend '" + tok + "' in 'backends' parameter");
    out.insert(tok);
    pos = comma + 1;
  }
  return out;
}

// Rotation matrix (column-major 3x3 float, as cuAprilTagsID_t::orientation) -> quaternion, the
// construction Eigen::Quaternion<float>(matrix) performs in ToTransformMsg (src/apriltag_node.cpp:409-427).
Quaternion quaternion_from_colmajor(const float* o) {
  auto m = [&](int r, int c) { return o[c * 3 + r]; };
  float q[4];  // x y z w
  float t = m(0, 0) + m(1, 1) + m(2, 2);
  if (t > 0.0f) {
    t = std::sqrt(t + 1.0f);
    q[3] = 0.5f * t;
    t = 0.5f / t;
    q[0] = (m(2, 1) - m(1, 2)) * t;
    q[1] = (m(0, 2) - m(2, 0)) * t;
    q[2] = (m(1, 0) - m(0, 1)) * t;
  } else {
    int i = 0;
    if (m(1, 1) > m(0, 0)) i = 1;
    if (m(2, 2) > m(i, i)) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0f);
    q[i] = 0.5f * t;
    t = 0.5f / t;
    q[3] = (m(k, j) - m(j, k)) * t;
    q[j] = (m(j, i) + m(i, j)) * t;
    q[k] = (m(k, i) + m(i, k)) * t;
  }
  Quaternion out;
  out.x = q[0]; out.y = q[1]; out.z = q[2]; out.w = q[3];
  return out;
}

// AprilTagDetectionArray + TF transforms of one frame from the detector's records (src/apriltag_node.cpp:499-549)
void assemble_messages(const amdAprilTagsID_t* tags, uint32_t num_detections, const Header& info_header, const std::string& family,
                       AprilTagDetectionArray* msg_out, std::vector<TransformStamped>* tfs_out) {
  AprilTagDetectionArray& msg = *msg_out;
  std::vector<TransformStamped>& tfs = *tfs_out;
  msg.header = info_header;  // camera_info header, not the image header (src/apriltag_node.cpp:501)
  for (uint32_t i = 0; i < num_detections; i++) {
    const amdAprilTagsID_t& d = tags[i];
    AprilTagDetection det;
    det.family = family;
    det.id = d.id;
    for (int c = 0; c < 4; c++) {
      det.corners[c].x = d.corners[c].x;
      det.corners[c].y = d.corners[c].y;
    }
    // The reference intersects the diagonals in slope/intercept form, which divides by zero when a
    // diagonal is vertical (src/apriltag_node.cpp:519-530); the library reports H(0,0) directly.
    det.center.x = d.center.x;
    det.center.y = d.center.y;
    TransformStamped tf;
    tf.header = info_header;
    tf.child_frame_id = family + ":" + std::to_string(d.id);
    tf.transform.translation.x = d.translation[0];
    tf.transform.translation.y = d.translation[1];
    tf.transform.translation.z = d.translation[2];
    tf.transform.rotation = quaternion_from_colmajor(d.orientation);
    tfs.push_back(tf);
    det.pose.pose.pose.position.x = tf.transform.translation.x;
    det.pose.pose.pose.position.y = tf.transform.translation.y;
    det.pose.pose.pose.position.z = tf.transform.translation.z;
    det.pose.pose.pose.orientation = tf.transform.rotation;
    msg.detections.push_back(det);
  }
}

// Backend / family validation of the constructor (src/apriltag_node.cpp:575-599): returns the family enum, throws with
// the reference's text otherwise.  cuapriltags_mode: exactly {CUDA} was asked for (tag36h11 only).
int validate_family(const NodeOptions& options, bool* cuapriltags_mode) {
  const std::set<std::string> backends = parse_backends(options.backends);
  *cuapriltags_mode = backends.size() == 1 && *backends.begin() == "CUDA";
  std::set<std::string> supported;
  if (*cuapriltags_mode) {
    supported.insert("tag36h11");
  } else {
    // any other backend list: the reference runs VPI, whose family table is src/apriltag_node.cpp:47-58;
    // here the families this library can decode = those with a code table (built in or registered by the host)
    for (const char* f : kKnownFamilyStrings)
      if (api().family_from_name(f) >= 0) supported.insert(f);
    if (api().family_from_name(options.tag_family.c_str()) >= 0) supported.insert(options.tag_family);
  }
  if (supported.find(options.tag_family) == supported.end()) {
    std::ostringstream os;
    os << "Tag family not supported by specified backend: '" << options.tag_family << "'" << std::endl;
    os << "'tag_family' parameter must be one of:" << std::endl;
    for (const auto& f : supported) os << f << std::endl;
    std::fprintf(stderr, "[apriltag_node] FATAL: Tag family not supported by specified backend: '%s'\n", options.tag_family.c_str());
    throw std::runtime_error(os.str());
  }
  return api().family_from_name(options.tag_family.c_str());
}

}  // namespace

struct AprilTagNode::Impl {
  NodeOptions opt;
  DetectionsCallback on_detections;
  TransformsCallback on_transforms;
  // lazily created on the first frame (src/apriltag_node.cpp:618-620)
  bool initialized = false;
  amdAprilTagsHandle detector = nullptr;
  int family_enum = -1;
  uint32_t width = 0, height = 0;
  void* d_input = nullptr;   // staging for host images
  size_t d_input_bytes = 0;
  amdAprilTagsStream stream = nullptr;   // the host-to-device copy and the detection are ordered on it: one host wait per frame

  // exactly {CUDA}: the reference runs cuAprilTags, which decodes tag36h11 only (src/apriltag_node.cpp:429-432)
  bool cuapriltags_mode = false;

  void Initialize(const Image& image, const CameraInfo& info) {
    if (opt.max_tags <= 0) throw std::runtime_error("'max_tags' must be positive");
    if (detector) { api().destroy(detector); detector = nullptr; }   // left over from a failed attempt
    // intrinsics from K, double -> float as the reference does (src/apriltag_node.cpp:442-447)
    amdAprilTagsConfig_t cfg;
    api().default_config(&cfg, info.width, info.height);
    cfg.tile_size = opt.tile_size;
    cfg.decimate = opt.decimate;
    cfg.num_families = 1;
    cfg.families[0] = static_cast<amdAprilTagsFamily>(family_enum);
    cfg.intrinsics.fx = static_cast<float>(info.k[0]);
    cfg.intrinsics.fy = static_cast<float>(info.k[4]);
    cfg.intrinsics.cx = static_cast<float>(info.k[2]);
    cfg.intrinsics.cy = static_cast<float>(info.k[5]);
    // the VPI path also passes the skew K[1] (src/apriltag_node.cpp:215-225); cuAprilTags has no such field
    cfg.skew = cuapriltags_mode ? 0.0f : static_cast<float>(info.k[1]);
    cfg.tag_size = static_cast<float>(opt.size);
    cfg.max_batch = 1;
    const int error = api().create_ex(&detector, &cfg);
    if (error != 0) {
      // same text as src/apriltag_node.cpp:453-457 with the library name replaced
      throw std::runtime_error("Failed to create AprilTags detector (error code " + std::to_string(error) + ")");
    }
    width = info.width;
    height = info.height;
    if (!stream && api().stream_create(&stream) != 0) throw std::runtime_error("stream creation failed");
    initialized = true;   // only now: a failed creation is retried (and reported) on the next frame
    (void)image;
  }

  void OnCameraFrame(const Image& image, const CameraInfo& info) {
    const int bpp = bytes_per_pixel(image.encoding);
    if (cuapriltags_mode && opt.strict_cuapriltags_encodings && image.encoding != "rgb8" && image.encoding != "bgr8") {
      // the reference's cuAprilTags branch, text and all (src/apriltag_node.cpp:469-476)
      std::fprintf(stderr, "[apriltag_node] Unsupported image encoding: %s (only 'rgb8' or 'bgr8' supported)\n", image.encoding.c_str());
      throw std::runtime_error("cuAprilTags detector only supports 'rgb8' or 'bgr8' image input");
    }
    if (bpp == 0) {
      // (a superset of the reference's cuAprilTags branch, which takes rgb8 / bgr8 only: NodeOptions::strict_cuapriltags_encodings)
      std::fprintf(stderr, "[apriltag_node] Unsupported image encoding: %s (supported: 'mono8', 'rgb8', 'bgr8', 'rgba8', 'bgra8'%s)\n",
                   image.encoding.c_str(), cuapriltags_mode ? "; the reference's cuAprilTags mode takes 'rgb8' / 'bgr8' only" : "");
      throw std::runtime_error("AprilTags detector only supports 'mono8', 'rgb8', 'bgr8', 'rgba8' or 'bgra8' image input");
    }
    // the detector and the conversion buffer are sized from camera_info at initialisation
    // (src/apriltag_node.cpp:228-231,257-260): a frame of another size is dropped before anything is written
    if (image.width != width || image.height != height || info.width != width || info.height != height ||
        static_cast<size_t>(image.step) < static_cast<size_t>(image.width) * bpp || image.data == nullptr) {
      std::fprintf(stderr, "[apriltag_node] image %ux%u (step %u) does not match the initialised size %ux%u: frame dropped\n",
                   image.width, image.height, image.step, width, height);
      return;
    }
    const uint8_t* dev_src = image.data;
    if (!image.is_device) {
      const size_t bytes = static_cast<size_t>(image.step) * image.height;
      if (bytes > d_input_bytes) {
        if (d_input) api().dev_free(d_input);
        d_input = nullptr;
        if (api().dev_alloc(&d_input, bytes) != 0) throw std::runtime_error("device allocation failed");
        d_input_bytes = bytes;
      }
      // enqueue-only, on the stream the detection runs on (image.data stays valid until the detection below has returned)
      if (api().copy_to_device_async(d_input, image.data, bytes, stream) != 0) {
        std::fprintf(stderr, "[apriltag_node] host-to-device copy failed\n");
        return;
      }
      dev_src = static_cast<const uint8_t*>(d_input);
    }
    // The frame goes to the detector in the encoding it arrived in -- the reference hands cuAprilTags its rgb8 / bgr8 uchar3 image
    // (src/apriltag_node.cpp:469-486), its VPI branch converts first (:275-282): here the threshold pass of the one call reads the
    // interleaved frame itself, there is no conversion launch (and no second device buffer) in between.
    amdAprilTagsImageInput_t input;
    input.width = image.width;
    input.height = image.height;
    input.dev_ptr = dev_src;
    input.pitch = image.step;
    uint32_t num_detections = 0;
    std::vector<amdAprilTagsID_t> tags(static_cast<size_t>(opt.max_tags));
    const int enc = api().encoding_from_name(image.encoding.c_str());
    const int error = api().detect_color(detector, &input, static_cast<amdAprilTagsEncoding>(enc), tags.data(), &num_detections,
                                         static_cast<uint32_t>(opt.max_tags), stream);
    if (error != 0) {
      // the reference logs and drops the frame (src/apriltag_node.cpp:494-497)
      std::fprintf(stderr, "[apriltag_node] Failed to run AprilTags detector (error code %d)\n", error);
      return;
    }
    AprilTagDetectionArray msg;
    std::vector<TransformStamped> tfs;
    assemble_messages(tags.data(), num_detections, info.header, opt.tag_family, &msg, &tfs);
    if (on_detections) on_detections(msg);
    if (on_transforms) on_transforms(tfs);
  }
};

AprilTagNode::AprilTagNode(const NodeOptions& options) : impl_(new Impl()) {
  impl_->opt = options;
  // Backend selection (src/apriltag_node.cpp:575-582): this build has exactly one implementation, the
  // HIP detector; CPU / PVA backends of VPI do not exist here and support no family.
  // cuAprilTags when exactly CUDA was asked for, VPI otherwise; the HIP detector stands behind both, whatever
  // names the list holds (there is one implementation and no CPU fallback).
  impl_->family_enum = validate_family(options, &impl_->cuapriltags_mode);
}

AprilTagNode::~AprilTagNode() {
  if (impl_) {
    if (impl_->detector) api().destroy(impl_->detector);
    if (impl_->stream) api().stream_destroy(impl_->stream);
    if (impl_->d_input) api().dev_free(impl_->d_input);
  }
}

void AprilTagNode::set_detections_callback(DetectionsCallback cb) { impl_->on_detections = std::move(cb); }
void AprilTagNode::set_transforms_callback(TransformsCallback cb) { impl_->on_transforms = std::move(cb); }
const NodeOptions& AprilTagNode::options() const { return impl_->opt; }
bool AprilTagNode::initialized() const { return impl_->initialized; }

bool AprilTagNode::CameraImageCallback(const Image& image, const CameraInfo& camera_info) {
  if (image.header.stamp.sec != camera_info.header.stamp.sec || image.header.stamp.nanosec != camera_info.header.stamp.nanosec)
    return false;  // ExactTime synchroniser would not fire
  if (!impl_->initialized) impl_->Initialize(image, camera_info);
  impl_->OnCameraFrame(image, camera_info);
  return true;
}

// ---- AprilTagMultiCameraNode: S streams, one submission per round -----------------------------------------------------
struct AprilTagMultiCameraNode::Impl {
  NodeOptions opt;
  uint32_t S = 0;
  DetectionsCallback on_detections;
  TransformsCallback on_transforms;
  bool cuapriltags_mode = false, auto_flush = true, initialized = false;
  int family_enum = -1;
  amdAprilTagsHandle detector = nullptr;
  uint32_t width = 0, height = 0;
  uint8_t* d_mono = nullptr;       // S mono8 slots
  size_t pitch = 0, slot_bytes = 0;
  void* d_input = nullptr;         // staging for host / colour frames
  size_t d_input_bytes = 0;
  struct Slot { bool pending = false; Header info_header; std::array<double, 9> k{}; };
  std::vector<Slot> slots;


  void Initialize(const CameraInfo& info) {
    if (opt.max_tags <= 0) throw std::runtime_error("'max_tags' must be positive");
    // left over from an attempt that threw after the handle existed (e.g. the slot allocation failed)
    if (detector) { api().destroy(detector); detector = nullptr; }
    if (d_mono) { api().dev_free(d_mono); d_mono = nullptr; }
    amdAprilTagsConfig_t cfg;
    api().default_config(&cfg, info.width, info.height);
    cfg.tile_size = opt.tile_size;
    cfg.decimate = opt.decimate;
    cfg.num_families = 1;
    cfg.families[0] = static_cast<amdAprilTagsFamily>(family_enum);
    cfg.intrinsics.fx = static_cast<float>(info.k[0]);
    cfg.intrinsics.fy = static_cast<float>(info.k[4]);
    cfg.intrinsics.cx = static_cast<float>(info.k[2]);
    cfg.intrinsics.cy = static_cast<float>(info.k[5]);
    // (VPI mode passes every camera's own skew K[1], src/apriltag_node.cpp:215-225: set per frame at every flush)
    cfg.skew = cuapriltags_mode ? 0.0f : static_cast<float>(info.k[1]);
    cfg.tag_size = static_cast<float>(opt.size);
    cfg.max_batch = S;
    const int error = api().create_ex(&detector, &cfg);
    if (error != 0) throw std::runtime_error("Failed to create AprilTags detector (error code " + std::to_string(error) + ")");
    width = info.width;
    height = info.height;
    pitch = (static_cast<size_t>(width) + 63) & ~static_cast<size_t>(63);
    slot_bytes = pitch * height;
    void* p = nullptr;
    if (api().dev_alloc(&p, slot_bytes * S) != 0) throw std::runtime_error("device allocation failed");
    d_mono = static_cast<uint8_t*>(p);
    initialized = true;
  }

  // the frame of `stream` ends up as mono8 in its device slot
  bool Stage(uint32_t stream, const Image& image, const CameraInfo& info) {
    const int bpp = bytes_per_pixel(image.encoding);
    if (bpp == 0) {
      std::fprintf(stderr, "[apriltag_node] Unsupported image encoding: %s\n", image.encoding.c_str());
      throw std::runtime_error("AprilTags detector only supports 'mono8', 'rgb8', 'bgr8', 'rgba8' or 'bgra8' image input");
    }
    if (image.width != width || image.height != height || info.width != width || info.height != height ||
        static_cast<size_t>(image.step) < static_cast<size_t>(image.width) * bpp || image.data == nullptr) {
      std::fprintf(stderr, "[apriltag_node] stream %u: image %ux%u (step %u) does not match the initialised size %ux%u: frame dropped\n",
                   stream, image.width, image.height, image.step, width, height);
      return false;
    }
    const uint8_t* dev_src = image.data;
    uint8_t* slot = d_mono + slot_bytes * stream;
    if (!image.is_device) {
      const size_t bytes = static_cast<size_t>(image.step) * image.height;
      if (bpp == 1 && image.step == pitch) {   // straight into the slot
        if (api().copy_to_device(slot, image.data, bytes, nullptr) != 0) return false;
        return true;
      }
      if (bytes > d_input_bytes) {
        if (d_input) api().dev_free(d_input);
        d_input = nullptr;
        if (api().dev_alloc(&d_input, bytes) != 0) throw std::runtime_error("device allocation failed");
        d_input_bytes = bytes;
      }
      if (api().copy_to_device(d_input, image.data, bytes, nullptr) != 0) return false;
      dev_src = static_cast<const uint8_t*>(d_input);
    }
    // mono8 with another pitch is repacked by the conversion entry point as well (one 8-bit channel in, one out)
    return api().to_mono8(dev_src, image.step, image.encoding.c_str(), image.width, image.height, slot, pitch, nullptr) == 0;
  }
};

AprilTagMultiCameraNode::AprilTagMultiCameraNode(const NodeOptions& options, uint32_t num_streams) : impl_(new Impl()) {
  if (num_streams == 0 || num_streams > 65535) throw std::runtime_error("'num_streams' must be 1..65535");
  impl_->opt = options;
  impl_->S = num_streams;
  impl_->slots.resize(num_streams);
  impl_->family_enum = validate_family(options, &impl_->cuapriltags_mode);
}

AprilTagMultiCameraNode::~AprilTagMultiCameraNode() {
  if (impl_) {
    if (impl_->detector) api().destroy(impl_->detector);
    if (impl_->d_input) api().dev_free(impl_->d_input);
    if (impl_->d_mono) api().dev_free(impl_->d_mono);
  }
}

void AprilTagMultiCameraNode::set_detections_callback(DetectionsCallback cb) { impl_->on_detections = std::move(cb); }
void AprilTagMultiCameraNode::set_transforms_callback(TransformsCallback cb) { impl_->on_transforms = std::move(cb); }
void AprilTagMultiCameraNode::set_auto_flush(bool on) { impl_->auto_flush = on; }
uint32_t AprilTagMultiCameraNode::num_streams() const { return impl_->S; }
const NodeOptions& AprilTagMultiCameraNode::options() const { return impl_->opt; }

bool AprilTagMultiCameraNode::CameraImageCallback(uint32_t stream, const Image& image, const CameraInfo& camera_info) {
  if (stream >= impl_->S) throw std::runtime_error("stream index out of range");
  if (image.header.stamp.sec != camera_info.header.stamp.sec || image.header.stamp.nanosec != camera_info.header.stamp.nanosec)
    return false;  // ExactTime synchroniser would not fire
  if (!impl_->initialized) impl_->Initialize(camera_info);
  Impl::Slot& sl = impl_->slots[stream];
  // the slot's device image is about to be overwritten: a frame staged earlier and not yet submitted is gone either way,
  // and a failed staging must not leave it pending under its old header
  sl.pending = false;
  if (!impl_->Stage(stream, image, camera_info)) return false;
  sl.pending = true;
  sl.info_header = camera_info.header;
  sl.k = camera_info.k;
  if (impl_->auto_flush) {
    bool all = true;
    for (const auto& x : impl_->slots) all &= x.pending;
    if (all) Flush();
  }
  return true;
}

uint32_t AprilTagMultiCameraNode::Flush() {
  Impl& I = *impl_;
  std::vector<uint32_t> who;
  for (uint32_t s = 0; s < I.S; s++) if (I.slots[s].pending) who.push_back(s);
  if (who.empty() || !I.initialized) return 0;
  const uint32_t n = static_cast<uint32_t>(who.size());
  std::vector<amdAprilTagsImageInput_t> imgs(n);
  std::vector<amdAprilTagsCameraIntrinsics_t> intr(n);
  for (uint32_t i = 0; i < n; i++) {
    const Impl::Slot& sl = I.slots[who[i]];
    imgs[i].width = I.width; imgs[i].height = I.height; imgs[i].dev_ptr = I.d_mono + I.slot_bytes * who[i]; imgs[i].pitch = I.pitch;
    // K of the stream's own CameraInfo, double -> float as the reference does (src/apriltag_node.cpp:442-447)
    intr[i].fx = static_cast<float>(sl.k[0]); intr[i].fy = static_cast<float>(sl.k[4]);
    intr[i].cx = static_cast<float>(sl.k[2]); intr[i].cy = static_cast<float>(sl.k[5]);
  }
  const uint32_t max_tags = static_cast<uint32_t>(I.opt.max_tags);
  std::vector<amdAprilTagsID_t> tags(static_cast<size_t>(n) * max_tags);
  std::vector<uint32_t> counts(n, 0);
  if (!I.cuapriltags_mode) {   // every stream's own K[1], as S independent VPI-mode nodes would pass it
    std::vector<float> skews(n);
    for (uint32_t i = 0; i < n; i++) skews[i] = static_cast<float>(I.slots[who[i]].k[1]);
    if (api().set_frame_skews(I.detector, n, skews.data()) != 0) {   // (a round whose skews were refused must not run with stale ones)
      std::fprintf(stderr, "[apriltag_node] per-stream skews refused: round dropped\n");
      for (uint32_t s : who) I.slots[s].pending = false;
      return 0;
    }
  }
  const int error = api().detect_batch(I.detector, n, imgs.data(), intr.data(), tags.data(), counts.data(), max_tags, nullptr);
  for (uint32_t s : who) I.slots[s].pending = false;
  if (error != 0) {
    // the reference logs and drops the frame (src/apriltag_node.cpp:494-497); here: the round
    std::fprintf(stderr, "[apriltag_node] Failed to run AprilTags detector (error code %d)\n", error);
    return 0;
  }
  for (uint32_t i = 0; i < n; i++) {
    AprilTagDetectionArray msg;
    std::vector<TransformStamped> tfs;
    assemble_messages(tags.data() + static_cast<size_t>(i) * max_tags, counts[i], I.slots[who[i]].info_header, I.opt.tag_family, &msg, &tfs);
    if (I.on_detections) I.on_detections(who[i], msg);
    if (I.on_transforms) I.on_transforms(who[i], tfs);
  }
  return n;
}

}  // namespace apriltag
}  // namespace isaac_ros
}  // namespace amd

// ---- flat C view of the shell for the Python test harness (tests/test_node_shell_*.py) ---------------
using amd::isaac_ros::apriltag::AprilTagDetectionArray;
using amd::isaac_ros::apriltag::AprilTagNode;
using amd::isaac_ros::apriltag::CameraInfo;
using amd::isaac_ros::apriltag::Image;
using amd::isaac_ros::apriltag::NodeOptions;
using amd::isaac_ros::apriltag::TransformStamped;

struct NodeShellHarness {
  std::unique_ptr<AprilTagNode> node;
  AprilTagDetectionArray last;
  std::vector<TransformStamped> last_tf;
  int publishes = 0;
};

struct NodeShellDetection {
  int32_t id;
  char family[32];
  double center[2];
  double corners[4][2];
  double position[3];
  double orientation_xyzw[4];
  char child_frame_id[48];
};

extern "C" {

// Returns nullptr and fills err on a constructor exception (mirrors test/apriltag_node_test.cpp).
NodeShellHarness* node_shell_create(int max_tags, double size, int tile_size, const char* tag_family, const char* backends,
                                    int decimate, char* err, size_t err_len) {
  try {
    NodeOptions o;
    o.max_tags = max_tags; o.size = size; o.tile_size = static_cast<uint16_t>(tile_size);
    o.tag_family = tag_family; o.backends = backends; o.decimate = static_cast<uint32_t>(decimate);
    auto* h = new NodeShellHarness();
    h->node.reset(new AprilTagNode(o));
    h->node->set_detections_callback([h](const AprilTagDetectionArray& m) { h->last = m; h->publishes++; });
    h->node->set_transforms_callback([h](const std::vector<TransformStamped>& t) { h->last_tf = t; });
    return h;
  } catch (const std::exception& e) {
    if (err && err_len) { std::strncpy(err, e.what(), err_len - 1); err[err_len - 1] = 0; }
    return nullptr;
  }
}

// the same with NodeOptions::strict_cuapriltags_encodings set
NodeShellHarness* node_shell_create_strict(int max_tags, double size, int tile_size, const char* tag_family, const char* backends,
                                           int decimate, char* err, size_t err_len) {
  try {
    NodeOptions o;
    o.max_tags = max_tags; o.size = size; o.tile_size = static_cast<uint16_t>(tile_size);
    o.tag_family = tag_family; o.backends = backends; o.decimate = static_cast<uint32_t>(decimate);
    o.strict_cuapriltags_encodings = true;
    auto* h = new NodeShellHarness();
    h->node.reset(new AprilTagNode(o));
    h->node->set_detections_callback([h](const AprilTagDetectionArray& m) { h->last = m; h->publishes++; });
    h->node->set_transforms_callback([h](const std::vector<TransformStamped>& t) { h->last_tf = t; });
    return h;
  } catch (const std::exception& e) {
    if (err && err_len) { std::strncpy(err, e.what(), err_len - 1); err[err_len - 1] = 0; }
    return nullptr;
  }
}

void node_shell_destroy(NodeShellHarness* h) { delete h; }

// Feeds one image + camera_info pair.  Returns the number of detections published, -1 if the stamps
// differ (no callback), -2 on an exception (message in err).
int node_shell_on_frame(NodeShellHarness* h, const uint8_t* data, int is_device, const char* encoding, uint32_t width,
                        uint32_t height, uint32_t step, const double* k9, const char* frame_id, int32_t sec, uint32_t nanosec,
                        int32_t info_sec, uint32_t info_nanosec, NodeShellDetection* out, int max_out, char* out_frame_id,
                        size_t frame_id_len, char* err, size_t err_len) {
  try {
    Image img;
    img.header.frame_id = "image_frame"; img.header.stamp.sec = sec; img.header.stamp.nanosec = nanosec;
    img.width = width; img.height = height; img.step = step; img.encoding = encoding; img.data = data; img.is_device = is_device != 0;
    CameraInfo info;
    info.header.frame_id = frame_id; info.header.stamp.sec = info_sec; info.header.stamp.nanosec = info_nanosec;
    info.width = width; info.height = height;
    for (int i = 0; i < 9; i++) info.k[i] = k9[i];
    const int before = h->publishes;
    if (!h->node->CameraImageCallback(img, info)) return -1;
    if (h->publishes == before) return 0;  // frame dropped
    const auto& m = h->last;
    if (out_frame_id && frame_id_len) { std::strncpy(out_frame_id, m.header.frame_id.c_str(), frame_id_len - 1); out_frame_id[frame_id_len - 1] = 0; }
    int n = static_cast<int>(m.detections.size());
    for (int i = 0; i < n && i < max_out; i++) {
      const auto& d = m.detections[i];
      NodeShellDetection& o = out[i];
      std::memset(&o, 0, sizeof(o));
      o.id = d.id;
      std::strncpy(o.family, d.family.c_str(), sizeof(o.family) - 1);
      o.center[0] = d.center.x; o.center[1] = d.center.y;
      for (int c = 0; c < 4; c++) { o.corners[c][0] = d.corners[c].x; o.corners[c][1] = d.corners[c].y; }
      o.position[0] = d.pose.pose.pose.position.x; o.position[1] = d.pose.pose.pose.position.y; o.position[2] = d.pose.pose.pose.position.z;
      o.orientation_xyzw[0] = d.pose.pose.pose.orientation.x; o.orientation_xyzw[1] = d.pose.pose.pose.orientation.y;
      o.orientation_xyzw[2] = d.pose.pose.pose.orientation.z; o.orientation_xyzw[3] = d.pose.pose.pose.orientation.w;
      std::strncpy(o.child_frame_id, h->last_tf[i].child_frame_id.c_str(), sizeof(o.child_frame_id) - 1);
    }
    return n;
  } catch (const std::exception& e) {
    if (err && err_len) { std::strncpy(err, e.what(), err_len - 1); err[err_len - 1] = 0; }
    return -2;
  }
}

// ---- the multi-camera node through the same flat view ----
struct MultiShellHarness {
  std::unique_ptr<amd::isaac_ros::apriltag::AprilTagMultiCameraNode> node;
  std::vector<AprilTagDetectionArray> last;
  std::vector<std::vector<TransformStamped>> last_tf;
  std::vector<int> publishes;
};

MultiShellHarness* node_shell_multi_create(int num_streams, int max_tags, double size, int tile_size, const char* tag_family,
                                           const char* backends, int decimate, int auto_flush, char* err, size_t err_len) {
  try {
    NodeOptions o;
    o.max_tags = max_tags; o.size = size; o.tile_size = static_cast<uint16_t>(tile_size);
    o.tag_family = tag_family; o.backends = backends; o.decimate = static_cast<uint32_t>(decimate);
    auto* h = new MultiShellHarness();
    h->node.reset(new amd::isaac_ros::apriltag::AprilTagMultiCameraNode(o, static_cast<uint32_t>(num_streams)));
    h->node->set_auto_flush(auto_flush != 0);
    h->last.resize(num_streams); h->last_tf.resize(num_streams); h->publishes.assign(num_streams, 0);
    h->node->set_detections_callback([h](uint32_t s, const AprilTagDetectionArray& m) { h->last[s] = m; h->publishes[s]++; });
    h->node->set_transforms_callback([h](uint32_t s, const std::vector<TransformStamped>& t) { h->last_tf[s] = t; });
    return h;
  } catch (const std::exception& e) {
    if (err && err_len) { std::strncpy(err, e.what(), err_len - 1); err[err_len - 1] = 0; }
    return nullptr;
  }
}

void node_shell_multi_destroy(MultiShellHarness* h) { delete h; }

// 1 staged, 0 not (stamps differ / dropped), -2 exception
int node_shell_multi_on_frame(MultiShellHarness* h, int stream, const uint8_t* data, int is_device, const char* encoding, uint32_t width,
                              uint32_t height, uint32_t step, const double* k9, const char* frame_id, int32_t sec, uint32_t nanosec,
                              int32_t info_sec, uint32_t info_nanosec, char* err, size_t err_len) {
  try {
    Image img;
    img.header.frame_id = "image_frame"; img.header.stamp.sec = sec; img.header.stamp.nanosec = nanosec;
    img.width = width; img.height = height; img.step = step; img.encoding = encoding; img.data = data; img.is_device = is_device != 0;
    CameraInfo info;
    info.header.frame_id = frame_id; info.header.stamp.sec = info_sec; info.header.stamp.nanosec = info_nanosec;
    info.width = width; info.height = height;
    for (int i = 0; i < 9; i++) info.k[i] = k9[i];
    return h->node->CameraImageCallback(static_cast<uint32_t>(stream), img, info) ? 1 : 0;
  } catch (const std::exception& e) {
    if (err && err_len) { std::strncpy(err, e.what(), err_len - 1); err[err_len - 1] = 0; }
    return -2;
  }
}

int node_shell_multi_flush(MultiShellHarness* h) { return static_cast<int>(h->node->Flush()); }
int node_shell_multi_publishes(MultiShellHarness* h, int stream) { return h->publishes[stream]; }

// the last message published for `stream`: number of detections (records in out), header frame / stamp through the outputs
int node_shell_multi_last(MultiShellHarness* h, int stream, NodeShellDetection* out, int max_out, char* out_frame_id, size_t frame_id_len,
                          int32_t* sec, uint32_t* nanosec) {
  const auto& m = h->last[stream];
  if (out_frame_id && frame_id_len) { std::strncpy(out_frame_id, m.header.frame_id.c_str(), frame_id_len - 1); out_frame_id[frame_id_len - 1] = 0; }
  if (sec) *sec = m.header.stamp.sec;
  if (nanosec) *nanosec = m.header.stamp.nanosec;
  const int n = static_cast<int>(m.detections.size());
  for (int i = 0; i < n && i < max_out; i++) {
    const auto& d = m.detections[i];
    NodeShellDetection& o = out[i];
    std::memset(&o, 0, sizeof(o));
    o.id = d.id;
    std::strncpy(o.family, d.family.c_str(), sizeof(o.family) - 1);
    o.center[0] = d.center.x; o.center[1] = d.center.y;
    for (int c = 0; c < 4; c++) { o.corners[c][0] = d.corners[c].x; o.corners[c][1] = d.corners[c].y; }
    o.position[0] = d.pose.pose.pose.position.x; o.position[1] = d.pose.pose.pose.position.y; o.position[2] = d.pose.pose.pose.position.z;
    o.orientation_xyzw[0] = d.pose.pose.pose.orientation.x; o.orientation_xyzw[1] = d.pose.pose.pose.orientation.y;
    o.orientation_xyzw[2] = d.pose.pose.pose.orientation.z; o.orientation_xyzw[3] = d.pose.pose.pose.orientation.w;
    std::strncpy(o.child_frame_id, h->last_tf[stream][i].child_frame_id.c_str(), sizeof(o.child_frame_id) - 1);
  }
  return n;
}

}  // extern "C"
