// apriltag_node_shell.hpp -- ROS-free C++ mirror of the reference node surface
// nvidia::isaac_ros::apriltag::AprilTagNode
//   (reference include/isaac_ros_apriltag/apriltag_node.hpp:48-91, src/apriltag_node.cpp:562-633),
// with the message types reduced to plain structs that carry the same field names the node reads and
// writes (sensor_msgs/Image, sensor_msgs/CameraInfo, isaac_ros_apriltag_interfaces/AprilTagDetection
// [Array], geometry_msgs/TransformStamped; fields as used at src/apriltag_node.cpp:324-363,500-546).
//
// ROS 2 is not present in the build image, so this shell keeps the node's logic -- parameters and their
// defaults, backend/family validation and its error text, lazy initialisation on the first frame from
// CameraInfo (K, not P), encoding check, message assembly, TF naming, error policy -- behind callbacks
// instead of rclcpp publishers.  An rclcpp component is a thin adapter over this class.
#pragma once
#include <array>
#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <vector>

namespace amd {
namespace isaac_ros {
namespace apriltag {

struct Time { int32_t sec = 0; uint32_t nanosec = 0; };
struct Header { Time stamp; std::string frame_id; };

// sensor_msgs/Image.  `data` may be host memory (is_device = false: copied to the GPU, as a plain
// sensor_msgs subscriber would) or a device pointer (is_device = true: the NITROS-handle case,
// src/apriltag_node.cpp:480-486).
struct Image {
  Header header;
  uint32_t height = 0, width = 0;
  std::string encoding;
  uint32_t step = 0;
  const uint8_t* data = nullptr;
  bool is_device = false;
};

struct CameraInfo {
  Header header;
  uint32_t height = 0, width = 0;
  std::array<double, 9> k{};  // the node reads K (k[0],k[4],k[2],k[5]), src/apriltag_node.cpp:442-446
};

struct Point { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Pose { Point position; Quaternion orientation; };
struct PoseWithCovariance { Pose pose; std::array<double, 36> covariance{}; };
struct PoseWithCovarianceStamped { Header header; PoseWithCovariance pose; };

struct AprilTagDetection {
  std::string family;
  int32_t id = 0;
  Point center;
  std::array<Point, 4> corners;
  PoseWithCovarianceStamped pose;
};
struct AprilTagDetectionArray { Header header; std::vector<AprilTagDetection> detections; };

struct Vector3 { double x = 0, y = 0, z = 0; };
struct Transform { Vector3 translation; Quaternion rotation; };
struct TransformStamped { Header header; std::string child_frame_id; Transform transform; };

// Parameters declared in the constructor of the reference node (src/apriltag_node.cpp:564-568).
struct NodeOptions {
  int max_tags = 64;
  double size = 0.22;
  uint16_t tile_size = 4;
  std::string tag_family = "tag36h11";
  std::string backends = "CUDA";  // reference default VPI_BACKEND_CUDA; "CUDA" | "HIP" | "GPU" select this library
  uint32_t decimate = 1;          // extension (AprilRobotics quad_decimate); 1 = cuAprilTags behaviour
  // The reference's cuAprilTags branch (backends == "CUDA") throws on every encoding but rgb8 / bgr8
  // (src/apriltag_node.cpp:469-476); its VPI branch takes the five of :76-82.  This shell takes the five in BOTH modes by default --
  // a superset: mono8 is what north_star feeds the detector, and the library's colour entry point reads rgba8 / bgra8 as well.
  // true: cuAprilTags mode refuses everything but rgb8 / bgr8 with the reference's own text.
  bool strict_cuapriltags_encodings = false;
};

class AprilTagNode {
 public:
  using DetectionsCallback = std::function<void(const AprilTagDetectionArray&)>;
  using TransformsCallback = std::function<void(const std::vector<TransformStamped>&)>;

  // Throws std::runtime_error whose what() contains
  // "Tag family not supported by specified backend" (src/apriltag_node.cpp:584-599).
  explicit AprilTagNode(const NodeOptions& options);
  ~AprilTagNode();

  // "tag_detections" publisher / TF broadcaster stand-ins (src/apriltag_node.cpp:548-549).
  void set_detections_callback(DetectionsCallback cb);
  void set_transforms_callback(TransformsCallback cb);

  // Synchronised image + camera_info (ExactTime policy upstream, include/.../apriltag_node.hpp:74-78):
  // returns false (and does nothing) unless both stamps are identical.
  bool CameraImageCallback(const Image& image, const CameraInfo& camera_info);

  const NodeOptions& options() const;
  bool initialized() const;

 private:
  struct Impl;
  std::unique_ptr<Impl> impl_;
};

// Batching front end: S camera streams of one image size on ONE GPU, one detector submission per round.
//
// The reference node -- and AprilTagNode above -- hands the detector one frame per call (src/apriltag_node.cpp:491-493);
// S cameras then mean S nodes and S one-frame submissions, each a chain of ~20 dependent kernels that leaves the GPU
// mostly idle (1 200 frames/s per call stream against ~10 000 batched).  This node keeps the reference's
// per-stream surface -- parameters, ExactTime pairing, the five encodings, lazy initialisation from the first
// CameraInfo, message assembly with the stream's own camera_info header, TF child "<family>:<id>" under the stream's
// camera frame (src/apriltag_node.cpp:499-549) -- but stages the latest frame of every stream on the device and submits
// all staged frames as ONE amdAprilTagsDetectBatch with per-frame intrinsics (each stream's own K).  Results are
// bit-identical to S independent AprilTagNode instances (tests/test_gpu_parity.py::test_multi_camera_node).
class AprilTagMultiCameraNode {
 public:
  using DetectionsCallback = std::function<void(uint32_t stream, const AprilTagDetectionArray&)>;
  using TransformsCallback = std::function<void(uint32_t stream, const std::vector<TransformStamped>&)>;

  // Same validation and error text as AprilTagNode.  All streams share the options (family, tag size, backends).
  AprilTagMultiCameraNode(const NodeOptions& options, uint32_t num_streams);
  ~AprilTagMultiCameraNode();
  void set_detections_callback(DetectionsCallback cb);
  void set_transforms_callback(TransformsCallback cb);

  // Stages one synchronised pair of `stream` (a newer pair replaces an unsubmitted older one: "latest frame").  Returns
  // false when the stamps differ (ExactTime would not fire) or the frame is dropped (size mismatch, as AprilTagNode).
  // With auto_flush (default) the round is submitted as soon as every stream has a staged frame.
  bool CameraImageCallback(uint32_t stream, const Image& image, const CameraInfo& camera_info);
  // Submits the staged frames of all streams that have one; publishes per stream; returns the number of streams served.
  uint32_t Flush();
  void set_auto_flush(bool on);

  uint32_t num_streams() const;
  const NodeOptions& options() const;

 private:
  struct Impl;
  std::unique_ptr<Impl> impl_;
};

}  // namespace apriltag
}  // namespace isaac_ros
}  // namespace amd
