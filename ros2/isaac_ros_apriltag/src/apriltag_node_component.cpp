// apriltag_node_component.cpp -- rclcpp component with the reference's plugin identity
// (class nvidia::isaac_ros::apriltag::AprilTagNode, node name "apriltag_node", topics image / camera_info
// in, tag_detections + /tf out; reference src/apriltag_node.cpp:562-633, launch/isaac_ros_apriltag.launch.py:24-41)
// as a thin adapter over the ROS-free shell amd::isaac_ros::apriltag::AprilTagNode
// (include/apriltag_node_shell.hpp), which owns all of the node logic and talks to libapriltag_amd.so.
//
// NOT BUILT IN THIS REPOSITORY'S IMAGE: ROS 2 (rclcpp, message_filters, tf2_ros,
// isaac_ros_apriltag_interfaces) is absent there, so this file is compiled only by the colcon build
// described in ros2/README.md.  It takes sensor_msgs/Image (host memory); the NITROS zero-copy type
// adaptation of the reference is NVIDIA-proprietary and out of scope.
#include <memory>
#include <string>
#include <vector>

#include "apriltag_node_shell.hpp"
#include "isaac_ros_apriltag_interfaces/msg/april_tag_detection_array.hpp"
#include "message_filters/subscriber.h"
#include "message_filters/sync_policies/exact_time.h"
#include "message_filters/synchronizer.h"
#include "rclcpp/rclcpp.hpp"
#include "sensor_msgs/msg/camera_info.hpp"
#include "sensor_msgs/msg/image.hpp"
#include "tf2_ros/transform_broadcaster.h"

namespace nvidia
{
namespace isaac_ros
{
namespace apriltag
{

namespace shell = amd::isaac_ros::apriltag;

class AprilTagNode : public rclcpp::Node
{
public:
  explicit AprilTagNode(const rclcpp::NodeOptions & options)
  : rclcpp::Node("apriltag_node", options),
    camera_image_sync_{ExactPolicy{3}, image_sub_, camera_info_sub_},
    detections_pub_{create_publisher<isaac_ros_apriltag_interfaces::msg::AprilTagDetectionArray>(
        "tag_detections", rclcpp::QoS(1))}
  {
    shell::NodeOptions opt;
    opt.max_tags = declare_parameter<int>("max_tags", 64);
    opt.size = declare_parameter<double>("size", 0.22);
    opt.tile_size = declare_parameter<uint16_t>("tile_size", 4);
    opt.tag_family = declare_parameter<std::string>("tag_family", "tag36h11");
    opt.backends = declare_parameter<std::string>("backends", "CUDA");  // name or comma list; exactly "CUDA" = cuAprilTags mode
    opt.decimate = static_cast<uint32_t>(declare_parameter<int>("decimate", 1));
    // throws std::runtime_error("Tag family not supported by specified backend ...") like the reference
    impl_ = std::make_unique<shell::AprilTagNode>(opt);
    tf_broadcaster_ = std::make_unique<tf2_ros::TransformBroadcaster>(this);

    impl_->set_detections_callback(
      [this](const shell::AprilTagDetectionArray & in) {
        isaac_ros_apriltag_interfaces::msg::AprilTagDetectionArray msg;
        msg.header = last_info_header_;
        for (const auto & d : in.detections) {
          isaac_ros_apriltag_interfaces::msg::AprilTagDetection m;
          m.family = d.family;
          m.id = d.id;
          m.center.x = d.center.x;
          m.center.y = d.center.y;
          for (int i = 0; i < 4; i++) {
            m.corners.data()[i].x = d.corners[i].x;
            m.corners.data()[i].y = d.corners[i].y;
          }
          m.pose.pose.pose.position.x = d.pose.pose.pose.position.x;
          m.pose.pose.pose.position.y = d.pose.pose.pose.position.y;
          m.pose.pose.pose.position.z = d.pose.pose.pose.position.z;
          m.pose.pose.pose.orientation.x = d.pose.pose.pose.orientation.x;
          m.pose.pose.pose.orientation.y = d.pose.pose.pose.orientation.y;
          m.pose.pose.pose.orientation.z = d.pose.pose.pose.orientation.z;
          m.pose.pose.pose.orientation.w = d.pose.pose.pose.orientation.w;
          msg.detections.push_back(m);
        }
        detections_pub_->publish(msg);
      });
    impl_->set_transforms_callback(
      [this](const std::vector<shell::TransformStamped> & in) {
        std::vector<geometry_msgs::msg::TransformStamped> tfs;
        for (const auto & t : in) {
          geometry_msgs::msg::TransformStamped tf;
          tf.header = last_info_header_;
          tf.child_frame_id = t.child_frame_id;
          tf.transform.translation.x = t.transform.translation.x;
          tf.transform.translation.y = t.transform.translation.y;
          tf.transform.translation.z = t.transform.translation.z;
          tf.transform.rotation.x = t.transform.rotation.x;
          tf.transform.rotation.y = t.transform.rotation.y;
          tf.transform.rotation.z = t.transform.rotation.z;
          tf.transform.rotation.w = t.transform.rotation.w;
          tfs.push_back(tf);
        }
        tf_broadcaster_->sendTransform(tfs);
      });

    camera_image_sync_.registerCallback(
      std::bind(&AprilTagNode::CameraImageCallback, this, std::placeholders::_1, std::placeholders::_2));
    image_sub_.subscribe(this, "image");
    camera_info_sub_.subscribe(this, "camera_info");
  }

private:
  void CameraImageCallback(
    const sensor_msgs::msg::Image::ConstSharedPtr & image,
    const sensor_msgs::msg::CameraInfo::ConstSharedPtr & camera_info)
  {
    shell::Image img;
    img.header.frame_id = image->header.frame_id;
    img.header.stamp.sec = image->header.stamp.sec;
    img.header.stamp.nanosec = image->header.stamp.nanosec;
    img.width = image->width;
    img.height = image->height;
    img.encoding = image->encoding;
    img.step = image->step;
    img.data = image->data.data();
    img.is_device = false;
    shell::CameraInfo info;
    info.header.frame_id = camera_info->header.frame_id;
    info.header.stamp.sec = camera_info->header.stamp.sec;
    info.header.stamp.nanosec = camera_info->header.stamp.nanosec;
    info.width = camera_info->width;
    info.height = camera_info->height;
    for (int i = 0; i < 9; i++) {info.k[i] = camera_info->k[i];}
    last_info_header_ = camera_info->header;  // output headers = camera_info header (reference :501,:534)
    impl_->CameraImageCallback(img, info);
  }

  using ExactPolicy = message_filters::sync_policies::ExactTime<sensor_msgs::msg::Image, sensor_msgs::msg::CameraInfo>;
  message_filters::Subscriber<sensor_msgs::msg::Image> image_sub_;
  message_filters::Subscriber<sensor_msgs::msg::CameraInfo> camera_info_sub_;
  message_filters::Synchronizer<ExactPolicy> camera_image_sync_;
  rclcpp::Publisher<isaac_ros_apriltag_interfaces::msg::AprilTagDetectionArray>::SharedPtr detections_pub_;
  std::unique_ptr<tf2_ros::TransformBroadcaster> tf_broadcaster_;
  std::unique_ptr<shell::AprilTagNode> impl_;
  std_msgs::msg::Header last_info_header_;
};

}  // namespace apriltag
}  // namespace isaac_ros
}  // namespace nvidia

#include "rclcpp_components/register_node_macro.hpp"
RCLCPP_COMPONENTS_REGISTER_NODE(nvidia::isaac_ros::apriltag::AprilTagNode)
