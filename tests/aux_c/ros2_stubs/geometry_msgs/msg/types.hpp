#pragma once
#include <array>
#include <string>
#include "std_msgs/msg/header.hpp"
namespace geometry_msgs { namespace msg {
struct Point { double x = 0, y = 0, z = 0; };
struct Vector3 { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Pose { Point position; Quaternion orientation; };
struct PoseWithCovariance { Pose pose; std::array<double, 36> covariance{}; };
struct PoseWithCovarianceStamped { std_msgs::msg::Header header; PoseWithCovariance pose; };
struct Transform { Vector3 translation; Quaternion rotation; };
struct TransformStamped { std_msgs::msg::Header header; std::string child_frame_id; Transform transform; };
} }
