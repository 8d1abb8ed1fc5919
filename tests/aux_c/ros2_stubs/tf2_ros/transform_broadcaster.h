#pragma once
#include <vector>
#include "geometry_msgs/msg/transform_stamped.hpp"
namespace tf2_ros {
class TransformBroadcaster {
public:
  template <class NodeT> explicit TransformBroadcaster(NodeT * node) { (void)node; }
  void sendTransform(const geometry_msgs::msg::TransformStamped &) {}
  void sendTransform(const std::vector<geometry_msgs::msg::TransformStamped> &) {}
};
}
