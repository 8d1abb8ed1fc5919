#pragma once
#include <type_traits>
#include "rclcpp/rclcpp.hpp"
// the real macro registers a factory that constructs the class from rclcpp::NodeOptions: check exactly that
#define RCLCPP_COMPONENTS_REGISTER_NODE(NodeClass) \
  static_assert(std::is_constructible<NodeClass, const rclcpp::NodeOptions &>::value && std::is_base_of<rclcpp::Node, NodeClass>::value, \
                "component class must derive from rclcpp::Node and be constructible from NodeOptions");
