#pragma once
#include <cstddef>
#include <memory>
#include <string>
namespace rclcpp {
class NodeOptions {};
class QoS { public: explicit QoS(size_t depth) : depth_(depth) {} size_t depth_; };
template <class MsgT> class Publisher {
public:
  using SharedPtr = std::shared_ptr<Publisher<MsgT>>;
  void publish(const MsgT &) {}
};
class Node {
public:
  Node(const std::string & name, const NodeOptions & options) { (void)name; (void)options; }
  virtual ~Node() = default;
  template <class T> T declare_parameter(const std::string & name, const T & default_value) { (void)name; return default_value; }
  template <class T> T declare_parameter(const std::string & name, const char * default_value) { (void)name; return T(default_value); }
  template <class MsgT> typename Publisher<MsgT>::SharedPtr create_publisher(const std::string & topic, const QoS & qos) {
    (void)topic; (void)qos; return std::make_shared<Publisher<MsgT>>();
  }
};
}  // namespace rclcpp
