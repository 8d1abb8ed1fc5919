#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>
#include "std_msgs/msg/header.hpp"
namespace sensor_msgs { namespace msg {
struct CameraInfo {
  std_msgs::msg::Header header;
  uint32_t height = 0, width = 0;
  std::string distortion_model;
  std::vector<double> d;
  std::array<double, 9> k{};
  std::array<double, 9> r{};
  std::array<double, 12> p{};
  using SharedPtr = std::shared_ptr<CameraInfo>;
  using ConstSharedPtr = std::shared_ptr<const CameraInfo>;
};
} }
