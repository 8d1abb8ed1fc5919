#pragma once
#include <cstdint>
namespace message_filters { namespace sync_policies {
template <class M0, class M1> struct ExactTime {
  using Message0 = M0; using Message1 = M1;
  explicit ExactTime(uint32_t queue_size) : queue_size_(queue_size) {}
  uint32_t queue_size_;
};
} }
