#pragma once
#include <functional>
#include <memory>
#include "message_filters/subscriber.h"
namespace message_filters {
template <class Policy> class Synchronizer {
public:
  using M0 = typename Policy::Message0; using M1 = typename Policy::Message1;
  Synchronizer(const Policy & policy, Subscriber<M0> & f0, Subscriber<M1> & f1) { (void)policy; (void)f0; (void)f1; }
  template <class C> void registerCallback(const C & callback) {
    // the callback must be invocable with the two messages' const shared pointers
    cb_ = callback;
  }
private:
  std::function<void(const std::shared_ptr<const M0> &, const std::shared_ptr<const M1> &)> cb_;
};
}
