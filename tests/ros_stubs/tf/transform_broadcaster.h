// stand-in: see tests/ros_stubs/README.md
#pragma once
#include <tf/transform_listener.h>
namespace tf {
class TransformBroadcaster {
   public:
    void sendTransform(const StampedTransform&) {}
};
}  // namespace tf
