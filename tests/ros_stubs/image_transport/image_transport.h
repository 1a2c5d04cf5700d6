// stand-in: see tests/ros_stubs/README.md
#pragma once
#include <sensor_msgs/CameraInfo.h>
namespace image_transport {
class CameraSubscriber {};
class ImageTransport {
   public:
    explicit ImageTransport(const ros::NodeHandle&) {}
    template <class T>
    CameraSubscriber subscribeCamera(const std::string&, uint32_t,
                                     void (T::*)(const sensor_msgs::ImageConstPtr&, const sensor_msgs::CameraInfoConstPtr&), T*) {
        return CameraSubscriber();
    }
};
}  // namespace image_transport
