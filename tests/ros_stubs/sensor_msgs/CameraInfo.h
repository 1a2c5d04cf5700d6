// stand-in: see tests/ros_stubs/README.md
#pragma once
#include <sensor_msgs/Image.h>
namespace sensor_msgs {
struct CameraInfo {
    std_msgs::Header header;
    uint32_t height, width;
    double K[9];
    double P[12];
    std::vector<double> D;
};
typedef std::shared_ptr<const CameraInfo> CameraInfoConstPtr;
}  // namespace sensor_msgs
