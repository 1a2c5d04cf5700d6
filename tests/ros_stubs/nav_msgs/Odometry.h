// stand-in: see tests/ros_stubs/README.md
#pragma once
#include <sensor_msgs/Imu.h>
namespace nav_msgs {
struct Odometry {
    std_msgs::Header header;
    std::string child_frame_id;
    struct { struct { geometry_msgs::Point position; geometry_msgs::Quaternion orientation; } pose; } pose;
    struct { struct { geometry_msgs::Vector3 linear, angular; } twist; } twist;
};
}  // namespace nav_msgs
