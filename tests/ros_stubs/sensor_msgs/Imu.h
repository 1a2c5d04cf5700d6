// stand-in: see tests/ros_stubs/README.md
#pragma once
#include <sensor_msgs/Image.h>
namespace geometry_msgs {
struct Vector3 { double x = 0, y = 0, z = 0; };
struct Point { double x = 0, y = 0, z = 0; };
struct Point32 { float x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
}  // namespace geometry_msgs
namespace sensor_msgs {
struct Imu {
    std_msgs::Header header;
    geometry_msgs::Vector3 angular_velocity, linear_acceleration;
};
typedef std::shared_ptr<const Imu> ImuConstPtr;
}  // namespace sensor_msgs
