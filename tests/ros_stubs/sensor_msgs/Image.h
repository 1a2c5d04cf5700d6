// stand-in: see tests/ros_stubs/README.md
#pragma once
#include <memory>
#include <ros/ros.h>
namespace std_msgs {
struct Header {
    ros::Time stamp;
    std::string frame_id;
};
}  // namespace std_msgs
namespace sensor_msgs {
struct Image {
    std_msgs::Header header;
    uint32_t height = 0, width = 0, step = 0;
    std::string encoding;
    std::vector<uint8_t> data;
};
typedef std::shared_ptr<const Image> ImageConstPtr;
}  // namespace sensor_msgs
