// stand-in: see tests/ros_stubs/README.md
#pragma once
#include <string>
namespace sensor_msgs {
namespace image_encodings {
const std::string MONO8 = "mono8", TYPE_8UC1 = "8UC1", RGB8 = "rgb8", BGR8 = "bgr8", RGBA8 = "rgba8", BGRA8 = "bgra8";
}
}  // namespace sensor_msgs
