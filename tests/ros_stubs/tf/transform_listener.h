// stand-in: see tests/ros_stubs/README.md
#pragma once
#include <stdexcept>
#include <ros/ros.h>
namespace tf {
struct Quaternion {
    Quaternion(double, double, double, double) {}
};
struct Vector3 {
    Vector3(double, double, double) {}
};
struct Transform {
    Transform() {}
    Transform(const Quaternion&, const Vector3&) {}
    Transform inverse() const { return Transform(); }
};
struct StampedTransform : Transform {
    StampedTransform() {}
    StampedTransform(const Transform&, const ros::Time&, const std::string&, const std::string&) {}
};
struct TransformException : std::runtime_error {
    using std::runtime_error::runtime_error;
};
class TransformListener {
   public:
    bool waitForTransform(const std::string&, const std::string&, const ros::Time&, const ros::Duration&) { return true; }
    void lookupTransform(const std::string&, const std::string&, const ros::Time&, StampedTransform&) {}
};
}  // namespace tf
