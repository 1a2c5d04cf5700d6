// stand-in: see tests/ros_stubs/README.md
#pragma once
#include <cstdint>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <vector>
namespace XmlRpc {
class XmlRpcValue {
   public:
    enum Type { TypeInvalid, TypeBoolean, TypeInt, TypeDouble, TypeString };
    Type getType() const { return TypeInvalid; }
    operator bool&();
    operator int&();
    operator double&();
    operator std::string&();
};
}  // namespace XmlRpc
namespace ros {
struct Duration {
    explicit Duration(double) {}
    double toSec() const { return 0; }
};
struct Time {
    Time() {}
    explicit Time(double) {}
    double toSec() const { return 0; }
    static Time now();
};
struct WallDuration {
    double toSec() const { return 0; }
};
struct WallTime {
    static WallTime now();
    WallDuration operator-(const WallTime&) const { return WallDuration(); }
};
class Publisher {
   public:
    template <class M>
    void publish(const M&) const {}
};
class Subscriber {};
class NodeHandle {
   public:
    template <class M>
    Publisher advertise(const std::string&, uint32_t) { return Publisher(); }
    template <class P, class T>
    Subscriber subscribe(const std::string&, uint32_t, void (T::*)(P), T*) { return Subscriber(); }
};
namespace param {
bool get(const std::string&, XmlRpc::XmlRpcValue&);
template <class T>
bool param(const std::string&, T&, const T&) { return true; }
}  // namespace param
void init(int&, char**, const std::string&);
void spin();
void shutdown();
}  // namespace ros
struct RosLogSink {
    template <class T>
    RosLogSink& operator<<(const T&) { return *this; }
};
#define ROS_INFO_STREAM(x) do { RosLogSink s_; s_ << x; } while (0)
#define ROS_WARN_STREAM(x) do { RosLogSink s_; s_ << x; } while (0)
#define ROS_ERROR_STREAM(x) do { RosLogSink s_; s_ << x; } while (0)
#define ROS_FATAL_STREAM(x) do { RosLogSink s_; s_ << x; } while (0)
#define ROS_ERROR_STREAM_THROTTLE(p, x) do { RosLogSink s_; s_ << x; } while (0)
#define ROS_ERROR_THROTTLE(p, ...) do { } while (0)
#define ROS_FATAL(...) do { } while (0)
