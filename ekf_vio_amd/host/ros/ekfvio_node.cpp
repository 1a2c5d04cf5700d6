// ekf_vio_amd/host/ros/ekfvio_node.cpp -- the ROS1 front end of the reference node (src/ekfvio_node.cpp:14-21,
// include/ekf_vio/EKFVIO.cpp:10-137 and :379-518) over the MI355X backend: same private parameters, same subscriptions
// (image_transport camera topic, queue 10; IMU topic, queue 1000), same publications (nav_msgs/Odometry, sensor_msgs/
// PointCloud with an "intensity" channel, optional insight image + camera info, the world -> odom transform) and the same
// start-up wait for the base -> camera transform.  Everything between the callbacks and the publishers is
// ekfvio::EKFVIO::addFrame, i.e. the C-ABI of libekfvio_hip.so.
//
// Built only inside a catkin workspace (INTEGRATION.md section 5 has the CMake lines); this repository's own build and
// tests never compile it: neither the build container nor the GPU boxes have ROS.  Without <ros/ros.h> on the include
// path the translation unit is empty.
//
// Names: the reference's launch files still start `invio_node` from package `invio` (launch/simLaunch.launch:9) while
// its CMake builds `ekfvio_node` in package `ekf_vio` (CMakeLists.txt:2,77); the topics keep their `invio/...` defaults
// (Params.h:19-20,111-113).  This node is `ekfvio_node`, and the topic defaults are the reference's.
#if __has_include(<ros/ros.h>)
#include <image_transport/image_transport.h>
#include <nav_msgs/Odometry.h>
#include <ros/ros.h>
#include <sensor_msgs/CameraInfo.h>
#include <sensor_msgs/Image.h>
#include <sensor_msgs/Imu.h>
#include <sensor_msgs/PointCloud.h>
#include <sensor_msgs/image_encodings.h>
#include <tf/transform_broadcaster.h>
#include <tf/transform_listener.h>

#include <memory>
#include <sstream>

#include "../ekfvio.hpp"

namespace {

std::string to_text(XmlRpc::XmlRpcValue& v) {
    std::ostringstream os;
    os.precision(17);
    switch (v.getType()) {
        case XmlRpc::XmlRpcValue::TypeBoolean: os << (static_cast<bool>(v) ? "true" : "false"); break;
        case XmlRpc::XmlRpcValue::TypeInt: os << static_cast<int>(v); break;
        case XmlRpc::XmlRpcValue::TypeDouble: os << static_cast<double>(v); break;
        case XmlRpc::XmlRpcValue::TypeString: os << static_cast<std::string>(v); break;
        default: throw ekfvio::Error(EKFVIO_EINVAL, "parameter of unsupported type");
    }
    return os.str();
}

// 8-bit grey view of an incoming image.  The reference hands the image to OpenCV in whatever encoding arrives
// (EKFVIO.cpp:124); the device pyramid is single-channel, so colour is reduced with the BT.601 integer weights.
bool to_grey(const sensor_msgs::Image& img, std::vector<uint8_t>& grey, const uint8_t*& data, int& step) {
    namespace enc = sensor_msgs::image_encodings;
    if (img.encoding == enc::MONO8 || img.encoding == enc::TYPE_8UC1) {
        data = img.data.data();
        step = (int)img.step;
        return true;
    }
    const bool rgb = img.encoding == enc::RGB8, bgr = img.encoding == enc::BGR8;
    const bool rgba = img.encoding == enc::RGBA8, bgra = img.encoding == enc::BGRA8;
    if (!(rgb || bgr || rgba || bgra)) return false;
    const int ch = (rgb || bgr) ? 3 : 4;
    grey.resize((size_t)img.width * img.height);
    for (uint32_t y = 0; y < img.height; y++) {
        const uint8_t* row = img.data.data() + (size_t)y * img.step;
        for (uint32_t x = 0; x < img.width; x++) {
            const uint8_t* p = row + (size_t)x * ch;
            const int r = (rgb || rgba) ? p[0] : p[2], g = p[1], b = (rgb || rgba) ? p[2] : p[0];
            grey[(size_t)y * img.width + x] = (uint8_t)((77 * r + 150 * g + 29 * b + 128) >> 8);
        }
    }
    data = grey.data();
    step = (int)img.width;
    return true;
}

class EkfvioNode {
   public:
    EkfvioNode() : it_(nh_) {
        // the ~48 private parameters of EKFVIO::EKFVIO (EKFVIO.cpp:20-67) by the reference's names
        for (const std::string& name : ekfvio::Params::names()) {
            XmlRpc::XmlRpcValue v;
            if (ros::param::get("~" + name, v)) params_.set(name, to_text(v));
        }
        params_.cfg.replenish = 1;  // addFrame runs replenishFeatures itself (EKFVIO.cpp:154, :172)
        int device = 0;
        ros::param::param<int>("~hip_device", device, 0);
        vio_.reset(new ekfvio::EKFVIO(params_, device));
        const auto& p = params_.node;
        cam_sub_ = it_.subscribeCamera(p.at("camera_topic"), 10, &EkfvioNode::cameraCallback, this);
        publish_insight_ = p.at("publish_insight") == "true" || p.at("publish_insight") == "1";
        if (publish_insight_) {
            insight_pub_ = nh_.advertise<sensor_msgs::Image>(p.at("insight_topic"), 1);
            insight_cinfo_pub_ = nh_.advertise<sensor_msgs::CameraInfo>(p.at("insight_camera_info_topic"), 1);
        }
        if (p.at("use_imu") == "true" || p.at("use_imu") == "1")  // "if(USE_IMU)" (EKFVIO.cpp:79-81; default true, Params.h)
            imu_sub_ = nh_.subscribe(p.at("imu_topic"), 1000, &EkfvioNode::imuCallback, this);
        odom_pub_ = nh_.advertise<nav_msgs::Odometry>(p.at("odom_topic"), 1);
        points_pub_ = nh_.advertise<sensor_msgs::PointCloud>(p.at("point_topic"), 1);
    }

    // EKFVIO.cpp:87-107: wait up to 10 s for base -> camera, else shut down
    bool waitForCameraTransform() {
        const auto& p = params_.node;
        ROS_INFO_STREAM("waiting for the transform from " << p.at("base_frame") << " to " << p.at("camera_frame"));
        if (!tf_listener_.waitForTransform(p.at("base_frame"), p.at("camera_frame"), ros::Time(0), ros::Duration(10))) {
            ROS_FATAL("could not get the base -> camera transform");
            return false;
        }
        try {
            tf::StampedTransform st;
            tf_listener_.lookupTransform(p.at("base_frame"), p.at("camera_frame"), ros::Time(0), st);
            b2c_ = tf::Transform(st);
        } catch (tf::TransformException& e) {
            ROS_WARN_STREAM(e.what());
        }
        return true;
    }

   private:
    void imuCallback(const sensor_msgs::ImuConstPtr& msg) {
        const ekfvio::Vector3f gyro{(float)msg->angular_velocity.x, (float)msg->angular_velocity.y, (float)msg->angular_velocity.z};
        const ekfvio::Vector3f acc{(float)msg->linear_acceleration.x, (float)msg->linear_acceleration.y,
                                   (float)msg->linear_acceleration.z};
        // EKFVIO.cpp:113-115 (a logging stub there).  With imu_update the record is queued and applied in stamp order in
        // front of the next frame (ekfvio::EKFVIO::imu_callback); nothing it can throw may unwind through ros::spin
        try {
            vio_->imu_callback(msg->header.stamp.toSec(), gyro, acc);
        } catch (const ekfvio::Error& e) {
            ROS_ERROR_STREAM_THROTTLE(1.0, "ekfvio (imu): " << e.what());
        }
    }

    void cameraCallback(const sensor_msgs::ImageConstPtr& img, const sensor_msgs::CameraInfoConstPtr& cam) {
        const ros::WallTime start = ros::WallTime::now();
        ekfvio::Frame f;
        int step = 0;
        if (!to_grey(*img, grey_, f.img, step)) {
            ROS_ERROR_STREAM_THROTTLE(5.0, "unsupported image encoding " << img->encoding);
            return;
        }
        f.cols = (int)img->width;
        f.rows = (int)img->height;
        f.step = step;
        for (int i = 0; i < 9; i++) f.K[i] = (float)cam->K[i];
        f.t = img->header.stamp.toSec();
        try {
            if (!vio_->addFrame(f)) ROS_ERROR_THROTTLE(1.0, "innovation covariance not positive definite (NumericalIssue)");
        } catch (const ekfvio::Error& e) {
            ROS_ERROR_STREAM("ekfvio: " << e.what());
            return;
        }
        if (publish_insight_) publishInsight(img->header.stamp);
        publishOdometry(img->header.stamp);
        publishPoints(img->header.stamp);
        const double ms = (ros::WallTime::now() - start).toSec() * 1e3;
        dt_sum_ += ms;
        ROS_INFO_STREAM("average dt: " << dt_sum_ / ++dt_count_ << " this dt: " << ms);  // EKFVIO.cpp:132-136
    }

    // EKFVIO.cpp:444-477
    void publishOdometry(const ros::Time& stamp) {
        const auto& p = params_.node;
        const ekfvio::Odometry o = vio_->odometry();
        const tf::Transform pose(tf::Quaternion(o.orientation_wxyz[1], o.orientation_wxyz[2], o.orientation_wxyz[3], o.orientation_wxyz[0]),
                                 tf::Vector3(o.position[0], o.position[1], o.position[2]));
        br_.sendTransform(tf::StampedTransform(pose, stamp, p.at("world_frame"), p.at("odom_frame")));
        nav_msgs::Odometry msg;
        msg.header.stamp = stamp;
        msg.header.frame_id = p.at("world_frame");
        msg.child_frame_id = p.at("camera_frame");
        msg.pose.pose.position.x = o.position[0];
        msg.pose.pose.position.y = o.position[1];
        msg.pose.pose.position.z = o.position[2];
        msg.pose.pose.orientation.w = o.orientation_wxyz[0];
        msg.pose.pose.orientation.x = o.orientation_wxyz[1];
        msg.pose.pose.orientation.y = o.orientation_wxyz[2];
        msg.pose.pose.orientation.z = o.orientation_wxyz[3];
        msg.twist.twist.linear.x = o.linear[0];
        msg.twist.twist.linear.y = o.linear[1];
        msg.twist.twist.linear.z = o.linear[2];
        msg.twist.twist.angular.x = o.angular[0];
        msg.twist.twist.angular.y = o.angular[1];
        msg.twist.twist.angular.z = o.angular[2];
        odom_pub_.publish(msg);  // no covariance, as in the reference (:473)
    }

    // EKFVIO.cpp:479-518
    void publishPoints(const ros::Time& stamp) {
        const ekfvio::PointCloud pc = vio_->points();
        sensor_msgs::PointCloud msg;
        msg.header.stamp = stamp;
        msg.header.frame_id = params_.node.at("odom_frame");
        sensor_msgs::ChannelFloat32 ch;
        ch.name = "intensity";
        msg.points.resize(pc.points.size());
        for (size_t i = 0; i < pc.points.size(); i++) {
            msg.points[i].x = pc.points[i][0];
            msg.points[i].y = pc.points[i][1];
            msg.points[i].z = pc.points[i][2];
        }
        ch.values = pc.intensity;
        msg.channels.push_back(ch);
        points_pub_.publish(msg);
    }

    // EKFVIO.cpp:379-442: the resized frame as BGR8 with a green square marker at every landmark that is not flagged for
    // deletion, and a CameraInfo with the reference's K / P entries; frame id ODOM_FRAME on both (ekfvio::EKFVIO::insight)
    void publishInsight(const ros::Time& stamp) {
        const ekfvio::Insight in = vio_->insight();
        sensor_msgs::CameraInfo cinfo;
        cinfo.header.frame_id = params_.node.at("odom_frame");
        cinfo.header.stamp = stamp;
        cinfo.height = in.height;
        cinfo.width = in.width;
        for (int i = 0; i < 9; i++) cinfo.K[i] = in.K[i];
        for (int i = 0; i < 12; i++) cinfo.P[i] = in.P[i];
        insight_cinfo_pub_.publish(cinfo);
        sensor_msgs::Image out;
        out.header.frame_id = params_.node.at("odom_frame");
        out.header.stamp = stamp;
        out.width = in.width;
        out.height = in.height;
        out.encoding = sensor_msgs::image_encodings::BGR8;
        out.step = 3 * in.width;
        out.data = in.bgr;
        insight_pub_.publish(out);
    }

    ros::NodeHandle nh_;
    image_transport::ImageTransport it_;
    image_transport::CameraSubscriber cam_sub_;
    ros::Subscriber imu_sub_;
    ros::Publisher insight_pub_, insight_cinfo_pub_, odom_pub_, points_pub_;
    tf::TransformListener tf_listener_;
    tf::TransformBroadcaster br_;
    tf::Transform b2c_;
    ekfvio::Params params_;
    std::unique_ptr<ekfvio::EKFVIO> vio_;
    std::vector<uint8_t> grey_;
    bool publish_insight_ = false;
    double dt_sum_ = 0;
    int dt_count_ = 0;
};

}  // namespace

int main(int argc, char** argv) {
    ros::init(argc, argv, "ekfvio_node");
    try {
        EkfvioNode node;
        if (!node.waitForCameraTransform()) {
            ros::shutdown();
            return 1;
        }
        ros::spin();  // single-threaded spinner: callbacks are serialised, as in the reference (EKFVIO.cpp:109)
    } catch (const ekfvio::Error& e) {
        ROS_FATAL_STREAM("ekfvio: " << e.what());
        return 1;
    }
    return 0;
}
#endif  // __has_include(<ros/ros.h>)
