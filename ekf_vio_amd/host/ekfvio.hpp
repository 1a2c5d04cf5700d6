// ekf_vio_amd/host/ekfvio.hpp — header-only C++ host shim over the C-ABI (include/ekfvio.h).
//
// Mirrors the three classes EKFVIO::addFrame touches in the reference, with the same member
// names and argument meaning, so that the reference's node (include/ekf_vio/EKFVIO.cpp) can
// swap its `TightlyCoupledEKF tc_ekf; KLTTracker tracker;` members (EKFVIO.h:70-72) for these
// and keep its call sites:
//   reference                                              here
//   tc_ekf.process(dt)                    EKFVIO.cpp:163   TightlyCoupledEKF::process
//   tc_ekf.previousFeaturePositionVector  EKFVIO.cpp:216   (kept on the device; getter provided)
//   tracker.findNewFeaturePositions(...)  EKFVIO.cpp:216   KLTTracker::findNewFeaturePositions
//   tc_ekf.updateWithFeaturePositions     EKFVIO.cpp:218   TightlyCoupledEKF::updateWithFeaturePositions
//   tc_ekf.addNewFeatures                 EKFVIO.cpp:308   TightlyCoupledEKF::addNewFeatures
// Eigen/OpenCV types are replaced by plain structs with the same memory layout
// (Eigen::Vector2f = 2 floats, Eigen::Matrix2f = 4 floats column-major, cv::Mat 8UC1 = pointer
// + step).  Errors the reference would ROS_ASSERT on surface as ekfvio::Error.
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <deque>
#include <fstream>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/ekfvio.h"

namespace ekfvio {

using Vector2f = std::array<float, 2>;
using Matrix2f = std::array<float, 4>;  // column-major
using Vector3f = std::array<float, 3>;

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

// EKFVIO::EKFVIO's parameter block (EKFVIO.cpp:19-67, defaults Params.h:18-125) without ROS: the node's private
// parameter names (without the "~") mapped onto ekfvio_config; node-level names (topics, frames, switches) are kept as
// strings for the node shim; parameters of subsystems the reference no longer calls (moba / sba / depth updates /
// variance boxes, all unused on the hot path) are accepted and ignored.  Unknown names are an error (a typo in a launch
// file should not pass silently).  Values arrive as text, as in a launch file or a `rosparam dump`.
struct Params {
    ekfvio_config cfg;
    std::map<std::string, std::string> node;  // odom_topic, camera_topic, base_frame, use_imu, publish_insight, ...

    Params() {
        ekfvio_default_config(&cfg);
        cfg.inverse_image_scale = 4;  // D_INVERSE_IMAGE_SCALE (Params.h:28); the C-ABI's own default is 1
        node = {{"publish_insight", "true"}, {"insight_topic", "invio/insight"}, {"insight_camera_info_topic", "invio/camera_info"},
                {"analyze_runtime", "true"}, {"odom_topic", "invio/odom"}, {"odom_frame", "invio_odom"},
                {"point_topic", "invio/points"}, {"camera_topic", "/camera/image_rect"}, {"base_frame", "base_link"},
                {"world_frame", "world"}, {"camera_frame", "camera"}, {"imu_topic", "imu/measurement"},
                {"imu_frame", "imu"}, {"use_imu", "true"}};
    }

    static double number(const std::string& key, const std::string& v) {
        char* end = nullptr;
        const double x = std::strtod(v.c_str(), &end);
        if (end == v.c_str() || *end != '\0' || !std::isfinite(x)) throw Error(EKFVIO_EINVAL, "parameter " + key + ": not a number: " + v);
        return x;
    }
    static int integer(const std::string& key, const std::string& v) {
        const double x = number(key, v);
        if (x != std::floor(x) || std::fabs(x) > 1e9) throw Error(EKFVIO_EINVAL, "parameter " + key + ": not an integer: " + v);
        return (int)x;
    }

    void set(std::string key, const std::string& value) {
        while (!key.empty() && (key[0] == '~' || key[0] == '/')) key.erase(0, 1);
        const size_t slash = key.rfind('/');  // "/ekf_vio/num_features" from a rosparam dump
        if (slash != std::string::npos) key = key.substr(slash + 1);
        if (key == "num_features") cfg.max_features = integer(key, value);
        else if (key == "fast_threshold") cfg.fast_threshold = integer(key, value);
        else if (key == "fast_blur_sigma") cfg.fast_blur_sigma = (float)number(key, value);
        else if (key == "inverse_image_scale") cfg.inverse_image_scale = integer(key, value);  // Frame.cpp:15-42; integral scales only
        else if (key == "kill_pad") cfg.kill_pad = integer(key, value);
        else if (key == "min_klt_eigen_val") cfg.klt_min_eigen = (float)number(key, value);
        else if (key == "min_new_feature_dist") cfg.min_new_feature_dist = (int)number(key, value);  // a double there, used as a radius
        else if (key == "max_pyramids") cfg.klt_max_pyramid_level = integer(key, value);
        else if (key == "klt_window_size") cfg.klt_window_size = integer(key, value);
        else if (key == "default_point_depth") cfg.default_point_depth = (float)number(key, value);
        else if (key == "default_point_depth_variance") cfg.default_point_depth_variance = (float)number(key, value);
        else if (key == "default_point_homogenous_variance") cfg.default_point_homogenous_variance = (float)number(key, value);
        else if (key == "imu_update") cfg.use_imu = (value == "true" || value == "1") ? 1 : 0;  // not a reference parameter: SURVEY 8(f) F4
        else if (key == "imu_gyro_variance") cfg.imu_gyro_variance = (float)number(key, value);
        else if (key == "imu_accel_variance") cfg.imu_accel_variance = (float)number(key, value);
        else if (key == "gravity_x") cfg.gravity[0] = (float)number(key, value);
        else if (key == "gravity_y") cfg.gravity[1] = (float)number(key, value);
        else if (key == "gravity_z") cfg.gravity[2] = (float)number(key, value);
        else if (key == "frame_buffer_size") {
            if (integer(key, value) != 2) throw Error(EKFVIO_EINVAL, "frame_buffer_size: the device keeps exactly two frames (Params.h:58)");
        } else if (node.count(key)) node[key] = value;
        else if (ignored().count(key)) (void)number(key, value);
        else throw Error(EKFVIO_EINVAL, "unknown parameter: " + key);
    }

    // every private parameter name the node accepts (EKFVIO.cpp:20-67): what a ROS front end asks the parameter server for
    static std::vector<std::string> names() {
        std::vector<std::string> out = {"num_features", "fast_threshold", "fast_blur_sigma", "inverse_image_scale", "kill_pad",
                                        "min_klt_eigen_val", "min_new_feature_dist", "max_pyramids", "klt_window_size",
                                        "default_point_depth", "default_point_depth_variance",
                                        "default_point_homogenous_variance", "frame_buffer_size", "imu_update",
                                        "imu_gyro_variance", "imu_accel_variance", "gravity_x", "gravity_y", "gravity_z"};
        for (const auto& e : Params().node) out.push_back(e.first);
        for (const auto& e : ignored()) out.push_back(e.first);
        return out;
    }

    static Params fromMap(const std::map<std::string, std::string>& kv) {
        Params p;
        for (const auto& e : kv) p.set(e.first, e.second);
        return p;
    }
    // "key: value" lines (the flat YAML a launch file's <rosparam> block or `rosparam dump` of the node produces);
    // '#' starts a comment, quotes around values are dropped.
    static Params fromFile(const std::string& path) {
        std::ifstream in(path);
        if (!in) throw Error(EKFVIO_EINVAL, "cannot open " + path);
        Params p;
        std::string line;
        while (std::getline(in, line)) {
            const size_t hash = line.find('#');
            if (hash != std::string::npos) line.erase(hash);
            const size_t colon = line.find(':');
            if (colon == std::string::npos) {
                if (line.find_first_not_of(" \t\r") != std::string::npos) throw Error(EKFVIO_EINVAL, "not a 'key: value' line: " + line);
                continue;
            }
            auto trim = [](std::string t) {
                const size_t a = t.find_first_not_of(" \t\r\"'"), b = t.find_last_not_of(" \t\r\"'");
                return a == std::string::npos ? std::string() : t.substr(a, b - a + 1);
            };
            p.set(trim(line.substr(0, colon)), trim(line.substr(colon + 1)));
        }
        return p;
    }

   private:
    static const std::map<std::string, int>& ignored() {
        static const std::map<std::string, int> names = {
            {"border_weight_exponent", 0}, {"start_feature_count", 0}, {"dangerous_mature_feature_count", 0},
            {"minimum_trackable_features", 0}, {"minimum_keyframe_count_for_optimization", 0},
            {"maximum_keyframe_count_for_optimization", 0}, {"depth_translation_ratio", 0}, {"max_depth_updates_per_frame", 0},
            {"maximum_reprojection_error", 0}, {"moba_candidate_variance", 0}, {"maximum_candidate_reprojection_error", 0},
            {"eps_moba", 0}, {"eps_sba", 0}, {"huber_width", 0}, {"minumum_depth_determinant", 0}, {"moba_max_iterations", 0},
            {"sba_max_iterations", 0}, {"max_point_z", 0}, {"min_point_z", 0}, {"min_variance_box_size", 0},
            {"max_variance_box_size", 0}};
        return names;
    }
};

// Frame.h:25-41 without OpenCV: 8-bit single-channel image + intrinsics + stamp.
struct Frame {
    const uint8_t* img = nullptr;
    int cols = 0, rows = 0, step = 0;
    std::array<float, 9> K{};  // row-major, as sensor_msgs/CameraInfo.K
    double t = 0;
};

class TightlyCoupledEKF {
   public:
    explicit TightlyCoupledEKF(int max_features = 100, int device = 0, const ekfvio_config* cfg = nullptr) {
        ekfvio_config c;
        if (cfg) c = *cfg;
        else ekfvio_default_config(&c);
        if (!cfg) c.max_features = max_features;
        max_features_ = c.max_features;
        int rc = ekfvio_create(&c, device, nullptr, &h_);
        if (rc != EKFVIO_OK) {
            std::string msg = h_ ? ekfvio_last_error(h_) : "ekfvio_create failed";
            if (h_) ekfvio_destroy(h_);
            h_ = nullptr;
            throw Error(rc, msg);
        }
    }
    ~TightlyCoupledEKF() {
        if (h_) ekfvio_destroy(h_);
    }
    TightlyCoupledEKF(const TightlyCoupledEKF&) = delete;
    TightlyCoupledEKF& operator=(const TightlyCoupledEKF&) = delete;

    void initializeBaseState() { chk(ekfvio_reset(h_)); }
    void addNewFeatures(const std::vector<Vector2f>& new_homogenous_features) {
        chk(ekfvio_add_features(h_, new_homogenous_features.empty() ? nullptr : new_homogenous_features[0].data(),
                                (int)new_homogenous_features.size()));
    }
    void process(float dt) { chk(ekfvio_process(h_, dt)); }
    // returns false when the factorisation met a non-positive pivot (reference: ROS_ERROR_COND, continues)
    bool updateWithFeaturePositions(const std::vector<Vector2f>& measured_positions,
                                    const std::vector<Matrix2f>& estimated_covariance, const std::vector<uint8_t>& pass) {
        if (measured_positions.size() != estimated_covariance.size() || pass.size() != estimated_covariance.size())
            throw Error(EKFVIO_EINVAL, "size mismatch (TightlyCoupledEKF.cpp:478)");
        int rc = ekfvio_update(h_, measured_positions.empty() ? nullptr : measured_positions[0].data(),
                               estimated_covariance.empty() ? nullptr : estimated_covariance[0].data(), pass.data(),
                               (int)pass.size());
        if (rc == EKFVIO_ENUMERIC) return false;
        chk(rc);
        return true;
    }
    std::vector<int> formFeatureMeasurementMap(const std::vector<uint8_t>& measured) const {
        std::vector<int> idx(2 * measured.size() + 1);
        int rows = 0;
        chk(ekfvio_measurement_map(h_, measured.data(), (int)measured.size(), idx.data(), &rows));
        idx.resize(rows);
        return idx;
    }
    std::vector<Vector2f> previousFeaturePositionVector() {
        std::vector<Vector2f> out(numFeatures());
        chk(ekfvio_get_features(h_, nullptr, out.empty() ? nullptr : out[0].data(), nullptr));
        return out;
    }
    Matrix2f getFeatureHomogenousCovariance(int index) {
        Matrix2f m;
        chk(ekfvio_get_feature_cov(h_, index, m.data()));
        return m;
    }
    // TightlyCoupledEKF.cpp:668-676
    void setFeatureHomogenousCovariance(int index, const Matrix2f& cov) { chk(ekfvio_set_feature_cov(h_, index, cov.data())); }
    // TightlyCoupledEKF.cpp:683-697 (K row-major 3x3 as in CameraInfo.K; the reference returns a 2x2 sparse matrix)
    static Matrix2f getMetric2PixelMap(const std::array<float, 9>& K) {
        Matrix2f J;
        ekfvio_metric2pixel_map(K.data(), J.data());
        return J;
    }
    static Matrix2f getPixel2MetricMap(const std::array<float, 9>& K) {
        Matrix2f J;
        ekfvio_pixel2metric_map(K.data(), J.data());
        return J;
    }
    float getFeatureDepthVariance(int index) {
        float v = 0;
        chk(ekfvio_get_depth_variance(h_, index, &v));
        return v;
    }
    // checkSigma (TightlyCoupledEKF.cpp:699-714): true iff diag >= 0 and |S_ij - S_ji| <= 1e-3
    bool checkSigma(float* min_diag = nullptr, float* max_asym = nullptr) {
        float a = 0, b = 0;
        chk(ekfvio_check_sigma(h_, &a, &b));
        if (min_diag) *min_diag = a;
        if (max_asym) *max_asym = b;
        return a >= 0 && b <= 1e-3f;
    }
    std::array<float, EKFVIO_BASE_STATE_SIZE> base_mu() {
        std::array<float, EKFVIO_BASE_STATE_SIZE> b;
        chk(ekfvio_get_base_mu(h_, b.data()));
        return b;
    }
    std::vector<Vector3f> featureMus() {
        std::vector<Vector3f> out(numFeatures());
        chk(ekfvio_get_features(h_, out.empty() ? nullptr : out[0].data(), nullptr, nullptr));
        return out;
    }
    // Feature::getDeleteFlag per landmark (set when the tracker lost it, TightlyCoupledEKF.cpp:528; read by publishInsight)
    std::vector<uint8_t> deleteFlags() {
        std::vector<uint8_t> out(numFeatures());
        chk(ekfvio_get_features(h_, nullptr, nullptr, out.empty() ? nullptr : out.data()));
        return out;
    }
    int numFeatures() const { return ekfvio_num_features(h_); }
    int maxFeatures() const { return max_features_; }
    int dim() const { return ekfvio_dim(h_); }
    ekfvio_filter* handle() { return h_; }

    void chk(int rc) const {
        if (rc != EKFVIO_OK) throw Error(rc, ekfvio_last_error(h_));
    }

   private:
    ekfvio_filter* h_ = nullptr;
    int max_features_ = 0;
};

class KLTTracker {
   public:
    explicit KLTTracker(TightlyCoupledEKF& ekf) : ekf_(ekf) {}
    // Uploads `cf`; the previously pushed frame becomes `lf` (the reference passes both by reference).
    void pushFrame(const Frame& cf) {
        ekf_.chk(ekfvio_klt_push_frame(ekf_.handle(), cf.img, cf.cols, cf.rows, cf.step, cf.K.data()));
    }
    void findNewFeaturePositions(std::vector<Vector2f>& measured_positions, std::vector<Matrix2f>& estimated_uncertainty,
                                 std::vector<uint8_t>& passed) {
        const int N = ekf_.numFeatures();
        measured_positions.resize(N);
        estimated_uncertainty.resize(N);
        passed.resize(N);
        ekf_.chk(ekfvio_klt_track(ekf_.handle(), N ? measured_positions[0].data() : nullptr,
                                  N ? estimated_uncertainty[0].data() : nullptr, passed.data()));
    }

    // KLTTracker::estimateUncertaintySampleBased (KLTTracker.cpp:111-175) for n points between the two frames pushed
    // last: pixel-space covariances (px^2).  mu_ref in the previous frame, mu in the current one.
    void estimateUncertaintySampleBased(const std::vector<Vector2f>& mu_ref, const std::vector<Vector2f>& mu,
                                        std::vector<Matrix2f>& cov) {
        if (mu_ref.size() != mu.size()) throw Error(EKFVIO_EINVAL, "estimateUncertaintySampleBased: size mismatch");
        cov.resize(mu.size());
        if (mu.empty()) return;
        ekf_.chk(ekfvio_klt_uncertainty_points(ekf_.handle(), mu_ref[0].data(), mu[0].data(), (int32_t)mu.size(), cov[0].data()));
    }

   private:
    TightlyCoupledEKF& ekf_;
};

// publishOdometry's payload (EKFVIO.cpp:444-477) and publishPoints' (EKFVIO.cpp:479-518) as plain records: what the
// node copies into nav_msgs/Odometry (+ the world -> odom tf) and sensor_msgs/PointCloud with its "intensity" channel.
struct Odometry {
    double stamp = 0;
    Vector3f position{};
    std::array<float, 4> orientation_wxyz{};
    Vector3f linear{}, angular{};
};
struct PointCloud {
    double stamp = 0;
    std::vector<Vector3f> points;   // camera frame: (u/rho, v/rho, 1/rho)
    std::vector<float> intensity;   // image byte at the landmark's pixel
};

// publishInsight's payload (EKFVIO.cpp:379-442): the RESIZED frame as BGR8 with a green 22-pixel square marker
// (cv::drawMarker(..., MARKER_SQUARE, 22, 1)) at the pixel of every landmark that is not flagged for deletion, and the
// CameraInfo numbers the reference fills in (its K/P entries come from linear indexing of a column-major Matrix3f:
// K is published transposed, P(0,2) and P(1,2) are the zero entries K(2,0), K(2,1); reproduced as they are).
struct Insight {
    double stamp = 0;
    int width = 0, height = 0;
    std::vector<uint8_t> bgr;  // height x width x 3
    std::array<double, 9> K{};
    std::array<double, 12> P{};
};

// The step sequence of EKFVIO::addFrame (EKFVIO.cpp:139-196) with the ROS plumbing removed.
class EKFVIO {
   public:
    explicit EKFVIO(int max_features = 100, int device = 0, const ekfvio_config* cfg = nullptr)
        : tc_ekf(max_features, device, cfg), tracker(tc_ekf), use_imu_(cfg ? cfg->use_imu != 0 : false),
          scale_(cfg && cfg->inverse_image_scale > 1 ? cfg->inverse_image_scale : 1) {}
    // the node's constructor (EKFVIO.cpp:19-67): parameters by the reference's names (see Params)
    explicit EKFVIO(const Params& p, int device = 0)
        : tc_ekf(p.cfg.max_features, device, &p.cfg), tracker(tc_ekf), params(p.node), use_imu_(p.cfg.use_imu != 0),
          scale_(p.cfg.inverse_image_scale > 1 ? p.cfg.inverse_image_scale : 1) {}
    TightlyCoupledEKF tc_ekf;
    KLTTracker tracker;
    std::map<std::string, std::string> params;  // node-level parameters (topics, frames, switches) for the ROS side

    // EKFVIO::replenishFeatures (EKFVIO.cpp:224-311) on the frame pushed last: FAST-9/16 + occupancy first fit +
    // addNewFeatures, all on the device.  Returns the number of landmarks added.  With cfg.replenish = 1
    // addFrame() does this itself at the two places the reference does (:154, :172).
    int replenishFeatures(std::vector<int32_t>* new_pixels_xy = nullptr) {
        int32_t added = 0;
        if (new_pixels_xy) new_pixels_xy->assign(2 * (size_t)tc_ekf.maxFeatures(), 0);
        tc_ekf.chk(ekfvio_replenish(tc_ekf.handle(), &added, new_pixels_xy ? new_pixels_xy->data() : nullptr));
        if (new_pixels_xy) new_pixels_xy->resize(2 * (size_t)added);
        return added;
    }

    // returns false on a numeric warning (see updateWithFeaturePositions)
    bool addFrame(const Frame& f) {
        drainImu(f.t);  // IMU records up to this frame's stamp, in stamp order, before the frame itself
        int rc = ekfvio_step_image(tc_ekf.handle(), f.t, f.img, f.cols, f.rows, f.step, f.K.data());
        // ekfvio_step_image moves the device clock as soon as process(dt) is enqueued; only EINVAL / ECAPACITY refuse the frame
        // before that.  The queue clock follows whenever the predict went out, whatever happened behind it.
        if (rc != EKFVIO_EINVAL && rc != EKFVIO_ECAPACITY) {
            last_stamp_ = f.t;
            if (!have_time_ || f.t > t_filter_) t_filter_ = f.t;
            have_time_ = true;
            for (int i = 0; i < 9; i++) last_K_[i] = f.K[i];
        }
        if (rc == EKFVIO_ENUMERIC) return false;
        tc_ekf.chk(rc);
        return true;
    }
    // what publishOdometry(cf) / publishPoints(cf) send after addFrame (EKFVIO.cpp:182-188)
    Odometry odometry() {
        Odometry o;
        o.stamp = last_stamp_;
        tc_ekf.chk(ekfvio_get_odometry(tc_ekf.handle(), o.position.data(), o.orientation_wxyz.data(), o.linear.data(), o.angular.data()));
        return o;
    }
    PointCloud points() {
        PointCloud c;
        c.stamp = last_stamp_;
        const int N = tc_ekf.numFeatures();
        c.points.resize(N);
        c.intensity.resize(N);
        tc_ekf.chk(ekfvio_get_points(tc_ekf.handle(), N ? c.points[0].data() : nullptr, N ? c.intensity.data() : nullptr));
        return c;
    }
    // what publishInsight(cf) sends (EKFVIO.cpp:379-442); needs a frame
    Insight insight() {
        if (!have_time_) throw Error(EKFVIO_ESTATE, "insight() before the first frame");
        Insight o;
        o.stamp = last_stamp_;
        int32_t w = 0, h = 0;
        tc_ekf.chk(ekfvio_klt_get_level(tc_ekf.handle(), 0, &w, &h, nullptr, nullptr));
        std::vector<uint8_t> grey((size_t)w * h);
        tc_ekf.chk(ekfvio_klt_get_level(tc_ekf.handle(), 0, &w, &h, grey.data(), nullptr));
        o.width = w;
        o.height = h;
        o.bgr.resize((size_t)w * h * 3);
        for (size_t i = 0; i < grey.size(); i++) o.bgr[3 * i] = o.bgr[3 * i + 1] = o.bgr[3 * i + 2] = grey[i];  // CV_GRAY2BGR
        // Frame::Frame's K (Frame.cpp:26-33), then Feature::getPixel with the K(2) / K(5) indexing quirk (Feature.h:60-66)
        const float fx = (float)((double)last_K_[0] / scale_), fy = (float)((double)last_K_[4] / scale_);
        const float cx = (float)((double)last_K_[2] / scale_), cy = (float)((double)last_K_[5] / scale_);
        const std::vector<Vector3f> mus = tc_ekf.featureMus();
        const std::vector<uint8_t> flags = tc_ekf.deleteFlags();
        auto put = [&](int x, int y) {
            if (x < 0 || y < 0 || x >= w || y >= h) return;
            uint8_t* p = &o.bgr[3 * ((size_t)y * w + x)];
            p[0] = 0, p[1] = 255, p[2] = 0;  // cv::Scalar(0, 255, 0) in BGR
        };
        for (size_t i = 0; i < mus.size(); i++) {
            if (flags[i]) continue;  // if(!e.flaggedForDeletion()) (:386)
            const int px = (int)std::nearbyint(mus[i][0] * fx), py = (int)std::nearbyint(mus[i][1] * fy);  // cv::Point(Point2f): cvRound
            const int r = 22 / 2;  // drawMarker MARKER_SQUARE: the square (x +- size/2, y +- size/2), thickness 1
            for (int d = -r; d <= r; d++) {
                put(px + d, py - r), put(px + d, py + r);
                put(px - r, py + d), put(px + r, py + d);
            }
        }
        // cinfo.K.at(i) = f.K(i), linear index into a column-major Matrix3f (:408-416): the transpose of K
        const double Kcm[9] = {fx, 0, 0, 0, fy, 0, cx, cy, 1.0};
        for (int i = 0; i < 9; i++) o.K[i] = Kcm[i];
        o.P[0] = Kcm[0], o.P[2] = Kcm[2], o.P[5] = Kcm[4], o.P[6] = Kcm[5], o.P[10] = 1.0;  // (:419-423)
        return o;
    }
    // EKFVIO::imu_callback (EKFVIO.cpp:113-115).  With cfg.use_imu = 0 (the reference's behaviour) nothing happens.  With
    // cfg.use_imu = 1 the record is queued and applied, in stamp order, in front of the next frame whose stamp is not
    // older: ROS delivers 200 Hz IMU messages stamped AFTER an image before that image arrives, and a filter that had
    // already moved to the IMU stamp would have to refuse the image (dt < 0).  A record older than the filter's time
    // (it arrived after the frame that should have followed it) is dropped and counted, never an error.
    void imu_callback(double stamp, const Vector3f& gyro, const Vector3f& accel) {
        if (!use_imu_) {
            tc_ekf.chk(ekfvio_imu(tc_ekf.handle(), stamp, gyro.data(), accel.data()));  // the no-op, kept at the boundary
            return;
        }
        if (have_time_ && stamp < t_filter_) {
            dropped_imu_++;
            return;
        }
        ImuRecord r{stamp, gyro, accel};
        auto it = imu_queue_.end();
        while (it != imu_queue_.begin() && (it - 1)->stamp > stamp) --it;  // stable: equal stamps keep arrival order
        imu_queue_.insert(it, r);
        if (imu_queue_.size() > 1000) {  // the reference's subscriber queue length (EKFVIO.cpp:80)
            imu_queue_.pop_front();
            dropped_imu_++;
        }
    }
    // applies the queued IMU records with stamp <= until (addFrame does this itself)
    void drainImu(double until) {
        while (!imu_queue_.empty() && imu_queue_.front().stamp <= until) {
            const ImuRecord r = imu_queue_.front();
            imu_queue_.pop_front();
            if (have_time_ && r.stamp < t_filter_) {
                dropped_imu_++;
                continue;
            }
            tc_ekf.chk(ekfvio_imu(tc_ekf.handle(), r.stamp, r.gyro.data(), r.accel.data()));
            t_filter_ = r.stamp;
            have_time_ = true;
        }
    }
    int droppedImuRecords() const { return dropped_imu_; }
    size_t queuedImuRecords() const { return imu_queue_.size(); }

   private:
    struct ImuRecord {
        double stamp;
        Vector3f gyro, accel;
    };
    double last_stamp_ = 0;
    double t_filter_ = 0;  // the stamp the device state stands at (last frame or IMU record applied)
    bool have_time_ = false;
    bool use_imu_ = false;
    int scale_ = 1;
    std::array<float, 9> last_K_{};
    std::deque<ImuRecord> imu_queue_;
    int dropped_imu_ = 0;
};

}  // namespace ekfvio
