// ekf_vio_amd/host/replay_main.cpp — ROS-free replay / smoke driver for the C++ host shim.
//
// Runs the synthetic closed loop of the reference's test/analyzeEKFSimulation.cpp:10-125
// (N landmarks, truth integrated with the filter's own motion model, perfect projections,
// R = 1e-5 I, every landmark measured) through ekfvio::TightlyCoupledEKF and prints the
// final odometry next to the ground truth, the checkSigma numbers and the step rate.
// Usage: ekfvio_replay [landmarks=256] [frames=300] [seed=0] [dt=0.0333333]
//        ekfvio_replay --print-config [params.yaml]   (no GPU work: the node's parameter file -> ekfvio_config as JSON)
//        ekfvio_replay --records <dir> [--out <dir>] [--params params.yaml] [--device n] [--insight]
//
// --records replays what the node's two subscriptions deliver (EKFVIO.cpp:69-81) and writes what its two publishers
// send (EKFVIO.cpp:444-518), with no ROS in between.  <dir>/records.txt holds one record per line, in arrival order:
//     image <stamp> <file> <K0> ... <K8>      8-bit binary PGM (P5), intrinsics row-major as in sensor_msgs/CameraInfo.K
//     imu   <stamp> <gx> <gy> <gz> <ax> <ay> <az>
// Every image record goes through EKFVIO::addFrame (frame ingest, process(dt), KLT, update, FAST replenishment:
// cfg.replenish = 1); after it one odometry record and one point-cloud record are appended to
//     <out>/odom.txt     stamp px py pz qw qx qy qz vx vy vz wx wy wz numeric_ok
//     <out>/points.txt   "cloud <stamp> <n>" followed by n lines "x y z intensity"
// (floats printed with %.9g: exact).  IMU records go to imu_callback, a logging stub in the reference (with imu_update: 1 in
// the parameter file they are queued and applied in stamp order in front of the next frame).  --insight also writes what
// publishInsight sends, <out>/insight_NNN.ppm.
#include <cctype>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>

#include "ekfvio.hpp"

namespace {
struct SplitMix64 {
    uint64_t s;
    uint64_t next() {
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double uniform(double lo, double hi) { return lo + (hi - lo) * ((next() >> 11) * (1.0 / 9007199254740992.0)); }
    double gaussian(double sigma) {
        double u1 = std::fmax(uniform(0, 1), 1e-300), u2 = uniform(0, 1);
        return sigma * std::sqrt(-2.0 * std::log(u1)) * std::cos(2.0 * M_PI * u2);
    }
};
struct V3 {
    double x, y, z;
};
struct Q {
    double w, x, y, z;
};
V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
V3 rot(Q q, V3 v) {
    V3 qv{q.x, q.y, q.z}, uv = cross(qv, v);
    uv = {2 * uv.x, 2 * uv.y, 2 * uv.z};
    V3 c = cross(qv, uv);
    return {v.x + q.w * uv.x + c.x, v.y + q.w * uv.y + c.y, v.z + q.w * uv.z + c.z};
}
Q mul(Q a, Q b) {
    return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
}  // namespace

static int print_config(const char* path) {
    try {
        const ekfvio::Params p = path ? ekfvio::Params::fromFile(path) : ekfvio::Params();
        const ekfvio_config& c = p.cfg;
        std::printf("{\"max_features\": %d, \"fast_threshold\": %d, \"fast_blur_sigma\": %.9g, \"inverse_image_scale\": %d, "
                    "\"kill_pad\": %d, \"klt_min_eigen\": %.9g, \"min_new_feature_dist\": %d, \"klt_max_pyramid_level\": %d, "
                    "\"klt_window_size\": %d, \"default_point_depth\": %.9g, \"default_point_depth_variance\": %.9g, "
                    "\"default_point_homogenous_variance\": %.9g, \"sample_based_uncertainty\": %d, \"node\": {",
                    c.max_features, c.fast_threshold, (double)c.fast_blur_sigma, c.inverse_image_scale, c.kill_pad,
                    (double)c.klt_min_eigen, c.min_new_feature_dist, c.klt_max_pyramid_level, c.klt_window_size,
                    (double)c.default_point_depth, (double)c.default_point_depth_variance,
                    (double)c.default_point_homogenous_variance, c.sample_based_uncertainty);
        bool first = true;
        for (const auto& e : p.node) {
            std::printf("%s\"%s\": \"%s\"", first ? "" : ", ", e.first.c_str(), e.second.c_str());
            first = false;
        }
        std::printf("}}\n");
        return 0;
    } catch (const ekfvio::Error& e) {
        std::fprintf(stderr, "error %d: %s\n", e.code, e.what());
        return 2;
    }
}


// 8-bit binary PGM ("P5", optional comments, maxval 255)
static bool read_pgm(const std::string& path, std::vector<uint8_t>& px, int& w, int& h) {
    std::ifstream in(path, std::ios::binary);
    if (!in) return false;
    std::string magic;
    in >> magic;
    if (magic != "P5") return false;
    int vals[3], got = 0;
    while (got < 3) {
        int c = in.peek();
        if (c == '#') {
            std::string skip;
            std::getline(in, skip);
        } else if (std::isspace(c)) {
            in.get();
        } else if (!(in >> vals[got++])) {
            return false;
        }
    }
    in.get();  // the single whitespace byte behind maxval
    w = vals[0];
    h = vals[1];
    if (w < 1 || h < 1 || vals[2] != 255) return false;
    px.resize((size_t)w * h);
    in.read(reinterpret_cast<char*>(px.data()), (std::streamsize)px.size());
    return (size_t)in.gcount() == px.size();
}

static int replay_records(const std::string& dir, const std::string& out_dir, const char* params_path, int device, bool insight) {
    try {
        ekfvio::Params p = params_path ? ekfvio::Params::fromFile(params_path) : ekfvio::Params();
        if (!params_path) p.cfg.inverse_image_scale = 1;  // no parameter file: frames are used as recorded
        p.cfg.replenish = 1;                              // addFrame runs replenishFeatures itself (EKFVIO.cpp:154, :172)
        std::ifstream rec(dir + "/records.txt");
        if (!rec) throw ekfvio::Error(EKFVIO_EINVAL, "cannot open " + dir + "/records.txt");
        // no image size anywhere: the device planes grow with the first frame that needs it (Frame::Frame takes any image)
        std::vector<uint8_t> px;
        ekfvio::EKFVIO node(p, device);
        FILE* fo = std::fopen((out_dir + "/odom.txt").c_str(), "w");
        FILE* fp = std::fopen((out_dir + "/points.txt").c_str(), "w");
        if (!fo || !fp) throw ekfvio::Error(EKFVIO_EINVAL, "cannot write into " + out_dir);
        std::string line;
        int images = 0, imus = 0, lineno = 0;
        const auto t0 = std::chrono::steady_clock::now();
        while (std::getline(rec, line)) {
            lineno++;
            std::istringstream ls(line);
            std::string kind;
            if (!(ls >> kind) || kind[0] == '#') continue;
            double stamp;
            if (!(ls >> stamp)) throw ekfvio::Error(EKFVIO_EINVAL, "records.txt line " + std::to_string(lineno) + ": no stamp");
            if (kind == "imu") {
                ekfvio::Vector3f g, a;
                if (!(ls >> g[0] >> g[1] >> g[2] >> a[0] >> a[1] >> a[2]))
                    throw ekfvio::Error(EKFVIO_EINVAL, "records.txt line " + std::to_string(lineno) + ": imu needs 6 numbers");
                node.imu_callback(stamp, g, a);
                imus++;
            } else if (kind == "image") {
                std::string file;
                ekfvio::Frame f;
                ls >> file;
                for (int i = 0; i < 9; i++)
                    if (!(ls >> f.K[i])) throw ekfvio::Error(EKFVIO_EINVAL, "records.txt line " + std::to_string(lineno) + ": image needs a file and K (9 numbers)");
                int fw, fh;
                if (!read_pgm(dir + "/" + file, px, fw, fh)) throw ekfvio::Error(EKFVIO_EINVAL, "cannot read " + file);
                f.img = px.data();
                f.cols = fw;
                f.rows = fh;
                f.step = fw;
                f.t = stamp;
                const bool ok = node.addFrame(f);
                const ekfvio::Odometry od = node.odometry();     // publishOdometry (EKFVIO.cpp:444-477)
                const ekfvio::PointCloud pc = node.points();     // publishPoints   (EKFVIO.cpp:479-518)
                std::fprintf(fo, "%.9f %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %d\n", od.stamp, od.position[0],
                             od.position[1], od.position[2], od.orientation_wxyz[0], od.orientation_wxyz[1], od.orientation_wxyz[2],
                             od.orientation_wxyz[3], od.linear[0], od.linear[1], od.linear[2], od.angular[0], od.angular[1],
                             od.angular[2], (int)ok);
                std::fprintf(fp, "cloud %.9f %zu\n", pc.stamp, pc.points.size());
                for (size_t i = 0; i < pc.points.size(); i++)
                    std::fprintf(fp, "%.9g %.9g %.9g %.9g\n", pc.points[i][0], pc.points[i][1], pc.points[i][2], pc.intensity[i]);
                if (insight) {  // publishInsight (EKFVIO.cpp:379-442) as a binary PPM (RGB order on disk)
                    const ekfvio::Insight in = node.insight();
                    char name[64];
                    std::snprintf(name, sizeof name, "/insight_%03d.ppm", images);
                    FILE* fi = std::fopen((out_dir + name).c_str(), "wb");
                    if (!fi) throw ekfvio::Error(EKFVIO_EINVAL, "cannot write into " + out_dir);
                    std::fprintf(fi, "P6\n%d %d\n255\n", in.width, in.height);
                    std::vector<uint8_t> rgb(in.bgr.size());
                    for (size_t i = 0; i + 2 < rgb.size(); i += 3) rgb[i] = in.bgr[i + 2], rgb[i + 1] = in.bgr[i + 1], rgb[i + 2] = in.bgr[i];
                    std::fwrite(rgb.data(), 1, rgb.size(), fi);
                    std::fclose(fi);
                }
                images++;
            } else {
                throw ekfvio::Error(EKFVIO_EINVAL, "records.txt line " + std::to_string(lineno) + ": unknown record kind " + kind);
            }
        }
        std::fclose(fo);
        std::fclose(fp);
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("records: %d images, %d imu; %d landmarks; %.1f frames/s (file reading included)\n", images, imus,
                    node.tc_ekf.numFeatures(), images / el);
        if (p.cfg.use_imu)
            std::printf("imu: %d records dropped (older than the filter's time), %zu still queued behind the last frame\n",
                        node.droppedImuRecords(), node.queuedImuRecords());
        return 0;
    } catch (const ekfvio::Error& e) {
        std::fprintf(stderr, "ekfvio error %d: %s\n", e.code, e.what());
        return 1;
    }
}

int main(int argc, char** argv) {
    if (argc > 1 && std::strcmp(argv[1], "--print-config") == 0) return print_config(argc > 2 ? argv[2] : nullptr);
    if (argc > 2 && std::strcmp(argv[1], "--records") == 0) {
        std::string dir = argv[2], out = argv[2];
        const char* params = nullptr;
        int device = 0;
        bool insight = false;
        for (int i = 3; i < argc; i += 2) {
            if (std::strcmp(argv[i], "--insight") == 0) {
                insight = true;
                i--;
                continue;
            }
            if (i + 1 >= argc) {
                std::fprintf(stderr, "option %s needs a value\n", argv[i]);
                return 2;
            }
            if (std::strcmp(argv[i], "--out") == 0) out = argv[i + 1];
            else if (std::strcmp(argv[i], "--params") == 0) params = argv[i + 1];
            else if (std::strcmp(argv[i], "--device") == 0) device = std::atoi(argv[i + 1]);
            else {
                std::fprintf(stderr, "unknown option %s\n", argv[i]);
                return 2;
            }
        }
        return replay_records(dir, out, params, device, insight);
    }
    const int N = argc > 1 ? std::atoi(argv[1]) : 256;
    const int frames = argc > 2 ? std::atoi(argv[2]) : 300;
    const uint64_t seed = argc > 3 ? std::strtoull(argv[3], nullptr, 10) : 0;
    const float dt = argc > 4 ? (float)std::atof(argv[4]) : (float)(1.0 / 30.0);
    try {
        ekfvio::TightlyCoupledEKF ekf(N, 0);
        SplitMix64 rng{seed};
        std::vector<V3> pts(N);
        std::vector<ekfvio::Vector2f> uv(N);
        for (int i = 0; i < N; i++) {
            double z = 0.5 + rng.gaussian(0.01);
            double x = rng.uniform(-1.5, 1.5) * z, y = rng.uniform(-1.5, 1.5) * z;
            pts[i] = {x, y, z};
            uv[i] = {(float)(x / z), (float)(y / z)};
        }
        ekf.addNewFeatures(uv);
        V3 pos{0, 0, 0}, vel{-0.1, 0, -0.1}, acc{0, 0, 0}, om{0, 0.1, 0};
        Q quat{1, 0, 0, 0};
        std::vector<ekfvio::Vector2f> z(N);
        std::vector<ekfvio::Matrix2f> R(N, ekfvio::Matrix2f{1e-5f, 0, 0, 1e-5f});
        std::vector<uint8_t> pass(N, 1);
        bool numeric_ok = true;
        auto t0 = std::chrono::steady_clock::now();
        for (int s = 0; s < frames; s++) {
            const double d = dt;
            V3 tr{d * vel.x + 0.5 * d * d * acc.x, d * vel.y + 0.5 * d * d * acc.y, d * vel.z + 0.5 * d * d * acc.z};
            V3 dp = rot(quat, tr);
            pos = {pos.x + dp.x, pos.y + dp.y, pos.z + dp.z};
            const double on = std::sqrt(om.x * om.x + om.y * om.y + om.z * om.z), th = d * on;
            Q dq = on < 1e-10 ? Q{1, 0, 0, 0} : Q{std::cos(th / 2), om.x / on * std::sin(th / 2), om.y / on * std::sin(th / 2),
                                                   om.z / on * std::sin(th / 2)};
            Q dqi{dq.w, -dq.x, -dq.y, -dq.z};
            vel = rot(dqi, {vel.x + d * acc.x, vel.y + d * acc.y, vel.z + d * acc.z});
            acc = rot(dqi, acc);
            quat = mul(quat, dq);
            Q qi{quat.w, -quat.x, -quat.y, -quat.z};
            for (int i = 0; i < N; i++) {
                V3 fp = rot(qi, {pts[i].x - pos.x, pts[i].y - pos.y, pts[i].z - pos.z});
                z[i] = {(float)(fp.x / fp.z), (float)(fp.y / fp.z)};
            }
            ekf.process(dt);
            numeric_ok &= ekf.updateWithFeaturePositions(z, R, pass);
        }
        auto b = ekf.base_mu();
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        float md = 0, ma = 0;
        ekf.checkSigma(&md, &ma);
        std::printf("landmarks %d frames %d steps_per_s %.1f numeric_ok %d\n", N, frames, frames / el, (int)numeric_ok);
        std::printf("est_pos %.6f %.6f %.6f truth_pos %.6f %.6f %.6f\n", b[0], b[1], b[2], pos.x, pos.y, pos.z);
        std::printf("est_vel %.6f %.6f %.6f truth_vel %.6f %.6f %.6f\n", b[7], b[8], b[9], vel.x, vel.y, vel.z);
        std::printf("est_quat %.6f %.6f %.6f %.6f truth_quat %.6f %.6f %.6f %.6f\n", b[3], b[4], b[5], b[6], quat.w, quat.x,
                    quat.y, quat.z);
        std::printf("min_diag %.3e max_asym %.3e\n", md, ma);
        const double perr = std::fmax(std::fabs(b[0] - pos.x), std::fmax(std::fabs(b[1] - pos.y), std::fabs(b[2] - pos.z)));
        return (frames >= 60 && perr > 0.02) || md < 0 ? 2 : 0;
    } catch (const ekfvio::Error& e) {
        std::fprintf(stderr, "ekfvio error %d: %s\n", e.code, e.what());
        return 1;
    }
}
