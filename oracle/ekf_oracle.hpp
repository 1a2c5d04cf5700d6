// oracle/ekf_oracle.hpp
//
// TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference filter's arithmetic
// (k-sheridan/ekf_vio, include/ekf_vio/TightlyCoupledEKF.cpp).  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this code; the
// product (ekf_vio_amd/csrc) never links, imports or calls it.
//
// PARITY STATUS: the reference cannot be built in this image (needs ROS, Eigen and
// OpenCV, none present), so this restatement is pinned by
//   * the reference's only known-answer test (H map, test/test_ekf.cpp:44-63),
//   * the initial-covariance / process-noise constants of the reference source,
//   * scenario known answers derived from the reference formulas
//     (test/test_ekf.cpp:154-204 inputs; expected values in tests/golden/),
//   * the checkSigma invariants (TightlyCoupledEKF.cpp:699-714) on the simulation
//     scenarios of test/analyzeEKFSimulation.cpp:233-244,
//   * an independent numpy fp64 restatement (oracle/np_oracle.py).
// Against the reference *binary* the numerics are "parity unpinned".
//
// Conventions restated from Eigen (the reference's only arithmetic dependency):
//   * Quaternion ctor order (w,x,y,z); q*v = v + w*uv + qv x uv with uv = 2(qv x v);
//     inverse = conjugate / squaredNorm; a*=b is the Hamilton product a.b.
//   * Sparse products accumulate res(i,j) += lhs(i,k)*rhs(k,j) in ascending k with a
//     separate multiply and add (x86-64 GCC without -mfma never contracts), so a dense
//     ascending-k loop without FMA contraction reproduces them (structural zeros add
//     exact zeros).  Build this file with -ffp-contract=off.
//   * prune(ref,eps)/sparseView(ref,eps) keep x iff |x| > |ref|*eps = 1e-8*1e-5; on a
//     dense store this is a flush-to-zero below 1e-13.
//   * Arithmetic the reference performs in double because of double literals or the C
//     `sin/cos/sqrt(double)` overloads is performed in double here and narrowed.
//
// Template parameter T = float reproduces the reference precision; T = double is the
// yardstick used to size parity tolerances.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "amd_order.hpp"

namespace oracle {

constexpr int BASE = 22;  // TightlyCoupledEKF.h:12 BASE_STATE_SIZE

struct Config {
    double default_point_depth = 0.5;                 // Params.h:83
    double default_point_depth_variance = 100.0;      // Params.h:84
    double default_point_homogenous_variance = 1e-5;  // Params.h:86
    int emulate_static_cache = 1;  // TightlyCoupledEKF.cpp:400-403 function-static dq_inv
    // --- which Eigen build the restatement follows (the reference pins neither Eigen's version nor its
    // vectorisation; CMakeLists.txt:9 asks for -msse..-mssse3 but :15 overwrites CMAKE_CXX_FLAGS, so a Release
    // build is plain x86-64 = SSE2, EIGEN_VECTORIZE_SSE defined, no SSE3 / FMA).  tests/test_oracle_variants_cpu.py
    // measures how far the variants are apart.
    // eigen_sse_quat = 1 (default): Geometry/arch/Geometry_SSE.h quat_product<Architecture::SSE, ., ., float> for
    //   `quat *= dq` (:362), and the 4-float squaredNorm() inside dq.inverse() (:357) / dq.normalize() (:347, :432)
    //   reduced by SSE2 predux: (c0+c2)+(c1+c3) over the coefficient order (x,y,z,w).  0 = Eigen's generic
    //   product and the unrolled scalar reduction (c0+c1)+(c2+c3) (EIGEN_DONT_VECTORIZE / a non-x86 build).
    int eigen_sse_quat = 1;
    // trig_float = 1: `sin(theta/2)` / `cos(theta/2)` (:352-354, :437-439) resolve to the float overloads
    //   (a <math.h> that pulls std::sin(float) into the global namespace, GCC >= 6 with the C header included
    //   directly); 0 (default): ::sin(double) on the promoted argument, narrowed on assignment (GCC 5, <cmath> only).
    int trig_float = 0;
    // div_reciprocal = 1: `vector /= scalar` (:198, :214, :243, :277, :293, :309 and normalize()) multiplies by
    //   Scalar(1)/scalar (Eigen <= 3.2.x SelfCwiseBinaryOp.h); 0 (default): a true division (Eigen >= 3.2.90).
    int div_reciprocal = 0;
    // ldlt_amd_order = 1 (default): SimplicialLDLT's default ordering (AMDOrdering, TightlyCoupledEKF.cpp:577: the solver is declared
    //   without an ordering argument) -- the matrix factored is S^T(P, P) with P from oracle/amd_order.hpp on the structural pattern
    //   of S^T's lower triangle (a dense-array restatement has no structure of its own: an entry counts as structural when it is
    //   non-zero -- Sigma is pruned at :117 / :625, so its exact zeros ARE its structural zeros -- or lies in one of R's 2 x 2
    //   blocks, whose four entries are inserted whatever their value, :513-519).  For a numerically dense S with m > ~100 rows every
    //   node is "dense" to AMD and P is the identity; below that, and for the block-diagonal S of an update straight from the
    //   diagonal prior (test/test_ekf.cpp:66-141), a real minimum-degree permutation runs.  0: natural order (rounds 1-5).
    int ldlt_amd_order = 1;
    // amd_keep_diagonal = 1 (default): the pattern AMD sees includes the diagonal (Eigen's Ordering.h has the `prune(keep_diag())`
    //   that would drop it commented out); 0: textbook cs_amd.  Moves the "every node is dense" threshold between m >= 101 and 103
    //   and, below it, nothing but tie-breaks.
    int amd_keep_diagonal = 1;
    int ldlt_general_path = 0;  // test switch: the sparse-form factorisation loop even where the dense loop applies (same bits, asserted)
};

// the three arithmetic choices above, passed down to the free functions
struct Arith {
    int sse_quat = 1, trig_float = 0, div_reciprocal = 0;
};
static inline Arith arith_of(const Config& c) { return Arith{c.eigen_sse_quat, c.trig_float, c.div_reciprocal}; }

template <class T>
struct Quat {
    T w, x, y, z;
};

template <class T>
struct Vec3 {
    T x, y, z;
};

template <class T>
static inline Vec3<T> cross(const Vec3<T>& a, const Vec3<T>& b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// Eigen QuaternionBase::_transformVector
template <class T>
static inline Vec3<T> rotate(const Quat<T>& q, const Vec3<T>& v) {
    Vec3<T> qv{q.x, q.y, q.z};
    Vec3<T> uv = cross(qv, v);
    uv = {uv.x + uv.x, uv.y + uv.y, uv.z + uv.z};
    Vec3<T> c = cross(qv, uv);
    return {v.x + q.w * uv.x + c.x, v.y + q.w * uv.y + c.y, v.z + q.w * uv.z + c.z};
}

// Vector4f::squaredNorm() of the coefficients (x,y,z,w).  Vectorised (SSE2, no SSE3): one packet of squares,
// predux = a + movehl(a) then lane 0 + lane 1 = (x2+z2)+(y2+w2).  Scalar: redux_novec_unroller halves the range,
// (x2+y2)+(z2+w2).
template <class T>
static inline T squared_norm4(const Quat<T>& q, const Arith& ar) {
    if (ar.sse_quat) return (q.x * q.x + q.z * q.z) + (q.y * q.y + q.w * q.w);
    return (q.x * q.x + q.y * q.y) + (q.z * q.z + q.w * q.w);
}

// Eigen QuaternionBase::inverse(): conjugate().coeffs() / squaredNorm()  (operator/ is a true division in every
// Eigen 3.x; the SSE conjugate is a sign-bit xor, exact)
template <class T>
static inline Quat<T> inverse(const Quat<T>& q, const Arith& ar) {
    T n2 = squared_norm4(q, ar);
    if (n2 > T(0)) return {q.w / n2, -q.x / n2, -q.y / n2, -q.z / n2};
    return {T(0), T(0), T(0), T(0)};
}

// coeffs().normalize(): `*this /= norm()` (3.2) / `z = squaredNorm(); if (z > 0) *this /= sqrt(z)` (3.3)
template <class T>
static inline Quat<T> normalized(const Quat<T>& q, const Arith& ar) {
    T n = std::sqrt(squared_norm4(q, ar));
    if (ar.div_reciprocal) {
        T r = T(1) / n;
        return {q.w * r, q.x * r, q.y * r, q.z * r};
    }
    return {q.w / n, q.x / n, q.y / n, q.z / n};
}

// a * b.  sse_quat: Geometry_SSE.h (3.2 and 3.3 round identically: 3.2 flips the sign of the two w products before
// adding them, 3.3 after):  res = (a * b.wwww - a.zxyx * b.yzxx) + (+,+,+,-)(a.yzxz * b.zxyz + a.wwwy * b.xyzy).
// Otherwise Eigen's generic quat_product (Hamilton, left to right).
template <class T>
static inline Quat<T> qmul(const Quat<T>& a, const Quat<T>& b, const Arith& ar) {
    if (ar.sse_quat) {
        Quat<T> r;
        r.x = (a.x * b.w - a.z * b.y) + (a.y * b.z + a.w * b.x);
        r.y = (a.y * b.w - a.x * b.z) + (a.z * b.x + a.w * b.y);
        r.z = (a.z * b.w - a.y * b.x) + (a.x * b.y + a.w * b.z);
        r.w = (a.w * b.w - a.x * b.x) + (-(a.z * b.z + a.y * b.y));
        return r;
    }
    return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
            a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
            a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}

static inline float sin_as(float x, int trig_float) { return trig_float ? sinf(x) : (float)std::sin((double)x); }
static inline float cos_as(float x, int trig_float) { return trig_float ? cosf(x) : (float)std::cos((double)x); }
static inline double sin_as(double x, int) { return std::sin(x); }
static inline double cos_as(double x, int) { return std::cos(x); }

// `v /= s` on an Eigen vector (the finite differences' `/= 2*DELTA_SHIFT`)
template <class T>
static inline T vec_div(T x, T s, const Arith& ar) {
    return ar.div_reciprocal ? x * (T(1) / s) : x / s;
}

// Vector3f::norm(): Eigen's unrolled reduction for 3 coefficients is c0 + (c1 + c2)
template <class T>
static inline T norm3(const Vec3<T>& v) {
    return std::sqrt(v.x * v.x + (v.y * v.y + v.z * v.z));
}

// "dt*vel + 0.5*dt*dt*accel": the 0.5*dt*dt factor is a double expression narrowed to
// the matrix scalar type (TightlyCoupledEKF.cpp:338, :420)
template <class T>
static inline Vec3<T> translation(const Vec3<T>& vel, const Vec3<T>& acc, T dt) {
    T h = (T)(0.5 * (double)dt * (double)dt);
    return {dt * vel.x + h * acc.x, dt * vel.y + h * acc.y, dt * vel.z + h * acc.z};
}

// dq = exp(omega*dt) as the reference builds it (TightlyCoupledEKF.cpp:340-355); sign=-1
// gives the convolveFeature variant (:427-440) which negates the vector part.
template <class T>
static inline Quat<T> delta_quat(const Vec3<T>& omega, T dt, T sign, const Arith& ar) {
    T on = norm3(omega);
    if (on < (T)1e-10) {
        Quat<T> q{T(1), sign * omega.x * dt, sign * omega.y * dt, sign * omega.z * dt};
        return normalized(q, ar);
    }
    T theta = dt * on;
    Vec3<T> oh{omega.x / on, omega.y / on, omega.z / on};
    T half = theta / 2;
    T st2 = sin_as(half, ar.trig_float);
    T ct2 = cos_as(half, ar.trig_float);
    return {ct2, sign * oh.x * st2, sign * oh.y * st2, sign * oh.z * st2};
}

template <class T>
struct Filter {
    Config cfg;
    T base_mu[BASE];
    std::vector<T> feat_mu;        // 3 per landmark: [u, v, 1/depth]   (Feature.h:41)
    std::vector<T> last_klt;       // 2 per landmark (Feature.h:43)
    std::vector<uint8_t> del_flag; // Feature.h:46
    std::vector<T> Sigma;          // dense column-major n x n
    int n = BASE;
    int last_update_info = 0;      // bit 0: the LDLT met a non-positive pivot (what the HIP path reports as EKFVIO_ENUMERIC);
                                   // bit 1: a pivot was EXACTLY zero -- the only case in which Eigen's simplicial LDLT sets
                                   // NumericalIssue (SimplicialCholesky_impl.h: `if(d == RealScalar(0)) { ok = false; break; }`),
                                   // i.e. the only case in which the reference's ROS_ERROR_COND at :579 fires.  (Eigen then abandons
                                   // the factorisation and the solve at :580 reads D entries it never wrote: undefined upstream.
                                   // Here the elimination carries on through the division by zero: non-finite, deterministic.)
    std::vector<int> last_perm;    // the ordering the last update's LDLT used (P[k] = measurement row of the k-th pivot)

    // emulation of the function-static cache in convolveFeature (:400-403)
    T cache_om[3] = {0, 0, 0};
    Quat<T> cache_dq_inv{T(1), T(0), T(0), T(0)};

    explicit Filter(const Config& c) : cfg(c) { initialize_base_state(); }

    int num_features() const { return (int)(feat_mu.size() / 3); }
    T& S(int i, int j) { return Sigma[(size_t)j * n + i]; }

    // TightlyCoupledEKF.cpp:10-56
    void initialize_base_state() {
        n = BASE;
        Sigma.assign((size_t)n * n, T(0));
        for (int i = 0; i < BASE; i++) base_mu[i] = T(0);
        base_mu[3] = T(1);
        for (int i = 7; i <= 15; i++) S(i, i) = T(30);
        for (int i = 16; i <= 21; i++) S(i, i) = T(0.5);
        feat_mu.clear();
        last_klt.clear();
        del_flag.clear();
    }

    // TightlyCoupledEKF.cpp:58-94 (+ Feature.cpp:14-20)
    void add_new_features(const T* uv, int k) {
        if (k <= 0) return;
        int n_new = n + 3 * k;
        std::vector<T> Snew((size_t)n_new * n_new, T(0));
        for (int j = 0; j < n; j++)
            std::memcpy(&Snew[(size_t)j * n_new], &Sigma[(size_t)j * n], sizeof(T) * n);
        int idx = n;
        Sigma.swap(Snew);
        n = n_new;
        T avg_depth = (T)cfg.default_point_depth;  // float average_scene_depth = DEFAULT_POINT_DEPTH
        for (int f = 0; f < k; f++) {
            feat_mu.push_back(uv[2 * f]);
            feat_mu.push_back(uv[2 * f + 1]);
            feat_mu.push_back((T)(1.0 / (double)avg_depth));  // mu(2) = 1.0/depth
            last_klt.push_back(uv[2 * f]);
            last_klt.push_back(uv[2 * f + 1]);
            del_flag.push_back(0);
            S(idx, idx) = (T)cfg.default_point_homogenous_variance; idx++;
            S(idx, idx) = (T)cfg.default_point_homogenous_variance; idx++;
            S(idx, idx) = (T)cfg.default_point_depth_variance; idx++;
        }
    }

    // TightlyCoupledEKF.cpp:328-395
    void convolve_base_state(const T* last, T dt, T* out) const {
        Vec3<T> pos{last[0], last[1], last[2]};
        Quat<T> quat{last[3], last[4], last[5], last[6]};
        Vec3<T> vel{last[7], last[8], last[9]};
        Vec3<T> omega{last[10], last[11], last[12]};
        Vec3<T> accel{last[13], last[14], last[15]};

        Vec3<T> d = rotate(quat, translation(vel, accel, dt));
        pos = {pos.x + d.x, pos.y + d.y, pos.z + d.z};

        const Arith ar = arith_of(cfg);
        Quat<T> dq = delta_quat(omega, dt, T(1), ar);
        Quat<T> dq_inv = inverse(dq, ar);

        Vec3<T> va{vel.x + dt * accel.x, vel.y + dt * accel.y, vel.z + dt * accel.z};
        vel = rotate(dq_inv, va);
        accel = rotate(dq_inv, accel);
        quat = qmul(quat, dq, ar);

        out[0] = pos.x; out[1] = pos.y; out[2] = pos.z;
        out[3] = quat.w; out[4] = quat.x; out[5] = quat.y; out[6] = quat.z;
        out[7] = vel.x; out[8] = vel.y; out[9] = vel.z;
        out[10] = last[10]; out[11] = last[11]; out[12] = last[12];
        out[13] = accel.x; out[14] = accel.y; out[15] = accel.z;
        for (int i = 16; i < BASE; i++) out[i] = last[i];
    }

    // TightlyCoupledEKF.cpp:397-460
    void convolve_feature(const T* base, const T* feat, T dt, T* out) {
        Vec3<T> vel{base[7], base[8], base[9]};
        Vec3<T> accel{base[13], base[14], base[15]};
        Vec3<T> p{feat[0], feat[1], feat[2]};
        p.z = (T)(1.0 / (double)p.z);
        p.x = p.x * p.z;
        p.y = p.y * p.z;
        Vec3<T> tr = translation(vel, accel, dt);

        Quat<T> dq_inv;
        if (cfg.emulate_static_cache) {
            if (cache_om[0] != base[10] || cache_om[1] != base[11] || cache_om[2] != base[12]) {
                cache_dq_inv = delta_quat(Vec3<T>{base[10], base[11], base[12]}, dt, T(-1), arith_of(cfg));
                cache_om[0] = base[10]; cache_om[1] = base[11]; cache_om[2] = base[12];
            }
            dq_inv = cache_dq_inv;
        } else {
            dq_inv = delta_quat(Vec3<T>{base[10], base[11], base[12]}, dt, T(-1), arith_of(cfg));
        }

        Vec3<T> a = rotate(dq_inv, p);
        Vec3<T> b = rotate(dq_inv, tr);
        p = {a.x + (-b.x), a.y + (-b.y), a.z + (-b.z)};
        p.x /= p.z;
        p.y /= p.z;
        p.z = (T)(1.0 / (double)p.z);
        out[0] = p.x; out[1] = p.y; out[2] = p.z;
    }

    // TightlyCoupledEKF.cpp:123-174: diagonal of Q(dt)
    void process_noise_diag(T dt, std::vector<T>& q) const {
        q.assign(n, T(0));
        T low = (T)(0.0001 * (double)dt), posn = (T)(0.0001 * (double)dt);
        T veln = (T)(0.01 * (double)dt), omn = (T)(5 * dt), accn = (T)(5 * dt);
        T bias = (T)(0.001 * (double)dt);
        for (int i = 0; i < 7; i++) q[i] = posn;
        for (int i = 7; i < 10; i++) q[i] = veln;
        for (int i = 10; i < 13; i++) q[i] = omn;
        for (int i = 13; i < 16; i++) q[i] = accn;
        for (int i = 16; i < 22; i++) q[i] = bias;
        for (int i = BASE; i < n; i++) q[i] = low;
    }

    // FD helpers: `x += DELTA_SHIFT` is float(double(x)+1e-3) etc. (:182,193-198)
    static T plus_delta(T x) { return (T)((double)x + 1e-3); }
    static T minus_2delta(T x) { return (T)((double)x - 2 * 1e-3); }
    static T two_delta() { return (T)(2 * 1e-3); }

    // TightlyCoupledEKF.cpp:176-325; F dense column-major n x n (zero elsewhere)
    void linearize(T dt, std::vector<T>& F) {
        const int N = num_features();
        const Arith ar = arith_of(cfg);
        F.assign((size_t)n * n, T(0));
        auto Fat = [&](int i, int j) -> T& { return F[(size_t)j * n + i]; };
        T test_mu[BASE];
        for (int i = 0; i < BASE; i++) test_mu[i] = base_mu[i];
        T hi[BASE], lo[BASE];
        std::vector<T> fd((size_t)3 * N), tmp(3);
        for (int j = 0; j < BASE; j++) {
            if (j <= 6) {
                test_mu[j] = plus_delta(test_mu[j]);
                convolve_base_state(test_mu, dt, hi);
                test_mu[j] = minus_2delta(test_mu[j]);
                convolve_base_state(test_mu, dt, lo);
                test_mu[j] = base_mu[j];
                for (int i = 0; i < BASE; i++) Fat(i, j) = vec_div((T)(hi[i] - lo[i]), two_delta(), ar);
            } else if (j <= 15) {
                test_mu[j] = plus_delta(test_mu[j]);
                convolve_base_state(test_mu, dt, hi);
                test_mu[j] = minus_2delta(test_mu[j]);
                convolve_base_state(test_mu, dt, lo);
                test_mu[j] = base_mu[j];
                for (int i = 0; i < BASE; i++) Fat(i, j) = vec_div((T)(hi[i] - lo[i]), two_delta(), ar);

                test_mu[j] = plus_delta(test_mu[j]);
                for (int f = 0; f < N; f++) convolve_feature(test_mu, &feat_mu[3 * f], dt, &fd[3 * f]);
                test_mu[j] = minus_2delta(test_mu[j]);
                for (int f = 0; f < N; f++) {
                    convolve_feature(test_mu, &feat_mu[3 * f], dt, tmp.data());
                    fd[3 * f] -= tmp[0]; fd[3 * f + 1] -= tmp[1]; fd[3 * f + 2] -= tmp[2];
                }
                test_mu[j] = base_mu[j];
                for (int r = 0; r < 3 * N; r++) Fat(BASE + r, j) = vec_div(fd[r], two_delta(), ar);
            } else {
                Fat(j, j) = T(1);
            }
        }
        int col = BASE;
        for (int f = 0; f < N; f++) {
            T basef[3] = {feat_mu[3 * f], feat_mu[3 * f + 1], feat_mu[3 * f + 2]};
            T test[3] = {basef[0], basef[1], basef[2]};
            int row = col;
            for (int c = 0; c < 3; c++) {
                T dhi[3], dlo[3];
                test[c] = plus_delta(test[c]);
                convolve_feature(base_mu, test, dt, dhi);
                test[c] = minus_2delta(test[c]);
                convolve_feature(base_mu, test, dt, dlo);
                test[c] = basef[c];
                for (int r = 0; r < 3; r++) Fat(row + r, col) = vec_div((T)(dhi[r] - dlo[r]), two_delta(), ar);
                col++;
            }
        }
    }

    T flush_thresh() const { return (T)1e-8 * (T)1e-5; }  // SPARSE_THRESH*SPARSE_EPS
    void prune(std::vector<T>& M) const {
        const T th = flush_thresh();
        for (auto& x : M)
            if (!(std::fabs(x) > th)) x = T(0);
    }

    // Nonzero pattern of the left operand, column by column (what Eigen's column-major
    // sparse storage would iterate).  Skipping exact zeros is arithmetically identical to
    // adding them (up to the sign of a zero sum).
    struct Pattern {
        bool dense = true;
        std::vector<int> colptr, rowidx;
    };
    static void build_pattern(const T* A, int nr, int nc, int lda, Pattern& p) {
        p.dense = false;
        p.colptr.assign(nc + 1, 0);
        p.rowidx.clear();
        for (int k = 0; k < nc; k++) {
            for (int i = 0; i < nr; i++)
                if (A[(size_t)k * lda + i] != T(0)) p.rowidx.push_back(i);
            p.colptr[k + 1] = (int)p.rowidx.size();
        }
    }
    // C(nr x np) = A(nr x nk) * op(B); op(B)(k,j) = B(k,j) or, with bt, B(j,k).
    // Column-major; for every output column the sum runs over ascending k with a separate
    // multiply and add, i.e. Eigen's sparse*sparse accumulation order.
    static void matmul(int nr, int nk, int np, const T* A, int lda, const Pattern& pa, const T* B, int ldb,
                       bool bt, T* C, int ldc) {
#pragma omp parallel for schedule(static)
        for (int j = 0; j < np; j++) {
            T* c = C + (size_t)j * ldc;
            for (int i = 0; i < nr; i++) c[i] = T(0);
            for (int k = 0; k < nk; k++) {
                T b = bt ? B[(size_t)k * ldb + j] : B[(size_t)j * ldb + k];
                if (b == T(0)) continue;
                const T* a = A + (size_t)k * lda;
                if (pa.dense) {
                    for (int i = 0; i < nr; i++) c[i] += a[i] * b;
                } else {
                    for (int q = pa.colptr[k]; q < pa.colptr[k + 1]; q++) {
                        int i = pa.rowidx[q];
                        c[i] += a[i] * b;
                    }
                }
            }
        }
    }

    // TightlyCoupledEKF.cpp:96-121
    void process(T dt) {
        std::vector<T> F;
        linearize(dt, F);
        const int N = num_features();
        for (int f = 0; f < N; f++) {
            T out[3];
            convolve_feature(base_mu, &feat_mu[3 * f], dt, out);
            feat_mu[3 * f] = out[0]; feat_mu[3 * f + 1] = out[1]; feat_mu[3 * f + 2] = out[2];
        }
        T nb[BASE];
        convolve_base_state(base_mu, dt, nb);
        for (int i = 0; i < BASE; i++) base_mu[i] = nb[i];

        std::vector<T> X((size_t)n * n), P((size_t)n * n);
        Pattern pf, pd;
        build_pattern(F.data(), n, n, n, pf);
        matmul(n, n, n, F.data(), n, pf, Sigma.data(), n, false, X.data(), n);  // F*Sigma
        matmul(n, n, n, X.data(), n, pd, F.data(), n, true, P.data(), n);       // (F*Sigma)*F^T
        std::vector<T> q;
        process_noise_diag(dt, q);
        for (int i = 0; i < n; i++) P[(size_t)i * n + i] += q[i];
        Sigma.swap(P);
        prune(Sigma);
    }

    // TightlyCoupledEKF.cpp:634-661: state index of row r of H (one 1.0 per row)
    void form_measurement_map(const uint8_t* measured, std::vector<int>& idx) const {
        idx.clear();
        const int N = num_features();
        for (int i = 0; i < N; i++)
            if (measured[i]) {
                idx.push_back(i * 3 + BASE);
                idx.push_back(i * 3 + BASE + 1);
            }
    }

    // TightlyCoupledEKF.cpp:475-628.  z: 2 per landmark, R: 4 per landmark (col-major
    // 2x2), pass: 1 per landmark.  Returns last_update_info: bit 0 = the LDLT met a pivot <= 0 (the reference says
    // nothing and continues), bit 1 = a pivot exactly zero (the reference: ROS_ERROR_COND at :579, then continues).
    int update(const T* z_in, const T* R_in, const uint8_t* pass) {
        const int N = num_features();
        std::vector<int> idx;
        form_measurement_map(pass, idx);
        const int m = (int)idx.size();
        std::vector<T> z(m), mu(n), Rm((size_t)m * m, T(0));
        for (int i = 0; i < BASE; i++) mu[i] = base_mu[i];
        int j = 0;
        for (int i = 0; i < N; i++) {
            if (pass[i]) {
                last_klt[2 * i] = z_in[2 * i];
                last_klt[2 * i + 1] = z_in[2 * i + 1];
                z[j] = z_in[2 * i];
                Rm[(size_t)j * m + j] = R_in[4 * i + 0];  // (0,0)
                j++;
                z[j] = z_in[2 * i + 1];
                Rm[(size_t)j * m + j] = R_in[4 * i + 3];          // (1,1)
                Rm[(size_t)j * m + (j - 1)] = R_in[4 * i + 2];    // R(j-1,j) = cov(0,1)
                Rm[(size_t)(j - 1) * m + j] = R_in[4 * i + 1];    // R(j,j-1) = cov(1,0)
                j++;
            } else {
                del_flag[i] = 1;
            }
            mu[BASE + 3 * i] = feat_mu[3 * i];
            mu[BASE + 3 * i + 1] = feat_mu[3 * i + 1];
            mu[BASE + 3 * i + 2] = feat_mu[3 * i + 2];
        }
        last_update_info = 0;
        if (m == 0) {
            // reference logs an error and carries on with empty matrices: every product
            // is empty, Sigma = I*Sigma*I, quaternion renormalised.
            T qn = std::sqrt(mu[3] * mu[3] + mu[4] * mu[4] + mu[5] * mu[5] + mu[6] * mu[6]);
            for (int i = 3; i <= 6; i++) base_mu[i] = mu[i] / qn;
            prune(Sigma);
            return 0;
        }
        // y = z - H*mu
        std::vector<T> y(m);
        for (int r = 0; r < m; r++) y[r] = z[r] - mu[idx[r]];
        // S = H*Sigma*H^T + R
        std::vector<T> Sm((size_t)m * m);
        for (int c = 0; c < m; c++)
            for (int r = 0; r < m; r++) Sm[(size_t)c * m + r] = S(idx[r], idx[c]) + Rm[(size_t)c * m + r];
        // SimplicialLDLT(S^T): up-looking LDL^T of the lower triangle of S^T -- i.e. of S's UPPER triangle, mirrored -- restated
        // dense.  Lr is row-major: Lr[r*m+c] = L(r,c), in the PERMUTED numbering when an ordering applies.
        std::vector<int> perm(m);
        for (int r = 0; r < m; r++) perm[r] = r;
        bool natural = true, dense_pattern = true;
        std::vector<int> cp, ri;  // structural pattern of the symmetric matrix (full, rows ascending per column)
        if (cfg.ldlt_amd_order) {
            // structural(r,c), r <= c: S(r,c) != 0, or the diagonal, or inside one of R's 2 x 2 blocks (rows 2j, 2j+1)
            auto structural = [&](int r, int c) {
                const int lo = r < c ? r : c, hi = r < c ? c : r;
                return lo == hi || (lo >> 1) == (hi >> 1) || Sm[(size_t)hi * m + lo] != T(0);
            };
            cp.assign(m + 1, 0);
            for (int c = 0; c < m; c++) {
                cp[c] = (int)ri.size();
                for (int r = 0; r < m; r++)
                    if (structural(r, c)) ri.push_back(r);
                    else dense_pattern = false;
            }
            cp[m] = (int)ri.size();
            perm = ekf_oracle::amd_order(m, cp, ri, cfg.amd_keep_diagonal != 0);
            for (int r = 0; r < m; r++) natural = natural && perm[r] == r;
        }
        last_perm = perm;
        std::vector<T> Lr((size_t)m * m, T(0)), D(m), yrow(m);
        if (natural && dense_pattern && !cfg.ldlt_general_path) {
            for (int r = 0; r < m; r++) {
                T d = Sm[(size_t)r * m + r];
                for (int c = 0; c < r; c++) {
                    T yc = Sm[(size_t)r * m + c];  // S^T(r,c) = S(c,r)
                    const T* lc = &Lr[(size_t)c * m];
                    for (int k = 0; k < c; k++) yc -= lc[k] * yrow[k];
                    yrow[c] = yc;
                    T l = yc / D[c];
                    Lr[(size_t)r * m + c] = l;
                    d -= l * yc;
                }
                D[r] = d;
                Lr[(size_t)r * m + r] = T(1);
                if (!(d > T(0))) last_update_info |= 1;
                if (d == T(0)) last_update_info |= 2;
            }
        } else {
            // The general form, as SimplicialCholesky_impl.h's analyzePattern_preordered / factorize_preordered run it on
            // ap = upper triangle of S^T(P, P): elimination tree, then row k of L from the entries of column k of ap (ascending row
            // index), its pattern gathered along the tree in topological order, one sparse triangular solve per row.  On a dense
            // pattern in natural order this visits 0 .. k-1 ascending: the loop above, bit for bit (tests/test_oracle_amd_cpu.py).
            auto apv = [&](int i, int k) {  // ap(i,k), i <= k: S's upper triangle at the permuted pair
                const int a = perm[i], b = perm[k];
                return a <= b ? Sm[(size_t)b * m + a] : Sm[(size_t)a * m + b];
            };
            auto in_ap = [&](int i, int k) {
                if (!cfg.ldlt_amd_order) return true;
                const int a = perm[i], b = perm[k];
                const int lo = a < b ? a : b, hi = a < b ? b : a;
                return lo == hi || (lo >> 1) == (hi >> 1) || Sm[(size_t)hi * m + lo] != T(0);
            };
            std::vector<int> parent(m, -1), tags(m, -1), pattern(m);
            for (int k = 0; k < m; k++) {  // elimination tree
                parent[k] = -1;
                tags[k] = k;
                for (int i0 = 0; i0 < k; i0++) {
                    if (!in_ap(i0, k)) continue;
                    for (int i = i0; tags[i] != k; i = parent[i]) {
                        if (parent[i] == -1) parent[i] = k;
                        tags[i] = k;
                    }
                }
            }
            std::vector<std::vector<int>> Lrow(m);
            std::vector<std::vector<T>> Lval(m);
            std::vector<T> y(m, T(0));
            std::fill(tags.begin(), tags.end(), -1);
            for (int k = 0; k < m; k++) {
                y[k] = T(0);
                int top = m;
                tags[k] = k;
                for (int i0 = 0; i0 <= k; i0++) {
                    if (!in_ap(i0, k)) continue;
                    y[i0] += apv(i0, k);
                    int len = 0;
                    for (int i = i0; tags[i] != k; i = parent[i]) {
                        pattern[len++] = i;
                        tags[i] = k;
                    }
                    while (len > 0) pattern[--top] = pattern[--len];
                }
                T d = y[k];
                y[k] = T(0);
                for (; top < m; ++top) {
                    const int i = pattern[top];
                    const T yi = y[i];
                    y[i] = T(0);
                    const T l_ki = yi / D[i];
                    for (size_t q = 0; q < Lrow[i].size(); q++) y[Lrow[i][q]] -= Lval[i][q] * yi;
                    d -= l_ki * yi;
                    Lrow[i].push_back(k);
                    Lval[i].push_back(l_ki);
                    Lr[(size_t)k * m + i] = l_ki;
                }
                D[k] = d;
                Lr[(size_t)k * m + k] = T(1);
                if (!(d > T(0))) last_update_info |= 1;
                if (d == T(0)) last_update_info |= 2;
            }
        }
        std::vector<T> Lc((size_t)m * m);  // column-major copy: Lc[c*m+r] = L(r,c)
        for (int r = 0; r < m; r++)
            for (int c = 0; c < m; c++) Lc[(size_t)c * m + r] = Lr[(size_t)r * m + c];
        // K^T = (S^T)^-1 (Sigma*H^T)^T : one solve per state row i
        std::vector<T> K((size_t)n * m);  // n x m column-major
#pragma omp parallel
        {
            std::vector<T> rhs(m);
#pragma omp for schedule(static)
            for (int i = 0; i < n; i++) {
                for (int r = 0; r < m; r++) rhs[r] = Sigma[(size_t)idx[perm[r]] * n + i];  // (Sigma*H^T)(i, P[r]): dest = m_P * b
                for (int r = 0; r < m; r++) {                                          // L w = rhs
                    T v = rhs[r];
                    const T* lr = &Lr[(size_t)r * m];
                    for (int k = 0; k < r; k++) v -= lr[k] * rhs[k];
                    rhs[r] = v;
                }
                for (int r = 0; r < m; r++) rhs[r] = rhs[r] * (T(1) / D[r]);           // D^-1 w
                for (int r = m - 1; r >= 0; r--) {                                     // L^T x = w
                    T v = rhs[r];
                    const T* lc = &Lc[(size_t)r * m];
                    for (int k = r + 1; k < m; k++) v -= lc[k] * rhs[k];
                    rhs[r] = v;
                }
                for (int r = 0; r < m; r++) K[(size_t)perm[r] * n + i] = rhs[r];  // dest = m_Pinv * dest
            }
        }
        prune(K);  // .sparseView(SPARSE_THRESH, SPARSE_EPS)
        // I_KH = I - K*H ; prune
        std::vector<T> IKH((size_t)n * n, T(0));
        for (int i = 0; i < n; i++) IKH[(size_t)i * n + i] = T(1);
        for (int r = 0; r < m; r++)
            for (int i = 0; i < n; i++) IKH[(size_t)idx[r] * n + i] -= K[(size_t)r * n + i];
        prune(IKH);
        // Sigma = I_KH*Sigma*I_KH^T + K*R*K^T
        std::vector<T> T1((size_t)n * n), T2((size_t)n * n);
        Pattern pikh, pdense;
        build_pattern(IKH.data(), n, n, n, pikh);
        matmul(n, n, n, IKH.data(), n, pikh, Sigma.data(), n, false, T1.data(), n);
        matmul(n, n, n, T1.data(), n, pdense, IKH.data(), n, true, T2.data(), n);
        std::vector<T> KR((size_t)n * m), EN((size_t)n * n);
        matmul(n, m, m, K.data(), n, pdense, Rm.data(), m, false, KR.data(), n);
        matmul(n, m, n, KR.data(), n, pdense, K.data(), n, true, EN.data(), n);
        for (size_t e = 0; e < T2.size(); e++) T2[e] += EN[e];
        Sigma.swap(T2);
        // mu += K*y
        std::vector<T> Ky(n, T(0));
        for (int r = 0; r < m; r++)
            for (int i = 0; i < n; i++) Ky[i] += K[(size_t)r * n + i] * y[r];
        for (int i = 0; i < n; i++) mu[i] += Ky[i];
        T qn = std::sqrt(mu[3] * mu[3] + mu[4] * mu[4] + mu[5] * mu[5] + mu[6] * mu[6]);
        for (int i = 3; i <= 6; i++) mu[i] /= qn;
        for (int i = 0; i < BASE; i++) base_mu[i] = mu[i];
        for (int i = 0; i < N; i++) {
            feat_mu[3 * i] = mu[BASE + 3 * i];
            feat_mu[3 * i + 1] = mu[BASE + 3 * i + 1];
            feat_mu[3 * i + 2] = mu[BASE + 3 * i + 2];
        }
        prune(Sigma);
        return last_update_info;
    }

    // ---- SURVEY 8(f) F4, second half: an IMU measurement update.  NOT a restatement of reference code: the reference's
    // imu_callback is a logging stub (EKFVIO.cpp:113-115) and its imu_update_buffer is never touched (EKFVIO.h:59-64).
    // This is the specification the HIP path (cfg.use_imu) is tested against, written in the reference's style:
    //   z = [gyro; accel],  h(x) = [omega + b_gyr ;  a + b_acc - R(q)^T g]      (state indices: q 3-6, omega 10-12,
    //                                                                           a 13-15, b_acc 16-18, b_gyr 19-21)
    // with R(q)^T g evaluated like every rotation of the filter (Eigen's q*v formula on the conjugate, q not normalised),
    // H its analytic Jacobian (identities on omega/b_gyr and a/b_acc, -d(R^T g)/dq on the quaternion), the noise
    // diag(gyro_var x3, accel_var x3), and the update in the reference's Joseph form (:559-609):
    //   S = H Sigma H^T + R, K = Sigma H^T S^-1, Sigma = (I-KH) Sigma (I-KH)^T + K R K^T evaluated as T = Sigma - K (H Sigma),
    //   G = K R - T H^T, Sigma = T + G K^T;  mu += K (z - h);  quaternion renormalised;  prune.
    static void rt_gravity(const T* q4, const T* g3, T* out3, T* jac12 /* d out / d(w,x,y,z), row-major 3x4, may be null */) {
        const T w = q4[0];
        const Vec3<T> c{-q4[1], -q4[2], -q4[3]};  // conjugate's vector part
        const Vec3<T> v{g3[0], g3[1], g3[2]};
        Vec3<T> uv = cross(c, v);
        uv = {uv.x + uv.x, uv.y + uv.y, uv.z + uv.z};
        const Vec3<T> cu = cross(c, uv);
        out3[0] = v.x + w * uv.x + cu.x;
        out3[1] = v.y + w * uv.y + cu.y;
        out3[2] = v.z + w * uv.z + cu.z;
        if (!jac12) return;
        jac12[0] = uv.x;  // d/dw
        jac12[4] = uv.y;
        jac12[8] = uv.z;
        for (int k = 0; k < 3; k++) {
            const Vec3<T> e{k == 0 ? T(1) : T(0), k == 1 ? T(1) : T(0), k == 2 ? T(1) : T(0)};
            Vec3<T> ev = cross(e, v);
            ev = {ev.x + ev.x, ev.y + ev.y, ev.z + ev.z};          // d uv / d c_k
            const Vec3<T> a = cross(e, uv), b = cross(c, ev);
            // d out / d c_k = w * ev + e_k x uv + c x ev;  c = -(x,y,z), so d/d(x,y,z)_k is its negative
            jac12[0 + 1 + k] = -(w * ev.x + a.x + b.x);
            jac12[4 + 1 + k] = -(w * ev.y + a.y + b.y);
            jac12[8 + 1 + k] = -(w * ev.z + a.z + b.z);
        }
    }

    void imu_update(const T* gyro3, const T* accel3, T gyro_var, T accel_var, const T* gravity3) {
        const int m = 6;
        // the 16 state columns H touches
        static const int cols[16] = {3, 4, 5, 6, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21};
        T Hm[6][16] = {};
        T rg[3], jac[12];
        rt_gravity(&base_mu[3], gravity3, rg, jac);
        for (int r = 0; r < 3; r++) {
            Hm[r][4 + r] = T(1);       // omega
            Hm[r][13 + r] = T(1);      // b_gyr
            Hm[3 + r][7 + r] = T(1);   // a
            Hm[3 + r][10 + r] = T(1);  // b_acc
            for (int k = 0; k < 4; k++) Hm[3 + r][k] = -jac[4 * r + k];
        }
        T y[6];
        for (int r = 0; r < 3; r++) {
            y[r] = gyro3[r] - (base_mu[10 + r] + base_mu[19 + r]);
            y[3 + r] = accel3[r] - ((base_mu[13 + r] + base_mu[16 + r]) - rg[r]);
        }
        const T Rd[6] = {gyro_var, gyro_var, gyro_var, accel_var, accel_var, accel_var};
        // X = Sigma H^T (n x 6), W = H Sigma (6 x n)
        std::vector<T> X((size_t)n * m, T(0)), W((size_t)m * n, T(0));
        for (int r = 0; r < m; r++)
            for (int c = 0; c < 16; c++) {
                const T h = Hm[r][c];
                if (h == T(0)) continue;
                for (int i = 0; i < n; i++) {
                    X[(size_t)r * n + i] += S(i, cols[c]) * h;
                    W[(size_t)i * m + r] += h * S(cols[c], i);
                }
            }
        // S = H X + R, its Cholesky, K = X S^-1
        T Sm[6][6], L[6][6] = {};
        for (int r = 0; r < m; r++)
            for (int s = 0; s < m; s++) {
                T acc = T(0);
                for (int c = 0; c < 16; c++) acc += Hm[r][c] * X[(size_t)s * n + cols[c]];
                Sm[r][s] = acc + (r == s ? Rd[r] : T(0));
            }
        for (int j = 0; j < m; j++) {
            T d = Sm[j][j];
            for (int k = 0; k < j; k++) d -= L[j][k] * L[j][k];
            L[j][j] = std::sqrt(d);
            for (int i = j + 1; i < m; i++) {
                T v = Sm[i][j];  // lower triangle of S
                for (int k = 0; k < j; k++) v -= L[i][k] * L[j][k];
                L[i][j] = v / L[j][j];
            }
        }
        std::vector<T> K((size_t)n * m);
        for (int i = 0; i < n; i++) {
            T t[6];
            for (int r = 0; r < m; r++) {  // t L^T = x  (row vector): forward in r
                T v = X[(size_t)r * n + i];
                for (int k = 0; k < r; k++) v -= t[k] * L[r][k];
                t[r] = v / L[r][r];
            }
            for (int r = m - 1; r >= 0; r--) {  // k L = t
                T v = t[r];
                for (int k = r + 1; k < m; k++) v -= t[k] * L[k][r];  // t holds the solution from r+1 on
                t[r] = v / L[r][r];
            }
            for (int r = 0; r < m; r++) K[(size_t)r * n + i] = t[r];
        }
        // T = Sigma - K W on the 16 columns G needs; G = K R - T H^T
        std::vector<T> G((size_t)n * m);
        for (int i = 0; i < n; i++)
            for (int r = 0; r < m; r++) {
                T th = T(0);
                for (int c = 0; c < 16; c++) {
                    const T h = Hm[r][c];
                    if (h == T(0)) continue;
                    T t = S(i, cols[c]);
                    for (int s = 0; s < m; s++) t -= K[(size_t)s * n + i] * W[(size_t)cols[c] * m + s];
                    th += t * h;
                }
                G[(size_t)r * n + i] = K[(size_t)r * n + i] * Rd[r] - th;
            }
        // Sigma' = Sigma - K W + G K^T, pruned
        for (int j = 0; j < n; j++)
            for (int i = 0; i < n; i++) {
                T v = S(i, j);
                for (int s = 0; s < m; s++) v -= K[(size_t)s * n + i] * W[(size_t)j * m + s];
                for (int s = 0; s < m; s++) v += G[(size_t)s * n + i] * K[(size_t)s * n + j];
                Sigma[(size_t)j * n + i] = v;
            }
        prune(Sigma);
        // mu += K y; quaternion renormalised (:600-609)
        const int N = num_features();
        std::vector<T> mu(n);
        for (int i = 0; i < BASE; i++) mu[i] = base_mu[i];
        for (int i = 0; i < 3 * N; i++) mu[BASE + i] = feat_mu[i];
        for (int i = 0; i < n; i++) {
            T acc = T(0);
            for (int r = 0; r < m; r++) acc += K[(size_t)r * n + i] * y[r];
            mu[i] += acc;
        }
        const T qn = std::sqrt(mu[3] * mu[3] + mu[4] * mu[4] + mu[5] * mu[5] + mu[6] * mu[6]);
        for (int i = 3; i <= 6; i++) mu[i] /= qn;
        for (int i = 0; i < BASE; i++) base_mu[i] = mu[i];
        for (int i = 0; i < 3 * N; i++) feat_mu[i] = mu[BASE + i];
    }

    // TightlyCoupledEKF.cpp:699-714: returns min diagonal and max |S(i,j)-S(j,i)|
    void check_sigma(T* min_diag, T* max_asym) {
        T md = S(0, 0), ma = T(0);
        for (int i = 0; i < n; i++) {
            if (S(i, i) < md) md = S(i, i);
            for (int j = i + 1; j < n; j++) {
                T d = std::fabs(S(i, j) - S(j, i));
                if (d > ma) ma = d;
            }
        }
        *min_diag = md;
        *max_asym = ma;
    }
};

}  // namespace oracle
