// oracle/ekf_oracle_capi.cpp — TEST INFRASTRUCTURE ONLY (see ekf_oracle.hpp header).
// Flat C entry points over oracle::Filter<float> (orc32_*) and <double> (orc64_*) so that
// tests/ and bench.py's cpu_baseline leg can drive the restatement through ctypes.
#include <chrono>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "ekf_oracle.hpp"

using oracle::BASE;

#define ORC_API(P, T)                                                                                      \
    extern "C" void* P##_create(double depth, double depth_var, double homog_var, int emulate_cache) {     \
        oracle::Config c;                                                                                  \
        c.default_point_depth = depth;                                                                     \
        c.default_point_depth_variance = depth_var;                                                        \
        c.default_point_homogenous_variance = homog_var;                                                   \
        c.emulate_static_cache = emulate_cache;                                                            \
        return new oracle::Filter<T>(c);                                                                   \
    }                                                                                                      \
    extern "C" void P##_destroy(void* h) { delete (oracle::Filter<T>*)h; }                                 \
    /* which Eigen build the restatement follows (ekf_oracle.hpp Config); -1 leaves a switch as it is */   \
    extern "C" void P##_set_variant(void* h, int sse_quat, int trig_float, int div_reciprocal) {           \
        auto* f = (oracle::Filter<T>*)h;                                                                   \
        if (sse_quat >= 0) f->cfg.eigen_sse_quat = sse_quat;                                               \
        if (trig_float >= 0) f->cfg.trig_float = trig_float;                                               \
        if (div_reciprocal >= 0) f->cfg.div_reciprocal = div_reciprocal;                                   \
    }                                                                                                      \
    /* the LDLT's ordering: -1 leaves a switch as it is */                                                  \
    extern "C" void P##_set_ldlt_order(void* h, int amd_order, int keep_diagonal, int general_path) {      \
        auto* f = (oracle::Filter<T>*)h;                                                                   \
        if (amd_order >= 0) f->cfg.ldlt_amd_order = amd_order;                                             \
        if (keep_diagonal >= 0) f->cfg.amd_keep_diagonal = keep_diagonal;                                  \
        if (general_path >= 0) f->cfg.ldlt_general_path = general_path;                                    \
    }                                                                                                      \
    /* the ordering the last update used: P[k] = measurement row of the k-th pivot; returns its length */  \
    extern "C" int P##_last_perm(void* h, int* out, int cap) {                                             \
        auto* f = (oracle::Filter<T>*)h;                                                                   \
        const int m = (int)f->last_perm.size();                                                            \
        for (int i = 0; i < m && i < cap; i++) out[i] = f->last_perm[i];                                   \
        return m;                                                                                          \
    }                                                                                                      \
    extern "C" int P##_num_features(void* h) { return ((oracle::Filter<T>*)h)->num_features(); }           \
    extern "C" int P##_dim(void* h) { return ((oracle::Filter<T>*)h)->n; }                                 \
    extern "C" void P##_add_features(void* h, const T* uv, int k) {                                        \
        ((oracle::Filter<T>*)h)->add_new_features(uv, k);                                                  \
    }                                                                                                      \
    extern "C" void P##_process(void* h, T dt) { ((oracle::Filter<T>*)h)->process(dt); }                   \
    extern "C" int P##_update(void* h, const T* z, const T* R, const uint8_t* pass) {                      \
        return ((oracle::Filter<T>*)h)->update(z, R, pass);                                                \
    }                                                                                                      \
    extern "C" void P##_linearize(void* h, T dt, T* Fdense) {                                              \
        std::vector<T> F;                                                                                  \
        auto* f = (oracle::Filter<T>*)h;                                                                   \
        f->linearize(dt, F);                                                                               \
        std::memcpy(Fdense, F.data(), sizeof(T) * F.size());                                               \
    }                                                                                                      \
    extern "C" void P##_convolve_base(void* h, const T* in22, T dt, T* out22) {                            \
        ((oracle::Filter<T>*)h)->convolve_base_state(in22, dt, out22);                                     \
    }                                                                                                      \
    extern "C" void P##_convolve_feature(void* h, const T* base22, const T* feat3, T dt, T* out3) {        \
        ((oracle::Filter<T>*)h)->convolve_feature(base22, feat3, dt, out3);                                \
    }                                                                                                      \
    extern "C" void P##_process_noise_diag(void* h, T dt, T* q) {                                          \
        std::vector<T> v;                                                                                  \
        ((oracle::Filter<T>*)h)->process_noise_diag(dt, v);                                                \
        std::memcpy(q, v.data(), sizeof(T) * v.size());                                                    \
    }                                                                                                      \
    extern "C" int P##_measurement_map(void* h, const uint8_t* measured, int* idx) {                       \
        std::vector<int> v;                                                                                \
        ((oracle::Filter<T>*)h)->form_measurement_map(measured, v);                                        \
        for (size_t i = 0; i < v.size(); i++) idx[i] = v[i];                                               \
        return (int)v.size();                                                                              \
    }                                                                                                      \
    extern "C" void P##_get_state(void* h, T* base22, T* feat3N, T* klt2N, uint8_t* delN, T* sigma) {      \
        auto* f = (oracle::Filter<T>*)h;                                                                   \
        if (base22) std::memcpy(base22, f->base_mu, sizeof(T) * BASE);                                     \
        if (feat3N) std::memcpy(feat3N, f->feat_mu.data(), sizeof(T) * f->feat_mu.size());                 \
        if (klt2N) std::memcpy(klt2N, f->last_klt.data(), sizeof(T) * f->last_klt.size());                 \
        if (delN) std::memcpy(delN, f->del_flag.data(), f->del_flag.size());                               \
        if (sigma) std::memcpy(sigma, f->Sigma.data(), sizeof(T) * f->Sigma.size());                       \
    }                                                                                                      \
    extern "C" void P##_set_state(void* h, int N, const T* base22, const T* feat3N, const T* klt2N,        \
                                  const uint8_t* delN, const T* sigma) {                                   \
        auto* f = (oracle::Filter<T>*)h;                                                                   \
        f->n = BASE + 3 * N;                                                                               \
        std::memcpy(f->base_mu, base22, sizeof(T) * BASE);                                                 \
        f->feat_mu.assign(feat3N, feat3N + 3 * N);                                                         \
        f->last_klt.assign(klt2N, klt2N + 2 * N);                                                          \
        f->del_flag.assign(delN, delN + N);                                                                \
        f->Sigma.assign(sigma, sigma + (size_t)f->n * f->n);                                               \
    }                                                                                                      \
    extern "C" void P##_imu_update(void* h, const T* gyro3, const T* accel3, T gyro_var, T accel_var,      \
                                   const T* gravity3) {                                                    \
        ((oracle::Filter<T>*)h)->imu_update(gyro3, accel3, gyro_var, accel_var, gravity3);                 \
    }                                                                                                      \
    extern "C" void P##_rt_gravity(const T* q4, const T* g3, T* out3, T* jac12) {                          \
        oracle::Filter<T>::rt_gravity(q4, g3, out3, jac12);                                                \
    }                                                                                                      \
    extern "C" void P##_check_sigma(void* h, T* min_diag, T* max_asym) {                                   \
        ((oracle::Filter<T>*)h)->check_sigma(min_diag, max_asym);                                          \
    }                                                                                                      \
    /* times `steps` repetitions of process(dt)+update(all measured) from the current state */             \
    extern "C" double P##_time_steps(void* h, int steps, T dt, const T* z, const T* R,                     \
                                     const uint8_t* pass) {                                                \
        auto* f = (oracle::Filter<T>*)h;                                                                   \
        auto t0 = std::chrono::steady_clock::now();                                                        \
        for (int s = 0; s < steps; s++) {                                                                  \
            f->process(dt);                                                                                \
            f->update(z, R, pass);                                                                         \
        }                                                                                                  \
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();               \
    }

ORC_API(orc32, float)
ORC_API(orc64, double)

extern "C" void orc_set_threads(int t) {
#ifdef _OPENMP
    omp_set_num_threads(t > 0 ? t : 1);
#else
    (void)t;
#endif
}
extern "C" int orc_max_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

// The ordering alone, for tests/test_oracle_amd_cpu.py: pattern in compressed-column form (rows ascending per column).
extern "C" void orc_amd_order(int n, const int* col_ptr, const int* row_idx, int keep_diagonal, int* perm_out) {
    std::vector<int> cp(col_ptr, col_ptr + n + 1), ri(row_idx, row_idx + col_ptr[n]);
    const std::vector<int> p = ekf_oracle::amd_order(n, cp, ri, keep_diagonal != 0);
    for (int i = 0; i < n; i++) perm_out[i] = p[i];
}
