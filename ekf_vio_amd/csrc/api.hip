// ekf_vio_amd/csrc/api.hip — the C-ABI of libekfvio_hip.so (include/ekfvio.h).
// Host-side orchestration only: every numeric step is a HIP kernel on the handle's stream.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>

#include <unistd.h>

#include "common.h"

#define HIPC(f, expr)                                                                              \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess) {                                                                   \
            (f)->last_error = std::string(#expr) + ": " + hipGetErrorString(e__);                  \
            return EKFVIO_EDEVICE;                                                                 \
        }                                                                                          \
    } while (0)

namespace {
// Allocation + zero fill ordered on the handle's (non-blocking) stream: nothing in this
// library relies on the legacy null stream, which a non-blocking stream does not wait for.
template <class T>
hipError_t dev_alloc(hipStream_t s, T** p, size_t count) {
    hipError_t e = hipMalloc((void**)p, sizeof(T) * (count ? count : 1));
    if (e == hipSuccess) e = hipMemsetAsync(*p, 0, sizeof(T) * (count ? count : 1), s);
    return e;
}

__global__ void init_sigma_kernel(float* P, int ld) {
    // initializeBaseState (TightlyCoupledEKF.cpp:23-56): diag [0x7, 30x9, 0.5x6]
    int i = threadIdx.x;
    if (i >= 7 && i <= 15) P[(size_t)i * ld + i] = 30.f;
    if (i >= 16 && i <= 21) P[(size_t)i * ld + i] = 0.5f;
}

// addNewFeatures (:58-94): new rows/columns are zero, new diagonal [hv, hv, dv]
// k_dev != nullptr: the count comes from device memory (the first-fit selection's result, ekfvio_step_image: the host
// learns it with the frame's status word, not in the middle of the frame)
__global__ void add_features_kernel(float* P, int ld, int n_old, int k, float hv, float dv, float* mu, float* last_klt,
                                    uint8_t* del_flag, const float* uv, int N_old, float inv_depth, const int* k_dev) {
    if (k_dev) k = *k_dev;
    const int n_new = n_old + 3 * k;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < (size_t)3 * k * n_new;
         e += (size_t)gridDim.x * blockDim.x) {
        int q = e / n_new;      // which new row/col
        int i = e % n_new;
        int s = n_old + q;
        P[(size_t)s * ld + i] = (i == s) ? ((q % 3 == 2) ? dv : hv) : 0.f;  // new column
        P[(size_t)i * ld + s] = (i == s) ? ((q % 3 == 2) ? dv : hv) : 0.f;  // new row
    }
    for (int fidx = blockIdx.x * blockDim.x + threadIdx.x; fidx < k; fidx += gridDim.x * blockDim.x) {
        float u = uv[2 * fidx], v = uv[2 * fidx + 1];
        mu[n_old + 3 * fidx] = u;
        mu[n_old + 3 * fidx + 1] = v;
        mu[n_old + 3 * fidx + 2] = inv_depth;
        last_klt[2 * (N_old + fidx)] = u;
        last_klt[2 * (N_old + fidx) + 1] = v;
        del_flag[N_old + fidx] = 0;
    }
}

int count_rows(const uint8_t* pass, int N) {
    int c = 0;
    for (int i = 0; i < N; i++) c += pass[i] ? 1 : 0;
    return 2 * c;
}
}  // namespace

static void drop_graph(ekfvio_filter* f);

static const char* kProfNames[PC_COUNT] = {"linearize",   "predict_structured", "gemm_predict", "gather",
                                           "cholesky",    "solve",              "gemm_update",  "update_misc",
                                           "klt_pyramid", "klt_track"};

// addNewFeatures (:58-94) with the `count` new (u,v) pairs already in f->zmeas on the device
int add_features_device(ekfvio_filter* f, int count) {
    if (count <= 0) return EKFVIO_OK;
    if (f->N + count > f->cfg.max_features) return EKFVIO_ECAPACITY;
    const float depth = f->cfg.default_point_depth;
    const float inv_depth = (float)(1.0 / (double)depth);  // Feature.cpp:18  mu(2) = 1.0/depth
    hipLaunchKernelGGL(add_features_kernel, dim3(64), dim3(256), 0, f->stream, f->P, f->ldp, f->n, count,
                       f->cfg.default_point_homogenous_variance, f->cfg.default_point_depth_variance, f->mu,
                       f->last_klt, f->del_flag, f->zmeas, f->N, inv_depth, nullptr);
    HIPC(f, hipStreamSynchronize(f->stream));
    f->N += count;
    f->n += 3 * count;
    return EKFVIO_OK;
}
// The same with the count in device memory (at most max_features - N, by construction of the selection): enqueued only.
// The caller adds the count to f->N / f->n once it has read it (wait_status's extra word).
void add_features_enqueue_device_count(ekfvio_filter* f, const int* count_dev) {
    const float inv_depth = (float)(1.0 / (double)f->cfg.default_point_depth);
    hipLaunchKernelGGL(add_features_kernel, dim3(64), dim3(256), 0, f->stream, f->P, f->ldp, f->n, 0,
                       f->cfg.default_point_homogenous_variance, f->cfg.default_point_depth_variance, f->mu,
                       f->last_klt, f->del_flag, f->zmeas, f->N, inv_depth, count_dev);
}

// one workgroup; publishes the status word itself once its own writes are out (see wait_status / poll_status)
__global__ void copy_small_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, const int* __restrict__ info,
                                  int* host_word, int seq) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        host_word[0] = info[0];
        host_word[2] = 0;
        __hip_atomic_store(host_word + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// The host's wait at the end of a frame.  A one-thread kernel behind everything else on the stream writes the status
// word and a sequence number straight into pinned host memory; the host polls the sequence number.  Against a 4-byte
// device-to-host copy plus hipStreamSynchronize this saves the copy's command and the interrupt-driven wake-up (≈10 us
// per frame).  Polling is bounded: after 300 us (a long device-resident run is in flight) the host blocks in
// hipStreamSynchronize like before.
__global__ void publish_status_kernel(const int* __restrict__ info, int* host_word, int seq, const int* __restrict__ extra) {
    host_word[0] = info[0];
    host_word[2] = extra ? *extra : 0;
    __hip_atomic_store(host_word + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
void launch_publish_status(ekfvio_filter* f, int seq) {
    hipLaunchKernelGGL(publish_status_kernel, dim3(1), dim3(1), 0, f->stream, f->info, f->d_hinfo, seq, (const int*)nullptr);
}
int wait_status(ekfvio_filter* f, int* status, const int* extra_dev, int* extra_out) {
    const int seq = ++f->status_seq;
    hipLaunchKernelGGL(publish_status_kernel, dim3(1), dim3(1), 0, f->stream, f->info, f->d_hinfo, seq, extra_dev);
    return poll_status(f, seq, status, extra_out);
}
// The polling half: for a caller whose own last kernel publishes sequence number `seq` (from next_status_seq).
int next_status_seq(ekfvio_filter* f) { return ++f->status_seq; }
int poll_status(ekfvio_filter* f, int seq, int* status, int* extra_out) {
    HIPC(f, hipGetLastError());
    volatile int* hw = f->h_info;
    const auto t0 = std::chrono::steady_clock::now();
    int spins = 0;
    while (__atomic_load_n(&hw[1], __ATOMIC_ACQUIRE) != seq) {
        if ((++spins & 1023) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(300)) {
            HIPC(f, hipStreamSynchronize(f->stream));
            break;
        }
    }
    if (__atomic_load_n(&hw[1], __ATOMIC_ACQUIRE) != seq) {
        f->last_error = "status word not published";
        return EKFVIO_EDEVICE;
    }
    *status = hw[0];
    if (extra_out) *extra_out = hw[2];
    return EKFVIO_OK;
}

extern "C" {

int ekfvio_default_config(ekfvio_config* c) {
    if (!c) return EKFVIO_EINVAL;
    c->max_features = 100;                        // D_NUM_FEATURES
    c->default_point_depth = 0.5f;                // D_DEFAULT_POINT_DEPTH
    c->default_point_depth_variance = 100.f;      // D_DEFAULT_POINT_DEPTH_VARIANCE
    c->default_point_homogenous_variance = 1e-5f; // D_DEFAULT_POINT_HOMOGENOUS_VARIANCE
    c->predict_mode = EKFVIO_PREDICT_STRUCTURED;
    c->klt_window_size = 21;
    c->klt_max_pyramid_level = 3;
    c->klt_max_iterations = 30;
    c->klt_epsilon = 0.01f;
    c->klt_min_eigen = 1e-4f;
    c->kill_pad = 11;
    c->max_image_width = 640;
    c->max_image_height = 480;
    c->use_principal_point = 0;
    c->inverse_image_scale = 1;   // D_INVERSE_IMAGE_SCALE is 4; 1 = the caller hands over frames already at working size
    c->fast_threshold = 50;       // D_FAST_THRESHOLD
    c->min_new_feature_dist = 30; // D_MIN_NEW_FEATURE_DIST
    c->fast_blur_sigma = 0.f;     // D_FAST_BLUR_SIGMA (0 = no blur)
    c->replenish = 0;             // 1: ekfvio_step_image runs replenishFeatures (EKFVIO.cpp:154,172) itself
    c->sample_based_uncertainty = 0;  // reference behaviour: estimateUncertainty's constant (KLTTracker.cpp:100-106)
    c->use_imu = 0;                   // reference behaviour: imu_callback only logs (EKFVIO.cpp:113-115)
    c->imu_gyro_variance = 1e-4f;
    c->imu_accel_variance = 1e-2f;
    c->gravity[0] = 0.f;
    c->gravity[1] = 9.81f;
    c->gravity[2] = 0.f;
    return EKFVIO_OK;
}

static int create_body(ekfvio_filter* f, const ekfvio_config* cfg, int device, void* stream) {
    HIPC(f, hipSetDevice(device));
    if (stream) {
        f->stream = (hipStream_t)stream;
    } else {
        HIPC(f, hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking));
        f->own_stream = true;
    }
    const int maxf = cfg->max_features;
    f->n_cap = EKF_BASE + 3 * maxf;
    f->ldp = round_up(f->n_cap + 1, 64);  // one spare row/column: the K*y column of the Joseph-1 GEMM
    f->m_cap = round_up(2 * maxf > 0 ? 2 * maxf : 1, 64);
    const size_t pp = (size_t)f->ldp * f->ldp, pm = (size_t)f->ldp * f->m_cap;
    HIPC(f, dev_alloc(f->stream, &f->mu, f->ldp));
    HIPC(f, dev_alloc(f->stream, &f->mu_next, f->ldp));
    HIPC(f, dev_alloc(f->stream, &f->last_klt, 2 * (size_t)maxf));
    HIPC(f, dev_alloc(f->stream, &f->del_flag, (size_t)maxf));
    HIPC(f, dev_alloc(f->stream, &f->P, pp));
    HIPC(f, dev_alloc(f->stream, &f->P2, pp));
    HIPC(f, dev_alloc(f->stream, &f->FA, EKF_BASE * EKF_BASE));
    HIPC(f, dev_alloc(f->stream, &f->FB, 27 * (size_t)maxf));
    HIPC(f, dev_alloc(f->stream, &f->FD, 9 * (size_t)maxf));
    HIPC(f, dev_alloc(f->stream, &f->Fdense, pp));
    HIPC(f, dev_alloc(f->stream, &f->idx, (size_t)f->m_cap));
    HIPC(f, dev_alloc(f->stream, &f->inv_idx, (size_t)f->ldp));
    HIPC(f, dev_alloc(f->stream, &f->zmeas, 2 * (size_t)maxf));
    HIPC(f, dev_alloc(f->stream, &f->Rmeas, 4 * (size_t)maxf));
    HIPC(f, dev_alloc(f->stream, &f->pass, (size_t)maxf));
    HIPC(f, dev_alloc(f->stream, &f->yres, (size_t)f->m_cap));
    HIPC(f, dev_alloc(f->stream, &f->Rm, 2 * (size_t)f->m_cap));
    f->ld_aug = f->m_cap + f->ldp + f->m_cap;
    HIPC(f, dev_alloc(f->stream, &f->Saug, (size_t)f->ld_aug * f->m_cap));
    HIPC(f, dev_alloc(f->stream, &f->Laug, (size_t)f->ld_aug * f->m_cap));
    HIPC(f, dev_alloc(f->stream, &f->Linv, 64 * (size_t)f->m_cap));
    HIPC(f, dev_alloc(f->stream, &f->Lsign, (size_t)(f->m_cap / 64 > 256 ? f->m_cap / 64 : 256)));  // 256: the raw-solve test hook goes up to m = 16384
    {   // flags of the persistent sweeps (chol_persist.inc): ready[mb] + 2 x [row blocks x mb] + abort word
        const size_t mb2 = (size_t)(f->m_cap / 64);
        f->sweep_sync_words = std::max((size_t)1024, mb2 + 2 * (size_t)(f->ld_aug / 64 + 1) * mb2 + 8);  // (>= 1024: the test hooks sweep matrices that are not the filter's)
        HIPC(f, dev_alloc(f->stream, &f->sweep_sync, f->sweep_sync_words));
    }
    {
        hipDeviceProp_t prop;
        HIPC(f, hipGetDeviceProperties(&prop, device));
        f->num_cus = prop.multiProcessorCount;
        const char* e = getenv("EKFVIO_EARLY_STATUS");  // 0: ekfvio_update publishes its status behind the update's last kernel (rounds 1-3)
        if (e) f->early_status = atoi(e) ? 1 : 0;
        e = getenv("EKFVIO_UPLOAD_KERNEL");  // 0: the frame travels by the copy engine (hipMemcpyAsync) instead of the upload kernel
        if (e) f->upload_kernel = atoi(e) ? 1 : 0;
        e = getenv("EKFVIO_SWEEP_WAIT_MS");  // the persistent sweep's patience per wait where the update can be run again (default 3 ms)
        if (e && atof(e) > 0) f->sweep_wait_ticks = (int)std::min(2.0e9, 1e5 * atof(e));
        e = getenv("EKFVIO_SWEEP_RETRY_S");  // the first pause before a handle whose persistent sweep gave up tries it again (default 2 s)
        if (e && atof(e) > 0) f->sweep_retry_first_s = atof(e);
        e = getenv("EKFVIO_SWEEP");  // tuning knob: 0 = one launch per block step
        if (e) f->sweep_mode = atoi(e) ? 2 : 0;
        e = getenv("EKFVIO_FUSE_GATHER");  // tuning knob: 0 = gather and first diagonal tile in separate launches
        if (e) f->fuse_gather = atoi(e) ? 1 : 0;
        e = getenv("EKFVIO_T2");  // A/B knob: 0 = gain in the sweep, two Joseph GEMMs behind it (rounds 4-5)
        if (e) f->t2_flow = atoi(e) ? 1 : 0;
        e = getenv("EKFVIO_SCHUR");  // tuning knob: 1 = Sigma (I - K H)^T and K as Schur tiles inside the sweep instead of two GEMMs behind it
        if (e) f->schur = atoi(e) ? 1 : 0;
        e = getenv("EKFVIO_FRAME_OUTPUTS");  // tuning knob: 0 = the frame's last kernel publishes the status word only
        if (e) f->frame_outputs = atoi(e) ? 1 : 0;
        e = getenv("EKFVIO_FUSE_SWEEP");  // tuning knob: 0 = gather + first diagonal tile and the persistent sweep as two launches (round 3)
        if (e) f->fuse_sweep = atoi(e) ? 1 : 0;
        e = getenv("EKFVIO_PERSIST_OVERSUB");
        if (e) f->persist_oversub = atoi(e);
        e = getenv("EKFVIO_EARLY_OUTPUTS");  // tuning knob: 0 = a frame's outputs behind its last kernel (rounds 1-3)
        if (e) f->early_outputs = atoi(e) ? 1 : 0;
        e = getenv("EKFVIO_PERSIST_GAIN");
        if (e) f->persist_gain = atoi(e) ? 1 : 0;
        e = getenv("EKFVIO_PERSIST_EARLY");
        if (e) f->persist_early = atoi(e) ? 1 : 0;
        e = getenv("EKFVIO_GEMM_ORDER2D");  // tuning knob: 0 = the throughput-regime GEMM's tiles in block-index order (rounds 1-4)
        if (e) f->gemm_order2d = atoi(e) ? 1 : 0;
        e = getenv("EKFVIO_LIN_OVERLAP");  // 0 = every process(dt) of a device-resident run linearises for itself (rounds 3-5)
        if (e) f->lin_overlap = atoi(e) ? 1 : 0;
        e = getenv("EKFVIO_SYM_JOSEPH");  // 0 = the throughput regime's second Joseph GEMM forms both triangles (rounds 1-5)
        if (e) f->sym_joseph = atoi(e) ? 1 : 0;
        e = getenv("EKFVIO_FUSE_LINEARIZE");  // tuning knob: 0 = linearize_kernel and the propagation as two launches
        if (e) f->fuse_linearize = atoi(e) ? 1 : 0;
    }
    HIPC(f, dev_alloc(f->stream, &f->Km, pm));
    HIPC(f, dev_alloc(f->stream, &f->Wt, pm));
    HIPC(f, dev_alloc(f->stream, &f->Gm, pm));
    HIPC(f, dev_alloc(f->stream, &f->info, 4));
    HIPC(f, hipHostMalloc((void**)&f->h_info, 16 * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent));
    memset(f->h_info, 0, 16 * sizeof(int));
    HIPC(f, hipHostGetDevicePointer((void**)&f->d_hinfo, f->h_info, 0));
    {
        const size_t words = EKF_BASE + 4 * (size_t)(f->cfg.max_features > 0 ? f->cfg.max_features : 1);
        HIPC(f, hipHostMalloc((void**)&f->h_out, words * sizeof(float), hipHostMallocMapped | hipHostMallocCoherent));
        HIPC(f, hipHostGetDevicePointer((void**)&f->d_out, f->h_out, 0));
    }
    {
        const size_t cap = (size_t)(f->cfg.max_features > 0 ? f->cfg.max_features : 1);
        HIPC(f, hipHostMalloc((void**)&f->h_meas, 25 * cap, hipHostMallocDefault));
        HIPC(f, hipMalloc((void**)&f->d_meas, 25 * cap));
    }
    HIPC(f, hipEventCreate(&f->ev0));
    HIPC(f, hipEventCreate(&f->ev1));
    int rc = klt_alloc(f);
    if (rc != EKFVIO_OK) return rc;
    rc = fast_alloc(f);
    if (rc != EKFVIO_OK) return rc;
    return ekfvio_reset(f);
}

}  // extern "C"

// Live handles per device in this process.  The persistent sweep needs all its workgroups resident together; two such
// launches from different handles' streams can each get part of the compute units and wait for workgroups the other one
// keeps out (every wait is bounded, but that ends in an aborted update, not in a result).  It is therefore used only by
// a device's sole handle; with several handles on a device every one of them takes one launch per block step.
// Live handles per device, PROCESS-wide: the product library and the hooks build of the same sources can be loaded into one
// process (the tests do; a node that links one and dlopens a tool built on the other could), each with its own statics -- two copies of a
// per-library count would each see a "sole handle" and let two persistent launches loose on one device.  The first copy to need the
// registry allocates it and leaves its address in the process environment under a name that carries the pid (so a forked child's or an
// exec'ed program's copy of the variable is ignored); every later copy picks it up there.  No exported symbol, no file.
static std::atomic<int>* live_registry() {
    static std::atomic<int>* reg = [] {
        char name[64], val[32];
        snprintf(name, sizeof(name), "EKFVIO_LIVE_HANDLES_%ld", (long)getpid());
        if (const char* e = getenv(name)) {
            void* p = nullptr;
            if (sscanf(e, "%p", &p) == 1 && p) return static_cast<std::atomic<int>*>(p);
        }
        auto* a = new std::atomic<int>[64];
        for (int i = 0; i < 64; i++) a[i].store(0, std::memory_order_relaxed);
        snprintf(val, sizeof(val), "%p", static_cast<void*>(a));
        setenv(name, val, 1);
        return a;
    }();
    return reg;
}
int live_handles_on(int device) { return (device >= 0 && device < 64) ? live_registry()[device].load(std::memory_order_relaxed) : 2; }

extern "C" {

int ekfvio_create(const ekfvio_config* cfg, int device, void* stream, ekfvio_filter** out) {
    if (!out) return EKFVIO_EINVAL;
    *out = nullptr;
    if (!cfg || cfg->max_features < 0) return EKFVIO_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return EKFVIO_EDEVICE;
    ekfvio_filter* f = new ekfvio_filter();
    f->cfg = *cfg;
    f->device = device;
    if (device < 64) live_registry()[device].fetch_add(1, std::memory_order_relaxed);
    const int rc = create_body(f, cfg, device, stream);
    if (rc != EKFVIO_OK) {
        // a half-built handle never leaves the library: whatever was allocated is released here
        fprintf(stderr, "ekfvio_create: %s\n", f->last_error.c_str());
        (void)ekfvio_destroy(f);
        return rc;
    }
    *out = f;
    return EKFVIO_OK;
}

int ekfvio_destroy(ekfvio_filter* f) {
    if (!f) return EKFVIO_EINVAL;
    (void)hipSetDevice(f->device);
    if (f->stream) (void)hipStreamSynchronize(f->stream);
    void* ptrs[] = {f->mu, f->mu_next, f->last_klt, f->del_flag, f->P,  f->P2, f->FA, f->FB, f->FD,   f->Fdense,
                    f->idx, f->inv_idx, f->zmeas,  f->Rmeas,    f->pass,     f->yres, f->Rm, f->Saug,  f->Laug,  f->Linv, f->Lsign, f->Km, f->sweep_sync, f->sweep_dbg,
                    f->Wt,  f->Gm,     f->info,     f->seq_z,    f->seq_R, f->seq_pass};
    for (void* p : ptrs)
        if (p) hipFree(p);
    if (f->h_info) hipHostFree(f->h_info);
    if (f->h_out) hipHostFree(f->h_out);
    if (f->h_meas) hipHostFree(f->h_meas);
    if (f->d_meas) (void)hipFree(f->d_meas);
    klt_free(f);
    fast_free(f);
    drop_graph(f);
    if (f->ev0) hipEventDestroy(f->ev0);
    if (f->ev1) hipEventDestroy(f->ev1);
    if (f->own_stream && f->stream) hipStreamDestroy(f->stream);
    if (f->device >= 0 && f->device < 64) live_registry()[f->device].fetch_sub(1, std::memory_order_relaxed);
    delete f;
    return EKFVIO_OK;
}

const char* ekfvio_last_error(const ekfvio_filter* f) {
    return f ? f->last_error.c_str() : "null handle (a failed ekfvio_create leaves none behind: its return code says why)";
}

int ekfvio_reset(ekfvio_filter* f) {
    if (!f) return EKFVIO_EINVAL;
    f->out_fresh = false;
    HIPC(f, hipSetDevice(f->device));
    f->N = 0;
    f->n = EKF_BASE;
    HIPC(f, hipMemsetAsync(f->P, 0, sizeof(float) * (size_t)f->ldp * f->ldp, f->stream));
    HIPC(f, hipMemsetAsync(f->P2, 0, sizeof(float) * (size_t)f->ldp * f->ldp, f->stream));  // ping-pong partner: padding must be zero
    HIPC(f, hipMemsetAsync(f->mu, 0, sizeof(float) * f->ldp, f->stream));
    const float one = 1.f;
    HIPC(f, hipMemcpyAsync(f->mu + 3, &one, sizeof(float), hipMemcpyHostToDevice, f->stream));
    hipLaunchKernelGGL(init_sigma_kernel, dim3(1), dim3(32), 0, f->stream, f->P, f->ldp);
    HIPC(f, hipMemsetAsync(f->info, 0, 4 * sizeof(int), f->stream));
    HIPC(f, hipStreamSynchronize(f->stream));
    f->have_stamp = false;
    f->frames[0].valid = f->frames[1].valid = false;
    return EKFVIO_OK;
}

int ekfvio_add_features(ekfvio_filter* f, const float* uv, int32_t count) {
    if (!f || count < 0 || (count > 0 && !uv)) return EKFVIO_EINVAL;
    f->out_fresh = false;
    if (count == 0) return EKFVIO_OK;  // reference: early return (:59)
    if (f->N + count > f->cfg.max_features) return EKFVIO_ECAPACITY;
    HIPC(f, hipSetDevice(f->device));
    // stage uv through the (free between updates) zmeas buffer in chunks of max_features
    HIPC(f, hipMemcpyAsync(f->zmeas, uv, sizeof(float) * 2 * count, hipMemcpyHostToDevice, f->stream));
    return add_features_device(f, count);
}

int ekfvio_process(ekfvio_filter* f, float dt) {
    if (!f || !(dt >= 0.f)) return EKFVIO_EINVAL;  // ROS_ASSERT(dt >= 0) EKFVIO.cpp:162
    f->out_fresh = false;
    HIPC(f, hipSetDevice(f->device));
    launch_predict(f, dt);
    HIPC(f, hipGetLastError());
    return EKFVIO_OK;
}

int ekfvio_linearize(ekfvio_filter* f, float dt, float* F_dense) {
    if (!f || !F_dense) return EKFVIO_EINVAL;
    HIPC(f, hipSetDevice(f->device));
    launch_linearize(f, dt);
    launch_build_dense_F(f, f->Fdense);
    HIPC(f, hipMemcpy2DAsync(F_dense, sizeof(float) * f->n, f->Fdense, sizeof(float) * f->ldp, sizeof(float) * f->n,
                             f->n, hipMemcpyDeviceToHost, f->stream));
    HIPC(f, hipStreamSynchronize(f->stream));
    return EKFVIO_OK;
}

}  // extern "C"

// The status word f->info[0]: bit 0 = a non-positive pivot was met (a warning: the reference logs and carries on,
// TightlyCoupledEKF.cpp:577-580), bit 1 = the persistent sweep ran out of patience (its workgroups were not all
// resident: another process or stream held compute units).  Behind an aborted sweep the Joseph GEMMs write nothing, so
// Sigma, mu and the frame counter stand where process(dt) left them, and the handle stops using the persistent launch.
void sweep_abort_latch(ekfvio_filter* f) {
    f->sweep_mode = 0;  // the following sweeps of this handle: one launch per block step (no co-residency needed)
    drop_graph(f);      // captured steps contain the persistent launch
    // ... and a way back: what kept the workgroups from being resident -- another process or stream on the device, a
    // compute-unit mask -- may be gone later.  The persistent launch is tried again after a pause that doubles with every abort
    // (2 s, 4 s, ... 64 s), checked at the entry points that enqueue updates (sweep_maybe_retry).
    f->sweep_retry_pause_s = f->sweep_retry_pause_s <= 0 ? f->sweep_retry_first_s : std::min(64.0, 2.0 * f->sweep_retry_pause_s);
    f->sweep_retry_at = std::chrono::steady_clock::now() + std::chrono::milliseconds((long long)(1e3 * f->sweep_retry_pause_s));
    f->sweep_retry_armed = true;
}
// Called ONLY from the entry points that can run an aborted update again (ekfvio_update, ekfvio_step_image): a retry that fails costs
// one recovery there, nothing else.
void sweep_maybe_retry(ekfvio_filter* f) {
    if (!f->sweep_retry_armed || f->sweep_mode != 0 || std::chrono::steady_clock::now() < f->sweep_retry_at) return;
    f->sweep_retry_armed = false;
    f->sweep_mode = 2;
    f->sweep_probation = 8;  // clean persistent sweeps until the pause starts from its first value again (sweep_clean_update)
    drop_graph(f);  // captured steps contain the per-step sweep
    // the flags AND the abort word (which no kernel ever zeroes) start from zero again
    (void)hipMemsetAsync(f->sweep_sync, 0, sizeof(int) * f->sweep_sync_words, f->stream);
    f->sweep_flags_clean = false;
}
// An update whose persistent sweep came through: after a few of them in a row behind a retry the pause of the NEXT abort starts from
// its first value again (it only ever doubled, so a handle that met a few transient aborts early waited 64 s on the slow
// sweep after any later one, even after hours of clean persistent sweeps).
void sweep_clean_update(ekfvio_filter* f) {
    if (f->sweep_probation > 0 && f->sweep_mode == 2 && --f->sweep_probation == 0) f->sweep_retry_pause_s = 0.0;
}
// Waits for the stream and reads the status word.  An aborted update is enqueued again by `rerun` (null: the caller cannot,
// e.g. a graph replay of many steps: EKFVIO_EABORTED) with the per-step sweep and awaited: fresh launches, same inputs,
// and the state comes out as if the per-step sweep had run in the first place.
int finish_update_rerun(ekfvio_filter* f, void (*rerun)(ekfvio_filter*, void*), void* ctx, int published_seq) {
    int bad = 0;
    int rc = published_seq ? poll_status(f, published_seq, &bad, nullptr) : wait_status(f, &bad);  // (published_seq: the caller's launches publish it)
    if (rc != EKFVIO_OK) return rc;
    if (bad) HIPC(f, hipMemsetAsync(f->info, 0, sizeof(int), f->stream));
    if (bad & 2) {
        sweep_abort_latch(f);
        if (!rerun) {
            f->last_error = "persistent sweep aborted (compute units shared with other work): updates behind it were skipped, the state is "
                            "the propagated one; this handle now takes one launch per block step";
            return EKFVIO_EABORTED;
        }
        f->sweep_recoveries++;
        rerun(f, ctx);
        HIPC(f, hipGetLastError());
        rc = wait_status(f, &bad);
        if (rc != EKFVIO_OK) return rc;
        if (bad) HIPC(f, hipMemsetAsync(f->info, 0, sizeof(int), f->stream));
        if (bad & 2) return EKFVIO_EABORTED;  // (cannot happen: the per-step sweep has no waits)
    } else if (rerun) sweep_clean_update(f);
    return (bad & 1) ? EKFVIO_ENUMERIC : EKFVIO_OK;
}

extern "C" {

static int finish_update(ekfvio_filter* f) { return finish_update_rerun(f, nullptr, nullptr, 0); }

int ekfvio_update(ekfvio_filter* f, const float* z, const float* R, const uint8_t* pass, int32_t count) {
    if (!f || count != f->N) return EKFVIO_EINVAL;  // ROS_ASSERT :478
    f->out_fresh = false;
    if (count > 0 && (!z || !R || !pass)) return EKFVIO_EINVAL;
    HIPC(f, hipSetDevice(f->device));
    sweep_maybe_retry(f);
    const int m = count_rows(pass, count);
    // the frame's three host arrays travel as ONE copy from pinned memory (three pageable copies cost ~25 us per step);
    // the previous call's synchronisation (finish_update) guarantees the staging buffer is free
    const size_t cap = (size_t)(f->cfg.max_features > 0 ? f->cfg.max_features : 1);
    float* dz = reinterpret_cast<float*>(f->d_meas);
    float* dR = reinterpret_cast<float*>(f->d_meas + 8 * cap);
    uint8_t* dp = f->d_meas + 24 * cap;
    if (count > 0) {
        memcpy(f->h_meas, z, sizeof(float) * 2 * count);
        memcpy(f->h_meas + 8 * cap, R, sizeof(float) * 4 * count);
        memcpy(f->h_meas + 24 * cap, pass, count);
        HIPC(f, hipMemcpyAsync(f->d_meas, f->h_meas, 24 * cap + count, hipMemcpyHostToDevice, f->stream));
    }
    // The status of an update -- a non-positive pivot, a sweep that gave up -- is final when the Cholesky sweep is: the status word goes to
    // the host right behind the sweep, and the call returns while the two Joseph GEMMs are still running (everything the caller can do next
    // is ordered behind them on the handle's stream, or synchronises).  A step-by-step caller then keeps the GPU fed: the next step's
    // launches are in the queue before this step's last kernel ends (EKFVIO_EARLY_STATUS=0: publish behind the last kernel, as rounds 1-3).
    const bool early = f->early_status != 0;  // (per handle, read at create like every other knob: ADVICE r04)
    const int seq = next_status_seq(f);
    f->publish_after_sweep_seq = early ? seq : 0;
    launch_update(f, m, dz, dR, dp);
    if (f->publish_after_sweep_seq != 0 || !early) launch_publish_status(f, seq);  // (no sweep ran: nothing was published yet)
    f->publish_after_sweep_seq = 0;
    HIPC(f, hipGetLastError());
    struct Ctx { int m; float *dz, *dR; uint8_t* dp; } ctx{m, dz, dR, dp};
    return finish_update_rerun(f, [](ekfvio_filter* g, void* c) {
        const Ctx* x = static_cast<const Ctx*>(c);
        launch_update(g, x->m, x->dz, x->dR, x->dp);  // the staged measurement is still in d_meas; the bookkeeping is idempotent
    }, &ctx, seq);
}

int ekfvio_measurement_map(const ekfvio_filter* f, const uint8_t* measured, int32_t count, int32_t* idx, int32_t* rows) {
    if (!f || !measured || !idx || !rows || count != f->N) return EKFVIO_EINVAL;  // ROS_ASSERT :636
    int r = 0;
    for (int i = 0; i < count; i++)
        if (measured[i]) {
            idx[r++] = EKF_BASE + 3 * i;
            idx[r++] = EKF_BASE + 3 * i + 1;
        }
    *rows = r;
    return EKFVIO_OK;
}

int ekfvio_num_features(const ekfvio_filter* f) { return f ? f->N : -1; }
int ekfvio_dim(const ekfvio_filter* f) { return f ? f->n : -1; }

int ekfvio_get_base_mu(ekfvio_filter* f, float* base_mu) {
    if (!f || !base_mu) return EKFVIO_EINVAL;
    if (f->out_fresh) {  // written with the frame's status word (ekfvio_step_image): nothing to launch, nothing to wait for
        memcpy(base_mu, f->h_out, sizeof(float) * EKF_BASE);
        return EKFVIO_OK;
    }
    HIPC(f, hipSetDevice(f->device));
    // 22 floats: written by a kernel straight into pinned host memory and awaited like the status word (the node reads
    // its odometry every frame: a device-to-host copy into pageable memory plus a synchronise cost three times as much)
    const int seq = next_status_seq(f);
    hipLaunchKernelGGL(copy_small_kernel, dim3(1), dim3(64), 0, f->stream, f->mu, f->d_out, EKF_BASE, f->info, f->d_hinfo, seq);
    int bad = 0;
    const int rc = poll_status(f, seq, &bad, nullptr);
    if (rc != EKFVIO_OK) return rc;
    memcpy(base_mu, f->h_out, sizeof(float) * EKF_BASE);
    return EKFVIO_OK;
}

int ekfvio_get_features(ekfvio_filter* f, float* mu3N, float* last_klt2N, uint8_t* delN) {
    if (!f) return EKFVIO_EINVAL;
    HIPC(f, hipSetDevice(f->device));
    if (f->N > 0) {
        if (mu3N) HIPC(f, hipMemcpyAsync(mu3N, f->mu + EKF_BASE, sizeof(float) * 3 * f->N, hipMemcpyDeviceToHost, f->stream));
        if (last_klt2N) HIPC(f, hipMemcpyAsync(last_klt2N, f->last_klt, sizeof(float) * 2 * f->N, hipMemcpyDeviceToHost, f->stream));
        if (delN) HIPC(f, hipMemcpyAsync(delN, f->del_flag, f->N, hipMemcpyDeviceToHost, f->stream));
    }
    HIPC(f, hipStreamSynchronize(f->stream));
    return EKFVIO_OK;
}

int ekfvio_get_sigma(ekfvio_filter* f, float* sigma, int32_t ld) {
    if (!f || !sigma || ld < f->n) return EKFVIO_EINVAL;
    HIPC(f, hipSetDevice(f->device));
    HIPC(f, hipMemcpy2DAsync(sigma, sizeof(float) * ld, f->P, sizeof(float) * f->ldp, sizeof(float) * f->n, f->n,
                             hipMemcpyDeviceToHost, f->stream));
    HIPC(f, hipStreamSynchronize(f->stream));
    return EKFVIO_OK;
}

int ekfvio_get_feature_cov(ekfvio_filter* f, int32_t index, float cov[4]) {
    if (!f || !cov || index < 0 || index >= f->N) return EKFVIO_EINVAL;
    HIPC(f, hipSetDevice(f->device));
    const int s = EKF_BASE + 3 * index;  // Sigma.block(start,start,2,2), column-major out
    HIPC(f, hipMemcpy2DAsync(cov, sizeof(float) * 2, f->P + (size_t)s * f->ldp + s, sizeof(float) * f->ldp,
                             sizeof(float) * 2, 2, hipMemcpyDeviceToHost, f->stream));
    HIPC(f, hipStreamSynchronize(f->stream));
    return EKFVIO_OK;
}

int ekfvio_get_depth_variance(ekfvio_filter* f, int32_t index, float* var) {
    if (!f || !var || index < 0 || index >= f->N) return EKFVIO_EINVAL;
    HIPC(f, hipSetDevice(f->device));
    const int s = EKF_BASE + 3 * index + 2;
    HIPC(f, hipMemcpyAsync(var, f->P + (size_t)s * f->ldp + s, sizeof(float), hipMemcpyDeviceToHost, f->stream));
    HIPC(f, hipStreamSynchronize(f->stream));
    return EKFVIO_OK;
}

// setFeatureHomogenousCovariance (TightlyCoupledEKF.cpp:668-676): the four coeffRef writes of the (u,v) block
int ekfvio_set_feature_cov(ekfvio_filter* f, int32_t index, const float cov[4]) {
    if (!f || !cov || index < 0 || index >= f->N) return EKFVIO_EINVAL;
    HIPC(f, hipSetDevice(f->device));
    const int s = EKF_BASE + 3 * index;  // column-major in, column-major block of Sigma out
    HIPC(f, hipMemcpy2DAsync(f->P + (size_t)s * f->ldp + s, sizeof(float) * f->ldp, cov, sizeof(float) * 2, sizeof(float) * 2, 2,
                             hipMemcpyHostToDevice, f->stream));
    HIPC(f, hipStreamSynchronize(f->stream));  // cov is the caller's memory
    return EKFVIO_OK;
}

// getMetric2PixelMap / getPixel2MetricMap (TightlyCoupledEKF.cpp:683-697): J = diag(K(0,0), K(1,1)) and
// diag(1.0f / K(0,0), 1.0f / K(1,1)) -- proper 2-D indexing of K here (unlike Feature.h:60-66), no state involved.
int ekfvio_metric2pixel_map(const float K[9], float J[4]) {
    if (!K || !J) return EKFVIO_EINVAL;
    J[0] = K[0]; J[1] = 0.f; J[2] = 0.f; J[3] = K[4];
    return EKFVIO_OK;
}
int ekfvio_pixel2metric_map(const float K[9], float J[4]) {
    if (!K || !J) return EKFVIO_EINVAL;
    J[0] = 1.0f / K[0]; J[1] = 0.f; J[2] = 0.f; J[3] = 1.0f / K[4];
    return EKFVIO_OK;
}

// publishOdometry's payload (EKFVIO.cpp:444-477): the slices of base_mu the message is filled from
int ekfvio_get_odometry(ekfvio_filter* f, float position[3], float orientation_wxyz[4], float linear[3], float angular[3]) {
    if (!f) return EKFVIO_EINVAL;
    float b[EKF_BASE];
    const int rc = ekfvio_get_base_mu(f, b);
    if (rc != EKFVIO_OK) return rc;
    for (int i = 0; i < 3; i++) {
        if (position) position[i] = b[i];
        if (linear) linear[i] = b[7 + i];
        if (angular) angular[i] = b[10 + i];
    }
    if (orientation_wxyz)
        for (int i = 0; i < 4; i++) orientation_wxyz[i] = b[3 + i];
    return EKFVIO_OK;
}

int ekfvio_check_sigma(ekfvio_filter* f, float* min_diag, float* max_asym) {
    if (!f || !min_diag || !max_asym) return EKFVIO_EINVAL;
    HIPC(f, hipSetDevice(f->device));
    launch_check_sigma(f, f->yres);  // yres is free between updates
    float out[2];
    HIPC(f, hipMemcpyAsync(out, f->yres, 2 * sizeof(float), hipMemcpyDeviceToHost, f->stream));
    HIPC(f, hipStreamSynchronize(f->stream));
    *min_diag = out[0];
    *max_asym = out[1];
    return EKFVIO_OK;
}

int ekfvio_set_state(ekfvio_filter* f, int32_t N, const float* base_mu, const float* mu3N, const float* last_klt2N,
                     const uint8_t* delN, const float* sigma, int32_t ld) {
    if (!f || N < 0 || !base_mu || !sigma) return EKFVIO_EINVAL;
    f->out_fresh = false;
    if (N > f->cfg.max_features) return EKFVIO_ECAPACITY;
    const int n = EKF_BASE + 3 * N;
    if (ld < n || (N > 0 && (!mu3N || !last_klt2N || !delN))) return EKFVIO_EINVAL;
    HIPC(f, hipSetDevice(f->device));
    HIPC(f, hipMemsetAsync(f->P, 0, sizeof(float) * (size_t)f->ldp * f->ldp, f->stream));
    HIPC(f, hipMemsetAsync(f->P2, 0, sizeof(float) * (size_t)f->ldp * f->ldp, f->stream));
    HIPC(f, hipMemsetAsync(f->mu, 0, sizeof(float) * f->ldp, f->stream));
    HIPC(f, hipMemcpyAsync(f->mu, base_mu, sizeof(float) * EKF_BASE, hipMemcpyHostToDevice, f->stream));
    if (N > 0) {
        HIPC(f, hipMemcpyAsync(f->mu + EKF_BASE, mu3N, sizeof(float) * 3 * N, hipMemcpyHostToDevice, f->stream));
        HIPC(f, hipMemcpyAsync(f->last_klt, last_klt2N, sizeof(float) * 2 * N, hipMemcpyHostToDevice, f->stream));
        HIPC(f, hipMemcpyAsync(f->del_flag, delN, N, hipMemcpyHostToDevice, f->stream));
    }
    HIPC(f, hipMemcpy2DAsync(f->P, sizeof(float) * f->ldp, sigma, sizeof(float) * ld, sizeof(float) * n, n,
                             hipMemcpyHostToDevice, f->stream));
    HIPC(f, hipStreamSynchronize(f->stream));
    f->N = N;
    f->n = n;
    return EKFVIO_OK;
}

int ekfvio_imu_update(ekfvio_filter* f, const float* gyro, const float* accel) {
    if (!f || !gyro || !accel) return EKFVIO_EINVAL;
    f->out_fresh = false;
    HIPC(f, hipSetDevice(f->device));
    launch_imu_update(f, gyro, accel);
    HIPC(f, hipGetLastError());
    return EKFVIO_OK;
}

int ekfvio_imu(ekfvio_filter* f, double stamp, const float* gyro, const float* accel) {
    if (!f) return EKFVIO_EINVAL;
    f->out_fresh = false;
    if (!f->cfg.use_imu) return EKFVIO_OK;  // the reference's callback only logs
    if (!gyro || !accel) return EKFVIO_EINVAL;
    if (!f->have_stamp) {  // tc_ekf.t is set by the first message (EKFVIO.cpp:148-150 does it for the first frame)
        f->t_stamp = stamp;
        f->have_stamp = true;
        return EKFVIO_OK;
    }
    const double dt = stamp - f->t_stamp;
    if (!(dt >= 0)) return EKFVIO_EINVAL;
    HIPC(f, hipSetDevice(f->device));
    launch_predict(f, (float)dt);
    f->t_stamp = stamp;
    launch_imu_update(f, gyro, accel);
    HIPC(f, hipGetLastError());
    return EKFVIO_OK;
}

// ---- uploaded measurement sequences ---------------------------------------------------
int ekfvio_upload_measurements(ekfvio_filter* f, int32_t frames, const float* z, const float* R, const uint8_t* pass) {
    if (!f || frames <= 0 || !z || !R || !pass || f->N <= 0) return EKFVIO_EINVAL;
    HIPC(f, hipSetDevice(f->device));
    HIPC(f, hipStreamSynchronize(f->stream));
    drop_graph(f);  // the captured launches hold pointers into the old sequence buffers
    if (f->seq_z) hipFree(f->seq_z);
    if (f->seq_R) hipFree(f->seq_R);
    if (f->seq_pass) hipFree(f->seq_pass);
    f->seq_z = f->seq_R = nullptr;
    f->seq_pass = nullptr;
    const size_t N = f->N;
    HIPC(f, dev_alloc(f->stream, &f->seq_z, frames * 2 * N));
    HIPC(f, dev_alloc(f->stream, &f->seq_R, frames * 4 * N));
    HIPC(f, dev_alloc(f->stream, &f->seq_pass, frames * N));
    HIPC(f, hipMemcpyAsync(f->seq_z, z, sizeof(float) * frames * 2 * N, hipMemcpyHostToDevice, f->stream));
    HIPC(f, hipMemcpyAsync(f->seq_R, R, sizeof(float) * frames * 4 * N, hipMemcpyHostToDevice, f->stream));
    HIPC(f, hipMemcpyAsync(f->seq_pass, pass, frames * N, hipMemcpyHostToDevice, f->stream));
    HIPC(f, hipStreamSynchronize(f->stream));
    f->seq_frames = frames;
    f->seq_N = (int)N;
    f->seq_m.resize(frames);
    for (int i = 0; i < frames; i++) f->seq_m[i] = count_rows(pass + (size_t)i * N, (int)N);
    return EKFVIO_OK;
}

// steps per captured graph: even (the mean / covariance ping-pong is back in its starting orientation)
// and large enough to amortise the ~8 us a hipGraphLaunch costs between replays
#ifndef EKF_GRAPH_STEPS
#define EKF_GRAPH_STEPS 8
#endif
#define EKF_GRAPH_STEPS_BIG 32  // second, longer graph for runs of at least two of them (+0.6 % steps/s at N=256)

static void drop_graph(ekfvio_filter* f) {
    if (f->step_graph) {
        (void)hipGraphExecDestroy(f->step_graph);
        f->step_graph = nullptr;
    }
    if (f->step_graph_big) {
        (void)hipGraphExecDestroy(f->step_graph_big);
        f->step_graph_big = nullptr;
    }
    if (f->step_graph_pair) {
        (void)hipGraphExecDestroy(f->step_graph_pair);
        f->step_graph_pair = nullptr;
    }
}

// Captures `steps` filter steps (process + update with m measurement rows each, bookkeeping driven by the device-side
// frame counter) into an executable graph.  `steps` is even: the mean / covariance ping-pong nets to zero.
static int capture_steps(ekfvio_filter* f, int steps, int m, float dt, int* counter, hipGraphExec_t* out) {
    hipGraph_t g = nullptr;
    HIPC(f, hipStreamSynchronize(f->stream));
    // sweep_flags_clean mirrors DEVICE state on the host, and a capture records launches without running them: the
    // graph must not inherit the mirror's value of the moment it was captured at -- its first sweep always zeroes the flags itself (one
    // memset node per replay) -- and the mirror must come out of the capture as it went in.  What a replay leaves behind is recorded
    // separately (graph_leaves_flags_clean) and applied after every hipGraphLaunch.
    const bool mirror = f->sweep_flags_clean;
    f->sweep_flags_clean = false;
    HIPC(f, hipStreamBeginCapture(f->stream, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < steps; k++) {
        // the frame's measurement bookkeeping rides in the process(dt) launch
        const BookArgs bk = make_book_args(f, m, f->seq_z, f->seq_R, f->seq_pass, counter);
        launch_predict(f, dt, &bk);
        // round 6: inside a graph the next step's dt is known -- every update but the graph's last lets its last GEMM linearise for the step
        // behind it (launch_update decides whether the shape allows it; the graph's last update leaves the next replay's first step to
        // linearise for itself, so a graph needs nothing from whatever ran before it)
        f->lin_next_dt = (k + 1 < steps) ? dt : -1.f;
        launch_update(f, m, f->seq_z, f->seq_R, f->seq_pass, counter, f->seq_frames, true);
        f->lin_next_dt = -1.f;
    }
    f->prelinearized = false;
    hipError_t ce = hipStreamEndCapture(f->stream, &g);
    f->graph_leaves_flags_clean = f->sweep_flags_clean;  // (the same for every step count: the last update's last GEMM decides)
    f->sweep_flags_clean = mirror;
    if (ce != hipSuccess || !g) {
        f->last_error = std::string("graph capture: ") + hipGetErrorString(ce);
        return EKFVIO_EDEVICE;
    }
    HIPC(f, hipGraphInstantiate(out, g, nullptr, nullptr, 0));
    (void)hipGraphDestroy(g);
    return EKFVIO_OK;
}

int ekfvio_run_uploaded(ekfvio_filter* f, int32_t first, int32_t count, float dt) {
    if (!f || f->seq_frames <= 0 || f->seq_N != f->N || count < 0 || first < 0 || !(dt >= 0.f)) return EKFVIO_EINVAL;
    f->out_fresh = false;
    HIPC(f, hipSetDevice(f->device));
    // (no sweep_maybe_retry here -- ADVICE r05: a device-resident run cannot run an aborted update again, so a handle whose persistent
    // sweep once gave up keeps the per-step sweep for its graph replays until ekfvio_update / ekfvio_step_image, which CAN recover,
    // have tried the persistent launch again and it has come through; with a permanent obstacle (a compute-unit mask, a co-tenant
    // process) a retry from here would skip a replay's remaining updates every 2 .. 64 s for the handle's whole life)
    const size_t N = f->N;
    int s = 0;
    struct Unrec {  // the launches and captures of this call carry the long wait bound (common.h, sweep_wait_ticks): nothing can run their updates again
        ekfvio_filter* f;
        explicit Unrec(ekfvio_filter* f_) : f(f_) { f->sweep_unrecoverable = true; }
        ~Unrec() { f->sweep_unrecoverable = false; }
    } unrec(f);
    // hipGraph path: the same measurement-row count for every frame (one launch geometry),
    // profiling off.  The bookkeeping kernel reads the frame index from a device counter.
    // (count == 0 only prepares: the graphs are captured, nothing runs — callers that time a run call this first)
    bool uniform = f->use_graph && !f->prof_on && (count >= 2 || count == 0);
    for (int i = 0; uniform && i < f->seq_frames; i++) uniform = f->seq_m[i] == f->seq_m[0];
    if (uniform) {
        const int m = f->seq_m[0];
        int* counter = f->info + 1;
        // by value (a copy from a host scalar could be overtaken by the next call's write to that scalar)
        HIPC(f, hipMemsetD32Async((hipDeviceptr_t)counter, first % f->seq_frames, 1, f->stream));
        if (!f->step_graph || f->graph_N != f->N || f->graph_m != m || f->graph_dt != dt || f->graph_mu != f->mu || f->graph_P != f->P ||
            f->graph_seq != f->seq_z || f->graph_frames != f->seq_frames || f->graph_sole != (live_handles_on(f->device) <= 1)) {
            drop_graph(f);
            int rc = capture_steps(f, EKF_GRAPH_STEPS, m, dt, counter, &f->step_graph);
            if (rc != EKFVIO_OK) return rc;
            rc = capture_steps(f, 2, m, dt, counter, &f->step_graph_pair);
            if (rc != EKFVIO_OK) return rc;
            f->graph_N = f->N; f->graph_m = m; f->graph_dt = dt; f->graph_mu = f->mu; f->graph_P = f->P; f->graph_seq = f->seq_z;
            f->graph_frames = f->seq_frames;
            f->graph_sole = live_handles_on(f->device) <= 1;  // (decides which sweep the captured steps contain)
            // the captured launches did not execute: the pointer swaps of launch_predict netted to zero
        }
        // captured together with the short graph (i.e. in a caller's warm-up call) whenever the uploaded sequence is
        // long enough to make runs of that length plausible, never lazily inside a long run
        if (f->seq_frames >= 2 * EKF_GRAPH_STEPS_BIG && !f->step_graph_big) {
            int rc = capture_steps(f, EKF_GRAPH_STEPS_BIG, m, dt, counter, &f->step_graph_big);
            if (rc != EKFVIO_OK) return rc;
        }
        if (f->step_graph_big)
            for (; s + EKF_GRAPH_STEPS_BIG <= count; s += EKF_GRAPH_STEPS_BIG) HIPC(f, hipGraphLaunch(f->step_graph_big, f->stream));
        for (; s + EKF_GRAPH_STEPS <= count; s += EKF_GRAPH_STEPS) HIPC(f, hipGraphLaunch(f->step_graph, f->stream));
        for (; s + 2 <= count; s += 2) HIPC(f, hipGraphLaunch(f->step_graph_pair, f->stream));
        if (s > 0) f->sweep_flags_clean = f->graph_leaves_flags_clean;  // what the last replayed step left in device memory
        for (; s < count; s++) {  // an odd last step, eager, same counter-driven bookkeeping
            launch_predict(f, dt);
            launch_update(f, m, f->seq_z, f->seq_R, f->seq_pass, counter, f->seq_frames);
        }
        HIPC(f, hipGetLastError());
        return EKFVIO_OK;
    }
    for (; s < count; s++) {
        const int i = (first + s) % f->seq_frames;
        launch_predict(f, dt);
        launch_update(f, f->seq_m[i], f->seq_z + i * 2 * N, f->seq_R + i * 4 * N, f->seq_pass + i * N);
    }
    HIPC(f, hipGetLastError());
    return EKFVIO_OK;
}

int ekfvio_synchronize(ekfvio_filter* f) {
    if (!f) return EKFVIO_EINVAL;
    HIPC(f, hipSetDevice(f->device));
    // a non-positive pivot met by an asynchronous run (ekfvio_run_uploaded) surfaces here, once
    return finish_update(f);
}

// ---- instrumentation ------------------------------------------------------------------
int ekfvio_profile_enable(ekfvio_filter* f, int32_t on) {
    if (!f) return EKFVIO_EINVAL;
    hipStreamSynchronize(f->stream);
    f->prof_on = on != 0;
    if (f->prof_on) {
        // calibrate the cost of an (ev0, ev1) pair with nothing in between
        double acc = 0;
        const int reps = 32;
        for (int i = 0; i < reps + 4; i++) {
            (void)hipEventRecord(f->ev0, f->stream);
            (void)hipEventRecord(f->ev1, f->stream);
            (void)hipEventSynchronize(f->ev1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, f->ev0, f->ev1);
            if (i >= 4) acc += ms;
        }
        f->prof_overhead_ms = (float)(acc / reps);
    }
    return EKFVIO_OK;
}
// Mean launch duration of the two P-update GEMMs at the shape of the most recent update, measured the way the
// timed region runs them: `reps` pairs captured into one hipGraph (back-to-back launches on the handle's stream),
// replayed once between two HIP events.  Outputs go to scratch; the filter state is unchanged.
int ekfvio_profile_update_gemms(ekfvio_filter* f, int32_t reps, double* avg_launch_us, double* flops_per_launch) {
    if (!f || reps <= 0 || !avg_launch_us) return EKFVIO_EINVAL;
    if (f->last_m <= 0 || f->N <= 0) return EKFVIO_ESTATE;
    HIPC(f, hipSetDevice(f->device));
    HIPC(f, hipStreamSynchronize(f->stream));
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    HIPC(f, hipStreamBeginCapture(f->stream, hipStreamCaptureModeThreadLocal));
    const int per_rep = launch_update_gemms_scratch(f, f->last_m, reps);
    hipError_t ce = hipStreamEndCapture(f->stream, &g);
    if (ce != hipSuccess || !g) {
        f->last_error = std::string("graph capture: ") + hipGetErrorString(ce);
        return EKFVIO_EDEVICE;
    }
    HIPC(f, hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    (void)hipGraphDestroy(g);
    HIPC(f, hipGraphLaunch(ge, f->stream));  // warm-up
    HIPC(f, hipEventRecord(f->ev0, f->stream));
    HIPC(f, hipGraphLaunch(ge, f->stream));
    HIPC(f, hipEventRecord(f->ev1, f->stream));
    HIPC(f, hipEventSynchronize(f->ev1));
    float ms = 0;
    HIPC(f, hipEventElapsedTime(&ms, f->ev0, f->ev1));
    (void)hipGraphExecDestroy(ge);
    // P2 is the predict's scratch and must stay zero outside the live block
    HIPC(f, hipMemsetAsync(f->P2, 0, sizeof(float) * (size_t)f->ldp * f->ldp, f->stream));
    HIPC(f, hipStreamSynchronize(f->stream));
    *avg_launch_us = 1e3 * ms / ((double)per_rep * reps);
    const int m_pad = round_up(f->last_m, 64);
    if (flops_per_launch) {
        // EXECUTED flops, averaged over the launches of one update: the second Joseph GEMM of the throughput regime forms the lower triangle's
        // 64 x 64 tiles only (GemmEpi::sym)
        const double full = 2.0 * f->n * (double)f->n * m_pad;
        const int tn = (f->n + 63) / 64;
        const bool half = per_rep == 2 && f->sym_joseph && gemm_throughput_regime(f, f->n, f->n, m_pad);
        const double second = half ? 0.5 * tn * (tn + 1) * 2.0 * 64.0 * 64.0 * m_pad : full;
        *flops_per_launch = per_rep == 2 ? 0.5 * (full + second) : full;
    }
    return EKFVIO_OK;
}
int ekfvio_profile_reset(ekfvio_filter* f) {
    if (!f) return EKFVIO_EINVAL;
    for (int i = 0; i < PC_COUNT; i++) f->prof[i] = ProfSlot();
    return EKFVIO_OK;
}
int ekfvio_profile_count(void) { return PC_COUNT; }
const char* ekfvio_profile_name(int32_t cls) { return (cls >= 0 && cls < PC_COUNT) ? kProfNames[cls] : ""; }
int ekfvio_profile_get(ekfvio_filter* f, int32_t cls, double* total_ms, int64_t* launches, double* flops) {
    if (!f || cls < 0 || cls >= PC_COUNT) return EKFVIO_EINVAL;
    if (total_ms) *total_ms = f->prof[cls].ms;
    if (launches) *launches = f->prof[cls].launches;
    if (flops) *flops = f->prof[cls].flops;
    return EKFVIO_OK;
}

int ekfvio_get_counters(ekfvio_filter* f, int64_t counters[8]) {
    if (!f || !counters) return EKFVIO_EINVAL;
    counters[0] = f->persistent_sweeps;
    counters[1] = f->schur_sweeps;
    counters[2] = f->sweep_recoveries;
    counters[3] = f->sweep_mode;
    counters[4] = f->early_output_frames;
    counters[5] = f->t2_updates;
    counters[6] = counters[7] = 0;
    return EKFVIO_OK;
}

#ifdef EKFVIO_TEST_HOOKS  // include/ekfvio_test_hooks.h: only in libekfvio_hip_hooks.so
// ---- raw kernels for unit tests -------------------------------------------------------
int ekfvio_test_gemm(ekfvio_filter* f, int32_t transB, int32_t M, int32_t N, int32_t K, float alpha, const float* A,
                     int32_t lda, const float* B, int32_t ldb, float beta, float* C, int32_t ldc, int32_t variant) {
    if (!f || M <= 0 || N <= 0 || K <= 0 || !A || !B || !C) return EKFVIO_EINVAL;
    HIPC(f, hipSetDevice(f->device));
    // device copies padded to the kernels' contract (64-row/col slack, K to 64, zero fill);
    // packed on the host so that only flat copies are issued
    const int Mp = round_up(M, 64), Np = round_up(N, 64), Kp = round_up(K, 64);
    const int brows = transB ? Np : Kp, bcols = transB ? Kp : Np;
    std::vector<float> hA((size_t)Mp * Kp, 0.f), hB((size_t)brows * bcols, 0.f), hC((size_t)Mp * Np, 0.f);
    for (int k = 0; k < K; k++)
        for (int i = 0; i < M; i++) hA[(size_t)k * Mp + i] = A[(size_t)k * lda + i];
    for (int c = 0; c < (transB ? K : N); c++)
        for (int r = 0; r < (transB ? N : K); r++) hB[(size_t)c * brows + r] = B[(size_t)c * ldb + r];
    for (int j = 0; j < N; j++)
        for (int i = 0; i < M; i++) hC[(size_t)j * Mp + i] = C[(size_t)j * ldc + i];
    float *dA, *dB, *dC;
    HIPC(f, dev_alloc(f->stream, &dA, hA.size()));
    HIPC(f, dev_alloc(f->stream, &dB, hB.size()));
    HIPC(f, dev_alloc(f->stream, &dC, hC.size()));
    HIPC(f, hipMemcpyAsync(dA, hA.data(), sizeof(float) * hA.size(), hipMemcpyHostToDevice, f->stream));
    HIPC(f, hipMemcpyAsync(dB, hB.data(), sizeof(float) * hB.size(), hipMemcpyHostToDevice, f->stream));
    HIPC(f, hipMemcpyAsync(dC, hC.data(), sizeof(float) * hC.size(), hipMemcpyHostToDevice, f->stream));
    launch_gemm_variant(f, variant, transB, M, N, Kp, alpha, dA, Mp, dB, brows, beta, dC, Mp, dC, Mp, 0, 0);
    HIPC(f, hipMemcpyAsync(hC.data(), dC, sizeof(float) * hC.size(), hipMemcpyDeviceToHost, f->stream));
    HIPC(f, hipStreamSynchronize(f->stream));
    for (int j = 0; j < N; j++)
        for (int i = 0; i < M; i++) C[(size_t)j * ldc + i] = hC[(size_t)j * Mp + i];
    hipFree(dA);
    hipFree(dB);
    hipFree(dC);
    return EKFVIO_OK;
}

// Times `reps` back-to-back launches of the GEMM at one shape (operands resident, random
// fill); returns the mean launch-to-launch time in microseconds.  variant selects a tile
// configuration (0 = production).
int ekfvio_test_gemm_bench(ekfvio_filter* f, int32_t transB, int32_t lowerB, int32_t M, int32_t N, int32_t K,
                           int32_t reps, int32_t variant, double* mean_us) {
    if (!f || M <= 0 || N <= 0 || K <= 0 || reps <= 0 || !mean_us) return EKFVIO_EINVAL;
    HIPC(f, hipSetDevice(f->device));
    const int Mp = round_up(M, 128), Np = round_up(N, 128), Kp = round_up(K, 64);
    const int brows = transB ? Np : Kp, bcols = transB ? Kp : Np;
    std::vector<float> hA((size_t)Mp * Kp), hB((size_t)brows * bcols), hC((size_t)Mp * Np);
    uint32_t st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : hA) v = rnd();
    for (auto& v : hB) v = rnd();
    for (auto& v : hC) v = rnd();
    float *dA, *dB, *dC;
    HIPC(f, dev_alloc(f->stream, &dA, hA.size()));
    HIPC(f, dev_alloc(f->stream, &dB, hB.size()));
    HIPC(f, dev_alloc(f->stream, &dC, hC.size()));
    HIPC(f, hipMemcpyAsync(dA, hA.data(), sizeof(float) * hA.size(), hipMemcpyHostToDevice, f->stream));
    HIPC(f, hipMemcpyAsync(dB, hB.data(), sizeof(float) * hB.size(), hipMemcpyHostToDevice, f->stream));
    HIPC(f, hipMemcpyAsync(dC, hC.data(), sizeof(float) * hC.size(), hipMemcpyHostToDevice, f->stream));
    const bool stamped = variant >= 10000;  // library built with -DEKF_GEMM_STAMPS
    variant %= 10000;
    for (int w = 0; w < 3; w++)
        launch_gemm_variant(f, variant, transB, M, N, Kp, -1e-3f, dA, Mp, dB, brows, 1.f, dC, Mp, dC, Mp, 0, lowerB);
    HIPC(f, hipEventRecord(f->ev0, f->stream));
    for (int r = 0; r < reps; r++)
        launch_gemm_variant(f, variant, transB, M, N, Kp, -1e-3f, dA, Mp, dB, brows, 1.f, dC, Mp, dC, Mp, 0, lowerB);
    HIPC(f, hipEventRecord(f->ev1, f->stream));
    HIPC(f, hipEventSynchronize(f->ev1));
    float ms = 0;
    HIPC(f, hipEventElapsedTime(&ms, f->ev0, f->ev1));
    *mean_us = 1e3 * ms / reps;
    if (stamped) {  // diagnostic: one stamped launch, stamps returned through mean_us[1..40]
        long long* dst;
        HIPC(f, dev_alloc(f->stream, &dst, 40));
        f->gemm_stamps = dst;
        launch_gemm_variant(f, variant, transB, M, N, Kp, -1e-3f, dA, Mp, dB, brows, 1.f, dC, Mp, dC, Mp, 0, lowerB);
        long long hst[40];
        HIPC(f, hipMemcpyAsync(hst, dst, sizeof(hst), hipMemcpyDeviceToHost, f->stream));
        HIPC(f, hipStreamSynchronize(f->stream));
        f->gemm_stamps = nullptr;
        for (int i = 0; i < 40; i++) mean_us[1 + i] = (double)hst[i];
        hipFree(dst);
    }
    hipFree(dA);
    hipFree(dB);
    hipFree(dC);
    return EKFVIO_OK;
}

// Diagnostic: cycle stamps (s_memtime) of the phases of one 64x64 diagonal-block factorisation
// on an SPD test block; stamps[0..11]: start, loaded, then after each panel factor / trailing
// update (x4), inverse done, stored; stamps[16 + 16*wave + i]: per-wavefront end of work of
// phase i (2p: factor phase of panel p, 2p+1: column-update phase, 8: final phase).
int ekfvio_test_potrf_stamps(ekfvio_filter* f, int64_t stamps[80]) {
    if (!f || !stamps) return EKFVIO_EINVAL;
    HIPC(f, hipSetDevice(f->device));
    std::vector<float> hs(64 * 64);
    for (int c = 0; c < 64; c++)
        for (int r = 0; r < 64; r++) hs[c * 64 + r] = (r == c ? 2.0f : 0.f) + 0.3f / (1.0f + (r > c ? r - c : c - r));
    float *dS, *dL, *dLi;
    long long* dst;
    HIPC(f, dev_alloc(f->stream, &dS, hs.size()));
    HIPC(f, dev_alloc(f->stream, &dL, hs.size()));
    HIPC(f, dev_alloc(f->stream, &dLi, 4096));
    HIPC(f, dev_alloc(f->stream, &dst, 80));
    HIPC(f, hipMemsetAsync(dst, 0, sizeof(long long) * 80, f->stream));
    HIPC(f, hipMemcpyAsync(dS, hs.data(), sizeof(float) * hs.size(), hipMemcpyHostToDevice, f->stream));
    for (int rep = 0; rep < 3; rep++) launch_potrf_stamps(f, dS, 64, dL, dLi, dst);
    HIPC(f, hipMemcpyAsync(stamps, dst, sizeof(long long) * 80, hipMemcpyDeviceToHost, f->stream));
    std::vector<uint32_t> hl(64 * 64), hi(4096);
    HIPC(f, hipMemcpyAsync(hl.data(), dL, sizeof(float) * hl.size(), hipMemcpyDeviceToHost, f->stream));
    HIPC(f, hipMemcpyAsync(hi.data(), dLi, sizeof(float) * hi.size(), hipMemcpyDeviceToHost, f->stream));
    HIPC(f, hipStreamSynchronize(f->stream));
    // stamps[12], [13]: FNV-1a hashes of the bits of L and of the 16x16 inverses (variants must agree bit for bit)
    uint64_t h1 = 1469598103934665603ull, h2 = h1;
    for (uint32_t v : hl) h1 = (h1 ^ v) * 1099511628211ull;
    for (uint32_t v : hi) h2 = (h2 ^ v) * 1099511628211ull;
    stamps[12] = (int64_t)h1;
    stamps[13] = (int64_t)h2;
    hipFree(dS); hipFree(dL); hipFree(dLi); hipFree(dst);
    return EKFVIO_OK;
}

// Diagnostic: stamps[512] of the persistent sweep's last run (chain: [0],[1], then 8 per block step
// from [8]; first helper: 8 per block step from [256]).  enable != 0 allocates and arms the buffer.
int ekfvio_test_sweep_stamps(ekfvio_filter* f, int enable, int64_t* stamps /* [1024] */) {
    if (!f) return EKFVIO_EINVAL;
    HIPC(f, hipSetDevice(f->device));
    if (enable && !f->sweep_dbg) {
        HIPC(f, dev_alloc(f->stream, &f->sweep_dbg, 1024));
        HIPC(f, hipMemsetAsync(f->sweep_dbg, 0, 1024 * sizeof(long long), f->stream));
    }
    if (stamps && f->sweep_dbg) {
        HIPC(f, hipMemcpyAsync(stamps, f->sweep_dbg, 1024 * sizeof(long long), hipMemcpyDeviceToHost, f->stream));
        HIPC(f, hipStreamSynchronize(f->stream));
    }
    return EKFVIO_OK;
}

int ekfvio_test_sweep_fault(ekfvio_filter* f, int32_t spin_limit, int32_t stall_workgroup) {
    if (!f || spin_limit < 0) return EKFVIO_EINVAL;
    f->sweep_spin_limit = spin_limit;
    f->sweep_stall_wg = stall_workgroup;
    drop_graph(f);  // captured launches carry the old arguments
    return EKFVIO_OK;
}

int ekfvio_test_cholesky_solve(ekfvio_filter* f, int32_t m, int32_t nrhs, const float* S, const float* Crhs,
                               float* L_out, float* X_out, int32_t* info) {
    // S: m x m (ld m) SPD; Crhs: nrhs x m (ld nrhs); L_out m x m; X_out = Crhs * S^-1 (nrhs x m).
    // Runs the production path: augmented sweep + gain GEMMs.
    if (!f || m <= 0 || nrhs <= 0 || !S || !Crhs) return EKFVIO_EINVAL;
    HIPC(f, hipSetDevice(f->device));
    const int mp = round_up(m, 64), rp = round_up(nrhs, 64);
    const int ld = mp + rp + mp;
    std::vector<float> hs((size_t)ld * mp, 0.f);
    for (int c = 0; c < mp; c++) {
        for (int r = 0; r < mp; r++) hs[(size_t)c * ld + r] = (r < m && c < m) ? S[(size_t)c * m + r] : (r == c ? 1.f : 0.f);
        if (c < m)
            for (int r = 0; r < nrhs; r++) hs[(size_t)c * ld + mp + r] = Crhs[(size_t)c * nrhs + r];
        hs[(size_t)c * ld + mp + rp + c] = 1.f;
    }
    float *dS, *dL, *dLi, *dK, *dW;
    HIPC(f, dev_alloc(f->stream, &dS, hs.size()));
    HIPC(f, dev_alloc(f->stream, &dL, hs.size()));
    HIPC(f, dev_alloc(f->stream, &dLi, (size_t)64 * mp));
    HIPC(f, dev_alloc(f->stream, &dK, (size_t)rp * mp));
    HIPC(f, dev_alloc(f->stream, &dW, (size_t)rp * mp));
    HIPC(f, hipMemcpyAsync(dS, hs.data(), sizeof(float) * hs.size(), hipMemcpyHostToDevice, f->stream));
    launch_chol_sweep(f, dS, dL, dLi, mp, rp, ld);
    // one step of residual refinement against L, unless a block went through the U S U^T path (K L = Y S then, not Y)
    std::vector<unsigned long long> hsign(mp / 64);
    HIPC(f, hipMemcpyAsync(hsign.data(), f->Lsign, sizeof(unsigned long long) * hsign.size(), hipMemcpyDeviceToHost, f->stream));
    HIPC(f, hipStreamSynchronize(f->stream));
    int refine = 1;
    for (unsigned long long v : hsign) refine &= (v == 0ull);
    launch_gain_from_sweep(f, dL, mp, rp, ld, nrhs, dK, dW, rp, refine);
    std::vector<float> hl(hs.size()), hk((size_t)rp * mp);
    HIPC(f, hipMemcpyAsync(hl.data(), dL, sizeof(float) * hl.size(), hipMemcpyDeviceToHost, f->stream));
    HIPC(f, hipMemcpyAsync(hk.data(), dK, sizeof(float) * hk.size(), hipMemcpyDeviceToHost, f->stream));
    if (info) {
        HIPC(f, hipMemcpyAsync(info, f->info, sizeof(int), hipMemcpyDeviceToHost, f->stream));
        HIPC(f, hipMemsetAsync(f->info, 0, sizeof(int), f->stream));
    }
    HIPC(f, hipStreamSynchronize(f->stream));
    if (const char* dp = getenv("EKFVIO_DEBUG_DUMP_LAUG")) {  // diagnostic: the whole swept matrix [L; Y; L^-T]
        if (FILE* fp = fopen(dp, "wb")) {
            const int hdr[3] = {ld, mp, rp};
            fwrite(hdr, sizeof(int), 3, fp);
            fwrite(hl.data(), sizeof(float), hl.size(), fp);
            fclose(fp);
        }
    }
    if (L_out)
        for (int c = 0; c < m; c++)
            for (int r = 0; r < m; r++) L_out[(size_t)c * m + r] = (r >= c) ? hl[(size_t)c * ld + r] : 0.f;
    if (X_out)
        for (int c = 0; c < m; c++)
            for (int r = 0; r < nrhs; r++) X_out[(size_t)c * nrhs + r] = hk[(size_t)c * rp + r];
    hipFree(dS);
    hipFree(dL);
    hipFree(dLi);
    hipFree(dK);
    hipFree(dW);
    return EKFVIO_OK;
}
#endif  // EKFVIO_TEST_HOOKS

}  // extern "C"
