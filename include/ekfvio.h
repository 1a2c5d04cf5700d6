/* include/ekfvio.h — C-ABI of libekfvio_hip.so, the MI355X (gfx950) backend for the
 * per-frame hot path of k-sheridan/ekf_vio (predict -> KLT measure -> EKF update).
 *
 * The reference has no plugin/FFI layer: the seam is the public C++ surface of
 * `class TightlyCoupledEKF` (include/ekf_vio/TightlyCoupledEKF.h:25-68) and
 * `KLTTracker::findNewFeaturePositions` (include/ekf_vio/KLTTracker.h:88-90), both called
 * only from `EKFVIO::addFrame` (include/ekf_vio/EKFVIO.cpp:139-219).  Every entry point
 * below names the reference member it replaces.  Plain pointers and sizes only; the
 * caller owns every host buffer; all device state (mean, dense covariance, image
 * pyramids, work buffers) is owned by the opaque handle.  One handle = one HIP stream on
 * one device; calls on one handle must be serialised by the caller; different handles are
 * independent: no handle reads another's state.  The library's only process-wide state is a per-device count of live
 * handles (a device's SOLE handle may run the Cholesky sweep as one persistent launch, which needs every compute unit it
 * asks for; with several handles on a device each takes one launch per block step: same results) and the values of
 * diagnostic environment switches (EKFVIO_*), read once.
 *
 * Layouts: vectors of 2-D points are x,y interleaved f32 (std::vector<Eigen::Vector2f>);
 * 2x2 covariances are column-major f32 (std::vector<Eigen::Matrix2f>); flags are one byte
 * each (std::vector<bool>); dense matrices are column-major with an explicit leading
 * dimension; images are 8-bit single channel with a row stride in bytes.
 *
 * Errors: the reference aborts through ROS_ASSERT or logs and continues.  Here every
 * function returns an int status and never aborts or throws across the boundary.
 */
#ifndef EKFVIO_H_
#define EKFVIO_H_

#include <stdint.h>

/* The library is compiled with -fvisibility=hidden: the entry points marked EKFVIO_API are its ONLY dynamic symbols
 * (tests/test_abi_cpu.py compares the whole `nm -D --defined-only` set with this header), so nothing of its internals can
 * interpose on, or be interposed by, what else a ROS node links (OpenCV, tf, ...). */
#ifndef EKFVIO_API
#define EKFVIO_API __attribute__((visibility("default")))
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define EKFVIO_BASE_STATE_SIZE 22 /* TightlyCoupledEKF.h:12 */

enum {
    EKFVIO_OK = 0,
    EKFVIO_EINVAL = 1,   /* what the reference ROS_ASSERTs (size mismatch, dt < 0, null) */
    EKFVIO_ENUMERIC = 2, /* The factorisation of S met a NON-POSITIVE pivot; the state is still updated, through a signed factor
                            S = U diag(+-1) U^T, as the reference's LDL^T carries a negative pivot along.  STRICTER than the reference's
                            log line: `ROS_ERROR_COND(solver.info() == Eigen::NumericalIssue, ...)` (TightlyCoupledEKF.cpp:579) fires only
                            for a pivot that is EXACTLY zero -- Eigen's simplicial LDL^T sets NumericalIssue for `d == 0` alone and then
                            abandons the factorisation, so what the reference computes behind it (:580) is undefined (it reads D entries
                            it never wrote); a negative pivot passes there without a word.  Here both are reported with this one code; on
                            an exactly zero pivot the state behind it is not finite (1 / 0 in the factor), as undefined as upstream's.
                            (tests/test_oracle_amd_cpu.py, tests/test_gpu_shapes.py: test_exactly_singular_s_*) */
    EKFVIO_ECAPACITY = 3,/* more landmarks than config.max_features */
    EKFVIO_EDEVICE = 4,  /* HIP runtime failure; ekfvio_last_error() has the text */
    EKFVIO_ESTATE = 5,   /* call sequence error (e.g. KLT track before two frames exist) */
    EKFVIO_EABORTED = 6  /* the persistent Cholesky sweep could not get its workgroups resident together (another process or
                            stream held compute units) and gave up.  NOT a numeric warning: the updates behind it were
                            skipped -- Sigma, mu are the propagated ones, never a half-finished factor's.  ekfvio_update and
                            ekfvio_step_image do not return it: they run the update again at once with one launch per block
                            step.  ekfvio_synchronize returns it for a device-resident run (ekfvio_run_uploaded), whose skipped
                            updates cannot be replayed.  Either way the handle uses the per-step sweep from then on. */
};

enum { EKFVIO_PREDICT_STRUCTURED = 0, /* exploits F = [[A,0],[B,D]] (nnz = 358+36N) */
       EKFVIO_PREDICT_DENSE = 1       /* dense F P F^T through the MFMA GEMM (north-star form) */ };

typedef struct ekfvio_filter ekfvio_filter; /* opaque */

/* The handful of Params.h globals the hot path reads, as a plain struct. */
typedef struct ekfvio_config {
    int32_t max_features;                  /* capacity; NUM_FEATURES (Params.h:46) */
    float default_point_depth;             /* Params.h:83, 0.5 */
    float default_point_depth_variance;    /* Params.h:84, 100 */
    float default_point_homogenous_variance; /* Params.h:86, 1e-5 */
    int32_t predict_mode;                  /* EKFVIO_PREDICT_* */
    /* KLT (KLTTracker.cpp:61-64, Params.h:33,36,103-104) */
    int32_t klt_window_size;               /* WINDOW_SIZE 21 */
    int32_t klt_max_pyramid_level;         /* MAX_PYRAMID_LEVEL 3 */
    int32_t klt_max_iterations;            /* 30 */
    float klt_epsilon;                     /* 0.01 */
    float klt_min_eigen;                   /* KLT_MIN_EIGEN 1e-4 */
    int32_t kill_pad;                      /* KILL_PAD 11 */
    int32_t max_image_width;               /* device pyramid capacity */
    int32_t max_image_height;
    int32_t use_principal_point;           /* 0 reproduces Feature.h:60-66 (cx,cy ignored) */
    /* frame ingest and landmark replenishment (Frame.cpp:15-42, EKFVIO.cpp:224-311, Params.h:24-28,43) */
    int32_t inverse_image_scale;           /* INVERSE_IMAGE_SCALE: frames are resized to (w/s, h/s), K/s; reference default 4, here 1 */
    int32_t fast_threshold;                /* FAST_THRESHOLD 50 */
    int32_t min_new_feature_dist;          /* MIN_NEW_FEATURE_DIST 30 (radius of the occupancy circles) */
    float fast_blur_sigma;                 /* FAST_BLUR_SIGMA: 0 = off (reference default), > 0: cv::GaussianBlur(5x5, sigma) before FAST */
    int32_t replenish;                     /* 1: ekfvio_step_image also runs replenishFeatures */
    int32_t sample_based_uncertainty;      /* 0 (reference behaviour): R = 1e-5 I px^2 (estimateUncertainty, KLTTracker.cpp:100-106);
                                              1: R from estimateUncertaintySampleBased (:111-175, dead code there) */
    /* IMU measurement update (the reference's imu_callback is a logging stub, EKFVIO.cpp:113-115) */
    int32_t use_imu;                       /* 0 (reference behaviour): ekfvio_imu does nothing; 1: propagate to the record's stamp, then update */
    float imu_gyro_variance;               /* (rad/s)^2 per axis, default 1e-4 */
    float imu_accel_variance;              /* (m/s^2)^2 per axis, default 1e-2 */
    float gravity[3];                      /* gravity in the filter's world frame (= the first camera frame), default (0, 9.81, 0):
                                              an optical frame, y down; the accelerometer model is a + b_acc - R(q)^T gravity */
} ekfvio_config;

/* Fills `cfg` with the reference defaults (Params.h D_* values). */
EKFVIO_API int ekfvio_default_config(ekfvio_config* cfg);

/* TightlyCoupledEKF::TightlyCoupledEKF() + initializeBaseState()
 * (TightlyCoupledEKF.cpp:10-56).  `device` is a HIP device ordinal.  `stream` is an
 * existing hipStream_t to enqueue on, or NULL to let the handle create its own.
 * On failure nothing is left allocated and *out is NULL. */
EKFVIO_API int ekfvio_create(const ekfvio_config* cfg, int device, void* stream, ekfvio_filter** out);
EKFVIO_API int ekfvio_destroy(ekfvio_filter* f);
/* initializeBaseState(): back to mu = [0,0,0,1,0...], Sigma diag [0x7,30x9,0.5x6], no landmarks. */
EKFVIO_API int ekfvio_reset(ekfvio_filter* f);
EKFVIO_API const char* ekfvio_last_error(const ekfvio_filter* f);

/* addNewFeatures(std::vector<Eigen::Vector2f>) (TightlyCoupledEKF.cpp:58-94). */
EKFVIO_API int ekfvio_add_features(ekfvio_filter* f, const float* uv, int32_t count);

/* process(float dt) (TightlyCoupledEKF.cpp:96-121): FD linearisation, mean propagation,
 * Sigma <- F Sigma F^T + Q(dt), flush below 1e-13. */
EKFVIO_API int ekfvio_process(ekfvio_filter* f, float dt);

/* numericallyLinearizeProcess (TightlyCoupledEKF.cpp:176-325): writes the dense n x n
 * Jacobian (column-major, ld = n) to host memory.  Does not change the state. */
EKFVIO_API int ekfvio_linearize(ekfvio_filter* f, float dt, float* F_dense);

/* updateWithFeaturePositions(z, R, pass) (TightlyCoupledEKF.cpp:475-628).  `count` must
 * equal the number of landmarks (reference ROS_ASSERT at :478).  Entries of z/R for
 * failed landmarks are ignored.  Returns EKFVIO_OK or EKFVIO_ENUMERIC (an aborted persistent sweep is
 * recovered inside the call, see EKFVIO_EABORTED).  The call returns as soon as the update's status is known -- when the
 * Cholesky sweep has ended; the covariance update behind it may still be running: every later call on this handle is ordered
 * behind it or waits for it (ekfvio_synchronize waits explicitly). */
EKFVIO_API int ekfvio_update(ekfvio_filter* f, const float* z, const float* R, const uint8_t* pass, int32_t count);

/* formFeatureMeasurementMap (TightlyCoupledEKF.cpp:634-661): state index of the single 1.0
 * in each row of H.  Writes 2*(#measured) ints, returns that count through *rows. */
EKFVIO_API int ekfvio_measurement_map(const ekfvio_filter* f, const uint8_t* measured, int32_t count, int32_t* idx, int32_t* rows);

/* previousFeaturePositionVector() (TightlyCoupledEKF.cpp:462-470) and the landmark records
 * (Feature.h:41-46).  Any pointer may be NULL. */
EKFVIO_API int ekfvio_num_features(const ekfvio_filter* f);
EKFVIO_API int ekfvio_dim(const ekfvio_filter* f); /* 22 + 3N */
EKFVIO_API int ekfvio_get_base_mu(ekfvio_filter* f, float base_mu[EKFVIO_BASE_STATE_SIZE]);
EKFVIO_API int ekfvio_get_features(ekfvio_filter* f, float* mu3N, float* last_klt2N, uint8_t* delete_flagN);
EKFVIO_API int ekfvio_get_sigma(ekfvio_filter* f, float* sigma, int32_t ld);
/* getFeatureHomogenousCovariance / getFeatureDepthVariance (TightlyCoupledEKF.cpp:663-681). */
EKFVIO_API int ekfvio_get_feature_cov(ekfvio_filter* f, int32_t index, float cov2x2[4]);
EKFVIO_API int ekfvio_get_depth_variance(ekfvio_filter* f, int32_t index, float* var);
/* setFeatureHomogenousCovariance(index, cov) (TightlyCoupledEKF.cpp:668-676): overwrites the 2x2 (u,v) block of Sigma
 * (column-major, like Eigen::Matrix2f). */
EKFVIO_API int ekfvio_set_feature_cov(ekfvio_filter* f, int32_t index, const float cov2x2[4]);
/* getMetric2PixelMap(K) / getPixel2MetricMap(K) (TightlyCoupledEKF.cpp:683-697): J = diag(K(0,0), K(1,1)) and
 * diag(1/K(0,0), 1/K(1,1)), column-major 2x2; K row-major 3x3 as in CameraInfo.K.  Pure functions of K. */
EKFVIO_API int ekfvio_metric2pixel_map(const float K[9], float J2x2[4]);
EKFVIO_API int ekfvio_pixel2metric_map(const float K[9], float J2x2[4]);
/* What EKFVIO::publishOdometry puts into nav_msgs/Odometry (EKFVIO.cpp:444-477): position base_mu[0..2], orientation
 * (w,x,y,z) base_mu[3..6], linear twist base_mu[7..9], angular twist base_mu[10..12].  Any pointer may be NULL. */
EKFVIO_API int ekfvio_get_odometry(ekfvio_filter* f, float position[3], float orientation_wxyz[4], float linear[3], float angular[3]);
/* What EKFVIO::publishPoints puts into sensor_msgs/PointCloud (EKFVIO.cpp:479-518), formed on the device: camera-frame
 * xyz = (u/rho, v/rho, 1/rho) per landmark and the "intensity" channel = byte of the current (resized) frame at the
 * landmark's pixel (Feature::getPixel, rounded like cv::Point(cv::Point2f)); 0 for a pixel outside the image (the
 * reference reads unchecked there) or before the first frame.  Either pointer may be NULL. */
EKFVIO_API int ekfvio_get_points(ekfvio_filter* f, float* xyz3N, float* intensityN);
/* checkSigma (TightlyCoupledEKF.cpp:699-714) as numbers: min diagonal, max |S_ij - S_ji|. */
EKFVIO_API int ekfvio_check_sigma(ekfvio_filter* f, float* min_diag, float* max_asym);

/* Checkpoint / teacher-forcing hook (the reference has none; its members are public). */
EKFVIO_API int ekfvio_set_state(ekfvio_filter* f, int32_t n_features, const float* base_mu, const float* mu3N,
                     const float* last_klt2N, const uint8_t* delete_flagN, const float* sigma, int32_t ld);

/* ---- KLT (KLTTracker::findNewFeaturePositions, KLTTracker.cpp:29-95) ----------------- */
/* Uploads a frame (Frame.h:25-41: image + intrinsics K row-major 3x3 as in CameraInfo.K),
 * builds its pyramid and Scharr derivatives on the device and makes it the current frame;
 * the former current frame becomes the previous one (frame_buffer depth 2, Params.h:58). */
EKFVIO_API int ekfvio_klt_push_frame(ekfvio_filter* f, const uint8_t* image, int32_t width, int32_t height, int32_t stride,
                          const float K[9]);
/* Tracks every landmark from the previous into the current frame: reference points are the
 * landmarks' last KLT results, initial guesses the EKF-predicted positions.  Outputs metric
 * z (2N), metric R (4N, 1e-5 px^2 scaled by 1/fx^2, 1/fy^2), pass (N).  Host pointers; any may be NULL. */
EKFVIO_API int ekfvio_klt_track(ekfvio_filter* f, float* z2N, float* R4N, uint8_t* passN);
/* Pixel-space tracking of arbitrary points (calcOpticalFlowPyrLK semantics with
 * OPTFLOW_USE_INITIAL_FLOW) between the two resident frames; for tests. */
EKFVIO_API int ekfvio_klt_track_points(ekfvio_filter* f, const float* prev_px, const float* init_px, int32_t count,
                            float* out_px, uint8_t* status);

/* KLTTracker::estimateUncertaintySampleBased (KLTTracker.cpp:111-175) between the two resident frames: for every
 * point a pixel-space 2x2 covariance (row-major, px^2) from 25 samples (offsets -10..10 step 5 around cur_px) of 5x5
 * sub-pixel patches (cv::getRectSubPix semantics) weighted by exp(-0.01 * mean squared difference to the 5x5
 * reference patch at ref_px in the previous frame).  Host pointers. */
EKFVIO_API int ekfvio_klt_uncertainty_points(ekfvio_filter* f, const float* ref_px, const float* cur_px, int32_t count, float* cov4);

/* Test hook: interior of pyramid level `level` of the current frame (8-bit image, w*h, and
 * interleaved int16 Scharr dx,dy, w*h*2).  Either output may be NULL. */
EKFVIO_API int ekfvio_klt_get_level(ekfvio_filter* f, int32_t level, int32_t* w, int32_t* h, uint8_t* img, int16_t* deriv);

/* EKFVIO::addFrame + updateStateWithNewImage (EKFVIO.cpp:139-219) without the ROS
 * publishing: first frame only stores the image and stamp; later frames run
 * process(dt = stamp - t), then KLT + update if landmarks exist.  With cfg.replenish = 1 the
 * two replenishFeatures calls of addFrame (:154, :172) run on the device as well; with 0 the
 * caller adds landmarks (ekfvio_replenish, or its own detector + ekfvio_add_features).
 * The image is copied before the call returns; the call waits for the device once, at its end
 * (status word), the pass flags of the tracker never travel to the host.  That wait (here, in ekfvio_update and in
 * ekfvio_synchronize) polls a word the device writes into pinned host memory for up to 300 us of the calling
 * thread's time, then blocks in hipStreamSynchronize.
 * Returns EKFVIO_OK or EKFVIO_ENUMERIC. */
EKFVIO_API int ekfvio_step_image(ekfvio_filter* f, double stamp, const uint8_t* image, int32_t width, int32_t height,
                      int32_t stride, const float K[9]);

/* EKFVIO::replenishFeatures (EKFVIO.cpp:224-311) on the current frame: cv::FAST(threshold, nonmax) on the
 * (resized) image, occupancy circles of radius min_new_feature_dist around the landmarks' pixels, first fit
 * in detector order with the kill-box test, addNewFeatures(pixel2Metric(.)) until max_features landmarks
 * exist.  `added` (may be NULL) receives the number of new landmarks, new_px_xy (may be NULL, room for
 * 2*max_features ints) their pixels. */
EKFVIO_API int ekfvio_replenish(ekfvio_filter* f, int32_t* added, int32_t* new_px_xy);
/* Test hook: cv::FAST(level 0 of the current frame, blurred first if cfg.fast_blur_sigma != 0, threshold, nonmax),
 * TYPE_9_16, keypoints in raster order. */
EKFVIO_API int ekfvio_fast_detect(ekfvio_filter* f, int32_t threshold, int32_t nonmax, int32_t cap, int32_t* xy, int32_t* score,
                       int32_t* count);
/* EKFVIO::imu_callback (EKFVIO.cpp:113-115), a logging stub in the reference: with cfg.use_imu = 0 (default) this does
 * nothing.  With cfg.use_imu = 1 the record first propagates the filter to its stamp -- process(dt = stamp - t),
 * the reference's motion model; the first record or frame only sets t -- and then updates with z = [gyro; accel],
 * h = [omega + b_gyr; a + b_acc - R(q)^T gravity], Joseph form (specification: oracle/ekf_oracle.hpp imu_update).
 * EKFVIO_EINVAL for a stamp before the filter's time.  Asynchronous. */
EKFVIO_API int ekfvio_imu(ekfvio_filter* f, double stamp, const float gyro[3], const float accel[3]);
/* The update alone (no propagation), whatever cfg.use_imu says: for tests and callers that propagate themselves. */
EKFVIO_API int ekfvio_imu_update(ekfvio_filter* f, const float gyro[3], const float accel[3]);

/* ---- device-resident measurement sequences (benchmark / replay) ---------------------- */
/* Copies `frames` consecutive (z, R, pass) triples for the current landmark count to HBM. */
EKFVIO_API int ekfvio_upload_measurements(ekfvio_filter* f, int32_t frames, const float* z, const float* R, const uint8_t* pass);
/* Runs process(dt) + update(frame i) for i = first .. first+count-1 (indices wrap modulo the
 * uploaded frame count) with no host<->device traffic.  Asynchronous; pair with
 * ekfvio_synchronize.  count = 0 runs nothing but prepares the launch graphs the runs replay (their
 * capture costs milliseconds: a caller that times a run prepares first). */
EKFVIO_API int ekfvio_run_uploaded(ekfvio_filter* f, int32_t first, int32_t count, float dt);
/* Waits for the handle's stream.  Returns EKFVIO_ENUMERIC (once) if a run since the last status read met a
 * non-positive pivot (reference: ROS_ERROR_COND at TightlyCoupledEKF.cpp:579, continues); EKFVIO_EABORTED if a persistent
 * sweep of the run gave up (the updates behind it were skipped; see the enum); else EKFVIO_OK. */
EKFVIO_API int ekfvio_synchronize(ekfvio_filter* f);

/* ---- instrumentation ----------------------------------------------------------------- */
/* Per-kernel-class device time accumulated with HIP events on the handle's stream while
 * profiling is on.  Classes: see ekfvio_profile_name.  Times in milliseconds. */
EKFVIO_API int ekfvio_profile_enable(ekfvio_filter* f, int32_t on);
EKFVIO_API int ekfvio_profile_reset(ekfvio_filter* f);
EKFVIO_API int ekfvio_profile_count(void);
EKFVIO_API const char* ekfvio_profile_name(int32_t cls);
EKFVIO_API int ekfvio_profile_get(ekfvio_filter* f, int32_t cls, double* total_ms, int64_t* launches, double* flops);
/* Mean launch duration (us) of the P-update GEMM(s) at the shape of the most recent update -- Sigma' = T + G K^T, and,
 * where the sweep does not produce T itself (EKFVIO_SCHUR=0, m >= 1024), T = Sigma - K W as well -- `reps` repetitions
 * replayed back to back from one hipGraph between two HIP events on the handle's stream; results go to scratch, the
 * state is untouched.  flops_per_launch (may be NULL) = the flops a launch EXECUTES, averaged over the update's P-update
 * launches: 2 n n m_pad, less where the second Joseph GEMM forms the lower triangle's tiles only (N > 334, round 6). */
EKFVIO_API int ekfvio_profile_update_gemms(ekfvio_filter* f, int32_t reps, double* avg_launch_us, double* flops_per_launch);

/* Diagnostic counters of a handle (no device work): counters[0] Cholesky sweeps that went out as the single persistent launch
 * (chol_persist_kernel; the others took one launch per block step), [1] sweeps with Sigma and the gain as Schur tiles (EKFVIO_SCHUR=1),
 * [2] updates run again behind an aborted persistent sweep, [3] the handle's sweep mode now (2: persistent where it applies, 0: one launch
 * per block step), [4] frames of ekfvio_step_image whose outputs and status were published in front of the update's last GEMM,
 * [5] updates whose right Joseph factor T2 = Sigma (I - K H)^T came out of the Cholesky sweep's launch (or the tile kernel that stands in for
 * it behind a per-step sweep), leaving ONE P-update GEMM behind it (where the fused persistent launch forms the gain: about 65 .. 265
 * landmarks, all measured), [6..7] reserved (0). */
EKFVIO_API int ekfvio_get_counters(ekfvio_filter* f, int64_t counters[8]);

#ifdef __cplusplus
}
#endif
#endif /* EKFVIO_H_ */
