// ekf_vio_amd/csrc/common.h — internal declarations shared by the HIP translation units.
// gfx950 (MI355X / CDNA4) only: 64-wide wavefronts, fp32 MFMA, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <chrono>
#include <string>
#include <vector>

#include "../../include/ekfvio.h"
#ifdef EKFVIO_TEST_HOOKS
#include "../../include/ekfvio_test_hooks.h"
#endif

#define EKF_BASE 22
#define EKF_TILE 64  // block size of every blocked algorithm (GEMM tile edge, Cholesky nb)
// prune(SPARSE_THRESH, SPARSE_EPS) keeps |x| > 1e-8f*1e-5f (TightlyCoupledEKF.h:13-14, .cpp:117,580,591,625)
#ifndef EKF_POTRF_FV
#define EKF_POTRF_FV 12       // factor-phase variant of potrf64_lds (chol.hip): 12 = generated stream incl. its LDS traffic, LDS latency off the
                              // pivot chain (gen_ls3); 13 = the same with wider fill slots; 10 = round 1's stream; 8 = without the LDS traffic; 0 = plain
#endif
#ifndef EKF_SWEEP_SPLIT_MB
#define EKF_SWEEP_SPLIT_MB 16  // from this many 64-wide block steps on, the sweep solves each panel block once (chol.hip)
#endif
#define EKF_FLUSH_THRESH (1e-8f * 1e-5f)

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// Kernel classes for the built-in event profiler (ekfvio_profile_*).
enum ProfClass {
    PC_LINEARIZE = 0,
    PC_PREDICT,
    PC_GEMM_PREDICT,
    PC_GATHER,
    PC_CHOL,
    PC_SOLVE,
    PC_GEMM_UPDATE,
    PC_UPDATE_MISC,
    PC_KLT_PYRAMID,
    PC_KLT_TRACK,
    PC_COUNT
};

struct ProfSlot {
    double ms = 0;
    int64_t launches = 0;
    double flops = 0;
};

struct KltFrame {
    // Pyramid level l: 8-bit image with a border of `border` pixels on every side
    // (reflect-101), and interleaved int16 (dx,dy) Scharr derivatives with a zero border.
    uint8_t* img[8] = {nullptr};
    short* deriv[8] = {nullptr};
    int w[8] = {0}, h[8] = {0};
    int levels = 0;  // number of valid levels (maxLevel+1)
    float K[9] = {0};
    bool valid = false;
    int cap_w = 0, cap_h = 0;  // level-0 size the planes were allocated for (they grow with the first larger frame)
};

struct ekfvio_filter {
    ekfvio_config cfg;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string last_error;

    int N = 0;      // landmarks
    int n = EKF_BASE;
    int n_cap = 0;  // 22 + 3*max_features
    int ldp = 0;    // leading dimension of every n-row matrix (multiple of 64)
    int m_cap = 0;  // 2*max_features rounded up to 64; leading dimension of S/L

    // --- state (device) ---
    float* mu = nullptr;       // [ldp]  base (22) then [u,v,1/d] per landmark
    float* mu_next = nullptr;  // [ldp]  scratch for the propagated mean
    float* last_klt = nullptr; // [2*max_features]
    uint8_t* del_flag = nullptr;  // [max_features]
    float* P = nullptr;        // [ldp*ldp] dense covariance, column-major
    float* P2 = nullptr;       // [ldp*ldp] ping-pong / scratch (X = F P, dense F)
    // --- process Jacobian blocks ---
    float* FA = nullptr;       // [22*22] column-major
    float* FB = nullptr;       // [max_features*27]  per landmark [9 cols (state 7..15)][3 rows]
    float* FD = nullptr;       // [max_features*9]   per landmark [3 cols][3 rows]
    float* Fdense = nullptr;   // [ldp*ldp] only in dense predict mode / ekfvio_linearize
    // --- update work ---
    int* idx = nullptr;        // [m_cap] state index of measurement row r
    int* inv_idx = nullptr;    // [ldp]   measurement row of state index j, or -1
    float* zmeas = nullptr;    // [2*max_features] device copy of z
    float* Rmeas = nullptr;    // [4*max_features]
    uint8_t* pass = nullptr;   // [max_features]
    float* yres = nullptr;     // [m_cap] measured coordinate per measurement row (the residual is formed in gather_kernel)
    float* Rm = nullptr;       // [m_cap*2] per measurement row r: R(r,r) and the off-diagonal partner
    // Augmented sweep matrices, ld_aug x m_cap, column-major.  Row blocks of Saug:
    //   [0, m_pad)                 A = (H Sigma H^T + R)^T           -> Laug: L (Cholesky factor)
    //   [m_pad, m_pad+n_pad)       Sigma H^T                         -> Laug: Y = Sigma H^T L^-T
    //   [m_pad+n_pad, +m_pad)      identity                          -> Laug: L^-T
    float* Saug = nullptr;
    float* Laug = nullptr;
    int ld_aug = 0;            // m_cap + ldp + m_cap
    float* Linv = nullptr;     // [64*m_cap] inverses of the 16x16 diagonal blocks of L
    unsigned long long* Lsign = nullptr;  // [>= m_cap/64] per block column: mask of negative pivots (0 = positive definite block)
    int* sweep_sync = nullptr; // flags of the persistent sweep: ready[mb], fin[row blocks x mb], abort word (sweep_sync_words ints)
    int sweep_mode = 2;        // 2: ONE persistent launch with per-tile hand-offs behind the first diagonal tile (chol_persist.inc), where it pays and
                               // applies (3 .. 15 block columns, grid co-resident; two block columns: 43.5 against 43.0 us per step); 0 (EKFVIO_SWEEP=0): one launch per block step
    size_t sweep_sync_words = 0;
    float lin_next_dt = -1.f;     // >= 0 (set by capture_steps around launch_update): the next process(dt)'s dt -- the update's last GEMM may linearise for it
    bool prelinearized = false;   // ... and did: FA / FB / FD / mu_next hold the next step's Jacobian blocks and propagated mean (launch_predict then skips its own)
    int lin_overlap = 1;          // EKFVIO_LIN_OVERLAP=0 turns that off (A/B)
    int sym_joseph = 1;           // EKFVIO_SYM_JOSEPH=0: the second Joseph GEMM of the throughput regime forms both triangles (rounds 1-5)
    int gemm_order2d = 1;         // EKFVIO_GEMM_ORDER2D=0: the 64 x 64 GEMM's tiles in block-index order (rounds 1-4)
    long long persistent_sweeps = 0;  // sweeps enqueued (or captured) as chol_persist_kernel: ekfvio_test_persistent_sweeps
    long long schur_sweeps = 0;       // sweeps enqueued with Sigma and the gain as Schur tiles (EKFVIO_SCHUR=1): ekfvio_test_sweep_counts
    long long sweep_recoveries = 0;   // updates run again with the per-step sweep behind an aborted persistent launch
    int early_outputs = 1;            // EKFVIO_EARLY_OUTPUTS: a frame's outputs and status go out between the update's two Joseph GEMMs (klt.hip)
    void (*between_joseph)(ekfvio_filter*) = nullptr;  // ekfvio_step_image: called by launch_update between the update's two Joseph GEMMs (the frame's outputs)
    int hook_kyp_blocks = 0;          // > 0 while between_joseph is called from the T2 flow: K y is Wt's first rows (one per block column), not column n of P
    int between_joseph_seq = 0;       // ... and the status sequence number its launch publishes (0: not called)
    long long early_output_frames = 0;  // frames whose outputs went out that way (test hook)
    int publish_after_sweep_seq = 0;  // ekfvio_update: launch_update publishes the status word with this sequence number right behind the sweep (0: not asked)
    int sweep_spin_limit = 0;         // > 0: looks per wait of the persistent sweep (test hook ekfvio_test_sweep_fault); 0: SWEEP_SPIN_LIMIT
    int sweep_stall_wg = -1;          // fault injection: this workgroup of the persistent launch never raises its flag
    const int* sweep_abort_word = nullptr;  // abort word of the persistent sweep enqueued last by launch_chol_sweep (null: another sweep)
    int early_status = 1;              // EKFVIO_EARLY_STATUS: ekfvio_update returns behind the sweep, the Joseph GEMMs still running (api.hip)
    int upload_kernel = 1;             // EKFVIO_UPLOAD_KERNEL: the frame's trip to device memory is a kernel reading mapped host memory (klt.hip)
    bool sweep_retry_armed = false;    // an aborted persistent sweep latched sweep_mode to 0: tried again at sweep_retry_at (api.hip, sweep_maybe_retry)
    double sweep_retry_pause_s = 0.0, sweep_retry_first_s = 2.0;
    int sweep_probation = 0;           // > 0: clean persistent sweeps still to come behind a retry before sweep_retry_pause_s starts over (api.hip)
    int sweep_wait_ticks = 300000;     // the persistent sweep's patience per wait in 100 MHz ticks (3 ms; EKFVIO_SWEEP_WAIT_MS) where an aborted update
                                       // is run again at once (ekfvio_update, ekfvio_step_image); a device-resident run (ekfvio_run_uploaded), which
                                       // cannot, waits SWEEP_WAIT_TICKS_UNRECOVERABLE (100 ms): the counter also runs while a healthy sweep's
                                       // wavefronts are descheduled (time slicing, a profiler's serialisation), ADVICE r05
    bool sweep_unrecoverable = false;  // set by ekfvio_run_uploaded around its launches / captures
    std::chrono::steady_clock::time_point sweep_retry_at;
    bool graph_leaves_flags_clean = false;  // sweep_flags_clean as a replay of the captured step graphs leaves it (api.hip, capture_steps)
    bool sweep_flags_clean = false;   // the persistent sweep's flags are zero for the launch enqueued next (zeroed by the last GEMM of the
                                      // previous update, or by gather_potrf_kernel in front); otherwise the launcher enqueues a memset
    int fuse_sweep = 1;               // 1: gather + first diagonal tile + sweep in ONE launch where the persistent sweep applies (EKFVIO_FUSE_SWEEP)
    bool persist_attr_set = false;
    bool gain2_attr_set = false;
    int persist_oversub = 0;          // > 0 (EKFVIO_PERSIST_OVERSUB): owner workgroups allowed per compute unit's worth of the persistent launch (chol.hip, persist_shape)
    int persist_gain = 1;             // 1 (EKFVIO_PERSIST_GAIN=0 turns it off): the gain's tiles are formed inside the fused persistent launch (chol_persist.inc)
    bool gain_in_sweep = false;       // the launch enqueued last did (launch_update then skips the gain kernel)
    int persist_early = 1;            // 1 (EKFVIO_PERSIST_EARLY=0 turns it off): owners fetch their panel sources in front of the wait for ready[k] (chol_persist.inc)
    int fuse_gather = 1;       // 1: the gather and the first diagonal tile's factorisation share a launch (EKFVIO_FUSE_GATHER)
    bool gather_attr_set = false;
    int t2_flow = 1;           // 1 (EKFVIO_T2=0 turns it off): where the fused persistent launch forms the gain, freed owners also form T2 = Sigma (I - K H)^T
                               // (chol_persist.inc, t2_tile): Sigma' = T2 + K G'^T is the ONE P-update GEMM behind the launch (round 6)
    long long t2_updates = 0;  // updates enqueued (or captured) with the T2 flow: ONE P-update GEMM behind the sweep (ekfvio_get_counters [5])
    bool t2_in_sweep = false;  // the sweep enqueued last formed T2 itself (launch_update then skips gain2_t2_tiles_kernel)
    int schur = 0;             // 1 (EKFVIO_SCHUR=1): T2 and K as Schur tiles of the sweep; 0: gain GEMM + first Joseph GEMM behind it.
                               // Measured equal in step time at N = 256, so the simpler flow is the default.
    int fuse_linearize = 1;    // 1: structured process(dt) is one launch, the Jacobian blocks are formed inside it (EKFVIO_FUSE_LINEARIZE)
    int num_cus = 0;
    int last_m = 0;            // measurement rows of the most recent update (shape of its GEMMs)
    long long* sweep_dbg = nullptr;  // [512] s_memtime stamps of the persistent sweep (diagnostic; null = off)
    long long* gemm_stamps = nullptr;  // diagnostic stamp buffer handed to the next GEMM launches (null = off)
    float* Km = nullptr;       // [ldp*m_cap]  Sigma H^T, solved in place into the Kalman gain
    float* Wt = nullptr;       // [ldp*m_cap]  (H Sigma)^T
    float* Gm = nullptr;       // [ldp*m_cap]  K R - T[:,idx]
    int* info = nullptr;       // [4] device words: [0] non-positive pivot seen, [1] frame counter of uploaded sequences, [2] device-side m
    int* h_info = nullptr;     // pinned, device-mapped: [0] status word, [1] sequence number (publish_status_kernel)
    int* d_hinfo = nullptr;    // the device's address of h_info
    int status_seq = 0;
    // small per-frame outputs (odometry, point cloud): kernels write them straight into pinned host memory and the host
    // waits with wait_status: no device-to-host copy into pageable memory, no interrupt-driven synchronise
    float* h_out = nullptr;    // pinned, device-mapped: EKF_BASE + 4 * max_features floats
    float* d_out = nullptr;    // the device's address of h_out
    int frame_outputs = 1;     // 1: ekfvio_step_image's last kernel also writes the node's outputs (EKFVIO_FRAME_OUTPUTS=0: status word only)
    bool out_fresh = false;    // h_out holds base_mu and the point cloud of the CURRENT state and frame: ekfvio_step_image's last
                               // kernel writes them with the status word, and every call that changes the state or the frame
                               // clears the flag; while it is set the node's getters cost a memcpy
    unsigned char* h_meas = nullptr;  // pinned staging for one frame's (z, R, pass): one H2D copy per ekfvio_update
    unsigned char* d_meas = nullptr;  // its device image: z at 0, R at 8N_cap, pass at 24N_cap bytes
    // uploaded measurement sequences
    float* seq_z = nullptr;
    float* seq_R = nullptr;
    uint8_t* seq_pass = nullptr;
    int seq_frames = 0;
    int seq_N = 0;
    std::vector<int> seq_m;                 // measurement rows per uploaded frame
    std::vector<std::vector<uint8_t>> seq_pass_host;

    // --- KLT ---
    KltFrame frames[2];
    int cur = 0;          // index of the current frame in frames[]
    int klt_border = 0;
    float* klt_prev_px = nullptr;  // [2*max_features]
    float* klt_next_px = nullptr;  // [2*max_features]
    uint8_t* klt_status = nullptr; // [max_features]
    float* klt_cov_px = nullptr;   // [4*max_features] sample-based pixel covariances (cfg.sample_based_uncertainty)
    uint8_t* staging = nullptr;    // device staging for the uploaded image
    uint8_t* h_image = nullptr;    // pinned host staging: the caller's frame is copied here, so its buffer is free on return without a stream sync
    size_t src_cap = 0;            // bytes of staging / h_image (the uploaded, un-resized frame)
    // --- frame ingest + replenishment (fast.hip) ---
    uint8_t* blurred = nullptr;    // replenishFeatures' cv::GaussianBlur output (only with cfg.fast_blur_sigma != 0)
    unsigned* fast_row_kp = nullptr;  // [w*h] per image row its keypoints in x order, (score << 16) | x
    int* fast_kp_xy = nullptr;     // keypoints in raster order
    short* fast_kp_score = nullptr;
    int fast_kp_cap = 0;
    uint8_t* occ_mask = nullptr;   // replenishFeatures' checkImg as a bit mask (global fallback for large frames)
    int* fast_row_cnt = nullptr;   // [max_image_height] keypoints per row
    int* new_xy = nullptr;         // pixels of the landmarks added by the last replenishment
    int* fast_counts = nullptr;    // [0] keypoints found, [1] landmarks added
    int fast_cap_w = 0, fast_cap_h = 0;  // level-0 size the detector's buffers were allocated for
    double t_stamp = 0;
    bool have_stamp = false;

    // --- hipGraph replay of device-resident sequences (an even number of steps per graph: the mean
    //     ping-pong mu <-> mu_next is back in its starting orientation after an even count) ---
    hipGraphExec_t step_graph = nullptr;      // EKF_GRAPH_STEPS filter steps
    hipGraphExec_t step_graph_big = nullptr;  // EKF_GRAPH_STEPS_BIG filter steps (long runs: fewer graph launches)
    hipGraphExec_t step_graph_pair = nullptr; // 2 filter steps (tails and short runs)
    int graph_N = -1, graph_m = -1, graph_frames = -1;
    bool graph_sole = true;   // the graphs were captured while this was the device's only handle (persistent sweep inside)
    float graph_dt = -1.f;
    float* graph_mu = nullptr;      // orientation of the mean / covariance ping-pong at capture time
    float* graph_P = nullptr;
    const void* graph_seq = nullptr;
    int use_graph = 1;

    // --- profiler ---
    bool prof_on = false;
    float prof_overhead_ms = 0.f;  // elapsed time of an empty event pair, subtracted from every scope
    ProfSlot prof[PC_COUNT];
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

// ---- launchers implemented across the translation units -------------------------------
// C[MxN] = beta*Cin + alpha * A[MxK] * op(B); all column-major.  transB: B is [NxK]
// (op = transpose) else [KxN].  K must be a multiple of 32 and the K-padding of both
// operands finite*0-safe (zero).  flush != 0 applies the reference's prune (|x|<=1e-13 -> 0).
// Filter-specific epilogues of the two Joseph GEMMs (mode 0 = none):
//  mode 1  T = Sigma - K (H Sigma): besides T, writes G = K R - T[:, idx] for the measured
//          columns (inv_idx: state index -> measurement row or -1) so that no separate pass
//          re-reads T; the residual rides as an extra row of (H Sigma)^T, so output column n
//          receives K*y.
//  mode 2  Sigma' = T + G K^T: workgroup (0,0) also finishes the mean: mu += column n,
//          quaternion renormalised (:600-609), column n zeroed again, frame counter advanced.
//  mode 3  the same, K y taken from per-column-block partial sums (Schur flow).
//  (Round 4's symmetric Joseph flow -- `sym`, and mode 4, the gain GEMM that left K y as partial sums for it -- was measured, rejected on parity
//  and left the product in round 5: scripts/experiments/joseph_sym.txt, profiles/r04_symmetric_joseph_experiment.txt.)
struct GemmEpi {
    int mode = 0;
    const int* inv_idx = nullptr;
    const float* Rm = nullptr;
    float* G = nullptr;
    int ldg = 0;
    float* mu = nullptr;
    float* Pcol = nullptr;  // column n of P
    const float* Kyp = nullptr;  // mode 3: K y as `kyp_blocks` partial sums (rows of ld kyp_ld), added in order, instead of Pcol
    int kyp_blocks = 0, kyp_ld = 0;
    int n = 0;
    int* frame_counter = nullptr;
    int frames = 0;
    long long* stamps = nullptr;  // diagnostic s_memtime stamps (library built with -DEKF_GEMM_STAMPS), per handle
    int* zero_words = nullptr;    // modes 2-3: the persistent sweep's flags, zeroed by workgroup (0,0) for the NEXT update's sweep
    int n_zero = 0;               // (everything but the abort word, which only ever goes up and retires the persistent path)
    int order2d = 0;              // gemm_f32_mfma_kernel: each XCD's run of tiles is a compact 2-D patch (launch_gemm_cfg decides)
    // round 6 (mode 3, gemm16_kernel): the linearisation of the NEXT process(dt) in `lin_blocks` workgroups behind the tiles' and the mean's -- a device-resident run
    // knows the next dt, and K y is final before the launch (Kyp), so numericallyLinearizeProcess (:176-325) and the mean propagation at mu + K y run while
    // this launch's tiles do, and the covariance propagation behind it only has the strips left (motion_model.inc, launch_update)
    int lin_blocks = 0;
    int lin_N = 0;
    float lin_dt = 0.f;
    float* lin_FA = nullptr;
    float* lin_FB = nullptr;
    float* lin_FD = nullptr;
    float* lin_mu_next = nullptr;
    int mean_keep = 0;            // (set by launch_gemm_cfg with lin_blocks) gemm16_finish_mean leaves mu alone
    int sym_w = 1;                // ... in strips of sym_w tile columns, each walked row by row
    int sym = 0;                  // mode 2, gemm_f32_mfma_kernel: only the lower triangle's tiles are formed, each also writes its transpose
    const int* abort = nullptr;   // modes 1-3: abort word of the persistent sweep in front (non-zero: the factor is unfinished) --
                                  // the kernel then writes nothing: Sigma, mu and the frame counter stay as process(dt) left them
};
void launch_gemm(ekfvio_filter* f, int transB, int M, int N, int K, float alpha, const float* A, int lda, const float* B,
                 int ldb, float beta, const float* Cin, int ldcin, float* C, int ldc, int flush, int lowerB = 0,
                 const GemmEpi* epi = nullptr);

bool gemm_throughput_regime(const ekfvio_filter* f, int M, int N, int K);
// would launch_gemm run A * B^T of this shape as ONE wave of gemm16_kernel workgroups with `extra` more workgroups still inside that wave?
bool gemm_single_round_with(const ekfvio_filter* f, int M, int N, int K, int extra);

// same, selecting a tile configuration (0 = production default chosen by shape)
void launch_gemm_variant(ekfvio_filter* f, int variant, int transB, int M, int N, int K, float alpha, const float* A, int lda,
                         const float* B, int ldb, float beta, const float* Cin, int ldcin, float* C, int ldc, int flush,
                         int lowerB);


// Measurement bookkeeping of one update (device pointers); see bookkeeping_body in ekf_kernels.hip
struct BookArgs {
    int enabled = 0;
    int N = 0, m_pad = 0;
    const float* z = nullptr;
    const float* R = nullptr;
    const uint8_t* pass = nullptr;
    float* last_klt = nullptr;
    uint8_t* del_flag = nullptr;
    int* idx = nullptr;
    int* inv_idx = nullptr;
    float* zrow = nullptr;  // measured coordinate per measurement row (f->yres)
    float* Rm = nullptr;
    const int* frame_counter = nullptr;
    int* m_out = nullptr;   // receives the number of measurement rows 2 * (#passed) (device-side m, ekfvio_step_image)
};
BookArgs make_book_args(ekfvio_filter* f, int m, const float* d_z, const float* d_R, const uint8_t* d_pass, const int* d_frame_counter);
void launch_linearize(ekfvio_filter* f, float dt, const BookArgs* book = nullptr);
void launch_build_dense_F(ekfvio_filter* f, float* Fdense);
void launch_predict(ekfvio_filter* f, float dt, const BookArgs* book = nullptr);
// m_on_device: the host does not know how many landmarks passed (no D2H of the flags): launches are sized for m = 2N,
// the kernels read the true row count from f->info[2] (written by the bookkeeping) and treat the rest as padding
void launch_update(ekfvio_filter* f, int m, const float* d_z, const float* d_R, const uint8_t* d_pass,
                   int* d_frame_counter = nullptr, int frames = 0, bool bookkeeping_done = false, bool m_on_device = false);
void launch_check_sigma(ekfvio_filter* f, float* d_out);
// klt.hip helpers shared with fast.hip
int klt_level_pitch(int w);
int klt_border();
void klt_intrinsics(const ekfvio_filter* f, const float* K, float* fx, float* fy, float* cx, float* cy);
// imu.hip
void launch_imu_update(ekfvio_filter* f, const float gyro[3], const float accel[3]);
// fast.hip
int fast_alloc(ekfvio_filter* f);
void fast_free(ekfvio_filter* f);
int fast_ensure(ekfvio_filter* f, int w, int h);  // grows the detector's per-pixel buffers to a w x h level 0 (synchronises if it must)
// Waits for everything on the handle's stream and returns the factorisation's status word through *status (api.hip).
// extra_dev (may be null): one more device word delivered with it through *extra_out.
void launch_publish_status(ekfvio_filter* f, int seq);  // the status word and `seq` to pinned host memory, stream-ordered (api.hip)
int wait_status(ekfvio_filter* f, int* status, const int* extra_dev = nullptr, int* extra_out = nullptr);
// the same in two halves, for a caller whose own (single-workgroup) kernel publishes the words itself
int next_status_seq(ekfvio_filter* f);
// api.hip: what a host does when the status word says the persistent sweep gave up (bit 1): the handle goes to the per-step
// sweep for good and its captured graphs are dropped
void sweep_abort_latch(ekfvio_filter* f);
void sweep_clean_update(ekfvio_filter* f);  // an update whose persistent sweep came through (api.hip)
void sweep_maybe_retry(ekfvio_filter* f);  // at the entry points that enqueue updates: the persistent sweep again, some time after an abort
int poll_status(ekfvio_filter* f, int seq, int* status, int* extra_out);
// api.hip: addNewFeatures with the k new (u,v) already in f->zmeas on the device
int add_features_device(ekfvio_filter* f, int k);
void add_features_enqueue_device_count(ekfvio_filter* f, const int* count_dev);  // enqueue only; the caller updates N / n
int replenish_enqueue(ekfvio_filter* f, int* enqueued);  // fast.hip: FAST + first-fit selection, count left in f->fast_counts[1]
// The two P-update GEMM launches of an update with m measurement rows, `reps` times, into scratch (P2, Gm): the
// filter state is not touched.  For timing the kernel under its production shape (ekfvio_profile_update_gemms).
int launch_update_gemms_scratch(ekfvio_filter* f, int m, int reps);  // returns the GEMM launches per repetition (1 with the Schur sweep, else 2)
// Augmented blocked Cholesky sweep (chol.hip): Saug = [A; X; I] (row blocks of 64; A is
// m_pad x m_pad, X has n_pad rows) -> Laug = [L; X L^-T; L^-T], both ld x m_pad column-major.
// schur: T = Sigma - X A^-1 X^T (in place in f->P) and K = X A^-1 (f->Km) come out of the sweep itself as Schur tiles
// (chol.hip); not in the split sweep (m_pad >= 64 * EKF_SWEEP_SPLIT_MB) nor the persistent one
void launch_chol_sweep(ekfvio_filter* f, float* Saug, float* Laug, float* Linv, int m_pad, int n_pad, int ld,
                       bool first_tile_done = false, bool schur = false);
bool sweep_supports_schur(const ekfvio_filter* f, int m_pad);
bool sweep_is_persistent(const ekfvio_filter* f, int m_pad, int n_pad);
int persist_zero_words(int m_pad, int n_pad);
void launch_persist_fused(ekfvio_filter* f, int m, int m_pad, int n_pad, bool m_on_device);
int live_handles_on(int device);  // api.hip: handles alive on that device in this process
// K pruned, G = K R - T[:, idx], K y partial sums (one row of f->Wt per 64 measurement columns)
void launch_joseph_g(ekfvio_filter* f, int m, int m_pad, int n_pad, bool m_on_device, const float* T = nullptr);  // T: the covariance G' is taken from (null: f->P)
bool t2_flow_shape(const ekfvio_filter* f, int m_pad, int n_pad);  // chol.hip: T2 = Sigma (I - K H)^T comes out of the sweep (or gain2_t2_tiles_kernel), ONE Joseph GEMM behind it
// Where T2 lives between the sweep and the one GEMM behind it: the dense-F buffer, dead during an update in either predict mode (the dense
// mode rebuilds F at the next process(dt)).  Not f->P2: that is process(dt)'s next output, and with T2 written there by other XCDs moments
// earlier process(dt) measured 0.6 us longer (same-box rocprofv3: 9.51 against 8.90 us).
inline float* t2_buffer(ekfvio_filter* f) { return f->Fdense; }
void launch_gain2_tiles(ekfvio_filter* f, int m, int m_pad, int n_pad, bool m_on_device);
void launch_gather_potrf(ekfvio_filter* f, int m, int m_pad, int n_pad, bool m_on_device = false, bool with_wt = true);
void launch_potrf_stamps(ekfvio_filter* f, const float* S, int ld, float* L, float* Linv, long long* d_stamps);
// K = X A^-1 (n rows, ldk) from the sweep output: K = Y L^-1 (+ optional residual refinement).
void launch_gain_from_sweep(ekfvio_filter* f, const float* Laug, int m_pad, int n_pad, int ld, int n, float* K,
                            float* scratch, int ldk, int refine, const GemmEpi* epi = nullptr);

int klt_alloc(ekfvio_filter* f);  // klt.hip
void klt_free(ekfvio_filter* f);

struct ProfScope {
    ekfvio_filter* f;
    int cls;
    int launches;
    ProfScope(ekfvio_filter* f_, int cls_, double flops = 0, int launches_ = 1) : f(f_), cls(cls_), launches(launches_) {
        if (f->prof_on) {
            (void)hipEventRecord(f->ev0, f->stream);
            f->prof[cls].flops += flops;
        }
    }
    ~ProfScope() {
        if (f->prof_on) {
            (void)hipEventRecord(f->ev1, f->stream);
            (void)hipEventSynchronize(f->ev1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, f->ev0, f->ev1);
            f->prof[cls].ms += ms;  // raw: includes the event-pair overhead (f->prof_overhead_ms, reported separately)
            f->prof[cls].launches += launches;
        }
    }
};
