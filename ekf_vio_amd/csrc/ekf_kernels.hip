// ekf_vio_amd/csrc/ekf_kernels.hip — process model, finite-difference linearisation,
// structured covariance propagation and the gather/scatter kernels of the update.
//
// Reference: include/ekf_vio/TightlyCoupledEKF.cpp — process :96-121, Q :123-174, FD
// Jacobian :176-325, convolveBaseState :328-395, convolveFeature :397-460, update
// bookkeeping :486-541, residual :554-555, mean update :600-620.
//
// This translation unit is compiled with -ffp-contract=off and evaluates every expression
// in the reference's operation order (including the places where a double literal makes the
// reference compute in double), so the Jacobian blocks and propagated means agree with an
// x86-64 build of the same formulas to the last bit wherever the math library does.
#include <algorithm>
#include <utility>

#include "common.h"

namespace {

#include "motion_model.inc"

// ---------------------------------------------------------------------------------------
// numericallyLinearizeProcess + mean propagation in one launch.
// Workgroups 0 .. ceil(N/8)-1: 8 landmarks each, 32 lanes per landmark.  Every workgroup
// first evaluates the 19 base-state variants (unperturbed + 9 columns x {+,-}) the landmark
// rows need into LDS; then lane e < 25 of a landmark evaluates ONE motion (18 base
// perturbations, 6 own perturbations, the propagated mean) and the finite differences are
// formed through LDS.  The last workgroup produces the 22x22 base block A (32 evaluations of
// convolveBaseState on 32 lanes) and the propagated base mean.
// ---------------------------------------------------------------------------------------
// Body of the bookkeeping (formFeatureMeasurementMap :634-661 and the per-landmark loop of :492-530) for one
// workgroup of NT threads.  zrow receives the measured coordinate of every measurement row (the residual
// z - H mu is formed by gather_kernel, once the propagated mean exists), so this depends only on the frame's
// measurements: it can run as an extra workgroup of the linearisation launch (BookArgs) or on its own.
template <int NT>
__device__ __forceinline__ void bookkeeping_body(const BookArgs& a) {
    __shared__ int s_cnt[32];
    __shared__ int s_total;
    const int tid = threadIdx.x;
    const int N = a.N, m_pad = a.m_pad;
    const float* z = a.z;
    const float* R = a.R;
    const uint8_t* pass = a.pass;
    if (a.frame_counter) {
        const int fi = *a.frame_counter;
        z += (size_t)fi * 2 * N;
        R += (size_t)fi * 4 * N;
        pass += (size_t)fi * N;
    }
    const int per = (N + NT - 1) / NT;
    const int lo = tid * per;
    const int hi = min(N, lo + per);
    int c = 0;
    for (int i = lo; i < hi; i++) c += pass[i] ? 1 : 0;
    // exclusive scan of the NT per-thread counts: shuffle scan inside each wavefront, then
    // the wavefront totals are scanned by wavefront 0 (two barriers in all)
    const int lane = tid & 63, wv = tid >> 6;
    constexpr int NW = NT / 64;
    int incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) s_cnt[wv] = incl;
    __syncthreads();
    if (wv == 0) {
        int t = (lane < NW) ? s_cnt[lane] : 0;
        int ti = t;
#pragma unroll
        for (int off = 1; off < NW; off <<= 1) {
            const int v = __shfl_up(ti, off, 64);
            if (lane >= off) ti += v;
        }
        if (lane < NW) s_cnt[16 + lane] = ti - t;  // exclusive prefix of the wavefront totals
        if (lane == NW - 1) s_total = ti;
    }
    __syncthreads();
    int base = s_cnt[16 + wv] + incl - c;
    for (int i = lo; i < hi; i++) {
        if (pass[i]) {
            const int r = 2 * base;
            const int s = EKF_BASE + 3 * i;
            a.last_klt[2 * i] = z[2 * i];
            a.last_klt[2 * i + 1] = z[2 * i + 1];
            a.idx[r] = s;
            a.idx[r + 1] = s + 1;
            a.zrow[r] = z[2 * i];
            a.zrow[r + 1] = z[2 * i + 1];
            // Rm[2c] = R(c,c), Rm[2c+1] = R(c^1,c) (the off-diagonal element of column c);
            // input is a column-major 2x2: [cov(0,0), cov(1,0), cov(0,1), cov(1,1)]
            a.Rm[2 * r] = R[4 * i + 0];
            a.Rm[2 * r + 1] = R[4 * i + 1];
            a.Rm[2 * r + 2] = R[4 * i + 3];
            a.Rm[2 * r + 3] = R[4 * i + 2];
            a.inv_idx[s] = r;
            a.inv_idx[s + 1] = r + 1;
            a.inv_idx[s + 2] = -1;
            base++;
        } else {
            a.del_flag[i] = 1;
            a.inv_idx[EKF_BASE + 3 * i] = a.inv_idx[EKF_BASE + 3 * i + 1] = a.inv_idx[EKF_BASE + 3 * i + 2] = -1;
        }
    }
    __syncthreads();
    const int m = 2 * s_total;
    if (tid == 0 && a.m_out) *a.m_out = m;
    for (int r = m + tid; r < m_pad; r += NT) {
        a.idx[r] = -1;
        a.zrow[r] = 0.f;
        a.Rm[2 * r] = 0.f;
        a.Rm[2 * r + 1] = 0.f;
    }
    if (tid <= EKF_BASE) a.inv_idx[tid == EKF_BASE ? EKF_BASE + 3 * N : tid] = -1;  // base state and the K*y column
}

__global__ __launch_bounds__(256) void linearize_kernel(const float* __restrict__ mu, int N, float dt, float* FA,
                                                        float* FB, float* FD, float* mu_next, BookArgs book) {
    // device-resident sequences: the measurement bookkeeping of the coming update rides along as one more
    // workgroup (it does not depend on the propagated state)
    if (book.enabled && blockIdx.x == gridDim.x - 1) {
        bookkeeping_body<256>(book);
        return;
    }
    const int base_block = (int)gridDim.x - 1 - (book.enabled ? 1 : 0);
    __shared__ LinLds l;
    linearize_block([mu](int e) { return mu[e]; }, N, dt, FA, FB, FD, mu_next, (int)blockIdx.x, base_block, l, (int)threadIdx.x);
}

// Scatter the Jacobian blocks into a dense n x n matrix (column-major, ld), zero elsewhere.
__global__ void build_dense_F_kernel(const float* FA, const float* FB, const float* FD, int N, int n, int ld, float* F) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int j = blockIdx.y;
    if (i >= ld || j >= ld) return;
    float v = 0.f;
    if (i < n && j < n) {
        if (i < EKF_BASE) {
            if (j < EKF_BASE) v = FA[j * EKF_BASE + i];
        } else {
            int f = (i - EKF_BASE) / 3, r = (i - EKF_BASE) % 3;
            if (j >= 7 && j <= 15)
                v = FB[(size_t)f * 27 + (j - 7) * 3 + r];
            else if (j >= EKF_BASE && (j - EKF_BASE) / 3 == f)
                v = FD[(size_t)f * 9 + ((j - EKF_BASE) % 3) * 3 + r];
        }
    }
    F[(size_t)j * ld + i] = v;
}

// ---------------------------------------------------------------------------------------
// Structured covariance propagation.  F = [[A 0],[B D]] with B non-zero only in state
// columns 7..15 and D block-diagonal 3x3, so
//   X = F*P      : row i<22 : sum_{k<22} A(i,k) P(k,j);  landmark row: 9 + 3 terms
//   P'= X*F^T + Q: col j<22 : sum_{k<22} X(i,k) A(j,k);  landmark col: 9 + 3 terms
// Terms are accumulated in ascending state index with separate multiply and add, i.e. the
// order of the reference's sparse products (and of the dense oracle, whose extra terms are
// exact zeros).
// ---------------------------------------------------------------------------------------
__device__ inline float process_noise(int i, float dt) {
    // generateProcessNoise (:123-174)
    if (i < 7) return (float)(0.0001 * (double)dt);
    if (i < 10) return (float)(0.01 * (double)dt);
    if (i < 16) return 5 * dt;
    if (i < 22) return (float)(0.001 * (double)dt);
    return (float)(0.0001 * (double)dt);
}

// ---------------------------------------------------------------------------------------
// Fused structured propagation  P' = F P F^T + Q  in ONE pass (out of place).
// Because F = [[A 0],[B D]], the 3x3 block of P' for the landmark pair (f,g) depends only on
// the 12x12 sub-block P[{7..15} U own(f), {7..15} U own(g)], so a workgroup owning 16x16
// landmark pairs stages the shared parts (9x9 base block, 9x3 / 3x9 strips, Jacobian blocks)
// in LDS and every thread finishes one 3x3 block with the intermediate X = F P kept in
// registers.  Base rows / columns are handled by extra workgroups of the same launch, one
// thread per element.  Every sum runs in ascending state index with separate multiply and
// add (X = F*P first, then X*F^T), i.e. the order of the reference's sparse products.
// ---------------------------------------------------------------------------------------
// ---- the pieces of numericallyLinearizeProcess (:176-325) as device functions, shared by linearize_kernel's
// formulation (one motion per lane) and by the propagation kernel when it linearises in place (LIN = true) ----
// Derivative column e of landmark (u, v, rho): e = 0..8 -> d/d(base state 7+e) (a column of B, :223-253),
// e = 9..11 -> d/d(own component e-9) (a column of D, :262-321).  bm[0] is the unperturbed base motion,
// bm[1+2c] / bm[2+2c] the ones with base state 7+c at +delta / -delta.
__device__ __forceinline__ void lin_column(const BaseMotion* bm, float u, float v, float rho, int e, float td, float out[3]) {
    V3 hi, lo;
    if (e < 9) {
        hi = convolve_feature(bm[1 + 2 * e], u, v, rho);
        lo = convolve_feature(bm[2 + 2 * e], u, v, rho);
    } else {
        const int c = e - 9;
        float t0 = u, t1 = v, t2 = rho;
        if (c == 0) t0 = plus_delta(t0);
        if (c == 1) t1 = plus_delta(t1);
        if (c == 2) t2 = plus_delta(t2);
        hi = convolve_feature(bm[0], t0, t1, t2);
        if (c == 0) t0 = minus_2delta(t0);
        if (c == 1) t1 = minus_2delta(t1);
        if (c == 2) t2 = minus_2delta(t2);
        lo = convolve_feature(bm[0], t0, t1, t2);
    }
    out[0] = (hi.x - lo.x) / td;
    out[1] = (hi.y - lo.y) / td;
    out[2] = (hi.z - lo.z) / td;
}
// base motion variant v of the 19 the landmark rows need (0: unperturbed, 1+2c: column 7+c plus, 2+2c: minus)
__device__ __forceinline__ BaseMotion lin_base_motion(const float* s_base, int v, float dt) {
    float t[EKF_BASE];
#pragma unroll
    for (int i = 0; i < EKF_BASE; i++) t[i] = s_base[i];
    if (v > 0) {
        const int c = 7 + (v - 1) / 2;
#pragma unroll
        for (int i = 7; i < 16; i++) {
            if (i == c) {
                t[i] = plus_delta(t[i]);
                if (((v - 1) & 1) == 1) t[i] = minus_2delta(t[i]);
            }
        }
    }
    return base_motion(t, dt);
}
// convolveBaseState at test point q (column q>>1 at +delta for even q, -delta for odd q) -> out[22]
__device__ __forceinline__ void lin_base_test_point(const float* s_base, int q, float dt, float* out) {
    const int j = q >> 1;
    float t[EKF_BASE], o[EKF_BASE];
#pragma unroll
    for (int i = 0; i < EKF_BASE; i++) t[i] = (i == j) ? plus_delta(s_base[i]) : s_base[i];
    if (q & 1) {
#pragma unroll
        for (int i = 0; i < EKF_BASE; i++) t[i] = (i == j) ? minus_2delta(t[i]) : t[i];
    }
    convolve_base(t, dt, o);
#pragma unroll
    for (int i = 0; i < EKF_BASE; i++) out[i] = o[i];
}

#define PT 16  // landmarks per tile side (landmark x landmark tiles)
#define PC 3   // landmarks per base-row / base-column workgroup (3*PC <= 64)

__device__ __forceinline__ float predict_finish(float acc, int i, int j, float dt) {
    if (i == j) acc = acc + process_noise(i, dt);
    return (fabsf(acc) > EKF_FLUSH_THRESH) ? acc : 0.f;
}

// grid.x = tiles_side^2 landmark tiles + (1 + chunks) base-row workgroups + chunks base-column workgroups
//          (+ 1 bookkeeping workgroup with LIN and book.enabled)
// LIN = true: process(dt) in ONE launch.  Every workgroup forms the Jacobian blocks it needs itself, with the device
// functions linearize_kernel uses (same values, same bits) and straight into LDS: a landmark tile its 2 x 16 B and D
// blocks (384 derivative columns over 256 threads), a base workgroup A (and the B, D of its three landmarks).  The
// propagated mean is written by the diagonal tiles (landmarks) and base workgroup 0; the coming update's measurement
// bookkeeping rides along as one more workgroup.  The redundant arithmetic (each landmark's columns are formed by
// 2 * tiles_side workgroups) is ~1 % of an MI355X-microsecond; what it buys is a kernel boundary and a trip through HBM.
template <bool LIN>
__global__ __launch_bounds__(256) void predict_fused_kernel(const float* __restrict__ P, int ld, int N, int n,
                                                            const float* __restrict__ FA, const float* __restrict__ FB,
                                                            const float* __restrict__ FD, float dt,
                                                            float* __restrict__ Pout, int tiles_side, int chunks,
                                                            const float* __restrict__ mu, float* __restrict__ mu_next, BookArgs book,
                                                            long long* dbg) {
    const int tid = threadIdx.x;
    const int ntile = tiles_side * tiles_side;
    // diagnostic phase stamps (scripts/predict_stamps.py; dbg is null in production): one diagonal tile, one off-diagonal
    // tile, base workgroup 0, one base-row and one base-column workgroup
#define PSTAMP_(slot)                                                                                               \
    do {                                                                                                            \
        if (dbg && tid == 0) {                                                                                      \
            const int b_ = (int)blockIdx.x;                                                                         \
            const int w_ = b_ == 0 ? 0 : b_ == 1 ? 1 : b_ == ntile ? 2 : b_ == ntile + 1 ? 3 : b_ == ntile + chunks + 1 ? 4 : -1; \
            if (w_ >= 0) dbg[900 + 8 * w_ + (slot)] = (long long)__builtin_amdgcn_s_memtime();                      \
        }                                                                                                           \
    } while (0)
    PSTAMP_(0);
    __shared__ float sm[4096];
    __shared__ float s_base[EKF_BASE];
    __shared__ BaseMotion s_bm[19];
    const float td = two_delta();
    if (book.enabled && blockIdx.x == gridDim.x - 1) {  // (LIN, or a step whose linearisation ran inside the previous update's last GEMM)
        bookkeeping_body<256>(book);
        return;
    }
    if (LIN) {
        if (tid < EKF_BASE) s_base[tid] = mu[tid];
    }
    if ((int)blockIdx.x < ntile) {
        float* sPbb = sm;                 // [81]      P(7+a, 7+b) at a*9+b
        float* sPbg = sm + 96;            // [PT][27]  P(7+a, 22+3g+s) at a*3+s
        float* sPfb = sPbg + PT * 27;     // [PT][27]  P(22+3f+q, 7+b) at q*9+b
        float* sBf = sPfb + PT * 27;      // [PT][27]  B_f(r,c) at c*3+r
        float* sBg = sBf + PT * 27;
        float* sDf = sBg + PT * 27;       // [PT][9]   D_f(r,q) at q*3+r
        float* sDg = sDf + PT * 9;
        float* sXb = sDg + PT * 9;        // [PT][27]  X(22+3f+r, 7+b) at r*9+b
        const int tI = blockIdx.x / tiles_side, tJ = blockIdx.x % tiles_side;
        const int f0 = tI * PT, g0 = tJ * PT;
        if (tid < 81) sPbb[tid] = P[(size_t)(7 + tid % 9) * ld + 7 + tid / 9];
        // LIN: the two derivative columns this thread forms: task = side*192 + landmark*12 + column, tasks tid and
        // tid + 256 (< 384); on the diagonal tiles threads 128..143 propagate the landmarks' means in their second slot
        float lu[2] = {0.f, 0.f}, lv[2] = {0.f, 0.f}, lrho[2] = {1.f, 1.f};
        if (LIN) {
#pragma unroll
            for (int sl = 0; sl < 2; sl++) {
                const int task = tid + 256 * sl;
                int lmk = -1;
                if (task < 384) lmk = ((task >= 192) ? g0 : f0) + (task % 192) / 12;
                else if (tI == tJ && tid >= 128 && tid < 128 + PT) lmk = f0 + tid - 128;
                if (lmk >= 0 && lmk < N) {
                    lu[sl] = mu[EKF_BASE + 3 * lmk];
                    lv[sl] = mu[EKF_BASE + 3 * lmk + 1];
                    lrho[sl] = mu[EKF_BASE + 3 * lmk + 2];
                }
            }
        }
#ifndef EKF_PREDICT_LIN_TWO_BARRIERS
        if (LIN) {
            // the staged operands are requested first; while they fly, 19 threads of the last wavefront form the base
            // motions from the base state read straight from global memory (no LDS hop, no extra barrier)
            float pg[2], pf[2];
#pragma unroll
            for (int it = 0; it < 2; it++) {
                const int e = tid + 256 * it;
                const int l = e / 27, w = e % 27;
                const int f = f0 + l, g = g0 + l;
                const bool in = e < PT * 27;
                pg[it] = (in && g < N) ? P[(size_t)(EKF_BASE + 3 * g + w % 3) * ld + 7 + w / 3] : 0.f;
                pf[it] = (in && f < N) ? P[(size_t)(7 + w % 9) * ld + EKF_BASE + 3 * f + w / 9] : 0.f;
            }
            if (tid >= 192 && tid < 192 + 19) {
                float b[EKF_BASE];
#pragma unroll
                for (int i = 0; i < EKF_BASE; i++) b[i] = mu[i];
                s_bm[tid - 192] = lin_base_motion(b, tid - 192, dt);
            }
#pragma unroll
            for (int it = 0; it < 2; it++) {
                const int e = tid + 256 * it;
                if (e < PT * 27) {
                    sPbg[e] = pg[it];
                    sPfb[e] = pf[it];
                }
            }
        } else
#endif
        for (int e = tid; e < PT * 27; e += 256) {
            const int l = e / 27, w = e % 27;
            const int f = f0 + l, g = g0 + l;
            sPbg[e] = (g < N) ? P[(size_t)(EKF_BASE + 3 * g + w % 3) * ld + 7 + w / 3] : 0.f;
            sPfb[e] = (f < N) ? P[(size_t)(7 + w % 9) * ld + EKF_BASE + 3 * f + w / 9] : 0.f;
            if (!LIN) {
                sBf[e] = (f < N) ? FB[(size_t)f * 27 + w] : 0.f;
                sBg[e] = (g < N) ? FB[(size_t)g * 27 + w] : 0.f;
            }
        }
        if (!LIN) {
            for (int e = tid; e < PT * 9; e += 256) {
                const int l = e / 9, w = e % 9;
                sDf[e] = (f0 + l < N) ? FD[(size_t)(f0 + l) * 9 + w] : 0.f;
                sDg[e] = (g0 + l < N) ? FD[(size_t)(g0 + l) * 9 + w] : 0.f;
            }
        }
        __syncthreads();
        PSTAMP_(1);
        if (LIN) {
#ifdef EKF_PREDICT_LIN_TWO_BARRIERS
            if (tid < 19) s_bm[tid] = lin_base_motion(s_base, tid, dt);
            __syncthreads();
#endif
#pragma unroll
            for (int sl = 0; sl < 2; sl++) {
                const int task = tid + 256 * sl;
                if (task < 384) {
                    const int side = task >= 192, l = (task % 192) / 12, e = task % 12;
                    const int lmk = (side ? g0 : f0) + l;
                    float o[3] = {0.f, 0.f, 0.f};
                    if (lmk < N) lin_column(s_bm, lu[sl], lv[sl], lrho[sl], e, td, o);
                    float* dst = (e < 9) ? ((side ? sBg : sBf) + l * 27 + e * 3) : ((side ? sDg : sDf) + l * 9 + (e - 9) * 3);
                    dst[0] = o[0];
                    dst[1] = o[1];
                    dst[2] = o[2];
                } else if (tI == tJ && tid >= 128 && tid < 128 + PT && f0 + tid - 128 < N) {
                    // mean propagation with the OLD base state (:102-104)
                    const V3 o = convolve_feature(s_bm[0], lu[sl], lv[sl], lrho[sl]);
                    const int lmk = f0 + tid - 128;
                    mu_next[EKF_BASE + 3 * lmk] = o.x;
                    mu_next[EKF_BASE + 3 * lmk + 1] = o.y;
                    mu_next[EKF_BASE + 3 * lmk + 2] = o.z;
                }
            }
            __syncthreads();
        }
        PSTAMP_(2);
        // X on the base columns, once per landmark row of the tile
        for (int e = tid; e < PT * 27; e += 256) {
            const int l = e / 27, r = (e % 27) / 9, b = e % 9;
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < 9; c++) acc = acc + sBf[l * 27 + c * 3 + r] * sPbb[c * 9 + b];
#pragma unroll
            for (int q = 0; q < 3; q++) acc = acc + sDf[l * 9 + q * 3 + r] * sPfb[l * 27 + q * 9 + b];
            sXb[e] = acc;
        }
        __syncthreads();
        PSTAMP_(3);
        // Lanes run along f, the ROW landmark: the 16 lanes of a group read / write 48 contiguous floats of a column of Sigma (three
        // floats each, one 12-byte access), four columns groups per wavefront.  (Rounds 1-4 had the lanes along g: sixteen
        // columns per access, 12 bytes apart in each -- 1.8 TB/s of the 8 n^2 bytes at N = 1024, where Sigma is 38 MB.)
        // (Requesting this block in front of the staging instead -- it is not needed before here -- was tried in round 6 and is SLOWER, 9.0 -> 9.3 us
        // at N = 256 and 30 -> 35.5 us at N = 1024: vmcnt retires in order, so the staged operands' first barrier then waits for these loads too.)
        const int lf = tid % PT, lg = tid / PT;
        const int f = f0 + lf, g = g0 + lg;
        if (f >= N || g >= N) return;
        float Pfg[3][3];  // own block P(22+3f+q, 22+3g+s)
#pragma unroll
        for (int sIdx = 0; sIdx < 3; sIdx++) {
            const float3 c3 = *reinterpret_cast<const float3*>(P + (size_t)(EKF_BASE + 3 * g + sIdx) * ld + EKF_BASE + 3 * f);
            Pfg[0][sIdx] = c3.x;
            Pfg[1][sIdx] = c3.y;
            Pfg[2][sIdx] = c3.z;
        }
        float Xg[3][3];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int sIdx = 0; sIdx < 3; sIdx++) {
                float acc = 0.f;
#pragma unroll
                for (int c = 0; c < 9; c++) acc = acc + sBf[lf * 27 + c * 3 + r] * sPbg[lg * 27 + c * 3 + sIdx];
#pragma unroll
                for (int q = 0; q < 3; q++) acc = acc + sDf[lf * 9 + q * 3 + r] * Pfg[q][sIdx];
                Xg[r][sIdx] = acc;
            }
#pragma unroll
        for (int sIdx = 0; sIdx < 3; sIdx++) {
            float o3[3];
#pragma unroll
            for (int r = 0; r < 3; r++) {
                float acc = 0.f;
#pragma unroll
                for (int c = 0; c < 9; c++) acc = acc + sXb[lf * 27 + r * 9 + c] * sBg[lg * 27 + c * 3 + sIdx];
#pragma unroll
                for (int q = 0; q < 3; q++) acc = acc + Xg[r][q] * sDg[lg * 9 + q * 3 + sIdx];
                const int i = EKF_BASE + 3 * f + r, j = EKF_BASE + 3 * g + sIdx;
                o3[r] = predict_finish(acc, i, j, dt);
            }
            *reinterpret_cast<float3*>(Pout + (size_t)(EKF_BASE + 3 * g + sIdx) * ld + EKF_BASE + 3 * f) = make_float3(o3[0], o3[1], o3[2]);
        }
        PSTAMP_(4);
        return;
    }
    const int wb = (int)blockIdx.x - ntile;
    float* sA = sm;                        // [22*22]  A(i,l) at l*22+i
    float* sLB = sm + 3072;                // LIN: [PC][27] B and [PC][9] D of this workgroup's landmarks
    float* sLD = sLB + PC * 27;
    const float* FBp = FB;                 // where this workgroup reads its landmarks' B / D blocks from
    const float* FDp = FD;
    int lm0 = 0;                           // ... and the landmark index that corresponds to offset 0 there
    if (LIN) {
        float* s_hi = sm + 2048;           // [16][22] convolveBaseState at +delta / -delta
        float* s_lo = s_hi + 16 * EKF_BASE;
        lm0 = (wb == 0) ? 0 : ((wb <= chunks) ? (wb - 1) * PC : (wb - chunks - 1) * PC);
        const int lt = tid - 128;          // threads 128..163: the 36 derivative columns of the PC landmarks
        float u = 0.f, v = 0.f, rho = 1.f;
        const bool ltask = wb > 0 && lt >= 0 && lt < PC * 12 && lm0 + lt / 12 < N;
        if (ltask) {
            const int lmk = lm0 + lt / 12;
            u = mu[EKF_BASE + 3 * lmk];
            v = mu[EKF_BASE + 3 * lmk + 1];
            rho = mu[EKF_BASE + 3 * lmk + 2];
        }
        __syncthreads();                   // s_base
        if (tid < 32) {
            lin_base_test_point(s_base, tid, dt, ((tid & 1) ? s_lo : s_hi) + (tid >> 1) * EKF_BASE);
        } else if (tid == 32 && wb == 0) {
            float o[EKF_BASE];
            convolve_base(s_base, dt, o);
#pragma unroll
            for (int i = 0; i < EKF_BASE; i++) mu_next[i] = o[i];
        } else if (tid >= 64 && tid < 64 + 19) {
            s_bm[tid - 64] = lin_base_motion(s_base, tid - 64, dt);
        }
        __syncthreads();
        for (int e = tid; e < EKF_BASE * EKF_BASE; e += 256) {
            const int j = e / EKF_BASE, i = e % EKF_BASE;
            sA[e] = (j < 16) ? (s_hi[j * EKF_BASE + i] - s_lo[j * EKF_BASE + i]) / td : ((i == j) ? 1.f : 0.f);
        }
        if (wb > 0 && lt >= 0 && lt < PC * 12) {
            float o[3] = {0.f, 0.f, 0.f};
            const int l = lt / 12, e = lt % 12;
            if (ltask) lin_column(s_bm, u, v, rho, e, td, o);
            float* dst = (e < 9) ? (sLB + l * 27 + e * 3) : (sLD + l * 9 + (e - 9) * 3);
            dst[0] = o[0];
            dst[1] = o[1];
            dst[2] = o[2];
        }
        FBp = sLB;
        FDp = sLD;
        __syncthreads();  // A and the landmarks' B, D blocks are in LDS
        PSTAMP_(1);
    } else {
        for (int e = tid; e < EKF_BASE * EKF_BASE; e += 256) sA[e] = FA[e];
    }
    if (wb <= chunks) {
        // ---- base rows: i < 22; columns j < 22 (wb == 0) or the 63 columns of landmark chunk wb-1 ----
        float* sX1 = sm + 512;             // [22][22]  X(i,k), k < 22, at k*22+i
        float* sX2 = sm + 1024;            // [63][22]  X(i, j0+jc) at jc*22+i
        __syncthreads();
        for (int e = tid; e < EKF_BASE * EKF_BASE; e += 256) {
            const int i = e % EKF_BASE, k = e / EKF_BASE;
            const float* pc = P + (size_t)k * ld;
            float acc = 0.f;
            for (int l = 0; l < EKF_BASE; l++) acc = acc + sA[l * EKF_BASE + i] * pc[l];
            sX1[e] = acc;
        }
        if (wb == 0) {
            __syncthreads();
            for (int e = tid; e < EKF_BASE * EKF_BASE; e += 256) {
                const int i = e % EKF_BASE, j = e / EKF_BASE;
                float acc = 0.f;
                for (int k = 0; k < EKF_BASE; k++) acc = acc + sX1[k * EKF_BASE + i] * sA[k * EKF_BASE + j];
                Pout[(size_t)j * ld + i] = predict_finish(acc, i, j, dt);
            }
            PSTAMP_(4);
            return;
        }
        const int g0 = (wb - 1) * PC, j0 = EKF_BASE + 3 * g0;
        const int ncol = min(3 * PC, n - j0);
        for (int e = tid; e < EKF_BASE * ncol; e += 256) {
            const int i = e % EKF_BASE, jc = e / EKF_BASE;
            const float* pc = P + (size_t)(j0 + jc) * ld;
            float acc = 0.f;
            for (int l = 0; l < EKF_BASE; l++) acc = acc + sA[l * EKF_BASE + i] * pc[l];
            sX2[e] = acc;
        }
        __syncthreads();
        for (int e = tid; e < EKF_BASE * ncol; e += 256) {
            const int i = e % EKF_BASE, jc = e / EKF_BASE;
            const int g = g0 + jc / 3, sIdx = jc % 3, j = j0 + jc;
            const float* b = FBp + (size_t)(g - lm0) * 27 + sIdx;
            const float* d = FDp + (size_t)(g - lm0) * 9 + sIdx;
            float acc = 0.f;
            for (int c = 0; c < 9; c++) acc = acc + sX1[(7 + c) * EKF_BASE + i] * b[c * 3];
            for (int q = 0; q < 3; q++) acc = acc + sX2[(3 * (jc / 3) + q) * EKF_BASE + i] * d[q * 3];
            Pout[(size_t)j * ld + i] = predict_finish(acc, i, j, dt);
        }
        PSTAMP_(4);
        return;
    }
    // ---- base columns: j < 22; rows of landmark chunk wb-chunks-1 ----
    {
        float* sX3 = sm + 512;             // [22][64]  X(i0+ic, k), k < 22, at k*64+ic
        const int f0 = (wb - chunks - 1) * PC, i0 = EKF_BASE + 3 * f0;
        const int nrow = min(3 * PC, n - i0);
        for (int e = tid; e < nrow * EKF_BASE; e += 256) {
            const int ic = e % nrow, k = e / nrow;
            const int f = f0 + ic / 3, r = ic % 3;
            const float* pc = P + (size_t)k * ld;
            const float* b = FBp + (size_t)(f - lm0) * 27 + r;
            const float* d = FDp + (size_t)(f - lm0) * 9 + r;
            float acc = 0.f;
            for (int c = 0; c < 9; c++) acc = acc + b[c * 3] * pc[7 + c];
            for (int q = 0; q < 3; q++) acc = acc + d[q * 3] * pc[EKF_BASE + 3 * f + q];
            sX3[k * 64 + ic] = acc;
        }
        __syncthreads();
        for (int e = tid; e < nrow * EKF_BASE; e += 256) {
            const int ic = e % nrow, j = e / nrow;
            float acc = 0.f;
            for (int k = 0; k < EKF_BASE; k++) acc = acc + sX3[k * 64 + ic] * sA[k * EKF_BASE + j];
            const int i = i0 + ic;
            Pout[(size_t)j * ld + i] = predict_finish(acc, i, j, dt);
        }
        PSTAMP_(4);
    }
#undef PSTAMP_
}

// dense mode epilogue: P += Q(dt) on the diagonal, then prune
__global__ void add_noise_flush_kernel(float* P, int ld, int n, float dt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (i >= n) return;
    float v = P[(size_t)j * ld + i];
    if (i == j) v = v + process_noise(i, dt);
    if (!(fabsf(v) > EKF_FLUSH_THRESH)) v = 0.f;
    P[(size_t)j * ld + i] = v;
}

// ---------------------------------------------------------------------------------------
// Update bookkeeping (:486-541, :554-555): one workgroup scans the pass flags into the
// measurement map idx (formFeatureMeasurementMap :634-661), stores last_klt / delete flags,
// z - H*mu and the per-row measurement noise.  Integer work is bit-exact by construction.
// ---------------------------------------------------------------------------------------
// With a frame counter (device-resident sequences replayed from a hipGraph) the measurement
// of frame *frame_counter is used and the counter advances modulo `frames`.
__global__ __launch_bounds__(1024) void update_bookkeeping_kernel(BookArgs a) { bookkeeping_body<1024>(a); }

#include "gather_body.inc"

__global__ __launch_bounds__(256) void gather_kernel(GatherArgs a) {
    __shared__ float tile[64 * 65];
    gather_body(a, blockIdx.x, tile);
}

// (G = K*R - T[:, idx] and mu += K*y are epilogues of the two Joseph GEMMs: see GemmEpi.)
// Stand-alone mean update, used when no landmark was measured (m = 0):
// mu += K*y (:600), quaternion renormalisation (:605-609).  64 state rows per workgroup;
// the four wavefronts each sum a quarter of the measurement rows (coalesced along the state
// index), combined in ascending order.
__global__ __launch_bounds__(256) void mean_update_kernel(const float* __restrict__ K, int ld, int n, int m,
                                                          const float* __restrict__ y, float* mu, int* frame_counter,
                                                          int frames) {
    __shared__ float s_part[4][64];
    __shared__ float s_q[4];
    const int rl = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + rl;
    const int chunk = (m + 3) / 4;
    const int r0 = w * chunk, r1 = min(m, r0 + chunk);
    float acc = 0.f;
    if (i < n) {
#pragma unroll 8
        for (int r = r0; r < r1; r++) acc = acc + K[(size_t)r * ld + i] * y[r];
    }
    s_part[w][rl] = acc;
    __syncthreads();
    float v = 0.f;
    if (w == 0 && i < n) v = mu[i] + (((s_part[0][rl] + s_part[1][rl]) + s_part[2][rl]) + s_part[3][rl]);
    if (blockIdx.x == 0) {
        if (w == 0 && rl >= 3 && rl <= 6) s_q[rl - 3] = v;
        __syncthreads();
        if (w == 0 && rl >= 3 && rl <= 6) {
            float qn = sqrtf(s_q[0] * s_q[0] + s_q[1] * s_q[1] + s_q[2] * s_q[2] + s_q[3] * s_q[3]);
            v = v / qn;
        }
    }
    if (w == 0 && i < n) mu[i] = v;
    if (frame_counter && blockIdx.x == 0 && threadIdx.x == 0) {
        const int fi = *frame_counter + 1;
        *frame_counter = (fi >= frames) ? 0 : fi;
    }
}

__global__ void check_sigma_kernel(const float* P, int ld, int n, float* out) {
    // out[0] = min diagonal, out[1] = max |P(i,j)-P(j,i)|; single block
    __shared__ float s_min[256], s_max[256];
    float mn = 3.4e38f, mx = 0.f;
    for (size_t e = threadIdx.x; e < (size_t)n * n; e += 256) {
        int i = e % n, j = e / n;
        if (i == j) mn = fminf(mn, P[(size_t)j * ld + i]);
        if (i > j) mx = fmaxf(mx, fabsf(P[(size_t)j * ld + i] - P[(size_t)i * ld + j]));
    }
    s_min[threadIdx.x] = mn;
    s_max[threadIdx.x] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            s_min[threadIdx.x] = fminf(s_min[threadIdx.x], s_min[threadIdx.x + s]);
            s_max[threadIdx.x] = fmaxf(s_max[threadIdx.x], s_max[threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = s_min[0];
        out[1] = s_max[0];
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------
void launch_linearize(ekfvio_filter* f, float dt, const BookArgs* book) {
    ProfScope ps(f, PC_LINEARIZE);
    BookArgs b;
    if (book) b = *book;
    // landmark workgroups + the base-block workgroup (+ the bookkeeping workgroup)
    const int blocks = (f->N + LIN_LM - 1) / LIN_LM + 1 + (b.enabled ? 1 : 0);
    hipLaunchKernelGGL(linearize_kernel, dim3(blocks), dim3(256), 0, f->stream, f->mu, f->N, dt, f->FA, f->FB, f->FD,
                       f->mu_next, b);
}

BookArgs make_book_args(ekfvio_filter* f, int m, const float* d_z, const float* d_R, const uint8_t* d_pass, const int* d_frame_counter) {
    BookArgs a;
    a.enabled = 1;
    a.N = f->N;
    a.m_pad = round_up(m > 0 ? m : 1, EKF_TILE);
    a.z = d_z;
    a.R = d_R;
    a.pass = d_pass;
    a.last_klt = f->last_klt;
    a.del_flag = f->del_flag;
    a.idx = f->idx;
    a.inv_idx = f->inv_idx;
    a.zrow = f->yres;
    a.Rm = f->Rm;
    a.frame_counter = d_frame_counter;
    return a;
}

void launch_build_dense_F(ekfvio_filter* f, float* Fdense) {
    dim3 grid((f->ldp + 255) / 256, f->ldp);
    hipLaunchKernelGGL(build_dense_F_kernel, grid, dim3(256), 0, f->stream, f->FA, f->FB, f->FD, f->N, f->n, f->ldp,
                       Fdense);
}

// process(dt) (:96-121)
void launch_predict(ekfvio_filter* f, float dt, const BookArgs* book) {
    // structured mode linearises inside the propagation kernel (predict_fused_kernel<true>); the dense mode and
    // ekfvio_linearize keep the stand-alone linearize_kernel
    // (while the landmark tiles are a few rounds of workgroups: with thousands of tiles, N = 1024, the Jacobian blocks
    // formed 2 * 64 times over cost more than the launch they save)
    const int tiles_side = (f->N + PT - 1) / PT;
    // round 6: the previous update's last GEMM has linearised for this step already (launch_update, GemmEpi::lin_blocks): FA / FB / FD and the
    // propagated mean are in place, the propagation kernel only carries the bookkeeping along
    const bool pre = f->prelinearized && f->cfg.predict_mode != EKFVIO_PREDICT_DENSE;
    f->prelinearized = false;
    const bool lin_in_predict = !pre && f->cfg.predict_mode != EKFVIO_PREDICT_DENSE && f->fuse_linearize && tiles_side * tiles_side <= 4 * f->num_cus;
    if (!lin_in_predict && !pre) launch_linearize(f, dt, book);
    const int n = f->n, ld = f->ldp;
    dim3 grid((n + 255) / 256, n);
    if (f->cfg.predict_mode == EKFVIO_PREDICT_DENSE) {
        launch_build_dense_F(f, f->Fdense);
        const int np = round_up(n, 32);
        {
            ProfScope ps(f, PC_GEMM_PREDICT, 4.0 * n * (double)n * n);
            // X = F*P (B = P is [K x N]); P' = X*F^T (B = F is [N x K])
            launch_gemm(f, 0, n, n, np, 1.f, f->Fdense, ld, f->P, ld, 0.f, nullptr, 0, f->P2, ld, 0);
            launch_gemm(f, 1, n, n, np, 1.f, f->P2, ld, f->Fdense, ld, 0.f, nullptr, 0, f->P, ld, 0);
        }
        ProfScope ps(f, PC_PREDICT);
        hipLaunchKernelGGL(add_noise_flush_kernel, grid, dim3(256), 0, f->stream, f->P, ld, n, dt);
    } else {
        ProfScope ps(f, PC_PREDICT, 4.0 * n * (358.0 + 36.0 * f->N));
        const int ts = (f->N + PT - 1) / PT;
        const int chunks = (f->N + PC - 1) / PC;
        BookArgs b;
        if (book && (lin_in_predict || pre)) b = *book;
        if (lin_in_predict)
            hipLaunchKernelGGL(predict_fused_kernel<true>, dim3(ts * ts + 1 + 2 * chunks + (b.enabled ? 1 : 0)), dim3(256), 0, f->stream,
                               f->P, ld, f->N, n, f->FA, f->FB, f->FD, dt, f->P2, ts, chunks, f->mu, f->mu_next, b, f->sweep_dbg);
        else
            hipLaunchKernelGGL(predict_fused_kernel<false>, dim3(ts * ts + 1 + 2 * chunks + (b.enabled ? 1 : 0)), dim3(256), 0, f->stream, f->P, ld, f->N, n,
                               f->FA, f->FB, f->FD, dt, f->P2, ts, chunks, f->mu, f->mu_next, b, f->sweep_dbg);
        std::swap(f->P, f->P2);  // out of place; P2's padding is zero as well (never written outside n x n)
    }
    // the propagated mean becomes the state (landmarks used the OLD base state, :102-107)
    std::swap(f->mu, f->mu_next);
}

// updateWithFeaturePositions (:475-628) on device-resident z/R/pass; m = 2*(#passed) known to the host
void launch_update(ekfvio_filter* f, int m, const float* d_z, const float* d_R, const uint8_t* d_pass, int* d_frame_counter,
                   int frames, bool bookkeeping_done, bool m_on_device) {
    if (m_on_device) m = 2 * f->N;  // upper bound: sizes the launches; the kernels read the true count from f->info[2]
    const int n = f->n, ld = f->ldp;
    const int m_pad = round_up(m > 0 ? m : 1, EKF_TILE);
    const int n_pad = round_up(n, EKF_TILE);
    const int lda = f->ld_aug;
    f->last_m = m;
    // The gather and the first diagonal tile's factorisation share a launch while that launch is a single round of
    // workgroups at one per compute unit (chol.hip: gather_potrf_kernel); beyond that (N = 1024: thousands of gather
    // workgroups) the gather wants several workgroups per compute unit and the two stay separate.
    // T = Sigma - X A^-1 X^T and K = X A^-1 as Schur tiles of the sweep (no gain GEMM, no first Joseph GEMM, no (H Sigma)^T)
    const bool schur = m > 0 && sweep_supports_schur(f, m_pad);
    // round 4: where the persistent sweep applies, the gather and the first tile are part of ITS launch (chol.hip, launch_persist_fused)
    const bool fused_front = m > 0 && !schur && f->fuse_sweep && f->fuse_gather && sweep_is_persistent(f, m_pad, n_pad);
    bool fused_gather = false;
    if (m > 0 && f->fuse_gather && !fused_front) {
        const int gx = (std::max(ld, m_pad) + 255) / 256;
        fused_gather = 1 + gx * ((m_pad + GC * GCI - 1) / (GC * GCI)) + (schur ? 0 : (m_pad / 64) * (ld / 64)) <= f->num_cus;
    }
    {
        ProfScope ps(f, PC_GATHER);
        if (!bookkeeping_done) {
            BookArgs bk = make_book_args(f, m, d_z, d_R, d_pass, d_frame_counter);
            if (m_on_device) bk.m_out = f->info + 2;
            hipLaunchKernelGGL(update_bookkeeping_kernel, dim3(1), dim3(1024), 0, f->stream, bk);
        }
        if (m > 0 && !fused_front) {
            if (fused_gather) {
                // the gather and the factorisation of the first diagonal tile share one launch (chol.hip)
                launch_gather_potrf(f, m, m_pad, n_pad, m_on_device, !schur);
            } else {
                GatherArgs ga = make_gather_args(f, m, m_pad, n_pad);
                if (m_on_device) ga.m_dev = f->info + 2;
                const int nb2 = schur ? 0 : (m_pad / 64) * (ld / 64);  // 64x64 transposing tiles of Wt
                hipLaunchKernelGGL(gather_kernel, dim3(ga.nb1 + nb2), dim3(256), 0, f->stream, ga);
            }
        }
    }
    GemmEpi e2;
    e2.mode = 2;
    e2.mu = f->mu;
    e2.Pcol = f->P + (size_t)n * ld;
    e2.n = n;
    e2.frame_counter = d_frame_counter;
    e2.frames = frames;
    e2.sym = f->sym_joseph;  // (heeded by the throughput-regime kernel behind the two-GEMM flow: gemm.hip)
    if (schur) {
        // [A; Sigma H^T; I] swept with Sigma and the gain as Schur tiles (chol.hip): K = X A^-1 (:577-580) and, in place,
        // T2 = Sigma - X A^-1 X^T = Sigma (I - K H)^T, the RIGHT Joseph factor applied (X = Sigma H^T; Sigma is symmetric only
        // to rounding, so this is not the transpose of the reference's (I - K H) Sigma).  Then K pruned,
        // G' = K R^T - (H T2)^T and K y; then the left factor: Sigma' = (I - K H) T2 + K R K^T = T2 + K G'^T, which is the
        // reference's (I - K H) Sigma (I - K H)^T + K R K^T (:594-596), pruned (:625).  Its first workgroup finishes the
        // mean (:600-609).
        launch_chol_sweep(f, f->Saug, f->Laug, f->Linv, m_pad, n_pad, lda, fused_gather, true);
        if (f->publish_after_sweep_seq) launch_publish_status(f, f->publish_after_sweep_seq), f->publish_after_sweep_seq = 0;
        launch_joseph_g(f, m, m_pad, n_pad, m_on_device);
        ProfScope ps(f, PC_GEMM_UPDATE, 2.0 * n * (double)n * m_pad, 1);
        e2.abort = nullptr;
        e2.mode = 3;  // the mean takes K y from joseph_g_kernel's partial sums
        e2.Kyp = f->Wt;
        e2.kyp_blocks = m_pad / 64;
        e2.kyp_ld = ld;
        launch_gemm(f, 1, n, n, m_pad, 1.f, f->Km, ld, f->Gm, ld, 1.f, f->P, ld, f->P, ld, 1, 0, &e2);
    } else if (m > 0) {
        // [A; Sigma H^T; I] -> [L; Y; L^-T], then K = (Sigma H^T) A^-1  (:577-580)
        if (fused_front) launch_persist_fused(f, m, m_pad, n_pad, m_on_device);
        else launch_chol_sweep(f, f->Saug, f->Laug, f->Linv, m_pad, n_pad, lda, fused_gather);
        // (ekfvio_update: the update's status is final here -- the host gets it now and returns while the GEMMs below run)
        if (f->publish_after_sweep_seq) launch_publish_status(f, f->publish_after_sweep_seq), f->publish_after_sweep_seq = 0;
        if (f->sweep_abort_word) {  // a persistent sweep ran: the update's last GEMM leaves its flags zero for the next one
            e2.zero_words = f->sweep_sync;
            e2.n_zero = persist_zero_words(m_pad, n_pad);
            f->sweep_flags_clean = true;
        }
        GemmEpi e1;
        e1.mode = 1;
        e1.inv_idx = f->inv_idx;
        e1.Rm = f->Rm;
        e1.G = f->Gm;
        e1.ldg = ld;
        e1.abort = e2.abort = f->sweep_abort_word;  // a persistent sweep that gave up: both GEMMs write nothing
        if (t2_flow_shape(f, m_pad, n_pad)) {
            // Round 6: T2 = Sigma - Y S Y^T = Sigma (I - K H)^T came out of the persistent launch itself (freed owners, chol_persist.inc t2_tile; behind
            // any other sweep of such a shape gain2_t2_tiles_kernel forms the same T2 from the stored panel blocks), in the dense-F buffer; the gain
            // tiles of the same launch left K, G' = K R^T - (H T2)^T and the partial sums of K y.  The left Joseph factor is the ONE P-update GEMM:
            // Sigma' = (I - K H) T2 + K R K^T = T2 + K G'^T (:594-596), pruned (:625), into f->P; its extra workgroup finishes the mean (:600-609).
            f->t2_updates++;
            if (!f->t2_in_sweep) launch_gain2_tiles(f, m, m_pad, n_pad, m_on_device);  // (K, G', K y and T2 in one launch behind the per-step sweep)
            if (f->between_joseph) {  // (ekfvio_step_image: the frame's outputs, from mu and the partial sums of K y, in front of the one GEMM)
                f->hook_kyp_blocks = m_pad / 64;
                f->between_joseph(f);
                f->hook_kyp_blocks = 0;
            }
            ProfScope ps(f, PC_GEMM_UPDATE, 2.0 * n * (double)n * m_pad, 1);
            e2.mode = 3;  // the mean takes K y from the gain tiles' partial sums (gain_tile2: Kyp = Wt, one row of sums per block column)
            e2.Kyp = f->Wt;
            e2.kyp_blocks = m_pad / 64;
            e2.kyp_ld = ld;
            e2.abort = f->sweep_abort_word;  // a persistent sweep that gave up: the GEMM writes nothing (T2, K, G' are scratch)
            const int lin_blocks = (f->N + LIN_LM - 1) / LIN_LM + 1;
            if (f->lin_next_dt >= 0.f && f->lin_overlap && f->cfg.predict_mode != EKFVIO_PREDICT_DENSE && f->fuse_linearize &&
                gemm_single_round_with(f, n, n, m_pad, 1 + lin_blocks)) {
                // a device-resident run (capture_steps): the next process(dt)'s linearisation and mean propagation ride in this launch, in workgroups
                // of their own behind the tiles' (K y is final: the gain tiles' partial sums); launch_predict then only propagates Sigma
                e2.lin_blocks = lin_blocks;
                e2.lin_N = f->N;
                e2.lin_dt = f->lin_next_dt;
                e2.lin_FA = f->FA, e2.lin_FB = f->FB, e2.lin_FD = f->FD;
                e2.lin_mu_next = f->mu_next;
                f->prelinearized = true;
            }
            launch_gemm(f, 1, n, n, m_pad, 1.f, f->Km, ld, f->Gm, ld, 1.f, t2_buffer(f), ld, f->P, ld, 1, 0, &e2);
        } else {
            if (!f->gain_in_sweep) launch_gain_from_sweep(f, f->Laug, m_pad, n_pad, lda, n, f->Km, f->Gm, ld, 0);
            // The two P-update GEMMs, back to back (one profiler scope, two launches), both triangles:
            //   T = Sigma - K*(H Sigma)   (I_KH * Sigma, :594) in place; also G and K*y (column n)
            //   Sigma' = T + G*K^T, pruned (:594-596, :625); workgroup (0,0) finishes the mean
            ProfScope ps(f, PC_GEMM_UPDATE, 2.0 * n * (double)(n + 1) * m_pad + 2.0 * n * (double)n * m_pad, 2);
            launch_gemm(f, 1, n, n + 1, m_pad, -1.f, f->Km, ld, f->Wt, ld, 1.f, f->P, ld, f->P, ld, 0, 0, &e1);
            if (f->between_joseph) f->between_joseph(f);  // (ekfvio_step_image: the frame's outputs, from mu and K y in column n of P)
            launch_gemm(f, 1, n, n, m_pad, 1.f, f->Gm, ld, f->Km, ld, 1.f, f->P, ld, f->P, ld, 1, 0, &e2);
        }
    } else {
        // no measurement: products are empty, only the quaternion renormalisation remains (:605-609)
        ProfScope ps(f, PC_UPDATE_MISC);
        hipLaunchKernelGGL(mean_update_kernel, dim3((n + 63) / 64), dim3(256), 0, f->stream, f->Km, ld, n, 0, f->yres, f->mu,
                           d_frame_counter, frames);
    }
}

int launch_update_gemms_scratch(ekfvio_filter* f, int m, int reps) {
    const int n = f->n, ld = f->ldp;
    const int m_pad = round_up(m > 0 ? m : 1, EKF_TILE);
    GemmEpi e1, e2;
    if (sweep_supports_schur(f, m_pad) || t2_flow_shape(f, m_pad, round_up(n, EKF_TILE))) {
        // with the Schur sweep, and where T2 comes out of the persistent launch, the update has ONE P-update GEMM: Sigma' = T2 + K G'^T
        e2.mode = 2;
        for (int r = 0; r < reps; r++)
            launch_gemm(f, 1, n, n, m_pad, 1.f, f->Km, ld, f->Gm, ld, 1.f, f->P, ld, f->P2, ld, 1, 0, &e2);
        return 1;
    }
    e1.mode = 1;
    e1.inv_idx = f->inv_idx;
    e1.Rm = f->Rm;
    e1.G = f->Gm;
    e1.ldg = ld;
    e2.mode = 2;  // n = 0: no mean update, no frame counter
    e2.sym = f->sym_joseph;
    for (int r = 0; r < reps; r++) {
        launch_gemm(f, 1, n, n + 1, m_pad, -1.f, f->Km, ld, f->Wt, ld, 1.f, f->P, ld, f->P2, ld, 0, 0, &e1);
        launch_gemm(f, 1, n, n, m_pad, 1.f, f->Gm, ld, f->Km, ld, 1.f, f->P2, ld, f->P2, ld, 1, 0, &e2);
    }
    return 2;
}

void launch_check_sigma(ekfvio_filter* f, float* d_out) {
    hipLaunchKernelGGL(check_sigma_kernel, dim3(1), dim3(256), 0, f->stream, f->P, f->ldp, f->n, d_out);
}
