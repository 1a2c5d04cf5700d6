// ekf_vio_amd/csrc/imu.hip -- SURVEY 8(f) F4, second half: an IMU measurement update behind cfg.use_imu.
//
// The reference subscribes to the IMU topic (EKFVIO.cpp:79-81) but its callback only logs (:113-115) and its
// imu_update_buffer is never touched (EKFVIO.h:59-64): there is nothing to restate.  What runs here is specified by
// the CPU specification kept with the test oracle (imu_update in oracle/) and tested against it: every IMU record first propagates the filter to its stamp
// (process(dt), the reference's own motion model), then updates with
//     z = [gyro; accel],  h(x) = [omega + b_gyr ;  a + b_acc - R(q)^T g]
// (state indices: q 3-6, omega 10-12, a 13-15, b_acc 16-18, b_gyr 19-21; g = cfg.gravity in the world frame), H the
// analytic Jacobian, R = diag(gyro_var x3, accel_var x3), in the reference's Joseph form (:559-609).  Six measurement rows
// touch sixteen state columns, so the update is two small launches: gains per state row (S is 6 x 6: factored by every
// workgroup for itself), then one elementwise pass Sigma <- Sigma - K (H Sigma) + G K^T.  With cfg.use_imu = 0 (default)
// ekfvio_imu stays the reference's no-op.
#include "common.h"

namespace {

struct ImuArgs {
    float gyro[3], accel[3];
    float gyro_var, accel_var;
    float g[3];
};

__constant__ const int kImuCols[16] = {3, 4, 5, 6, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21};

__device__ inline void cross3f(const float* a, const float* b, float* o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

// R(q)^T g with the filter's rotation formula on the conjugate (q not normalised), and d/d(w,x,y,z) (row-major 3 x 4)
__device__ inline void rt_gravity(const float* q, const float* g, float* out, float* jac) {
    const float w = q[0];
    const float c[3] = {-q[1], -q[2], -q[3]};
    float uv[3], cu[3];
    cross3f(c, g, uv);
    for (int k = 0; k < 3; k++) uv[k] = uv[k] + uv[k];
    cross3f(c, uv, cu);
    for (int k = 0; k < 3; k++) {
        out[k] = g[k] + w * uv[k] + cu[k];
        jac[4 * k] = uv[k];
    }
    for (int k = 0; k < 3; k++) {
        const float e[3] = {k == 0 ? 1.f : 0.f, k == 1 ? 1.f : 0.f, k == 2 ? 1.f : 0.f};
        float ev[3], a[3], b[3];
        cross3f(e, g, ev);
        for (int r = 0; r < 3; r++) ev[r] = ev[r] + ev[r];
        cross3f(e, uv, a);
        cross3f(c, ev, b);
        for (int r = 0; r < 3; r++) jac[4 * r + 1 + k] = -(w * ev[r] + a[r] + b[r]);
    }
}

// gains, G = K R - T H^T, W = H Sigma and the updated mean, one state row per thread
__global__ __launch_bounds__(256) void imu_gain_kernel(const float* __restrict__ P, int ld, int n, const float* __restrict__ mu,
                                                       float* __restrict__ mu_out, ImuArgs a, float* __restrict__ K6,
                                                       float* __restrict__ G6, float* __restrict__ W6) {
    __shared__ float sH[6][16];      // H on its sixteen columns
    __shared__ float sPb[16][16];    // Sigma(cols[r], cols[c])
    __shared__ float sWb[6][16];     // (H Sigma)(s, cols[c])
    __shared__ float sS[6][6], sL[6][6];
    __shared__ float sy[6], sq[4];
    const int tid = threadIdx.x;
    const int i = blockIdx.x * 256 + tid;
    if (tid == 0) {
        float q[4] = {mu[3], mu[4], mu[5], mu[6]};
        float rg[3], jac[12];
        rt_gravity(q, a.g, rg, jac);
        for (int r = 0; r < 6; r++)
            for (int c = 0; c < 16; c++) sH[r][c] = 0.f;
        for (int r = 0; r < 3; r++) {
            sH[r][4 + r] = 1.f;
            sH[r][13 + r] = 1.f;
            sH[3 + r][7 + r] = 1.f;
            sH[3 + r][10 + r] = 1.f;
            for (int k = 0; k < 4; k++) sH[3 + r][k] = -jac[4 * r + k];
            sy[r] = a.gyro[r] - (mu[10 + r] + mu[19 + r]);
            sy[3 + r] = a.accel[r] - ((mu[13 + r] + mu[16 + r]) - rg[r]);
        }
    }
    {
        const int r = tid >> 4, c = tid & 15;  // 256 threads = the 16 x 16 block
        sPb[r][c] = P[(size_t)kImuCols[c] * ld + kImuCols[r]];
    }
    __syncthreads();
    if (tid < 96) {  // W on the sixteen columns: (H Sigma)(s, cols[c])
        const int s = tid / 16, c = tid % 16;
        float acc = 0.f;
        for (int cc = 0; cc < 16; cc++) acc = acc + sH[s][cc] * sPb[cc][c];
        sWb[s][c] = acc;
    }
    __syncthreads();
    if (tid < 36) {  // S = (H Sigma) H^T + R
        const int r = tid / 6, s = tid % 6;
        float acc = 0.f;
        for (int c = 0; c < 16; c++) acc = acc + sWb[r][c] * sH[s][c];
        sS[r][s] = acc + (r == s ? (r < 3 ? a.gyro_var : a.accel_var) : 0.f);
    }
    __syncthreads();
    if (tid == 0) {  // Cholesky of the lower triangle of S
        for (int j = 0; j < 6; j++) {
            float d = sS[j][j];
            for (int k = 0; k < j; k++) d = d - sL[j][k] * sL[j][k];
            sL[j][j] = sqrtf(d);
            for (int r = j + 1; r < 6; r++) {
                float v = sS[r][j];
                for (int k = 0; k < j; k++) v = v - sL[r][k] * sL[j][k];
                sL[r][j] = v / sL[j][j];
            }
        }
    }
    __syncthreads();
    if (i < n) {
        float pc[16], x[6], w[6], kk[6];
#pragma unroll
        for (int c = 0; c < 16; c++) pc[c] = P[(size_t)kImuCols[c] * ld + i];  // Sigma(i, cols[c]): coalesced over i
#pragma unroll
        for (int r = 0; r < 6; r++) {
            float ax = 0.f;
#pragma unroll
            for (int c = 0; c < 16; c++) ax = ax + pc[c] * sH[r][c];
            x[r] = ax;
        }
        float pr[16];
#pragma unroll
        for (int c = 0; c < 16; c++) pr[c] = P[(size_t)i * ld + kImuCols[c]];  // Sigma(cols[c], i)
#pragma unroll
        for (int r = 0; r < 6; r++) {
            float aw = 0.f;
#pragma unroll
            for (int c = 0; c < 16; c++) aw = aw + sH[r][c] * pr[c];
            w[r] = aw;
        }
        // k = x S^-1 = (x L^-T) L^-1
#pragma unroll
        for (int r = 0; r < 6; r++) {
            float v = x[r];
#pragma unroll
            for (int k = 0; k < r; k++) v = v - kk[k] * sL[r][k];
            kk[r] = v / sL[r][r];
        }
#pragma unroll
        for (int r = 5; r >= 0; r--) {
            float v = kk[r];
#pragma unroll
            for (int k = r + 1; k < 6; k++) v = v - kk[k] * sL[k][r];
            kk[r] = v / sL[r][r];
        }
        // T on the sixteen columns, G = K R - T H^T
        float tc[16];
#pragma unroll
        for (int c = 0; c < 16; c++) {
            float t = pc[c];
#pragma unroll
            for (int s = 0; s < 6; s++) t = t - kk[s] * sWb[s][c];
            tc[c] = t;
        }
        float dm = 0.f;
#pragma unroll
        for (int r = 0; r < 6; r++) {
            float th = 0.f;
#pragma unroll
            for (int c = 0; c < 16; c++) th = th + tc[c] * sH[r][c];
            const float g = kk[r] * (r < 3 ? a.gyro_var : a.accel_var) - th;
            K6[(size_t)r * ld + i] = kk[r];
            G6[(size_t)r * ld + i] = g;
            W6[(size_t)r * ld + i] = w[r];
            dm = dm + kk[r] * sy[r];
        }
        const float v = mu[i] + dm;
        if (i >= 3 && i <= 6) sq[i - 3] = v;  // block 0
        else mu_out[i] = v;
    }
    if (blockIdx.x == 0) {
        __syncthreads();
        if (tid < 4) {
            const float qn = sqrtf(sq[0] * sq[0] + sq[1] * sq[1] + sq[2] * sq[2] + sq[3] * sq[3]);
            mu_out[3 + tid] = sq[tid] / qn;
        }
    }
}

// Sigma <- Sigma - K (H Sigma) + G K^T, pruned: one element per thread, column per blockIdx.y
__global__ __launch_bounds__(256) void imu_joseph_kernel(float* __restrict__ P, int ld, int n, const float* __restrict__ K6,
                                                         const float* __restrict__ G6, const float* __restrict__ W6) {
    const int i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (i >= n) return;
    float v = P[(size_t)j * ld + i];
#pragma unroll
    for (int s = 0; s < 6; s++) v = v - K6[(size_t)s * ld + i] * W6[(size_t)s * ld + j];
#pragma unroll
    for (int s = 0; s < 6; s++) v = v + G6[(size_t)s * ld + i] * K6[(size_t)s * ld + j];
    P[(size_t)j * ld + i] = (fabsf(v) > EKF_FLUSH_THRESH) ? v : 0.f;
}

}  // namespace

void launch_imu_update(ekfvio_filter* f, const float gyro[3], const float accel[3]) {
    ImuArgs a;
    for (int k = 0; k < 3; k++) {
        a.gyro[k] = gyro[k];
        a.accel[k] = accel[k];
        a.g[k] = f->cfg.gravity[k];
    }
    a.gyro_var = f->cfg.imu_gyro_variance;
    a.accel_var = f->cfg.imu_accel_variance;
    const int n = f->n, ld = f->ldp;
    ProfScope ps(f, PC_UPDATE_MISC, 0, 2);
    // the three n x 6 work matrices live in the first six columns of the camera update's buffers (free between updates)
    hipLaunchKernelGGL(imu_gain_kernel, dim3((n + 255) / 256), dim3(256), 0, f->stream, f->P, ld, n, f->mu, f->mu_next, a, f->Km, f->Gm,
                       f->Wt);
    hipLaunchKernelGGL(imu_joseph_kernel, dim3((n + 255) / 256, n), dim3(256), 0, f->stream, f->P, ld, n, f->Km, f->Gm, f->Wt);
    std::swap(f->mu, f->mu_next);
}
