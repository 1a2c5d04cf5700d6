// ekf_vio_amd/csrc/chol.hip — blocked Cholesky of the innovation covariance S and the
// right-hand triangular solves that turn Sigma*H^T into the Kalman gain.
//
// Reference: SimplicialLDLT(S^T) + solve, TightlyCoupledEKF.cpp:577-580
//   K = (S^-T (Sigma H^T)^T)^T  ==  (Sigma H^T) S^-1.
// Here: S = L L^T by a right-looking blocked factorisation with 64-wide blocks.  The
// diagonal block is factored inside one workgroup in LDS (potrf64_kernel, which also
// forms the inverse of the 64x64 triangular factor); panel solves and trailing updates
// are fp32 MFMA GEMMs against those inverses.  K = ((Sigma H^T) L^-T) L^-1 is two blocked
// substitutions whose block steps are GEMMs as well.
#include "common.h"

namespace {

#define PB 64
#define PLD 65  // LDS row stride: conflict-free for both row- and column-wise sweeps

__device__ inline float lane_bcast(float v, int src_lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}

// Factor the 64x64 block at S (lower triangle read) -> L (lower, upper zeroed) and
// Linv = L^-1 (lower, ld 64).  One workgroup of 256 threads (4 wavefronts).
//
// Factorisation: four 16-column micro-panels.  Wavefront 0 holds one matrix row per lane
// (16 panel entries in registers) and runs the 16 pivot steps with v_readlane broadcasts:
// scaling the pivot column over all 64 lanes IS the panel's triangular solve, so no
// barrier or LDS round trip sits on the pivot chain.  The 48x48 (then 32x32, 16x16)
// trailing update is done by all four wavefronts from LDS.
//
// Inverse: formed in fp64 from the fp32 factor (16x16 diagonal blocks by substitution, one
// per wavefront; off-diagonal blocks level by level) and rounded once to fp32.  An
// inverse that is accurate to rounding makes "multiply by L_kk^-1" as good as a
// backward-stable triangular solve (error eps*|Y||L||L^-1|); an inverse computed in fp32
// would lose another factor cond(L_kk).
__global__ __launch_bounds__(256) void potrf64_kernel(const float* __restrict__ S, int lds, float* __restrict__ L,
                                                      int ldl, float* __restrict__ Linv, int* info) {
    __shared__ float A[PB * PLD];    // A[r*PLD + c]
    __shared__ double Xd[PB * PLD];  // L^-1
    __shared__ double Td[3 * 256];
    __shared__ double Dinv[PB];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    for (int e = tid; e < PB * PB; e += 256) {
        int r = e % PB, c = e / PB;
        A[r * PLD + c] = (r >= c) ? S[(size_t)c * lds + r] : 0.f;
        Xd[r * PLD + c] = 0.0;
    }
    __syncthreads();
    bool bad = false;
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const int c0 = 16 * p;
        if (wave == 0) {
            float a[16];
#pragma unroll
            for (int j = 0; j < 16; j++) a[j] = A[lane * PLD + c0 + j];
#pragma unroll
            for (int k = 0; k < 16; k++) {
                float d = lane_bcast(a[k], c0 + k);
                if (!(d > 0.f)) {
                    bad = true;
                    d = 1e-20f;
                }
                const float lkk = sqrtf(d);
                const float inv = 1.0f / lkk;
                a[k] = (lane == c0 + k) ? lkk : a[k] * inv;
#pragma unroll
                for (int j = k + 1; j < 16; j++) {
                    const float sj = lane_bcast(a[k], c0 + j);
                    a[j] = __builtin_fmaf(-a[k], sj, a[j]);
                }
            }
            if (lane >= c0) {
#pragma unroll
                for (int j = 0; j < 16; j++) A[lane * PLD + c0 + j] = (lane - c0 >= j) ? a[j] : 0.f;
            }
        }
        __syncthreads();
        const int r0 = c0 + 16;
        const int rem = PB - r0;
        for (int e = tid; e < rem * rem; e += 256) {
            const int r = r0 + e % rem, c = r0 + e / rem;
            if (c <= r) {
                float acc = A[r * PLD + c];
#pragma unroll
                for (int j = 0; j < 16; j++) acc = __builtin_fmaf(-A[r * PLD + c0 + j], A[c * PLD + c0 + j], acc);
                A[r * PLD + c] = acc;
            }
        }
        __syncthreads();
    }
    // ---- inverse in fp64 ----
    if (tid < PB) Dinv[tid] = 1.0 / (double)A[tid * PLD + tid];
    __syncthreads();
    if (lane < 16) {
        const int o = 16 * wave;
        double x[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            double acc = (r == lane) ? 1.0 : 0.0;
#pragma unroll
            for (int q = 0; q < r; q++) acc -= (double)A[(o + r) * PLD + o + q] * x[q];
            x[r] = acc * Dinv[o + r];
        }
#pragma unroll
        for (int r = 0; r < 16; r++) Xd[(o + r) * PLD + o + lane] = x[r];
    }
    __syncthreads();
    for (int d = 1; d <= 3; d++) {
        const int nblk = 4 - d;
        for (int e = tid; e < 256 * nblk; e += 256) {
            const int b = e >> 8, s = e & 15, c = (e >> 4) & 15;
            const int j0 = 16 * b, i0 = 16 * (b + d);
            double acc = 0.0;
            for (int q = j0; q < i0; q++) acc += (double)A[(i0 + s) * PLD + q] * Xd[q * PLD + j0 + c];
            Td[b * 256 + c * 16 + s] = acc;
        }
        __syncthreads();
        for (int e = tid; e < 256 * nblk; e += 256) {
            const int b = e >> 8, s = e & 15, c = (e >> 4) & 15;
            const int j0 = 16 * b, i0 = 16 * (b + d);
            double acc = 0.0;
#pragma unroll
            for (int t = 0; t < 16; t++) acc += Xd[(i0 + s) * PLD + i0 + t] * Td[b * 256 + c * 16 + t];
            Xd[(i0 + s) * PLD + j0 + c] = -acc;
        }
        __syncthreads();
    }
    for (int e = tid; e < PB * PB; e += 256) {
        int r = e % PB, c = e / PB;
        L[(size_t)c * ldl + r] = A[r * PLD + c];
        Linv[c * PB + r] = (float)Xd[r * PLD + c];
    }
    if (bad && tid == 0) atomicOr(info, 1);
}

}  // namespace

void launch_cholesky(ekfvio_filter* f, float* S, float* L, float* Linv, int m_pad, int lds) {
    ProfScope ps(f, PC_CHOL, (double)m_pad * m_pad * m_pad / 3.0);
    const int mb = m_pad / PB;
    for (int k = 0; k < mb; k++) {
        float* Skk = S + (size_t)k * PB * lds + k * PB;
        float* Lkk = L + (size_t)k * PB * lds + k * PB;
        float* Li = Linv + (size_t)k * PB * PB;
        hipLaunchKernelGGL(potrf64_kernel, dim3(1), dim3(256), 0, f->stream, Skk, lds, Lkk, lds, Li, f->info);
        const int rem = m_pad - (k + 1) * PB;
        if (rem > 0) {
            float* Spanel = Skk + PB;  // rows below the diagonal block, same columns
            float* Lpanel = Lkk + PB;
            float* S22 = S + (size_t)(k + 1) * PB * lds + (k + 1) * PB;
            // L_ik = S_ik * Linv_kk^T
            launch_gemm(f->stream, 1, rem, PB, PB, 1.f, Spanel, lds, Li, PB, 0.f, nullptr, 0, Lpanel, lds, 0);
            // S_22 -= L_21 * L_21^T
            launch_gemm(f->stream, 1, rem, rem, PB, -1.f, Lpanel, lds, Lpanel, lds, 1.f, S22, lds, S22, lds, 0);
        }
    }
}

// X <- X * S^-1 with S = L L^T; X is nrows x m_pad (ld = ldx), W scratch of the same shape.
void launch_solve_right(ekfvio_filter* f, const float* L, const float* Linv, int m_pad, int lds, float* X, float* W,
                        int nrows, int ldx) {
    ProfScope ps(f, PC_SOLVE, 2.0 * nrows * (double)m_pad * m_pad);
    const int mb = m_pad / PB;
    // forward: Y L^T = X   ->  W holds Y
    for (int k = 0; k < mb; k++) {
        const float* Li = Linv + (size_t)k * PB * PB;
        float* Xk = X + (size_t)k * PB * ldx;
        float* Wk = W + (size_t)k * PB * ldx;
        launch_gemm(f->stream, 1, nrows, PB, PB, 1.f, Xk, ldx, Li, PB, 0.f, nullptr, 0, Wk, ldx, 0);
        const int rem = m_pad - (k + 1) * PB;
        if (rem > 0) {
            const float* Lpanel = L + (size_t)k * PB * lds + (k + 1) * PB;  // [rem x 64]
            float* Xr = X + (size_t)(k + 1) * PB * ldx;
            launch_gemm(f->stream, 1, nrows, rem, PB, -1.f, Wk, ldx, Lpanel, lds, 1.f, Xr, ldx, Xr, ldx, 0);
        }
    }
    // backward: K L = Y   ->  X holds K
    for (int k = mb - 1; k >= 0; k--) {
        const float* Li = Linv + (size_t)k * PB * PB;
        float* Xk = X + (size_t)k * PB * ldx;
        const float* Wk = W + (size_t)k * PB * ldx;
        launch_gemm(f->stream, 0, nrows, PB, PB, 1.f, Wk, ldx, Li, PB, 0.f, nullptr, 0, Xk, ldx, 0);
        if (k > 0) {
            const float* Lrow = L + (size_t)k * PB;  // block row k, columns 0..k*64: [64 x k*64]
            launch_gemm(f->stream, 0, nrows, k * PB, PB, -1.f, Xk, ldx, Lrow, lds, 1.f, W, ldx, W, ldx, 0);
        }
    }
}
