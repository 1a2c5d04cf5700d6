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

// Factor the 64x64 block at S (lower triangle read) -> L (lower, upper zeroed) and
// Linv = L^-1 (lower, ld 64).  One workgroup of 256 threads.
__global__ __launch_bounds__(256) void potrf64_kernel(const float* __restrict__ S, int lds, float* __restrict__ L,
                                                      int ldl, float* __restrict__ Linv, int* info) {
    __shared__ float A[PB * PLD];  // A[r*PLD + c]
    __shared__ float X[PB * PLD];
    const int tid = threadIdx.x;
    for (int e = tid; e < PB * PB; e += 256) {
        int r = e % PB, c = e / PB;
        A[r * PLD + c] = (r >= c) ? S[(size_t)c * lds + r] : 0.f;
        X[r * PLD + c] = 0.f;
    }
    bool bad = false;
    for (int k = 0; k < PB; k++) {
        __syncthreads();
        float d = A[k * PLD + k];
        if (!(d > 0.f)) {
            bad = true;
            d = 1e-20f;
        }
        const float lkk = sqrtf(d);
        __syncthreads();
        if (tid < PB && tid >= k) A[tid * PLD + k] = (tid == k) ? lkk : A[tid * PLD + k] / lkk;
        __syncthreads();
        // trailing update of the lower triangle: A[r][c] -= L[r][k]*L[c][k], k < c <= r
        const int rem = PB - 1 - k;
        for (int e = tid; e < rem * rem; e += 256) {
            int r = k + 1 + e % rem, c = k + 1 + e / rem;
            if (r >= c) A[r * PLD + c] = A[r * PLD + c] - A[r * PLD + k] * A[c * PLD + k];
        }
    }
    __syncthreads();
    // X = L^-1 by forward substitution, one column per thread
    if (tid < PB) {
        const int c = tid;
        X[c * PLD + c] = 1.f / A[c * PLD + c];
        for (int r = c + 1; r < PB; r++) {
            float acc = 0.f;
            for (int q = c; q < r; q++) acc = acc + A[r * PLD + q] * X[q * PLD + c];
            X[r * PLD + c] = -acc / A[r * PLD + r];
        }
    }
    __syncthreads();
    for (int e = tid; e < PB * PB; e += 256) {
        int r = e % PB, c = e / PB;
        L[(size_t)c * ldl + r] = A[r * PLD + c];
        Linv[c * PB + r] = X[r * PLD + c];
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
