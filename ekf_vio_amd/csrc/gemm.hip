// ekf_vio_amd/csrc/gemm.hip — fp32 MFMA GEMM for gfx950.
//
// C[MxN] = beta*Cin + alpha * A[MxK] * op(B), everything column-major fp32 (the covariance
// and every gain/gather matrix of the filter are column-major with the state index
// contiguous).  This is the contraction behind the reference's covariance products:
//   Sigma = I_KH*Sigma*I_KH^T + K*R*K^T   (TightlyCoupledEKF.cpp:594-596)
//   Sigma = F*Sigma*F^T                   (TightlyCoupledEKF.cpp:113, dense predict mode)
// and the panel / trailing updates of the blocked Cholesky and triangular solves.
//
// Mapping to CDNA4: one 256-thread workgroup (4 wavefronts of 64) per 64x64 tile of C; each
// wavefront owns a 32x32 accumulator (16 VGPRs/lane) fed by v_mfma_f32_32x32x2_f32 (exact
// fp32 fmaf chain in k order, 64 cycles per issue per SIMD).  A and B^T tiles are staged
// [k][row] in LDS so that the MFMA operand read (lane -> consecutive rows, half-wave ->
// next k) is bank-conflict free; operands are passed swapped (D = B_frag x A_frag) so that
// the accumulator's lane index runs along C's contiguous dimension and every store is a
// 128-byte row segment.  Global->LDS staging is register-prefetched one K-tile ahead with
// a single barrier per K-tile (double-buffered LDS).
//
// Contract: K % 16 == 0; lda/ldb/ldc % 4 == 0 and 16-byte aligned bases; A has
// round_up(M,64) readable rows, B round_up(N,64) readable rows (transB) or columns; the
// K-padding of both operands is zero.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BM 64
#define BN 64
#define BK 16
#define LDS_A (BM)       // [BK][BM]
#define LDS_BT (BN)      // transB: [BK][BN], written with b128
#define LDS_BN (BN + 1)  // !transB: [BK][BN+1], written transposed with b32 (conflict-free)

template <bool TRANSB>
__global__ __launch_bounds__(256) void gemm_f32_mfma_kernel(int M, int N, int K, float alpha, const float* __restrict__ A,
                                                            int lda, const float* __restrict__ B, int ldb, float beta,
                                                            const float* Cin, int ldcin, float* C, int ldc, int flush) {
    __shared__ __attribute__((aligned(16))) float As[2][BK * LDS_A];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDS_BN];
    constexpr int LDB_S = TRANSB ? LDS_BT : LDS_BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave & 1;   // 32-row half of the tile
    const int wc = wave >> 1;  // 32-col half of the tile
    const int i0 = blockIdx.x * BM;
    const int j0 = blockIdx.y * BN;

    // staging coordinates
    const int a_k = tid >> 4;         // 0..15
    const int a_i = (tid & 15) * 4;   // 0..60
    const int b_j = tid >> 2;         // !transB: 0..63
    const int b_k = (tid & 3) * 4;    // !transB: 0,4,8,12

    const float* Ap = A + (size_t)a_k * lda + i0 + a_i;
    const float* Bp = TRANSB ? (B + (size_t)a_k * ldb + j0 + a_i) : (B + (size_t)(j0 + b_j) * ldb + b_k);

    float4 ra = *reinterpret_cast<const float4*>(Ap);
    float4 rb = *reinterpret_cast<const float4*>(Bp);

    auto stage = [&](int buf) {
        *reinterpret_cast<float4*>(&As[buf][a_k * LDS_A + a_i]) = ra;
        if (TRANSB) {
            *reinterpret_cast<float4*>(&Bs[buf][a_k * LDS_BT + a_i]) = rb;
        } else {
            Bs[buf][(b_k + 0) * LDS_BN + b_j] = rb.x;
            Bs[buf][(b_k + 1) * LDS_BN + b_j] = rb.y;
            Bs[buf][(b_k + 2) * LDS_BN + b_j] = rb.z;
            Bs[buf][(b_k + 3) * LDS_BN + b_j] = rb.w;
        }
    };
    stage(0);
    __syncthreads();

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;

    const int li = lane & 31;
    const int lk = lane >> 5;
    const int KT = K / BK;
    for (int kt = 0; kt < KT; kt++) {
        const int buf = kt & 1;
        if (kt + 1 < KT) {
            Ap += (size_t)BK * lda;
            Bp += TRANSB ? (size_t)BK * ldb : (size_t)BK;
            ra = *reinterpret_cast<const float4*>(Ap);
            rb = *reinterpret_cast<const float4*>(Bp);
        }
        const float* as = &As[buf][lk * LDS_A + wr * 32 + li];
        const float* bs = &Bs[buf][lk * LDB_S + wc * 32 + li];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a = as[kk * LDS_A];
            float b = bs[kk * LDB_S];
            // swapped operands: D[r = j][c = i] = sum_k B[j][k] * A[i][k]
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc, 0, 0, 0);
        }
        if (kt + 1 < KT) stage(buf ^ 1);
        __syncthreads();
    }

    // epilogue: lane -> row i (contiguous), register -> column j
    const int i = i0 + wr * 32 + li;
    if (i < M) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int j = j0 + wc * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (j < N) {
                float v = alpha * acc[r];
                if (beta != 0.f) v += beta * Cin[(size_t)j * ldcin + i];
                if (flush && !(fabsf(v) > EKF_FLUSH_THRESH)) v = 0.f;
                C[(size_t)j * ldc + i] = v;
            }
        }
    }
}

void launch_gemm(hipStream_t s, int transB, int M, int N, int K, float alpha, const float* A, int lda, const float* B,
                 int ldb, float beta, const float* Cin, int ldcin, float* C, int ldc, int flush) {
    if (M <= 0 || N <= 0 || K <= 0) return;
    dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN);
    if (transB)
        hipLaunchKernelGGL(gemm_f32_mfma_kernel<true>, grid, dim3(256), 0, s, M, N, K, alpha, A, lda, B, ldb, beta, Cin,
                           ldcin, C, ldc, flush);
    else
        hipLaunchKernelGGL(gemm_f32_mfma_kernel<false>, grid, dim3(256), 0, s, M, N, K, alpha, A, lda, B, ldb, beta,
                           Cin, ldcin, C, ldc, flush);
}
