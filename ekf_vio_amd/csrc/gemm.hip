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
// fp32 fmaf chain in k order, 64 cycles per issue per SIMD).  A and B^T tiles (64x32) are staged
// [k][row] in LDS so that the MFMA operand read (lane -> consecutive rows, half-wave ->
// next k) is bank-conflict free; operands are passed swapped (D = B_frag x A_frag) so that
// the accumulator's lane index runs along C's contiguous dimension and every store is a
// 128-byte row segment.  The main loop is a three-stage software pipeline (global ->
// registers -> LDS -> fragment registers -> MFMA), one barrier per 32-deep K-tile.
//
// Contract: K % 32 == 0 (and the lowerB start, a multiple of 64); lda/ldb/ldc % 4 == 0 and 16-byte aligned bases; A has
// round_up(M,64) readable rows, B round_up(N,64) readable rows (transB) or columns; the
// K-padding of both operands is zero.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BM 64
#define BN 64
#define BK 32
#define LDS_A (BM)       // [BK][BM]
#define LDS_BT (BN)      // transB: [BK][BN], written with b128
#define LDS_BN (BN + 1)  // !transB: [BK][BN+1], written transposed with b32 (conflict-free)

template <bool TRANSB>
__global__ __launch_bounds__(256) void gemm_f32_mfma_kernel(int M, int N, int K, float alpha, const float* __restrict__ A,
                                                            int lda, const float* __restrict__ B, int ldb, float beta,
                                                            const float* Cin, int ldcin, float* C, int ldc, int flush,
                                                            int lowerB) {
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

    // staging coordinates: A (and B when TRANSB): k = a_k and a_k+16, rows a_i..a_i+3
    const int a_k = tid >> 4;         // 0..15
    const int a_i = (tid & 15) * 4;   // 0..60
    // !TRANSB: column b_j, k = b_k..b_k+3 and b_k+16..b_k+19
    const int b_j = tid >> 2;         // 0..63
    const int b_k = (tid & 3) * 4;    // 0,4,8,12

    // lowerB: op(B)(k,j) is zero for k < j (B is L, or L^-T stored row-wise): start the
    // contraction at this tile's first column
    const int kbeg = lowerB ? min(j0, K) : 0;
    const int KT = (K - kbeg) / BK;
    const float* Ap = A + (size_t)(kbeg + a_k) * lda + i0 + a_i;
    const float* Bp = TRANSB ? (B + (size_t)(kbeg + a_k) * ldb + j0 + a_i) : (B + (size_t)(j0 + b_j) * ldb + kbeg + b_k);
    const size_t a_step = (size_t)BK * lda, b_step = TRANSB ? (size_t)BK * ldb : (size_t)BK;
    const size_t a_half = (size_t)16 * lda, b_half = TRANSB ? (size_t)16 * ldb : (size_t)16;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    const int li = lane & 31;
    const int lk = lane >> 5;

    // Software pipeline, one barrier per 32-deep K-tile.  In iteration t, between the 16
    // dependent MFMAs of tile t (each holds the wave's issue for 64 cycles) the wave also
    //   slots 0-7  : reads the operand fragments of tile t+1 from LDS buffer (t+1)%2,
    //   slots 8-11 : stages tile t+2 (already in registers) into LDS buffer t%2,
    //   slots 12-15: issues the global loads of tile t+3 into the same registers.
    // A wave alone on its SIMD (the filter's sizes put one workgroup on a CU) otherwise
    // executes all of that serially AFTER the MFMA block: measured 118 instead of 64 cycles
    // per MFMA.  __builtin_amdgcn_sched_barrier(0) pins the slot structure.
    // Tile u lives in LDS buffer u%2 and in fragment set u%2 (F even / G odd).
    float4 xa0, xa1, xb0, xb1;  // staging registers (one K-tile: 2 float4 of A, 2 of B)
    xa0 = xa1 = xb0 = xb1 = make_float4(0.f, 0.f, 0.f, 0.f);
    float Fa[BK / 2], Fb[BK / 2], Ga[BK / 2], Gb[BK / 2];
#pragma unroll
    for (int kk = 0; kk < BK / 2; kk++) Fa[kk] = Fb[kk] = Ga[kk] = Gb[kk] = 0.f;

#define GEMM_LOAD(kt)                                                      \
    do {                                                                   \
        const float* ap_ = Ap + (size_t)(kt) * a_step;                     \
        const float* bp_ = Bp + (size_t)(kt) * b_step;                     \
        xa0 = *reinterpret_cast<const float4*>(ap_);                       \
        xa1 = *reinterpret_cast<const float4*>(ap_ + a_half);              \
        xb0 = *reinterpret_cast<const float4*>(bp_);                       \
        xb1 = *reinterpret_cast<const float4*>(bp_ + b_half);              \
    } while (0)
#define GEMM_STAGE_A0(buf) *reinterpret_cast<float4*>(&As[buf][a_k * LDS_A + a_i]) = xa0
#define GEMM_STAGE_A1(buf) *reinterpret_cast<float4*>(&As[buf][(a_k + 16) * LDS_A + a_i]) = xa1
#define GEMM_STAGE_B0(buf)                                                                          \
    do {                                                                                            \
        if (TRANSB) {                                                                               \
            *reinterpret_cast<float4*>(&Bs[buf][a_k * LDS_BT + a_i]) = xb0;                         \
        } else {                                                                                    \
            float* b_ = &Bs[buf][b_k * LDS_BN + b_j];                                               \
            b_[0] = xb0.x; b_[LDS_BN] = xb0.y; b_[2 * LDS_BN] = xb0.z; b_[3 * LDS_BN] = xb0.w;      \
        }                                                                                           \
    } while (0)
#define GEMM_STAGE_B1(buf)                                                                          \
    do {                                                                                            \
        if (TRANSB) {                                                                               \
            *reinterpret_cast<float4*>(&Bs[buf][(a_k + 16) * LDS_BT + a_i]) = xb1;                  \
        } else {                                                                                    \
            float* b_ = &Bs[buf][(b_k + 16) * LDS_BN + b_j];                                        \
            b_[0] = xb1.x; b_[LDS_BN] = xb1.y; b_[2 * LDS_BN] = xb1.z; b_[3 * LDS_BN] = xb1.w;      \
        }                                                                                           \
    } while (0)
#define GEMM_STAGE(buf)    \
    do {                   \
        GEMM_STAGE_A0(buf); \
        GEMM_STAGE_A1(buf); \
        GEMM_STAGE_B0(buf); \
        GEMM_STAGE_B1(buf); \
    } while (0)
#define GEMM_FRAGS(S, buf)                                                         \
    do {                                                                           \
        const float* as_ = &As[buf][lk * LDS_A + wr * 32 + li];                    \
        const float* bs_ = &Bs[buf][lk * LDB_S + wc * 32 + li];                    \
        _Pragma("unroll") for (int kk = 0; kk < BK / 2; kk++) {                    \
            S##a[kk] = as_[2 * kk * LDS_A];                                        \
            S##b[kk] = bs_[2 * kk * LDB_S];                                        \
        }                                                                          \
    } while (0)
    // swapped operands: D[r = j][c = i] = sum_k B[j][k] * A[i][k]
#define GEMM_MFMA1(S, kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(S##b[kk], S##a[kk], acc, 0, 0, 0)
    // one K-tile: MFMAs from set C; fragments of the next tile -> set Nx from LDS[nb];
    // staging registers -> LDS[sb]; global loads of tile `lt`.  fr/stg/ld switch the stages on.
#define GEMM_TILE(C, Nx, nb, sb, fr, stg, ld, lt)                                              \
    do {                                                                                       \
        const float* as_ = &As[nb][lk * LDS_A + wr * 32 + li];                                 \
        const float* bs_ = &Bs[nb][lk * LDB_S + wc * 32 + li];                                 \
        const float* ap_ = Ap + (size_t)(lt) * a_step;                                         \
        const float* bp_ = Bp + (size_t)(lt) * b_step;                                         \
        _Pragma("unroll") for (int sl = 0; sl < 16; sl++) {                                    \
            GEMM_MFMA1(C, sl);                                                                 \
            if (sl < 8 && (fr)) {                                                              \
                Nx##a[2 * sl] = as_[(4 * sl) * LDS_A];                                         \
                Nx##a[2 * sl + 1] = as_[(4 * sl + 2) * LDS_A];                                 \
                Nx##b[2 * sl] = bs_[(4 * sl) * LDB_S];                                         \
                Nx##b[2 * sl + 1] = bs_[(4 * sl + 2) * LDB_S];                                 \
            }                                                                                  \
            if (stg) {                                                                         \
                if (sl == 8) GEMM_STAGE_A0(sb);                                                \
                if (sl == 9) GEMM_STAGE_A1(sb);                                                \
                if (sl == 10) GEMM_STAGE_B0(sb);                                               \
                if (sl == 11) GEMM_STAGE_B1(sb);                                               \
            }                                                                                  \
            if (ld) {                                                                          \
                if (sl == 12) xa0 = *reinterpret_cast<const float4*>(ap_);                     \
                if (sl == 13) xa1 = *reinterpret_cast<const float4*>(ap_ + a_half);            \
                if (sl == 14) xb0 = *reinterpret_cast<const float4*>(bp_);                     \
                if (sl == 15) xb1 = *reinterpret_cast<const float4*>(bp_ + b_half);            \
            }                                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                 \
        }                                                                                      \
    } while (0)

    // prologue: tile 0 staged and its fragments read; tile 1 staged; tile 2 in flight
    if (KT > 0) {
        GEMM_LOAD(0);
        GEMM_STAGE(0);
    }
    if (KT > 1) GEMM_LOAD(1);
    __syncthreads();
    if (KT > 0) GEMM_FRAGS(F, 0);
    if (KT > 1) GEMM_STAGE(1);
    if (KT > 2) GEMM_LOAD(2);
    __syncthreads();
    int kt = 0;
    for (; kt + 4 < KT; kt += 2) {  // steady state: every stage active for both tiles
        GEMM_TILE(F, G, 1, 0, true, true, true, kt + 3);
        __syncthreads();
        GEMM_TILE(G, F, 0, 1, true, true, true, kt + 4);
        __syncthreads();
    }
    for (; kt < KT; kt += 2) {  // tail: stages switch off as the tiles run out
        GEMM_TILE(F, G, 1, 0, kt + 1 < KT, kt + 2 < KT, kt + 3 < KT, kt + 3);
        __syncthreads();
        if (kt + 1 >= KT) break;
        GEMM_TILE(G, F, 0, 1, kt + 2 < KT, kt + 3 < KT, kt + 4 < KT, kt + 4);
        __syncthreads();
    }
#undef GEMM_LOAD
#undef GEMM_STAGE
#undef GEMM_STAGE_A0
#undef GEMM_STAGE_A1
#undef GEMM_STAGE_B0
#undef GEMM_STAGE_B1
#undef GEMM_FRAGS
#undef GEMM_MFMA1
#undef GEMM_TILE

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
                 int ldb, float beta, const float* Cin, int ldcin, float* C, int ldc, int flush, int lowerB) {
    if (M <= 0 || N <= 0 || K <= 0) return;
    dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN);
    if (transB)
        hipLaunchKernelGGL(gemm_f32_mfma_kernel<true>, grid, dim3(256), 0, s, M, N, K, alpha, A, lda, B, ldb, beta, Cin,
                           ldcin, C, ldc, flush, lowerB);
    else
        hipLaunchKernelGGL(gemm_f32_mfma_kernel<false>, grid, dim3(256), 0, s, M, N, K, alpha, A, lda, B, ldb, beta,
                           Cin, ldcin, C, ldc, flush, lowerB);
}

void launch_gemm_variant(hipStream_t s, int variant, int transB, int M, int N, int K, float alpha, const float* A, int lda,
                         const float* B, int ldb, float beta, const float* Cin, int ldcin, float* C, int ldc, int flush,
                         int lowerB) {
    (void)variant;
    launch_gemm(s, transB, M, N, K, alpha, A, lda, B, ldb, beta, Cin, ldcin, C, ldc, flush, lowerB);
}
