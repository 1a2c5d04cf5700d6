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

#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
#include "motion_model.inc"
}  // namespace

// Diagnostic cycle stamps (s_memtime) of one workgroup's wave 0 go to GemmEpi::stamps (per handle; null in production).
#ifdef EKF_GEMM_STAMPS  // build with -DEKF_GEMM_STAMPS for scripts/gemm_stamps.py
#define GSTAMP(i)                                                                                  \
    do {                                                                                           \
        if (stamps_ && (i) < 40) stamps_[i] = (long long)__builtin_amdgcn_s_memtime();             \
    } while (0)
#else
#define GSTAMP(i) do { } while (0)
#endif

#define BM 64
#define BN 64
#define BK 32
#define LDS_A (BM)       // [BK][BM]
#define LDS_BT (BN)      // transB: [BK][BN], written with b128
#define LDS_BN (BN + 1)  // !transB: [BK][BN+1], written transposed with b32 (conflict-free)

// GROUPS = 2: 512 threads; wavefronts 4-7 run the same pipeline on the second half of the
// K-tiles out of their own LDS buffers (two waves per SIMD fill each other's issue bubbles),
// and the two accumulators are summed through LDS before the epilogue.
template <bool TRANSB, int GROUPS, int EPI>
__global__ __launch_bounds__(256 * GROUPS) void gemm_f32_mfma_kernel(int M, int N, int K, float alpha, const float* __restrict__ A,
                                                            int lda, const float* __restrict__ B, int ldb, float beta,
                                                            const float* Cin, int ldcin, float* C, int ldc, int flush,
                                                            int lowerB, GemmEpi epi) {
    __shared__ __attribute__((aligned(16))) float AsG[GROUPS][2][BK * LDS_A];
    __shared__ __attribute__((aligned(16))) float BsG[GROUPS][2][BK * LDS_BN];
    constexpr int LDB_S = TRANSB ? LDS_BT : LDS_BN;

#ifdef EKF_GEMM_STAMPS
    long long* stamps_ = (blockIdx.x == 1 && blockIdx.y == 1 && threadIdx.x == 0) ? epi.stamps : nullptr;
#endif
    GSTAMP(0);
    const int grp = (GROUPS > 1) ? (threadIdx.x >> 8) : 0;
    float (*As)[BK * LDS_A] = AsG[grp];
    float (*Bs)[BK * LDS_BN] = BsG[grp];
    const int tid = threadIdx.x & 255;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave & 1;   // 32-row half of the tile
    const int wc = wave >> 1;  // 32-col half of the tile
    // Tile order.  Workgroups are dealt round-robin to the eight XCDs by their linear index x + y * gridDim.x: with the tiles in that order every
    // XCD's L2 sees every operand panel of the product, and a Joseph GEMM at N = 1024 (49 x 49 tiles, K = 2048) pulls 1.5 GB per launch
    // through the fabric for 127 MB of operands (profiles/r05_pmc_traffic_n1024.json).  order2d (round 5): XCD x takes the x-th run of
    // ceil(T / 8) tiles of an order that walks strips of ceil(tiles_n / 8) tile columns row by row, so the ~96 tiles an XCD has in flight share
    // ~14 row panels and ~7 column panels.  A permutation of which workgroup forms which tile: the same bits.
    int bx = blockIdx.x, by = blockIdx.y;
    if (EPI == 2 && epi.sym) {
        // Round 6, the symmetric second Joseph GEMM: Sigma' = T + G K^T is a congruence of Sigma plus K R K^T whatever K is (:594-596), symmetric
        // up to the rounding of the two products, so only the tiles (bx >= by) of the LOWER triangle are formed -- a 1-D grid over them -- and
        // each writes its transpose as well (epilogue).  XCD x takes the x-th contiguous run of the tile order below.
        const int T = gridDim.x, L = (int)blockIdx.x;
        const int bq = T >> 3, br = T & 7, xcd = L & 7;
        int t = xcd * bq + min(xcd, br) + (L >> 3);
        // the order: strips of sym_w tile columns, each walked row by row from its diagonal down (row c0 + r of a strip has min(r + 1, w) tiles)
        const int tn = (N + BN - 1) / BN;
        int c0 = 0, w = 1;
        for (;;) {
            w = min(epi.sym_w, tn - c0);
            const int cnt = w * (w - 1) / 2 + w * (tn - c0 - (w - 1));
            if (t < cnt || c0 + w >= tn) break;
            t -= cnt;
            c0 += w;
        }
        const int head = w * (w - 1) / 2;
        if (t < head) {
            int r = 0;
            while (t >= r + 1) {
                t -= r + 1;
                r++;
            }
            bx = c0 + r;
            by = c0 + t;
        } else {
            t -= head;
            const int r = t / w;
            bx = c0 + (w - 1) + r;
            by = c0 + (t - r * w);
        }
    } else if (epi.order2d) {
        const int tm = gridDim.x, tn = gridDim.y, T = tm * tn;
        const int L = (int)blockIdx.x + (int)blockIdx.y * tm;
        const int bq = T >> 3, br = T & 7, xcd = L & 7;
        const int swz = xcd * bq + min(xcd, br) + (L >> 3);
        const int W = (tn + 7) >> 3, per = tm * W;
        const int st = swz / per, within = swz - st * per;
        const int wcs = min(W, tn - st * W);
        bx = within / wcs;
        by = st * W + within - bx * wcs;
    }
    const int i0 = bx * BM;
    const int j0 = by * BN;

    // staging coordinates: A (and B when TRANSB): k = a_k and a_k+16, rows a_i..a_i+3
    const int a_k = tid >> 4;         // 0..15
    const int a_i = (tid & 15) * 4;   // 0..60
    // !TRANSB: column b_j, k = b_k..b_k+3 and b_k+16..b_k+19
    const int b_j = tid >> 2;         // 0..63
    const int b_k = (tid & 3) * 4;    // 0,4,8,12

    // lowerB: op(B)(k,j) is zero for k < j (B is L, or L^-T stored row-wise): start the
    // contraction at this tile's first column
    const int kbeg0 = lowerB ? min(j0, K) : 0;
    const int KTall = (K - kbeg0) / BK;
    const int KTm = (KTall + GROUPS - 1) / GROUPS;                 // iterations every group runs (barrier count)
    const int KT = max(0, min(KTm, KTall - grp * KTm));           // K-tiles this group really has
    const int kbeg = kbeg0 + grp * KTm * BK;
    const float* Ap = A + (size_t)(kbeg + a_k) * lda + i0 + a_i;
    const float* Bp = TRANSB ? (B + (size_t)(kbeg + a_k) * ldb + j0 + a_i) : (B + (size_t)(j0 + b_j) * ldb + kbeg + b_k);
    const size_t a_step = (size_t)BK * lda, b_step = TRANSB ? (size_t)BK * ldb : (size_t)BK;
    const size_t a_half = (size_t)16 * lda, b_half = TRANSB ? (size_t)16 * ldb : (size_t)16;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    const int li = lane & 31;
    const int lk = lane >> 5;
    // an aborted persistent sweep in front (chol_persist.inc): requested here, looked at in front of the epilogue's writes
    int sweep_aborted = 0;
    if (EPI != 0 && epi.abort) sweep_aborted = *epi.abort;

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
#define GEMM_TILE(C, Nx, nb, sb, act, fr, stg, ld, lt)                                         \
    do {                                                                                       \
        const float* as_ = &As[nb][lk * LDS_A + wr * 32 + li];                                 \
        const float* bs_ = &Bs[nb][lk * LDB_S + wc * 32 + li];                                 \
        const float* ap_ = Ap + (size_t)(lt) * a_step;                                         \
        const float* bp_ = Bp + (size_t)(lt) * b_step;                                         \
        _Pragma("unroll") for (int sl = 0; sl < 16; sl++) {                                    \
            if (act) GEMM_MFMA1(C, sl);                                                        \
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

    // The C tile this wave will update is requested first: vmcnt retires in order, so these
    // loads ride in front of the prologue's own round trip instead of stalling the first
    // staging wait of the K loop.
    const int ci_ = i0 + wr * 32 + li;
    float cpre[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int j = j0 + wc * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        cpre[r] = (beta != 0.f && ci_ < M && j < N) ? Cin[(size_t)j * ldcin + ci_] : 0.f;
    }
    // prologue: tiles 0 and 1 are requested together (one memory round trip); tile 0 staged
    // and its fragments read; tile 1 staged; tile 2 in flight
    {
        float4 ya0, ya1, yb0, yb1;
        ya0 = ya1 = yb0 = yb1 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (KT > 0) GEMM_LOAD(0);
        if (KT > 1) {
            const float* ap_ = Ap + a_step;
            const float* bp_ = Bp + b_step;
            ya0 = *reinterpret_cast<const float4*>(ap_);
            ya1 = *reinterpret_cast<const float4*>(ap_ + a_half);
            yb0 = *reinterpret_cast<const float4*>(bp_);
            yb1 = *reinterpret_cast<const float4*>(bp_ + b_half);
        }
        if (KT > 0) GEMM_STAGE(0);
        __syncthreads();
        if (KT > 0) GEMM_FRAGS(F, 0);
        xa0 = ya0; xa1 = ya1; xb0 = yb0; xb1 = yb1;
        if (KT > 1) GEMM_STAGE(1);
        if (KT > 2) GEMM_LOAD(2);
        __syncthreads();
    }
    GSTAMP(1);
    // Every K-tile runs the same fully active body: beyond the last tile the prefetch index
    // is clamped (a redundant reload of the last tile) and what gets staged / read as
    // fragments is never consumed, so the loop needs no per-stage conditions.
    const int klast = KT > 0 ? KT - 1 : 0;
    const bool any = KT > 0;  // a group without tiles touches no memory
    int kt = 0;
    for (; kt + 1 < KTm; kt += 2) {
        GEMM_TILE(F, G, 1, 0, kt < KT, any, any, any, min(kt + 3, klast));
        __syncthreads();
        GSTAMP(2 + kt);
        GEMM_TILE(G, F, 0, 1, kt + 1 < KT, any, any, any, min(kt + 4, klast));
        __syncthreads();
        GSTAMP(3 + kt);
    }
    if (kt < KTm) {  // odd tile count
        GEMM_TILE(F, G, 1, 0, kt < KT, false, false, false, klast);
        __syncthreads();
    }
    GSTAMP(36);
    if (GROUPS > 1) {
        // sum the groups' accumulators through LDS (group 0's staging buffers are free now)
        float* red = &AsG[0][0][0];
        if (grp == 1) {
#pragma unroll
            for (int r = 0; r < 16; r++) red[r * 256 + tid] = acc[r];
        }
        __syncthreads();
        if (grp == 1) return;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] += red[r * 256 + tid];
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

    GSTAMP(37);
    if (EPI != 0 && sweep_aborted) return;  // (uniform over the grid) the state stays as process(dt) left it
    // epilogue: lane -> row i (contiguous), register -> column j.  All 16 outputs are formed
    // first; interior tiles then store without per-element tests.
    const int i = i0 + wr * 32 + li;
    float vout[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        float v = alpha * acc[r];
        v = v + beta * cpre[r];
        if (flush) v = (fabsf(v) > EKF_FLUSH_THRESH) ? v : 0.f;
        vout[r] = v;
    }
    if (EPI == 1) {
        // G(i,q) = (K R)(i,q) - T(i, idx[q]) for the measured columns of this tile (A is K).
        // All loads are unconditional on clamped indices (so they are issued as one batch);
        // only the store is predicated.
        const int ic = min(i, M - 1);
        int qv[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int j = j0 + wc * 32 + 4 * lk + (r & 3) + 8 * (r >> 2);
            qv[r] = epi.inv_idx[min(j, N - 1)];
        }
        float kq[16], kp[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int q = max(qv[r], 0);
            kq[r] = A[(size_t)q * lda + ic] * epi.Rm[2 * q];
            kp[r] = A[(size_t)(q ^ 1) * lda + ic] * epi.Rm[2 * q + 1];
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int j = j0 + wc * 32 + 4 * lk + (r & 3) + 8 * (r >> 2);
            const int q = qv[r];
            const float kr = ((q ^ 1) < q) ? (kp[r] + kq[r]) : (kq[r] + kp[r]);  // ascending measurement index
            if (q >= 0 && j < N && i < M) epi.G[(size_t)q * epi.ldg + i] = kr - vout[r];
        }
    }
    if ((EPI == 2 || EPI == 3) && bx == 0 && by == 0 && epi.n > 0) {
        // mu += K*y, quaternion renormalised.  EPI 2: K*y is column n of P, left there by the mode-1 GEMM; EPI 3 (Schur
        // flow): it arrives as per-column-block partial sums (joseph_g_kernel), added here in block order
        __shared__ float s_q[4];
        for (int e = threadIdx.x; e < epi.n; e += 256 * GROUPS) {
            float v;
            if (EPI == 3) {
                float ky = epi.Kyp[e];
                for (int cb = 1; cb < epi.kyp_blocks; cb++) ky = ky + epi.Kyp[(size_t)cb * epi.kyp_ld + e];
                v = epi.mu[e] + ky;
            } else {
                v = epi.mu[e] + epi.Pcol[e];
                epi.Pcol[e] = 0.f;
            }
            if (e >= 3 && e <= 6) s_q[e - 3] = v;
            else epi.mu[e] = v;
        }
        __syncthreads();
        if (threadIdx.x < 4) {
            const float qn = sqrtf(s_q[0] * s_q[0] + s_q[1] * s_q[1] + s_q[2] * s_q[2] + s_q[3] * s_q[3]);
            epi.mu[3 + threadIdx.x] = s_q[threadIdx.x] / qn;
        }
        if (threadIdx.x == 0 && epi.frame_counter) {
            const int fi = *epi.frame_counter + 1;
            *epi.frame_counter = (fi >= epi.frames) ? 0 : fi;
        }
        for (int e = threadIdx.x; e < epi.n_zero; e += 256 * GROUPS) epi.zero_words[e] = 0;  // the next update's sweep starts from zero flags
    }
    float* cp = C + (size_t)(j0 + wc * 32 + 4 * lk) * ldc + i;
    if (i0 + BM <= M && j0 + BN <= N) {
#pragma unroll
        for (int r = 0; r < 16; r++) cp[(size_t)((r & 3) + 8 * (r >> 2)) * ldc] = vout[r];
    } else if (i < M) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int jo = (r & 3) + 8 * (r >> 2);
            if (j0 + wc * 32 + 4 * lk + jo < N) cp[(size_t)jo * ldc] = vout[r];
        }
    }
    if (EPI == 2 && epi.sym && bx != by) {
        // the mirror tile Sigma'(j0.., i0..) = this tile's transpose, turned through LDS (B's staging buffers, 64 x 65 floats, free since the K
        // loop) so that it is written along ITS contiguous dimension.  (In place: nobody reads the upper tiles of T in this launch.)
        static_assert(!TRANSB || 2 * BK * LDS_BN >= 64 * 65, "turning tile");
        float* tr = &BsG[0][0][0];
        __syncthreads();
        const int il = wr * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; r++) tr[il * 65 + wc * 32 + 4 * lk + (r & 3) + 8 * (r >> 2)] = vout[r];
        __syncthreads();
        const int jm = j0 + il;  // this lane's row of the mirror tile
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int io = wc * 32 + 4 * lk + (r & 3) + 8 * (r >> 2);  // its column i0 + io
            if (jm < N && i0 + io < M) C[(size_t)(i0 + io) * ldc + jm] = tr[io * 65 + il];
        }
    }
    GSTAMP(38);
}


// ---------------------------------------------------------------------------------------
// Second tiling, for A * B^T at the filter's sizes, where a launch is ONE wave of workgroups and
// its duration is one workgroup's latency: BMt x 64 tiles with BMt in {32, 48, 64} chosen so that
// the tile count just fits the CUs (790 x 790: 221 tiles of 48 x 64 instead of 169 of 64 x 64,
// i.e. 3/4 of the MFMA work on the critical path).  Each wavefront owns 16 columns and all BMt
// rows: BMt/16 independent 16x16 accumulators on v_mfma_f32_16x16x4_f32 (32-cycle issue, 40-cycle
// dependent latency: two or three independent chains keep the pipe full).  K-tiles are 64 deep
// (half the barriers of the 32-deep kernel above), LDS tiles are [k][row] with a row stride of
// 16 (mod 32) floats, which makes the operand read (16 lanes -> consecutive rows, next 16 lanes
// -> next k) conflict-free, filled by one ds_write_b128 per global float4.  Operand fragments
// are read two k-steps ahead of their MFMAs; staging of tile t+1 and the global loads of tile
// t+2 are issued at the top of tile t.
// Workgroups are dealt round-robin to the 8 XCDs; the 1-D block index is remapped so that every
// XCD works on a contiguous run of tiles and re-reads the same operand panels from ITS L2.
// The end of the update that is nobody's tile (round 5: was done by the workgroup of tile (0,0) in front of its tile's store, which made that
// workgroup the launch's last by a memory round trip; now by one more workgroup behind the tiles' -- 221 tiles leave 35 compute units idle at
// N = 256): mu += K*y, quaternion renormalised (EPI 2: K*y is column n of P, left there by the mode-1 GEMM; EPI 3, the Schur flow: per-column-block
// partial sums of joseph_g_kernel, added in block order), the device frame counter, and the next update's sweep flags zeroed.  256 threads.
template <int EPI>
__device__ __forceinline__ void gemm16_finish_mean(const GemmEpi& epi) {
    __shared__ float s_q[4];
    // Everything this workgroup reads is requested at once -- the abort word of the persistent sweep in front, the partial sums of K y and the mean
    // for up to four elements per thread (n <= 1024 wherever a launch is a single wave of tiles) -- and the stores follow: written as a loop over
    // e with the store inside, the iterations' round trips go out one behind the other (mu may alias what is loaded), after the abort word's own,
    // and inside each the block columns' partial sums one by one: dependent round trips by the dozen made this workgroup end about when the tiles
    // do (round 6, profiles/r06_lin_pre_experiment.txt).
    int aborted = 0;
    if (epi.abort) aborted = *epi.abort;  // (an aborted persistent sweep in front: this update writes nothing)
    // (mean_keep: the launch's linearising workgroups form mu + K y for themselves from the OLD mean -- nobody may write it under them -- and the
    // propagation behind this launch replaces the state's mean anyway: launch_update)
    const bool keep = EPI == 3 && epi.mean_keep;
    float v4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; u++) {
        if (keep) break;
        const int e = min((int)threadIdx.x + 256 * u, epi.n - 1);
        if (EPI == 3) {
            // (the partial sums of up to 16 block columns as one batch from clamped addresses, added in block order afterwards: a loop with a
            // run-time bound is compiled load - wait - add, one round trip per block column)
            float pk[16];
#pragma unroll
            for (int cb = 0; cb < 16; cb++) pk[cb] = epi.Kyp[(size_t)min(cb, epi.kyp_blocks - 1) * epi.kyp_ld + e];
            const float m0 = epi.mu[e];
            float ky = pk[0];
#pragma unroll
            for (int cb = 1; cb < 16; cb++) ky = (cb < epi.kyp_blocks) ? ky + pk[cb] : ky;
            for (int cb = 16; cb < epi.kyp_blocks; cb++) ky = ky + epi.Kyp[(size_t)cb * epi.kyp_ld + e];
            v4[u] = m0 + ky;
        } else {
            v4[u] = epi.mu[e] + epi.Pcol[e];
        }
    }
    if (aborted) return;  // (uniform)
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int e = (int)threadIdx.x + 256 * u;
        if (e < epi.n && !keep) {
            if (EPI != 3) epi.Pcol[e] = 0.f;
            if (e >= 3 && e <= 6) s_q[e - 3] = v4[u];
            else epi.mu[e] = v4[u];
        }
    }
    for (int e = threadIdx.x + 1024; e < (keep ? 0 : epi.n); e += 256) {  // (not reached by the shapes this kernel is chosen for)
        float v;
        if (EPI == 3) {
            float ky = epi.Kyp[e];
            for (int cb = 1; cb < epi.kyp_blocks; cb++) ky = ky + epi.Kyp[(size_t)cb * epi.kyp_ld + e];
            v = epi.mu[e] + ky;
        } else {
            v = epi.mu[e] + epi.Pcol[e];
            epi.Pcol[e] = 0.f;
        }
        epi.mu[e] = v;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): s_q written
    __builtin_amdgcn_s_barrier();        // (the workgroup's other wavefronts have returned)
    if (threadIdx.x < 4 && !keep) {
        const float qn = sqrtf(s_q[0] * s_q[0] + s_q[1] * s_q[1] + s_q[2] * s_q[2] + s_q[3] * s_q[3]);
        epi.mu[3 + threadIdx.x] = s_q[threadIdx.x] / qn;
    }
    if (threadIdx.x == 0 && epi.frame_counter) {
        const int fi = *epi.frame_counter + 1;
        *epi.frame_counter = (fi >= epi.frames) ? 0 : fi;
    }
    for (int e = threadIdx.x; e < epi.n_zero; e += 256) epi.zero_words[e] = 0;  // the next update's sweep starts from zero flags
}

template <int BMt, int WPS, int EPI>
__global__ __launch_bounds__(256 * WPS) void gemm16_kernel(int M, int N, int K, float alpha, const float* __restrict__ A, int lda,
                                                     const float* __restrict__ B, int ldb, float beta, const float* Cin,
                                                     int ldcin, float* C, int ldc, int flush, int lowerB, GemmEpi epi,
                                                     int tiles_x, int tiles_y) {
    constexpr int RB = BMt / 16;                            // 16-row blocks per wavefront
    constexpr int SA = (BMt % 32 == 16) ? BMt : BMt + 16;   // LDS row strides, both = 16 mod 32
    constexpr int SB = 80;
    constexpr int BKK = 64;
    constexpr int NT = 256 * WPS;                           // threads; WPS = wavefronts per SIMD
    constexpr int AVT = BKK * BMt / 4;                      // float4 per K-tile of A
    constexpr int AV = (AVT + NT - 1) / NT;                 // ... per thread (the last one may be partial)
    constexpr int BV = BKK * 64 / 4 / NT;                   // of B: 4 (256 threads) or 2 (512)
    constexpr int BKS = NT / 16;                            // k-rows of B covered by one float4 per thread
    constexpr int NS = BKK / 4 / WPS;                       // k-steps per wavefront per K-tile
    __shared__ __attribute__((aligned(16))) float As[3][BKK * SA];
    __shared__ __attribute__((aligned(16))) float Bs[3][BKK * SB];

    // (EPI 2 / 3 with a mean to finish: one workgroup more than tiles)
    if ((EPI == 2 || EPI == 3) && (int)blockIdx.x == tiles_x * tiles_y) {
        if (threadIdx.x >= 256) return;
        if (epi.n > 0) gemm16_finish_mean<EPI>(epi);
        return;
    }
    if (EPI == 3 && (int)blockIdx.x > tiles_x * tiles_y) {
        // Round 6: the NEXT process(dt)'s linearisation and mean propagation (GemmEpi::lin_blocks), at the updated mean mu + K y with the quaternion
        // renormalised -- formed here for the 22 + 3 LIN_LM elements this workgroup reads exactly as gemm16_finish_mean forms it (the same sums in the
        // same order; one batch of loads) and never written back: the finishing workgroup leaves the mean alone in such a launch (mean_keep -- it
        // would be writing under these workgroups' reads), and the propagation behind the launch replaces the state's mean.  Behind an aborted sweep
        // the update is skipped: the linearisation is then taken at mu itself, as process(dt) would.
        if (threadIdx.x >= 256) return;
        const int tid = threadIdx.x;
        const int lb = (int)blockIdx.x - tiles_x * tiles_y - 1;
        LinLds& l = *reinterpret_cast<LinLds*>(&As[0][0]);
        static_assert(sizeof(LinLds) <= sizeof(As), "linearisation scratch");
        float* s_mu = &Bs[0][0];  // [22 + 3 LIN_LM]
        int aborted = 0;
        if (epi.abort) aborted = *epi.abort;
        constexpr int NE = EKF_BASE + 3 * LIN_LM;
        if (tid < NE) {
            const int e = (tid < EKF_BASE) ? tid : min(EKF_BASE + 3 * LIN_LM * lb + (tid - EKF_BASE), epi.n - 1);
            float pk[16];
#pragma unroll
            for (int cb = 0; cb < 16; cb++) pk[cb] = epi.Kyp[(size_t)min(cb, epi.kyp_blocks - 1) * epi.kyp_ld + e];
            const float m0 = epi.mu[e];
            float ky = pk[0];
#pragma unroll
            for (int cb = 1; cb < 16; cb++) ky = (cb < epi.kyp_blocks) ? ky + pk[cb] : ky;
            for (int cb = 16; cb < epi.kyp_blocks; cb++) ky = ky + epi.Kyp[(size_t)cb * epi.kyp_ld + e];
            s_mu[tid] = aborted ? m0 : m0 + ky;
        }
        __syncthreads();
        if (tid < 4 && !aborted) {  // (one wavefront: every lane has read the four components before any of them writes)
            const float qn = sqrtf(s_mu[3] * s_mu[3] + s_mu[4] * s_mu[4] + s_mu[5] * s_mu[5] + s_mu[6] * s_mu[6]);
            const float qv = s_mu[3 + tid] / qn;
            s_mu[3 + tid] = qv;
        }
        __syncthreads();
        const int e0 = EKF_BASE + 3 * LIN_LM * lb;
        auto mean = [s_mu, e0](int e) { return (e < EKF_BASE) ? s_mu[e] : s_mu[EKF_BASE + (e - e0)]; };
        linearize_block(mean, epi.lin_N, epi.lin_dt, epi.lin_FA, epi.lin_FB, epi.lin_FD, epi.lin_mu_next, lb, epi.lin_blocks - 1, l, tid);
        return;
    }
    // XCD-aware tile order (bijective for any grid size)
    const int nwg = tiles_x * tiles_y;
    const int bq = nwg >> 3, br = nwg & 7, xcd = blockIdx.x & 7;
    const int swz = xcd * bq + min(xcd, br) + (blockIdx.x >> 3);
    // The run goes down whole tile columns.  (A compact 2-D patch per XCD -- strips of a few tile columns walked row by row -- cuts the
    // fabric-side traffic of a Joseph GEMM at 790 x 790 x 512 from 22.3 to 16.8 MB per launch and the pair replayed back to back from
    // 12.3 to 11.9 us, but inside the filter step the first Joseph GEMM gets SLOWER (12.76 -> 13.17 us, step 120.3 -> 121.0 us:
    // profiles/r03_gemm_tile_order_experiment.txt): the operands were written by the previous kernel, every L2 is cold for them either
    // way, and the whole-column runs are what leaves every L2 holding all of K for the second GEMM.  Removed in round 5.)
    const int ti = swz % tiles_x, tj = swz / tiles_x;
    const int i0 = ti * BMt;
    const int j0 = tj * 64;

#ifdef EKF_GEMM_STAMPS
    long long* stamps_ = (blockIdx.x == 9 && threadIdx.x == 0) ? epi.stamps : nullptr;
#endif
    GSTAMP(0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = (tid >> 6) & 3;   // 16-column slice of the tile
    const int kg = tid >> 8;           // WPS = 2: wavefronts 4-7 take the odd k-steps of every K-tile
    const int li = lane & 15, g = lane >> 4;
    const int Mread = (M + 63) & ~63;  // readable rows of A (contract)

    const int kbeg = lowerB ? min(j0, K) : 0;
    const int KT = (K - kbeg) / BKK;

    // staging coordinates (scalars, not arrays: hipcc sends a loop-carried float4 array to scratch)
#define G16_ACOORD(u)                                                                       \
    const int ea##u = min(tid + NT * u, AVT - 1);                                           \
    const int ao##u = ((ea##u / (BMt / 4)) * lda + min(i0 + (ea##u % (BMt / 4)) * 4, Mread - 4)) * 4; /* bytes */ \
    const int al##u = (ea##u / (BMt / 4)) * SA + (ea##u % (BMt / 4)) * 4;
    G16_ACOORD(0) G16_ACOORD(1) G16_ACOORD(2) G16_ACOORD(3)
#undef G16_ACOORD
    // operands are read through buffer descriptors: 32-bit per-lane offsets fixed for the whole K loop,
    // one scalar offset per K-tile, and reads past column K return zero without touching memory
    // (the prefetches issued beyond the last tile need no clamp)
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, K * lda * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B), 0, K * ldb * 4, 0x00020000);
    const int bo = ((tid >> 4) * ldb + j0 + (tid & 15) * 4) * 4;  // bytes; k-rows tid>>4, + BKS, ...
    const int bl = (tid >> 4) * SB + (tid & 15) * 4;
    const int a_step = BKK * lda * 4, b_step = BKK * ldb * 4, b_rows = BKS * ldb * 4;  // bytes
    const int a_base = kbeg * lda * 4, b_base = kbeg * ldb * 4;
    static_assert((BV == 4 || BV == 2) && AV >= 1 && AV <= 4, "staging is written for 1..4 float4 of A and 2 or 4 of B per thread");

    f32x4 acc[RB];
#pragma unroll
    for (int a = 0; a < RB; a++) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    // an aborted persistent sweep in front (chol_persist.inc): requested here, looked at in front of the epilogue's writes
    int sweep_aborted = 0;
    if (EPI != 0 && epi.abort) sweep_aborted = *epi.abort;

    // the C tile this wave will update is requested first (see the 32x32 kernel)
    float cpre[RB][4];
#pragma unroll
    for (int a = 0; a < RB; a++)
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const int i = i0 + 16 * a + li, j = j0 + 16 * wave + 4 * g + v;
            cpre[a][v] = (beta != 0.f && i < M && j < N) ? Cin[(size_t)j * ldcin + i] : 0.f;
        }

    // EPI 1: the K R terms of G = K R - T[:, idx] depend only on the inputs: their two dependent memory
    // round trips (inv_idx, then K and R) are paid here, under the K loop, not in the epilogue
    int qv[4] = {-1, -1, -1, -1};
    float kq[RB][4], kp[RB][4];
    if (EPI == 1 && kg == 0) {  // the epilogue belongs to wavefronts 0-3
        const int jbq = j0 + 16 * wave + 4 * g;
#pragma unroll
        for (int v = 0; v < 4; v++) qv[v] = epi.inv_idx[min(jbq + v, N - 1)];
#pragma unroll
        for (int a = 0; a < RB; a++) {
            const int ic = min(i0 + 16 * a + li, M - 1);
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const int q = max(qv[v], 0);
                kq[a][v] = A[(size_t)q * lda + ic] * epi.Rm[2 * q];
                kp[a][v] = A[(size_t)(q ^ 1) * lda + ic] * epi.Rm[2 * q + 1];
            }
        }
    }
    // Three staging register sets, X, Y and Z, rotate between iterations (tile t+1 sits in one, tile t+2 is in
    // flight into the second, tile t+3 is requested into the third).  With a single set hipcc loads into fresh registers and
    // copies them back at the loop edge, which parks a full memory round trip behind every K-tile.
    float4 Xa0, Xa1, Xa2, Xa3, Xb0, Xb1, Xb2, Xb3, Ya0, Ya1, Ya2, Ya3, Yb0, Yb1, Yb2, Yb3;
    float4 Za0, Za1, Za2, Za3, Zb0, Zb1, Zb2, Zb3;
    Xa0 = Xa1 = Xa2 = Xa3 = Xb0 = Xb1 = Xb2 = Xb3 = make_float4(0.f, 0.f, 0.f, 0.f);
    Ya0 = Ya1 = Ya2 = Ya3 = Yb0 = Yb1 = Yb2 = Yb3 = make_float4(0.f, 0.f, 0.f, 0.f);
    Za0 = Za1 = Za2 = Za3 = Zb0 = Zb1 = Zb2 = Zb3 = make_float4(0.f, 0.f, 0.f, 0.f);
#define G16_LOAD(S, kt)                                                                                \
    do {                                                                                               \
        const int sa_ = a_base + (kt) * a_step, sb_ = b_base + (kt) * b_step;                          \
        S##a0 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsA, ao0, sa_, 0));                                                               \
        if constexpr (AV > 1) S##a1 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsA, ao1, sa_, 0));                                         \
        if constexpr (AV > 2) S##a2 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsA, ao2, sa_, 0));                                         \
        if constexpr (AV > 3) S##a3 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsA, ao3, sa_, 0));                                         \
        S##b0 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsB, bo, sb_, 0));                                                                \
        S##b1 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsB, bo + b_rows, sb_, 0));                                                       \
        if constexpr (BV > 2) S##b2 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsB, bo + 2 * b_rows, sb_, 0));                             \
        if constexpr (BV > 2) S##b3 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsB, bo + 3 * b_rows, sb_, 0));                             \
    } while (0)
// item j of a K-tile's AV + BV float4 per thread: first the A pieces, then the B pieces
#define G16_LOADI(S, j, kt)                                                                            \
    do {                                                                                               \
        const int sa_ = a_base + (kt) * a_step, sb_ = b_base + (kt) * b_step;                          \
        if ((j) == 0) S##a0 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsA, ao0, sa_, 0));                                                 \
        if ((j) == 1 && AV > 1) S##a1 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsA, ao1, sa_, 0));                                       \
        if ((j) == 2 && AV > 2) S##a2 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsA, ao2, sa_, 0));                                       \
        if ((j) == 3 && AV > 3) S##a3 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsA, ao3, sa_, 0));                                       \
        if ((j) == AV) S##b0 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsB, bo, sb_, 0));                                                 \
        if ((j) == AV + 1) S##b1 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsB, bo + b_rows, sb_, 0));                                    \
        if ((j) == AV + 2 && BV > 2) S##b2 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsB, bo + 2 * b_rows, sb_, 0));                      \
        if ((j) == AV + 3 && BV > 2) S##b3 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsB, bo + 3 * b_rows, sb_, 0));                      \
    } while (0)
#define G16_STAGEI(S, j, buf)                                                                          \
    do {                                                                                               \
        if ((j) == 0) *reinterpret_cast<float4*>(&As[buf][al0]) = S##a0;                               \
        if ((j) == 1 && AV > 1) *reinterpret_cast<float4*>(&As[buf][al1]) = S##a1;                     \
        if ((j) == 2 && AV > 2) *reinterpret_cast<float4*>(&As[buf][al2]) = S##a2;                     \
        if ((j) == 3 && AV > 3) *reinterpret_cast<float4*>(&As[buf][al3]) = S##a3;                     \
        if ((j) == AV) *reinterpret_cast<float4*>(&Bs[buf][bl]) = S##b0;                               \
        if ((j) == AV + 1) *reinterpret_cast<float4*>(&Bs[buf][bl + BKS * SB]) = S##b1;                \
        if ((j) == AV + 2 && BV > 2) *reinterpret_cast<float4*>(&Bs[buf][bl + 2 * BKS * SB]) = S##b2;  \
        if ((j) == AV + 3 && BV > 2) *reinterpret_cast<float4*>(&Bs[buf][bl + 3 * BKS * SB]) = S##b3;  \
    } while (0)
#define G16_STAGE_A(S, buf)                                                                            \
    do {                                                                                               \
        *reinterpret_cast<float4*>(&As[buf][al0]) = S##a0; /* a clamped duplicate rewrites the same bytes */ \
        if constexpr (AV > 1) *reinterpret_cast<float4*>(&As[buf][al1]) = S##a1;                       \
        if constexpr (AV > 2) *reinterpret_cast<float4*>(&As[buf][al2]) = S##a2;                       \
        if constexpr (AV > 3) *reinterpret_cast<float4*>(&As[buf][al3]) = S##a3;                       \
    } while (0)
#define G16_STAGE_B(S, buf)                                                                            \
    do {                                                                                               \
        *reinterpret_cast<float4*>(&Bs[buf][bl]) = S##b0;                                              \
        *reinterpret_cast<float4*>(&Bs[buf][bl + BKS * SB]) = S##b1;                                   \
        if constexpr (BV > 2) *reinterpret_cast<float4*>(&Bs[buf][bl + 2 * BKS * SB]) = S##b2;         \
        if constexpr (BV > 2) *reinterpret_cast<float4*>(&Bs[buf][bl + 3 * BKS * SB]) = S##b3;         \
    } while (0)
#define G16_MFMA(c, b, a) c = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, c, 0, 0, 0)
// compile-time switches for pricing the parts of the K loop (scripts/gemm_loop_pricing.sh); the results are wrong
// with any of them set: -DG16X_NOLOAD=1 -DG16X_NOSTAGE=1 -DG16X_NOBAR=1 -DG16X_NOFR=1
#ifndef G16X_NOLOAD
#define G16X_NOLOAD 0
#endif
#ifndef G16X_NOSTAGE
#define G16X_NOSTAGE 0
#endif
#ifndef G16X_NOBAR
#define G16X_NOBAR 0
#endif
#ifndef G16X_NOFR
#define G16X_NOFR 0
#endif
#define G16_FR(P, st, slot)                                                               \
    do {                                                                                  \
        if (G16X_NOFR) break;                                                             \
        fb[slot] = P##b[(4 * WPS * (st)) * SB];                                           \
        _Pragma("unroll") for (int a = 0; a < RB; a++) fa[slot][a] = P##a[(4 * WPS * (st)) * SA + 16 * a]; \
    } while (0)
    // One K-tile t out of LDS buffer t % 3.  Per wavefront NS k-steps (every WPS-th of the tile); a
    // step's operand fragments are read two steps ahead, for steps 0 and 1 already during the
    // previous tile.  Around the MFMAs of
    //   steps 0..BAR-1  one global load of tile t+3 (register set S2) and one LDS write of tile t+1
    //               (register set S1, loaded two tiles ago, to buffer (t+1) % 3) per step,
    //   step BAR    the one barrier of the tile: tile t+1 is visible from here on, and every wavefront
    //               is done with buffer (t-1) % 3, the target of the NEXT tile's staging,
    //   steps NS-2, NS-1  the first two fragment sets of tile t+1 are fetched,
    // so no LDS or memory latency is left between the last MFMA of one tile and the first of the next.
#define G16_TILE(S1, S2, tcur)                                                            \
    do {                                                                                  \
        const int nxt = (cb == 2) ? 0 : cb + 1;                                           \
        na = &As[nxt][(g + 4 * kg) * SA + li];                                            \
        nb = &Bs[nxt][(g + 4 * kg) * SB + 16 * wave + li];                                \
        _Pragma("unroll") for (int st = 0; st < NS; st++) {                               \
            if (st + 2 < NS) G16_FR(p, st + 2, (st + 2) % 4);                             \
            _Pragma("unroll") for (int a = 0; a < RB; a++)                                \
                G16_MFMA(acc[a], fb[st % 4], fa[st % 4][a]);                              \
            /* past the last tile the prefetch index is clamped (a redundant reload) and what is  \
               staged is never read: a branch here would split the pinned instruction stream */   \
            /* one global load and one LDS staging write per step (a burst of them stalls the wave   \
               behind the CU's 64 B/clk vector-memory path and the MFMAs queue up behind it);         \
               past the last tile the loads fall outside the buffer (zeros, no traffic) and what is   \
               staged is never read: a branch here would split the pinned instruction stream */       \
            if (!G16X_NOLOAD) G16_LOADI(S2, st, (tcur) + 3);                              \
            if (!G16X_NOSTAGE) G16_STAGEI(S1, st, nxt);                                   \
            if (st == BAR && !G16X_NOBAR) __syncthreads();                                \
            if (st == NS - 2) G16_FR(n, 0, 0);                                            \
            if (st == NS - 1) G16_FR(n, 1, 1);                                            \
            __builtin_amdgcn_sched_barrier(0);                                            \
        }                                                                                 \
        cb = nxt;                                                                         \
        pa = na;                                                                          \
        pb = nb;                                                                          \
    } while (0)

    if (KT > 0) {
        constexpr int BAR = AV + BV;  // the step after the last staging write
        static_assert(BAR <= NS - 3 && NS % 4 == 0, "step schedule");
        // tiles 0, 1 and 2 are requested together (one memory round trip); from then on tile t+3 is requested while
        // tile t is computed and written to LDS during tile t+2: two tile times of memory latency are covered, which
        // a full grid needs (every XCD's first touch of an operand panel comes from the fabric, not from its L2)
        G16_LOAD(Z, 0);
        G16_LOAD(X, 1);
        G16_LOAD(Y, 2);
        G16_STAGE_A(Z, 0);
        G16_STAGE_B(Z, 0);
        __syncthreads();
        float fa[4][RB], fb[4];
        int cb = 0;
        const float* pa = &As[0][(g + 4 * kg) * SA + li];
        const float* pb = &Bs[0][(g + 4 * kg) * SB + 16 * wave + li];
        const float* na = pa;
        const float* nb = pb;
        G16_FR(n, 0, 0);
        G16_FR(n, 1, 1);
        GSTAMP(1);
        int t = 0;
        // register sets rotate: (stage, in flight, load) = (X, Y, Z) -> (Y, Z, X) -> (Z, X, Y)
        for (; t + 2 < KT; t += 3) {
            G16_TILE(X, Z, t);
            GSTAMP(2 + t);
            G16_TILE(Y, X, t + 1);
            GSTAMP(3 + t);
            G16_TILE(Z, Y, t + 2);
            GSTAMP(4 + t);
        }
        if (t < KT) {
            G16_TILE(X, Z, t);
            t++;
        }
        if (t < KT) G16_TILE(Y, X, t);
        GSTAMP(36);
        __syncthreads();  // every wavefront is done with LDS (the reduction below reuses it)
    }
#undef G16_TILE
#undef G16_FR
#undef G16_LOAD
#undef G16_LOADI
#undef G16_STAGEI
#undef G16_STAGE_A
#undef G16_STAGE_B
    if constexpr (WPS > 1) {
        // sum the two k-interleaved partial accumulators through LDS (the staging buffers are free)
        float* red = &As[0][0];
        static_assert(3 * BKK * SA >= RB * 4 * 256, "reduction scratch");
        if (kg == 1) {
#pragma unroll
            for (int a = 0; a < RB; a++)
#pragma unroll
                for (int v = 0; v < 4; v++) red[(a * 4 + v) * 256 + (tid & 255)] = acc[a][v];
        }
        __syncthreads();
        if (kg == 1) return;
#pragma unroll
        for (int a = 0; a < RB; a++)
#pragma unroll
            for (int v = 0; v < 4; v++) acc[a][v] += red[(a * 4 + v) * 256 + tid];
    }

    GSTAMP(37);
    if (EPI != 0 && sweep_aborted) return;  // (uniform over the grid) the state stays as process(dt) left it
    // epilogue: lane -> row (contiguous in memory), register -> column
    float vout[RB][4];
#pragma unroll
    for (int a = 0; a < RB; a++)
#pragma unroll
        for (int v = 0; v < 4; v++) {
            float x = alpha * acc[a][v];
            x = x + beta * cpre[a][v];
            if (flush) x = (fabsf(x) > EKF_FLUSH_THRESH) ? x : 0.f;
            vout[a][v] = x;
        }
    const int jb = j0 + 16 * wave + 4 * g;
    if (EPI == 1) {
        // G(i,q) = (K R)(i,q) - T(i, idx[q]) for the measured columns of this tile (A is K)
#pragma unroll
        for (int a = 0; a < RB; a++) {
            const int i = i0 + 16 * a + li;
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const int q = qv[v];
                const float kr = ((q ^ 1) < q) ? (kp[a][v] + kq[a][v]) : (kq[a][v] + kp[a][v]);  // ascending measurement index
                if (q >= 0 && jb + v < N && i < M) epi.G[(size_t)q * epi.ldg + i] = kr - vout[a][v];
            }
        }
    }
    float* cp = C + (size_t)jb * ldc + i0 + li;
    if (i0 + BMt <= M && j0 + 64 <= N) {  // interior tile: no per-element tests
#pragma unroll
        for (int a = 0; a < RB; a++)
#pragma unroll
            for (int v = 0; v < 4; v++) cp[(size_t)v * ldc + 16 * a] = vout[a][v];
    } else {
#pragma unroll
        for (int a = 0; a < RB; a++) {
            if (i0 + 16 * a + li < M) {
#pragma unroll
                for (int v = 0; v < 4; v++)
                    if (jb + v < N) cp[(size_t)v * ldc + 16 * a] = vout[a][v];
            }
        }
    }
    GSTAMP(38);
}

// true where launch_gemm forms A * B^T with the 64 x 64 kernel (more 64-row tiles than compute units): the regime in which GemmEpi::sym is heeded
bool gemm_throughput_regime(const ekfvio_filter* f, int M, int N, int K) {
    const int cus = f->num_cus > 0 ? f->num_cus : 256;
    return !(K % 64 == 0 && ((M + 63) / 64) * ((N + 63) / 64) <= cus);
}

bool gemm_single_round_with(const ekfvio_filter* f, int M, int N, int K, int extra) {
    if (K % 64 != 0) return false;
    const int cus = f->num_cus > 0 ? f->num_cus : 256;
    const int ty = (N + 63) / 64;
    for (int bm : {32, 48, 64})
        if (((M + bm - 1) / bm) * ty <= cus) return ((M + bm - 1) / bm) * ty + extra <= cus;
    return false;
}

// cfg: 0 = choose by shape; 1 / 2 = the 64x64 kernel with 256 / 512 threads; 32, 48, 64 = gemm16_kernel with that BM
// (512 threads); +100 = the same with 256 threads
static void launch_gemm_cfg(ekfvio_filter* f, int cfg, int transB, int M, int N, int K, float alpha, const float* A, int lda,
                            const float* B, int ldb, float beta, const float* Cin, int ldcin, float* C, int ldc, int flush,
                            int lowerB, const GemmEpi* epi) {
    if (M <= 0 || N <= 0 || K <= 0) return;
    GemmEpi e;
    if (epi) e = *epi;
    hipStream_t s = f->stream;
    e.stamps = f->gemm_stamps;
    const int cus = f->num_cus > 0 ? f->num_cus : 256;  // per handle: handles on different devices may differ
    const int ty = (N + 63) / 64;
    auto tiles = [&](int bm) { return ((M + bm - 1) / bm) * ty; };
    if (cfg == 0) {
        cfg = 1;
        if (transB && K % 64 == 0) {
            // latency regime: the smallest tile height whose grid is still a single wave of workgroups
            for (int bm : {32, 48, 64}) {
                if (tiles(bm) <= cus) {
                    cfg = bm;
                    break;
                }
            }
        }
    }
    if (cfg >= 32) {
        const int wps = cfg >= 100 ? 1 : 2;  // 132 / 148 / 164: one wavefront per SIMD (micro-benchmark only)
        const int bm = cfg % 100;
        const int tx = (M + bm - 1) / bm;
        if (!(e.mode == 3 && e.n > 0) || tiles(bm) + 1 + e.lin_blocks > cus) e.lin_blocks = 0;  // (launch_update asked gemm_single_round_with first)
        e.mean_keep = e.lin_blocks > 0 ? 1 : 0;
        dim3 grid(tiles(bm) + ((e.mode == 2 || e.mode == 3) && e.n > 0 ? 1 : 0) + e.lin_blocks);  // (+1: gemm16_finish_mean; + the next step's linearisation)
#define GEMM16_GO(BMv, W, EP)                                                                                           \
    hipLaunchKernelGGL((gemm16_kernel<BMv, W, EP>), grid, dim3(256 * W), 0, s, M, N, K, alpha, A, lda, B, ldb, beta, Cin, ldcin, C, \
                       ldc, flush, lowerB, e, tx, ty)
#define GEMM16_BM(W, EP)                          \
    do {                                          \
        if (bm == 32) GEMM16_GO(32, W, EP);       \
        else if (bm == 48) GEMM16_GO(48, W, EP);  \
        else GEMM16_GO(64, W, EP);                \
    } while (0)
        if (e.mode == 1) GEMM16_BM(2, 1);
        else if (e.mode == 2) GEMM16_BM(2, 2);
        else if (e.mode == 3) GEMM16_BM(2, 3);
        else if (wps == 2) GEMM16_BM(2, 0);
        else GEMM16_BM(1, 0);
#undef GEMM16_BM
#undef GEMM16_GO
        return;
    }
    const int groups = cfg == 2 ? 2 : 1;
    e.lin_blocks = 0, e.mean_keep = 0;  // (gemm16_kernel only)
    dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN);
    e.sym = (e.sym && e.mode == 2 && M == N && transB) ? 1 : 0;
    if (e.sym) {
        // strips of ceil(tn / 8) tile columns, as order2d's: measured at tn = 49 (FETCH_SIZE x 2 per launch, scripts/sym_w_sweep.sh) row by row 572 MB,
        // strips of 12 / 7 / 4 / 3 columns 296 / 258 / 277 / 312 MB, the launch's duration the same within 0.5 % (it is MFMA-bound)
        e.sym_w = ((int)grid.x + 7) >> 3;
        grid = dim3(grid.x * (grid.x + 1) / 2);  // the lower triangle's tiles
    }
    // throughput regime (several tiles per compute unit and a full contraction per tile): compact 2-D patches per XCD.  Not for the
    // triangular-aware gain GEMM: its tiles' work falls with the tile column, and strips of columns would load the XCDs unevenly.
    e.order2d = (!e.sym && f->gemm_order2d && !lowerB && (int)(grid.x * grid.y) >= 2 * cus) ? 1 : 0;
#define GEMM_GO(TB, G, EP)                                                                                             \
    hipLaunchKernelGGL((gemm_f32_mfma_kernel<TB, G, EP>), grid, dim3(256 * G), 0, s, M, N, K, alpha, A, lda, B, ldb, beta, \
                       Cin, ldcin, C, ldc, flush, lowerB, e)
    if (e.mode == 1) {
        GEMM_GO(true, 1, 1);  // the Joseph GEMMs are always A * B^T
    } else if (e.mode == 2) {
        GEMM_GO(true, 1, 2);
    } else if (e.mode == 3) {
        GEMM_GO(true, 1, 3);
    } else if (groups == 2) {
        if (transB) GEMM_GO(true, 2, 0);
        else GEMM_GO(false, 2, 0);
    } else {
        if (transB) GEMM_GO(true, 1, 0);
        else GEMM_GO(false, 1, 0);
    }
#undef GEMM_GO
}

void launch_gemm(ekfvio_filter* f, int transB, int M, int N, int K, float alpha, const float* A, int lda, const float* B,
                 int ldb, float beta, const float* Cin, int ldcin, float* C, int ldc, int flush, int lowerB,
                 const GemmEpi* epi) {
    launch_gemm_cfg(f, 0, transB, M, N, K, alpha, A, lda, B, ldb, beta, Cin, ldcin, C, ldc, flush, lowerB, epi);
}

// variant: 0 = production choice, 1 = 64x64 tiles / 256 threads, 2 = 64x64 / 512 threads, 32 / 48 / 64 = BM of gemm16_kernel
void launch_gemm_variant(ekfvio_filter* f, int variant, int transB, int M, int N, int K, float alpha, const float* A, int lda,
                         const float* B, int ldb, float beta, const float* Cin, int ldcin, float* C, int ldc, int flush,
                         int lowerB) {
    if (variant >= 32 && (!transB || K % 64 != 0)) variant = 1;
    if (variant >= 32 && variant % 100 != 32 && variant % 100 != 48 && variant % 100 != 64) variant = 0;
    launch_gemm_cfg(f, variant, transB, M, N, K, alpha, A, lda, B, ldb, beta, Cin, ldcin, C, ldc, flush, lowerB, nullptr);
}

