// ekf_vio_amd/csrc/chol.hip — blocked Cholesky of the innovation covariance and the
// right-hand triangular sweeps that turn Sigma*H^T into the Kalman gain.
//
// Reference: SimplicialLDLT(S^T) + solve, TightlyCoupledEKF.cpp:577-580
//   K = (S^-T (Sigma H^T)^T)^T  ==  (Sigma H^T) A^-1,  A = sym(lower(S^T)).
// Here A = L L^T by a right-looking blocked factorisation with 64-wide blocks, one launch
// per block step:
//   chol_step_kernel(k): every workgroup owns one 64x64 tile (i,j), i >= j > k, of the
//     trailing matrix: it forms the panel blocks L_ik = A_ik L_kk^-T and L_jk, applies
//     A_ij -= L_ik L_jk^T, and the workgroup that owns the next diagonal tile (k+1,k+1)
//     immediately factors it in LDS (look-ahead), so the sequential chain is one launch per
//     block step.
//   The sweep runs over the augmented matrix [A; Sigma H^T; I]: the extra row blocks are
//     just more tiles of the same launches (the forward substitution Y L^T = X IS the
//     panel/trailing step), so it also delivers Y = Sigma H^T L^-T and L^-T.  The gain is
//     then K = Y L^-1 as MFMA GEMMs (gemm.hip) with one residual-refinement step.
//
// Triangular solves against a 64x64 diagonal block are micro-blocked substitutions with
// 16-wide blocks: each wavefront owns 16 rows of the right-hand side in LDS and alternates
// "multiply by the inverse of a 16x16 diagonal block" and "eliminate from the remaining
// columns", both on v_mfma_f32_16x16x4_f32, with no workgroup barrier (rows never cross
// wavefronts).  Only 16x16 inverses are ever formed, in fp64 from the fp32 factor and
// rounded once (an inverse that is exact to rounding keeps the multiply as accurate as a
// backward-stable solve; formed in fp32 it costs a factor cond(L_kk): measured 18x larger
// state error on the first, ill-conditioned update).
// The 64x64x64 trailing products run on v_mfma_f32_32x32x2_f32.  LDS tiles are column-major
// with a row stride of 65 floats: conflict-free for both operand orientations.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define PB 64
#define PLD 65  // LDS tile stride: element (r,c) of a tile lives at c*PLD + r  (column-major)
#define ILD 17  // stride of a 16x16 inverse block in LDS
#define INV_LDS (4 * 16 * ILD)

__device__ __forceinline__ float lane_bcast(float v, int src_lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}
__device__ __forceinline__ double lane_bcast_d(double v, int src_lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}

// ---- tile movers (256 threads) ---------------------------------------------------------
__device__ __forceinline__ void load_tile(float* T, const float* __restrict__ G, int ld, int tid) {
    // 64x64 column-major global tile -> LDS (c*PLD + r); 16-byte global loads along r
#pragma unroll
    for (int it = 0; it < 4; it++) {
        const int e = tid + it * 256;
        const int c = e >> 4, r4 = (e & 15) * 4;
        const float4 v = *reinterpret_cast<const float4*>(G + (size_t)c * ld + r4);
        float* t = T + c * PLD + r4;
        t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
    }
}
__device__ __forceinline__ void store_tile(const float* T, float* __restrict__ G, int ld, int tid) {
#pragma unroll
    for (int it = 0; it < 4; it++) {
        const int e = tid + it * 256;
        const int c = e >> 4, r4 = (e & 15) * 4;
        const float* t = T + c * PLD + r4;
        *reinterpret_cast<float4*>(G + (size_t)c * ld + r4) = make_float4(t[0], t[1], t[2], t[3]);
    }
}
// four 16x16 inverse blocks (global: block p at p*256, column-major ld 16) <-> LDS (stride ILD)
__device__ __forceinline__ void load_inv(float* Tinv, const float* __restrict__ G, int tid) {
    const float4 v = *reinterpret_cast<const float4*>(G + tid * 4);
    const int p = tid >> 6, c = (tid >> 2) & 15, r4 = (tid & 3) * 4;
    float* t = Tinv + p * 16 * ILD + c * ILD + r4;
    t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
}
__device__ __forceinline__ void store_inv(const float* Tinv, float* __restrict__ G, int tid) {
    const int p = tid >> 6, c = (tid >> 2) & 15, r4 = (tid & 3) * 4;
    const float* t = Tinv + p * 16 * ILD + c * ILD + r4;
    *reinterpret_cast<float4*>(G + tid * 4) = make_float4(t[0], t[1], t[2], t[3]);
}

// acc(32x32 per wavefront) = sum_{q<64} A(i,q) * B(j,q);  A(i,q) at As[i*a_si + q*a_sq],
// B(j,q) at Bs[j*b_sj + q*b_sq].  Operands are passed swapped so that the accumulator's
// lane index runs along i:
//   acc[reg] <-> (i = wr*32 + lane%32, j = wc*32 + (reg&3) + 8*(reg>>2) + 4*(lane/32)).
__device__ __forceinline__ f32x16 mma64(const float* As, int a_si, int a_sq, const float* Bs, int b_sj, int b_sq, int wr, int wc,
                               int lane) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    const int li = lane & 31, lk = lane >> 5;
    const float* ap = As + (wr * 32 + li) * a_si + lk * a_sq;
    const float* bp = Bs + (wc * 32 + li) * b_sj + lk * b_sq;
#pragma unroll 8
    for (int q = 0; q < PB; q += 2) {
        const float a = ap[q * a_sq];
        const float b = bp[q * b_sq];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc, 0, 0, 0);
    }
    return acc;
}

// 16x16 tile: acc[reg] <-> (i = lane%16, j = 4*(lane/16) + reg);  acc = sum_{q<16} A(i,q)*B(j,q)
__device__ __forceinline__ f32x4 mma16(const float* As, int a_si, int a_sq, const float* Bs, int b_sj, int b_sq, int lane) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int li = lane & 15, lk = lane >> 4;
    const float* ap = As + li * a_si + lk * a_sq;
    const float* bp = Bs + li * b_sj + lk * b_sq;
#pragma unroll
    for (int q = 0; q < 16; q += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(bp[q * b_sq], ap[q * a_sq], acc, 0, 0, 0);
    return acc;
}

// X <- X L^-T for a 64x64 tile X (Tx) against the lower-triangular 64x64 block in Tl, given
// the inverses of its four 16x16 diagonal blocks (Tinv).  Wavefront w owns rows 16w..16w+15;
// no barrier inside.
__device__ __forceinline__ void tri_solve_fwd(float* Tx, const float* Tl, const float* Tinv, int wave, int lane) {
    const int r0 = 16 * wave, li = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int p = 0; p < 4; p++) {
        // Y_p(i,j) = sum_q X(r0+i, 16p+q) Inv_p(j,q)
        const f32x4 y = mma16(Tx + 16 * p * PLD + r0, 1, PLD, Tinv + p * 16 * ILD, 1, ILD, lane);
#pragma unroll
        for (int g = 0; g < 4; g++) Tx[(16 * p + 4 * lk + g) * PLD + r0 + li] = y[g];
#pragma unroll
        for (int t = p + 1; t < 4; t++) {
            // X(r0+i, 16t+j) -= sum_q Y_p(i,q) L(16t+j, 16p+q)
            const f32x4 u = mma16(Tx + 16 * p * PLD + r0, 1, PLD, Tl + 16 * p * PLD + 16 * t, 1, PLD, lane);
#pragma unroll
            for (int g = 0; g < 4; g++) Tx[(16 * t + 4 * lk + g) * PLD + r0 + li] -= u[g];
        }
    }
}
#define POTRF_STAMP(i)                                                         \
    do {                                                                       \
        if (stamps && tid == 0) stamps[i] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)

// ---- 64x64 Cholesky inside one workgroup ------------------------------------------------
// A: LDS tile holding (at least) the lower triangle of an SPD block; on return its lower
// triangle is L and the strict upper triangle is zero.  Tinv receives the inverses of the
// four 16x16 diagonal blocks of L.  Returns true if a pivot was <= 0.
//
// Four 16-column micro-panels.  Only the pivot chain is serial: wavefront 0 holds one matrix
// row per lane (16 panel entries in registers) and runs the 16 pivot steps with v_readlane
// broadcasts; scaling the pivot column over all 64 lanes IS the panel's triangular solve,
// so no barrier or LDS round trip sits on the chain.  Schedule per panel p:
//   phase F: wave 0 factors panel p   ||  waves 1-3 finish the trailing update of panel p-1
//            (the tiles right of the next panel) and invert diagonal block p-1;
//   phase C: the tiles that make up panel p+1's columns get panel p's update (<= 3 tiles).
// Trailing tiles are 16x16 MFMA tiles (v_mfma_f32_16x16x4_f32).
__device__ __forceinline__ void potrf_tile_update(float* A, int c0, int ti, int tj, int lane) {
    // A(rb.., cb..) -= P(rb.., c0..c0+15) P(cb.., c0..c0+15)^T
    const int li = lane & 15, lk = lane >> 4;
    const int rb = c0 + 16 + 16 * ti, cb = c0 + 16 + 16 * tj;
    const f32x4 u = mma16(A + c0 * PLD + rb, 1, PLD, A + c0 * PLD + cb, 1, PLD, lane);
#pragma unroll
    for (int g = 0; g < 4; g++) A[(cb + 4 * lk + g) * PLD + rb + li] -= u[g];
}

// Inverse of the 16x16 diagonal block b by forward substitution in registers: lane r (< 16)
// holds row r of the block, lane c owns column c of the inverse, L(r,q) reaches every lane
// through v_readlane.  fp32 is enough for 16x16 blocks (measured against fp64-formed
// inverses: no difference in the filter state; 64x64 inverses did need fp64).
__device__ __forceinline__ void potrf_inverse16(const float* A, float* Tinv, int b, int lane) {
    const int o = 16 * b, li = lane & 15;
    float lrow[16], acc[16], x[16];
#pragma unroll
    for (int q = 0; q < 16; q++) lrow[q] = A[(o + q) * PLD + o + li];  // L(o+li, o+q)
    const float dinv_own = 1.0f / A[(o + li) * PLD + o + li];
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = (r == li) ? 1.f : 0.f;
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const float xq = acc[q] * lane_bcast(dinv_own, q);
        x[q] = xq;
#pragma unroll
        for (int r = q + 1; r < 16; r++) acc[r] = __builtin_fmaf(-lane_bcast(lrow[q], r), xq, acc[r]);
    }
    if (lane < 16) {
#pragma unroll
        for (int r = 0; r < 16; r++) Tinv[b * 16 * ILD + li * ILD + r] = x[r];
    }
}

__device__ __forceinline__ bool potrf64_lds(float* A, float* Tinv, int tid, long long* stamps = nullptr) {
    const int lane = tid & 63;
    const int wave = tid >> 6;
    bool bad = false;
    POTRF_STAMP(1);
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const int c0 = 16 * p;
        // ---- phase F ----
        if (wave == 0) {
            float a[16];
#pragma unroll
            for (int j = 0; j < 16; j++) a[j] = A[(c0 + j) * PLD + lane];
#pragma unroll
            for (int k = 0; k < 16; k++) {
                float d = lane_bcast(a[k], c0 + k);
                if (!(d > 0.f)) {
                    bad = true;
                    d = 1e-20f;
                }
                const float inv = __builtin_amdgcn_rsqf(d);  // v_rsq_f32, <= 1 ulp
                const float lkk = d * inv;
                a[k] = (lane == c0 + k) ? lkk : a[k] * inv;
#pragma unroll
                for (int j = k + 1; j < 16; j++) {
                    const float sj = lane_bcast(a[k], c0 + j);
                    a[j] = __builtin_fmaf(-a[k], sj, a[j]);
                }
            }
            if (lane >= c0) {
#pragma unroll
                for (int j = 0; j < 16; j++) A[(c0 + j) * PLD + lane] = (lane - c0 >= j) ? a[j] : 0.f;
            }
        } else if (p == 1) {
            // rest of panel 0's trailing update: tiles (1,1) | (2,1),(2,2); wave 1 inverts block 0
            if (wave == 1) potrf_inverse16(A, Tinv, 0, lane);
            if (wave == 2) potrf_tile_update(A, 0, 1, 1, lane);
            if (wave == 3) {
                potrf_tile_update(A, 0, 2, 1, lane);
                potrf_tile_update(A, 0, 2, 2, lane);
            }
        } else if (p == 2) {
            if (wave == 1) potrf_tile_update(A, 16, 1, 1, lane);
            if (wave == 2) potrf_inverse16(A, Tinv, 1, lane);
        } else if (p == 3) {
            if (wave == 3) potrf_inverse16(A, Tinv, 2, lane);
        }
        __syncthreads();
        POTRF_STAMP(2 + 2 * p);
        // ---- phase C: panel p's update of the next panel's columns (tiles (ti,0)) ----
        if (p < 3) {
            if (wave < 3 - p) potrf_tile_update(A, c0, wave, 0, lane);
            __syncthreads();
        }
        POTRF_STAMP(3 + 2 * p);
    }
    // zero the strict upper triangle (the tile may have carried the symmetric upper part);
    // wave 0 inverts the last diagonal block meanwhile (it only reads on/below the diagonal)
    if (wave == 0) {
        potrf_inverse16(A, Tinv, 3, lane);
    } else {
        for (int e = tid - 64; e < PB * PB; e += 192) {
            const int r = e % PB, c = e / PB;
            if (r < c) A[c * PLD + r] = 0.f;
        }
    }
    __syncthreads();
    POTRF_STAMP(10);
    return bad;
}

// Diagnostic twin of potrf64_kernel: same work, s_memtime stamps after every phase.
__global__ __launch_bounds__(256) void potrf64_stamp_kernel(const float* __restrict__ S, int lds, float* __restrict__ L,
                                                            int ldl, float* __restrict__ Linv, long long* stamps) {
    __shared__ float A[PB * PLD];
    __shared__ float Tinv[INV_LDS];
    const int tid = threadIdx.x;
    if (tid == 0) stamps[0] = (long long)__builtin_amdgcn_s_memtime();
    load_tile(A, S, lds, tid);
    __syncthreads();
    potrf64_lds(A, Tinv, tid, stamps);
    store_tile(A, L, ldl, tid);
    store_inv(Tinv, Linv, tid);
    __syncthreads();
    if (tid == 0) stamps[11] = (long long)__builtin_amdgcn_s_memtime();
}

// Factor the first diagonal block (step "-1" of the sweep).
__global__ __launch_bounds__(256) void potrf64_kernel(const float* __restrict__ S, int lds, float* __restrict__ L,
                                                      int ldl, float* __restrict__ Linv, int* info) {
    __shared__ float A[PB * PLD];
    __shared__ float Tinv[INV_LDS];
    const int tid = threadIdx.x;
    load_tile(A, S, lds, tid);
    __syncthreads();
    const bool bad = potrf64_lds(A, Tinv, tid);
    store_tile(A, L, ldl, tid);
    store_inv(Tinv, Linv, tid);
    if (bad && tid == 0) atomicOr(info, 1);
}

// Block step k of the right-looking sweep over the augmented matrix [A; X; I].
// grid.x = r(r+1)/2 triangular tiles of A (r = mb-1-k) + rb*r rectangular tiles of the extra
// row blocks (X: the forward substitution Y L^T = X rides along; I: yields L^-T).
// idb0 = first identity row block: its block row q is still zero left of column block q.
__global__ __launch_bounds__(256) void chol_step_kernel(float* __restrict__ S, int lds, float* __restrict__ L, int ldl,
                                                        float* __restrict__ Linv, int k, int mb, int idb0, int* info) {
    __shared__ float Ti[PB * PLD];   // A_ik, then L_ik
    __shared__ float Tj[PB * PLD];   // A_jk, then L_jk
    __shared__ float Tl[PB * PLD];   // L_kk, later the updated next diagonal tile
    __shared__ float Tinv[INV_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave & 1, wc = wave >> 1;
    const int rr = mb - 1 - k, ntri = rr * (rr + 1) / 2;
    int t = blockIdx.x, i, j;
    if (t < ntri) {
        // lower-triangular tile index: t = ii(ii+1)/2 + jj, 0 <= jj <= ii
        int ii = 0;
        while ((ii + 1) * (ii + 2) / 2 <= t) ii++;
        i = k + 1 + ii;
        j = k + 1 + (t - ii * (ii + 1) / 2);
    } else {
        t -= ntri;
        i = mb + t / rr;
        j = k + 1 + t % rr;
        if (i >= idb0 && i - idb0 > k) return;  // identity block row: block (i,k) is still zero
    }

    load_tile(Tl, L + (size_t)k * PB * ldl + (size_t)k * PB, ldl, tid);
    load_inv(Tinv, Linv + (size_t)k * PB * PB, tid);
    load_tile(Ti, S + (size_t)k * PB * lds + (size_t)i * PB, lds, tid);
    if (i != j) load_tile(Tj, S + (size_t)k * PB * lds + (size_t)j * PB, lds, tid);
    __syncthreads();
    tri_solve_fwd(Ti, Tl, Tinv, wave, lane);             // L_ik = A_ik L_kk^-T
    if (i != j) tri_solve_fwd(Tj, Tl, Tinv, wave, lane);  // L_jk
    __syncthreads();
    if (j == k + 1) store_tile(Ti, L + (size_t)k * PB * ldl + (size_t)i * PB, ldl, tid);
    // A_ij(r,s) -= sum_c L_ik(r,c) L_jk(s,c)
    const float* Bj = (i != j) ? Tj : Ti;
    const f32x16 up = mma64(Ti, 1, PLD, Bj, 1, PLD, wr, wc, lane);
    float* Sij = S + (size_t)j * PB * lds + (size_t)i * PB;
    const int r = wr * 32 + (lane & 31);
    if (i == k + 1 && j == k + 1) {
        // next diagonal tile: update into LDS and factor it now (look-ahead)
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int c = wc * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
            Tl[c * PLD + r] = Sij[(size_t)c * lds + r] - up[q];
        }
        __syncthreads();
        const bool bad = potrf64_lds(Tl, Tinv, tid);
        store_tile(Tl, L + (size_t)j * PB * ldl + (size_t)i * PB, ldl, tid);
        store_inv(Tinv, Linv + (size_t)(k + 1) * PB * PB, tid);
        if (bad && tid == 0) atomicOr(info, 1);
    } else {
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int c = wc * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
            Sij[(size_t)c * lds + r] -= up[q];
        }
    }
}

// X_k <- X_k L_kk^-T for the extra row blocks at the last block column (no trailing tiles left)
__global__ __launch_bounds__(256) void chol_last_panel_kernel(const float* __restrict__ S, int lds, float* __restrict__ L,
                                                              int ldl, const float* __restrict__ Linv, int k, int mb) {
    __shared__ float Ti[PB * PLD];
    __shared__ float Tl[PB * PLD];
    __shared__ float Tinv[INV_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = mb + blockIdx.x;
    load_tile(Tl, L + (size_t)k * PB * ldl + (size_t)k * PB, ldl, tid);
    load_inv(Tinv, Linv + (size_t)k * PB * PB, tid);
    load_tile(Ti, S + (size_t)k * PB * lds + (size_t)i * PB, lds, tid);
    __syncthreads();
    tri_solve_fwd(Ti, Tl, Tinv, wave, lane);
    __syncthreads();
    store_tile(Ti, L + (size_t)k * PB * ldl + (size_t)i * PB, ldl, tid);
}

}  // namespace

void launch_potrf_stamps(ekfvio_filter* f, const float* S, int ld, float* L, float* Linv, long long* d_stamps) {
    hipLaunchKernelGGL(potrf64_stamp_kernel, dim3(1), dim3(256), 0, f->stream, S, ld, L, ld, Linv, d_stamps);
}

void launch_chol_sweep(ekfvio_filter* f, float* Saug, float* Laug, float* Linv, int m_pad, int n_pad, int ld) {
    ProfScope ps(f, PC_CHOL, (double)m_pad * m_pad * m_pad / 3.0 + (double)(n_pad + m_pad / 2) * m_pad * m_pad);
    const int mb = m_pad / PB;
    const int rb = n_pad / PB + mb;        // extra row blocks: X then I
    const int idb0 = mb + n_pad / PB;
    hipLaunchKernelGGL(potrf64_kernel, dim3(1), dim3(256), 0, f->stream, Saug, ld, Laug, ld, Linv, f->info);
    for (int k = 0; k + 1 < mb; k++) {
        const int r = mb - 1 - k;
        hipLaunchKernelGGL(chol_step_kernel, dim3(r * (r + 1) / 2 + rb * r), dim3(256), 0, f->stream, Saug, ld, Laug, ld,
                           Linv, k, mb, idb0, f->info);
    }
    hipLaunchKernelGGL(chol_last_panel_kernel, dim3(rb), dim3(256), 0, f->stream, Saug, ld, Laug, ld, Linv, mb - 1, mb);
}

// K = X A^-1 = Y L^-1 with Y = X L^-T and L^-T (both from the sweep): one MFMA GEMM that
// skips the structurally zero part of the triangular operand and prunes like sparseView
// (:580).  refine != 0 adds one step of residual refinement against L itself,
// K <- K + (Y - K L) L^-1 (two more GEMMs); on the filter's matrices it changes nothing
// measurable (the error is dominated by the fp32 factor itself), so it is off by default.
void launch_gain_from_sweep(ekfvio_filter* f, const float* Laug, int m_pad, int n_pad, int ld, int n, float* K,
                            float* scratch, int ldk, int refine) {
    ProfScope ps(f, PC_SOLVE, (refine ? 3.0 : 1.0) * n * (double)m_pad * m_pad);
    const float* Lf = Laug;
    const float* Y = Laug + m_pad;
    const float* LinvT = Laug + m_pad + n_pad;
    launch_gemm(f->stream, 1, n, m_pad, m_pad, 1.f, Y, ld, LinvT, ld, 0.f, nullptr, 0, K, ldk, refine ? 0 : 1, 1);
    if (refine) {
        launch_gemm(f->stream, 0, n, m_pad, m_pad, -1.f, K, ldk, Lf, ld, 1.f, Y, ld, scratch, ldk, 0, 1);     // Y - K L
        launch_gemm(f->stream, 1, n, m_pad, m_pad, 1.f, scratch, ldk, LinvT, ld, 1.f, K, ldk, K, ldk, 1, 1);  // + prune
    }
}
