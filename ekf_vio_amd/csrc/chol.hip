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
//
// A numerically indefinite A.  The reference's SimplicialLDLT carries negative pivots along (only an exactly zero one
// is a NumericalIssue), and with its tiny measurement noise (1e-5 px^2 / fx^2) or many landmarks on the raw prior the
// fp32 S really is indefinite (cond(S) ~ 1e7 .. 1e12).  A Cholesky would have to clamp such a pivot and destroy the
// gain.  So a diagonal tile whose fast factorisation meets a non-positive pivot is factored again by a slow, plain
// path as A = U S U^T, S = diag(+-1), U lower triangular with U_cc = sign_c sqrt|d_c| (column c of the eliminated tile
// scaled by rsqrt|d_c|): everything downstream keeps its form -- the substitutions use U and the inverses of its 16x16
// diagonal blocks, trailing updates become A_ij -= U_ik S_k U_jk^T (the sign of column c applied to the left operand as
// it is read, from a 64-bit mask per block column), and the gain K = Y S U^-1 gets S through the identity rows, whose
// panel blocks are stored with the column signs applied.  With no negative pivot every mask is zero and every branch
// on it is a scalar compare: the fast path's instruction streams are untouched.  The pivot flag (info bit 0) is still
// raised, as a warning that S was not positive definite in fp32.
#include "common.h"

#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define PB 64
#ifndef PLD
#define PLD 68  // LDS tile stride: element (r,c) of a tile lives at c*PLD + r  (column-major); a multiple of 4 keeps
                // 16-byte LDS accesses aligned (tile movers), 68 = 4 mod 32 spreads the columns over the banks
#endif
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

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E)
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}
// acc -= a(lane R of this lane's 16-lane row) * b   — one VALU op, no SGPR round trip.
// The DPP operand must not have been written by the VALU in the two preceding issue slots
// (CDNA3 ISA 4.5, "VALU writes VGPR -> DPP reads that VGPR"): callers keep it old.
template <int R>
__device__ __forceinline__ void dpp_fnma(float& acc, float a, float b) {
    asm volatile("v_fmac_f32_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b), "n"(R));
}
// a(lane R of this lane's 16-lane row) * b
template <int R>
__device__ __forceinline__ float dpp_mul(float a, float b) {
    float r;
    asm volatile("s_nop 1\n\tv_mul_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(a), "v"(b), "n"(R));
    return r;
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
#if PLD % 4 == 0
        *reinterpret_cast<float4*>(t) = v;  // one ds_write_b128
#else
        t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
#endif
    }
}
__device__ __forceinline__ void store_tile(const float* T, float* __restrict__ G, int ld, int tid) {
#pragma unroll
    for (int it = 0; it < 4; it++) {
        const int e = tid + it * 256;
        const int c = e >> 4, r4 = (e & 15) * 4;
        const float* t = T + c * PLD + r4;
#if PLD % 4 == 0
        *reinterpret_cast<float4*>(G + (size_t)c * ld + r4) = *reinterpret_cast<const float4*>(t);
#else
        *reinterpret_cast<float4*>(G + (size_t)c * ld + r4) = make_float4(t[0], t[1], t[2], t[3]);
#endif
    }
}
// The sign mask of a block column is requested like the tiles: a VECTOR load (every lane the same address), so that it
// retires with the tile loads issued around it.  As a scalar load it sat on lgkmcnt with the LDS traffic and its cold-L2
// latency (the mask was written by the previous launch, on another compute unit) landed on the chain: 2.3 us per filter
// step at N = 256 (same-box A/B, scripts/ab_lib.py).  The address goes through a VGPR zero the compiler cannot see
// through, which makes it an ordinary vector load whose wait the compiler places itself (an asm load would not be
// tracked: under register pressure the compiler may copy its destination before the data has arrived).
// sign_mask_request next to the tile loads, sign_mask_value (first use) after the barrier that follows them.
__device__ __forceinline__ unsigned long long sign_mask_request(const unsigned long long* p) {
    unsigned z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return p[z];
}
__device__ __forceinline__ unsigned long long sign_mask_value(unsigned long long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
// a panel block of the identity rows with the column signs of its block column applied (the gain GEMM K = Y S U^-1 takes
// S from here; `neg` is zero unless the block had negative pivots)
__device__ __forceinline__ void store_tile_signed(const float* T, float* __restrict__ G, int ld, int tid, unsigned long long neg) {
#pragma unroll
    for (int it = 0; it < 4; it++) {
        const int e = tid + it * 256;
        const int c = e >> 4, r4 = (e & 15) * 4;
        const float* t = T + c * PLD + r4;
        const float sg = ((neg >> c) & 1ull) ? -1.f : 1.f;
        *reinterpret_cast<float4*>(G + (size_t)c * ld + r4) = make_float4(sg * t[0], sg * t[1], sg * t[2], sg * t[3]);
    }
}
// lower triangle of a diagonal tile; the strict upper triangle is written as zero
__device__ __forceinline__ void store_tile_lower(const float* T, float* __restrict__ G, int ld, int tid) {
#pragma unroll
    for (int it = 0; it < 4; it++) {
        const int e = tid + it * 256;
        const int c = e >> 4, r4 = (e & 15) * 4;
        const float* t = T + c * PLD + r4;
#if PLD % 4 == 0
        const float4 q = *reinterpret_cast<const float4*>(t);
        *reinterpret_cast<float4*>(G + (size_t)c * ld + r4) =
            make_float4(r4 >= c ? q.x : 0.f, r4 + 1 >= c ? q.y : 0.f, r4 + 2 >= c ? q.z : 0.f, r4 + 3 >= c ? q.w : 0.f);
#else
        *reinterpret_cast<float4*>(G + (size_t)c * ld + r4) =
            make_float4(r4 >= c ? t[0] : 0.f, r4 + 1 >= c ? t[1] : 0.f, r4 + 2 >= c ? t[2] : 0.f, r4 + 3 >= c ? t[3] : 0.f);
#endif
    }
}
// Columns C0 .. C0+NC-1 of a tile by NT threads (t = 0 .. NT-1): what wavefronts without work in a phase of the 64x64
// factorisation store while wavefront 0 runs its pivot chain (potrf64_lds, `idle`).  LOWER: zero above the diagonal.
template <int C0, int NC, int NT, bool LOWER>
__device__ __forceinline__ void store_cols(const float* T, float* __restrict__ G, int ld, int t) {
    static_assert(PLD % 4 == 0, "vector LDS reads");
#pragma unroll
    for (int it = 0; it < (NC * 16 + NT - 1) / NT; it++) {
        const int e = t + it * NT;
        if ((NC * 16) % NT != 0 && e >= NC * 16) break;
        const int c = C0 + (e >> 4), r4 = (e & 15) * 4;
        const float4 q = *reinterpret_cast<const float4*>(T + c * PLD + r4);
        *reinterpret_cast<float4*>(G + (size_t)c * ld + r4) =
            LOWER ? make_float4(r4 >= c ? q.x : 0.f, r4 + 1 >= c ? q.y : 0.f, r4 + 2 >= c ? q.z : 0.f, r4 + 3 >= c ? q.w : 0.f) : q;
    }
}
// inverse blocks P0 .. P0+NP-1, thread t = 0 .. 64*NP-1
__device__ __forceinline__ void store_inv_blocks(const float* Tinv, float* __restrict__ G, int p0, int t) {
    const int p = p0 + (t >> 6), c = (t >> 2) & 15, r4 = (t & 3) * 4;
    const float* s = Tinv + p * 16 * ILD + c * ILD + r4;
    *reinterpret_cast<float4*>(G + (p0 * 64 + t) * 4) = make_float4(s[0], s[1], s[2], s[3]);
}
struct NoIdleWork {
    template <class PC>
    __device__ __forceinline__ void operator()(PC, int) const {}
};
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
    // completely unrolled: no loop, so loads a caller has in flight stay in flight across the product (chol_step_kernel)
#pragma unroll
    for (int q = 0; q < PB; q += 2) {
        const float a = ap[q * a_sq];
        const float b = bp[q * b_sq];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc, 0, 0, 0);
    }
    return acc;
}
// The same product with the column signs of an indefinite block applied to A's k index (bit q of `neg` set: column q
// of the panel belongs to a negative pivot): sum_q s_q A(i,q) B(j,q).  Rare path (see the header).
// Not inlined: the rare path must not shape the register allocation and scheduling of the kernels' hot loops.
__device__ __attribute__((noinline)) f32x16 mma64_signed(const float* As, int a_si, int a_sq, const float* Bs, int b_sj, int b_sq, int wr,
                                                         int wc, int lane, unsigned long long neg) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    const int li = lane & 31, lk = lane >> 5;
    const float* ap = As + (wr * 32 + li) * a_si + lk * a_sq;
    const float* bp = Bs + (wc * 32 + li) * b_sj + lk * b_sq;
#pragma unroll 8
    for (int q = 0; q < PB; q += 2) {
        float a = ap[q * a_sq];
        if ((neg >> (q + lk)) & 1ull) a = -a;
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
// no barrier inside.  The 16x64 row block lives in MFMA accumulator layout for the whole
// substitution (lane l, block b, register v <-> X(16w + l%16, 16b + 4*(l/16) + v)): a result
// register v is fed straight back as the k-slice {4g+v} of the next product's A operand (the
// B operand is read from LDS with the matching k permutation), so the dependent chain is
// MFMA -> MFMA with no LDS round trip; all B operands are preloaded.
// The substitution's B operands, read once into registers: binv = the four 16x16 inverse diagonal blocks, nl = the six
// off-diagonal 16x16 blocks of -L.  Separate from the substitution itself so that a caller can fetch them, then re-use
// L_kk's LDS tile for a row block before it solves (chol_step_la_kernel's near tiles).
struct TriOps {
    float binv[4][4];   // Inv_p(li, 4g+v)
    float nl[6][4];     // -L(16t+li, 16p+4g+v), (p,t) pairs in the order (0,1)(0,2)(0,3)(1,2)(1,3)(2,3)
};
__device__ __forceinline__ void tri_solve_preload(TriOps& o, const float* Tl, const float* Tinv, int lane) {
    const int li = lane & 15, g = lane >> 4;
#pragma unroll
    for (int p = 0; p < 4; p++)
#pragma unroll
        for (int v = 0; v < 4; v++) o.binv[p][v] = Tinv[p * 16 * ILD + (4 * g + v) * ILD + li];
    {
        int e = 0;
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int t = p + 1; t < 4; t++, e++)
#pragma unroll
                for (int v = 0; v < 4; v++) o.nl[e][v] = -Tl[(16 * p + 4 * g + v) * PLD + 16 * t + li];
    }
}
// pin every operand in a register here: otherwise the LDS reads are sunk next to their MFMA
// and their latency lands on the dependent chain
__device__ __forceinline__ void tri_solve_pin(TriOps& o) {
#pragma unroll
    for (int p = 0; p < 4; p++)
#pragma unroll
        for (int v = 0; v < 4; v++) asm volatile("" : "+v"(o.binv[p][v]));
#pragma unroll
    for (int q = 0; q < 6; q++)
#pragma unroll
        for (int v = 0; v < 4; v++) asm volatile("" : "+v"(o.nl[q][v]));
}
template <int NT>
__device__ __forceinline__ void tri_solve_load_x(f32x4 (&x)[NT][4], float* const (&Txs)[NT], int wave, int lane) {
    const int r0 = 16 * wave, li = lane & 15, g = lane >> 4;
#pragma unroll
    for (int n = 0; n < NT; n++)
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int v = 0; v < 4; v++) x[n][b][v] = Txs[n][(16 * b + 4 * g + v) * PLD + r0 + li];
}
// the dependent MFMA chain and the write-back
template <int NT>
__device__ __forceinline__ void tri_solve_run(f32x4 (&x)[NT][4], float* const (&Txs)[NT], const TriOps& o, int wave, int lane) {
    const int r0 = 16 * wave, li = lane & 15, g = lane >> 4;
    int e0 = 0;
#pragma unroll
    for (int p = 0; p < 4; p++) {
        f32x4 y[NT];
#pragma unroll
        for (int n = 0; n < NT; n++) y[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        // the NT tiles are independent chains: interleaved they keep the MFMA pipe busy
#pragma unroll
        for (int v = 0; v < 4; v++)
#pragma unroll
            for (int n = 0; n < NT; n++) y[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.binv[p][v], x[n][p][v], y[n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NT; n++) x[n][p] = y[n];
        int e = e0;
#pragma unroll
        for (int t = p + 1; t < 4; t++, e++)
#pragma unroll
            for (int v = 0; v < 4; v++)
#pragma unroll
                for (int n = 0; n < NT; n++) x[n][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.nl[e][v], y[n][v], x[n][t], 0, 0, 0);
        e0 = e;
    }
#pragma unroll
    for (int n = 0; n < NT; n++)
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int v = 0; v < 4; v++) Txs[n][(16 * b + 4 * g + v) * PLD + r0 + li] = x[n][b][v];
}
// (the same substitution in one piece: the sweep's hot kernels keep this form, whose schedule is the measured one)
template <int NT>
__device__ __forceinline__ void tri_solve_fwd_n(float* const (&Txs)[NT], const float* Tl, const float* Tinv, int wave, int lane,
                                                long long* st = nullptr) {
#define TSTAMP(i) do { if (st && wave == 0 && lane == 0) st[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
    TSTAMP(0);
    const int r0 = 16 * wave, li = lane & 15, g = lane >> 4;
    f32x4 x[NT][4];
    float binv[4][4];   // Inv_p(li, 4g+v)
    float nl[6][4];     // -L(16t+li, 16p+4g+v), (p,t) pairs in the order (0,1)(0,2)(0,3)(1,2)(1,3)(2,3)
#pragma unroll
    for (int n = 0; n < NT; n++)
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int v = 0; v < 4; v++) x[n][b][v] = Txs[n][(16 * b + 4 * g + v) * PLD + r0 + li];
#pragma unroll
    for (int p = 0; p < 4; p++)
#pragma unroll
        for (int v = 0; v < 4; v++) binv[p][v] = Tinv[p * 16 * ILD + (4 * g + v) * ILD + li];
    {
        int e = 0;
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int t = p + 1; t < 4; t++, e++)
#pragma unroll
                for (int v = 0; v < 4; v++) nl[e][v] = -Tl[(16 * p + 4 * g + v) * PLD + 16 * t + li];
    }
    // pin every operand in a register here: otherwise the LDS reads are sunk next to their MFMA
    // and their latency lands on the dependent chain
#pragma unroll
    for (int p = 0; p < 4; p++)
#pragma unroll
        for (int v = 0; v < 4; v++) asm volatile("" : "+v"(binv[p][v]));
#pragma unroll
    for (int q = 0; q < 6; q++)
#pragma unroll
        for (int v = 0; v < 4; v++) asm volatile("" : "+v"(nl[q][v]));
    TSTAMP(1);
    int e0 = 0;
#pragma unroll
    for (int p = 0; p < 4; p++) {
        f32x4 y[NT];
#pragma unroll
        for (int n = 0; n < NT; n++) y[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        // the NT tiles are independent chains: interleaved they keep the MFMA pipe busy
#pragma unroll
        for (int v = 0; v < 4; v++)
#pragma unroll
            for (int n = 0; n < NT; n++) y[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(binv[p][v], x[n][p][v], y[n], 0, 0, 0);
#pragma unroll
        for (int n = 0; n < NT; n++) x[n][p] = y[n];
        int e = e0;
#pragma unroll
        for (int t = p + 1; t < 4; t++, e++)
#pragma unroll
            for (int v = 0; v < 4; v++)
#pragma unroll
                for (int n = 0; n < NT; n++) x[n][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(nl[e][v], y[n][v], x[n][t], 0, 0, 0);
        e0 = e;
    }
    TSTAMP(2);
#pragma unroll
    for (int n = 0; n < NT; n++)
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int v = 0; v < 4; v++) Txs[n][(16 * b + 4 * g + v) * PLD + r0 + li] = x[n][b][v];
    TSTAMP(3);
}
__device__ __forceinline__ void tri_solve_fwd(float* Tx, const float* Tl, const float* Tinv, int wave, int lane, long long* st = nullptr) {
    float* const t[1] = {Tx};
    tri_solve_fwd_n<1>(t, Tl, Tinv, wave, lane, st);
}
__device__ __forceinline__ void tri_solve_fwd2(float* Tx, float* Ty, const float* Tl, const float* Tinv, int wave, int lane) {
    float* const t[2] = {Tx, Ty};
    tri_solve_fwd_n<2>(t, Tl, Tinv, wave, lane);
}
#define POTRF_STAMP(i)                                                         \
    do {                                                                       \
        if (stamps && tid == 0) stamps[i] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)

#define POTRF_WSTAMP(i)                                                                               \
    do {                                                                                              \
        if (stamps && lane == 0) stamps[16 + wave * 16 + (i)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)

// ---- 64x64 Cholesky inside one workgroup ------------------------------------------------
// A: LDS tile holding (at least) the lower triangle of an SPD block; on return its lower
// triangle is L (the strict upper triangle is left as it was: store with store_tile_lower).
// Tinv receives the inverses of the
// four 16x16 diagonal blocks of L.  Returns true if a pivot was <= 0.
//
// Four 16-column micro-panels.  Only the pivot chain is serial: wavefront 0 holds one matrix
// row per lane (16 panel entries in registers) and runs the 16 pivot steps with v_readlane
// broadcasts; scaling the pivot column over all 64 lanes IS the panel's triangular solve,
// so no barrier or LDS round trip sits on the chain.  Schedule per panel p:
//   phase F: wave 0 factors panel p   ||  waves 1-3 finish the trailing update of panel p-1
//            (the tiles right of the next panel); wave 1 inverts diagonal block 0 meanwhile, the
//            inverses of blocks 1-3 fall out of wave 0's own sweep (identity rows in its dead lanes);
//   phase C: the tiles that make up panel p+1's columns get panel p's update (<= 3 tiles).
// Trailing tiles are 16x16 MFMA tiles (v_mfma_f32_16x16x4_f32).
__device__ __forceinline__ void potrf_tile_update(float* A, int c0, int ti, int tj, int lane) {
    // A(rb.., cb..) -= P(rb.., c0..c0+15) P(cb.., c0..c0+15)^T
    const int li = lane & 15, lk = lane >> 4;
    const int rb = c0 + 16 + 16 * ti, cb = c0 + 16 + 16 * tj;
    const f32x4 u = mma16(A + c0 * PLD + rb, 1, PLD, A + c0 * PLD + cb, 1, PLD, lane);
    float t[4];
#pragma unroll
    for (int g = 0; g < 4; g++) t[g] = A[(cb + 4 * lk + g) * PLD + rb + li];  // all reads before the first write (see chol_step_kernel)
#pragma unroll
    for (int g = 0; g < 4; g++) A[(cb + 4 * lk + g) * PLD + rb + li] = t[g] - u[g];
}

// Inverse of the 16x16 diagonal block b by forward substitution in registers: lane r (< 16)
// holds row r of the block, lane c owns column c of the inverse, L(r,q) reaches every lane
// through v_readlane.  fp32 is enough for 16x16 blocks (measured against fp64-formed
// inverses: no difference in the filter state; 64x64 inverses did need fp64).
__device__ __forceinline__ void potrf_inverse16(const float* A, float* Tinv, int b, int lane) {
    const int o = 16 * b, li = lane & 15;
    float lrow[16], acc[16], x[16];
#pragma unroll
    for (int q = 0; q < 16; q++) lrow[q] = A[(o + q) * PLD + o + li];  // L(o+li, o+q), same in all four lane rows
    const float dd = A[(o + li) * PLD + o + li];
    float dinv_own = __builtin_amdgcn_rcpf(dd);
    dinv_own = __builtin_fmaf(__builtin_fmaf(-dd, dinv_own, 1.f), dinv_own, dinv_own);  // one Newton step: ~0.5 ulp
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = (r == li) ? 1.f : 0.f;
    // x(q) = acc(q) / L(q,q);  acc(r) -= L(r,q) x(q): L(r,q) sits in lane r of every 16-lane row and
    // reaches the other lanes as the DPP operand of the multiply-add itself (row_newbcast)
    static_for<0, 16>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const float xq = dpp_mul<q>(dinv_own, acc[q]);
        x[q] = xq;
        static_for<q + 1, 16>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            dpp_fnma<r>(acc[r], lrow[q], xq);
        });
    });
    if (lane < 16) {
#pragma unroll
        for (int r = 0; r < 16; r++) Tinv[b * 16 * ILD + li * ILD + r] = x[r];
    }
}

#include "potrf_chain.inc"

// idle(integral_constant<p>, wave): called by wavefronts 1-3 in phase F of panel p after their own share of it.  Columns
// left of panel p and the inverses of the blocks before p are final by then (read-only for everyone): a caller can have
// them stored while the chain runs (waves 2-3 are free in phase F of panel 2, waves 1-3 in that of panel 3).
// RAW_BARRIER: the barriers wait for this wavefront's LDS traffic only (s_waitcnt lgkmcnt(0); s_barrier) instead of
// __syncthreads(), which also drains its global loads: an idle hook can then leave loads in flight across phases.
template <int FV = EKF_POTRF_FV, class Idle = NoIdleWork, bool RAW_BARRIER = false>
__device__ __forceinline__ bool potrf64_lds(float* A, float* Tinv, int tid, long long* stamps = nullptr, Idle idle = Idle()) {
#define POTRF_BAR()                                                                                  \
    do {                                                                                             \
        if constexpr (RAW_BARRIER) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   \
        else __syncthreads();                                                                        \
    } while (0)
    const int lane = tid & 63;
    const int wave = tid >> 6;
    bool bad = false;
    POTRF_STAMP(1);
    static_for<0, 4>([&](auto pc) {
        constexpr int P = decltype(pc)::value;
        constexpr int p = P;
        constexpr int c0 = 16 * p;
        // ---- phase F ----
        if (FV >= 10 && wave == 0) {
            // FV 10: the generated stream with the panel's LDS loads and stores inside it.  Lanes of the panel (and
            // the dead lanes above it, harmlessly) write their matrix column, the identity lanes of panels 1-3 the rows
            // of the diagonal block's inverse.
            const unsigned lb = static_cast<unsigned>(reinterpret_cast<size_t>(A + c0 * PLD + lane));
            const bool inv_lane = p >= 1 && lane < 16;
            const unsigned sb = inv_lane ? static_cast<unsigned>(reinterpret_cast<size_t>(Tinv + p * 16 * ILD + lane * ILD)) : lb;
            // FV 12 / 13: the same panel rescheduled so that no LDS latency sits on the pivot chain (gen_ls3)
            if constexpr (FV == 12) potrf_panel16_chain_ls3<c0>(lb, sb, inv_lane ? 4u : (unsigned)(PLD * 4), (c0 + (lane & 15)) << 2);
            else if constexpr (FV == 13) potrf_panel16_chain_ls3w<c0>(lb, sb, inv_lane ? 4u : (unsigned)(PLD * 4), (c0 + (lane & 15)) << 2);
            else potrf_panel16_chain_ls<c0>(lb, sb, inv_lane ? 4u : (unsigned)(PLD * 4), (c0 + (lane & 15)) << 2);
        } else if (wave == 0) {
            float a[16];
#pragma unroll
            for (int j = 0; j < 16; j++) a[j] = A[(c0 + j) * PLD + lane];
            // From the second panel on the first 16 lanes hold rows above the panel (dead values of the symmetric
            // input).  Waves 1-3 have put the identity there during phase F of panel 0: the column sweep below turns row
            // i of it into row i of L_pp^-T, i.e. column i of the diagonal block's inverse, for free (the same forward
            // substitution potrf_inverse16 runs for block 0).
#pragma unroll
            for (int j = 0; j < 16; j++) asm volatile("" : "+v"(a[j]));  // all LDS reads issue before the chain
            // Column k of the panel: pivot (clamped: a non-positive or NaN pivot leaves L_kk = 1e-10 behind, which the
            // caller's final diagonal check reports), scale, then a[j] -= a[k] * L[c0+j][c0+k] for the columns j > k.
            //   FV 8 (production): the generated, hand-scheduled stream of potrf_chain.inc — column k+1 takes its
            //         multiplier through an SGPR, the others one DPP fma each from a ds_bpermute copy of the diagonal
            //         block's column, issued in the wait states of the next column's pivot chain;
            //   FV 0: the plain formulation (every multiplier v_readlane + v_fma), kept as the reference the generated
            //         stream is checked against: same fma per element in the same order, identical bits
            //         (scripts/potrf_stamps.py prints hashes of L and the inverses for both).
            if (stamps && p == 0 && lane == 0) stamps[14] = (long long)__builtin_amdgcn_s_memtime();
            if constexpr (FV == 8) {
                potrf_panel16_chain<c0>(a, (c0 + (lane & 15)) << 2);
            } else {
                static_for<0, 16>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    const float d = lane_bcast(fmaxf(a[k], 1e-20f), c0 + k);
                    const float inv = __builtin_amdgcn_rsqf(d);  // v_rsq_f32, <= 1 ulp
                    a[k] *= inv;                                 // lane c0+k: d * inv = L_kk
                    static_for<k + 1, 16>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        a[j] = __builtin_fmaf(-a[k], lane_bcast(a[k], c0 + j), a[j]);
                    });
                });
            }
            if (stamps && p == 0 && lane == 0) stamps[15] = (long long)__builtin_amdgcn_s_memtime();
            if (lane >= c0) {
#pragma unroll
                for (int j = 0; j < 16; j++) A[(c0 + j) * PLD + lane] = (lane - c0 >= j) ? a[j] : 0.f;
            }
            if constexpr (p >= 1) {
                if (lane < 16) {
#pragma unroll
                    for (int j = 0; j < 16; j++) Tinv[p * 16 * ILD + lane * ILD + j] = a[j];
                }
            }
        } else if (p == 0) {
            // identity into the dead blocks (rows 0-15 of the column panels 1-3): 768 entries over 192 threads
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int e = (tid - 64) + 192 * q;  // 0 .. 767
                const int c = 16 + (e >> 4), r = e & 15;
                A[c * PLD + r] = ((c & 15) == r) ? 1.f : 0.f;
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
        }
        if (wave != 0) idle(pc, wave);
        POTRF_WSTAMP(2 * p);
        POTRF_BAR();
        POTRF_STAMP(2 + 2 * p);
        // ---- phase C: panel p's update of the next panel's columns (tiles (ti,0)) ----
        if (p < 3) {
            if (wave < 3 - p) potrf_tile_update(A, c0, wave, 0, lane);
            POTRF_WSTAMP(2 * p + 1);
            POTRF_BAR();
        }
        POTRF_STAMP(3 + 2 * p);
    });
    // the strict upper triangle still holds the symmetric input: store_tile_lower drops it
    POTRF_STAMP(10);
    // every wavefront looks at the 64 diagonal entries itself: the answer is uniform over the workgroup without a barrier
    bad = __ballot(!(A[lane * PLD + lane] > 2e-10f)) != 0ull;
    return bad;
#undef POTRF_BAR
}

// Slow path for a tile the fast factorisation flagged (see the header): A = U S U^T by plain right-looking column
// elimination on wavefront 0 (one matrix row per lane, the tile in LDS), then the four 16x16 inverses of U's diagonal
// blocks, one per wavefront.  A must hold the lower triangle of the ORIGINAL tile again (callers rebuild it).  Returns
// the mask of negative pivots (uniform over the workgroup).  A pivot of magnitude below 1e-20 is treated as +-1e-20.
__device__ __attribute__((noinline)) unsigned long long potrf64_signed(float* A, float* Tinv, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    __syncthreads();
    if (wave == 0) {
        for (int k = 0; k < PB; k++) {
            const float d = A[k * PLD + k];
            const float inv = __builtin_amdgcn_rsqf(fmaxf(fabsf(d), 1e-20f));
            const float u = (lane >= k) ? A[k * PLD + lane] * inv : 0.f;   // column k of U (U_kk = sign * sqrt|d|)
            if (lane >= k) A[k * PLD + lane] = u;
            const float su = (d < 0.f) ? -u : u;
            for (int j = k + 1; j < PB; j++) {
                const float uj = lane_bcast(u, j);
                if (lane >= j) A[j * PLD + lane] = __builtin_fmaf(-su, uj, A[j * PLD + lane]);
            }
        }
    }
    __syncthreads();
    const unsigned long long neg = __ballot(A[lane * PLD + lane] < 0.f);
    potrf_inverse16(A, Tinv, wave, lane);
    __syncthreads();
    return neg;
}

// Diagnostic twin of potrf64_kernel: same work, s_memtime stamps after every phase (hooks build only).
#ifdef EKFVIO_TEST_HOOKS
template <int FV>
__global__ __launch_bounds__(256) void potrf64_stamp_kernel(const float* __restrict__ S, int lds, float* __restrict__ L,
                                                            int ldl, float* __restrict__ Linv, long long* stamps, int passes) {
    __shared__ __attribute__((aligned(16))) float A[PB * PLD];
    __shared__ float Tinv[INV_LDS];
    const int tid = threadIdx.x;
    // passes = 2 (EKFVIO_POTRF_WARM=1): the same factorisation twice, the stamps are the second pass's, which finds its
    // instructions in the instruction cache (is the once-through pivot chain waiting for its own code?)
#pragma unroll 1
    for (int pass = 0; pass < passes; pass++) {
        __syncthreads();
        if (tid == 0) stamps[0] = (long long)__builtin_amdgcn_s_memtime();
        load_tile(A, S, lds, tid);
        __syncthreads();
        potrf64_lds<FV>(A, Tinv, tid, stamps);
        store_tile_lower(A, L, ldl, tid);
        store_inv(Tinv, Linv, tid);
        __syncthreads();
        if (tid == 0) stamps[11] = (long long)__builtin_amdgcn_s_memtime();
    }
}
#endif  // EKFVIO_TEST_HOOKS

// Factor the first diagonal block (step "-1" of the sweep).
__global__ __launch_bounds__(256) void potrf64_kernel(const float* __restrict__ S, int lds, float* __restrict__ L,
                                                      int ldl, float* __restrict__ Linv, int* info, unsigned long long* Lsign) {
    __shared__ __attribute__((aligned(16))) float A[PB * PLD];
    __shared__ float Tinv[INV_LDS];
    const int tid = threadIdx.x;
    load_tile(A, S, lds, tid);
    __syncthreads();
    const bool bad = potrf64_lds(A, Tinv, tid);
    unsigned long long neg = 0ull;
    if (bad) {  // workgroup-uniform
        __syncthreads();
        load_tile(A, S, lds, tid);
        neg = potrf64_signed(A, Tinv, tid);
    }
    store_tile_lower(A, L, ldl, tid);
    store_inv(Tinv, Linv, tid);
    if (tid == 0) {
        Lsign[0] = neg;
        if (bad) atomicOr(info, 1);
    }
}

#include "gather_body.inc"

// The measurement gather and the factorisation of the first diagonal tile in ONE launch.  Workgroup 0 forms the first
// 64x64 tile of A = (H Sigma H^T + R)^T straight from Sigma (the same elements, through the same gather_a_elem, as the
// gather writes to Saug) and factors it; every other workgroup is a gather workgroup.  The pivot chain is
// latency-bound and runs 2x longer when its compute unit is shared, so the launch asks for more than half a compute
// unit's LDS per workgroup (EKF_GATHER_POTRF_LDS): one workgroup per compute unit, the chain has its own.
__global__ __launch_bounds__(256) void gather_potrf_kernel(GatherArgs ga, float* __restrict__ L, int ldl, float* __restrict__ Linv,
                                                           int* info, unsigned long long* Lsign, long long* dbg) {
    extern __shared__ __attribute__((aligned(16))) float dyn_lds[];
    const int tid = threadIdx.x;
#define GSTAMP_(slot)                                                                               \
    do {                                                                                            \
        if (dbg && tid == 0) dbg[1000 + (slot)] = (long long)__builtin_amdgcn_s_memtime();          \
    } while (0)
    if (blockIdx.x != 0) {
        // the flags of the persistent sweep behind this launch: zeroed here (a memset node between the two launches cost
        // 4.9 us of every filter step)
        if (blockIdx.x == gridDim.x - 1 && ga.zero_words)
            for (int e = tid; e < ga.n_zero; e += 256) ga.zero_words[e] = 0;
        gather_body(ga, (int)blockIdx.x - 1, dyn_lds);
        if (dbg && tid == 0 && blockIdx.x == gridDim.x - 1) dbg[1004] = (long long)__builtin_amdgcn_s_memtime();
        if (dbg && tid == 0 && blockIdx.x == 1) dbg[1005] = (long long)__builtin_amdgcn_s_memtime();
        return;
    }
    GSTAMP_(0);
    float* A = dyn_lds;
    float* Tinv = dyn_lds + PB * PLD;
    const int m = ga.m_dev ? *ga.m_dev : ga.m;
    // lanes along c: Sigma(idx[c], idx[r]) = P[idx[r]*ld + idx[c]] is (nearly) contiguous in c
    const int c = tid & 63;
    const int sc = (c < m) ? ga.idx[c] : 0;
    // two batches of loads (row indices and R entries, then the Sigma elements), each fully in flight before its first use:
    // clamped, unconditional addresses, the masks are applied to the values
    int ir[16];
    float v[16], rd[16], ro[16];
#pragma unroll
    for (int ps = 0; ps < 16; ps++) {
        const int rc = max(min((tid >> 6) + 4 * ps, m - 1), 0);  // m may be 0 when it is only known on the device
        ir[ps] = max(ga.idx[rc], 0);
        rd[ps] = ga.Rm[2 * rc];
        ro[ps] = ga.Rm[2 * rc + 1];
    }
#pragma unroll
    for (int ps = 0; ps < 16; ps++) asm volatile("" : "+v"(ir[ps]));
#pragma unroll
    for (int ps = 0; ps < 16; ps++) v[ps] = ga.P[(size_t)ir[ps] * ga.ld + sc];
#pragma unroll
    for (int ps = 0; ps < 16; ps++) asm volatile("" : "+v"(v[ps]));
#pragma unroll
    for (int ps = 0; ps < 16; ps++) {
        const int r = (tid >> 6) + 4 * ps;
        A[c * PLD + r] = gather_a_elem(v[ps], r, c, m, rd[ps], ro[ps]);
    }
    __syncthreads();
    GSTAMP_(1);
    const float* A_c = A;
    const float* Tinv_c = Tinv;
    auto idle = [&](auto pc, int wv) {  // finished columns and inverses are stored under the pivot chain (see chol_step_kernel)
        constexpr int p = decltype(pc)::value;
        if constexpr (p == 2) {
            if (wv >= 2) store_cols<0, 32, 128, true>(A_c, L, ldl, tid - 128);
        } else if constexpr (p == 3) {
            store_cols<32, 16, 192, true>(A_c, L, ldl, tid - 64);
            store_inv_blocks(Tinv_c, Linv, 0, tid - 64);
        }
    };
    const bool bad = potrf64_lds<EKF_POTRF_FV>(A, Tinv, tid, dbg ? dbg + 800 : nullptr, idle);
    unsigned long long neg = 0ull;
    if (bad) {  // workgroup-uniform, rare: the tile is gathered again (its loads were consumed) and factored as U S U^T
        __syncthreads();
#pragma unroll
        for (int ps = 0; ps < 16; ps++) {
            const int r = (tid >> 6) + 4 * ps;
            A[c * PLD + r] = gather_a_elem(ga.P[(size_t)ir[ps] * ga.ld + sc], r, c, m, rd[ps], ro[ps]);
        }
        neg = potrf64_signed(A, Tinv, tid);
        GSTAMP_(2);
        store_tile_lower(A, L, ldl, tid);
        store_inv(Tinv, Linv, tid);
    } else {
        GSTAMP_(2);
        store_cols<48, 16, 256, true>(A_c, L, ldl, tid);
        if (tid < 64) store_inv_blocks(Tinv_c, Linv, 3, tid);
    }
    if (tid == 0) {
        Lsign[0] = neg;
        if (bad) atomicOr(info, 1);
    }
    GSTAMP_(3);
#undef GSTAMP_
}

// Block step k of the right-looking sweep over the augmented matrix [A; X; I].
// grid.x = r(r+1)/2 triangular tiles of A (r = mb-1-k) + rb*r rectangular tiles of the extra
// row blocks (X: the forward substitution Y L^T = X rides along; I: yields L^-T).
// idb0 = first identity row block: its block row q is still zero left of column block q.
// SOLVE = true: every tile workgroup forms the two panel blocks it needs itself (one launch per block step: the
// latency-optimal form for few tiles).  SOLVE = false: the panel blocks L_ik were written by chol_panel_kernel(k)
// before (two launches per block step): with thousands of tiles per step (N = 1024: mb = 32) the per-tile panel
// solves are 60 % redundant work.
//
// Schur tiles (SchurArgs, SOLVE = true only).  Eliminating A from the symmetric matrix [[A, X^T, I], [X, Sigma, 0], [I, 0, 0]]
// leaves Sigma - X A^-1 X^T in the (X, X) block and -X A^-1 = -K in the (X, I) block: both are just more trailing
// tiles of this same sweep,
//     T2(a,b) -= Y_ak S_k Y_bk^T           nb(nb+1)/2 tiles (a >= b), the mirror image written from the same product,
//     K(a,c)  += Y_ak S_k Z_ck^T, c <= k   nb*(k+1) tiles (block (I_c, k) is zero before step c),
// with Y_ak = X_ak U_kk^-T and Z_ck = I_ck U_kk^-T formed by the tile itself like every other panel block.  They fill
// compute units the latency-bound chain leaves idle, and the gain GEMM and the first Joseph GEMM (20 us behind the
// sweep at N = 256) disappear; the last block step is then a launch of Schur tiles only.
// Which Joseph factor this is: X = Sigma H^T, so X A^-1 X^T = Sigma H^T K^T and T2 = Sigma (I - K H)^T -- the RIGHT factor
// applied, where the reference forms the left one first, (I - K H) Sigma (:594).  Sigma is symmetric only to rounding, so
// the two are not transposes of each other, and finishing T2 as if it were the left factor ((.)(I - K H)^T once more)
// never damps the rows of Sigma's antisymmetric part: the filter drifts apart within ~100 steps (measured).  The caller
// therefore applies the LEFT factor to T2: Sigma' = (I - K H) T2 + K R K^T = T2 + K (R K^T - H T2) (launch_update), which
// is the reference's (I - K H) Sigma (I - K H)^T + K R K^T term for term.
struct SchurArgs {
    float* P = nullptr;   // Sigma in, T out (in place), nb x nb tiles of 64, column-major, ldp
    int ldp = 0;
    float* K = nullptr;   // n_pad x m_pad, column-major, ldk; written, never read (first touch of a tile stores)
    int ldk = 0;
    int nb = 0;           // X row blocks = n_pad / 64
};
// SCHUR = false compiles every Schur-tile branch away: the default flow's kernel is the plain sweep step.
template <bool SOLVE, bool SCHUR = false>
__global__ __launch_bounds__(256) void chol_step_kernel(float* __restrict__ S, int lds, float* __restrict__ L, int ldl,
                                                        float* __restrict__ Linv, int k, int mb, int idb0, int* info,
                                                        unsigned long long* Lsign, long long* dbg, SchurArgs sc) {
    __shared__ __attribute__((aligned(16))) float Ti[PB * PLD];   // A_ik, then L_ik
    __shared__ __attribute__((aligned(16))) float Tj[PB * PLD];   // A_jk, then L_jk
    // L_kk, later the updated next diagonal tile.  Without the in-kernel panel solves (SOLVE = false) only the chain
    // workgroup needs it, and that one has i == j and leaves Tj unused: sharing its storage brings the kernel from 54
    // to 38 KB of LDS, four resident workgroups per compute unit instead of two for the thousands of tiles of a step.
    __shared__ __attribute__((aligned(16))) float Tl_own[SOLVE ? PB * PLD : 1];
    float* Tl = SOLVE ? Tl_own : Tj;
    __shared__ float Tinv[INV_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave & 1, wc = wave >> 1;
    const int rr = mb - 1 - k, ntri = rr * (rr + 1) / 2;
    const int rb = idb0;  // extra row blocks: idb0 - mb of X, then mb of I
    // Tile 0 is the chain (next diagonal tile: latency-bound, wants a compute unit of its own).  Workgroups beyond the
    // 256th double up on the compute units of the first ones, so in a larger grid the chain trades places with workgroup
    // 255, whose compute unit is the last to receive a second resident.
    int t = blockIdx.x;
    if (SCHUR && rr > 0 && gridDim.x > 256) {
        if (t == 0) t = 255;
        else if (t == 255) t = 0;
    }
    const bool chain = rr > 0 && t == 0;
    // diagnostic phase stamps of the chain workgroup (scripts/chol_step_stamps.py); dbg is null in production
#define CSTAMP(slot)                                                                                         \
    do {                                                                                                     \
        if (dbg && chain && threadIdx.x == 0) dbg[32 + 8 * k + (slot)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
    CSTAMP(0);
    int i, j;
    int kind = 0;  // 0: tile of the augmented matrix, 1: T tile, 2: K tile
    if (t < ntri) {
        // lower-triangular tile index: t = ii(ii+1)/2 + jj, 0 <= jj <= ii
        int ii = 0;
        while ((ii + 1) * (ii + 2) / 2 <= t) ii++;
        i = k + 1 + ii;
        j = k + 1 + (t - ii * (ii + 1) / 2);
    } else if (!SCHUR || t < ntri + rb * rr) {
        t -= ntri;
        i = mb + t / rr;
        j = k + 1 + t % rr;
        if (i >= idb0 && i - idb0 > k) return;  // identity block row: block (i,k) is still zero
    } else {
        // Schur tiles: both operands are panel blocks of extra row blocks
        int u = t - ntri - rb * rr;
        const int nT = sc.nb * (sc.nb + 1) / 2;
        if (u < nT) {
            int aa = 0;
            while ((aa + 1) * (aa + 2) / 2 <= u) aa++;
            i = mb + aa;
            j = mb + (u - aa * (aa + 1) / 2);
            kind = 1;
        } else {
            u -= nT;
            i = mb + u % sc.nb;
            j = idb0 + u / sc.nb;  // identity row block c = u / nb <= k
            kind = 2;
        }
    }

    // negative pivots of block column k (zero unless diagonal tile k went through the U S U^T path): a scalar load
    unsigned long long neg = sign_mask_request(Lsign + k);
    if (SOLVE) {
        load_tile(Tl, L + (size_t)k * PB * ldl + (size_t)k * PB, ldl, tid);
        load_inv(Tinv, Linv + (size_t)k * PB * PB, tid);
        load_tile(Ti, S + (size_t)k * PB * lds + (size_t)i * PB, lds, tid);
        if (i != j) load_tile(Tj, S + (size_t)k * PB * lds + (size_t)j * PB, lds, tid);
        __syncthreads();
        neg = sign_mask_value(neg);
#ifdef EKF_NO_SIGNED  // diagnostic build: prices the hooks of the rare path (scripts/ab_lib.py)
        neg = 0ull;
#endif
        CSTAMP(1);
        if (i != j)
            tri_solve_fwd2(Ti, Tj, Tl, Tinv, wave, lane);  // L_ik = A_ik L_kk^-T, L_jk
        else
            tri_solve_fwd(Ti, Tl, Tinv, wave, lane, (dbg && chain) ? dbg + 960 + 4 * k : nullptr);
        __syncthreads();
        CSTAMP(2);
        // the panel blocks of the extra rows are results of their own only without Schur tiles (the gain GEMM reads them)
        // (the chain workgroup stores its panel block later, from wavefronts that idle during the factorisation)
        if (kind == 0 && j == k + 1 && (i < mb || !SCHUR) && !chain) {
            float* dst = L + (size_t)k * PB * ldl + (size_t)i * PB;
            if (neg != 0ull && i >= idb0) store_tile_signed(Ti, dst, ldl, tid, neg);  // identity rows carry S into the gain
            else store_tile(Ti, dst, ldl, tid);
        }
    } else {
        load_tile(Ti, L + (size_t)k * PB * ldl + (size_t)i * PB, ldl, tid);
        if (i != j) load_tile(Tj, L + (size_t)k * PB * ldl + (size_t)j * PB, ldl, tid);
        __syncthreads();
        neg = sign_mask_value(neg);
    }
    // A_ij(r,s) -= sum_c L_ik(r,c) L_jk(s,c)
    const float* Bj = (i != j) ? Tj : Ti;
    // where the tile's target lives: the augmented matrix itself, Sigma (T tiles) or the gain (K tiles)
    float* Sij;
    int ldt;
    if (!SCHUR || kind == 0) {
        Sij = S + (size_t)j * PB * lds + (size_t)i * PB;
        ldt = lds;
    } else if (kind == 1) {
        Sij = sc.P + (size_t)(j - mb) * PB * sc.ldp + (size_t)(i - mb) * PB;
        ldt = sc.ldp;
    } else {
        Sij = sc.K + (size_t)(j - idb0) * PB * sc.ldk + (size_t)(i - mb) * PB;
        ldt = sc.ldk;
    }
    const int r = wr * 32 + (lane & 31);
    // The chain workgroup requests its target tile BEFORE the 64^3 product and uses it AFTER: the memory round trip runs
    // under the MFMA stream.  Plain loads: the compiler keeps their wait in front of the first use as long as nothing
    // between forces it earlier -- a loop does (it waits for every outstanding load before entering one), which is why
    // mma64 is unrolled completely.  (An earlier form issued the loads from inline asm and placed the wait by hand;
    // the compiler does not know such a register is still in flight and may copy it: scripts/handles_stress.py.)
    float tgt[16];
    if (chain) {
        // scalar base + 32-bit lane offset + a scalar step per element: one VALU add per load instead of a 64-bit
        // multiply-add chain (the tile spans at most 64 * lds floats: the offsets fit 32 bits)
        const unsigned off0 = ((unsigned)(wc * 32 + 4 * (lane >> 5)) * (unsigned)lds + (unsigned)r) * 4u;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const unsigned off = off0 + (unsigned)((q & 3) + 8 * (q >> 2)) * (unsigned)lds * 4u;
            tgt[q] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(Sij) + off);
        }
    }
    CSTAMP(7);
    f32x16 up;
    if (neg == 0ull) {
        up = mma64(Ti, 1, PLD, Bj, 1, PLD, wr, wc, lane);
    } else {
        up = mma64_signed(Ti, 1, PLD, Bj, 1, PLD, wr, wc, lane, neg);
    }
    CSTAMP(6);
    if (chain) {
        // next diagonal tile: update into LDS and factor it now (look-ahead)
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int c = wc * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
            Tl[c * PLD + r] = tgt[q] - up[q];
        }
        __syncthreads();
        CSTAMP(3);
        // Stores ride under the pivot chain: columns 0-31 of the factor by waves 2-3 in panel 2's phase, columns 32-47,
        // the inverses of blocks 0-2 and this workgroup's panel block L_ik (SOLVE) by waves 1-3 in panel 3's; only the
        // last 16 columns and the last inverse are left for the end.
        float* Ldst = L + (size_t)j * PB * ldl + (size_t)i * PB;
        float* Idst = Linv + (size_t)(k + 1) * PB * PB;
        float* Pdst = L + (size_t)k * PB * ldl + (size_t)i * PB;
        const float* Tl_c = Tl;
        const float* Ti_c = Ti;
        const float* Tinv_c = Tinv;
        auto idle = [&](auto pc, int wv) {
            constexpr int p = decltype(pc)::value;
            if constexpr (p == 2) {
                if (wv >= 2) store_cols<0, 32, 128, true>(Tl_c, Ldst, ldl, tid - 128);
            } else if constexpr (p == 3) {
                store_cols<32, 16, 192, true>(Tl_c, Ldst, ldl, tid - 64);
                store_inv_blocks(Tinv_c, Idst, 0, tid - 64);
                if (SOLVE && (i < mb || !SCHUR)) store_cols<0, 64, 192, false>(Ti_c, Pdst, ldl, tid - 64);
            }
        };
        const bool bad = potrf64_lds<EKF_POTRF_FV>(Tl, Tinv, tid, nullptr, idle);
        unsigned long long neg1 = 0ull;
        if (bad) {  // workgroup-uniform, rare: the updated tile is formed again and factored as U S U^T
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int c = wc * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
                Tl[c * PLD + r] = Sij[(size_t)c * lds + r] - up[q];
            }
            neg1 = potrf64_signed(Tl, Tinv, tid);
            CSTAMP(4);
            store_tile_lower(Tl, Ldst, ldl, tid);
            store_inv(Tinv, Idst, tid);
        } else {
            CSTAMP(4);
            store_cols<48, 16, 256, true>(Tl_c, Ldst, ldl, tid);
            if (tid < 64) store_inv_blocks(Tinv_c, Idst, 3, tid);
        }
        if (tid == 0) {
            Lsign[k + 1] = neg1;
            if (bad) atomicOr(info, 1);
        }
        CSTAMP(5);
    } else if (SCHUR && kind == 2) {
        // K(a,c) = sum_k Y_ak S_k Z_ck^T: the first step that reaches identity block row c stores, later ones add
        const bool first = (j - idb0) == k;
        float tv[16];
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int c = wc * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
            tv[q] = first ? 0.f : Sij[(size_t)c * ldt + r];
        }
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int c = wc * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
            Sij[(size_t)c * ldt + r] = tv[q] + up[q];
        }
    } else {
        // read all sixteen targets, then write them: written as sixteen `-=` the compiler cannot rule out that a store
        // aliases the next load (the stride is a run-time value) and serialises sixteen memory round trips
        float tv[16];
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int c = wc * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
            tv[q] = Sij[(size_t)c * ldt + r];
        }
        float tm[16];
        if (SCHUR && kind == 1 && i != j) {
            // the mirror image T(b,a) -= (Y_ak S Y_bk^T)^T from the same product: transposed through LDS (L_kk's tile is
            // free: every wavefront is past its substitution), then read back with the lanes along the mirror tile's rows
            float* Mji = sc.P + (size_t)(i - mb) * PB * sc.ldp + (size_t)(j - mb) * PB;
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int c = wc * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
                tm[q] = Mji[(size_t)c * ldt + r];
                Tl[r * PLD + c] = up[q];
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int c = wc * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
                Mji[(size_t)c * ldt + r] = tm[q] - Tl[c * PLD + r];
            }
        }
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int c = wc * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
            Sij[(size_t)c * ldt + r] = tv[q] - up[q];
        }
    }
}

// Block column k of the panel: L_ik = A_ik L_kk^-T for the row blocks below the diagonal (A rows k+1 .. mb-1, `na` of
// them; na = 0 for the last block column) and the extra row blocks (X, I).
// sign_irows: the panel blocks of the identity rows are stored with the block column's signs applied (what the gain GEMM
// reads); off in the split sweep, whose update launches read the stored blocks back as operands (launch_chol_sweep).
__global__ __launch_bounds__(256) void chol_panel_kernel(const float* __restrict__ S, int lds, float* __restrict__ L,
                                                         int ldl, const float* __restrict__ Linv, int k, int mb, int na, int idb0,
                                                         const unsigned long long* __restrict__ Lsign, int sign_irows) {
    __shared__ __attribute__((aligned(16))) float Ti[PB * PLD];
    __shared__ __attribute__((aligned(16))) float Tl[PB * PLD];
    __shared__ float Tinv[INV_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const int i = (b < na) ? k + 1 + b : mb + (b - na);
    if (i >= idb0 && i - idb0 > k) return;  // identity block row: block (i,k) is still zero and stays unused
    unsigned long long neg = sign_mask_request(Lsign + k);
    load_tile(Tl, L + (size_t)k * PB * ldl + (size_t)k * PB, ldl, tid);
    load_inv(Tinv, Linv + (size_t)k * PB * PB, tid);
    load_tile(Ti, S + (size_t)k * PB * lds + (size_t)i * PB, lds, tid);
    __syncthreads();
    neg = sign_mask_value(neg);
    tri_solve_fwd(Ti, Tl, Tinv, wave, lane);
    __syncthreads();
    if (sign_irows && neg != 0ull && i >= idb0) store_tile_signed(Ti, L + (size_t)k * PB * ldl + (size_t)i * PB, ldl, tid, neg);
    else store_tile(Ti, L + (size_t)k * PB * ldl + (size_t)i * PB, ldl, tid);
}

// Split sweep only: its panel blocks are stored unsigned (they are read back as operands), so the column signs of the
// identity rows' blocks (L^-T, what the gain GEMM multiplies by) are applied afterwards, one block column per
// blockIdx.y; a block column without negative pivots returns at once.
__global__ __launch_bounds__(256) void sign_irows_kernel(float* __restrict__ L, int ldl, int row0, int rows,
                                                         const unsigned long long* __restrict__ Lsign) {
    const unsigned long long neg = Lsign[blockIdx.y];
    if (neg == 0ull) return;
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    for (int c = 0; c < PB; c++)
        if ((neg >> c) & 1ull) {
            float* p = L + (size_t)(blockIdx.y * PB + c) * ldl + row0 + r;
            *p = -*p;
        }
}


// Every wait inside the persistent sweep is bounded (chol_persist.inc, persist_poll): by wall-clock time, PersistArgs::wait_ticks of the
// 100 MHz s_memrealtime counter = 3 ms per handle by default (a filter step is 0.1 ms, the longest legitimate wait ~10 us), and by a
// number of looks that only the fault-injection hook lowers into reach.
#define SWEEP_SPIN_LIMIT (1 << 30)
#define SWEEP_WAIT_TICKS_UNRECOVERABLE 10000000  // 100 ms: where nothing can run the update again (ekfvio_run_uploaded), PersistArgs::wait_ticks

#include "chol_persist.inc"
#include "chol_step_la.inc"

}  // namespace

#ifdef EKFVIO_TEST_HOOKS
void launch_potrf_stamps(ekfvio_filter* f, const float* S, int ld, float* L, float* Linv, long long* d_stamps) {
    const char* e = getenv("EKFVIO_POTRF_FV");  // diagnostic: which factor-phase variant to time
    const int fv = e ? atoi(e) : EKF_POTRF_FV;
    auto kern = fv == 0 ? potrf64_stamp_kernel<0> : fv == 10 ? potrf64_stamp_kernel<10> : fv == 12 ? potrf64_stamp_kernel<12>
              : fv == 13 ? potrf64_stamp_kernel<13> : potrf64_stamp_kernel<8>;
    const char* w = getenv("EKFVIO_POTRF_WARM");
    hipLaunchKernelGGL(kern, dim3(1), dim3(256), 0, f->stream, S, ld, L, ld, Linv, d_stamps, (w && atoi(w)) ? 2 : 1);
}
#endif  // EKFVIO_TEST_HOOKS

// layout: ready[mb], fin[rows x mb], pan[rows x mb] (panel blocks of the X / identity rows, for the gain tiles formed inside the
// launch), one spare word, the abort word (last: whoever zeroes the flags for the next sweep leaves it alone, persist_zero_words);
// a multiple of 16 bytes
static size_t persist_flag_words(int m_pad, int n_pad) {
    const int mb = m_pad / PB, rows = 2 * mb + n_pad / PB;
    return ((size_t)(mb + 2 * rows * mb + 2) + 3) & ~(size_t)3;
}
int persist_zero_words(int m_pad, int n_pad) {
    const int mb = m_pad / PB, rows = 2 * mb + n_pad / PB;
    return mb + 2 * rows * mb + 1;
}
static void persist_flag_pointers(ekfvio_filter* f, PersistArgs& pa, int mb, int rows) {
    pa.ready = f->sweep_sync;
    pa.fin = f->sweep_sync + mb;
    pa.pan = pa.fin + rows * mb;
    pa.abort_flag = pa.pan + rows * mb + 1;
}
#define EKF_GATHER_POTRF_LDS (84 * 1024)  // > half of a compute unit's 160 KB: one workgroup per compute unit
void launch_gather_potrf(ekfvio_filter* f, int m, int m_pad, int n_pad, bool m_on_device, bool with_wt) {
    if (!f->gather_attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gather_potrf_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  EKF_GATHER_POTRF_LDS);
        f->gather_attr_set = true;
    }
    GatherArgs ga = make_gather_args(f, m, m_pad, n_pad);
    if (m_on_device) ga.m_dev = f->info + 2;
    if (sweep_is_persistent(f, m_pad, n_pad)) {
        ga.zero_words = f->sweep_sync;
        ga.n_zero = (int)persist_flag_words(m_pad, n_pad);
        f->sweep_flags_clean = true;  // (launch_chol_sweep, called next, skips its memset)
    }
    const int nb2 = with_wt ? (m_pad / 64) * (f->ldp / 64) : 0;  // 64x64 transposing tiles of Wt (only the first Joseph GEMM reads it)
    hipLaunchKernelGGL(gather_potrf_kernel, dim3(1 + ga.nb1 + nb2), dim3(256), EKF_GATHER_POTRF_LDS, f->stream, ga, f->Laug, f->ld_aug,
                       f->Linv, f->info, f->Lsign, f->sweep_dbg);
}

// helpers of the persistent sweep (chol_persist.inc): per block column j the rows of A from the diagonal down (without
// (1,1)), the X row blocks and the identity row blocks 0 .. j
static int persist_helpers(int mb, int nX, bool compact = false) {  // (compact: without the identity rows' diagonal tiles, chol_persist.inc)
    int h = 0;
    for (int j = 1; j < mb; j++) h += (mb - j - (j == 1 ? 1 : 0)) + nX + (j + 1 - (compact ? 1 : 0));
    return h;
}
// The shapes the persistent launch takes: 3 .. EKF_SWEEP_SPLIT_MB - 1 block columns.  Up to round 3: chain + owners co-resident
// (1 + tiles <= compute units).  Round 4: up to EKF_PERSIST_OVERSUB x that.  An owner waits only for workgroups with a LOWER block index
// (the chain is workgroup 0, the owners are numbered block column by block column, and a tile's panel sources lie in earlier columns), so
// with workgroups dispatched in index order -- what the hardware does, though HIP does not promise it -- the resident ones can always
// finish, and the later columns' owners take over their compute units and catch up (their flags are all up: ~10 k cycles per step
// against the chain's ~15.5 k).  N = 400 (params/fast_with_insight.yaml: 13 block columns, 407 owners) runs the sweep in one launch
// this way.  Should the order ever not hold, the bounded waits end the launch and the update is run again per step (EKFVIO_EABORTED).
// Measured (profiles/r04_oversubscribed_persistent_sweep.txt): bit-identical, the sweep itself 155.7 -> 121.2 us at N = 400, but the gain
// kernel behind it then runs 2x longer on operands the persistent launch has left in other XCDs' L2s; +2.9 % per step in all.  Off by
// default (1); EKFVIO_PERSIST_OVERSUB=2 turns it on, tests/test_gpu_parity.py exercises it.
#ifndef EKF_PERSIST_OVERSUB
#define EKF_PERSIST_OVERSUB 1
#endif
static bool persist_shape(const ekfvio_filter* f, int m_pad, int n_pad) {
    const int mb = m_pad / PB;
    const int over = f->persist_oversub > 0 ? f->persist_oversub : EKF_PERSIST_OVERSUB;
    return mb >= 3 && mb < EKF_SWEEP_SPLIT_MB && 1 + persist_helpers(mb, n_pad / PB) <= (over > 1 ? over : 1) * f->num_cus &&
           persist_flag_words(m_pad, n_pad) <= f->sweep_sync_words;
}
// ... and of those, the shapes whose gain is formed inside the fused launch: (nearly) every workgroup of that launch finds a compute unit
// at once (the transposing / gain workgroups hold theirs to the end).  Behind every other sweep of such a shape gain_tiles_kernel runs
// the same arithmetic; all other shapes take the gain GEMM, whichever sweep ran.
static bool gain_in_sweep_shape(const ekfvio_filter* f, int m_pad, int n_pad) {
    const int mb = m_pad / PB;
    return persist_shape(f, m_pad, n_pad) && 1 + mb * (f->ldp / 64) + 2 + persist_helpers(mb, n_pad / PB) <= f->num_cus + 8;
}
// ... and of those, the shapes whose T2 = Sigma (I - K H)^T is formed by freed owners inside the launch (round 6, chol_persist.inc, t2_tile): there
// must be an owner per tile pair once enough of the first owners have LEFT for every workgroup of the launch to find a compute unit (the two
// step-0 gatherers and the gain workgroups beyond the state's row blocks leave at once).  Returns the number of owners that leave (PersistArgs::t2_skip),
// -1 where the flow does not apply.  The flow is a property of the SHAPE, not of the sweep that runs: behind a per-step sweep of such a shape
// gain2_t2_tiles_kernel forms the same T2, so a sequence's bits do not depend on what else the device runs.
static int t2_skip_owners(const ekfvio_filter* f, int m_pad, int n_pad) {
    if (!f->t2_flow || !f->persist_gain || f->schur || !gain_in_sweep_shape(f, m_pad, n_pad)) return -1;
    const int mb = m_pad / PB, nX = n_pad / PB;
    const int H = persist_helpers(mb, nX, true), gw = mb * (f->ldp / 64);
    if (1 + gw + H > f->num_cus) return -1;  // the compact launch (chol_persist.inc): every workgroup has its compute unit from the start
    return (nX * (nX + 1) / 2 <= H) ? 0 : -1;
}
bool t2_flow_shape(const ekfvio_filter* f, int m_pad, int n_pad) { return t2_skip_owners(f, m_pad, n_pad) >= 0; }
bool sweep_is_persistent(const ekfvio_filter* f, int m_pad, int n_pad) {
    // (only for a device's sole handle: two persistent launches in flight together could starve each other of compute units)
    return f->sweep_mode == 2 && !sweep_supports_schur(f, m_pad) && live_handles_on(f->device) <= 1 && persist_shape(f, m_pad, n_pad);
}

// The update's whole front in ONE launch (round 4): the measurement gather, the first diagonal tile and the sweep behind it.
// Nobody writes the whole augmented matrix [A; X; I] any more: workgroup 0 gathers tile (0,0) straight from Sigma, factors it, keeps
// it in LDS and goes on as the chain; every owner gathers ITS tile and the two block-column-0 panel sources of its first step itself
// (chol_persist.inc, GTile); two workgroups gather the chain's first two tiles, (1,0) and (1,1), and hand them over like finished
// tiles; workgroups 1 .. G transpose (H Sigma)^T into Wt for the first Joseph GEMM and wait for nothing.  Against gather_potrf_kernel + chol_persist_kernel: no kernel boundary on the chain, no store and cold reload of
// L_00, no 5.5 MB of Saug written and read back.  A few more workgroups than compute units at N = 256 (260): the transposing ones
// come first and never wait, the last owners get their compute unit microseconds later and are needed last.  The launch asks for
// > 80 KB of LDS so that every workgroup has a compute unit of its own (the pivot chain runs 2x longer on a shared one).
#define EKF_PERSIST_FUSED_DYN_LDS (28 * 1024)
void launch_persist_fused(ekfvio_filter* f, int m, int m_pad, int n_pad, bool m_on_device) {
    ProfScope ps(f, PC_CHOL, (double)m_pad * m_pad * m_pad / 3.0 + (double)(n_pad + m_pad / 2) * m_pad * m_pad);
    if (!f->persist_attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(chol_persist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  EKF_PERSIST_FUSED_DYN_LDS);
        f->persist_attr_set = true;
    }
    GatherArgs ga = make_gather_args(f, m, m_pad, n_pad);
    if (m_on_device) ga.m_dev = f->info + 2;
    const int nb2 = (m_pad / 64) * (f->ldp / 64);  // 64x64 transposing tiles of Wt
    const int mb = m_pad / PB, rb = n_pad / PB + mb, ld = f->ld_aug;
    PersistArgs pa;
    pa.S = f->Saug, pa.lds = ld, pa.L = f->Laug, pa.ldl = ld, pa.Linv = f->Linv;
    pa.mb = mb, pa.idb0 = mb + n_pad / PB, pa.nrows = mb + rb;
    pa.info = f->info, pa.Lsign = f->Lsign;
    persist_flag_pointers(f, pa, mb, mb + rb);
    pa.fused = 1, pa.gather_wgs = nb2;
    // the transposing workgroups stay and form the gain while the sweep runs, as long as (nearly) every workgroup of the launch finds a
    // compute unit at once: they hold theirs to the end
    pa.gain = (f->persist_gain && gain_in_sweep_shape(f, m_pad, n_pad)) ? 1 : 0;
    pa.K = f->Km, pa.ldk = f->ldp;
    f->gain_in_sweep = pa.gain != 0;
    pa.t2_skip = t2_skip_owners(f, m_pad, n_pad);
    pa.t2 = (pa.gain && pa.t2_skip >= 0) ? 1 : 0;
    pa.compact = pa.t2;
    pa.Sg = f->P, pa.ldsg = f->ldp, pa.T2 = t2_buffer(f), pa.ldt = f->ldp, pa.nstate = f->n;
    f->t2_in_sweep = pa.t2 != 0;
    pa.dbg = f->sweep_dbg;
    pa.bound.spin_limit = f->sweep_spin_limit > 0 ? f->sweep_spin_limit : SWEEP_SPIN_LIMIT;
    pa.bound.wait_ticks = f->sweep_unrecoverable ? std::max(f->sweep_wait_ticks, SWEEP_WAIT_TICKS_UNRECOVERABLE) : f->sweep_wait_ticks;
    pa.early_sources = f->persist_early;
    pa.stall_wg = f->sweep_stall_wg >= 0 ? f->sweep_stall_wg + pa.gather_wgs + (pa.compact ? 0 : 2) : -1;  // (the hook counts owners from workgroup 1)
    f->sweep_abort_word = pa.abort_flag;
    if (!f->sweep_flags_clean) (void)hipMemsetAsync(f->sweep_sync, 0, sizeof(int) * persist_flag_words(m_pad, n_pad), f->stream);
    f->sweep_flags_clean = false;
    hipLaunchKernelGGL(chol_persist_kernel, dim3(1 + pa.gather_wgs + (pa.compact ? 0 : 2) + persist_helpers(mb, n_pad / PB, pa.compact != 0)), dim3(256), EKF_PERSIST_FUSED_DYN_LDS,
                       f->stream, pa, ga);
    f->persistent_sweeps++;
}

void launch_chol_sweep(ekfvio_filter* f, float* Saug, float* Laug, float* Linv, int m_pad, int n_pad, int ld, bool first_tile_done,
                       bool schur) {
    ProfScope ps(f, PC_CHOL, (double)m_pad * m_pad * m_pad / 3.0 + (double)(n_pad + m_pad / 2) * m_pad * m_pad +
                                 (schur ? (double)n_pad * n_pad * m_pad + (double)n_pad * m_pad * m_pad : 0.0));
    const int mb = m_pad / PB;
    const int rb = n_pad / PB + mb;        // extra row blocks: X then I
    const int idb0 = mb + n_pad / PB;
    f->sweep_abort_word = nullptr;
    f->gain_in_sweep = false;
    f->t2_in_sweep = false;
    if (!first_tile_done)
        hipLaunchKernelGGL(potrf64_kernel, dim3(1), dim3(256), 0, f->stream, Saug, ld, Laug, ld, Linv, f->info, f->Lsign);
    if (!schur && sweep_is_persistent(f, m_pad, n_pad)) {
        // ONE launch for everything behind the first diagonal tile (chol_persist.inc): flags zeroed, then the chain and its helpers
        PersistArgs pa;
        pa.S = Saug, pa.lds = ld, pa.L = Laug, pa.ldl = ld, pa.Linv = Linv;
        pa.mb = mb, pa.idb0 = idb0, pa.nrows = mb + rb;
        pa.info = f->info, pa.Lsign = f->Lsign;
        persist_flag_pointers(f, pa, mb, mb + rb);
        pa.fused = 0, pa.gather_wgs = 0, pa.gain = 0, pa.K = nullptr, pa.ldk = 0, pa.t2 = 0, pa.compact = 0;
        pa.dbg = f->sweep_dbg;
        pa.bound.spin_limit = f->sweep_spin_limit > 0 ? f->sweep_spin_limit : SWEEP_SPIN_LIMIT;
        pa.bound.wait_ticks = f->sweep_unrecoverable ? std::max(f->sweep_wait_ticks, SWEEP_WAIT_TICKS_UNRECOVERABLE) : f->sweep_wait_ticks;
        pa.early_sources = f->persist_early;
        pa.stall_wg = f->sweep_stall_wg;
        f->sweep_abort_word = pa.abort_flag;  // the kernels behind this sweep leave the state alone if it is raised (launch_update)
        // (the flags are zero already behind an update of this handle: its last GEMM zeroes them, launch_update)
        if (!f->sweep_flags_clean) (void)hipMemsetAsync(f->sweep_sync, 0, sizeof(int) * persist_flag_words(m_pad, n_pad), f->stream);
        f->sweep_flags_clean = false;
        hipLaunchKernelGGL(chol_persist_kernel, dim3(1 + persist_helpers(mb, n_pad / PB)), dim3(256), 0, f->stream, pa, GatherArgs());
        f->persistent_sweeps++;
        return;
    }
    // many tiles per step: panel blocks once per step in a launch of their own instead of twice per tile
    const bool split = mb >= EKF_SWEEP_SPLIT_MB;
    SchurArgs sc;
    if (schur && !split) {
        f->schur_sweeps++;
        // T and K as Schur tiles of the sweep itself (chol_step_kernel): Sigma is updated in place, the gain lands in Km
        sc.P = f->P;
        sc.ldp = f->ldp;
        sc.K = f->Km;
        sc.ldk = f->ldp;
        sc.nb = n_pad / PB;
        const int nT = sc.nb * (sc.nb + 1) / 2;
        for (int k = 0; k < mb; k++) {
            const int r = mb - 1 - k;
            const dim3 grid(r * (r + 1) / 2 + rb * r + nT + sc.nb * (k + 1));
            hipLaunchKernelGGL((chol_step_kernel<true, true>), grid, dim3(256), 0, f->stream, Saug, ld, Laug, ld, Linv, k, mb, idb0,
                               f->info, f->Lsign, f->sweep_dbg, sc);
        }
        return;
    }
    // EKFVIO_SWEEP_LA=0 (diagnostic): the two-launch split sweep (panel launch + tile launch per block step)
    static const bool la_env = getenv("EKFVIO_SWEEP_LA") ? atoi(getenv("EKFVIO_SWEEP_LA")) != 0 : true;
    if (split && la_env) {
        LaArgs la;
        la.S = Saug, la.lds = ld, la.L = Laug, la.ldl = ld, la.Linv = Linv;
        la.mb = mb, la.rb = rb, la.idb0 = idb0, la.info = f->info, la.Lsign = f->Lsign;
        {
            for (int l = 0; l + 1 < mb; l++) {
                const int grid = la_near_tasks(mb, idb0, rb, l) + la_far_tasks(mb, idb0, rb, l);  // (chol_step_la.inc)
                hipLaunchKernelGGL(chol_step_la_kernel, dim3(grid), dim3(256), 0, f->stream, la, l);
            }
        }
        hipLaunchKernelGGL(chol_panel_kernel, dim3(rb), dim3(256), 0, f->stream, Saug, ld, Laug, ld, Linv, mb - 1, mb, 0, idb0, f->Lsign, 0);
        hipLaunchKernelGGL(sign_irows_kernel, dim3((m_pad + 255) / 256, mb), dim3(256), 0, f->stream, Laug, ld, idb0 * PB, m_pad,
                           f->Lsign);
        return;
    }
    for (int k = 0; k + 1 < mb; k++) {
        const int r = mb - 1 - k;
        const dim3 grid(r * (r + 1) / 2 + rb * r);
        if (split) {
            hipLaunchKernelGGL(chol_panel_kernel, dim3(r + rb), dim3(256), 0, f->stream, Saug, ld, Laug, ld, Linv, k, mb, r, idb0,
                               f->Lsign, 0);
            hipLaunchKernelGGL(chol_step_kernel<false>, grid, dim3(256), 0, f->stream, Saug, ld, Laug, ld, Linv, k, mb, idb0, f->info,
                               f->Lsign, f->sweep_dbg, sc);
        } else {
            hipLaunchKernelGGL(chol_step_kernel<true>, grid, dim3(256), 0, f->stream, Saug, ld, Laug, ld, Linv, k, mb, idb0, f->info,
                               f->Lsign, f->sweep_dbg, sc);
        }
    }
    hipLaunchKernelGGL(chol_panel_kernel, dim3(rb), dim3(256), 0, f->stream, Saug, ld, Laug, ld, Linv, mb - 1, mb, 0, idb0, f->Lsign,
                       split ? 0 : 1);
    if (split)  // the stored identity-row blocks were operands until now: their column signs go on last
        hipLaunchKernelGGL(sign_irows_kernel, dim3((m_pad + 255) / 256, mb), dim3(256), 0, f->stream, Laug, ld, idb0 * PB, m_pad,
                           f->Lsign);
}

// Behind the Schur sweep: the gain gets the reference's sparseView prune (:580), G' = K R^T - (H T2)^T, i.e.
// G'(i,q) = (K R^T)(i,q) - T2(idx[q], i) (the factor of the second Joseph term, Sigma' = T2 + K G'^T, see launch_update),
// and K y.  One workgroup per 64 x 64 tile of K.  The rows idx[q] of T2 are read with the lanes along q (consecutive
// measurement rows are 1-2 floats apart in a column of T2: coalesced) and turned through LDS; everything else has the
// lanes along the state index.  K y is summed per tile (four column phases, then the phases in order) into
// Kyp[column block][state index]; the second Joseph GEMM's first workgroup adds the column blocks in order: one fixed
// summation order, same bits every run.
__global__ __launch_bounds__(256) void joseph_g_kernel(float* __restrict__ K, int ldk, const float* __restrict__ T, int ldp,
                                                       const int* __restrict__ idx, const float* __restrict__ Rm,
                                                       const float* __restrict__ zrow, const float* __restrict__ mu, int m,
                                                       const int* __restrict__ m_dev, float* __restrict__ G, int ldg,
                                                       float* __restrict__ Kyp) {
    __shared__ float s_t[64 * 65];
    __shared__ float s_part[4][64];
    if (m_dev) m = *m_dev;
    const int rl = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int i0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int i = i0 + rl;
    // the gain's loads go first (they do not depend on anything)
    float kq[16], kp[16];
#pragma unroll
    for (int u = 0; u < 16; u++) {
        const int q = c0 + ph + 4 * u;
        kq[u] = K[(size_t)q * ldk + i];
        kp[u] = K[(size_t)(q ^ 1) * ldk + i];
    }
    {   // T2(idx[q], i0 + ii): lanes along q
        const int q = c0 + rl;
        const int sq = (q < m) ? idx[q] : 0;
        float tv[16];
#pragma unroll
        for (int u = 0; u < 16; u++) tv[u] = T[(size_t)(i0 + ph + 4 * u) * ldp + sq];
#pragma unroll
        for (int u = 0; u < 16; u++) s_t[(ph + 4 * u) * 65 + rl] = tv[u];
    }
    __syncthreads();
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < 16; u++) {
        const int q = c0 + ph + 4 * u;
        const float a = (fabsf(kq[u]) > EKF_FLUSH_THRESH) ? kq[u] : 0.f;   // .sparseView(SPARSE_THRESH, SPARSE_EPS)
        const float b = (fabsf(kp[u]) > EKF_FLUSH_THRESH) ? kp[u] : 0.f;   // the partner column, pruned the same way
        K[(size_t)q * ldk + i] = a;
        float g = 0.f;
        if (q < m) {
            // (K R^T)(i,q) = K(i,q) R(q,q) + K(i,q^1) R(q,q^1)
            const float x = a * Rm[2 * q], y = b * Rm[2 * (q ^ 1) + 1];
            const float kr = ((q ^ 1) < q) ? (y + x) : (x + y);            // ascending measurement index
            g = kr - s_t[rl * 65 + ph + 4 * u];
            acc = acc + a * (zrow[q] - mu[idx[q]]);                         // K (z - H mu), :554-555, :600
        }
        G[(size_t)q * ldg + i] = g;
    }
    s_part[ph][rl] = acc;
    __syncthreads();
    if (ph == 0) Kyp[(size_t)blockIdx.y * ldg + i] = ((s_part[0][rl] + s_part[1][rl]) + s_part[2][rl]) + s_part[3][rl];
}

// The gain, G', K y's partial sums AND T2 behind a sweep that did not form them itself (gain2_t2_tiles_kernel; T2 flow)
void launch_gain2_tiles(ekfvio_filter* f, int m, int m_pad, int n_pad, bool m_on_device) {
    // (with T2's tile pairs in the same launch: gain2_t2_tiles_kernel)
    ProfScope ps(f, PC_SOLVE, 2.0 * n_pad * (double)m_pad * m_pad + (double)n_pad * n_pad * m_pad);
    PersistArgs pa = PersistArgs();
    pa.L = f->Laug, pa.ldl = f->ld_aug;
    pa.mb = m_pad / PB, pa.idb0 = m_pad / PB + n_pad / PB;
    pa.K = f->Km, pa.ldk = f->ldp;
    pa.Lsign = f->Lsign;
    pa.Sg = f->P, pa.ldsg = f->ldp, pa.T2 = t2_buffer(f), pa.ldt = f->ldp, pa.nstate = f->n;
    GatherArgs ga = make_gather_args(f, m, m_pad, n_pad);
    if (m_on_device) ga.m_dev = f->info + 2;
    const int nX = n_pad / PB, gain_wgs = nX * pa.mb;
    const int total = gain_wgs + nX * (nX + 1) / 2;
    // (> 80 KB of LDS with the 53 KB of tiles: one workgroup per compute unit while the launch fits the compute units)
    const int dyn = total <= f->num_cus ? 30 * 1024 : 0;
    if (dyn && !f->gain2_attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gain2_t2_tiles_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 30 * 1024);
        f->gain2_attr_set = true;
    }
    hipLaunchKernelGGL(gain2_t2_tiles_kernel, dim3(total), dim3(256), dyn, f->stream, pa, ga, gain_wgs);
}

void launch_joseph_g(ekfvio_filter* f, int m, int m_pad, int n_pad, bool m_on_device, const float* T) {
    ProfScope ps(f, PC_SOLVE, 3.0 * n_pad * (double)m_pad);
    hipLaunchKernelGGL(joseph_g_kernel, dim3(n_pad / 64, m_pad / 64), dim3(256), 0, f->stream, f->Km, f->ldp, T ? T : f->P, f->ldp, f->idx,
                       f->Rm, f->yres, f->mu, m, m_on_device ? f->info + 2 : nullptr, f->Gm, f->ldp, f->Wt);
}

// K = X A^-1 = Y L^-1 with Y = X L^-T and L^-T (both from the sweep): one MFMA GEMM that
// skips the structurally zero part of the triangular operand and prunes like sparseView
// (:580).  refine != 0 adds one step of residual refinement against L itself,
// K <- K + (Y - K L) L^-1 (two more GEMMs); on the filter's matrices it changes nothing
// measurable (the error is dominated by the fp32 factor itself), so it is off by default.
bool sweep_supports_schur(const ekfvio_filter* f, int m_pad) {
    // (EKFVIO_SCHUR=1 takes precedence over the persistent launch: launch_chol_sweep skips that path when `schur` is set)
    return f->schur && m_pad / PB < EKF_SWEEP_SPLIT_MB;
}

void launch_gain_from_sweep(ekfvio_filter* f, const float* Laug, int m_pad, int n_pad, int ld, int n, float* K,
                            float* scratch, int ldk, int refine, const GemmEpi* epi) {
    ProfScope ps(f, PC_SOLVE, (refine ? 3.0 : 1.0) * n * (double)m_pad * m_pad);
    if (!refine && !epi && f->persist_gain && gain_in_sweep_shape(f, m_pad, n_pad)) {
        // a shape the persistent launch takes when the handle is alone on its device (this call: it is not, or EKFVIO_SWEEP=0): the
        // tile kernel whose arithmetic that launch's in-sweep gain shares (chol_persist.inc, gain_tile), so that a sequence gives the
        // same bits whichever sweep its updates take (eight handles on one GPU against each one's solo run, tests/test_gpu_shapes.py)
        PersistArgs pa = PersistArgs();
        pa.L = const_cast<float*>(Laug), pa.ldl = ld;
        pa.mb = m_pad / PB, pa.idb0 = m_pad / PB + n_pad / PB;
        pa.K = K, pa.ldk = ldk;
        hipLaunchKernelGGL(gain_tiles_kernel, dim3((n_pad / PB) * pa.mb), dim3(256), 0, f->stream, pa);
        return;
    }
    const float* Lf = Laug;
    const float* Y = Laug + m_pad;
    const float* LinvT = Laug + m_pad + n_pad;
    launch_gemm(f, 1, n, m_pad, m_pad, 1.f, Y, ld, LinvT, ld, 0.f, nullptr, 0, K, ldk, refine ? 0 : 1, 1, refine ? nullptr : epi);
    if (refine) {
        launch_gemm(f, 0, n, m_pad, m_pad, -1.f, K, ldk, Lf, ld, 1.f, Y, ld, scratch, ldk, 0, 1);     // Y - K L
        launch_gemm(f, 1, n, m_pad, m_pad, 1.f, scratch, ldk, LinvT, ld, 1.f, K, ldk, K, ldk, 1, 1);  // + prune
    }
}
