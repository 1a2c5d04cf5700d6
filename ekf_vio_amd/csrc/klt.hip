// ekf_vio_amd/csrc/klt.hip — pyramidal Lucas-Kanade tracker on the GPU.
//
// Replaces KLTTracker::findNewFeaturePositions (include/ekf_vio/KLTTracker.cpp:29-95), i.e.
// cv::calcOpticalFlowPyrLK(lf.img, cf.img, prev, init, Size(21,21), maxLevel 3,
// {COUNT+EPS, 30, 0.01}, OPTFLOW_USE_INITIAL_FLOW, minEig 1e-4) plus the pass / metric
// conversion loop around it (:72-92).  The arithmetic restates OpenCV 3.x's published
// algorithm (see oracle/klt_oracle.cpp for the itemised list); the gradient matrix and the
// mismatch vector are accumulated in exact 64-bit integers, which makes the result
// independent of the reduction order.
//
// Data layout in HBM, per frame and pyramid level: the 8-bit image with a border of
// KLT_BORDER pixels on every side, filled by reflect-101 (what OpenCV's pyramid border
// holds), rows padded to 64 B; and the int16 (dx,dy) Scharr derivatives with the same
// geometry and a ZERO border.  With the borders materialised the tracker's inner loops are
// branch-free: every window that passes OpenCV's bounds test lies inside the padded level.
//
// Kernels: klt_pyramid_kernel (all levels of a frame, padded images and Scharr derivatives, in
// one launch) and klt_track_kernel: one wavefront per landmark, all pyramid levels inside the kernel;
// the 21x21 window is spread over the 64 lanes (7 pixels per lane, template patch and
// gradients in registers), the search region of the current frame is staged in LDS once
// per level (re-staged only if the window walks out of it), and the per-iteration sums are
// 64-bit wave reductions.
#include <math.h>

#include <stdlib.h>
#include <string.h>

#include "common.h"

#define KLT_BORDER 24     // >= window size (21); multiple of 8
#define KLT_MAX_WIN 21
#define KLT_SLOTS 7       // ceil(21*21/64)
#define KLT_R 8           // staging margin around the window
#define KLT_RS (KLT_MAX_WIN + 1 + 2 * KLT_R)  // staged region edge (38)
#define KLT_RP 40         // LDS row pitch of the staged region

#define HIPK(f, expr)                                                              \
    do {                                                                           \
        hipError_t e__ = (expr);                                                   \
        if (e__ != hipSuccess) {                                                   \
            (f)->last_error = std::string(#expr) + ": " + hipGetErrorString(e__);  \
            return EKFVIO_EDEVICE;                                                 \
        }                                                                          \
    } while (0)

namespace {

struct LevelView {
    const uint8_t* img;  // points at padded (0,0)
    const short* der;    // interleaved dx,dy; padded (0,0)
    int w, h, pitch;     // pitch in pixels (same for img bytes and der short2)
};
struct PyrView {
    LevelView lv[4];
    int levels;
};

__device__ __host__ inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

// The whole pyramid of a frame in ONE launch: padded images (reflect-101 borders) and Scharr derivatives of every level.
// A workgroup owns a 32x32 tile of level 0 and the pixels of the coarser levels below it (16x16, 8x8, 4x4); it forms
// every level it needs itself, out of LDS, on a region with a halo that shrinks with the level (22, 10, 4, 1 pixels:
// h_l = 2 h_(l+1) + 2, the 5x5 pyrDown support of the next level's region, which ends with the 1-pixel ring Scharr
// needs at the coarsest level).  Region entries hold the LEVEL's value at the reflect-101 image of their coordinate,
// i.e. what the padded level holds there: level 0 entries are loaded through the reflected index; an out-of-image entry
// of a coarser level is copied from the in-image entry it mirrors (always inside the same region: the halo is shorter
// than the tile plus halo on the other side), NOT computed at its own position: pyrDown and reflection commute only when
// the finer level's size is odd.  pyrDown: [1 4 6 4 1]^2 / 256 with (v + 128) >> 8 (exact integers, so the evaluation
// order is free); Scharr: 3/10/3, int16, zero outside the image.  Every owned pixel is written to its own place and to
// every border position that mirrors it (and a zero derivative there), so a level's padded planes are complete however
// the frame size changed since the buffers were last used.
// Was four launches, one per level (a thread per padded pixel, up to 9 x 25 taps each): 23-27 us of launch latency
// per frame; the levels of a 640x480 frame are 300 workgroups of ~7 KB LDS here.
#define PYR_T 32
#define PYR_NT 512  // threads per workgroup: two wavefronts per SIMD, so one's LDS and memory latency hides under the other's arithmetic
struct PyrOut {
    uint8_t* img[4];
    short* der[4];
    int w[4], h[4], pitch[4];
    int levels;
};
template <int L>
struct PyrGeom {
    static constexpr int H = (L == 0) ? 22 : (L == 1) ? 10 : (L == 2) ? 4 : 1;  // halo
    static constexpr int T = PYR_T >> L;                                         // owned tile edge
    static constexpr int S = T + 2 * H;                                          // region edge: 76, 36, 16, 6
};
// Frame::Frame's cv::resize(img, 1 / inverse_image_scale) (Frame.cpp:15-42), OpenCV 3.x resize.cpp 8-bit INTER_LINEAR: 11-bit
// fixed-point weights (cvRound, saturated to short), horizontal pass kept at 16 fractional bits less 4, vertical pass with
// the two-stage rounding of its VResizeLinear.  Destination pixel (dx, dy) of the (sw x sh) source.  The pyramid kernel
// forms level 0 with this on the fly (a resize launch of its own used to write the resized frame first).
struct ResizeArgs {
    int on = 0, sw = 0, sh = 0;
    double scale_x = 1.0, scale_y = 1.0;
};
__device__ inline short sat_short_rn(float v) {
    const int i = __float2int_rn(v);  // round half to even = cvRound
    return (short)min(32767, max(-32768, i));
}
// The separable part of a destination coordinate: source index pair and the two weights (x: clamped as resize.cpp's
// xofs / alpha tables are; y: the row indices are clamped, the weights are not).
struct ResizeTap {
    int s0, s1;
    short w0, w1;
};
__device__ inline ResizeTap resize_tap_x(const ResizeArgs& ra, int dx) {
    float fx = (float)((dx + 0.5) * ra.scale_x - 0.5);
    int sx = (int)floorf(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= ra.sw - 1) { fx = 0; sx = ra.sw - 1; }
    return {sx, min(sx + 1, ra.sw - 1), sat_short_rn((1.f - fx) * 2048), sat_short_rn(fx * 2048)};
}
__device__ inline ResizeTap resize_tap_y(const ResizeArgs& ra, int dy) {
    float fy = (float)((dy + 0.5) * ra.scale_y - 0.5);
    const int sy = (int)floorf(fy);
    fy -= sy;
    return {min(max(sy, 0), ra.sh - 1), min(max(sy + 1, 0), ra.sh - 1), sat_short_rn((1.f - fy) * 2048), sat_short_rn(fy * 2048)};
}
__device__ inline uint8_t resize_combine(const uint8_t* __restrict__ src, int sstride, const ResizeTap& tx, const ResizeTap& ty) {
    const uint8_t *S0 = src + (size_t)ty.s0 * sstride, *S1 = src + (size_t)ty.s1 * sstride;
    const int a0 = tx.w0, a1 = tx.w1, b0 = ty.w0, b1 = ty.w1;
    const int r0 = S0[tx.s0] * a0 + S0[tx.s1] * a1, r1 = S1[tx.s0] * a0 + S1[tx.s1] * a1;
    return (uint8_t)((((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2);
}

// calls fn(X) for every X in [-KLT_BORDER, w + KLT_BORDER) other than x itself whose reflect-101 image is x (0 <= x < w)
template <class F>
__device__ __forceinline__ void for_mirror_images(int x, int w, F&& fn) {
    if (w >= 14) {
        // period P = 2 (w - 1) >= 26 against a padded width of w + 48: the images x + k P and -x + k P with k in -1 .. 2 are
        // all there can be (no division: this is the path of every pyramid level narrower than two borders plus two)
        const int P = 2 * (w - 1);
#pragma unroll
        for (int k = -1; k <= 2; k++) {
            const int X1 = x + k * P, X2 = -x + k * P;
            if (X1 != x && X1 >= -KLT_BORDER && X1 < w + KLT_BORDER) fn(X1);
            if (x != 0 && x != w - 1 && X2 >= -KLT_BORDER && X2 < w + KLT_BORDER) fn(X2);
        }
        return;
    }
    if (w == 1) {
        for (int X = -KLT_BORDER; X < 1 + KLT_BORDER; X++)
            if (X != x) fn(X);
        return;
    }
    const int P = 2 * (w - 1);
    for (int X = x - ((x + KLT_BORDER) / P) * P; X < w + KLT_BORDER; X += P)
        if (X != x) fn(X);
    if (x != 0 && x != w - 1) {  // otherwise -x is in x's residue class
        const int y = -x;        // y + KLT_BORDER may be negative: first member of the class at or above -KLT_BORDER
        int X = (y >= -KLT_BORDER) ? y - ((y + KLT_BORDER) / P) * P : y + ((-KLT_BORDER - y + P - 1) / P) * P;
        for (; X < w + KLT_BORDER; X += P) fn(X);
    }
}
// Border copies of a tile's pixels for a level narrower or lower than two borders plus two (several reflections can land
// on one pixel): small frames and crops only, kept out of line.
__device__ __attribute__((noinline)) void pyr_emit_border_general(const uint8_t* R, int S, int H, int T, int w, int h, int pitch,
                                                                  uint8_t* img, short* der, int ax, int ay, int tid) {
    for (int e = tid; e < T * T; e += PYR_NT) {
        const int lx = e % T, ly = e / T;
        const int x = ax + lx, y = ay + ly;
        if (x >= w || y >= h) continue;
        const uint8_t c1 = R[(ly + H) * S + lx + H];
        auto put = [&](int X, int Y) {
            const size_t at = (size_t)(Y + KLT_BORDER) * pitch + (X + KLT_BORDER);
            img[at] = c1;
            *reinterpret_cast<int*>(der + at * 2) = 0;
        };
        for_mirror_images(x, w, [&](int X) { put(X, y); });
        for_mirror_images(y, h, [&](int Y) {
            put(x, Y);
            for_mirror_images(x, w, [&](int X) { put(X, Y); });
        });
    }
}
template <int L>
__device__ __forceinline__ void pyr_emit_level(const uint8_t* R, const PyrOut& o, int tx, int ty, int tid) {
    using G = PyrGeom<L>;
    constexpr int B = KLT_BORDER;
    const int w = o.w[L], h = o.h[L], pitch = o.pitch[L];
    const int ax = tx * G::T, ay = ty * G::T;
    uint8_t* img = o.img[L];
    short* der = o.der[L];
    // A level at least two borders plus two wide has at most ONE mirror image per pixel and axis in the border: x in [1, B]
    // at -x, x in [w-1-B, w-2] at 2(w-1)-x.  Then an owned pixel goes to its own place and, if the tile lies near an edge
    // of the image (workgroup-uniform), to up to three border positions (image byte, zero derivative).
    const bool wide = w >= 2 * B + 2 && h >= 2 * B + 2;
    const bool edge = wide && (ax <= B || ax + G::T - 1 >= w - 1 - B || ay <= B || ay + G::T - 1 >= h - 1 - B);
#pragma unroll
    for (int it = 0; it < (G::T * G::T + PYR_NT - 1) / PYR_NT; it++) {
        const int e = tid + PYR_NT * it;
        const int lx = e % G::T, ly = e / G::T;
        const int x = ax + lx, y = ay + ly;
        if (e < G::T * G::T && x < w && y < h) {
            // the 3x3 neighbourhood as three byte windows cut out of aligned dword pairs (3 LDS reads instead of 9)
            const unsigned* R32 = reinterpret_cast<const unsigned*>(R);
            const int o0 = (ly + G::H - 1) * G::S + lx + G::H - 1;
            unsigned rw[3];
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const int ob = o0 + r * G::S;
                rw[r] = __builtin_amdgcn_alignbyte(R32[(ob >> 2) + 1], R32[ob >> 2], ob & 3);
            }
            const int u0 = rw[0] & 0xff, u1 = (rw[0] >> 8) & 0xff, u2 = (rw[0] >> 16) & 0xff;
            const int c0 = rw[1] & 0xff, c1 = (rw[1] >> 8) & 0xff, c2 = (rw[1] >> 16) & 0xff;
            const int d0 = rw[2] & 0xff, d1 = (rw[2] >> 8) & 0xff, d2 = (rw[2] >> 16) & 0xff;
            const int t0m = (u0 + d0) * 3 + c0 * 10, t0p = (u2 + d2) * 3 + c2 * 10;
            const int t1m = d0 - u0, t1c = d1 - u1, t1p = d2 - u2;
            const size_t at = (size_t)(y + B) * pitch + (x + B);
            img[at] = (uint8_t)c1;
            short2 g;
            g.x = (short)(t0p - t0m);
            g.y = (short)((t1p + t1m) * 3 + t1c * 10);
            *reinterpret_cast<short2*>(der + at * 2) = g;
            if (edge) {
                const int X = (x >= 1 && x <= B) ? -x : ((x >= w - 1 - B && x <= w - 2) ? 2 * (w - 1) - x : x);
                const int Y = (y >= 1 && y <= B) ? -y : ((y >= h - 1 - B && y <= h - 2) ? 2 * (h - 1) - y : y);
                const size_t ax_ = (size_t)(y + B) * pitch + (X + B), ay_ = (size_t)(Y + B) * pitch + (x + B),
                             axy = (size_t)(Y + B) * pitch + (X + B);
                if (X != x) {
                    img[ax_] = (uint8_t)c1;
                    *reinterpret_cast<int*>(der + ax_ * 2) = 0;
                }
                if (Y != y) {
                    img[ay_] = (uint8_t)c1;
                    *reinterpret_cast<int*>(der + ay_ * 2) = 0;
                }
                if (X != x && Y != y) {
                    img[axy] = (uint8_t)c1;
                    *reinterpret_cast<int*>(der + axy * 2) = 0;
                }
            }
        }
    }
    if (!wide) pyr_emit_border_general(R, G::S, G::H, G::T, w, h, pitch, img, der, ax, ay, tid);
}
// region of level L (Rd) from the region of level L-1 (Rs)
template <int L>
__device__ __forceinline__ void pyr_down_level(const uint8_t* Rs, uint8_t* Rd, const PyrOut& o, int tx, int ty, int tid) {
    using G = PyrGeom<L>;
    using Gs = PyrGeom<L - 1>;
    const int w = o.w[L], h = o.h[L];
    const int ox = tx * G::T - G::H, oy = ty * G::T - G::H;
    for (int e = tid; e < G::S * G::S; e += PYR_NT) {
        const int jx = e % G::S, jy = e / G::S;
        const int qx = ox + jx, qy = oy + jy;
        if (qx < 0 || qx >= w || qy < 0 || qy >= h) continue;
        // source rows 2 jy .. 2 jy + 4, bytes 2 jx .. 2 jx + 4 of each (the entry for (2 qx, 2 qy) is at (2 jy + 2, 2 jx + 2)):
        // a five-byte window at an even offset lies inside one aligned dword pair
        const unsigned* Rs32 = reinterpret_cast<const unsigned*>(Rs);
        const int o0 = 2 * jy * Gs::S + 2 * jx;
        int v = 0;
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const int ob = o0 + j * Gs::S;
            const unsigned d0 = Rs32[ob >> 2], d1 = Rs32[(ob >> 2) + 1];
            const unsigned lo = __builtin_amdgcn_alignbyte(d1, d0, ob & 3);
            const unsigned b4 = ((ob & 2) ? (d1 >> 16) : d1) & 0xff;
            const int rs = (int)((lo & 0xff) + b4 + 4 * (((lo >> 8) & 0xff) + (lo >> 24)) + 6 * ((lo >> 16) & 0xff));
            const int wj = (j == 2) ? 6 : ((j == 1 || j == 3) ? 4 : 1);
            v += wj * rs;
        }
        Rd[e] = (uint8_t)((v + 128) >> 8);
    }
    __syncthreads();
    // a region that lies inside the level (workgroup-uniform) has no mirrored entries
    if (ox >= 0 && oy >= 0 && ox + G::S <= w && oy + G::S <= h) return;
    for (int e = tid; e < G::S * G::S; e += PYR_NT) {
        const int jx = e % G::S, jy = e / G::S;
        const int qx = ox + jx, qy = oy + jy;
        if (qx >= 0 && qx < w && qy >= 0 && qy < h) continue;
        const int rx = reflect101(qx, w), ry = reflect101(qy, h);
        Rd[e] = Rd[(ry - oy) * G::S + (rx - ox)];
    }
    __syncthreads();
}
__global__ __launch_bounds__(PYR_NT) void klt_pyramid_kernel(const uint8_t* __restrict__ src, int spitch, PyrOut o, long long* dbg,
                                                             ResizeArgs ra) {
    // diagnostic phase stamps of one interior workgroup (scripts/klt_timing.py); dbg is null in production
#define PSTAMP(slot)                                                                                                     \
    do {                                                                                                                 \
        if (dbg && blockIdx.x == 2 && blockIdx.y == 2 && threadIdx.x == 0) dbg[940 + (slot)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
    PSTAMP(0);
    const int wg_ = blockIdx.y * gridDim.x + blockIdx.x;
    if (dbg && threadIdx.x == 0 && wg_ < 450) dbg[2 * wg_] = (long long)__builtin_amdgcn_s_memrealtime();
    // regions are read through aligned dword pairs: 8 bytes of slack behind each (a window's second dword may lie past the end)
    __shared__ __attribute__((aligned(8))) uint8_t R0[PyrGeom<0>::S * PyrGeom<0>::S + 8];
    __shared__ __attribute__((aligned(8))) uint8_t R1[PyrGeom<1>::S * PyrGeom<1>::S + 8];
    __shared__ __attribute__((aligned(8))) uint8_t R2[PyrGeom<2>::S * PyrGeom<2>::S + 8];
    __shared__ __attribute__((aligned(8))) uint8_t R3[PyrGeom<3>::S * PyrGeom<3>::S + 8];
    const int tid = threadIdx.x, tx = blockIdx.x, ty = blockIdx.y;
    {
        using G = PyrGeom<0>;
        const int w = o.w[0], h = o.h[0];
        const int ox = tx * G::T - G::H, oy = ty * G::T - G::H;
        // Frames at least as large as the region: all loads of a thread in flight before the first LDS write, one reflection
        // on either side, then a clamp for the region entries nothing reads (the region is 76 wide wherever the image
        // ends).  Smaller frames take the general reflection, rolled.
        constexpr int NL = (G::S * G::S + PYR_NT - 1) / PYR_NT;
        if (ra.on) {
            // level 0 is the resized frame: every region entry is resized out of the full-size source on the fly (unrolled:
            // the four taps of all of a thread's entries are in flight together)
            // the coordinate arithmetic (double) is separable: one table entry per region column and per region row
            __shared__ ResizeTap s_tx[G::S], s_ty[G::S];
            if (tid < G::S) s_tx[tid] = resize_tap_x(ra, reflect101(ox + tid, w));
            else if (tid < 2 * G::S) s_ty[tid - G::S] = resize_tap_y(ra, reflect101(oy + tid - G::S, h));
            __syncthreads();
            uint8_t v[NL];
#pragma unroll
            for (int it = 0; it < NL; it++) {
                const int e = min(tid + PYR_NT * it, G::S * G::S - 1);
                v[it] = resize_combine(src, spitch, s_tx[e % G::S], s_ty[e / G::S]);
            }
#pragma unroll
            for (int it = 0; it < NL; it++) {
                const int e = tid + PYR_NT * it;
                if (e < G::S * G::S) R0[e] = v[it];
            }
        } else if (w >= G::S && h >= G::S) {
            uint8_t v[NL];
#pragma unroll
            for (int it = 0; it < NL; it++) {
                const int e = min(tid + PYR_NT * it, G::S * G::S - 1);
                const int px = ox + e % G::S, py = oy + e / G::S;
                int rx = px < 0 ? -px : px;
                rx = max(rx >= w ? 2 * w - 2 - rx : rx, 0);
                int ry = py < 0 ? -py : py;
                ry = max(ry >= h ? 2 * h - 2 - ry : ry, 0);
                v[it] = src[(size_t)ry * spitch + rx];
            }
#pragma unroll
            for (int it = 0; it < NL; it++) {
                const int e = tid + PYR_NT * it;
                if (e < G::S * G::S) R0[e] = v[it];
            }
        } else {
#pragma unroll 1
            for (int e = tid; e < G::S * G::S; e += PYR_NT)
                R0[e] = src[(size_t)reflect101(oy + e / G::S, h) * spitch + reflect101(ox + e % G::S, w)];
        }
        __syncthreads();
    }
    PSTAMP(1);
    pyr_emit_level<0>(R0, o, tx, ty, tid);
    PSTAMP(2);
    if (o.levels > 1) {
        pyr_down_level<1>(R0, R1, o, tx, ty, tid);
        PSTAMP(3);
        pyr_emit_level<1>(R1, o, tx, ty, tid);
    }
    PSTAMP(4);
    if (o.levels > 2) {
        pyr_down_level<2>(R1, R2, o, tx, ty, tid);
        pyr_emit_level<2>(R2, o, tx, ty, tid);
    }
    PSTAMP(5);
    if (o.levels > 3) {
        pyr_down_level<3>(R2, R3, o, tx, ty, tid);
        pyr_emit_level<3>(R3, o, tx, ty, tid);
    }
    PSTAMP(6);
    if (dbg && threadIdx.x == 0 && wg_ < 450) dbg[2 * wg_ + 1] = (long long)__builtin_amdgcn_s_memrealtime();
#undef PSTAMP
}

// Exact sum over the wavefront of per-lane int32 partials, |v| < 2^28.  The value is split into its low 16 bits
// (0 .. 65535) and the arithmetic rest (|v >> 16| < 2^12): each half sums over 64 lanes inside int32, so both reductions
// run entirely in the vector unit -- a DPP butterfly inside each row of 16 lanes (quad_perm xor 1, xor 2, row_half_mirror,
// row_mirror), then row_bcast:15 / row_bcast:31 carry the row sums into lane 63 -- and the two results are put together
// exactly: (sum of rests << 16) + sum of low halves.  (Round 2 read the eight 8-lane sums out with v_readlane and added
// them as 64-bit scalars: 35 instructions per value on the Gauss-Newton step's serial path instead of 19; exact either
// way, so the bits are the same: 1.6 k -> 1.4 k cycles per step, 25.2 -> 22.5 us per 256 points.)  Wave-uniform results.
__device__ inline int wave_sum_small_i32(int v) {  // |sum| must fit int32
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);    // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);   // row_half_mirror: lane i <-> 7 - i
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);   // row_mirror: lane i <-> 15 - i: every lane holds its row's sum
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3: lane 63 holds the total
    return __builtin_amdgcn_readlane(v, 63);
}
// (float) of the exact sum, i.e. the 64-bit integer (hi << 16) + lo rounded ONCE to float, as OpenCV's (float)(int64 sum)
// would be: formed in double (hi * 2^16 + lo is exact there: |sum| < 2^34) and narrowed -- four vector instructions where
// the 64-bit integer -> float conversion is a normalising shift sequence of seventeen
__device__ inline float wave_sum_i32_as_float(int v) {
    const int lo = wave_sum_small_i32(v & 0xffff), hi = wave_sum_small_i32(v >> 16);
    return (float)((double)hi * 65536.0 + (double)lo);
}
// Two such sums at once (the Gauss-Newton step's b1, b2): the four 16-bit halves go through the butterfly in lockstep, so that every DPP
// step finds three independent ones between itself and the step that reads its result -- one after the other, each step waits two
// issue slots for its operand (the DPP read-after-write hazard): ~20 slots of the ~200 on the iteration's serial path.
__device__ inline void wave_sum2_i32_as_float(int a, int b, float& fa, float& fb) {
    int v0 = a & 0xffff, v1 = a >> 16, v2 = b & 0xffff, v3 = b >> 16;
#define KLT_DPP4(ctrl, rmask, bound)                                        \
    do {                                                                    \
        const int t0 = __builtin_amdgcn_update_dpp(0, v0, ctrl, rmask, 0xf, bound); \
        const int t1 = __builtin_amdgcn_update_dpp(0, v1, ctrl, rmask, 0xf, bound); \
        const int t2 = __builtin_amdgcn_update_dpp(0, v2, ctrl, rmask, 0xf, bound); \
        const int t3 = __builtin_amdgcn_update_dpp(0, v3, ctrl, rmask, 0xf, bound); \
        v0 += t0, v1 += t1, v2 += t2, v3 += t3;                             \
    } while (0)
    KLT_DPP4(0xB1, 0xf, true);
    KLT_DPP4(0x4E, 0xf, true);
    KLT_DPP4(0x141, 0xf, true);
    KLT_DPP4(0x140, 0xf, true);
    KLT_DPP4(0x142, 0xa, false);
    KLT_DPP4(0x143, 0xc, false);
#undef KLT_DPP4
    const int s0 = __builtin_amdgcn_readlane(v0, 63), s1 = __builtin_amdgcn_readlane(v1, 63);
    const int s2 = __builtin_amdgcn_readlane(v2, 63), s3 = __builtin_amdgcn_readlane(v3, 63);
    fa = (float)((double)s1 * 65536.0 + (double)s0);
    fb = (float)((double)s3 * 65536.0 + (double)s2);
}
__device__ inline int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// Copies `rows` rows of `nbytes` bytes starting at byte address src (row stride `pitch`, a multiple of 4) into LDS
// with aligned 4-byte loads: LDS row r holds the aligned dwords that cover the span, LDP dwords per row.  Returns
// the byte offset of the first wanted byte inside an LDS row (the same for every row).  MAXR x LDP bounds the
// element count at compile time, so that all loads of a lane are issued before the first LDS write.
template <int MAXR, int LDP>
struct RowStage {
    static constexpr int NL = (MAXR * LDP + 63) / 64;  // loads per lane
    unsigned v[NL];
    int shift;
    __device__ inline void issue(const uint8_t* src, int pitch, int rows, int nbytes, int lane) {
        const unsigned long long a0 = (unsigned long long)src;
        shift = (int)(a0 & 3ull);
        // an integer -> pointer cast yields a generic pointer (flat_load: counted on lgkmcnt with the LDS traffic as well);
        // the pyramid lives in global memory
        typedef const __attribute__((address_space(1))) unsigned* global_u32;
        const global_u32 base = (global_u32)(a0 - shift);
        const int nd = (shift + nbytes + 3) >> 2;  // dwords per row actually needed (<= LDP)
        const int pd = pitch >> 2;
#pragma unroll
        for (int q = 0; q < NL; q++) {
            const int e = lane + 64 * q, r = e / LDP, c = e % LDP;
            // clamped instead of predicated: every load is unconditional and in flight with the others
            v[q] = base[(size_t)min(r, rows - 1) * pd + min(c, nd - 1)];
        }
    }
    __device__ inline void commit(unsigned* lds, int lane) const {
#pragma unroll
        for (int q = 0; q < NL; q++) {
            const int e = lane + 64 * q;
            if (e < MAXR * LDP) lds[e] = v[q];
        }
    }
};

#define KLT_TW (KLT_MAX_WIN + 1)                 // template rows / columns incl. the bilinear neighbour (22)
#define KLT_TIP 8                                // dwords per LDS row of the template image (22 + 3 bytes -> 7)
#define KLT_JP 11                                // dwords per LDS row of the search region (38 + 3 bytes -> 11)

// What findNewFeaturePositions does around calcOpticalFlowPyrLK, done by the point's own wavefront instead of a launch
// before and one after the tracker (each ~4 us of launch latency for a few hundred flops):
//   from_state: reference pixel from the last KLT result (previous frame's K) and initial guess from the predicted
//               landmark (current frame's K) (KLTTracker.cpp:53-59);
//   finish:     pass test with the kill box, pixel -> metric conversion and the constant measurement covariance of
//               estimateUncertainty (:72-92, :100-106).  Off with the sample-based covariance, whose kernel runs between.
struct TrackFuse {
    const float* last_klt = nullptr;
    const float* mu = nullptr;
    float fxp = 0, fyp = 0, cxp = 0, cyp = 0, fxc = 0, fyc = 0, cxc = 0, cyc = 0;
    int from_state = 0, finish = 0;
    int w = 0, h = 0, kill_pad = 0;
    float r0 = 0, r1 = 0;
    float* z = nullptr;
    float* R = nullptr;
    uint8_t* pass = nullptr;
};

// One wavefront per point, three lanes per window row.  Mirrors LKTrackerInvoker; every scalar expression is evaluated
// redundantly (and identically) by all 64 lanes.  Per level everything the wavefront reads from memory
// (template image, its derivatives, the search region) is fetched with aligned 4-byte loads into LDS first;
// the per-pixel work then runs out of LDS.
__global__ __launch_bounds__(64) void klt_track_kernel(PyrView P, PyrView Q, float* prev_px, float* next_px, uint8_t* status, int n,
                                                       int win, int max_iter, float eps2, float min_eig, long long* dbg, TrackFuse tf) {
#define KSTAMP(slot)                                                                                   \
    do {                                                                                               \
        if (dbg && blockIdx.x == 0 && threadIdx.x == 0) dbg[900 + (slot)] = (long long)__builtin_amdgcn_s_memtime(); \
    } while (0)
    KSTAMP(0);
    __shared__ unsigned regJ[KLT_RS * KLT_JP];         // search region of J, rows of aligned dwords
    __shared__ unsigned regI[KLT_TW * KLT_TIP];        // template image rows
    __shared__ unsigned regD[KLT_TW * KLT_TW];         // template derivatives, one (dx,dy) short pair per pixel
    const int pt = blockIdx.x;
    const int lane = threadIdx.x;
    if (pt >= n) return;
    const int levels = min(P.levels, Q.levels);
    const int W_BITS = 14;
    const float FLT_SCALE = 1.f / (1 << 20);
    const float half = (win - 1) * 0.5f;
    // Lane geometry: three lanes per window row, each a run of `chunk` = ceil(win / 3) consecutive pixels of that row
    // (win = 21: 63 lanes x 7 pixels).  A run's bilinear samples are chunk + 1 consecutive bytes of two image rows: two
    // 8-byte windows cut out of aligned dwords (6 LDS reads) where a pixel-per-slot layout reads 4 bytes per pixel (28).
    const int chunk = (win + 2) / 3;
    const int lrow = min(lane / 3, win - 1), x0 = (lane % 3) * chunk;
    bool sv[KLT_SLOTS];
#pragma unroll
    for (int t = 0; t < KLT_SLOTS; t++) sv[t] = lane < 3 * win && t < chunk && x0 + t < win;
    // bytes o .. o + 7 of an LDS byte array (o + 11 stays inside the staged row: see the row pitches)
    auto window8 = [](const unsigned* base, int o, unsigned& w0, unsigned& w1) {
        const unsigned d0 = base[o >> 2], d1 = base[(o >> 2) + 1], d2 = base[(o >> 2) + 2];
        w0 = __builtin_amdgcn_alignbyte(d1, d0, o & 3);
        w1 = __builtin_amdgcn_alignbyte(d2, d1, o & 3);
    };
    auto byte_of = [](unsigned w0, unsigned w1, int k) -> int { return (int)(((k < 4 ? w0 : w1) >> (8 * (k & 3))) & 0xffu); };
    float ppx0, ppy0, ox, oy;
    if (tf.from_state) {
        ppx0 = tf.last_klt[2 * pt] * tf.fxp + tf.cxp;
        ppy0 = tf.last_klt[2 * pt + 1] * tf.fyp + tf.cyp;
        ox = tf.fxc * tf.mu[EKF_BASE + 3 * pt] + tf.cxc;
        oy = tf.fyc * tf.mu[EKF_BASE + 3 * pt + 1] + tf.cyc;
        if (lane == 0) {  // the sample-based covariance reads them
            prev_px[2 * pt] = ppx0;
            prev_px[2 * pt + 1] = ppy0;
        }
    } else {
        ppx0 = prev_px[2 * pt];
        ppy0 = prev_px[2 * pt + 1];
        ox = next_px[2 * pt];
        oy = next_px[2 * pt + 1];
    }
    bool ok = true;
    unsigned itpack = 0u;
    for (int level = levels - 1; level >= 0; level--) {
        const LevelView I = P.lv[level];
        const LevelView J = Q.lv[level];
        const float sc = (float)(1. / (1 << level));
        float ppx = ppx0 * sc, ppy = ppy0 * sc;
        if (level == levels - 1) {
            ox = ox * sc;
            oy = oy * sc;
        } else {
            ox = ox * 2.f;
            oy = oy * 2.f;
        }
        ppx -= half;
        ppy -= half;
        const int ipx = (int)floorf(ppx), ipy = (int)floorf(ppy);
        if (ipx < -win || ipx >= I.w || ipy < -win || ipy >= I.h) {
            if (level == 0) ok = false;
            continue;
        }
        KSTAMP(1 + 8 * level);
        // ---- one batch of aligned loads: template image, template derivatives, first search region ----
        float nx = ox - half, ny = oy - half;
        int rx0 = 0, ry0 = 0, shJ = 0;
        bool staged = false;
        __syncthreads();  // previous level's LDS readers are done
        const size_t o0 = (size_t)(ipy + KLT_BORDER) * I.pitch + (ipx + KLT_BORDER);
        RowStage<KLT_TW, KLT_TIP> stI;
        RowStage<KLT_TW, KLT_TW> stD;
        RowStage<KLT_RS, KLT_JP> stJ;
        stI.issue(I.img + o0, I.pitch, win + 1, win + 1, lane);
        stD.issue((const uint8_t*)(I.der + o0 * 2), I.pitch * 4, win + 1, 4 * (win + 1), lane);  // one dword per pixel
        {
            const int inx = (int)floorf(nx), iny = (int)floorf(ny);
            if (!(inx < -win || inx >= J.w || iny < -win || iny >= J.h)) {
                rx0 = min(max(inx - KLT_R, -KLT_BORDER), J.w + KLT_BORDER - KLT_RS);
                ry0 = min(max(iny - KLT_R, -KLT_BORDER), J.h + KLT_BORDER - KLT_RS);
                staged = true;
            }
        }
        // (when the start point is outside J nothing will be read from regJ: the iteration loop leaves first)
        stJ.issue(J.img + (size_t)(ry0 + KLT_BORDER) * J.pitch + rx0 + KLT_BORDER, J.pitch, KLT_RS, KLT_RS, lane);
        stI.commit(regI, lane);
        stD.commit(regD, lane);
        stJ.commit(regJ, lane);
        const int shI = stI.shift;
        shJ = stJ.shift;
        __syncthreads();
        KSTAMP(2 + 8 * level);
        float a = ppx - ipx, b = ppy - ipy;
        int iw00 = __float2int_rn((1.f - a) * (1.f - b) * (1 << W_BITS));
        int iw01 = __float2int_rn(a * (1.f - b) * (1 << W_BITS));
        int iw10 = __float2int_rn((1.f - a) * b * (1 << W_BITS));
        int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
        int Iv[KLT_SLOTS], Ix[KLT_SLOTS], Iy[KLT_SLOTS];
        int pA11 = 0, pA12 = 0, pA22 = 0;  // per-lane partials: 7 terms of at most 4080^2 each
        {
            unsigned ia0, ia1, ib0, ib1;  // image rows lrow and lrow + 1, bytes x0 .. x0 + 7
            const int oI = lrow * (KLT_TIP * 4) + shI + x0;
            window8(regI, oI, ia0, ia1);
            window8(regI, oI + KLT_TIP * 4, ib0, ib1);
            unsigned dA[KLT_SLOTS + 1], dB[KLT_SLOTS + 1];  // derivative pairs of the same pixels
#pragma unroll
            for (int t = 0; t <= KLT_SLOTS; t++) {
                const int xc = min(x0 + t, KLT_TW - 1);  // runs of a short window's last lane: clamped, masked below
                dA[t] = regD[lrow * KLT_TW + xc];
                dB[t] = regD[(lrow + 1) * KLT_TW + xc];
            }
#pragma unroll
            for (int t = 0; t < KLT_SLOTS; t++) {
                // every factor fits 24 bits (pixels <= 255, weights <= 2^14, derivatives <= 4080, differences <= 8160):
                // v_mul_i32_i24 / v_mad_i32_i24 run at full rate, v_mul_lo_u32 (what `*` compiles to) at a quarter
                Iv[t] = descale(__mul24(byte_of(ia0, ia1, t), iw00) + __mul24(byte_of(ia0, ia1, t + 1), iw01) +
                                    __mul24(byte_of(ib0, ib1, t), iw10) + __mul24(byte_of(ib0, ib1, t + 1), iw11), W_BITS - 5);
                const unsigned d00 = dA[t], d01 = dA[t + 1], d10 = dB[t], d11 = dB[t + 1];
                Ix[t] = descale(__mul24((short)(d00 & 0xffff), iw00) + __mul24((short)(d01 & 0xffff), iw01) +
                                    __mul24((short)(d10 & 0xffff), iw10) + __mul24((short)(d11 & 0xffff), iw11), W_BITS);
                Iy[t] = descale(__mul24((short)(d00 >> 16), iw00) + __mul24((short)(d01 >> 16), iw01) +
                                    __mul24((short)(d10 >> 16), iw10) + __mul24((short)(d11 >> 16), iw11), W_BITS);
                if (!sv[t]) Iv[t] = Ix[t] = Iy[t] = 0;
                pA11 += __mul24(Ix[t], Ix[t]);
                pA12 += __mul24(Ix[t], Iy[t]);
                pA22 += __mul24(Iy[t], Iy[t]);
            }
        }
        const float A11 = wave_sum_i32_as_float(pA11) * FLT_SCALE, A12 = wave_sum_i32_as_float(pA12) * FLT_SCALE,
                    A22 = wave_sum_i32_as_float(pA22) * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * win * win);
        if (minEig < min_eig || D < 1.1920929e-07f) {
            if (level == 0) ok = false;
            continue;
        }
        D = 1.f / D;
        KSTAMP(3 + 8 * level);
        float pdx = 0.f, pdy = 0.f;
        int itc = 0;
        for (int j = 0; j < max_iter; j++) {
            const int inx = (int)floorf(nx), iny = (int)floorf(ny);
            if (inx < -win || inx >= J.w || iny < -win || iny >= J.h) {
                if (level == 0) ok = false;
                break;
            }
            if (!staged || inx < rx0 || iny < ry0 || inx + win > rx0 + KLT_RS - 1 || iny + win > ry0 + KLT_RS - 1) {
                // (re)stage the search region [rx0, rx0+RS) x [ry0, ry0+RS) of J into LDS, clamped to the padded level
                rx0 = min(max(inx - KLT_R, -KLT_BORDER), J.w + KLT_BORDER - KLT_RS);
                ry0 = min(max(iny - KLT_R, -KLT_BORDER), J.h + KLT_BORDER - KLT_RS);
                __syncthreads();
                RowStage<KLT_RS, KLT_JP> st2;
                st2.issue(J.img + (size_t)(ry0 + KLT_BORDER) * J.pitch + rx0 + KLT_BORDER, J.pitch, KLT_RS, KLT_RS, lane);
                st2.commit(regJ, lane);
                shJ = st2.shift;
                __syncthreads();
                staged = true;
            }
            a = nx - inx;
            b = ny - iny;
            iw00 = __float2int_rn((1.f - a) * (1.f - b) * (1 << W_BITS));
            iw01 = __float2int_rn(a * (1.f - b) * (1 << W_BITS));
            iw10 = __float2int_rn((1.f - a) * b * (1 << W_BITS));
            iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            int pb1 = 0, pb2 = 0;  // per-lane partials: 7 terms of at most 8160 * 4080 each
            {
                unsigned ja0, ja1, jb0, jb1;  // search-region rows, bytes of the run and its right neighbour
                const int oJ = (iny - ry0 + lrow) * (KLT_JP * 4) + inx - rx0 + shJ + x0;
                window8(regJ, oJ, ja0, ja1);
                window8(regJ, oJ + KLT_JP * 4, jb0, jb1);
#pragma unroll
                for (int t = 0; t < KLT_SLOTS; t++) {  // Ix = Iy = 0 in the slots past the window: their terms vanish
                    const int diff = descale(__mul24(byte_of(ja0, ja1, t), iw00) + __mul24(byte_of(ja0, ja1, t + 1), iw01) +
                                                 __mul24(byte_of(jb0, jb1, t), iw10) + __mul24(byte_of(jb0, jb1, t + 1), iw11), W_BITS - 5) - Iv[t];
                    pb1 += __mul24(diff, Ix[t]);
                    pb2 += __mul24(diff, Iy[t]);
                }
            }
            itc++;
            float b1, b2;
            wave_sum2_i32_as_float(pb1, pb2, b1, b2);
            b1 *= FLT_SCALE;
            b2 *= FLT_SCALE;
            const float dx = (A12 * b2 - A22 * b1) * D;
            const float dy = (A12 * b1 - A11 * b2) * D;
            nx += dx;
            ny += dy;
            ox = nx + half;
            oy = ny + half;
            if (dx * dx + dy * dy <= eps2) break;
            if (j > 0 && fabsf(dx + pdx) < 0.01f && fabsf(dy + pdy) < 0.01f) {
                ox -= dx * 0.5f;
                oy -= dy * 0.5f;
                break;
            }
            pdx = dx;
            pdy = dy;
        }
        KSTAMP(4 + 8 * level);
        if (dbg && blockIdx.x == 0 && threadIdx.x == 0) dbg[900 + 5 + 8 * level] = itc;
        itpack |= (unsigned)itc << (8 * level);  // diagnostic: this landmark's Gauss-Newton steps, a byte per level
        if (ok && level == 0) {
            const int fx = (int)floorf(ox - half), fy = (int)floorf(oy - half);
            if (fx < -win || fx >= J.w || fy < -win || fy >= J.h) ok = false;
        }
    }
    if (dbg && lane == 0 && pt < 896) dbg[pt] = (long long)itpack;  // (scripts/klt_iterations.py)
    if (lane == 0) {
        next_px[2 * pt] = ox;
        next_px[2 * pt + 1] = oy;
        status[pt] = ok ? 1 : 0;
        if (tf.finish) {
            const bool p = ok && !(ox < tf.kill_pad || oy < tf.kill_pad || tf.w - ox < tf.kill_pad || tf.h - oy < tf.kill_pad);
            tf.pass[pt] = p ? 1 : 0;
            tf.z[2 * pt] = p ? (ox - tf.cxc) / tf.fxc : 0.f;
            tf.z[2 * pt + 1] = p ? (oy - tf.cyc) / tf.fyc : 0.f;
            tf.R[4 * pt] = p ? tf.r0 : 0.f;
            tf.R[4 * pt + 1] = 0.f;  // the off-diagonals of estimateUncertainty are zero, scaled or not
            tf.R[4 * pt + 2] = 0.f;
            tf.R[4 * pt + 3] = p ? tf.r1 : 0.f;
        }
    }
}


// EKFVIO::publishPoints (EKFVIO.cpp:479-518): camera-frame point (u/rho, v/rho, 1/rho) per landmark -- p(2) = 1.0/p(2)
// in double, narrowed, then two float products -- and the "intensity" channel f.img.at<uchar>(e.getPixel(f)): the byte
// at the landmark's pixel (Feature::getPixel: K(0)*mu(0) + K(2), K(4)*mu(1) + K(5); cv::Point2f -> cv::Point rounds
// to nearest even, cvRound).  The reference reads outside the image unchecked; here such a landmark gets intensity 0.
// img = pixel (0,0) of level 0 of the current frame, or null when no frame has been pushed (intensity 0 then).
__device__ inline void points_one(const float* __restrict__ mu, int i, const uint8_t* __restrict__ img, int pitch, int w, int h,
                                  float fx, float fy, float cx, float cy, float* __restrict__ xyz, float* __restrict__ intensity) {
    const float u = mu[EKF_BASE + 3 * i], v = mu[EKF_BASE + 3 * i + 1], rho = mu[EKF_BASE + 3 * i + 2];
    const float z = (float)(1.0 / (double)rho);
    xyz[3 * i] = u * z;
    xyz[3 * i + 1] = v * z;
    xyz[3 * i + 2] = z;
    float in = 0.f;
    if (img) {
        const float px = fx * u + cx, py = fy * v + cy;
        // cvRound; compared as floats first so that NaN / huge values never reach the conversion
        if (px >= -0.5f && px < (float)w && py >= -0.5f && py < (float)h) {
            const int ix = __float2int_rn(px), iy = __float2int_rn(py);
            if (ix >= 0 && ix < w && iy >= 0 && iy < h) in = (float)img[(size_t)iy * pitch + ix];
        }
    }
    intensity[i] = in;
}
// host_word != nullptr (single-workgroup launches only): the kernel publishes the status word itself behind its writes
__global__ void points_kernel(const float* __restrict__ mu, int N, const uint8_t* __restrict__ img, int pitch, int w, int h,
                              float fx, float fy, float cx, float cy, float* __restrict__ xyz, float* __restrict__ intensity,
                              const int* __restrict__ info, int* host_word, int seq) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) points_one(mu, i, img, pitch, w, h, fx, fy, cx, cy, xyz, intensity);
    if (host_word) {
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
            host_word[0] = info[0];
            host_word[2] = 0;
            __hip_atomic_store(host_word + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// The last kernel of a frame (ekfvio_step_image): what the node publishes after addFrame -- publishOdometry's slices of
// base_mu and publishPoints' cloud (EKFVIO.cpp:444-518) -- written into pinned host memory TOGETHER with the status word
// the host is waiting for anyway.  ekfvio_get_odometry / ekfvio_get_points then cost a memcpy instead of a launch and a
// wait each (the node's loop with outputs: 164 -> 140 us per frame at the node's defaults).  One workgroup; the number of
// landmarks is N_old plus what the replenishment just added (added_dev, may be null).
// Layout of out: base_mu[22], xyz[3 N], intensity[N].
// Pcol != null (round 4): launched BETWEEN the update's two Joseph GEMMs.  The mean update mu += K y, quaternion renormalised, is the
// second GEMM's (gemm.hip, mode 2, :600-609) from K y in column n of P, which the first one has just left there: the same sums are
// formed here (into LDS, mu itself is not touched), so the frame's outputs and status reach the host while the second GEMM still runs
// -- the host's next frame (copying the image, ~15 us) starts that much earlier.  Same values, same bits as behind the update.
__global__ __launch_bounds__(256) void frame_outputs_kernel(const float* __restrict__ mu, int N_old, const int* __restrict__ added_dev,
                                                            const uint8_t* __restrict__ img, int pitch, int w, int h, float fx, float fy,
                                                            float cx, float cy, float* __restrict__ out, const int* __restrict__ info,
                                                            int* host_word, int seq, const float* __restrict__ Pcol,
                                                            const float* __restrict__ Kyp = nullptr, int kyp_blocks = 0, int kyp_ld = 0) {
    extern __shared__ float s_mu[];  // Pcol / Kyp: the updated mean, EKF_BASE + 3 N floats
    const int added = added_dev ? *added_dev : 0;
    const int N = N_old + added;
    if (Pcol || Kyp) {
        const int n = EKF_BASE + 3 * N;
        if (Kyp) {
            // (round 6, T2 flow: K y as the gain tiles' partial sums, one row per block column, added as gemm16_finish_mean<3> adds them -- and requested as
            // it requests them: up to four elements per thread and sixteen block columns each as ONE batch from clamped addresses.  The loop with a
            // run-time bound it replaces was compiled load - wait - add: a memory round trip per block column and element, ~30 in a row at N = 256;
            // image loop at N = 256, same box: 5 622 -> 5 798 frames/s.)
            float v4[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int e = min((int)threadIdx.x + 256 * u, n - 1);
                float pk[16];
#pragma unroll
                for (int cb = 0; cb < 16; cb++) pk[cb] = Kyp[(size_t)min(cb, kyp_blocks - 1) * kyp_ld + e];
                const float m0 = mu[e];
                float ky = pk[0];
#pragma unroll
                for (int cb = 1; cb < 16; cb++) ky = (cb < kyp_blocks) ? ky + pk[cb] : ky;
                for (int cb = 16; cb < kyp_blocks; cb++) ky = ky + Kyp[(size_t)cb * kyp_ld + e];
                v4[u] = m0 + ky;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int e = (int)threadIdx.x + 256 * u;
                if (e < n) s_mu[e] = v4[u];
            }
            for (int e = threadIdx.x + 1024; e < n; e += 256) {
                float ky = Kyp[e];
                for (int cb = 1; cb < kyp_blocks; cb++) ky = ky + Kyp[(size_t)cb * kyp_ld + e];
                s_mu[e] = mu[e] + ky;
            }
        } else {
            for (int e = threadIdx.x; e < n; e += 256) s_mu[e] = mu[e] + Pcol[e];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const float q0 = s_mu[3], q1 = s_mu[4], q2 = s_mu[5], q3 = s_mu[6];
            const float qn = sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
            s_mu[3] = q0 / qn, s_mu[4] = q1 / qn, s_mu[5] = q2 / qn, s_mu[6] = q3 / qn;
        }
        __syncthreads();
        mu = s_mu;
    }
    if (threadIdx.x < EKF_BASE) out[threadIdx.x] = mu[threadIdx.x];
    float* xyz = out + EKF_BASE;
    float* inten = xyz + 3 * (size_t)N;
    for (int i = threadIdx.x; i < N; i += 256) points_one(mu, i, img, pitch, w, h, fx, fy, cx, cy, xyz, inten);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        host_word[0] = info[0];
        host_word[2] = added;
        __hip_atomic_store(host_word + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---- KLTTracker::estimateUncertaintySampleBased (KLTTracker.cpp:111-175) -------------------
// cv::getRectSubPix(8-bit, Size(5,5), center, CV_32F) (OpenCV 3.x samplers.cpp, getRectSubPix_8u32f): one patch row.
// Inside the image the horizontal interpolation is carried from pixel to pixel through a double factor
// (prev = (float)(t * s), s = (1 - a) / a); at the border: replicate-clamped bilinear taps.  `img` points at pixel
// (0,0) of level 0 (the pyramid's reflect-101 border is not what getRectSubPix sees, so coordinates are clamped).
__device__ __forceinline__ void rect_subpix5_row(const uint8_t* __restrict__ img, int pitch, int w, int h, float cx, float cy, int i,
                                                 float out[5]) {
    cx -= 2.0f;
    cy -= 2.0f;
    const int ipx = (int)floorf(cx), ipy = (int)floorf(cy);
    if (0 <= ipx && ipx + 5 < w && 0 <= ipy && ipy + 5 < h) {
        float a = cx - (float)ipx;
        const float b = cy - (float)ipy;
        a = fmaxf(a, 0.0001f);
        const float a12 = a * (1.f - b), a22 = a * b, b1 = 1.f - b, b2 = b;
        const double sc = (1. - (double)a) / (double)a;
        const uint8_t* src = img + (size_t)(ipy + i) * pitch + ipx;
        float prev = (1 - a) * (b1 * (float)src[0] + b2 * (float)src[pitch]);
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const float t = a12 * (float)src[j + 1] + a22 * (float)src[j + 1 + pitch];
            out[j] = prev + t;
            prev = (float)((double)t * sc);
        }
    } else {
        const float a = cx - (float)ipx, b = cy - (float)ipy;
        const float a11 = (1.f - a) * (1.f - b), a12 = a * (1.f - b), a21 = (1.f - a) * b, a22 = a * b;
        const int y0 = min(max(ipy + i, 0), h - 1), y1 = min(max(ipy + i + 1, 0), h - 1);
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const int x0 = min(max(ipx + j, 0), w - 1), x1 = min(max(ipx + j + 1, 0), w - 1);
            out[j] = (float)img[(size_t)y0 * pitch + x0] * a11 + (float)img[(size_t)y0 * pitch + x1] * a12 +
                     (float)img[(size_t)y1 * pitch + x0] * a21 + (float)img[(size_t)y1 * pitch + x1] * a22;
        }
    }
}

// 32 threads per point, two points per 64-thread block.  Threads 0-4 of a group form the reference patch (one row
// each) in LDS; threads 0-24 then own one of the 25 samples (du = 5*(s/5) - 10 outer, dv = 5*(s%5) - 10 inner), form
// its patch row by row and accumulate the squared differences in the reference's (i, j) order: pow(float, 2) and
// exp(float) promote to double there, the sums are float; thread 0 adds the 25 weighted samples in loop order.
__global__ __launch_bounds__(64) void klt_uncertainty_kernel(const uint8_t* __restrict__ prev, int ppitch, int pw, int ph,
                                                             const uint8_t* __restrict__ cur, int cpitch, int cw, int ch,
                                                             const float* __restrict__ ref_px, const float* __restrict__ cur_px,
                                                             int n, float* __restrict__ cov) {
    __shared__ float ref[2][25];
    __shared__ float rd_s[2][25];
    const int g = threadIdx.x >> 5, t = threadIdx.x & 31;
    const int pt = blockIdx.x * 2 + g;
    const bool live = pt < n;
    if (live && t < 5) {
        float row[5];
        rect_subpix5_row(prev, ppitch, pw, ph, ref_px[2 * pt], ref_px[2 * pt + 1], t, row);
#pragma unroll
        for (int j = 0; j < 5; j++) ref[g][t * 5 + j] = row[j];
    }
    __syncthreads();
    if (live && t < 25) {
        const float du = (float)(5 * (t / 5) - 10), dv = (float)(5 * (t % 5) - 10);
        const float sx = cur_px[2 * pt] + du, sy = cur_px[2 * pt + 1] + dv;
        float ssd = 0.f;
        for (int i = 0; i < 5; i++) {
            float row[5];
            rect_subpix5_row(cur, cpitch, cw, ch, sx, sy, i, row);
#pragma unroll
            for (int j = 0; j < 5; j++) {
                const double d = (double)(ref[g][i * 5 + j] - row[j]);
                ssd = (float)((double)ssd + d * d);
            }
        }
        ssd /= 25.f;
        rd_s[g][t] = (float)exp((double)(-0.01f * ssd));
    }
    __syncthreads();
    if (live && t == 0) {
        float sum_rd = 0.f, sum_xx = 0.f, sum_yy = 0.f, sum_xy = 0.f;
        for (int s = 0; s < 25; s++) {
            const float du = (float)(5 * (s / 5) - 10), dv = (float)(5 * (s % 5) - 10);
            const float rd = rd_s[g][s];
            sum_rd += rd;
            sum_xx += rd * du * du;
            sum_yy += rd * dv * dv;
            sum_xy += rd * du * dv;
        }
        cov[4 * pt] = sum_xx / sum_rd;
        cov[4 * pt + 1] = sum_xy / sum_rd;
        cov[4 * pt + 2] = sum_xy / sum_rd;
        cov[4 * pt + 3] = sum_yy / sum_rd;
    }
}

// KLTTracker.cpp:72-92: pass = status && inside the kill pad; z = pixel2Metric; R = 1e-5 I px^2
// scaled by (1/fx)^2 on row 0 and (1/fy)^2 on row 1.
__global__ void klt_finish_kernel(const float* __restrict__ next_px, const uint8_t* __restrict__ status, int N, int w,
                                  int h, int kill_pad, float fx, float fy, float cx, float cy, float r0, float r1,
                                  const float* __restrict__ cov_px, float s0, float s1, float* z, float* R, uint8_t* pass) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float x = next_px[2 * i], y = next_px[2 * i + 1];
    const bool p = status[i] == 1 && !(x < kill_pad || y < kill_pad || w - x < kill_pad || h - y < kill_pad);
    pass[i] = p ? 1 : 0;
    if (p) {
        z[2 * i] = (x - cx) / fx;
        z[2 * i + 1] = (y - cy) / fy;
        if (cov_px) {  // estimateUncertaintySampleBased's matrix through the same conversion (:79-84): row 0 by
                       // (1/fx)^2, row 1 by (1/fy)^2
            R[4 * i] = cov_px[4 * i] * s0;
            R[4 * i + 1] = cov_px[4 * i + 1] * s0;
            R[4 * i + 2] = cov_px[4 * i + 2] * s1;
            R[4 * i + 3] = cov_px[4 * i + 3] * s1;
        } else {
            R[4 * i] = r0;
            R[4 * i + 1] = 0.f;  // the off-diagonals of estimateUncertainty are zero, scaled or not
            R[4 * i + 2] = 0.f;
            R[4 * i + 3] = r1;
        }
    } else {
        z[2 * i] = 0.f;
        z[2 * i + 1] = 0.f;
        R[4 * i] = R[4 * i + 1] = R[4 * i + 2] = R[4 * i + 3] = 0.f;
    }
}

int level_pitch(int w) { return round_up(w + 2 * KLT_BORDER, 64); }

PyrView make_view(const KltFrame& fr) {
    PyrView v;
    v.levels = fr.levels;
    for (int l = 0; l < 4; l++) {
        v.lv[l].img = fr.img[l];
        v.lv[l].der = fr.deriv[l];
        v.lv[l].w = fr.w[l];
        v.lv[l].h = fr.h[l];
        v.lv[l].pitch = (l < fr.levels) ? level_pitch(fr.w[l]) : 0;
    }
    return v;
}

}  // namespace

int klt_level_pitch(int w) { return level_pitch(w); }
int klt_border() { return KLT_BORDER; }
void klt_intrinsics(const ekfvio_filter* f, const float* K, float* fx, float* fy, float* cx, float* cy);

// Planes of one frame for a level 0 of up to w x h pixels (every level the configuration allows).
static int frame_planes_alloc(ekfvio_filter* f, KltFrame& fr, int w, int h) {
    for (int l = 0; l < 8; l++) {
        if (fr.img[l]) (void)hipFree(fr.img[l]);
        if (fr.deriv[l]) (void)hipFree(fr.deriv[l]);
        fr.img[l] = nullptr;
        fr.deriv[l] = nullptr;
    }
    fr.cap_w = fr.cap_h = 0;
    fr.valid = false;
    int lw = w, lh = h;
    for (int l = 0; l <= f->cfg.klt_max_pyramid_level; l++) {
        // + one row of slack: the tracker's aligned 4-byte loads may touch up to 3 bytes past a row's last pixel
        const size_t px = (size_t)level_pitch(lw) * (lh + 2 * KLT_BORDER + 1);
        HIPK(f, hipMalloc((void**)&fr.img[l], px));
        HIPK(f, hipMalloc((void**)&fr.deriv[l], px * 2 * sizeof(short)));
        HIPK(f, hipMemsetAsync(fr.img[l], 0, px, f->stream));
        HIPK(f, hipMemsetAsync(fr.deriv[l], 0, px * 2 * sizeof(short), f->stream));
        lw = (lw + 1) / 2;
        lh = (lh + 1) / 2;
    }
    fr.cap_w = w;
    fr.cap_h = h;
    return EKFVIO_OK;
}

// Frame::Frame takes any image (Frame.cpp:15-42): cfg.max_image_width x max_image_height is only what is allocated up
// front.  A frame that needs more grows the upload staging, the planes of the frame slot it will be written to (the
// other slot, the previous frame, keeps its planes and its contents) and the detector's buffers; this synchronises the
// stream and allocates, once per size.
static int ensure_frame_capacity(ekfvio_filter* f, int sw, int sh, int w, int h) {
    const size_t src = (size_t)sw * sh;
    KltFrame& fr = f->frames[f->cur ^ 1];
    if (src <= f->src_cap && w <= fr.cap_w && h <= fr.cap_h && w <= f->fast_cap_w && h <= f->fast_cap_h) return EKFVIO_OK;
    HIPK(f, hipSetDevice(f->device));
    HIPK(f, hipStreamSynchronize(f->stream));
    if (src > f->src_cap) {
        if (f->staging) (void)hipFree(f->staging);
        if (f->h_image) (void)hipHostFree(f->h_image);
        f->staging = nullptr, f->h_image = nullptr, f->src_cap = 0;
        HIPK(f, hipMalloc((void**)&f->staging, src + 16));  // (+16: the upload kernel moves whole 16-byte pieces)
        HIPK(f, hipHostMalloc((void**)&f->h_image, src + 16, hipHostMallocMapped));
        f->src_cap = src;
    }
    if (w > fr.cap_w || h > fr.cap_h) {
        const int rc = frame_planes_alloc(f, fr, std::max(w, fr.cap_w), std::max(h, fr.cap_h));
        if (rc != EKFVIO_OK) return rc;
    }
    return fast_ensure(f, w, h);
}

int klt_alloc(ekfvio_filter* f) {
    const ekfvio_config& c = f->cfg;
    if (c.klt_window_size < 3 || c.klt_window_size > KLT_MAX_WIN || (c.klt_window_size & 1) == 0 ||
        c.klt_max_pyramid_level < 0 || c.klt_max_pyramid_level > 3 || c.max_image_width < 1 || c.max_image_height < 1) {
        f->last_error = "KLT config: odd window <= 21 and max level <= 3 are supported";
        return EKFVIO_EINVAL;
    }
    for (int fr = 0; fr < 2; fr++) {
        const int rc = frame_planes_alloc(f, f->frames[fr], c.max_image_width, c.max_image_height);
        if (rc != EKFVIO_OK) return rc;
    }
    const size_t maxf = f->cfg.max_features > 0 ? f->cfg.max_features : 1;
    HIPK(f, hipMalloc((void**)&f->klt_prev_px, sizeof(float) * 2 * maxf));
    HIPK(f, hipMalloc((void**)&f->klt_next_px, sizeof(float) * 2 * maxf));
    HIPK(f, hipMalloc((void**)&f->klt_status, maxf));
    HIPK(f, hipMalloc((void**)&f->klt_cov_px, sizeof(float) * 4 * maxf));
    f->src_cap = (size_t)c.max_image_width * c.max_image_height;
    HIPK(f, hipMalloc((void**)&f->staging, f->src_cap + 16));
    HIPK(f, hipHostMalloc((void**)&f->h_image, f->src_cap + 16, hipHostMallocMapped));
    return EKFVIO_OK;
}

void klt_free(ekfvio_filter* f) {
    for (int fr = 0; fr < 2; fr++)
        for (int l = 0; l < 8; l++) {
            if (f->frames[fr].img[l]) (void)hipFree(f->frames[fr].img[l]);
            if (f->frames[fr].deriv[l]) (void)hipFree(f->frames[fr].deriv[l]);
        }
    if (f->klt_prev_px) (void)hipFree(f->klt_prev_px);
    if (f->klt_next_px) (void)hipFree(f->klt_next_px);
    if (f->klt_status) (void)hipFree(f->klt_status);
    if (f->klt_cov_px) (void)hipFree(f->klt_cov_px);
    if (f->staging) (void)hipFree(f->staging);
    if (f->h_image) (void)hipHostFree(f->h_image);
}

// Device-side part of klt_push_frame: staging -> pyramid + derivatives of frames[cur]
// src: the uploaded frame (sw x sh, tightly packed); the pyramid's level 0 is w x h = (sw / s) x (sh / s)
static int build_pyramid(ekfvio_filter* f, KltFrame& fr, const uint8_t* src, int sw, int sh, int w, int h, hipStream_t st) {
    const int win = f->cfg.klt_window_size;
    fr.w[0] = w;
    fr.h[0] = h;
    fr.levels = 1;
    for (int l = 1; l <= f->cfg.klt_max_pyramid_level; l++) {
        const int lw = (fr.w[l - 1] + 1) / 2, lh = (fr.h[l - 1] + 1) / 2;
        if (lw <= win || lh <= win) break;  // buildOpticalFlowPyramid: level not larger than the window
        fr.w[l] = lw;
        fr.h[l] = lh;
        fr.levels = l + 1;
    }
    ProfScope ps(f, PC_KLT_PYRAMID);
    PyrOut o;
    o.levels = fr.levels;
    for (int l = 0; l < 4; l++) {
        o.img[l] = fr.img[l];
        o.der[l] = fr.deriv[l];
        o.w[l] = (l < fr.levels) ? fr.w[l] : 0;
        o.h[l] = (l < fr.levels) ? fr.h[l] : 0;
        o.pitch[l] = (l < fr.levels) ? level_pitch(fr.w[l]) : 0;
    }
    ResizeArgs ra;  // Frame::Frame (Frame.cpp:15-42): level 0 = cv::resize(frame, (cols / s, rows / s))
    if (w != sw || h != sh) {
        ra.on = 1;
        ra.sw = sw;
        ra.sh = sh;
        ra.scale_x = (double)sw / w;
        ra.scale_y = (double)sh / h;
    }
    hipLaunchKernelGGL(klt_pyramid_kernel, dim3((w + PYR_T - 1) / PYR_T, (h + PYR_T - 1) / PYR_T), dim3(PYR_NT), 0, st, src, sw, o, f->sweep_dbg, ra);
    return EKFVIO_OK;
}

static void intrinsics(const ekfvio_filter* f, const float* K, float* fx, float* fy, float* cx, float* cy);
void klt_intrinsics(const ekfvio_filter* f, const float* K, float* fx, float* fy, float* cx, float* cy) {
    intrinsics(f, K, fx, fy, cx, cy);
}
static void intrinsics(const ekfvio_filter* f, const float* K, float* fx, float* fy, float* cx, float* cy) {
    // CameraInfo.K is row-major [fx 0 cx; 0 fy cy; 0 0 1].  Feature.h:60-66 reads K(0), K(4)
    // and, for the offsets, K(2) and K(5) of a column-major Matrix3f = the zero entries K[2,0]
    // and K[2,1]: the principal point is ignored by the reference (self-consistently).
    *fx = K[0];
    *fy = K[4];
    *cx = f->cfg.use_principal_point ? K[2] : 0.f;
    *cy = f->cfg.use_principal_point ? K[5] : 0.f;
}

static int track_points_device(ekfvio_filter* f, int n, const TrackFuse& tf = TrackFuse()) {
    const KltFrame& prev = f->frames[f->cur ^ 1];
    const KltFrame& cur = f->frames[f->cur];
    const float eps = f->cfg.klt_epsilon;
    ProfScope ps(f, PC_KLT_TRACK);
    hipLaunchKernelGGL(klt_track_kernel, dim3(n), dim3(64), 0, f->stream, make_view(prev), make_view(cur), f->klt_prev_px,
                       f->klt_next_px, f->klt_status, n, f->cfg.klt_window_size, f->cfg.klt_max_iterations, eps * eps,
                       f->cfg.klt_min_eigen, f->sweep_dbg, tf);
    return EKFVIO_OK;
}

// Pixel covariances of the n points in f->klt_prev_px (previous frame) / f->klt_next_px (current frame) -> f->klt_cov_px.
static void uncertainty_device(ekfvio_filter* f, int n) {
    const KltFrame& prev = f->frames[f->cur ^ 1];
    const KltFrame& cur = f->frames[f->cur];
    const int pp = level_pitch(prev.w[0]), cp = level_pitch(cur.w[0]);
    hipLaunchKernelGGL(klt_uncertainty_kernel, dim3((n + 1) / 2), dim3(64), 0, f->stream,
                       prev.img[0] + (size_t)KLT_BORDER * pp + KLT_BORDER, pp, prev.w[0], prev.h[0],
                       cur.img[0] + (size_t)KLT_BORDER * cp + KLT_BORDER, cp, cur.w[0], cur.h[0], f->klt_prev_px, f->klt_next_px, n,
                       f->klt_cov_px);
}

// Runs the tracker for the current landmarks; results land in f->zmeas / Rmeas / pass (device).
int klt_track_device(ekfvio_filter* f) {
    const KltFrame& prev = f->frames[f->cur ^ 1];
    const KltFrame& cur = f->frames[f->cur];
    if (!prev.valid || !cur.valid) {
        f->last_error = "KLT needs two frames";
        return EKFVIO_ESTATE;
    }
    const int N = f->N;
    if (N == 0) return EKFVIO_OK;
    float fxp, fyp, cxp, cyp, fxc, fyc, cxc, cyc;
    intrinsics(f, prev.K, &fxp, &fyp, &cxp, &cyp);
    intrinsics(f, cur.K, &fxc, &fyc, &cxc, &cyc);
    // estimateUncertainty (:100-106) = 1e-5 I; scale = pow(1.0/K(0,0), 2) in double, narrowed
    const float r0 = 0.00001f * (float)pow(1.0 / (double)cur.K[0], 2);
    const float r1 = 0.00001f * (float)pow(1.0 / (double)cur.K[4], 2);
    const float s0 = (float)pow(1.0 / (double)cur.K[0], 2), s1 = (float)pow(1.0 / (double)cur.K[4], 2);
    TrackFuse tf;
    tf.last_klt = f->last_klt;
    tf.mu = f->mu;
    tf.fxp = fxp, tf.fyp = fyp, tf.cxp = cxp, tf.cyp = cyp, tf.fxc = fxc, tf.fyc = fyc, tf.cxc = cxc, tf.cyc = cyc;
    tf.from_state = 1;
    tf.finish = f->cfg.sample_based_uncertainty ? 0 : 1;
    tf.w = cur.w[0], tf.h = cur.h[0], tf.kill_pad = f->cfg.kill_pad;
    tf.r0 = r0, tf.r1 = r1;
    tf.z = f->zmeas, tf.R = f->Rmeas, tf.pass = f->pass;
    track_points_device(f, N, tf);
    if (f->cfg.sample_based_uncertainty) {  // estimateUncertaintySampleBased(lf, prev_fts[i], cf, new_fts[i])
        uncertainty_device(f, N);
        hipLaunchKernelGGL(klt_finish_kernel, dim3((N + 255) / 256), dim3(256), 0, f->stream, f->klt_next_px, f->klt_status, N,
                           cur.w[0], cur.h[0], f->cfg.kill_pad, fxc, fyc, cxc, cyc, r0, r1, f->klt_cov_px, s0, s1, f->zmeas,
                           f->Rmeas, f->pass);
    }
    return EKFVIO_OK;
}

// Frame ingest without a host wait: the caller's image is copied into the handle's pinned staging buffer (so the
// caller may reuse its buffer on return), then H2D, resize and pyramid are enqueued.  The pinned buffer is rewritten by
// the next frame, so the previous frame's H2D copy must have completed by then: the callers below synchronise the
// stream before they return.
// (Measured and dropped: the upload and the pyramid on a stream of their own beside process(dt), joined by an event in
// front of the tracker.  The two cross-stream waits cost more than the 12 us of overlap they buy: 161 instead of 141 us
// per frame at N = 64.)
// The frame's trip to device memory: a kernel reading the pinned (mapped) host buffer, 16 bytes per lane.  The copy engine's
// hipMemcpyAsync took 12.6 us for a 640 x 480 frame (rocprofv3), this ~6: the image loop gained 11 - 13 us per frame (same-box A/B,
// N = 256 with outputs 198 -> 186 us, node defaults 136 -> 123 us).  EKFVIO_UPLOAD_KERNEL=0 brings the copy engine back.
__global__ __launch_bounds__(256) void upload_frame_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int n16) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}

static int push_frame_check(ekfvio_filter* f, const uint8_t* image, int32_t width, int32_t height, int32_t stride,
                            const float K[9]) {
    if (!f || !image || !K || width < 1 || height < 1 || stride < width) return EKFVIO_EINVAL;
    if (width > 16384 || height > 16384) return EKFVIO_ECAPACITY;  // keypoints carry x in 16 bits (fast.hip)
    const int s = f->cfg.inverse_image_scale > 1 ? f->cfg.inverse_image_scale : 1;
    if (width / s < 1 || height / s < 1) return EKFVIO_EINVAL;
    return ensure_frame_capacity(f, width, height, width / s, height / s);
}
static int push_frame_enqueue(ekfvio_filter* f, const uint8_t* image, int32_t width, int32_t height, int32_t stride,
                              const float K[9]) {
    const int rcc = push_frame_check(f, image, width, height, stride, K);
    if (rcc != EKFVIO_OK) return rcc;
    // Frame::Frame (Frame.cpp:15-42): cv::resize to (cols / s, rows / s), K(0,0), K(0,2), K(1,1), K(1,2) divided by s
    const int s = f->cfg.inverse_image_scale > 1 ? f->cfg.inverse_image_scale : 1;
    const int w = width / s, h = height / s;
    HIPK(f, hipSetDevice(f->device));
    if (stride == width) {
        memcpy(f->h_image, image, (size_t)width * height);
    } else {
        for (int y = 0; y < height; y++) memcpy(f->h_image + (size_t)y * width, image + (size_t)y * stride, width);
    }
    hipStream_t st = f->stream;
    const int upload_kernel = f->upload_kernel;  // (per handle, read at create)
    if (upload_kernel) {
        const int n16 = (int)(((size_t)width * height + 15) / 16);
        void* dsrc = nullptr;  // the device's address of the mapped buffer (the same pointer under unified addressing; asked for, not assumed)
        HIPK(f, hipHostGetDevicePointer(&dsrc, f->h_image, 0));
        hipLaunchKernelGGL(upload_frame_kernel, dim3((n16 + 255) / 256), dim3(256), 0, st, reinterpret_cast<const uint4*>(dsrc),
                           reinterpret_cast<uint4*>(f->staging), n16);
    } else {
        HIPK(f, hipMemcpyAsync(f->staging, f->h_image, (size_t)width * height, hipMemcpyHostToDevice, st));
    }
    f->cur ^= 1;  // the former current frame becomes the previous one (frame_buffer depth 2)
    KltFrame& fr = f->frames[f->cur];
    for (int i = 0; i < 9; i++) fr.K[i] = K[i];
    if (s > 1) {
        fr.K[0] = (float)((double)K[0] / s);
        fr.K[2] = (float)((double)K[2] / s);
        fr.K[4] = (float)((double)K[4] / s);
        fr.K[5] = (float)((double)K[5] / s);
    }
    build_pyramid(f, fr, f->staging, width, height, w, h, st);
    fr.valid = true;
    HIPK(f, hipGetLastError());
    return EKFVIO_OK;
}

extern "C" {

int ekfvio_klt_push_frame(ekfvio_filter* f, const uint8_t* image, int32_t width, int32_t height, int32_t stride,
                          const float K[9]) {
    if (f) f->out_fresh = false;
    const int rc = push_frame_enqueue(f, image, width, height, stride, K);
    if (rc != EKFVIO_OK) return rc;
    HIPK(f, hipStreamSynchronize(f->stream));
    return EKFVIO_OK;
}

// publishPoints' payload (EKFVIO.cpp:479-518), formed on the device: xyz3N = (u/rho, v/rho, 1/rho) per landmark in the
// camera frame, intensityN = the current (resized) frame's byte at the landmark's pixel (0 outside the image or before
// the first frame).  Either pointer may be NULL.
int ekfvio_get_points(ekfvio_filter* f, float* xyz3N, float* intensityN) {
    if (!f) return EKFVIO_EINVAL;
    const int N = f->N;
    if (N == 0) return EKFVIO_OK;
    if (f->out_fresh) {  // the frame's last kernel wrote them with the status word (frame_outputs_kernel)
        if (xyz3N) memcpy(xyz3N, f->h_out + EKF_BASE, sizeof(float) * 3 * N);
        if (intensityN) memcpy(intensityN, f->h_out + EKF_BASE + 3 * (size_t)N, sizeof(float) * N);
        return EKFVIO_OK;
    }
    HIPK(f, hipSetDevice(f->device));
    const KltFrame& fr = f->frames[f->cur];
    const uint8_t* img = nullptr;
    int pitch = 0, w = 0, h = 0;
    float fx = 0.f, fy = 0.f, cx = 0.f, cy = 0.f;
    if (fr.valid) {
        pitch = level_pitch(fr.w[0]);
        img = fr.img[0] + (size_t)KLT_BORDER * pitch + KLT_BORDER;
        w = fr.w[0];
        h = fr.h[0];
        intrinsics(f, fr.K, &fx, &fy, &cx, &cy);
    }
    // the kernel writes into pinned host memory (behind the 22 floats ekfvio_get_base_mu uses): xyz, then the intensities
    float* d_xyz = f->d_out + EKF_BASE;
    float* d_int = d_xyz + 3 * (size_t)N;
    int bad = 0, rc;
    if (N <= 256) {  // one workgroup: it publishes the status word itself
        const int seq = next_status_seq(f);
        hipLaunchKernelGGL(points_kernel, dim3(1), dim3(256), 0, f->stream, f->mu, N, img, pitch, w, h, fx, fy, cx, cy, d_xyz, d_int,
                           f->info, f->d_hinfo, seq);
        rc = poll_status(f, seq, &bad, nullptr);
    } else {
        hipLaunchKernelGGL(points_kernel, dim3((N + 255) / 256), dim3(256), 0, f->stream, f->mu, N, img, pitch, w, h, fx, fy, cx, cy,
                           d_xyz, d_int, nullptr, nullptr, 0);
        rc = wait_status(f, &bad);
    }
    if (rc != EKFVIO_OK) return rc;
    if (xyz3N) memcpy(xyz3N, f->h_out + EKF_BASE, sizeof(float) * 3 * N);
    if (intensityN) memcpy(intensityN, f->h_out + EKF_BASE + 3 * (size_t)N, sizeof(float) * N);
    return EKFVIO_OK;
}

int ekfvio_klt_track(ekfvio_filter* f, float* z2N, float* R4N, uint8_t* passN) {
    if (!f) return EKFVIO_EINVAL;
    HIPK(f, hipSetDevice(f->device));
    int rc = klt_track_device(f);
    if (rc != EKFVIO_OK) return rc;
    const int N = f->N;
    if (N > 0) {
        if (z2N) HIPK(f, hipMemcpyAsync(z2N, f->zmeas, sizeof(float) * 2 * N, hipMemcpyDeviceToHost, f->stream));
        if (R4N) HIPK(f, hipMemcpyAsync(R4N, f->Rmeas, sizeof(float) * 4 * N, hipMemcpyDeviceToHost, f->stream));
        if (passN) HIPK(f, hipMemcpyAsync(passN, f->pass, N, hipMemcpyDeviceToHost, f->stream));
    }
    HIPK(f, hipGetLastError());
    HIPK(f, hipStreamSynchronize(f->stream));
    return EKFVIO_OK;
}

int ekfvio_klt_track_points(ekfvio_filter* f, const float* prev_px, const float* init_px, int32_t count, float* out_px,
                            uint8_t* status) {
    if (!f || !prev_px || !init_px || count < 0) return EKFVIO_EINVAL;
    if (count > f->cfg.max_features) return EKFVIO_ECAPACITY;
    if (!f->frames[0].valid || !f->frames[1].valid) {
        f->last_error = "KLT needs two frames";
        return EKFVIO_ESTATE;
    }
    if (count == 0) return EKFVIO_OK;
    HIPK(f, hipSetDevice(f->device));
    HIPK(f, hipMemcpyAsync(f->klt_prev_px, prev_px, sizeof(float) * 2 * count, hipMemcpyHostToDevice, f->stream));
    HIPK(f, hipMemcpyAsync(f->klt_next_px, init_px, sizeof(float) * 2 * count, hipMemcpyHostToDevice, f->stream));
    track_points_device(f, count);
    if (out_px) HIPK(f, hipMemcpyAsync(out_px, f->klt_next_px, sizeof(float) * 2 * count, hipMemcpyDeviceToHost, f->stream));
    if (status) HIPK(f, hipMemcpyAsync(status, f->klt_status, count, hipMemcpyDeviceToHost, f->stream));
    HIPK(f, hipGetLastError());
    HIPK(f, hipStreamSynchronize(f->stream));
    return EKFVIO_OK;
}

int ekfvio_klt_uncertainty_points(ekfvio_filter* f, const float* ref_px, const float* cur_px, int32_t count, float* cov4) {
    if (!f || !ref_px || !cur_px || !cov4 || count < 0) return EKFVIO_EINVAL;
    if (count > f->cfg.max_features) return EKFVIO_ECAPACITY;
    if (!f->frames[0].valid || !f->frames[1].valid) {
        f->last_error = "the sample-based uncertainty needs two frames";
        return EKFVIO_ESTATE;
    }
    if (count == 0) return EKFVIO_OK;
    HIPK(f, hipSetDevice(f->device));
    HIPK(f, hipMemcpyAsync(f->klt_prev_px, ref_px, sizeof(float) * 2 * count, hipMemcpyHostToDevice, f->stream));
    HIPK(f, hipMemcpyAsync(f->klt_next_px, cur_px, sizeof(float) * 2 * count, hipMemcpyHostToDevice, f->stream));
    uncertainty_device(f, count);
    HIPK(f, hipMemcpyAsync(cov4, f->klt_cov_px, sizeof(float) * 4 * count, hipMemcpyDeviceToHost, f->stream));
    HIPK(f, hipGetLastError());
    HIPK(f, hipStreamSynchronize(f->stream));
    return EKFVIO_OK;
}

// Test hook: copies pyramid level `level` of the current frame (interior only) to the host.
int ekfvio_klt_get_level(ekfvio_filter* f, int32_t level, int32_t* w, int32_t* h, uint8_t* img, int16_t* deriv) {
    if (!f || level < 0) return EKFVIO_EINVAL;
    const KltFrame& fr = f->frames[f->cur];
    if (!fr.valid || level >= fr.levels) return EKFVIO_ESTATE;
    HIPK(f, hipSetDevice(f->device));
    const int lw = fr.w[level], lh = fr.h[level], pitch = level_pitch(lw);
    if (w) *w = lw;
    if (h) *h = lh;
    if (img)
        HIPK(f, hipMemcpy2DAsync(img, lw, fr.img[level] + (size_t)KLT_BORDER * pitch + KLT_BORDER, pitch, lw, lh,
                                 hipMemcpyDeviceToHost, f->stream));
    if (deriv)
        HIPK(f, hipMemcpy2DAsync(deriv, (size_t)lw * 4, fr.deriv[level] + ((size_t)KLT_BORDER * pitch + KLT_BORDER) * 2,
                                 (size_t)pitch * 4, (size_t)lw * 4, lh, hipMemcpyDeviceToHost, f->stream));
    HIPK(f, hipStreamSynchronize(f->stream));
    return EKFVIO_OK;
}

#ifdef EKFVIO_TEST_HOOKS  // include/ekfvio_test_hooks.h: only in libekfvio_hip_hooks.so
int ekfvio_test_klt_padded_level(ekfvio_filter* f, int32_t level, int32_t* border, uint8_t* img, int16_t* deriv) {
    if (!f || level < 0) return EKFVIO_EINVAL;
    const KltFrame& fr = f->frames[f->cur];
    if (!fr.valid || level >= fr.levels) return EKFVIO_ESTATE;
    HIPK(f, hipSetDevice(f->device));
    const int pw = fr.w[level] + 2 * KLT_BORDER, ph = fr.h[level] + 2 * KLT_BORDER, pitch = level_pitch(fr.w[level]);
    if (border) *border = KLT_BORDER;
    if (img) HIPK(f, hipMemcpy2DAsync(img, pw, fr.img[level], pitch, pw, ph, hipMemcpyDeviceToHost, f->stream));
    if (deriv)
        HIPK(f, hipMemcpy2DAsync(deriv, (size_t)pw * 4, fr.deriv[level], (size_t)pitch * 4, (size_t)pw * 4, ph, hipMemcpyDeviceToHost,
                                 f->stream));
    HIPK(f, hipStreamSynchronize(f->stream));
    return EKFVIO_OK;
}
#endif

// EKFVIO::addFrame + updateStateWithNewImage (EKFVIO.cpp:139-219) minus ROS publishing; with
// cfg.replenish the FAST replenishment of :154 / :172 runs on the device too (otherwise the caller
// adds landmarks with ekfvio_add_features).
int ekfvio_step_image(ekfvio_filter* f, double stamp, const uint8_t* image, int32_t width, int32_t height, int32_t stride,
                      const float K[9]) {
    if (!f) return EKFVIO_EINVAL;
    f->out_fresh = false;
    const bool first = !f->frames[f->cur].valid;
    if (!first && f->have_stamp && !(stamp - f->t_stamp >= 0)) return EKFVIO_EINVAL;  // ROS_ASSERT(dt >= 0)
    // nothing below waits for the device until the status word is read at the very end: process(dt), the frame
    // upload, the pyramid, the tracker and the update are enqueued back to back.  process(dt) goes first: it does not
    // depend on the image, so the device runs it while the host copies the frame into the pinned buffer.
    // everything that can refuse the frame (sizes, capacity growth) comes before the first launch
    int rc = push_frame_check(f, image, width, height, stride, K);
    if (rc != EKFVIO_OK) return rc;
    sweep_maybe_retry(f);
    const float dt = first ? 0.f : (float)(stamp - f->t_stamp);
    // an error return from here on has work enqueued behind it: wait for it (the pinned frame buffer is rewritten by the
    // next call) and leave the stamp with the state it belongs to
    auto fail = [&](int code) {
        (void)hipStreamSynchronize(f->stream);
        return code;
    };
    if (!first) {
        HIPK(f, hipSetDevice(f->device));
        launch_predict(f, dt);
        f->t_stamp = stamp;  // tc_ekf.t = f.t (:164): the state now stands at this stamp whatever happens below
        f->have_stamp = true;
    }
    rc = push_frame_enqueue(f, image, width, height, stride, K);
    if (rc != EKFVIO_OK) return fail(rc);
    if (first) {
        // first frame: remember the stamp (tc_ekf.t = f.t) and return; the caller replenishes
        if (!f->have_stamp) {
            f->t_stamp = stamp;
            f->have_stamp = true;
        }
        if (f->cfg.replenish) return ekfvio_replenish(f, nullptr, nullptr);  // replenishFeatures(first frame) (:154)
        HIPK(f, hipStreamSynchronize(f->stream));
        return EKFVIO_OK;
    }
    int status = EKFVIO_OK;
    int early_seq = 0;
    if (f->N > 0) {  // "run update if we have enough features" (EKFVIO.cpp:166)
        rc = klt_track_device(f);
        if (rc != EKFVIO_OK) return fail(rc);
        // the pass flags stay on the device: the update is launched for m = 2N measurement rows and its kernels take
        // the true count from the bookkeeping (rows beyond it are identity padding, exact zeros in every product)
        // A frame that adds no landmarks publishes its outputs and its status BETWEEN the update's two Joseph GEMMs (frame_outputs_kernel,
        // Pcol): launch_update calls back there.  (With landmarks to add the selection reads the updated mean and the outputs carry the count:
        // behind the update, as before.  EKFVIO_EARLY_OUTPUTS=0: always behind.)
        if (f->early_outputs && f->frame_outputs && !(f->cfg.replenish && f->N < f->cfg.max_features) &&
            sizeof(float) * (size_t)f->n <= 48 * 1024) {  // (the updated mean is formed in LDS)
            f->between_joseph = [](ekfvio_filter* g) {
                const KltFrame& fr = g->frames[g->cur];
                const int pitch = level_pitch(fr.w[0]);
                float fx, fy, cx, cy;
                intrinsics(g, fr.K, &fx, &fy, &cx, &cy);
                g->between_joseph_seq = next_status_seq(g);
                // (two-GEMM flow: K y is column n of P, left there by the first Joseph GEMM; T2 flow: the gain tiles' partial sums in Wt, and the
                // outputs go out in front of the update's ONE GEMM)
                const bool kyp = g->hook_kyp_blocks > 0;
                hipLaunchKernelGGL(frame_outputs_kernel, dim3(1), dim3(256), sizeof(float) * (size_t)g->n, g->stream, g->mu, g->N, (const int*)nullptr,
                                   fr.img[0] + (size_t)KLT_BORDER * pitch + KLT_BORDER, pitch, fr.w[0], fr.h[0], fx, fy, cx, cy, g->d_out, g->info,
                                   g->d_hinfo, g->between_joseph_seq, kyp ? (const float*)nullptr : (const float*)(g->P + (size_t)g->n * g->ldp),
                                   kyp ? (const float*)g->Wt : (const float*)nullptr, g->hook_kyp_blocks, g->ldp);
            };
            f->between_joseph_seq = 0;
        }
        launch_update(f, 0, f->zmeas, f->Rmeas, f->pass, nullptr, 0, false, true);
        f->between_joseph = nullptr;
        early_seq = f->between_joseph_seq;  // 0: the hook was not reached (no measurement, another flow of the update)
        if (early_seq) f->early_output_frames++;
        f->between_joseph_seq = 0;
    }
    if (hipGetLastError() != hipSuccess) {
        f->last_error = "launch failed in ekfvio_step_image";
        return fail(EKFVIO_EDEVICE);
    }
    // "try to get more features if needed" (:172): detection, first-fit selection and the growth of the state are
    // enqueued behind the update with the number of new landmarks left on the device; the host reads it with the status
    // word, in the frame's single wait, and only then counts the landmarks in
    int replenishing = 0;
    if (f->cfg.replenish) {
        rc = replenish_enqueue(f, &replenishing);
        if (rc != EKFVIO_OK) return fail(rc);
        if (replenishing) add_features_enqueue_device_count(f, f->fast_counts + 1);
    }
    // the frame's one wait: the status word, the number of new landmarks and what the node publishes after addFrame
    // (odometry, point cloud) arrive together in pinned host memory (frame_outputs_kernel)
    int bad = 0, added = 0;
    if (!f->frame_outputs) {
        rc = wait_status(f, &bad, replenishing ? f->fast_counts + 1 : nullptr, &added);
    } else {
        const KltFrame& fr = f->frames[f->cur];
        const int pitch = level_pitch(fr.w[0]);
        float fx, fy, cx, cy;
        intrinsics(f, fr.K, &fx, &fy, &cx, &cy);
        if (early_seq) {  // (the outputs went out between the Joseph GEMMs: launch_update's hook)
            rc = poll_status(f, early_seq, &bad, &added);
        } else {
            const int seq = next_status_seq(f);
            hipLaunchKernelGGL(frame_outputs_kernel, dim3(1), dim3(256), 0, f->stream, f->mu, f->N, replenishing ? f->fast_counts + 1 : nullptr,
                               fr.img[0] + (size_t)KLT_BORDER * pitch + KLT_BORDER, pitch, fr.w[0], fr.h[0], fx, fy, cx, cy, f->d_out, f->info,
                               f->d_hinfo, seq, (const float*)nullptr);
            rc = poll_status(f, seq, &bad, &added);
        }
    }
    if (rc != EKFVIO_OK) return rc;
    f->out_fresh = f->frame_outputs != 0;
    if (bad) HIPK(f, hipMemsetAsync(f->info, 0, sizeof(int), f->stream));
    if (bad & 2) {
        // The persistent sweep gave up (chol_persist.inc): the Joseph GEMMs wrote nothing, the state is the propagated one.
        // The update runs again now, with one launch per block step, over the landmarks it was enqueued for (the count is
        // bumped below): idx, the measured coordinates per row, R and the row count are where the bookkeeping left them.
        // Landmarks the replenishment has added meanwhile lie outside n, with ONE exception the re-run relies on an invariant for
        // (ADVICE r04): the first Joseph GEMM writes n + 1 columns -- column n carries K y -- and the second zeroes rows 0 .. n-1 of
        // column n again; column n is now the first new landmark's column.  addNewFeatures leaves a new landmark's cross-covariances
        // exactly zero and its measurement-map entries at -1, so "zero before, K y in between, zero after" is what an update without
        // new landmarks does to that padding column too (tests/test_gpu_sweep_abort.py, the replenishing-frame case).  (Those
        // landmarks were picked around the predicted, not the updated, landmark pixels: a valid state, not bit for bit the
        // frame an unshared GPU produces.)
        sweep_abort_latch(f);
        f->sweep_recoveries++;
        f->out_fresh = false;
        launch_update(f, 0, f->zmeas, f->Rmeas, f->pass, nullptr, 0, true, true);
        HIPK(f, hipGetLastError());
        rc = wait_status(f, &bad);
        if (rc != EKFVIO_OK) return rc;
        if (bad) HIPK(f, hipMemsetAsync(f->info, 0, sizeof(int), f->stream));
        if (bad & 2) return EKFVIO_EABORTED;
    } else if (f->N > 0) sweep_clean_update(f);
    if (added > 0) {
        f->N += added;
        f->n += 3 * added;
    }
    if (bad & 1) status = EKFVIO_ENUMERIC;
    return status;
}

}  // extern "C"
