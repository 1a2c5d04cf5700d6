// ekf_vio_amd/csrc/fast.hip — landmark replenishment on the device.  (Frame::Frame's resize, F2, is part
// of the pyramid kernel: klt.hip, resize_pixel.)
//
// Reference:
//   EKFVIO::replenishFeatures (include/ekf_vio/EKFVIO.cpp:224-311): cv::FAST(img, kp, FAST_THRESHOLD, true),
//     occupancy image of filled circles (radius MIN_NEW_FEATURE_DIST) around the landmarks' pixels, first fit
//     over the keypoints in detector (raster) order, kill-box test, addNewFeatures(pixel2Metric(...)).
// The arithmetic is OpenCV 3.x's (fast.cpp / fast_score.cpp TYPE_9_16, drawing.cpp Circle): everything is integer, so the kernels are bit-exact against oracle/fast_oracle.cpp.
//
// Mapping: detection is two launches: a workgroup per image row scores the row and its two neighbours in LDS (the 16
// ring pixels become two 16-bit masks, "9 contiguous" is four shift-ands on the doubled mask), suppresses non-maxima and
// appends the survivors in x order to the row's own list; one workgroup then scans the row counts and copies the lists in
// row order, which keeps cv::FAST's keypoint order without a sort.  The first-fit selection is inherently sequential
// in keypoint order: one wavefront takes 64 keypoints at a time, tests them against the occupancy mask in parallel,
// accepts the first free one, stamps its circle (one lane per circle row, bit mask in LDS) and re-tests the rest; four
// wavefronts prepare the mask (existing landmarks' circles) before that.
#include "common.h"

#define HIPF(f, expr)                                                              \
    do {                                                                           \
        hipError_t e_ = (expr);                                                    \
        if (e_ != hipSuccess) {                                                    \
            (f)->last_error = std::string(#expr) + ": " + hipGetErrorString(e_);   \
            return EKFVIO_EDEVICE;                                                 \
        }                                                                          \
    } while (0)

namespace {

// FAST-9/16 segment test + cornerScore<16> at pixel (x, y): -1 = no corner (or inside the 3-pixel frame cv::FAST skips).
// img points at pixel (0,0) of a pitched image.
__device__ inline short fast_score_at(const uint8_t* __restrict__ img, int w, int h, int pitch, int threshold, int x, int y) {
    short out = -1;
    if (x >= 3 && y >= 3 && x < w - 3 && y < h - 3) {
        const uint8_t* p = img + (size_t)y * pitch + x;
        const int v = p[0];
        // ring in cv::FAST's order (fast.cpp makeOffsets, patternSize 16)
        int q[16];
        q[0] = p[3 * pitch];      q[1] = p[3 * pitch + 1];   q[2] = p[2 * pitch + 2];   q[3] = p[pitch + 3];
        q[4] = p[3];              q[5] = p[-pitch + 3];      q[6] = p[-2 * pitch + 2];  q[7] = p[-3 * pitch + 1];
        q[8] = p[-3 * pitch];     q[9] = p[-3 * pitch - 1];  q[10] = p[-2 * pitch - 2]; q[11] = p[-pitch - 3];
        q[12] = p[-3];            q[13] = p[pitch - 3];      q[14] = p[2 * pitch - 2];  q[15] = p[3 * pitch - 1];
        unsigned br = 0, dk = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            br |= (q[k] > v + threshold ? 1u : 0u) << k;
            dk |= (q[k] < v - threshold ? 1u : 0u) << k;
        }
        auto run9 = [](unsigned m16) {
            const unsigned m = m16 | (m16 << 16);
            unsigned t = m & (m >> 1);   // runs >= 2
            t &= t >> 2;                 // >= 4
            t &= t >> 4;                 // >= 8
            return (t & (m >> 8)) != 0;  // >= 9
        };
        if (run9(br) || run9(dk)) {
            int d[25];
#pragma unroll
            for (int k = 0; k < 25; k++) d[k] = v - q[k & 15];
            int a0 = threshold;
#pragma unroll
            for (int k = 0; k < 16; k += 2) {
                int a = min(d[k + 1], d[k + 2]);
                a = min(a, d[k + 3]);
                a = min(a, d[k + 4]);
                a = min(a, d[k + 5]);
                a = min(a, d[k + 6]);
                a = min(a, d[k + 7]);
                a = min(a, d[k + 8]);
                a0 = max(a0, min(a, d[k]));
                a0 = max(a0, min(a, d[k + 9]));
            }
            int b0 = -a0;
#pragma unroll
            for (int k = 0; k < 16; k += 2) {
                int b = max(d[k + 1], d[k + 2]);
                b = max(b, d[k + 3]);
                b = max(b, d[k + 4]);
                b = max(b, d[k + 5]);
                b = max(b, d[k + 6]);
                b = max(b, d[k + 7]);
                b = max(b, d[k + 8]);
                b0 = min(b0, max(b, d[k]));
                b0 = min(b0, max(b, d[k + 9]));
            }
            out = (short)(-b0 - 1);
        }
    }
    return out;
}

// Detection in two launches (was four: score map, keypoints per row, scan, ordered write).
// fast_rows_kernel: one workgroup per image row.  In chunks of 256 columns it scores the row and its two neighbours into
// LDS (each row's scores are formed three times over the grid: cheaper than a launch and a score map in HBM), applies the
// non-maximum suppression (strict > against the 8 neighbours, fast.cpp) and appends the survivors, in x order, to the
// row's own list (x and score packed into one word) with a ballot prefix; row_cnt[y] = their number.
// fast_gather_kernel: one workgroup scans the row counts and copies the rows' lists, in row order, into the keypoint
// arrays: cv::FAST's raster order without a sort.
#define FAST_CHUNK 256
__global__ __launch_bounds__(FAST_CHUNK) void fast_rows_kernel(const uint8_t* __restrict__ img, int w, int h, int pitch, int threshold,
                                                               int nonmax, unsigned* __restrict__ row_kp, int* __restrict__ row_cnt) {
    __shared__ short sc[3][FAST_CHUNK + 2];
    __shared__ int s_w[FAST_CHUNK / 64];
    __shared__ int s_base;
    const int y = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (y < 3 || y >= h - 3) {  // cv::FAST's frame: no keypoints
        if (tid == 0) row_cnt[y] = 0;
        return;
    }
    if (tid == 0) s_base = 0;
    unsigned* out = row_kp + (size_t)y * w;
    for (int x0 = 0; x0 < w; x0 += FAST_CHUNK) {
        __syncthreads();  // the previous chunk's readers are done; s_base is visible
        // columns x0 - 1 .. x0 + FAST_CHUNK of rows y - 1, y, y + 1 (the two extra columns by the first two threads)
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const int x = x0 + tid;
            sc[r][tid + 1] = (x < w) ? fast_score_at(img, w, h, pitch, threshold, x, y - 1 + r) : (short)-1;
        }
        if (tid < 2) {
            const int x = tid == 0 ? x0 - 1 : x0 + FAST_CHUNK;
#pragma unroll
            for (int r = 0; r < 3; r++)
                sc[r][tid == 0 ? 0 : FAST_CHUNK + 1] = (x >= 0 && x < w) ? fast_score_at(img, w, h, pitch, threshold, x, y - 1 + r) : (short)-1;
        }
        __syncthreads();
        const int x = x0 + tid;
        const short s = sc[1][tid + 1];
        bool keep = x >= 3 && x < w - 3 && s >= 0;
        if (keep && nonmax)
            keep = s > sc[0][tid] && s > sc[0][tid + 1] && s > sc[0][tid + 2] && s > sc[1][tid] && s > sc[1][tid + 2] && s > sc[2][tid] &&
                   s > sc[2][tid + 1] && s > sc[2][tid + 2];
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) s_w[wv] = __popcll(bal);
        __syncthreads();
        int before = 0, tot = 0;
#pragma unroll
        for (int q = 0; q < FAST_CHUNK / 64; q++) {
            before += (q < wv) ? s_w[q] : 0;
            tot += s_w[q];
        }
        const int base = s_base;
        if (keep) out[base + before + __popcll(bal & ((1ull << lane) - 1ull))] = ((unsigned)(unsigned short)s << 16) | (unsigned)x;
        __syncthreads();
        if (tid == 0) s_base = base + tot;
    }
    __syncthreads();
    if (tid == 0) row_cnt[y] = s_base;
}
// the row lists copied, in row order, into the keypoint arrays; NT threads of one workgroup
template <int NT>
__device__ inline void fast_gather_body(const unsigned* __restrict__ row_kp, const int* __restrict__ row_cnt, int w, int h, int cap,
                                        int* __restrict__ kp_xy, short* __restrict__ kp_score, int* total, int* s_w, int* s_carry) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) *s_carry = 0;
    __syncthreads();
    for (int y0 = 0; y0 < h; y0 += NT) {
        const int y = y0 + tid;
        const int c = (y < h) ? row_cnt[y] : 0;
        int incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
        }
        if (lane == 63) s_w[wv] = incl;
        __syncthreads();
        int before = 0, tot = 0;
#pragma unroll
        for (int q = 0; q < NT / 64; q++) {
            before += (q < wv) ? s_w[q] : 0;
            tot += s_w[q];
        }
        const int carry = *s_carry;
        const int off0 = carry + before + incl - c;  // this row's first slot
        const unsigned* src = row_kp + (size_t)y * w;
        for (int i = 0; i < c; i++) {
            const int o = off0 + i;
            if (o < cap) {
                const unsigned v = src[i];
                kp_xy[2 * o] = (int)(v & 0xffffu);
                kp_xy[2 * o + 1] = y;
                kp_score[o] = (short)(v >> 16);
            }
        }
        __syncthreads();
        if (tid == 0) *s_carry = carry + tot;
        __syncthreads();
    }
    if (tid == 0) *total = *s_carry;
}
__global__ __launch_bounds__(1024) void fast_gather_kernel(const unsigned* __restrict__ row_kp, const int* __restrict__ row_cnt, int w, int h,
                                                           int cap, int* __restrict__ kp_xy, short* __restrict__ kp_score, int* total) {
    __shared__ int s_w[16];
    __shared__ int s_carry;
    fast_gather_body<1024>(row_kp, row_cnt, w, h, cap, kp_xy, kp_score, total, s_w, &s_carry);
}

// drawing.cpp Circle(), filled: half-width of the span in row cy + dyrow (-1: the row is outside the circle)
__device__ inline int circle_half_width(int radius, int dyrow) {
    // The midpoint walk emits, for each step (dx, dy): rows +-dy get half-width dx, rows +-dx get half-width dy.
    // A row |r| is reached as a "dy row" while dx >= dy, and as a "dx row" each time dx takes the value |r|
    // (then with the LAST dy written before dx decrements being the widest).  Re-walk it: the circle is small.
    const int ar = dyrow < 0 ? -dyrow : dyrow;
    int best = -1;
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    while (dx >= dy) {
        if (dy == ar) best = max(best, dx);
        if (dx == ar) best = max(best, dy);
        dy++;
        err += plus;
        plus += 2;
        const int m = (err <= 0) - 1;
        err -= minus & m;
        dx += m;
        minus -= m & 2;
    }
    return best;
}

// The occupancy image (checkImg) as a bit mask: in LDS when it fits (w*h bits <= 64 KB: every 640x480-class
// frame), otherwise in global memory behind agent-scope atomics (one wavefront reads what it just wrote).
#define OCC_LDS_WORDS 16384
template <bool LDSMASK>
struct OccMask {
    unsigned* words;  // LDS or global, (w + 31) / 32 words per row
    int wpr;
    __device__ inline bool test(int x, int y) const {
        const unsigned* p = words + (size_t)y * wpr + (x >> 5);
        const unsigned v = LDSMASK ? *p : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return (v >> (x & 31)) & 1u;
    }
    __device__ inline void set_span(int y, int xa, int xb) const {  // bits xa..xb of row y
        for (int wd = xa >> 5; wd <= (xb >> 5); wd++) {
            const int lo = max(xa, wd * 32) - wd * 32, hi = min(xb, wd * 32 + 31) - wd * 32;
            const unsigned m = (hi == 31 ? 0xffffffffu : ((1u << (hi + 1)) - 1u)) & ~((1u << lo) - 1u);
            unsigned* p = words + (size_t)y * wpr + wd;
            if (LDSMASK) atomicOr(p, m);
            else __hip_atomic_fetch_or(p, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
};

// one lane per row of the circle: lane l of pass q owns row offset -radius + 64 q + l; its half-width is fixed for
// the whole kernel (hw[q], -1 = no such row), so stamping a circle is one to three word ORs per lane
#define OCC_PASSES 2  // radii up to 63
template <bool LDSMASK>
__device__ inline void stamp_circle(const OccMask<LDSMASK>& mask, int w, int h, int cx, int cy, int radius, int lane,
                                    const int (&hw)[OCC_PASSES]) {
#pragma unroll
    for (int q = 0; q < OCC_PASSES; q++) {
        const int y = cy - radius + 64 * q + lane;
        if (hw[q] < 0 || y < 0 || y >= h) continue;
        const int xa = max(cx - hw[q], 0), xb = min(cx + hw[q], w - 1);
        if (xa <= xb) mask.set_span(y, xa, xb);
    }
}

// replenishFeatures' selection.  Four wavefronts clear the mask and stamp the existing landmarks' circles (order-free:
// bit ORs; every wavefront fetches 64 landmark pixels at once and takes every fourth); the first fit itself is sequential
// by definition and stays with wavefront 0.  The global mask (LDSMASK = false) is zeroed by the caller.
template <bool LDSMASK>
__global__ __launch_bounds__(256) void replenish_select_kernel(const int* kp_xy, const int* kp_count, int cap_kp,
                                                              const float* __restrict__ mu, int N_old, float fx, float fy, float cx,
                                                              float cy, int num_features, int w, int h, int radius, int kill_pad,
                                                              unsigned* gmask, int* new_xy, float* new_uv, int* new_count,
                                                              const unsigned* __restrict__ row_kp, const int* __restrict__ row_cnt,
                                                              int* kp_xy_out, short* kp_score_out, int* kp_total_out) {
    __shared__ unsigned lmask[LDSMASK ? OCC_LDS_WORDS : 1];
    __shared__ int s_gw[4];
    __shared__ int s_gcarry;
    // row_kp != nullptr: the detector's row lists are flattened here, by the same four wavefronts that prepare the mask
    // (fast_gather_kernel's launch saved); kp_xy / kp_count then alias the arrays written here
    if (row_kp) {
        fast_gather_body<256>(row_kp, row_cnt, w, h, cap_kp, kp_xy_out, kp_score_out, kp_total_out, s_gw, &s_gcarry);
        __threadfence();
        __syncthreads();
        kp_xy = kp_xy_out;
        kp_count = kp_total_out;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    OccMask<LDSMASK> mask;
    mask.wpr = (w + 31) / 32;
    mask.words = LDSMASK ? lmask : gmask;
    if (LDSMASK) {
        for (int i = threadIdx.x; i < mask.wpr * h; i += 256) lmask[i] = 0u;
        __syncthreads();
    }
    int hw[OCC_PASSES];
#pragma unroll
    for (int q = 0; q < OCC_PASSES; q++) {
        const int rr = -radius + 64 * q + lane;
        hw[q] = (rr <= radius) ? circle_half_width(radius, rr) : -1;
    }
    // occupancy of the existing landmarks: circle at cv::Point(getPixel(f)) = cvRound of (u*fx + cx, v*fy + cy)
    for (int i0 = 0; i0 < N_old; i0 += 64) {
        const int il = min(i0 + lane, N_old - 1);
        const float u = mu[EKF_BASE + 3 * il], v = mu[EKF_BASE + 3 * il + 1];
        const int mypx = __float2int_rn(u * fx + cx), mypy = __float2int_rn(v * fy + cy);
        const int cnt = min(64, N_old - i0);
        for (int j = wave; j < cnt; j += 4) stamp_circle(mask, w, h, __shfl(mypx, j, 64), __shfl(mypy, j, 64), radius, lane, hw);
    }
    if (!LDSMASK) __threadfence();  // the other wavefronts' ORs went to the global mask
    __syncthreads();
    if (wave != 0) return;  // (no workgroup barrier below this line)
    int wanted = num_features - N_old, added = 0;
    const int nk = min(*kp_count, cap_kp);
    for (int c0 = 0; c0 < nk && added < wanted; c0 += 64) {
        const int k = c0 + lane;
        int x = 0, y = 0;
        bool cand = false;
        if (k < nk) {
            x = kp_xy[2 * k];
            y = kp_xy[2 * k + 1];
            cand = !(x < kill_pad || y < kill_pad || w - x < kill_pad || h - y < kill_pad);  // Frame::isPixelInBox
        }
        unsigned long long todo = __ballot(cand);
        while (todo && added < wanted) {
            // test the remaining candidates against the current mask
            const bool is_free = cand && ((todo >> lane) & 1ull) && !mask.test(x, y);
            const unsigned long long fb = __ballot(is_free);
            if (!fb) break;
            const int L = __ffsll((long long)fb) - 1;
            const int ax = __shfl(x, L, 64), ay = __shfl(y, L, 64);
            if (lane == 0) {
                new_xy[2 * added] = ax;
                new_xy[2 * added + 1] = ay;
                // Feature::pixel2Metric (Feature.h:60-62)
                new_uv[2 * added] = ((float)ax - cx) / fx;
                new_uv[2 * added + 1] = ((float)ay - cy) / fy;
            }
            added++;
            stamp_circle(mask, w, h, ax, ay, radius, lane, hw);
            // this wavefront's own ORs must have landed before it tests the mask again; wavefronts 1-3 have exited, so
            // a workgroup barrier here would be a barrier after a partial exit: wait for the wave's own memory operations
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            todo &= ~((2ull << L) - 1ull);  // everything up to and including L is consumed
        }
    }
    if (lane == 0) *new_count = added;
}

// cv::GaussianBlur(img, Size(5,5), sigma) on level 0 (EKFVIO.cpp:228-232, FAST_BLUR_SIGMA != 0): OpenCV 3.x runs
// symmetric smoothing kernels on 8-bit data in fixed point (taps k = round(256 * g), exact integer row and column
// sums, (v + 2^15) >> 16), so the separable filter equals this single 25-tap integer sum.  `img` is level 0 inside its
// reflect-101 border (>= 2 pixels: BORDER_DEFAULT needs no special case); four outputs per thread.
__global__ __launch_bounds__(256) void gauss5_kernel(const uint8_t* __restrict__ img, int pitch, int w, int h, int k0, int k1, int k2,
                                                     uint8_t* __restrict__ out) {
    const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4, y = blockIdx.y;
    if (x0 >= w) return;
    const int kk[5] = {k0, k1, k2, k1, k0};
    int acc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const uint8_t* r = img + (ptrdiff_t)(y + i - 2) * pitch + x0 - 2;
        int v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = r[j];
#pragma unroll
        for (int o = 0; o < 4; o++) {
            int rs = 0;
#pragma unroll
            for (int j = 0; j < 5; j++) rs += kk[j] * v[o + j];
            acc[o] += kk[i] * rs;
        }
    }
#pragma unroll
    for (int o = 0; o < 4; o++)
        if (x0 + o < w) out[(size_t)y * w + x0 + o] = (uint8_t)min(max((acc[o] + (1 << 15)) >> 16, 0), 255);
}

}  // namespace

// getGaussianKernel(5, sigma, CV_32F) (smooth.cpp) -> the fixed-point taps of createSeparableLinearFilter's 8-bit path
static void gauss5_taps(float sigma, int k[5]) {
    float cf[5];
    const double scale2x = -0.5 / ((double)sigma * (double)sigma);
    double sum = 0;
    for (int i = 0; i < 5; i++) {
        const double x = i - 2.0;
        cf[i] = (float)std::exp(scale2x * x * x);
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < 5; i++) {
        cf[i] = (float)(cf[i] * sum);
        k[i] = (int)std::nearbyint((double)(cf[i] * 256.f));  // cvRound: half to even
    }
}

// The detector's per-pixel buffers for a level 0 of up to w x h pixels.  Frame::Frame takes any image (Frame.cpp:15-42):
// the buffers start at cfg.max_image_width x max_image_height and grow with the first frame that needs more.
int fast_ensure(ekfvio_filter* f, int w, int h) {
    if (w <= f->fast_cap_w && h <= f->fast_cap_h) return EKFVIO_OK;
    const ekfvio_config& c = f->cfg;
    w = std::max(w, f->fast_cap_w);
    h = std::max(h, f->fast_cap_h);
    if (f->stream) HIPF(f, hipStreamSynchronize(f->stream));  // nothing in flight may still read the old buffers
    void* old[] = {f->blurred, f->fast_row_kp, f->fast_kp_xy, f->fast_kp_score, f->occ_mask, f->fast_row_cnt};
    for (void* p : old)
        if (p) (void)hipFree(p);
    f->blurred = nullptr, f->fast_row_kp = nullptr, f->fast_kp_xy = nullptr, f->fast_kp_score = nullptr, f->occ_mask = nullptr,
    f->fast_row_cnt = nullptr;
    f->fast_cap_w = f->fast_cap_h = 0;
    const size_t px = (size_t)w * h;
    if (c.fast_blur_sigma != 0.f) HIPF(f, hipMalloc((void**)&f->blurred, px));
    HIPF(f, hipMalloc((void**)&f->fast_row_kp, px * sizeof(unsigned)));  // per image row: its keypoints in x order, (score << 16) | x
    f->fast_kp_cap = (int)(px / 4 + 1);
    HIPF(f, hipMalloc((void**)&f->fast_kp_xy, (size_t)f->fast_kp_cap * 2 * sizeof(int)));
    HIPF(f, hipMalloc((void**)&f->fast_kp_score, (size_t)f->fast_kp_cap * sizeof(short)));
    HIPF(f, hipMalloc((void**)&f->occ_mask, ((size_t)(w + 31) / 32) * h * sizeof(unsigned)));
    HIPF(f, hipMalloc((void**)&f->fast_row_cnt, (size_t)h * sizeof(int)));
    f->fast_cap_w = w;
    f->fast_cap_h = h;
    return EKFVIO_OK;
}

int fast_alloc(ekfvio_filter* f) {
    const ekfvio_config& c = f->cfg;
    // what replenishFeatures can be asked for is checked here, once, not in the middle of a frame
    if (c.fast_blur_sigma != 0.f && !(c.fast_blur_sigma > 0.f)) {
        f->last_error = "fast_blur_sigma must be >= 0";  // cv::GaussianBlur would derive sigma from the kernel size
        return EKFVIO_EINVAL;
    }
    if (c.min_new_feature_dist < 0 || c.min_new_feature_dist > 64 * OCC_PASSES / 2 - 1) {
        f->last_error = "min_new_feature_dist must be in [0, 63]";
        return EKFVIO_EINVAL;
    }
    const int maxf = c.max_features > 0 ? c.max_features : 1;
    HIPF(f, hipMalloc((void**)&f->new_xy, (size_t)maxf * 2 * sizeof(int)));
    HIPF(f, hipMalloc((void**)&f->fast_counts, 4 * sizeof(int)));
    return fast_ensure(f, c.max_image_width, c.max_image_height);
}

void fast_free(ekfvio_filter* f) {
    void* ptrs[] = {f->blurred, f->fast_row_kp, f->fast_kp_xy, f->fast_kp_score, f->occ_mask, f->new_xy, f->fast_counts, f->fast_row_cnt};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
}

// cv::FAST on level 0 of the current frame -> f->fast_kp_* (raster order), count in f->fast_counts[0]
// gather = false: the row lists are left for the caller's own kernel to flatten (replenish_select_kernel)
int fast_detect_device(ekfvio_filter* f, int threshold, int nonmax, bool blur, bool gather = true) {
    const KltFrame& fr = f->frames[f->cur];
    if (!fr.valid) {
        f->last_error = "FAST needs a frame";
        return EKFVIO_ESTATE;
    }
    const int w = fr.w[0], h = fr.h[0];
    int pitch = klt_level_pitch(w);
    const uint8_t* img0 = fr.img[0] + (size_t)klt_border() * pitch + klt_border();
    if (blur) {  // replenishFeatures: cv::GaussianBlur(f.img, img, Size(5,5), FAST_BLUR_SIGMA) (EKFVIO.cpp:228-232)
        int k[5];
        gauss5_taps(f->cfg.fast_blur_sigma, k);
        hipLaunchKernelGGL(gauss5_kernel, dim3((w + 1023) / 1024, h), dim3(256), 0, f->stream, img0, pitch, w, h, k[0], k[1], k[2],
                           f->blurred);
        img0 = f->blurred;
        pitch = w;
    }
    hipLaunchKernelGGL(fast_rows_kernel, dim3(h), dim3(FAST_CHUNK), 0, f->stream, img0, w, h, pitch, threshold, nonmax, f->fast_row_kp,
                       f->fast_row_cnt);
    if (gather)
        hipLaunchKernelGGL(fast_gather_kernel, dim3(1), dim3(1024), 0, f->stream, f->fast_row_kp, f->fast_row_cnt, w, h, f->fast_kp_cap,
                           f->fast_kp_xy, f->fast_kp_score, f->fast_counts);
    return EKFVIO_OK;
}

// FAST on the current frame and the first-fit selection, enqueued only: the new landmarks' pixels are in f->new_xy, their
// metric positions in f->zmeas, their number in f->fast_counts[1].  *enqueued = 0 if the filter is full (nothing to do).
int replenish_enqueue(ekfvio_filter* f, int* enqueued) {
    *enqueued = 0;
    if (!f) return EKFVIO_EINVAL;
    // (fast_blur_sigma and min_new_feature_dist were validated by ekfvio_create: fast_alloc)
    HIPF(f, hipSetDevice(f->device));
    if (f->N >= f->cfg.max_features) return EKFVIO_OK;  // "if (tc_ekf.features.size() < NUM_FEATURES)" (:236)
    int rc = fast_detect_device(f, f->cfg.fast_threshold, 1, f->cfg.fast_blur_sigma != 0.f, false);
    if (rc != EKFVIO_OK) return rc;
    const KltFrame& fr = f->frames[f->cur];
    const int w = fr.w[0], h = fr.h[0];
    float fx, fy, cx, cy;
    klt_intrinsics(f, fr.K, &fx, &fy, &cx, &cy);
    const size_t mask_words = (size_t)((w + 31) / 32) * h;
    if (mask_words <= OCC_LDS_WORDS) {
        hipLaunchKernelGGL(replenish_select_kernel<true>, dim3(1), dim3(256), 0, f->stream, f->fast_kp_xy, f->fast_counts, f->fast_kp_cap,
                           f->mu, f->N, fx, fy, cx, cy, f->cfg.max_features, w, h, f->cfg.min_new_feature_dist, f->cfg.kill_pad,
                           (unsigned*)f->occ_mask, f->new_xy, f->zmeas, f->fast_counts + 1, f->fast_row_kp, f->fast_row_cnt, f->fast_kp_xy,
                           f->fast_kp_score, f->fast_counts);
    } else {
        HIPF(f, hipMemsetAsync(f->occ_mask, 0, mask_words * sizeof(unsigned), f->stream));
        hipLaunchKernelGGL(replenish_select_kernel<false>, dim3(1), dim3(256), 0, f->stream, f->fast_kp_xy, f->fast_counts, f->fast_kp_cap,
                           f->mu, f->N, fx, fy, cx, cy, f->cfg.max_features, w, h, f->cfg.min_new_feature_dist, f->cfg.kill_pad,
                           (unsigned*)f->occ_mask, f->new_xy, f->zmeas, f->fast_counts + 1, f->fast_row_kp, f->fast_row_cnt, f->fast_kp_xy,
                           f->fast_kp_score, f->fast_counts);
    }
    *enqueued = 1;
    return EKFVIO_OK;
}

extern "C" {

int ekfvio_fast_detect(ekfvio_filter* f, int32_t threshold, int32_t nonmax, int32_t cap, int32_t* xy, int32_t* score, int32_t* count) {
    if (!f || !count || cap < 0) return EKFVIO_EINVAL;
    HIPF(f, hipSetDevice(f->device));
    int rc = fast_detect_device(f, threshold, nonmax, f->cfg.fast_blur_sigma != 0.f);
    if (rc != EKFVIO_OK) return rc;
    int n = 0;
    HIPF(f, hipMemcpyAsync(&n, f->fast_counts, sizeof(int), hipMemcpyDeviceToHost, f->stream));
    HIPF(f, hipStreamSynchronize(f->stream));
    *count = n;
    const int k = std::min(std::min(n, cap), f->fast_kp_cap);
    if (k > 0 && xy) HIPF(f, hipMemcpyAsync(xy, f->fast_kp_xy, sizeof(int) * 2 * k, hipMemcpyDeviceToHost, f->stream));
    if (k > 0 && score) {
        std::vector<short> hs(k);
        HIPF(f, hipMemcpyAsync(hs.data(), f->fast_kp_score, sizeof(short) * k, hipMemcpyDeviceToHost, f->stream));
        HIPF(f, hipStreamSynchronize(f->stream));
        for (int i = 0; i < k; i++) score[i] = hs[i];
    }
    HIPF(f, hipStreamSynchronize(f->stream));
    return EKFVIO_OK;
}

// Test hook: the Gaussian-blurred level 0 the last FAST run saw (w*h bytes, tightly packed); needs cfg.fast_blur_sigma != 0.
#ifdef EKFVIO_TEST_HOOKS  // include/ekfvio_test_hooks.h: only in libekfvio_hip_hooks.so
int ekfvio_test_blurred_level0(ekfvio_filter* f, uint8_t* out) {
    if (!f || !out) return EKFVIO_EINVAL;
    const KltFrame& fr = f->frames[f->cur];
    if (!fr.valid || !f->blurred) return EKFVIO_ESTATE;
    HIPF(f, hipSetDevice(f->device));
    HIPF(f, hipMemcpyAsync(out, f->blurred, (size_t)fr.w[0] * fr.h[0], hipMemcpyDeviceToHost, f->stream));
    HIPF(f, hipStreamSynchronize(f->stream));
    return EKFVIO_OK;
}
#endif

// EKFVIO::replenishFeatures (EKFVIO.cpp:224-311) on the current frame
int ekfvio_replenish(ekfvio_filter* f, int32_t* added, int32_t* new_px_xy) {
    if (!f) return EKFVIO_EINVAL;
    f->out_fresh = false;
    if (added) *added = 0;
    int enq = 0;
    int rc = replenish_enqueue(f, &enq);
    if (rc != EKFVIO_OK || !enq) return rc;
    int k = 0, bad = 0;
    rc = wait_status(f, &bad, f->fast_counts + 1, &k);  // the count through the polled host word (no pageable copy)
    if (rc != EKFVIO_OK) return rc;
    if (k > 0 && new_px_xy) HIPF(f, hipMemcpyAsync(new_px_xy, f->new_xy, sizeof(int) * 2 * k, hipMemcpyDeviceToHost, f->stream));
    if (k > 0) {
        rc = add_features_device(f, k);  // uv already in f->zmeas
        if (rc != EKFVIO_OK) return rc;
    }
    HIPF(f, hipStreamSynchronize(f->stream));
    if (added) *added = k;
    return EKFVIO_OK;
}

}  // extern "C"
