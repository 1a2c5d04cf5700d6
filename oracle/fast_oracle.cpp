// oracle/fast_oracle.cpp — TEST INFRASTRUCTURE ONLY (see ekf_oracle.hpp header).
//
// CPU restatement of the image work either side of the filter's hot path:
//   * Frame::Frame (include/ekf_vio/Frame.cpp:15-42): cv::resize(img, Size(cols/s, rows/s)) with
//     the default INTER_LINEAR, K scaled by 1/s;
//   * EKFVIO::replenishFeatures (include/ekf_vio/EKFVIO.cpp:224-311): cv::FAST(img, kp, FAST_THRESHOLD,
//     nonmaxSuppression = true) (:242), an occupancy image of filled circles of radius
//     MIN_NEW_FEATURE_DIST around the existing landmarks' pixels (:257-260), then a first-fit pass over
//     the keypoints in detector order that skips occupied or kill-box pixels and stamps a circle for
//     every accepted one (:262-305), until NUM_FEATURES landmarks exist.
// OpenCV is an un-vendored dependency of the reference (CMakeLists.txt:31, version unpinned, 3.x API
// era) and is absent from this image, so this file restates the published algorithms of OpenCV 3.x:
//   * modules/imgproc/src/resize.cpp, 8-bit INTER_LINEAR: source coordinate (dx + 0.5) * scale - 0.5
//     in float, 11-bit fixed-point weights through saturate_cast<short> (round half to even),
//     horizontal pass in int, vertical pass ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2;
//   * modules/features2d/src/fast.cpp / fast_score.cpp, TYPE_9_16: Bresenham circle of radius 3,
//     a pixel is a corner when 9 contiguous circle pixels are all > v + t or all < v - t, score =
//     the largest t' for which it still is (cornerScore<16>), non-maximum suppression against the 8
//     neighbours with strict >, keypoints emitted in raster order, 3-pixel image border excluded;
//   * modules/imgproc/src/drawing.cpp, Circle(): midpoint circle with horizontal fill spans, clipped.
// PARITY STATUS: "parity unpinned" against OpenCV itself (no OpenCV here, and the reference has no
// test of this path).  Pinned by construction properties checked in tests/test_fast_oracle_cpu.py
// (exact 2x2 box mean at integer scale 2 and 4, the FAST segment test against a brute-force
// definition, circle area/symmetry, selection invariants).
// FAST_BLUR_SIGMA != 0 (cv::GaussianBlur, EKFVIO.cpp:228-232; default 0 = off) is not restated.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

inline int cv_round(double v) { return (int)std::nearbyint(v); }  // round half to even (default FP mode)
inline short sat_short(float v) {
    const int i = cv_round(v);
    return (short)std::min(32767, std::max(-32768, i));
}

// resize.cpp: 8u INTER_LINEAR to (dw, dh)
void resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh) {
    const double scale_x = (double)sw / dw, scale_y = (double)sh / dh;
    std::vector<int> xofs(dw), yofs(dh);
    std::vector<short> ialpha(2 * dw), ibeta(2 * dh);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)std::floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        ialpha[2 * dx] = sat_short((1.f - fx) * 2048);
        ialpha[2 * dx + 1] = sat_short(fx * 2048);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)std::floor(fy);
        fy -= sy;
        yofs[dy] = sy;
        ibeta[2 * dy] = sat_short((1.f - fy) * 2048);
        ibeta[2 * dy + 1] = sat_short(fy * 2048);
    }
    std::vector<int> r0(dw), r1(dw);
    for (int dy = 0; dy < dh; dy++) {
        const int sy0 = std::min(std::max(yofs[dy], 0), sh - 1), sy1 = std::min(std::max(yofs[dy] + 1, 0), sh - 1);
        const uint8_t *S0 = src + (size_t)sy0 * sstride, *S1 = src + (size_t)sy1 * sstride;
        for (int dx = 0; dx < dw; dx++) {
            const int sx = xofs[dx], sx1 = std::min(sx + 1, sw - 1);
            const int a0 = ialpha[2 * dx], a1 = ialpha[2 * dx + 1];
            r0[dx] = S0[sx] * a0 + S0[sx1] * a1;
            r1[dx] = S1[sx] * a0 + S1[sx1] * a1;
        }
        const int b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
        for (int dx = 0; dx < dw; dx++)
            dst[(size_t)dy * dw + dx] = (uint8_t)((((b0 * (r0[dx] >> 4)) >> 16) + ((b1 * (r1[dx] >> 4)) >> 16) + 2) >> 2);
    }
}

// fast.cpp makeOffsets, patternSize 16 (x, y)
const int kOff[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                         {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

// segment test: 9 contiguous of the 16 all brighter than v + t, or all darker than v - t
bool is_corner(const uint8_t* p, int stride, int t) {
    const int v = p[0];
    int d[16];
    for (int k = 0; k < 16; k++) {
        const int q = p[kOff[k][1] * stride + kOff[k][0]];
        d[k] = (q > v + t) ? 1 : ((q < v - t) ? 2 : 0);
    }
    for (int cls = 1; cls <= 2; cls++) {
        int run = 0;
        for (int k = 0; k < 16 + 8; k++) {
            if (d[k & 15] == cls) {
                if (++run >= 9) return true;
            } else {
                run = 0;
            }
        }
    }
    return false;
}

// fast_score.cpp cornerScore<16>: the largest threshold for which the pixel is still a corner
int corner_score(const uint8_t* p, int stride, int threshold) {
    const int K = 8, N = K * 3 + 1;
    const int v = p[0];
    short d[N];
    for (int k = 0; k < N; k++) d[k] = (short)(v - p[kOff[k & 15][1] * stride + kOff[k & 15][0]]);
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = std::min((int)d[k + 1], (int)d[k + 2]);
        a = std::min(a, (int)d[k + 3]);
        if (a <= a0) continue;
        a = std::min(a, (int)d[k + 4]);
        a = std::min(a, (int)d[k + 5]);
        a = std::min(a, (int)d[k + 6]);
        a = std::min(a, (int)d[k + 7]);
        a = std::min(a, (int)d[k + 8]);
        a0 = std::max(a0, std::min(a, (int)d[k]));
        a0 = std::max(a0, std::min(a, (int)d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = std::max((int)d[k + 1], (int)d[k + 2]);
        b = std::max(b, (int)d[k + 3]);
        b = std::max(b, (int)d[k + 4]);
        b = std::max(b, (int)d[k + 5]);
        if (b >= b0) continue;
        b = std::max(b, (int)d[k + 6]);
        b = std::max(b, (int)d[k + 7]);
        b = std::max(b, (int)d[k + 8]);
        b0 = std::min(b0, std::max(b, (int)d[k]));
        b0 = std::min(b0, std::max(b, (int)d[k + 9]));
    }
    return -b0 - 1;
}

// score map: -1 = not a corner; corners carry cornerScore (>= threshold)
void fast_score_map(const uint8_t* img, int w, int h, int stride, int threshold, std::vector<int>& score) {
    score.assign((size_t)w * h, -1);
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            const uint8_t* p = img + (size_t)y * stride + x;
            if (is_corner(p, stride, threshold)) score[(size_t)y * w + x] = corner_score(p, stride, threshold);
        }
}

// drawing.cpp Circle(), filled, value 255, clipped to the image
void circle_fill(uint8_t* mask, int w, int h, int cx, int cy, int radius) {
    auto hline = [&](int y, int x0, int x1) {
        if (y < 0 || y >= h) return;
        x0 = std::max(x0, 0);
        x1 = std::min(x1, w - 1);
        for (int x = x0; x <= x1; x++) mask[(size_t)y * w + x] = 255;
    };
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    while (dx >= dy) {
        hline(cy - dy, cx - dx, cx + dx);
        hline(cy + dy, cx - dx, cx + dx);
        hline(cy - dx, cx - dy, cx + dy);
        hline(cy + dx, cx - dy, cx + dy);
        dy++;
        err += plus;
        plus += 2;
        const int m = (err <= 0) - 1;
        err -= minus & m;
        dx += m;
        minus -= m & 2;
    }
}

}  // namespace

extern "C" {

// Frame::Frame image part: dst is (w / inv) x (h / inv), tightly packed
int orc_frame_resize(const uint8_t* src, int w, int h, int stride, int inv_scale, uint8_t* dst) {
    if (!src || !dst || inv_scale < 1 || w / inv_scale < 1 || h / inv_scale < 1) return -1;
    const int dw = w / inv_scale, dh = h / inv_scale;
    if (inv_scale == 1) {
        for (int y = 0; y < h; y++) std::memcpy(dst + (size_t)y * w, src + (size_t)y * stride, w);
        return 0;
    }
    resize_linear_u8(src, w, h, stride, dst, dw, dh);
    return 0;
}

// cv::FAST(img, kp, threshold, nonmax): keypoints in raster order; returns the number found
// (xy/score filled up to `cap` entries)
// cv::GaussianBlur(img, out, Size(5,5), sigma) on an 8-bit image as replenishFeatures calls it when
// FAST_BLUR_SIGMA != 0 (EKFVIO.cpp:228-232).  OpenCV 3.x (the reference's era: ROS Kinetic ships 3.3.1), 8-bit path of
// createSeparableLinearFilter (modules/imgproc/src/filter.cpp) with getGaussianKernel (smooth.cpp):
//   * float kernel: cf[i] = (float)exp(-0.5/sigma^2 * (i-2)^2), sum in double, cf[i] = (float)(cf[i] * (1/sum));
//   * symmetric smoothing kernels on 8-bit data run in fixed point: k[i] = cvRound(cf[i] * 256) (float product, round
//     half to even), row pass = exact int sums, column pass = exact int sums, result (v + 2^15) >> 16, saturated;
//   * BORDER_DEFAULT = BORDER_REFLECT_101.
// No intermediate rounding happens between the passes, so the result is the exact 2-D integer sum.
// PARITY STATUS: unpinned (OpenCV absent, version unpinned; 3.4.2+ switched 8-bit Gaussians to a ufixedpoint16 kernel
// whose taps are normalised to sum to 256 exactly).
void orc_gauss5_kernel(float sigma, int k[5]) {
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
        k[i] = cv_round(cf[i] * 256.f);
    }
}

int orc_gaussian_blur5(const uint8_t* img, int w, int h, int stride, float sigma, uint8_t* out) {
    if (!(sigma > 0.f) || w < 2 || h < 2) return -1;
    int k[5];
    orc_gauss5_kernel(sigma, k);
    auto r101 = [](int p, int len) {
        while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
        return p;
    };
    std::vector<int> row((size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int a = 0;
            for (int j = 0; j < 5; j++) a += k[j] * (int)img[(size_t)y * stride + r101(x + j - 2, w)];
            row[(size_t)y * w + x] = a;
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int a = 0;
            for (int i = 0; i < 5; i++) a += k[i] * row[(size_t)r101(y + i - 2, h) * w + x];
            a = (a + (1 << 15)) >> 16;
            out[(size_t)y * w + x] = (uint8_t)(a < 0 ? 0 : (a > 255 ? 255 : a));
        }
    return 0;
}

int orc_fast_detect(const uint8_t* img, int w, int h, int stride, int threshold, int nonmax, int cap, int* xy, int* score_out) {
    std::vector<int> sc;
    fast_score_map(img, w, h, stride, threshold, sc);
    int n = 0;
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            const int s = sc[(size_t)y * w + x];
            if (s < 0) continue;
            if (nonmax) {
                bool keep = true;
                for (int j = -1; j <= 1 && keep; j++)
                    for (int i = -1; i <= 1; i++)
                        if ((i || j) && !(s > sc[(size_t)(y + j) * w + x + i])) {
                            keep = false;
                            break;
                        }
                if (!keep) continue;
            }
            if (n < cap) {
                if (xy) { xy[2 * n] = x; xy[2 * n + 1] = y; }
                if (score_out) score_out[n] = s;
            }
            n++;
        }
    return n;
}

void orc_circle_fill(uint8_t* mask, int w, int h, int cx, int cy, int radius) { circle_fill(mask, w, h, cx, cy, radius); }

// replenishFeatures on an already scaled frame: existing landmark pixels (getPixel, float), the
// number of landmarks wanted in total; writes the accepted pixels (ints) and returns how many
int orc_replenish(const uint8_t* img, int w, int h, int stride, const float* existing_px, int n_existing, int num_features,
                  int threshold, int min_dist, int kill_pad, int cap, int* new_xy) {
    if (n_existing >= num_features) return 0;
    std::vector<int> xy(2 * ((size_t)w * h / 4 + 1));  // non-maximum suppression leaves at most one keypoint per 2x2
    const int nk = std::min(orc_fast_detect(img, w, h, stride, threshold, 1, (int)xy.size() / 2, xy.data(), nullptr), (int)xy.size() / 2);
    int needed = num_features - n_existing;
    std::vector<uint8_t> mask((size_t)w * h, 0);
    for (int i = 0; i < n_existing; i++)
        circle_fill(mask.data(), w, h, cv_round(existing_px[2 * i]), cv_round(existing_px[2 * i + 1]), min_dist);
    int added = 0;
    for (int i = 0; i < needed && i < nk; i++) {
        const int x = xy[2 * i], y = xy[2 * i + 1];
        if (mask[(size_t)y * w + x]) {  // a landmark is already close (:282)
            needed++;
            continue;
        }
        if (x < kill_pad || y < kill_pad || w - x < kill_pad || h - y < kill_pad) {  // Frame::isPixelInBox (:290)
            needed++;
            continue;
        }
        circle_fill(mask.data(), w, h, x, y, min_dist);
        if (added < cap) {
            new_xy[2 * added] = x;
            new_xy[2 * added + 1] = y;
        }
        added++;
    }
    return added;
}

}  // extern "C"
