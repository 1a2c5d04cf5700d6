// oracle/klt_oracle.cpp — TEST INFRASTRUCTURE ONLY (see ekf_oracle.hpp header).
//
// CPU restatement of the arithmetic behind the reference's only image operation on the hot
// path: cv::calcOpticalFlowPyrLK as called at include/ekf_vio/KLTTracker.cpp:61-64
//   (prevImg, nextImg, prevPts, nextPts, status, err, Size(21,21), maxLevel 3,
//    TermCriteria(COUNT+EPS, 30, 0.01), OPTFLOW_USE_INITIAL_FLOW, minEigThreshold 1e-4).
// OpenCV is an un-vendored dependency of the reference (CMakeLists.txt:31, version
// unpinned, 3.x API era) and is absent from this image, so this file restates the
// published algorithm of OpenCV 3.x modules/video/src/lkpyramid.cpp and
// modules/imgproc/src/pyramids.cpp:
//   * pyramid: pyrDown = separable [1 4 6 4 1]/16, integer, (sum + 128) >> 8, size
//     ((w+1)/2,(h+1)/2), BORDER_REFLECT_101; levels stop when a level is not larger than
//     the window (buildOpticalFlowPyramid);
//   * derivatives: Scharr 3/10/3, int16, interleaved (dx,dy), reflect-101 at the image
//     edge, ZERO outside the image (copyMakeBorder BORDER_CONSTANT);
//   * tracker (LKTrackerInvoker): W_BITS = 14 fixed-point bilinear weights via cvRound
//     (round-half-even), patch descale >> 9 (image, 5 extra bits) and >> 14 (derivatives),
//     FLT_SCALE = 2^-20, minEig test, <= maxCount Gauss-Newton steps, stop at |delta|^2 <=
//     eps^2 or on the oscillation test, status only cleared at level 0.
// PARITY STATUS: "parity unpinned" against OpenCV itself — the reference's only KLT test
// (test/klt_test.cpp) asserts nothing and never calls the tracker.  Pinned instead by the
// reference's test image pair (images/640_480_test.png -> 640_480_moved_test.png is a pure
// translation by (-21,-7) px) and by exact integer properties of the pyramid.
//
// One deliberate, documented choice: the 2x2 gradient matrix and the mismatch vector are
// accumulated in exact 64-bit integers (accum_mode 0) before the single conversion to
// float.  OpenCV's own accumulation order is build dependent (scalar float adds vs SSE2
// madd_epi16 pairs vs NEON), so no order is "the" reference; exact accumulation is
// order-free, which lets the HIP wave reduction agree with this oracle bit for bit.
// accum_mode 1 reproduces OpenCV's scalar path (sequential float adds) for comparison.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

struct Level {
    int w = 0, h = 0;
    std::vector<uint8_t> img;    // w*h
    std::vector<int16_t> deriv;  // w*h*2 (dx,dy)
};

struct Frame {
    std::vector<Level> lv;
};

inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

void pyr_down(const Level& s, Level& d) {
    d.w = (s.w + 1) / 2;
    d.h = (s.h + 1) / 2;
    d.img.assign((size_t)d.w * d.h, 0);
    std::vector<int> row((size_t)5 * d.w);
    for (int y = 0; y < d.h; y++) {
        for (int k = 0; k < 5; k++) {
            const int sy = reflect101(2 * y + k - 2, s.h);
            const uint8_t* sp = &s.img[(size_t)sy * s.w];
            for (int x = 0; x < d.w; x++) {
                const int x0 = reflect101(2 * x - 2, s.w), x1 = reflect101(2 * x - 1, s.w), x2 = reflect101(2 * x, s.w),
                          x3 = reflect101(2 * x + 1, s.w), x4 = reflect101(2 * x + 2, s.w);
                row[(size_t)k * d.w + x] = sp[x0] + sp[x4] + 4 * (sp[x1] + sp[x3]) + 6 * sp[x2];
            }
        }
        for (int x = 0; x < d.w; x++) {
            const int v = row[x] + row[(size_t)4 * d.w + x] + 4 * (row[(size_t)d.w + x] + row[(size_t)3 * d.w + x]) +
                          6 * row[(size_t)2 * d.w + x];
            d.img[(size_t)y * d.w + x] = (uint8_t)((v + 128) >> 8);
        }
    }
}

void scharr(Level& l) {
    l.deriv.assign((size_t)l.w * l.h * 2, 0);
    auto px = [&](int x, int y) { return (int)l.img[(size_t)reflect101(y, l.h) * l.w + reflect101(x, l.w)]; };
    for (int y = 0; y < l.h; y++)
        for (int x = 0; x < l.w; x++) {
            // trow0 = (up + down)*3 + mid*10 ; trow1 = down - up   (per column), then
            // dx = trow0[x+1] - trow0[x-1] ; dy = (trow1[x+1] + trow1[x-1])*3 + trow1[x]*10
            auto t0 = [&](int xx) { return (px(xx, y - 1) + px(xx, y + 1)) * 3 + px(xx, y) * 10; };
            auto t1 = [&](int xx) { return px(xx, y + 1) - px(xx, y - 1); };
            l.deriv[((size_t)y * l.w + x) * 2] = (int16_t)(t0(x + 1) - t0(x - 1));
            l.deriv[((size_t)y * l.w + x) * 2 + 1] = (int16_t)((t1(x + 1) + t1(x - 1)) * 3 + t1(x) * 10);
        }
}

// image read with the reflect-101 border of the pyramid level, derivative read with zero border
inline int img_at(const Level& l, int x, int y) { return l.img[(size_t)reflect101(y, l.h) * l.w + reflect101(x, l.w)]; }
inline int der_at(const Level& l, int x, int y, int c) {
    if (x < 0 || y < 0 || x >= l.w || y >= l.h) return 0;
    return l.deriv[((size_t)y * l.w + x) * 2 + c];
}
inline int cv_round(float v) { return (int)std::nearbyint((double)v); }  // lrint: half to even
inline int cv_floor(float v) { return (int)std::floor(v); }
inline int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

}  // namespace

extern "C" {

void* orc_klt_frame_create(const uint8_t* img, int w, int h, int stride, int win, int max_level) {
    Frame* f = new Frame();
    Level l0;
    l0.w = w;
    l0.h = h;
    l0.img.resize((size_t)w * h);
    for (int y = 0; y < h; y++) std::memcpy(&l0.img[(size_t)y * w], img + (size_t)y * stride, w);
    f->lv.push_back(l0);
    for (int lev = 1; lev <= max_level; lev++) {
        Level d;
        pyr_down(f->lv.back(), d);
        if (d.w <= win || d.h <= win) break;  // buildOpticalFlowPyramid: level too small for the window
        f->lv.push_back(d);
    }
    for (auto& l : f->lv) scharr(l);
    return f;
}
void orc_klt_frame_destroy(void* p) { delete (Frame*)p; }
int orc_klt_levels(void* p) { return (int)((Frame*)p)->lv.size(); }
void orc_klt_level_size(void* p, int level, int* w, int* h) {
    *w = ((Frame*)p)->lv[level].w;
    *h = ((Frame*)p)->lv[level].h;
}
void orc_klt_get_level(void* p, int level, uint8_t* img, int16_t* deriv) {
    const Level& l = ((Frame*)p)->lv[level];
    if (img) std::memcpy(img, l.img.data(), l.img.size());
    if (deriv) std::memcpy(deriv, l.deriv.data(), l.deriv.size() * sizeof(int16_t));
}

// prev_px: n x 2 reference positions in the previous frame; next_px: n x 2 initial guesses
// (OPTFLOW_USE_INITIAL_FLOW), overwritten with the result; status: n bytes.
void orc_klt_track(void* prev_p, void* next_p, const float* prev_px, float* next_px, int n, int win, int max_iter,
                   float epsilon, float min_eig, int accum_mode, uint8_t* status, int* iters_out) {
    const Frame& P = *(Frame*)prev_p;
    const Frame& Q = *(Frame*)next_p;
    const int levels = (int)std::min(P.lv.size(), Q.lv.size());
    const float eps2 = epsilon * epsilon;  // criteria.epsilon *= criteria.epsilon
    const int W_BITS = 14;
    const float FLT_SCALE = 1.f / (1 << 20);
    std::vector<int> Ipatch((size_t)win * win), dIx((size_t)win * win), dIy((size_t)win * win);
    for (int i = 0; i < n; i++) {
        status[i] = 1;
        int total_iters = 0;
        float ox = 0, oy = 0;  // nextPts[ptidx]
        for (int level = levels - 1; level >= 0; level--) {
            const Level& I = P.lv[level];
            const Level& J = Q.lv[level];
            const float sc = (float)(1. / (1 << level));
            float ppx = prev_px[2 * i] * sc, ppy = prev_px[2 * i + 1] * sc;
            if (level == levels - 1) {
                ox = next_px[2 * i] * sc;
                oy = next_px[2 * i + 1] * sc;
            } else {
                ox = ox * 2.f;
                oy = oy * 2.f;
            }
            const float half = (win - 1) * 0.5f;
            ppx -= half;
            ppy -= half;
            const int ipx = cv_floor(ppx), ipy = cv_floor(ppy);
            if (ipx < -win || ipx >= I.w || ipy < -win || ipy >= I.h) {
                if (level == 0) status[i] = 0;
                continue;
            }
            float a = ppx - ipx, b = ppy - ipy;
            int iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
            int iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
            int iw10 = cv_round((1.f - a) * b * (1 << W_BITS));
            int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            long long sA11 = 0, sA12 = 0, sA22 = 0;
            float fA11 = 0, fA12 = 0, fA22 = 0;
            for (int y = 0; y < win; y++)
                for (int x = 0; x < win; x++) {
                    const int X = ipx + x, Y = ipy + y;
                    const int ival = descale(img_at(I, X, Y) * iw00 + img_at(I, X + 1, Y) * iw01 +
                                                 img_at(I, X, Y + 1) * iw10 + img_at(I, X + 1, Y + 1) * iw11,
                                             W_BITS - 5);
                    const int ixv = descale(der_at(I, X, Y, 0) * iw00 + der_at(I, X + 1, Y, 0) * iw01 +
                                                der_at(I, X, Y + 1, 0) * iw10 + der_at(I, X + 1, Y + 1, 0) * iw11,
                                            W_BITS);
                    const int iyv = descale(der_at(I, X, Y, 1) * iw00 + der_at(I, X + 1, Y, 1) * iw01 +
                                                der_at(I, X, Y + 1, 1) * iw10 + der_at(I, X + 1, Y + 1, 1) * iw11,
                                            W_BITS);
                    Ipatch[(size_t)y * win + x] = ival;
                    dIx[(size_t)y * win + x] = ixv;
                    dIy[(size_t)y * win + x] = iyv;
                    sA11 += (long long)ixv * ixv;
                    sA12 += (long long)ixv * iyv;
                    sA22 += (long long)iyv * iyv;
                    fA11 += (float)(ixv * ixv);
                    fA12 += (float)(ixv * iyv);
                    fA22 += (float)(iyv * iyv);
                }
            float A11, A12, A22;
            if (accum_mode == 0) {
                A11 = (float)sA11 * FLT_SCALE; A12 = (float)sA12 * FLT_SCALE; A22 = (float)sA22 * FLT_SCALE;
            } else {
                A11 = fA11 * FLT_SCALE; A12 = fA12 * FLT_SCALE; A22 = fA22 * FLT_SCALE;
            }
            float D = A11 * A22 - A12 * A12;
            const float minEig = (A22 + A11 - std::sqrt((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * win * win);
            if (minEig < min_eig || D < 1.1920929e-07f) {  // FLT_EPSILON
                if (level == 0) status[i] = 0;
                continue;
            }
            D = 1.f / D;
            float nx = ox - half, ny = oy - half;  // nextPt -= halfWin (nextPts keeps the un-shifted value)
            float pdx = 0, pdy = 0;
            for (int j = 0; j < max_iter; j++) {
                const int inx = cv_floor(nx), iny = cv_floor(ny);
                if (inx < -win || inx >= J.w || iny < -win || iny >= J.h) {
                    if (level == 0) status[i] = 0;
                    break;
                }
                total_iters++;
                a = nx - inx;
                b = ny - iny;
                iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
                iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
                iw10 = cv_round((1.f - a) * b * (1 << W_BITS));
                iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
                long long sb1 = 0, sb2 = 0;
                float fb1 = 0, fb2 = 0;
                for (int y = 0; y < win; y++)
                    for (int x = 0; x < win; x++) {
                        const int X = inx + x, Y = iny + y;
                        const int diff = descale(img_at(J, X, Y) * iw00 + img_at(J, X + 1, Y) * iw01 +
                                                     img_at(J, X, Y + 1) * iw10 + img_at(J, X + 1, Y + 1) * iw11,
                                                 W_BITS - 5) -
                                         Ipatch[(size_t)y * win + x];
                        sb1 += (long long)diff * dIx[(size_t)y * win + x];
                        sb2 += (long long)diff * dIy[(size_t)y * win + x];
                        fb1 += (float)(diff * dIx[(size_t)y * win + x]);
                        fb2 += (float)(diff * dIy[(size_t)y * win + x]);
                    }
                const float b1 = (accum_mode == 0 ? (float)sb1 : fb1) * FLT_SCALE;
                const float b2 = (accum_mode == 0 ? (float)sb2 : fb2) * FLT_SCALE;
                const float dx = (A12 * b2 - A22 * b1) * D;
                const float dy = (A12 * b1 - A11 * b2) * D;
                nx += dx;
                ny += dy;
                ox = nx + half;  // nextPts[ptidx] = nextPt + halfWin
                oy = ny + half;
                if (dx * dx + dy * dy <= eps2) break;
                if (j > 0 && std::fabs(dx + pdx) < 0.01f && std::fabs(dy + pdy) < 0.01f) {
                    ox -= dx * 0.5f;
                    oy -= dy * 0.5f;
                    break;
                }
                pdx = dx;
                pdy = dy;
            }
            if (status[i] && level == 0) {
                // the error pass of LKTrackerInvoker also clears the status of points whose
                // final window left the image
                const int fx = cv_floor(ox - half), fy = cv_floor(oy - half);
                if (fx < -win || fx >= J.w || fy < -win || fy >= J.h) status[i] = 0;
            }
        }
        next_px[2 * i] = ox;
        next_px[2 * i + 1] = oy;
        if (iters_out) iters_out[i] = total_iters;
    }
}

// ---- KLTTracker::estimateUncertaintySampleBased (KLTTracker.cpp:111-175) -------------------
// Dead code in the reference (estimateUncertainty, :100-106, returns the constant 1e-5 I instead); restated so that
// the optional sample-based measurement covariance of the HIP path has a checker.
//
// cv::getRectSubPix(8-bit image, Size(5,5), center, patch, CV_32F) as published in OpenCV 3.x
// modules/imgproc/src/samplers.cpp (getRectSubPix_8u32f): the patch's top-left sample sits at
// center - (win-1)/2; ip = floor, a, b the fractions, a = max(a, 1e-4f).  Window strictly inside the image: per row
//     prev = (1-a) * (b1*src[0] + b2*src[step]);  for j: t = a12*src[j+1] + a22*src[j+1+step]; dst[j] = prev + t;
//     prev = (float)(t * s)   with   double s = (1 - a) / a
// (the horizontal interpolation is carried from pixel to pixel).  Otherwise: border-replicated bilinear taps with
// weights a11 = (1-a)(1-b), a12 = a(1-b), a21 = (1-a)b, a22 = ab — equal to OpenCV's adjustRect path in exact
// arithmetic; its summation order there is not restated ("parity unpinned", OpenCV absent and unpinned).
static void rect_subpix5(const uint8_t* img, int w, int h, int stride, float cx, float cy, float* dst /* 5x5 row-major */) {
    const int win = 5;
    cx -= (win - 1) * 0.5f;
    cy -= (win - 1) * 0.5f;
    const int ipx = cv_floor(cx), ipy = cv_floor(cy);
    if (0 <= ipx && ipx + win < w && 0 <= ipy && ipy + win < h) {
        float a = cx - ipx;
        const float b = cy - ipy;
        a = a > 0.0001f ? a : 0.0001f;
        const float a12 = a * (1.f - b), a22 = a * b, b1 = 1.f - b, b2 = b;
        const double sc = (1. - a) / a;
        const uint8_t* src = img + (size_t)ipy * stride + ipx;
        for (int i = 0; i < win; i++, src += stride) {
            float prev = (1 - a) * (b1 * src[0] + b2 * src[stride]);
            for (int j = 0; j < win; j++) {
                const float t = a12 * src[j + 1] + a22 * src[j + 1 + stride];
                dst[i * win + j] = prev + t;
                prev = (float)(t * sc);
            }
        }
    } else {
        const float a = cx - ipx, b = cy - ipy;
        const float a11 = (1.f - a) * (1.f - b), a12 = a * (1.f - b), a21 = (1.f - a) * b, a22 = a * b;
        auto px = [&](int x, int y) -> float {
            x = x < 0 ? 0 : (x >= w ? w - 1 : x);
            y = y < 0 ? 0 : (y >= h ? h - 1 : y);
            return (float)img[(size_t)y * stride + x];
        };
        for (int i = 0; i < win; i++)
            for (int j = 0; j < win; j++)
                dst[i * win + j] = px(ipx + j, ipy + i) * a11 + px(ipx + j + 1, ipy + i) * a12 + px(ipx + j, ipy + i + 1) * a21 +
                                   px(ipx + j + 1, ipy + i + 1) * a22;
    }
}

// ref_px / cur_px: n x 2 pixel positions in the previous / current level-0 image; cov: n x 4 (row-major 2x2, px^2).
// Arithmetic as written at :127-166: pow(float, 2) and exp(float) promote to double, the sums are float, samples run
// du outer / dv inner over {-10,-5,0,5,10}.
void orc_klt_uncertainty(void* prev_p, void* cur_p, const float* ref_px, const float* cur_px, int n, float* cov) {
    const Level& P = ((Frame*)prev_p)->lv[0];
    const Level& Cc = ((Frame*)cur_p)->lv[0];
    for (int t = 0; t < n; t++) {
        float ref[25], smp[25];
        rect_subpix5(P.img.data(), P.w, P.h, P.w, ref_px[2 * t], ref_px[2 * t + 1], ref);
        const float window_area = 25.f, k = 0.01f;
        float sum_rd = 0, sum_xx = 0, sum_yy = 0, sum_xy = 0;
        for (float du = -10; du <= 10; du += 5) {
            for (float dv = -10; dv <= 10; dv += 5) {
                rect_subpix5(Cc.img.data(), Cc.w, Cc.h, Cc.w, cur_px[2 * t] + du, cur_px[2 * t + 1] + dv, smp);
                float ssd = 0;
                for (int i = 0; i < 5; i++)
                    for (int j = 0; j < 5; j++) {
                        const double d = (double)(ref[i * 5 + j] - smp[i * 5 + j]);
                        ssd = (float)((double)ssd + d * d);
                    }
                ssd /= window_area;
                const float rd = (float)std::exp((double)(-k * ssd));
                sum_rd += rd;
                sum_xx += rd * du * du;
                sum_yy += rd * dv * dv;
                sum_xy += rd * du * dv;
            }
        }
        cov[4 * t + 0] = sum_xx / sum_rd;
        cov[4 * t + 3] = sum_yy / sum_rd;
        cov[4 * t + 1] = sum_xy / sum_rd;
        cov[4 * t + 2] = cov[4 * t + 1];
    }
}

}  // extern "C"
