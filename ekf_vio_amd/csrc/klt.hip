// ekf_vio_amd/csrc/klt.hip — pyramidal Lucas-Kanade tracker (placeholder until the HIP
// tracker lands; the entry points fail loudly rather than fall back to anything).
#include "common.h"

int klt_alloc(ekfvio_filter*) { return EKFVIO_OK; }
void klt_free(ekfvio_filter*) {}

extern "C" {
int ekfvio_klt_push_frame(ekfvio_filter* f, const uint8_t*, int32_t, int32_t, int32_t, const float*) {
    if (f) f->last_error = "KLT not built";
    return EKFVIO_ESTATE;
}
int ekfvio_klt_track(ekfvio_filter* f, float*, float*, uint8_t*) {
    if (f) f->last_error = "KLT not built";
    return EKFVIO_ESTATE;
}
int ekfvio_klt_track_points(ekfvio_filter* f, const float*, const float*, int32_t, float*, uint8_t*) {
    if (f) f->last_error = "KLT not built";
    return EKFVIO_ESTATE;
}
int ekfvio_step_image(ekfvio_filter* f, double, const uint8_t*, int32_t, int32_t, int32_t, const float*) {
    if (f) f->last_error = "KLT not built";
    return EKFVIO_ESTATE;
}
}
