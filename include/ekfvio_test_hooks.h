/* ekfvio_test_hooks.h -- raw kernels, in-kernel stamps and fault injection for the tests and the profiling scripts.
 *
 * NOT part of the drop-in boundary (include/ekfvio.h) and NOT in the product library: these entry points exist only in
 * libekfvio_hip_hooks.so, the same sources compiled with -DEKFVIO_TEST_HOOKS (ekf_vio_amd/_build.py builds both; the
 * Python mirror loads the hooks build only for TightlyCoupledEKF(..., hooks=True)).  libekfvio_hip.so exports no
 * ekfvio_test_* symbol (tests/test_abi_cpu.py).
 */
#ifndef EKFVIO_TEST_HOOKS_H_
#define EKFVIO_TEST_HOOKS_H_
#include "ekfvio.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Test hook: the same level WITH its border as the tracker reads it: (w + 2 border) x (h + 2 border) image bytes
 * (reflect-101 border) and int16 pairs (zero border).  `border` (may be NULL) receives the border width (24). */
EKFVIO_API int ekfvio_test_klt_padded_level(ekfvio_filter* f, int32_t level, int32_t* border, uint8_t* img, int16_t* deriv);
/* Test hook: the blurred level 0 (w*h bytes) the last FAST run saw; cfg.fast_blur_sigma must be non-zero. */
EKFVIO_API int ekfvio_test_blurred_level0(ekfvio_filter* f, uint8_t* out);
/* Raw kernels for unit tests (column-major, device copies made internally).  variant: 0 = the
 * production tile choice, 1 / 2 = 64x64 tiles with 256 / 512 threads, 32 / 48 / 64 = BM x 64 tiles. */
EKFVIO_API int ekfvio_test_gemm(ekfvio_filter* f, int32_t transB, int32_t M, int32_t N, int32_t K, float alpha, const float* A,
                     int32_t lda, const float* B, int32_t ldb, float beta, float* C, int32_t ldc, int32_t variant);
/* Mean time (us) of `reps` back-to-back GEMM launches at one shape, operands resident. */
EKFVIO_API int ekfvio_test_gemm_bench(ekfvio_filter* f, int32_t transB, int32_t lowerB, int32_t M, int32_t N, int32_t K,
                           int32_t reps, int32_t variant, double* mean_us);
/* Diagnostic: s_memtime stamps of the phases of one 64x64 diagonal-block factorisation. */
EKFVIO_API int ekfvio_test_potrf_stamps(ekfvio_filter* f, int64_t stamps[80]);
EKFVIO_API int ekfvio_test_sweep_stamps(ekfvio_filter* f, int enable, int64_t stamps[1024]);
/* Fault injection for the persistent sweep: at most `spin_limit` looks per wait (0: the production limit), and workgroup
   `stall_workgroup` of the launch never raises its tile's flag (-1: none), so every wait behind it runs out. */
EKFVIO_API int ekfvio_test_sweep_fault(ekfvio_filter* f, int32_t spin_limit, int32_t stall_workgroup);
EKFVIO_API int ekfvio_test_cholesky_solve(ekfvio_filter* f, int32_t m, int32_t nrhs, const float* S, const float* Crhs,
                               float* L_out, float* X_out, int32_t* info);


#ifdef __cplusplus
}
#endif
#endif /* EKFVIO_TEST_HOOKS_H_ */
