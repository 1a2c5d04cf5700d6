#include <hip/hip_runtime.h>
#include <cstdio>
#define T0() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory")
#define T1() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory")
// 16 groups of: 8 independent v_fmac_dpp + ONE probe instruction
#define GROUP(P) "v_fmac_f32_dpp %0, -%8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, -%8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n" \
  "v_fmac_f32_dpp %2, -%8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, -%8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n" \
  "v_fmac_f32_dpp %4, -%8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %5, -%8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n" \
  "v_fmac_f32_dpp %6, -%8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %7, -%8, %9 row_newbcast:8 row_mask:0xf bank_mask:0xf\n" P "\n"
#define RUN(name, P, CL...)                                                                                   \
    T0();                                                                                                     \
    asm volatile(".rept 16\n" GROUP(P) ".endr"                                                                \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)             \
                 : "v"(b), "v"(c), "v"(addr), "v"(addr8) : "memory", "v200", "v201", "v202", "v203", "m0");   \
    T1(); if (lane == 0 && wave == 0) out[n] = t1 - t0; n++;
__global__ __launch_bounds__(256) void probes(float* io, long long* out) {
    __shared__ float lds[8192];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = io[i & 255];
    __syncthreads();
    unsigned long long t0, t1;
    float a0 = io[lane], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = io[64 + lane], c = io[128 + lane];
    const unsigned addr = (unsigned)(size_t)(lds + lane), addr8 = (unsigned)(size_t)(lds + 2 * (lane & 15));
    int n = 0;
    if (wave == 0) {
        asm volatile("s_mov_b32 m0, %0" :: "s"((unsigned)(size_t)lds) : "m0");
        RUN(0, "s_nop 0")
        RUN(1, "ds_write_b32 %10, %0")
        RUN(2, "ds_write_b32 %10, %8")                 // data not just written by the VALU
        RUN(3, "ds_write_addtid_b32 %8 offset:256")
        RUN(4, "ds_write_b64 %11, v[200:201]")
        RUN(5, "ds_write_b128 %11, v[200:203]")
        RUN(6, "ds_read_b32 v200, %10")
        RUN(7, "ds_read_b64 v[200:201], %11")
        RUN(8, "ds_read2_b32 v[200:201], %10 offset0:0 offset1:16")
        RUN(9, "ds_read_b32 v200, %10\n s_waitcnt lgkmcnt(0)")
        RUN(10, "ds_bpermute_b32 v200, %10, %8")
        RUN(11, "ds_write2_b32 %10, %8, %9 offset0:0 offset1:64")
        RUN(12, "v_mov_b32_dpp v200, %8 row_bcast:15 row_mask:0x2 bank_mask:0xf")
        RUN(13, "ds_read_addtid_b32 v200 offset:256")
        RUN(14, "ds_write_b32 %10, %8\n ds_write_b32 %10, %9 offset:512")
    }
    io[512 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main() {
    float* io; long long* out;
    hipMalloc(&io, 4096 * sizeof(float)); hipMalloc(&out, 32 * sizeof(long long));
    float h[4096];
    for (int i = 0; i < 4096; i++) h[i] = 1.0f + 1e-3f * (i % 61);
    hipMemcpy(io, h, sizeof(h), hipMemcpyHostToDevice);
    const char* names[] = {"s_nop 0", "ds_write_b32 (fresh VALU data)", "ds_write_b32 (old data)", "ds_write_addtid_b32", "ds_write_b64", "ds_write_b128", "ds_read_b32",
                           "ds_read_b64", "ds_read2_b32", "ds_read_b32 + wait", "ds_bpermute_b32", "ds_write2_b32", "v_mov_dpp row_bcast:15", "ds_read_addtid_b32", "2 x ds_write_b32"};
    for (int rep = 0; rep < 2; rep++) {
        hipMemset(out, 0, 32 * sizeof(long long));
        hipLaunchKernelGGL(probes, dim3(1), dim3(256), 0, 0, io, out);
        hipDeviceSynchronize();
    }
    long long ho[32];
    hipMemcpy(ho, out, sizeof(ho), hipMemcpyDeviceToHost);
    for (int i = 0; i < 15; i++) printf("%-34s %8lld ticks  marginal %6.1f per probe instruction\n", names[i], ho[i], (double)(ho[i] - ho[0]) / 16.0 + 4.0);
    return 0;
}
