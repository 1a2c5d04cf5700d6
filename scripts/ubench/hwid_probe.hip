#include <hip/hip_runtime.h>
__global__ void k(unsigned* o) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) { o[2 * blockIdx.x] = hw; o[2 * blockIdx.x + 1] = xcc; }
}
int main() {
    unsigned* d; hipMalloc(&d, 8 * 2048);
    hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, d);
    unsigned h[4096]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 24; b++) printf("wg %d hw %08x cu %u sh %u se %u xcc %u\n", b, h[2*b], (h[2*b]>>8)&15, (h[2*b]>>12)&1, (h[2*b]>>13)&7, h[2*b+1]&15);
    // count distinct keys
    int cnt[4096] = {0};
    for (int b = 0; b < 2048; b++) cnt[((h[2*b+1]&15)<<8) | ((h[2*b]>>8)&0xff)]++;
    int keys = 0, mx = 0; for (int i = 0; i < 4096; i++) if (cnt[i]) { keys++; if (cnt[i] > mx) mx = cnt[i]; }
    printf("distinct cu keys %d, max wgs per key %d\n", keys, mx);
    return 0;
}
