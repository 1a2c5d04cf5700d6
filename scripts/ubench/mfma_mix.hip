// MFMA issue-rate probes for gfx950 (diagnostic): how fast do v_mfma_f32_16x16x4_f32 / 32x32x2 issue from one or
// two wavefronts per SIMD, alone and interleaved with the LDS operand reads of a GEMM inner loop?
// Build: hipcc -O2 --offload-arch=gfx950 -o /tmp/mfma_mix scripts/ubench/mfma_mix.hip
#include <hip/hip_runtime.h>

#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define T0() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory")
#define T1() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory")

// mode 0: 192 x 16x16x4, 3 independent accumulators, register operands
// mode 1: the same + 4 ds_read_b32 per 3 MFMAs (operands for two steps later), compiler-counted waits
// mode 2: 96 x 32x32x2, 1 accumulator, register operands
// mode 3: 96 x 32x32x2 + 2 ds_read_b32 per MFMA
// mode 4: 192 x 16x16x4 with 4 independent accumulators
__global__ void probe(float* io, long long* out, int mode) {
    __shared__ float lds[16384];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 16384; i += blockDim.x) lds[i] = io[i & 1023];
    __syncthreads();
    unsigned long long t0, t1;
    float b = io[lane], c = io[64 + lane];
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    f32x16 A;
    for (int i = 0; i < 16; i++) A[i] = 0;
    const float* p = lds + (lane & 15) + (lane >> 4) * 48 + (tid >> 6) * 16;
    T0();
    if (mode == 0) {
#pragma unroll
        for (int s = 0; s < 64; s++) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, c, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, c, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, c, a2, 0, 0, 0);
        }
    } else if (mode == 1) {
        float f[4][4];
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int r = 0; r < 4; r++) f[q][r] = p[(4 * q) * 48 + 16 * r];
#pragma unroll
        for (int s = 0; s < 64; s++) {
            if (s + 2 < 64) {
#pragma unroll
                for (int r = 0; r < 4; r++) f[(s + 2) % 4][r] = p[(4 * ((s + 2) % 16)) * 48 + 16 * r];
            }
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f[s % 4][3], f[s % 4][0], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f[s % 4][3], f[s % 4][1], a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(f[s % 4][3], f[s % 4][2], a2, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if (mode == 2) {
#pragma unroll
        for (int s = 0; s < 96; s++) A = __builtin_amdgcn_mfma_f32_32x32x2f32(b, c, A, 0, 0, 0);
    } else if (mode == 3) {
        float f[4][2];
#pragma unroll
        for (int q = 0; q < 2; q++) { f[q][0] = p[q * 130]; f[q][1] = p[q * 130 + 65]; }
#pragma unroll
        for (int s = 0; s < 96; s++) {
            if (s + 2 < 96) { f[(s + 2) % 4][0] = p[((s + 2) % 16) * 130]; f[(s + 2) % 4][1] = p[((s + 2) % 16) * 130 + 65]; }
            A = __builtin_amdgcn_mfma_f32_32x32x2f32(f[s % 4][0], f[s % 4][1], A, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
#pragma unroll
        for (int s = 0; s < 48; s++) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, c, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, c, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, c, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, c, a3, 0, 0, 0);
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    T1();
    if (lane == 0) out[tid >> 6] = t1 - t0;
    io[2048 + tid] = a0[0] + a1[0] + a2[0] + a3[0] + A[0];
}

int main() {
    float* io; long long* out;
    hipMalloc(&io, 8192 * sizeof(float)); hipMalloc(&out, 64 * sizeof(long long));
    hipMemset(io, 0, 8192 * sizeof(float));
    const char* names[] = {"16x16x4 x192, 3 acc, reg operands", "16x16x4 x192, 3 acc + 4 ds_read/step", "32x32x2 x96, reg operands",
                           "32x32x2 x96 + 2 ds_read/MFMA", "16x16x4 x192, 4 acc, reg operands"};
    const int mf[] = {192, 192, 96, 96, 192};
    for (int mode = 0; mode < 5; mode++)
        for (int threads : {256, 512}) {
            long long ho[8];
            for (int rep = 0; rep < 2; rep++) {
                hipLaunchKernelGGL(probe, dim3(1), dim3(threads), 0, 0, io, out, mode);
                hipDeviceSynchronize();
            }
            hipMemcpy(ho, out, sizeof(ho), hipMemcpyDeviceToHost);
            long long mx = 0;
            for (int w = 0; w < threads / 64; w++) mx = ho[w] > mx ? ho[w] : mx;
            printf("%-40s %d waves/SIMD: %6lld ticks, %6.1f per MFMA per wave, pipe busy %.0f%%\n", names[mode], threads / 256, mx,
                   (double)mx / mf[mode], 100.0 * mf[mode] * (threads / 256) * (mode == 2 || mode == 3 ? 64 : 32) / mx);
        }
    return 0;
}
