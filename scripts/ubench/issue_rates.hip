// Single-wavefront instruction issue / latency probes for gfx950 (diagnostic, not part of the library).
// Build: hipcc -O2 --offload-arch=gfx950 -o /tmp/issue_rates scripts/ubench/issue_rates.hip
// Each probe runs REP copies of a small instruction group between two s_memtime reads in ONE wave
// (or 4 waves for the barrier probe) and reports shader-clock ticks per instruction.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define T0() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory")
#define T1() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory")

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void probes(float* io, long long* out) {
    __shared__ float lds[4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = io[i & 255];
    __syncthreads();
    unsigned long long t0, t1;
    float a0 = io[lane], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = io[64 + lane], c = io[128 + lane];
    int n = 0;
    if (wave == 0) {
        // 0: empty
        T0(); T1(); if (lane == 0) out[n] = t1 - t0; n++;
        // 1: 256 independent v_fma (8 accumulators)
        T0();
        asm volatile(".rept 32\n v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n"
                     "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        T1(); if (lane == 0) out[n] = t1 - t0; n++;
        // 2: 256 dependent v_fma
        T0();
        asm volatile(".rept 256\n v_fma_f32 %0, %1, %2, %0\n.endr" : "+v"(a0) : "v"(b), "v"(c));
        T1(); if (lane == 0) out[n] = t1 - t0; n++;
        // 3: 256 independent v_fmac_dpp row_newbcast
        T0();
        asm volatile(".rept 32\n v_fmac_f32_dpp %0, -%8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, -%8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
                     "v_fmac_f32_dpp %2, -%8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, -%8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                     "v_fmac_f32_dpp %4, -%8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %5, -%8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
                     "v_fmac_f32_dpp %6, -%8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %7, -%8, %9 row_newbcast:8 row_mask:0xf bank_mask:0xf\n.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        T1(); if (lane == 0) out[n] = t1 - t0; n++;
        // 4: 256 v_readlane (independent, 8 SGPR destinations)
        T0();
        asm volatile(".rept 32\n v_readlane_b32 s20, %0, 1\n v_readlane_b32 s21, %0, 2\n v_readlane_b32 s22, %0, 3\n v_readlane_b32 s23, %0, 4\n"
                     "v_readlane_b32 s24, %0, 5\n v_readlane_b32 s25, %0, 6\n v_readlane_b32 s26, %0, 7\n v_readlane_b32 s27, %0, 8\n.endr"
                     : : "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        T1(); if (lane == 0) out[n] = t1 - t0; n++;
        // 5: 128 x (v_readlane ; v_fma with that SGPR)  = 256 instructions, independent accumulators
        T0();
        asm volatile(".rept 32\n v_readlane_b32 s20, %4, 1\n v_readlane_b32 s21, %4, 2\n v_readlane_b32 s22, %4, 3\n v_readlane_b32 s23, %4, 4\n"
                     "v_fma_f32 %0, %5, s20, %0\n v_fma_f32 %1, %5, s21, %1\n v_fma_f32 %2, %5, s22, %2\n v_fma_f32 %3, %5, s23, %3\n.endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "s20", "s21", "s22", "s23");
        T1(); if (lane == 0) out[n] = t1 - t0; n++;
        // 6: 64 dependent v_rsq
        T0();
        asm volatile(".rept 64\n v_rsq_f32 %0, %0\n s_nop 0\n.endr" : "+v"(a1));
        T1(); if (lane == 0) out[n] = t1 - t0; n++;
        // 7: 256 independent v_mov_dpp
        T0();
        asm volatile(".rept 64\n v_mov_b32_dpp %0, %4 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
                     "v_mov_b32_dpp %2, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 row_newbcast:4 row_mask:0xf bank_mask:0xf\n.endr"
                     : "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        T1(); if (lane == 0) out[n] = t1 - t0; n++;
        // 8: 64 dependent mfma 16x16x4 (acc chain)
        {
            f32x4 acc = {0, 0, 0, 0};
            T0();
            asm volatile(".rept 64\n v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n.endr\n s_nop 7" : "+v"(acc) : "v"(b), "v"(c));
            T1(); if (lane == 0) out[n] = t1 - t0; n++;
            a2 += acc[0];
        }
        // 9: 64 dependent mfma 32x32x2
        {
            f32x16 acc;
            for (int i = 0; i < 16; i++) acc[i] = 0;
            T0();
            asm volatile(".rept 64\n v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n.endr\n s_nop 15" : "+v"(acc) : "v"(b), "v"(c));
            T1(); if (lane == 0) out[n] = t1 - t0; n++;
            a3 += acc[0];
        }
        // 10: 64 dependent LDS reads (pointer chase through zeros): latency
        {
            int addr = lane * 4;
            for (int i = threadIdx.x; i < 64; i += 64) ((int*)lds)[i] = i * 4;
            T0();
            asm volatile(".rept 64\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n.endr" : "+v"(addr));
            T1(); if (lane == 0) out[n] = t1 - t0; n++;
            a4 += addr;
        }
        // 11: 64 x (ds_write ; ds_read same address ; wait): LDS round trip
        {
            int addr = 1024 + lane * 4;
            T0();
            asm volatile(".rept 64\n ds_write_b32 %1, %0\n ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n.endr" : "+v"(a5) : "v"(addr));
            T1(); if (lane == 0) out[n] = t1 - t0; n++;
        }
        // 12: 256 dependent v_fmac_dpp chain (same accumulator)
        T0();
        asm volatile(".rept 256\n v_fmac_f32_dpp %0, -%1, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf\n.endr" : "+v"(a0) : "v"(b), "v"(c));
        T1(); if (lane == 0) out[n] = t1 - t0; n++;
        // 13: 128 x (permlane32_swap) independent
        T0();
        asm volatile(".rept 64\n v_permlane32_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n.endr" : "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        T1(); if (lane == 0) out[n] = t1 - t0; n++;
        // 14: 256 x s_nop 0
        T0();
        asm volatile(".rept 256\n s_nop 0\n.endr");
        T1(); if (lane == 0) out[n] = t1 - t0; n++;
        // 15: 256 SALU adds (dependent)
        T0();
        asm volatile(".rept 256\n s_add_u32 s20, s20, 1\n.endr" ::: "s20", "scc");
        T1(); if (lane == 0) out[n] = t1 - t0; n++;
        // 16: 128 x (v_mul ; dependent v_readlane ; dependent v_mul using sgpr) = pivot-chain pattern, 384 instr
        T0();
        asm volatile(".rept 128\n v_mul_f32 %0, %0, %1\n s_nop 0\n v_readlane_b32 s20, %0, 3\n s_nop 1\n v_mul_f32 %0, s20, %0\n.endr" : "+v"(a6) : "v"(c) : "s20");
        T1(); if (lane == 0) out[n] = t1 - t0; n++;
    } else {
        n = 17;
    }
    // 17: 64 barriers with all waves of the block
    __syncthreads();
    T0();
    asm volatile(".rept 64\n s_barrier\n.endr");
    T1(); if (threadIdx.x == 0) out[17] = t1 - t0;
    io[512 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main() {
    float* io; long long* out;
    hipMalloc(&io, 4096 * sizeof(float)); hipMalloc(&out, 32 * sizeof(long long));
    std::vector<float> h(4096, 1.0f);
    for (int i = 0; i < 4096; i++) h[i] = 1.0f + 1e-3f * (i % 61);
    hipMemcpy(io, h.data(), 4096 * sizeof(float), hipMemcpyHostToDevice);
    const char* names[] = {"empty", "fma indep x256", "fma dep x256", "fmac_dpp indep x256", "readlane indep x256", "readlane+fma x128 pairs (256)",
                           "rsq dep x64 (+nop)", "mov_dpp indep x256", "mfma16x16x4 dep x64", "mfma32x32x2 dep x64", "lds read chase x64",
                           "lds write+read x64", "fmac_dpp dep x256", "permlane32/16 swap x128", "s_nop x256", "salu add dep x256",
                           "mul->readlane->mul x128 (384)", "s_barrier x64 (4 waves)"};
    const int cnt[] = {1, 256, 256, 256, 256, 256, 64, 256, 64, 64, 64, 64, 256, 128, 256, 256, 384, 64};
    for (int rep = 0; rep < 2; rep++) {
        hipMemset(out, 0, 32 * sizeof(long long));
        hipLaunchKernelGGL(probes, dim3(1), dim3(256), 0, 0, io, out);
        hipDeviceSynchronize();
    }
    long long ho[32];
    hipMemcpy(ho, out, sizeof(ho), hipMemcpyDeviceToHost);
    for (int i = 0; i < 18; i++) printf("%-34s %8lld ticks  %7.2f / instr\n", names[i], ho[i], (double)(ho[i] - ho[0]) / cnt[i]);
    return 0;
}
