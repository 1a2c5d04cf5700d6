"""Writes /tmp/split_probe.hip: single-wavefront timing of the split panel's two instruction streams (scripts/gen_potrf_chain.py,
gen_split_diag / gen_split_x) and of variants with one instruction class removed, to price the classes.
  python scripts/ubench/gen_split_probe.py > /tmp/split_probe.hip && hipcc -O2 --offload-arch=gfx950 -o /tmp/split_probe /tmp/split_probe.hip"""
import os, re, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gen_potrf_chain as G

def diag_variant(kind):
    out = []
    for ln in G.gen_split_diag():
        if kind == "nowrite" and ln.startswith("ds_write"):
            ln = "s_nop 0"
        if kind == "nodpp" and "_dpp" in ln:
            ln = re.sub(r" row_newbcast.*", "", ln.replace("_dpp", "")).replace("-%[", "%[")
        if kind == "norsq" and ln.startswith("v_rsq"):
            ln = re.sub(r"v_rsq_f32_dpp (\S+), (\S+) row.*", r"v_mov_b32 \1 \2", ln)
        if kind == "fmaconly" and not ln.startswith("v_fmac"):
            continue
        if kind == "nowait" and ln.startswith("s_waitcnt"):
            continue
        out.append(ln)
    return out

def x_variant(kind):
    out = []
    skipping = False
    for ln in G.gen_split_x():
        if kind == "nopoll" and (ln.startswith("v_cmp") or ln.startswith("s_cbranch")):
            continue
        if kind == "fmaconly" and not ln.startswith("v_fmac"):
            continue
        if kind == "nopoll" and (ln.startswith("s_cbranch") or ln.startswith("v_cmp")):
            continue
        out.append(ln)
    return out

def block(lines, regs_out, extra_in, clob):
    s = "    asm volatile(\n"
    for ln in lines:
        s += '        "%s\\n\\t"\n' % ln
    s += "        : " + regs_out + "\n        : " + extra_in + "\n        : " + clob + ");\n"
    return s

L = ", ".join('[l%d] "+v"(l[%d])' % (i, i) for i in range(16)) + ', [t] "+v"(t)'
X = ", ".join('[x%d] "+v"(x[%d])' % (i, i) for i in range(16)) + ", " + ", ".join('[t%d] "=&v"(rt[%d]), [l%d] "=&v"(rl[%d])' % (i, i, i, i) for i in range(G.SPLIT_X_RING)) 
print(r'''#include <hip/hip_runtime.h>
#include <cstdio>
#define PLD 68
#define T0() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory")
#define T1() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory")
__global__ __launch_bounds__(256) void probes(float* io, long long* out, int nwaves) {
    __shared__ float A[64 * PLD];
    __shared__ float Ex[1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * PLD; i += blockDim.x) A[i] = 1.0f + 0.001f * (i % 37) + ((i / PLD) == (i % PLD) ? 40.f : 0.f);
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) Ex[i] = 0.5f;
    __syncthreads();
    unsigned long long t0, t1;
    float l[16], x[16], t = 1.f, rt[4], rl[4];
    for (int j = 0; j < 16; j++) { l[j] = io[lane + j]; x[j] = io[64 + lane + j]; }
    const unsigned lbd = (unsigned)(size_t)(A + (lane & 15));
    const unsigned exw = (unsigned)(size_t)(Ex + lane), exr = (unsigned)(size_t)(Ex + (lane & 15)), ext = (unsigned)(size_t)(Ex + 16);
    unsigned sb = (unsigned)(size_t)(A + 16 * PLD + lane);
    const unsigned ss = 4 * PLD;
    int n = 0;
    if (wave < nwaves) {
''')
names = []
for kind in ("full", "nowrite", "nodpp", "norsq", "fmaconly", "nowait"):
    print("        for (int j = 0; j < 16; j++) l[j] = io[lane + j] + ((lane & 15) == j ? 40.f : 0.f);")
    print("        T0();")
    print(block(diag_variant(kind), L, '[lbd] "v"(lbd), [exw] "v"(exw), [p4] "n"(4 * PLD)', '"memory"'))
    print("        T1(); if (lane == 0 && wave == 0) out[n] = t1 - t0; n++;")
    names.append("diag " + kind + " (%d instr)" % len([q for q in diag_variant(kind) if not q.startswith(";")]))
for kind in ("full", "nopoll", "fmaconly"):
    print("        __builtin_amdgcn_s_waitcnt(0); for (int q = lane; q < 1024; q += 64) Ex[q] = 0.5f;")
    print("        for (int j = 0; j < 16; j++) x[j] = io[64 + lane + j];")
    print("        __builtin_amdgcn_s_waitcnt(0); sb = (unsigned)(size_t)(A + 16 * PLD + lane); T0();")
    print(block(x_variant(kind), X, '[exr] "v"(exr), [ext] "v"(ext), [sb] "v"(sb), [p4] "n"(4 * PLD)', '"memory", "vcc"'))
    print("        T1(); if (lane == 0 && wave == 0) out[n] = t1 - t0; n++;")
    names.append("x " + kind + " (%d instr)" % len([q for q in x_variant(kind) if not q.startswith(";") and not q.startswith(".L")]))
print(r'''    }
    float acc = t;
    for (int j = 0; j < 16; j++) acc += l[j] + x[j] + rt[j & 3] + rl[j & 3];
    io[512 + threadIdx.x] = acc;
}
int main() {
    float* io; long long* out;
    hipMalloc(&io, 4096 * sizeof(float)); hipMalloc(&out, 32 * sizeof(long long));
    float h[4096];
    for (int i = 0; i < 4096; i++) h[i] = 1.0f + 1e-3f * (i %% 61);
    hipMemcpy(io, h, sizeof(h), hipMemcpyHostToDevice);
    const char* names[] = {%s};
    for (int nw = 1; nw <= 4; nw *= 2) {
        for (int rep = 0; rep < 2; rep++) {
            hipMemset(out, 0, 32 * sizeof(long long));
            hipLaunchKernelGGL(probes, dim3(1), dim3(256), 0, 0, io, out, nw);
            hipDeviceSynchronize();
        }
        long long ho[32];
        hipMemcpy(ho, out, sizeof(ho), hipMemcpyDeviceToHost);
        printf("-- %%d wavefront(s) running the stream\n", nw);
        for (int i = 0; i < %d; i++) printf("%%-34s %%8lld ticks\n", names[i], ho[i]);
    }
    return 0;
}''' % (", ".join('"%s"' % n for n in names), len(names)))
