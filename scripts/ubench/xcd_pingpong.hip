// Hand-off latency between two workgroups, same XCD against another XCD, for the store / load flavours the persistent sweep could use.
// One "hop": the producer's 256 threads store a 16 KB tile (4 x 16 B per thread), every storing wavefront waits for vmcnt(0), a barrier, one lane
// raises a flag; the consumer polls the flag (one lane, s_sleep 2 between looks), barrier, its 256 threads load the tile (4 x 16 B per thread) and
// check it.  A round trip = two hops (A -> B -> A); the printed figure is the mean hop in shader cycles (s_memtime) over REPS round trips.
//   flavour 0: sc1 (write-through) stores, sc1 flag, sc1 loads          -- what chol_persist_kernel does today
//   flavour 1: plain stores (stay in the XCD's L2), sc1 flag store, sc1 loads (bypass L1, L2-served)  -- valid ONLY between workgroups of one XCD
//   flavour 2: plain stores, sc0 sc1 flag (atomic), plain-sc1 loads: as 1 with the flag as an agent-scope atomic add
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/bin/xcd_pingpong scripts/ubench/xcd_pingpong.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define REPS 200
#define TILE_FLOATS 4096

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const float* p, unsigned floats) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, floats * 4u, 0x00020000);
}
template <int AUX>
__device__ __forceinline__ void st(__amdgpu_buffer_rsrc_t b, unsigned off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), b, off * 4u, 0, AUX);
}
template <int AUX>
__device__ __forceinline__ f32x4 ld(__amdgpu_buffer_rsrc_t b, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b, off * 4u, 0, AUX));
}

struct Args {
    float* tiles;   // [2][TILE_FLOATS] : A's outbox, B's outbox (each pair has its own)
    int* flags;     // [2] per pair
    unsigned* info; // per block: xcc id
    long long* out; // per pair: cycles per hop, errors
    int b_of_pair[8];  // block index of the B side of pair p (the A side is block p)
    int npairs;
};

template <int FL>
__device__ void hop_send(__amdgpu_buffer_rsrc_t box, int* flag, int seq, int tid) {
    const f32x4 v = {(float)seq, (float)(seq + tid), (float)tid, 1.0f};
#pragma unroll
    for (int it = 0; it < 4; it++) {
        if (FL == 0) st<16>(box, (unsigned)(tid + it * 256) * 4u, v);
        else st<0>(box, (unsigned)(tid + it * 256) * 4u, v);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        if (FL == 2) __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
template <int FL>
__device__ int hop_recv(__amdgpu_buffer_rsrc_t box, int* flag, int seq, int tid) {
    if (tid < 64) {
        int spins = 0;
        for (;;) {
            int v = seq;
            if (tid == 0) v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__ballot(v < seq) == 0ull || ++spins > (1 << 22)) break;
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    int bad = 0;
#pragma unroll
    for (int it = 0; it < 4; it++) {
        const f32x4 v = ld<16>(box, (unsigned)(tid + it * 256) * 4u);
        bad += (v[0] != (float)seq) || (v[1] != (float)(seq + tid));
    }
    return bad;
}

template <int FL>
__global__ __launch_bounds__(256) void pingpong(Args a) {
    extern __shared__ float pad[];  // > half the LDS: one workgroup per compute unit
    const int tid = threadIdx.x, b = blockIdx.x;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (tid == 0) a.info[b] = xcc & 15;
    int pair = -1, side = 0;
    for (int p = 0; p < a.npairs; p++) {
        if (b == p) pair = p, side = 0;
        if (b == a.b_of_pair[p]) pair = p, side = 1;
    }
    if (pair < 0) return;
    float* base = a.tiles + (size_t)pair * 2 * TILE_FLOATS;
    const __amdgpu_buffer_rsrc_t boxA = rsrc(base, TILE_FLOATS), boxB = rsrc(base + TILE_FLOATS, TILE_FLOATS);
    int* fA = a.flags + 64 * (2 * pair), *fB = a.flags + 64 * (2 * pair + 1);
    int bad = 0;
    long long t0 = 0;
    for (int r = 1; r <= REPS + 10; r++) {
        if (r == 11 && side == 0) t0 = (long long)__builtin_amdgcn_s_memtime();
        const int seq = (FL == 2) ? r : r;
        if (side == 0) {
            hop_send<FL>(boxA, fA, seq, tid);
            bad += hop_recv<FL>(boxB, fB, seq, tid);
        } else {
            bad += hop_recv<FL>(boxA, fA, seq, tid);
            hop_send<FL>(boxB, fB, seq, tid);
        }
    }
    if (side == 0 && tid == 0) a.out[2 * pair] = ((long long)__builtin_amdgcn_s_memtime() - t0) / (2 * REPS);
    bad = __syncthreads_count(bad != 0);
    if (tid == 0) atomicAdd(reinterpret_cast<unsigned long long*>(a.out + 2 * pair + 1), (unsigned long long)bad);
}

int main() {
    const int NB = 64;
    Args a;
    a.npairs = 4;
    // pairs: A side = blocks 0..3; B side: 8, 16+1 (same b % 8 as 1 -> same XCD if round-robin), and two on other XCDs
    const int bs[4] = {8, 17, 4 + 8 * 3, 5 + 8 * 4};  // pair 0: (0,8) same; pair 1: (1,17) same; pair 2: (2,28) other; pair 3: (3,37) other
    for (int p = 0; p < 8; p++) a.b_of_pair[p] = p < 4 ? bs[p] : -1;
    hipMalloc(&a.tiles, sizeof(float) * 8 * 2 * TILE_FLOATS);
    hipMalloc(&a.flags, sizeof(int) * 64 * 16);
    hipMalloc(&a.info, sizeof(unsigned) * NB);
    hipMalloc(&a.out, sizeof(long long) * 16);
    const char* names[3] = {"sc1 stores + sc1 flag + sc1 loads (today)", "plain stores + sc1 flag + sc1 loads", "plain stores + atomic flag + sc1 loads"};
    for (int fl = 0; fl < 3; fl++) {
        hipMemset(a.tiles, 0, sizeof(float) * 8 * 2 * TILE_FLOATS);
        hipMemset(a.flags, 0, sizeof(int) * 64 * 16);
        hipMemset(a.out, 0, sizeof(long long) * 16);
        const size_t lds = 84 * 1024;
        if (fl == 0) { hipFuncSetAttribute(reinterpret_cast<const void*>(pingpong<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); hipLaunchKernelGGL(pingpong<0>, dim3(NB), dim3(256), lds, 0, a); }
        if (fl == 1) { hipFuncSetAttribute(reinterpret_cast<const void*>(pingpong<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); hipLaunchKernelGGL(pingpong<1>, dim3(NB), dim3(256), lds, 0, a); }
        if (fl == 2) { hipFuncSetAttribute(reinterpret_cast<const void*>(pingpong<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds); hipLaunchKernelGGL(pingpong<2>, dim3(NB), dim3(256), lds, 0, a); }
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        std::vector<unsigned> info(NB);
        long long out[16];
        hipMemcpy(info.data(), a.info, sizeof(unsigned) * NB, hipMemcpyDeviceToHost);
        hipMemcpy(out, a.out, sizeof(out), hipMemcpyDeviceToHost);
        printf("%s\n", names[fl]);
        for (int p = 0; p < 4; p++)
            printf("   pair (%d,%d): xcc %u / %u  %s  %lld cycles per hop (16 KB tile + flag), %lld threads saw stale data\n", p, bs[p], info[p], info[bs[p]],
                   info[p] == info[bs[p]] ? "SAME XCD " : "other XCD", out[2 * p], out[2 * p + 1]);
    }
    return 0;
}
