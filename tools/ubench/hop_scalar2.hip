// hop_scalar2.hip -- scalar-path polling across XCDs, the hard cases: (a) one launch per epoch (the polled line still holds
// the previous launch's tag when the poller starts), (b) the line was ALSO read with vector sc1 loads by the poller's CU
// before the update (a clean copy in the poller's L2?), (c) 64 granules swept by one wavefront with s_load_dwordx16.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
__device__ __forceinline__ u64 get_vec(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 get_scalar(const u64* p) {
    u64 v;
    asm volatile("s_load_dwordx2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}
// mode bit 0: poller reads the line with a vector sc1 load first (sees the old tag most likely), bit 1: producer delays
__global__ void k(u64* g, unsigned epoch, int mode, unsigned* result, int consumer_block) {
    const int b = blockIdx.x;
    if (threadIdx.x != 0) return;
    if (b == 0) {
        if (mode & 2) __builtin_amdgcn_s_sleep(127);
        for (int i = 0; i < 64; ++i) __hip_atomic_store(g + i, ((u64)epoch << 32) | (u64)(epoch * 64 + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (b == consumer_block) {
        unsigned vecseen = 0;
        if (mode & 1) vecseen = (unsigned)(get_vec(g + 63) >> 32);
        unsigned spin = 0;
        bool ok = false;
        while (spin < 2000000) {
            const u64 v = get_scalar(g + 63);
            if ((unsigned)(v >> 32) == epoch) { ok = true; break; }
            ++spin;
        }
        // all 64 granules through s_load_dwordx16 x 8
        unsigned bad = 0;
        if (ok) {
            for (int c = 0; c < 8; ++c) {
                unsigned d[16];
                typedef unsigned u16v __attribute__((ext_vector_type(16)));
                u16v v;
                asm volatile("s_load_dwordx16 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(g + c * 8) : "memory");
                for (int i = 0; i < 8; ++i) { if (v[2 * i + 1] != epoch || v[2 * i] != epoch * 64 + c * 8 + i) ++bad; }
            }
        }
        result[0] = ok ? spin : 0xffffffffu;
        result[1] = bad;
        result[2] = vecseen;
    }
}
int main() {
    u64* g; unsigned* res;
    hipMalloc(&g, 4096); hipMalloc(&res, 64); hipMemset(g, 0, 4096);
    for (int consumer : {1, 8}) for (int mode = 0; mode < 4; ++mode) {
        int stale = 0, badsum = 0; double spins = 0;
        for (unsigned e = 1; e <= 200; ++e) {
            hipLaunchKernelGGL(k, dim3(16), dim3(64), 0, 0, g, e + consumer * 1000 + mode * 10000, mode, res, consumer);
            hipDeviceSynchronize();
            unsigned h[3]; hipMemcpy(h, res, 12, hipMemcpyDeviceToHost);
            if (h[0] == 0xffffffffu) ++stale; else spins += h[0];
            badsum += h[1];
        }
        printf("consumer block %d mode %d (vec-read-first %d, producer-delay %d): %d / 200 launches never saw the new tag, bad granules in x16 sweeps %d, mean spins %.1f\n",
               consumer, mode, mode & 1, (mode >> 1) & 1, stale, badsum, spins / (200 - stale + 1e-9));
    }
    return 0;
}
