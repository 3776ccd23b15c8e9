// hop_lat.hip -- one-way latency of an inter-workgroup hand-off (8-byte tagged granule) on gfx950:
// ping-pong between workgroup 0 and a partner on the same XCD (block 8) or another XCD (block 1),
// for combinations of store kind and poll kind.
//   store 0: relaxed agent-scope atomic store (global_store sc1, write-through)   [what the kernels use]
//   store 1: relaxed workgroup-scope atomic store (plain store, lands in the XCD's L2)
//   poll  0: relaxed agent-scope atomic load (global_load sc1)                     [what the kernels use]
//   poll  1: workgroup-scope fetch_or(0) (atomic executed in the XCD's L2)
//   poll  2: workgroup-scope atomic load (may be served by the CU's L1)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
template <int ST>
__device__ __forceinline__ void put(u64* p, u64 v) {
    if (ST == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <int PL>
__device__ __forceinline__ u64 get(u64* p) {
    if (PL == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (PL == 1) return __hip_atomic_fetch_or(p, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <int ST, int PL>
__global__ void k(u64* ping, u64* pong, int partner, int iters, u64* out, unsigned* xcc_out) {
    const int b = blockIdx.x;
    if (threadIdx.x != 0 || (b != 0 && b != partner)) return;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc_out[b != 0] = xcc & 15;
    u64 t0 = 0;
    bool fail = false;
    for (int i = 1; i <= iters && !fail; ++i) {
        if (i == 11) t0 = __builtin_amdgcn_s_memrealtime();
        if (b == 0) {
            put<ST>(ping, (u64)i);
            int spin = 0;
            while (get<PL>(pong) != (u64)i) if (++spin > 2000000) { fail = true; break; }
        } else {
            int spin = 0;
            while (get<PL>(ping) != (u64)i) if (++spin > 2000000) { fail = true; break; }
            put<ST>(pong, (u64)i);
        }
    }
    if (b == 0) { out[0] = __builtin_amdgcn_s_memrealtime() - t0; out[1] = fail; }
}
template <int ST, int PL>
void run(u64* buf, u64* out, unsigned* xo, int partner) {
    hipMemset(buf, 0, 4096);
    const int iters = 2010;
    hipLaunchKernelGGL((k<ST, PL>), dim3(16), dim3(64), 0, 0, buf, buf + 64, partner, iters, out, xo);
    hipDeviceSynchronize();
    u64 h[2]; unsigned x[2];
    hipMemcpy(h, out, 16, hipMemcpyDeviceToHost); hipMemcpy(x, xo, 8, hipMemcpyDeviceToHost);
    printf("store %d poll %d partner block %2d (xcc %u vs %u): one-way %.3f us%s\n", ST, PL, partner, x[0], x[1],
           h[0] / 100.0 / 2000 / 2, h[1] ? "  ** TIMED OUT (never became visible) **" : "");
}
int main() {
    u64 *buf, *out; unsigned* xo;
    hipMalloc(&buf, 4096); hipMalloc(&out, 64); hipMalloc(&xo, 8);
    for (int partner : {8, 1}) {
        run<0, 0>(buf, out, xo, partner);
        run<0, 1>(buf, out, xo, partner);
        run<1, 1>(buf, out, xo, partner);
        run<0, 2>(buf, out, xo, partner);
        run<1, 2>(buf, out, xo, partner);
        run<1, 0>(buf, out, xo, partner);
    }
    return 0;
}
