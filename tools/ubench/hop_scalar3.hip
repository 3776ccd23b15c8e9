// hop_scalar3.hip -- can a hand-off be PUBLISHED through the scalar path too (the CU's vector queue delays stores behind
// the bulk loads of all its wavefronts)?  Ping-pong between workgroup 0 and a partner on another XCD (block 1) / the same
// XCD (block 8); the 7 other wavefronts of both workgroups and all other workgroups stream HBM.
//   store 0: vector relaxed agent-scope atomic store (global_store sc1)           [what the kernels use]
//   store 1: s_store_dwordx2 glc + s_dcache_wb
//   store 2: s_atomic_swap_x2 (executes in the XCD's L2)
//   poll  0: vector sc1 load, 1: s_load_dwordx2 glc
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
typedef _Float16 h16;
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
#define G __attribute__((address_space(1)))
template <int ST>
__device__ __forceinline__ void put(u64* p, u64 v) {
    if (ST == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (ST == 1) asm volatile("s_store_dwordx2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)\n\ts_dcache_wb\n\ts_waitcnt lgkmcnt(0)" :: "s"(v), "s"(p) : "memory");
    else { u64 t = v; asm volatile("s_atomic_swap_x2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(t) : "s"(p) : "memory"); }
}
template <int PL>
__device__ __forceinline__ u64 get(const u64* p) {
    if (PL == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    u64 v;
    asm volatile("s_load_dwordx2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}
template <int ST, int PL>
__global__ __launch_bounds__(512) void k(u64* ping, u64* pong, int partner, int iters, u64* out, unsigned* xcc_out, const h16* w, int load_rows, float* sink) {
    const int b = blockIdx.x;
    const bool player = b == 0 || b == partner;
    if ((!player || threadIdx.x >= 64) && load_rows) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        float acc = 0.f;
        for (int rep = 0; rep < (player ? 3 : 1); ++rep)
        for (int r = 0; r < load_rows; r += 2) {
            h16x8 v[2][8];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    v[i][j] = __builtin_nontemporal_load((const G h16x8*)(w + ((size_t)(b * 8 + wave) * load_rows + r + i) * 4096 + (j * 64 + lane) * 8));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += (float)v[i][j][0];
        }
        if (acc == 1234.5f) sink[0] = acc;
        return;
    }
    if (!player || threadIdx.x != 0) return;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc_out[b != 0] = xcc & 15;
    u64 t0 = 0;
    bool fail = false;
    const u64* mine = b == 0 ? pong : ping;
    u64* theirs = b == 0 ? ping : pong;
    for (int i = 1; i <= iters && !fail; ++i) {
        if (i == 11) t0 = __builtin_amdgcn_s_memrealtime();
        if (b == 0) put<ST>(theirs, (u64)i);
        int spin = 0;
        while (true) {
            if (get<PL>(mine) == (u64)i) break;
            if (++spin > 1000000) { fail = true; break; }
        }
        if (b != 0 && !fail) put<ST>(theirs, (u64)i);
    }
    if (b == 0) { out[0] = __builtin_amdgcn_s_memrealtime() - t0; out[1] = fail; }
}
template <int ST, int PL>
void run(u64* buf, u64* out, unsigned* xo, int partner, const h16* w, int load_rows, float* sink) {
    hipMemset(buf, 0, 4096);
    hipDeviceSynchronize();
    const int iters = load_rows ? 40 : 510;
    hipLaunchKernelGGL((k<ST, PL>), dim3(256), dim3(512), 0, 0, buf, buf + 64, partner, iters, out, xo, w, load_rows, sink);
    hipError_t e = hipDeviceSynchronize();
    u64 h[2]; unsigned x[2];
    hipMemcpy(h, out, 16, hipMemcpyDeviceToHost); hipMemcpy(x, xo, 8, hipMemcpyDeviceToHost);
    const char* sn[3] = {"vector sc1", "s_store+wb", "s_atomic_swap"};
    printf("store %-13s poll %-6s partner %2d (xcc %u vs %u) %-9s: one-way %.3f us%s %s\n", sn[ST], PL ? "scalar" : "vector", partner, x[0], x[1],
           load_rows ? "streaming" : "idle", h[0] / 100.0 / (iters - 10) / 2, h[1] ? "  ** NEVER VISIBLE **" : "", e == hipSuccess ? "" : hipGetErrorString(e));
    fflush(stdout);
}
int main() {
    u64 *out, *buf; unsigned* xo; h16* w; float* sink;
    hipMalloc(&out, 64); hipMalloc(&xo, 8); hipMalloc(&sink, 4); hipMalloc(&buf, 4096);
    const size_t wbytes = (size_t)256 * 8 * 48 * 8192;
    hipMalloc(&w, wbytes); hipMemset(w, 1, wbytes);
    for (int partner : {1, 8})
        for (int load : {0, 48}) {
            run<0, 0>(buf, out, xo, partner, w, load, sink);
            run<0, 1>(buf, out, xo, partner, w, load, sink);
            run<1, 1>(buf, out, xo, partner, w, load, sink);
            run<2, 1>(buf, out, xo, partner, w, load, sink);
            run<1, 0>(buf, out, xo, partner, w, load, sink);
            run<2, 0>(buf, out, xo, partner, w, load, sink);
        }
    return 0;
}
