// hop_scalar.hip -- can a hand-off be polled through the SCALAR memory path (s_load_dwordx2 glc: SMEM, lgkmcnt -- it does not
// queue behind the wavefront's / the CU's vector loads) and still see another XCD's stores?  Ping-pong between workgroup 0
// and a partner on another XCD (block 1) or the same XCD (block 8), tagged 8-byte granules at FIXED addresses (so a
// stale cached copy would show), for memory kinds {default, fine-grained, uncached} and poll kinds {vector sc1, scalar glc},
// idle and with the other 254 workgroups streaming HBM (each CU ~128 KB in flight).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
typedef _Float16 h16;
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
#define G __attribute__((address_space(1)))
__device__ __forceinline__ u64 get_vec(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 get_scalar(const u64* p) {
    u64 v;
    asm volatile("s_load_dwordx2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}
template <int PL>
__global__ __launch_bounds__(512) void k(u64* ping, u64* pong, int partner, int iters, u64* out, unsigned* xcc_out, const h16* w, int load_rows, float* sink) {
    const int b = blockIdx.x;
    if (b != 0 && b != partner) {   // background: stream HBM
        if (!load_rows) return;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        float acc = 0.f;
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
    if (threadIdx.x >= 64 && load_rows) {   // the ping-pong workgroup's other 7 wavefronts stream too (the CU's vector queue is deep)
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        float acc = 0.f;
        for (int rep = 0; rep < 3; ++rep)
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
    if (threadIdx.x != 0) return;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc_out[b != 0] = xcc & 15;
    u64 t0 = 0;
    bool fail = false;
    const u64* mine = b == 0 ? pong : ping;
    u64* theirs = b == 0 ? ping : pong;
    for (int i = 1; i <= iters && !fail; ++i) {
        if (i == 11) t0 = __builtin_amdgcn_s_memrealtime();
        if (b == 0) __hip_atomic_store(theirs, (u64)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spin = 0;
        while (true) {
            const u64 v = PL == 0 ? get_vec(mine) : get_scalar(mine);
            if (v == (u64)i) break;
            if (++spin > 3000000) { fail = true; break; }
        }
        if (b != 0 && !fail) __hip_atomic_store(theirs, (u64)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (b == 0) { out[0] = __builtin_amdgcn_s_memrealtime() - t0; out[1] = fail; }
}
template <int PL>
void run(const char* mem, u64* buf, u64* out, unsigned* xo, int partner, const h16* w, int load_rows, float* sink) {
    hipMemset(buf, 0, 4096);
    hipDeviceSynchronize();
    const int iters = load_rows ? 40 : 1010;
    hipLaunchKernelGGL((k<PL>), dim3(256), dim3(512), 0, 0, buf, buf + 64, partner, iters, out, xo, w, load_rows, sink);
    hipError_t e = hipDeviceSynchronize();
    u64 h[2]; unsigned x[2];
    hipMemcpy(h, out, 16, hipMemcpyDeviceToHost); hipMemcpy(x, xo, 8, hipMemcpyDeviceToHost);
    printf("%-12s poll %-7s partner block %2d (xcc %u vs %u) %-9s: one-way %.3f us%s %s\n", mem, PL ? "scalar" : "vector", partner, x[0], x[1],
           load_rows ? "streaming" : "idle", h[0] / 100.0 / (iters - 10) / 2, h[1] ? "  ** TIMED OUT (stale) **" : "", e == hipSuccess ? "" : hipGetErrorString(e));
    fflush(stdout);
}
int main() {
    u64 *out; unsigned* xo; h16* w; float* sink;
    hipMalloc(&out, 64); hipMalloc(&xo, 8); hipMalloc(&sink, 4);
    const size_t wbytes = (size_t)256 * 8 * 48 * 8192;   // 48 rows per wavefront = 805 MB
    hipMalloc(&w, wbytes); hipMemset(w, 1, wbytes);
    const char* names[3] = {"default", "fine-grained", "uncached"};
    const unsigned flags[3] = {hipDeviceMallocDefault, hipDeviceMallocFinegrained, hipDeviceMallocUncached};
    for (int m = 0; m < 3; ++m) {
        u64* buf = nullptr;
        hipError_t e = hipExtMallocWithFlags((void**)&buf, 4096, flags[m]);
        if (e != hipSuccess) { printf("%s: alloc failed: %s\n", names[m], hipGetErrorString(e)); continue; }
        for (int partner : {1, 8})
            for (int load : {0, 48}) {
                run<0>(names[m], buf, out, xo, partner, w, load, sink);
                run<1>(names[m], buf, out, xo, partner, w, load, sink);
            }
        hipFree(buf);
    }
    return 0;
}
