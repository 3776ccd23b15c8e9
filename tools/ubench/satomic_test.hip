// satomic_test.hip -- do scalar memory atomics (s_atomic_add ... glc) work on gfx950 as XCD-local ticket counters?
// Every wavefront takes N tickets from the counter of the XCD it runs on; tickets of a pool must be 0..total-1, each once.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__device__ __forceinline__ unsigned xcc_id() { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); return x & 7u; }
__global__ __launch_bounds__(512) void k(unsigned* ctr, unsigned* tickets, unsigned* where, int n, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned xcd = xcc_id();
    unsigned* c = ctr + xcd * 64;
    const int gw = blockIdx.x * 8 + wave;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        unsigned v = 1;
        asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(c) : "memory");
        if (lane == 0) tickets[(size_t)gw * n + i] = v;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { where[gw] = xcd; cyc[gw] = t1 - t0; }
}
int main() {
    const int blocks = 256, n = 64, nw = blocks * 8;
    unsigned *ctr, *tk, *wh; unsigned long long* cyc;
    hipMalloc(&ctr, 8 * 256); hipMalloc(&tk, nw * n * 4); hipMalloc(&wh, nw * 4); hipMalloc(&cyc, nw * 8);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(ctr, 0, 8 * 256); hipMemset(tk, 0xff, nw * n * 4);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, ctr, tk, wh, n, cyc);
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { printf("FAILED: %s\n", hipGetErrorString(e)); return 1; }
        std::vector<unsigned> t(nw * n), w(nw), c(16 * 8); std::vector<unsigned long long> cy(nw);
        hipMemcpy(t.data(), tk, nw * n * 4, hipMemcpyDeviceToHost); hipMemcpy(w.data(), wh, nw * 4, hipMemcpyDeviceToHost);
        hipMemcpy(cy.data(), cyc, nw * 8, hipMemcpyDeviceToHost);
        hipMemcpy(c.data(), ctr, 8 * 64 * 4 > 16 * 8 * 4 ? 16 * 8 * 4 : 8 * 64 * 4, hipMemcpyDeviceToHost);
        bool ok = true;
        for (int x = 0; x < 8; ++x) {
            std::vector<unsigned> v;
            for (int g = 0; g < nw; ++g) if (w[g] == (unsigned)x) for (int i = 0; i < n; ++i) v.push_back(t[(size_t)g * n + i]);
            std::sort(v.begin(), v.end());
            bool good = true;
            for (size_t i = 0; i < v.size(); ++i) good &= v[i] == i;
            printf("xcd %d: %zu tickets, %s (first %u last %u)\n", x, v.size(), good ? "unique 0..n-1" : "BROKEN", v.empty() ? 0 : v.front(), v.empty() ? 0 : v.back());
            ok &= good;
        }
        double avg = 0; for (auto q : cy) avg += q; avg /= nw;
        printf("rep %d: %s; avg cycles per ticket (2048 waves contending) %.0f\n", rep, ok ? "OK" : "FAIL", avg / n);
    }
    return 0;
}
