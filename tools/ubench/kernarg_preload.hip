// kernarg_preload.hip -- what the first dependent load of a kernel costs when its pointer comes from the kernarg segment through
// s_load (cold at every kernel start: the scalar cache is invalidated) and when the CP preloads it into SGPRs (gfx950, code object
// v5+: -mllvm -amdgpu-kernarg-preload-count=N; the Makefile builds this file both ways).  256 workgroups x 512 threads, 32 launches
// per graph replay; every workgroup stamps s_memrealtime at its first instruction, when its first load has been ISSUED (the pointer
// is there) and when the data is back.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
struct Tail { const float* q[8]; char pad[600]; };
__global__ __launch_bounds__(512, 2) void k(const float* a, unsigned long long* st, Tail t) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const float v = __builtin_nontemporal_load(a + (size_t)blockIdx.x * 4096 + threadIdx.x);
    asm volatile("" ::"v"(v));      // (issued: the address is known)
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    float w = v * 2.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(w));
    const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { st[blockIdx.x * 4] = t0; st[blockIdx.x * 4 + 1] = t1; st[blockIdx.x * 4 + 2] = t2; }
    if (w == 12345.f) st[1 << 20] = (unsigned long long)t.q[1];
}
int main() {
    float* a; unsigned long long* st;
    hipMalloc(&a, 256 * 4096 * 4 * 40); hipMalloc(&st, ((1 << 20) + 8) * 8);
    hipMemset(a, 0, 256 * 4096 * 4 * 40);
    Tail t{};
    hipStream_t s; hipStreamCreate(&s);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 32; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, s, a + (size_t)i * 256 * 4096, st, t);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    std::vector<double> issue, data;
    for (int rep = 0; rep < 30; ++rep) {
        hipGraphLaunch(ge, s);
        hipStreamSynchronize(s);
        unsigned long long h[1024];
        hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
        if (rep < 5) continue;
        for (int b = 0; b < 256; ++b) { issue.push_back((h[b * 4 + 1] - h[b * 4]) / 100.0); data.push_back((h[b * 4 + 2] - h[b * 4]) / 100.0); }
    }
    std::sort(issue.begin(), issue.end()); std::sort(data.begin(), data.end());
    auto q = [](std::vector<double>& v, double p) { return v[(size_t)(p * (v.size() - 1))]; };
#ifdef PRELOAD
    printf("kernarg PRELOADED into SGPRs: ");
#else
    printf("kernarg through s_load:       ");
#endif
    printf("first load issued %.2f / %.2f / %.2f us after the workgroup's start (p10 / median / p90), data back %.2f / %.2f / %.2f\n",
           q(issue, .1), q(issue, .5), q(issue, .9), q(data, .1), q(data, .5), q(data, .9));
    return 0;
}
