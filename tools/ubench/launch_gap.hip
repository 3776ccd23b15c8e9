// launch_gap.hip -- what one kernel node of a hipGraph costs when the kernel itself does (almost) nothing: the floor under
// "us per call" of a persistent one-launch-per-layer kernel.  256 workgroups x 512 threads, 84 KB dynamic LDS (one per CU),
// a ~700-byte argument block like FusedArgs; variants: empty / every workgroup spins T us / dirty lines left in L2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct Args { unsigned long long* p; int spin_ticks; int dirty; char pad[680]; };
__global__ __launch_bounds__(512, 2) void k(Args a) {
    extern __shared__ char smem[];
    if (a.spin_ticks) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < a.spin_ticks) __builtin_amdgcn_s_sleep(8);
    }
    if (a.dirty) {   // plain (write-back) stores: dirty lines in this XCD's L2, written back at the end of the kernel
        for (int i = threadIdx.x; i < a.dirty; i += 512) a.p[(size_t)blockIdx.x * 4096 + i] = i;
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) a.p[1 << 22] = smem[0];
}
int main() {
    unsigned long long* d;
    hipMalloc(&d, (size_t)(4 << 20) * 8 + 64);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 84 * 1024);
    hipStream_t st;
    hipStreamCreate(&st);
    const int cfg[][2] = {{0, 0}, {500, 0}, {1000, 0}, {0, 512}, {0, 4096}, {1000, 512}};
    for (auto& c : cfg) {
        Args a{d, c[0], c[1], {}};
        hipGraph_t g;
        hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        for (int i = 0; i < 32; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(512), 84 * 1024, st, a);
        hipStreamEndCapture(st, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        for (int i = 0; i < 3; ++i) hipGraphLaunch(ge, st);
        hipStreamSynchronize(st);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0, st);
        const int reps = 50;
        for (int i = 0; i < reps; ++i) hipGraphLaunch(ge, st);
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("spin %4.1f us, dirty %4d x 8 B per workgroup: %.2f us per kernel node (overhead %.2f us)\n", c[0] / 100.0, c[1],
               ms * 1e3 / (reps * 32), ms * 1e3 / (reps * 32) - c[0] / 100.0);
    }
    return 0;
}
