// launch_gap2.hip -- where the time BETWEEN two kernel nodes of a replayed hipGraph goes for a one-workgroup-per-CU persistent
// kernel (k_fused_decode_g<8,4> shows 2.0 us per launch outside its in-kernel stamps, the headline kernel 1.0).
// 256 workgroups x 512 threads; every workgroup stamps s_memrealtime at its first and last instruction and spins T us in
// between, so   period (HIP events / launches) - span (last end - first start)   is the cost of the node itself.
// One factor at a time: dynamic LDS bytes, VGPRs the kernel declares, dirty L2 lines left behind (plain stores, written back at
// the end of the kernel), size of the argument block, how ragged the workgroups' ends are.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

struct Args {
    unsigned long long* stamps;   // [64 launches][256][2]
    unsigned* counter;            // launch id (block 0 bumps it at its end)
    unsigned long long* dirty;
    int spin_ticks, dirty_words, ragged_ticks, pad0;
};
struct ArgsBig { Args a; char pad[680]; };

template <int NV>
__device__ __forceinline__ void body(const Args& a) {
    extern __shared__ char smem[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned id = __builtin_amdgcn_readfirstlane(*(volatile unsigned*)a.counter) & 63u;
    if (NV > 128) asm volatile("v_mov_b32 v250, 0" ::: "v250");
    else if (NV > 64) asm volatile("v_mov_b32 v120, 0" ::: "v120");
    const int spin = a.spin_ticks + (a.ragged_ticks ? (int)((blockIdx.x * 2654435761u >> 16) % (unsigned)a.ragged_ticks) : 0);
    if (spin) {
        while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < spin) __builtin_amdgcn_s_sleep(8);
    }
    if (a.dirty_words) {
        for (int i = threadIdx.x; i < a.dirty_words; i += 512) a.dirty[(size_t)blockIdx.x * 8192 + i] = i;
    }
    if (threadIdx.x == 0) {
        if (blockIdx.x == 0) {
            a.dirty[(size_t)256 * 8192] = smem[0];
            __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        a.stamps[((size_t)id * 256 + blockIdx.x) * 2] = t0;
        a.stamps[((size_t)id * 256 + blockIdx.x) * 2 + 1] = __builtin_amdgcn_s_memrealtime();
    }
}
template <int NV> __global__ __launch_bounds__(512, 2) void k_small(Args a) { body<NV>(a); }
template <int NV> __global__ __launch_bounds__(512, 2) void k_big(ArgsBig a) { body<NV>(a.a); }

template <class K, class A>
void run(const char* what, K kern, A args, int lds, Args* inner, hipStream_t st) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipMemsetAsync(inner->counter, 0, 4, st);
    hipGraph_t g;
    hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 32; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds, st, args);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < 10; ++i) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, st);
    const int reps = 40;
    for (int i = 0; i < reps; ++i) hipGraphLaunch(ge, st);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(64 * 256 * 2);
    hipMemcpy(h.data(), inner->stamps, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> spans, starts;
    for (int id = 0; id < 64; ++id) {
        unsigned long long lo = ~0ull, hi = 0, slo = 0;
        for (int b = 0; b < 256; ++b) {
            lo = std::min(lo, h[(id * 256 + b) * 2]);
            hi = std::max(hi, h[(id * 256 + b) * 2 + 1]);
            slo = std::max(slo, h[(id * 256 + b) * 2]);
        }
        if (hi > lo && hi - lo < 100000) { spans.push_back((hi - lo) / 100.0); starts.push_back((slo - lo) / 100.0); }
    }
    std::sort(spans.begin(), spans.end());
    std::sort(starts.begin(), starts.end());
    const double period = ms * 1e3 / (reps * 32), span = spans.empty() ? 0 : spans[spans.size() / 2];
    printf("%-78s period %6.2f us  span %6.2f  start spread %4.2f  outside the span %5.2f us\n", what, period, span,
           starts.empty() ? 0 : starts[starts.size() / 2], period - span);
    hipGraphExecDestroy(ge);
    hipGraphDestroy(g);
}

int main() {
    Args a{};
    hipMalloc(&a.stamps, 64 * 256 * 2 * 8);
    hipMalloc(&a.counter, 64);
    hipMalloc(&a.dirty, ((size_t)256 * 8192 + 8) * 8);
    hipStream_t st;
    hipStreamCreate(&st);
    auto S = [&](int spin_us, int dirty_words, int ragged_us) {
        Args x = a;
        x.spin_ticks = spin_us * 100;
        x.dirty_words = dirty_words;
        x.ragged_ticks = ragged_us * 100;
        return x;
    };
    auto B = [&](Args x) { ArgsBig b{}; b.a = x; return b; };
    Args x;
    x = S(20, 0, 0);
    run("base: 20 us spin, 84 KB LDS, few VGPRs, 48-B args", k_small<32>, x, 84 * 1024, &x, st);
    run("  ... 700-B argument block (FusedArgs)", k_big<32>, B(x), 84 * 1024, &x, st);
    run("  ... 120 VGPRs", k_big<100>, B(x), 84 * 1024, &x, st);
    run("  ... 250 VGPRs", k_big<200>, B(x), 84 * 1024, &x, st);
    run("  ... 250 VGPRs, 120 KB LDS", k_big<200>, B(x), 120 * 1024, &x, st);
    run("  ... 250 VGPRs, 140 KB LDS (k_fused_decode_g<8,4>)", k_big<200>, B(x), 140 * 1024, &x, st);
    run("  ... 250 VGPRs, 160 KB LDS", k_big<200>, B(x), 160 * 1024, &x, st);
    run("  ... 250 VGPRs, 8 KB LDS (2 workgroups could share a CU)", k_big<200>, B(x), 8 * 1024, &x, st);
    x = S(20, 264, 0);
    run("250 VGPRs, 140 KB LDS, 0.54 MB of dirty L2 lines left (the records of config 4)", k_big<200>, B(x), 140 * 1024, &x, st);
    x = S(20, 2048, 0);
    run("250 VGPRs, 140 KB LDS, 4 MB of dirty L2 lines left", k_big<200>, B(x), 140 * 1024, &x, st);
    x = S(20, 0, 3);
    run("250 VGPRs, 140 KB LDS, ends ragged over 3 us", k_big<200>, B(x), 140 * 1024, &x, st);
    x = S(0, 0, 0);
    run("empty kernel, 250 VGPRs, 140 KB LDS", k_big<200>, B(x), 140 * 1024, &x, st);
    run("empty kernel, few VGPRs, 84 KB LDS", k_big<32>, B(x), 84 * 1024, &x, st);
    return 0;
}
