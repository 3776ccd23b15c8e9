// hbm_map.hip -- which addresses share a path to memory on MI355X, measured: two >= 1 us effects in the
// persistent kernels are address-map effects -- the "dense deal" of phase 1's rows (profiles/r05_experiments.md section 12) and the
// K/V pieces at (address >> 8) & 3 == 1 that stream ~17 % slower (cf_fused_kernel_g.h) -- and were tuned around blind.
//
//   hbm_map stride                 256 workgroups stream 256-B pieces that are R bytes apart, R = 256 B .. 64 KB: the classic
//                                  interleave probe -- the rate collapses when R is a multiple of (granule x channels)
//   hbm_map class R                256-B pieces R bytes apart at offset c x 256, c = 0 .. R/256 - 1 (R = 8192: the K/V pieces of
//                                  head c of a [S, 4096] cache; R = 2048: kv head c of Llama-3-8B): one class at a time
//   hbm_map classes R              ... all classes at once, workgroup b on class b % (R/256): duration by class
//   hbm_map deal runs|dense|...    phase 1's weight stream: 6144 row pairs of 16 KB dealt to 256 x 8 wavefronts x 3 slots
//   hbm_map one <mode...> [reps]   a single configuration, for a rocprofv3 --pmc pass (kernel names say the configuration)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
typedef _Float16 h16;
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
#define GAS __attribute__((address_space(1)))
__device__ __forceinline__ h16x8 ld(const char* p) { return __builtin_nontemporal_load((const GAS h16x8*)p); }

struct Stamps { unsigned long long* st; float* sink; };

// ---- 256-B pieces R bytes apart ---------------------------------------------------------------------------------------------
// wavefront (b, w), iteration it, load u, lane group q = lane / 16: piece ((b * 8 + w) * ITERS + it) * 64 + 4 u + q
template <int ITERS>
__global__ __launch_bounds__(512, 2) void k_pieces(const char* base, size_t R, int cls, int ncls_mix, Stamps s) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const int c = ncls_mix ? b % ncls_mix : cls;
    float acc = 0.f;
    for (int it = 0; it < ITERS; ++it) {
        h16x8 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const size_t piece = ((size_t)(b * 8 + wave) * ITERS + it) * 64 + 4 * u + (lane >> 4);
            v[u] = ld(base + piece * R + (size_t)c * 256 + (lane & 15) * 16);
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += (float)v[u][0] + (float)v[u][7];
    }
    if (acc == 12345.678f) s.sink[0] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { s.st[b * 2] = t0; s.st[b * 2 + 1] = __builtin_amdgcn_s_memrealtime(); }
}

// ---- phase 1's weight stream: wavefront (b, w) streams SLOTS row pairs of 16 KB ------------------------------------------------
// deal 0 "runs": pair 24 b + w + 8 s (256 separate 384-KB runs); 1 "dense": pair 2048 s + 8 b + w (one 32-MB region at a time);
// 2 "runs8": pair 24 b + 3 w + s (a wavefront's slots adjacent); 3 "xcd": pair 2048 s + 256 (b % 8) + 8 (b / 8) + w (an XCD's
// 32 workgroups read one 4-MB region at a time)
template <int DEAL>
__global__ __launch_bounds__(512, 2) void k_deal(const char* base, Stamps s) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    auto pair = [&](int sl) -> size_t {
        if (DEAL == 0) return 24 * b + wave + 8 * sl;
        if (DEAL == 1) return 2048 * sl + 8 * b + wave;
        if (DEAL == 2) return 24 * b + 3 * wave + sl;
        return 2048 * sl + 256 * (b & 7) + 8 * (b >> 3) + wave;
    };
    float acc = 0.f;
    h16x8 x[16], y[16];
    const char* p0 = base + pair(0) * 16384 + lane * 16;
#pragma unroll
    for (int u = 0; u < 16; ++u) x[u] = ld(p0 + u * 1024);
    const char* p1 = base + pair(1) * 16384 + lane * 16;
#pragma unroll
    for (int u = 0; u < 16; ++u) y[u] = ld(p1 + u * 1024);
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += (float)x[u][0] + (float)x[u][7];
    const char* p2 = base + pair(2) * 16384 + lane * 16;
#pragma unroll
    for (int u = 0; u < 16; ++u) x[u] = ld(p2 + u * 1024);
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += (float)y[u][0] + (float)y[u][7];
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += (float)x[u][0] + (float)x[u][7];
    if (acc == 12345.678f) s.sink[0] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { s.st[b * 2] = t0; s.st[b * 2 + 1] = __builtin_amdgcn_s_memrealtime(); }
}

struct Result { double span_us, gbs, wg_med, wg_p10, wg_p90; std::vector<double> by_class; };

template <class F>
Result measure(F launch, double bytes, int reps, unsigned long long* d_st, int ncls) {
    std::vector<double> spans, wg;
    std::vector<std::vector<double>> cls(ncls > 0 ? ncls : 1);
    for (int rep = 0; rep < reps + 4; ++rep) {
        launch(rep);
        hipDeviceSynchronize();
        unsigned long long h[512];
        hipMemcpy(h, d_st, sizeof(h), hipMemcpyDeviceToHost);
        if (rep < 4) continue;
        unsigned long long lo = ~0ull, hi = 0;
        for (int b = 0; b < 256; ++b) {
            lo = std::min(lo, h[b * 2]);
            hi = std::max(hi, h[b * 2 + 1]);
            wg.push_back((h[b * 2 + 1] - h[b * 2]) / 100.0);
            if (ncls > 0) cls[b % ncls].push_back((h[b * 2 + 1] - h[b * 2]) / 100.0);
        }
        spans.push_back((hi - lo) / 100.0);
    }
    std::sort(spans.begin(), spans.end());
    std::sort(wg.begin(), wg.end());
    Result r;
    r.span_us = spans[spans.size() / 2];
    r.gbs = bytes / r.span_us / 1e3;
    r.wg_med = wg[wg.size() / 2];
    r.wg_p10 = wg[wg.size() / 10];
    r.wg_p90 = wg[wg.size() * 9 / 10];
    if (ncls > 0)
        for (auto& v : cls) {
            std::sort(v.begin(), v.end());
            r.by_class.push_back(v.empty() ? 0 : v[v.size() / 2]);
        }
    return r;
}

int main(int argc, char** argv) {
    const std::string mode = argc > 1 ? argv[1] : "stride";
    constexpr int ITERS = 4;                                   // 256 x 8 x 4 x 64 pieces = 524288 pieces = 128 MiB per launch
    const size_t pieces = (size_t)256 * 8 * ITERS * 64;
    const size_t pool_bytes = (size_t)36 << 30;
    char* pool;
    if (hipMalloc(&pool, pool_bytes) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMemset(pool, 1, pool_bytes);
    Stamps s;
    hipMalloc(&s.st, 512 * 8);
    hipMalloc(&s.sink, 4);
    hipDeviceSynchronize();
    const double piece_bytes = (double)pieces * 256;
    auto run_pieces = [&](size_t R, int cls, int mix, int reps, bool rotate_cls = false) {
        const size_t span = pieces * R;
        const int nbase = (int)std::max<size_t>(1, std::min<size_t>(8, pool_bytes / span));
        const int ncls = (int)(R / 256);
        // every launch reads lines the 256-MiB Infinity Cache cannot still hold: another base, or (few bases) another class
        return measure([&](int rep) {
            const int c = rotate_cls ? (cls + rep / nbase) % ncls : cls;
            hipLaunchKernelGGL(k_pieces<ITERS>, dim3(256), dim3(512), 0, 0, pool + (size_t)(rep % nbase) * span, R, c, mix, s); },
                       piece_bytes, reps, s.st, mix);
    };
    auto run_deal = [&](int deal, int reps) {
        const size_t span = (size_t)6144 * 16384;
        auto L = [&](int rep) {
            const char* bp = pool + (size_t)(rep % 16) * span;
            if (deal == 0) hipLaunchKernelGGL(k_deal<0>, dim3(256), dim3(512), 0, 0, bp, s);
            else if (deal == 1) hipLaunchKernelGGL(k_deal<1>, dim3(256), dim3(512), 0, 0, bp, s);
            else if (deal == 2) hipLaunchKernelGGL(k_deal<2>, dim3(256), dim3(512), 0, 0, bp, s);
            else hipLaunchKernelGGL(k_deal<3>, dim3(256), dim3(512), 0, 0, bp, s);
        };
        return measure(L, (double)span, reps, s.st, 8);
    };
    const char* deal_names[] = {"runs (24 b + w + 8 s)", "dense (2048 s + 8 b + w)", "runs8 (24 b + 3 w + s)", "xcd (2048 s + 256 (b % 8) + 8 (b / 8) + w)"};
    if (mode == "stride") {
        printf("256-B pieces R bytes apart, 128 MiB of pieces per launch, 256 workgroups x 8 wavefronts x 64 pieces in flight\n");
        printf("%10s %10s %10s   workgroup duration p10 / median / p90 (us)\n", "R", "span us", "GB/s");
        for (size_t R : {256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536}) {
            const Result r = run_pieces(R, 0, 0, 24, true);
            printf("%10zu %10.2f %10.0f   %.2f / %.2f / %.2f\n", R, r.span_us, r.gbs, r.wg_p10, r.wg_med, r.wg_p90);
        }
    } else if (mode == "class" || mode == "classes") {
        const size_t R = argc > 2 ? strtoul(argv[2], nullptr, 0) : 8192;
        const int ncls = (int)(R / 256);
        if (mode == "class") {
            printf("256-B pieces %zu bytes apart, ONE class (offset c x 256) per launch\n%6s %10s %10s\n", R, "c", "span us", "GB/s");
            for (int c = 0; c < ncls; ++c) {
                const Result r = run_pieces(R, c, 0, 16);
                printf("%6d %10.2f %10.0f\n", c, r.span_us, r.gbs);
            }
        } else {
            const Result r = run_pieces(R, 0, ncls, 40);
            printf("256-B pieces %zu bytes apart, all %d classes at once (workgroup b: class b %% %d): span %.2f us, %.0f GB/s\n  median workgroup "
                   "duration by class:", R, ncls, ncls, r.span_us, r.gbs);
            for (int c = 0; c < ncls; ++c) printf(" %.2f", r.by_class[c]);
            printf("\n");
        }
    } else if (mode == "deal") {
        printf("phase 1's weight stream, 6144 row pairs of 16 KB (100.7 MB), 3 slots per wavefront, 2 in flight\n");
        for (int round = 0; round < 3; ++round)
            for (int d = 0; d < 4; ++d) {
                const Result r = run_deal(d, 48);
                printf("round %d  %-46s span %6.2f us  %5.0f GB/s  workgroup p10 / med / p90 %.2f / %.2f / %.2f   by XCD (b %% 8):", round, deal_names[d],
                       r.span_us, r.gbs, r.wg_p10, r.wg_med, r.wg_p90);
                for (double v : r.by_class) printf(" %.2f", v);
                printf("\n");
            }
    } else if (mode == "one") {      // hbm_map one deal <d> | one class <R> <c> | one classes <R>    (for counter passes)
        const std::string what = argc > 2 ? argv[2] : "deal";
        const int reps = 12;
        if (what == "deal") {
            const int d = argc > 3 ? atoi(argv[3]) : 0;
            const Result r = run_deal(d, reps);
            printf("deal %s: %.2f us %.0f GB/s\n", deal_names[d], r.span_us, r.gbs);
        } else if (what == "class") {
            const size_t R = strtoul(argv[3], nullptr, 0);
            const int c = atoi(argv[4]);
            const Result r = run_pieces(R, c, 0, reps);
            printf("class R %zu c %d: %.2f us %.0f GB/s\n", R, c, r.span_us, r.gbs);
        } else {
            const size_t R = strtoul(argv[3], nullptr, 0);
            const Result r = run_pieces(R, 0, (int)(R / 256), reps);
            printf("classes R %zu: %.2f us %.0f GB/s\n", R, r.span_us, r.gbs);
        }
    }
    return 0;
}
