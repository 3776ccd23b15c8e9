// xcd_bw.hip -- per-XCD streaming rate: 256 workgroups x 512 threads, each streams `rows` 8-KB rows;
// every workgroup stamps start/end (100 MHz) and its HW XCC id.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef _Float16 h16;
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
#define G __attribute__((address_space(1)))
__device__ __forceinline__ h16x8 ld(const h16* p) { return __builtin_nontemporal_load((const G h16x8*)p); }
__global__ __launch_bounds__(512, 2) void k(const h16* __restrict__ w, int rows_per_wave, unsigned long long* st, float* out, int mode) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    // mode 0: workgroup b streams a contiguous chunk; mode 1: chunks interleaved so XCD (b%8) owns 1/8 stripes
    size_t w0 = ((size_t)b * 8 + wave) * rows_per_wave;
    const bool inter = mode == 3;
    float acc = 0.f;
    for (int r = 0; r < rows_per_wave; r += 2) {
        h16x8 v[2][8];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int jr = j;
                if (mode == 1) jr = (j + b + wave) & 7;          // rotate the 1-KB pieces of a row per wavefront
                if (mode == 2) jr = (j + (b >> 3)) & 7;           // rotate per workgroup index within XCD
                const size_t row = inter ? (size_t)(r + i) * 2048 + (size_t)b * 8 + wave : w0 + r + i;
                v[i][j] = ld(w + row * 4096 + (jr * 64 + lane) * 8);
            }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += (float)v[i][j][0] + (float)v[i][j][7];
    }
    if (acc == 12345.678f) out[0] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        st[b * 4 + 0] = t0;
        st[b * 4 + 1] = __builtin_amdgcn_s_memrealtime();
        st[b * 4 + 2] = xcc & 0xf;
    }
}
int main() {
    const size_t bytes = (size_t)2 << 30;
    h16* w; float* out; unsigned long long* st;
    hipMalloc(&w, bytes + (64 << 20)); hipMalloc(&out, 4); hipMalloc(&st, 256 * 4 * 8); hipMemset(w, 1, bytes);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    for (int mode : {0, 1, 2, 3})
    for (int rows : {6, 12}) {
        std::vector<double> dur[8], start[8];
        int mism = 0;
        const size_t win = (size_t)256 * 8 * rows * 8192;
        for (int rep = 0; rep < 30; ++rep) {
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 96 * 1024, 0, w + (size_t)(rep % (int)(bytes / win)) * (win / 2), rows, st, out, mode);
            hipDeviceSynchronize();
            unsigned long long h[256 * 4];
            hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
            if (rep < 5) continue;
            unsigned long long tmin = ~0ull;
            for (int b = 0; b < 256; ++b) tmin = std::min(tmin, h[b * 4]);
            for (int b = 0; b < 256; ++b) {
                int x = (int)h[b * 4 + 2];
                if (x != b % 8) ++mism;
                dur[x & 7].push_back((h[b * 4 + 1] - h[b * 4]) / 100.0);
                start[x & 7].push_back((h[b * 4] - tmin) / 100.0);
            }
        }
        printf("   per-XCD median WG duration:");
        for (int x = 0; x < 8; ++x) { std::vector<double> v = dur[x]; std::sort(v.begin(), v.end()); printf(" %.2f", v[v.size() / 2]); }
        printf("\n");
        std::vector<double> all, ends;
        for (int x = 0; x < 8; ++x) for (size_t i = 0; i < dur[x].size(); ++i) { all.push_back(dur[x][i]); ends.push_back(dur[x][i] + start[x][i]); }
        std::sort(all.begin(), all.end()); std::sort(ends.begin(), ends.end());
        size_t n = all.size();
        printf("mode %d rows/wave=%2d (%.0f MB): WG duration min %.2f p10 %.2f med %.2f p90 %.2f max %.2f | end-time med %.2f p99 %.2f max %.2f -> %.0f GB/s at max\n", mode, rows, win / 1e6,
               all[0], all[n / 10], all[n / 2], all[n * 9 / 10], all[n - 1], ends[n / 2], ends[n * 99 / 100], ends[n - 1], win / ends[n - 1] / 1e3);
    }
    return 0;
}
