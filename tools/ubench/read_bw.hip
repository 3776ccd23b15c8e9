// read_bw.hip -- what a pure READ stream reaches on this chip, in the access shapes the decode
// kernels use.  hipcc --offload-arch=gfx950 -O3 read_bw.hip -o read_bw && ./read_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h16;
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
#define G __attribute__((address_space(1)))
template <bool NT> __device__ __forceinline__ h16x8 ld(const h16* p) {
    if (NT) return __builtin_nontemporal_load((const G h16x8*)p);
    return *(const G h16x8*)p;
}
// each wave streams `rows` consecutive 8-KB rows starting at its own offset; R rows in flight x 2
template <bool NT, int R>
__global__ __launch_bounds__(512, 2) void k_rows(const h16* __restrict__ w, size_t rows_per_wave, float* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t w0 = ((size_t)blockIdx.x * (blockDim.x >> 6) + wave) * rows_per_wave;
    float acc = 0.f;
    for (size_t r = 0; r < rows_per_wave; r += R) {
        h16x8 v[R][8];
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] = ld<NT>(w + (w0 + r + i) * 4096 + (j * 64 + lane) * 8);
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += (float)v[i][j][0] + (float)v[i][j][7];
    }
    if (acc == 12345.678f) out[0] = acc;
}
// grid-stride: consecutive waves read consecutive KBs
template <bool NT, int U>
__global__ __launch_bounds__(256) void k_stride(const h16* __restrict__ w, size_t n_vec, float* out) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t step = (size_t)gridDim.x * 256;
    float acc = 0.f;
    for (; i + (U - 1) * step < n_vec; i += U * step) {
        h16x8 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld<NT>(w + (i + u * step) * 8);
#pragma unroll
        for (int u = 0; u < U; ++u) acc += (float)v[u][0] + (float)v[u][7];
    }
    if (acc == 12345.678f) out[0] = acc;
}
template <class F> float time_ms(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main() {
    const size_t bytes = (size_t)3 << 30;    // 3 GiB, cycled: every launch reads a fresh 100-MB window
    h16* w; float* out; hipMalloc(&w, bytes + (64 << 20)); hipMalloc(&out, 4); hipMemset(w, 1, bytes);
    const size_t win = (size_t)12288 * 8192;  // 100.66 MB = Wqkv of one layer
    const int nwin = bytes / win;
    int k = 0;
    auto rows = [&](auto kern, int blocks, int threads, const char* name) {
        size_t rpw = 12288 / ((size_t)blocks * (threads / 64));
        float ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, w + (size_t)(k++ % nwin) * (win / 2), rpw, out); }, 200);
        printf("%-44s %7.2f us  %7.1f GB/s\n", name, ms * 1e3, win / ms / 1e6);
    };
    auto stride = [&](auto kern, int blocks, const char* name) {
        float ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, w + (size_t)(k++ % nwin) * (win / 2), win / 16, out); }, 200);
        printf("%-44s %7.2f us  %7.1f GB/s\n", name, ms * 1e3, win / ms / 1e6);
    };
    printf("one launch = 100.66 MB read (distinct window each launch), time incl. launch ramp\n");
    rows(k_rows<true, 2>, 256, 512, "rows nt  R=2 256x512 (6 rows/wave)");
    rows(k_rows<false, 2>, 256, 512, "rows def R=2 256x512");
    rows(k_rows<true, 3>, 256, 512, "rows nt  R=3 256x512");
    rows(k_rows<true, 2>, 512, 256, "rows nt  R=2 512x256");
    rows(k_rows<true, 1>, 512, 512, "rows nt  R=1 512x512 (3 rows/wave)");
    rows(k_rows<true, 2>, 1024, 256, "rows nt  R=2 1024x256 (3 rows/wave)");
    stride(k_stride<true, 8>, 1024, "grid-stride nt  U=8 1024 blocks");
    stride(k_stride<false, 8>, 1024, "grid-stride def U=8 1024 blocks");
    stride(k_stride<true, 8>, 2048, "grid-stride nt  U=8 2048 blocks");
    stride(k_stride<true, 16>, 512, "grid-stride nt  U=16 512 blocks");
    stride(k_stride<true, 4>, 4096, "grid-stride nt  U=4 4096 blocks");
    // long stream: 1.2 GB in one launch (steady state, ramp amortised)
    {
        const size_t big = win * 12;
        float ms = time_ms([&] { hipLaunchKernelGGL((k_stride<true, 8>), dim3(2048), dim3(256), 0, 0, w, big / 16, out); }, 20);
        printf("%-44s %7.2f us  %7.1f GB/s\n", "grid-stride nt U=8 2048 blocks, 1.2 GB", ms * 1e3, big / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL((k_stride<false, 8>), dim3(2048), dim3(256), 0, 0, w, big / 16, out); }, 20);
        printf("%-44s %7.2f us  %7.1f GB/s\n", "grid-stride def U=8 2048 blocks, 1.2 GB", ms * 1e3, big / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL((k_rows<true, 2>), dim3(256), dim3(512), 0, 0, w, (size_t)72, out); }, 20);
        printf("%-44s %7.2f us  %7.1f GB/s\n", "rows nt R=2 256x512, 72 rows/wave (1.2 GB)", ms * 1e3, big / ms / 1e6);
    }
    return 0;
}
