// sweep_bw.hip -- single-launch wall time (first WG start -> last WG end, device stamps) of a pure
// 8-KB-row streaming read, over launch geometries and row->wave assignments.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef _Float16 h16;
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
#define G __attribute__((address_space(1)))
__device__ __forceinline__ h16x8 ld(const h16* p) { return __builtin_nontemporal_load((const G h16x8*)p); }
// total_rows rows of 8 KB; wave gw of NW handles rows: inter ? gw + k*NW : gw*rpw + k
template <int R, int THREADS, int MINW>
__global__ __launch_bounds__(THREADS, MINW) void k(const h16* __restrict__ w, int total_rows, int inter, unsigned long long* st, float* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int WPB = THREADS / 64;
    const int NW = gridDim.x * WPB, gw = blockIdx.x * WPB + wave;
    const int rpw = total_rows / NW;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float acc = 0.f;
    for (int r = 0; r < rpw; r += R) {
        h16x8 v[R][8];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const size_t row = inter ? (size_t)(r + i) * NW + gw : (size_t)gw * rpw + r + i;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] = ld(w + row * 4096 + (j * 64 + lane) * 8);
        }
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += (float)v[i][j][0] + (float)v[i][j][7];
    }
    if (acc == 12345.678f) out[0] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { st[blockIdx.x * 2] = t0; st[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime(); }
}
template <int R, int THREADS, int MINW>
void run(const char* name, int blocks, int total_rows, int inter, const h16* w, size_t bytes, unsigned long long* st, float* out) {
    std::vector<unsigned long long> h(blocks * 2);
    std::vector<double> walls;
    const size_t win = (size_t)total_rows * 8192;
    for (int rep = 0; rep < 25; ++rep) {
        hipLaunchKernelGGL((k<R, THREADS, MINW>), dim3(blocks), dim3(THREADS), 0, 0, w + (size_t)(rep % (int)(bytes / win)) * (win / 2), total_rows, inter, st, out);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), st, blocks * 16, hipMemcpyDeviceToHost);
        if (rep < 5) continue;
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int b = 0; b < blocks; ++b) { t0 = std::min(t0, h[b * 2]); t1 = std::max(t1, h[b * 2 + 1]); }
        walls.push_back((t1 - t0) / 100.0);
    }
    std::sort(walls.begin(), walls.end());
    double med = walls[walls.size() / 2];
    printf("%-34s blocks=%5d rows=%5d %-6s wall med %.2f us (min %.2f max %.2f) -> %.0f GB/s\n", name, blocks, total_rows, inter ? "inter" : "contig",
           med, walls.front(), walls.back(), win / med / 1e3);
}
int main() {
    const size_t bytes = (size_t)2 << 30;
    h16* w; float* out; unsigned long long* st;
    hipMalloc(&w, bytes + (256 << 20)); hipMalloc(&out, 4); hipMalloc(&st, 8192 * 16); hipMemset(w, 1, bytes);
    for (int rows : {12288, 24576}) {
        for (int inter : {0, 1}) {
            run<2, 512, 2>("R=2 512thr (2w/simd)", 256, rows, inter, w, bytes, st, out);
            run<2, 256, 2>("R=2 256thr", 512, rows, inter, w, bytes, st, out);
            run<1, 256, 4>("R=1 256thr (<=128 vgpr)", 1024, rows, inter, w, bytes, st, out);
            run<1, 256, 8>("R=1 256thr (<=64 vgpr)", 2048, rows, inter, w, bytes, st, out);
            run<1, 256, 8>("R=1 256thr (<=64 vgpr)", 3072, rows, inter, w, bytes, st, out);
            run<1, 128, 8>("R=1 128thr (<=64 vgpr)", 6144, rows, inter, w, bytes, st, out);
            run<2, 256, 4>("R=2 256thr (<=128 vgpr)", 1536, rows, inter, w, bytes, st, out);
        }
    }
    return 0;
}
