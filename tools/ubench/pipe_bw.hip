// pipe_bw.hip -- does the stream rate depend on the number of wavefronts per CU when the loads are software-
// pipelined (one 8-KB row always in flight per wavefront beside the one being consumed), as in the fused kernel?
// 201 MB of 8-KB rows, wavefront gw streams rows gw, gw + NW, ... ; device-stamped first-start -> last-end time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef _Float16 h16;
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
#define G __attribute__((address_space(1)))
__device__ __forceinline__ h16x8 ld(const h16* p) { return __builtin_nontemporal_load((const G h16x8*)p); }
template <int THREADS, int MINW, int DEPTH>
__global__ __launch_bounds__(THREADS, MINW) void k(const h16* __restrict__ w, int total_rows, unsigned long long* st, float* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int WPB = THREADS / 64;
    const int NW = gridDim.x * WPB, gw = blockIdx.x * WPB + wave;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float acc = 0.f;
    h16x8 buf[DEPTH][8];
    auto load = [&](h16x8 (&t)[8], int r) {
        const size_t row = r < total_rows ? r : total_rows - 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = ld(w + row * 4096 + (j * 64 + lane) * 8);
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load(buf[d], gw + d * NW);
    for (int r = gw; r < total_rows; r += DEPTH * NW) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += (float)buf[d][j][0] + (float)buf[d][j][7];
            load(buf[d], r + (d + DEPTH) * NW);
        }
    }
    if (acc == 12345.678f) out[0] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { st[blockIdx.x * 2] = t0; st[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime(); }
}
template <int THREADS, int MINW, int DEPTH>
void run(const char* name, int blocks, int total_rows, const h16* w, size_t bytes, unsigned long long* st, float* out) {
    std::vector<unsigned long long> h(blocks * 2);
    std::vector<double> walls;
    const size_t win = (size_t)total_rows * 8192;
    for (int rep = 0; rep < 25; ++rep) {
        hipLaunchKernelGGL((k<THREADS, MINW, DEPTH>), dim3(blocks), dim3(THREADS), 0, 0, w + (size_t)(rep % (int)(bytes / win)) * (win / 2), total_rows, st, out);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), st, blocks * 16, hipMemcpyDeviceToHost);
        if (rep < 5) continue;
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int b = 0; b < blocks; ++b) { t0 = std::min(t0, h[b * 2]); t1 = std::max(t1, h[b * 2 + 1]); }
        walls.push_back((t1 - t0) / 100.0);
    }
    std::sort(walls.begin(), walls.end());
    const double med = walls[walls.size() / 2];
    printf("%-40s blocks=%5d wall med %.2f us (min %.2f) -> %.0f GB/s\n", name, blocks, med, walls.front(), win / med / 1e3);
}
int main() {
    const size_t bytes = (size_t)2 << 30;
    h16* w; float* out; unsigned long long* st;
    hipMalloc(&w, bytes + (256 << 20)); hipMalloc(&out, 4); hipMalloc(&st, 8192 * 16); hipMemset(w, 1, bytes);
    const int rows = 24576;   // 201 MB
    run<512, 2, 2>("8 waves/CU, 2 rows in flight", 256, rows, w, bytes, st, out);
    run<512, 2, 4>("8 waves/CU, 4 rows in flight", 256, rows, w, bytes, st, out);
    run<1024, 4, 2>("16 waves/CU (1 WG), 2 rows in flight", 256, rows, w, bytes, st, out);
    run<256, 4, 2>("16 waves/CU (4 WGs), 2 rows in flight", 1024, rows, w, bytes, st, out);
    run<1024, 4, 1>("16 waves/CU (1 WG), 1 row in flight", 256, rows, w, bytes, st, out);
    run<512, 4, 2>("16 waves/CU (2 WGs), 2 rows in flight", 512, rows, w, bytes, st, out);
    return 0;
}
