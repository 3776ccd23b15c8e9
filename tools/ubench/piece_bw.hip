// piece_bw.hip -- phase 1 of the [in,out] kernel in isolation: 256 workgroups stream 384 KB each as strided pieces of
// 256 / 512 / 1024 B per 8-KB weight row (one / two / four heads of an XCD); workgroup duration by piece position.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef _Float16 h16;
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
#define G __attribute__((address_space(1)))
__device__ __forceinline__ h16x8 ld(const h16* p) { return __builtin_nontemporal_load((const G h16x8*)p); }
// PIECE = bytes per row a workgroup reads (256: 16 lanes/row, 512: 32 lanes/row, 1024: 64 lanes/row); every WG reads 384 KB
template <int PIECE>
__global__ __launch_bounds__(512, 2) void k(const h16* __restrict__ w, unsigned long long* st, float* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.x;
    constexpr int LPR = PIECE / 16, RPI = 64 / LPR;            // lanes per row, rows per instruction
    constexpr int NP = 8192 / PIECE;                           // pieces per 8-KB row
    const int piece = (b & 7) * (NP / 8) + ((b >> 3) % (NP / 8));   // XCD-local pieces
    const int ks = (b >> 3) / (NP / 8), nks = 32 / (NP / 8);   // K-slices per piece
    const int rows_per_wg = 12288 / nks;                       // rows of the [12288, 4096] matrix
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float acc = 0.f;
    const int r0 = ks * rows_per_wg + wave * (rows_per_wg / 8) + lane / LPR;
    for (int i = 0; i < rows_per_wg / 8; i += RPI * 16) {
        h16x8 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = ld(w + (size_t)(r0 + i + u * RPI) * 4096 + piece * (PIECE / 2) + (lane % LPR) * 8);
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += (float)v[u][0] + (float)v[u][7];
    }
    if (acc == 12345.678f) out[0] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { st[b * 2] = t0; st[b * 2 + 1] = __builtin_amdgcn_s_memrealtime(); }
}
template <int PIECE>
void run(const h16* w, unsigned long long* st, float* out) {
    constexpr int NP = 8192 / PIECE;
    std::vector<double> d[8], all;
    for (int rep = 0; rep < 40; ++rep) {
        hipLaunchKernelGGL((k<PIECE>), dim3(256), dim3(512), 0, 0, w + (size_t)(rep % 8) * 12288 * 4096, st, out);
        hipDeviceSynchronize();
        unsigned long long h[512];
        hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
        if (rep < 6) continue;
        unsigned long long a0 = ~0ull, a1 = 0;
        for (int b = 0; b < 256; ++b) {
            const int piece = (b & 7) * (NP / 8) + ((b >> 3) % (NP / 8));
            d[piece % (NP / 8 > 8 ? 8 : NP / 8)].push_back((h[b * 2 + 1] - h[b * 2]) / 100.0);
            a0 = std::min(a0, h[b * 2]); a1 = std::max(a1, h[b * 2 + 1]);
        }
        all.push_back((a1 - a0) / 100.0);
    }
    std::sort(all.begin(), all.end());
    printf("piece %4d B: span med %.2f us | WG duration median by piece-in-XCD:", PIECE, all[all.size() / 2]);
    for (int q = 0; q < NP / 8 && q < 8; ++q) { std::sort(d[q].begin(), d[q].end()); printf(" %.2f", d[q][d[q].size() / 2]); }
    printf("\n");
}
int main() {
    h16* w; float* out; unsigned long long* st;
    hipMalloc(&w, (size_t)9 * 12288 * 8192); hipMemset(w, 1, (size_t)9 * 12288 * 8192);
    hipMalloc(&out, 4); hipMalloc(&st, 512 * 8);
    run<256>(w, st, out); run<512>(w, st, out); run<1024>(w, st, out); run<256>(w, st, out);
    return 0;
}
