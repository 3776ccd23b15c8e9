// kv_bw.hip -- the K/V read pattern of phase 2 in isolation: 256 workgroups, workgroup (head c, split j)
// reads the 256-B chunk c of 512 token rows (8 KB apart) from a K and a V pool.  Reports the median
// workgroup duration by chunk index mod 4 -- is the (c % 4 == 1) slowness seen in the fused kernel's
// timeline a property of the address pattern?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef _Float16 h16;
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
#define G __attribute__((address_space(1)))
__device__ __forceinline__ h16x8 ld(const h16* p) { return __builtin_nontemporal_load((const G h16x8*)p); }
__global__ __launch_bounds__(512, 2) void k(const h16* __restrict__ kp, const h16* __restrict__ vp, unsigned long long* st, float* out,
                                             int map, int paged, int chunk_bytes_shift) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    int c = (b & 7) * 4 + (b >> 6);
    if (map == 1) c ^= 1;
    if (map == 2) c = b >> 3;
    const int j = (b >> 3) & 7, gid = wave * 4 + (lane >> 4), l16 = lane & 15;
    h16x8 kk[16], vv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        int tok = j * 512 + u * 32 + gid;
        if (map == 3) { c = (b & 7) * 4 + (wave & 3); tok = (b >> 3) * 128 + (wave >> 2) * 64 + u * 4 + (lane >> 4); }   // every workgroup reads all 4 chunk classes
        if (map == 4) { c = ((b >> 3) & 7) * 4 + (wave & 3); tok = ((b & 7) * 4 + (b >> 6)) * 128 + (wave >> 2) * 64 + u * 4 + (lane >> 4); }
        size_t row = tok;
        if (paged) { unsigned pg = tok >> 4; pg = (pg * 2654435761u) >> 20 & 255u; row = ((size_t)pg << 4) | (tok & 15); }   // 256 pages of 16
        const size_t off = row * 4096 + c * 128 + l16 * 8;
        kk[u] = ld(kp + off);
        vv[u] = ld(vp + off);
    }
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += (float)kk[u][0] + (float)vv[u][7];
    if (acc == 12345.678f) out[0] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { st[b * 2] = t0; st[b * 2 + 1] = __builtin_amdgcn_s_memrealtime(); }
}
int main() {
    const size_t layer = (size_t)4096 * 8192;   // one K (or V) cache of 4096 tokens
    const int L = 24;
    h16 *kp, *vp; float* out; unsigned long long* st;
    hipMalloc(&kp, layer * L); hipMalloc(&vp, layer * L); hipMalloc(&out, 4); hipMalloc(&st, 256 * 2 * 8);
    hipMemset(kp, 1, layer * L); hipMemset(vp, 1, layer * L);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    for (int paged : {0, 1})
    for (int map : {0, 3, 4}) {
        std::vector<double> d[4], all, dc[32];
        for (int rep = 0; rep < 60; ++rep) {
            const size_t o = (size_t)(rep % L) * (layer / 2);
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 96 * 1024, 0, kp + o, vp + o, st, out, map, paged, 0);
            hipDeviceSynchronize();
            unsigned long long h[512];
            hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
            if (rep < 6) continue;
            unsigned long long t0 = ~0ull, t1 = 0;
            for (int b = 0; b < 256; ++b) {
                int c = (b & 7) * 4 + (b >> 6);
                if (map == 1) c ^= 1;
                if (map == 2) c = b >> 3;
                if (map >= 3) c = b >> 6;
                d[c & 3].push_back((h[b * 2 + 1] - h[b * 2]) / 100.0);
                if (map == 0) dc[c & 31].push_back((h[b * 2 + 1] - h[b * 2]) / 100.0);
                t0 = std::min(t0, h[b * 2]); t1 = std::max(t1, h[b * 2 + 1]);
            }
            all.push_back((t1 - t0) / 100.0);
        }
        std::sort(all.begin(), all.end());
        printf("paged=%d map=%d: span med %.2f us (%.0f GB/s) | WG duration median by chunk%%4:", paged, map, all[all.size() / 2],
               2.0 * layer / all[all.size() / 2] / 1e3);
        for (int q = 0; q < 4; ++q) { std::sort(d[q].begin(), d[q].end()); printf(" %.2f", d[q][d[q].size() / 2]); }
        printf("\n");
        if (map == 0 && paged == 1) {   // per-chunk medians (chunk = head index = 256-B piece of the 8-KB token row)
            printf("  per chunk 0..31:");
            for (int c = 0; c < 32; ++c) {
                std::vector<double> v;
                for (size_t i = 0; i < dc[c].size(); ++i) v.push_back(dc[c][i]);
                std::sort(v.begin(), v.end());
                printf(" %.1f", v[v.size() / 2]);
            }
            printf("\n");
        }
    }
    return 0;
}
