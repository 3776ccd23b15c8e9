// skel_bw.hip -- the byte stream of the fused MHA kernel without its exchanges: per workgroup 48 rows of Wqkv
// -> 1/8 of a head's K and V (256-B pieces, S = 4096) -> 16 rows of Wo, in the request order of the kernel.  What is
// the floor of a launch geometry / request schedule before any hand-off costs anything?  201 MB per launch,
// random data, device stamps (first start -> last end).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <type_traits>
typedef _Float16 h16;
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
#define G __attribute__((address_space(1)))
__device__ __forceinline__ h16x8 ld_nt(const h16* p) { return __builtin_nontemporal_load((const G h16x8*)p); }

struct Args {
    const h16* wqkv;   // [12288][4096]
    const h16* kc;     // [4096][4096]
    const h16* vc;     // [4096][4096]
    const h16* wo;     // [4096][4096]
    unsigned long long* st;
    float* out;
};
__device__ __forceinline__ float use(const h16x8& v) { return (float)v[0] + (float)v[7]; }

// ---- A: the kernel's schedule today: 8 wavefronts, row PAIRS (16 loads) x 2 in flight, two 256-token tiles, 2 Wo rows ----
__global__ __launch_bounds__(512, 2) void k_skel8(Args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.x;
    const int h = (b & 7) * 4 + (b >> 6), j = (b >> 3) & 7;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float acc = 0.f;
    h16x8 ga[16], gb[16], ka[8], va[8], kb[8], vb[8], go[16];
    auto lrow2 = [&](h16x8 (&t)[16], const h16* base, int pair) {
#pragma unroll
        for (int i = 0; i < 16; ++i) t[i] = ld_nt(base + (size_t)pair * 8192 + (i * 64 + lane) * 8);
    };
    auto crow2 = [&](const h16x8 (&t)[16]) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += use(t[i]);
    };
    const int gid = wave * 4 + (lane >> 4), d0 = (lane & 15) * 8;
    auto ltile = [&](h16x8 (&k)[8], h16x8 (&v)[8], int tb) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const size_t tok = (size_t)(512 * j + tb + u * 32 + gid);
            k[u] = ld_nt(a.kc + tok * 4096 + h * 128 + d0);
            v[u] = ld_nt(a.vc + tok * 4096 + h * 128 + d0);
        }
    };
    auto ctile = [&](const h16x8 (&k)[8], const h16x8 (&v)[8]) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += use(k[u]) + use(v[u]);
    };
    const int p0 = 24 * b + wave;   // 24 row pairs per workgroup, pair p0 + 8 i
    unsigned long long* tr = a.st + 4096 + b * 8;
#define STAMP(i) do { asm volatile("" : "+v"(acc)); if (threadIdx.x == 0) tr[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
    lrow2(ga, a.wqkv, p0);
    lrow2(gb, a.wqkv, p0 + 8);
    crow2(ga); STAMP(0);
    lrow2(ga, a.wqkv, p0 + 16);
    crow2(gb); STAMP(1);
    ltile(ka, va, 0);
    crow2(ga); STAMP(2);
    ltile(kb, vb, 256);
    ctile(ka, va); STAMP(3);
    lrow2(go, a.wo, 8 * b + wave);
    ctile(kb, vb); STAMP(4);
    crow2(go); STAMP(5);
    a.out[b * 1024 + threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { a.st[b * 2] = t0; a.st[b * 2 + 1] = __builtin_amdgcn_s_memrealtime(); }
}

// ---- B: 16 wavefronts (1024 threads), single rows, 3 rows per wavefront, one 256... 512-token slice = 8 K + 8 V loads per
//      wavefront, 1 Wo row per wavefront; register budget 128 ------------------------------------------------------------
template <int PRE>   // PRE = rows requested up front (2 or 3)
__global__ __launch_bounds__(1024, 4) void k_skel16(Args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.x;
    const int h = (b & 7) * 4 + (b >> 6), j = (b >> 3) & 7;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float acc = 0.f;
    h16x8 r0[8], r1[8], r2[8], kk[8], vv[8];
    auto lrow = [&](h16x8 (&t)[8], const h16* base, int row) {
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = ld_nt(base + (size_t)row * 4096 + (i * 64 + lane) * 8);
    };
    auto crow = [&](const h16x8 (&t)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += use(t[i]);
    };
    const int gid = wave * 4 + (lane >> 4), d0 = (lane & 15) * 8;   // 64 lane groups
    auto ltile = [&]() {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const size_t tok = (size_t)(512 * j + u * 64 + gid);
            kk[u] = ld_nt(a.kc + tok * 4096 + h * 128 + d0);
            vv[u] = ld_nt(a.vc + tok * 4096 + h * 128 + d0);
        }
    };
    const int q0 = 48 * b + wave;   // rows q0, q0 + 16, q0 + 32
    lrow(r0, a.wqkv, q0);
    lrow(r1, a.wqkv, q0 + 16);
    if constexpr (PRE == 3) lrow(r2, a.wqkv, q0 + 32);
    crow(r0);
    if constexpr (PRE == 2) lrow(r2, a.wqkv, q0 + 32);
    crow(r1);
    ltile();                        // 16 loads (r0, r1 free)
    crow(r2);
    lrow(r0, a.wo, 16 * b + wave);  // Wo row
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += use(kk[u]) + use(vv[u]);
    crow(r0);
    a.out[b * 1024 + threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { a.st[b * 2] = t0; a.st[b * 2 + 1] = __builtin_amdgcn_s_memrealtime(); }
}

// ---- C: 8 wavefronts, but single-row granularity (ring of 3 rows = 24 KB per wavefront), K/V as 4 quarter tiles --------
__global__ __launch_bounds__(512, 2) void k_skel8f(Args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.x;
    const int h = (b & 7) * 4 + (b >> 6), j = (b >> 3) & 7;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float acc = 0.f;
    h16x8 r[4][8], kq[4][4], vq[4][4];
    auto lrow = [&](h16x8 (&t)[8], const h16* base, int row) {
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = ld_nt(base + (size_t)row * 4096 + (i * 64 + lane) * 8);
    };
    auto crow = [&](const h16x8 (&t)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += use(t[i]);
    };
    const int gid = wave * 4 + (lane >> 4), d0 = (lane & 15) * 8;
    auto lq = [&](int qd) {   // quarter tile: 4 K + 4 V loads, tokens qd*128 + u*32 + gid
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t tok = (size_t)(512 * j + qd * 128 + u * 32 + gid);
            kq[qd][u] = ld_nt(a.kc + tok * 4096 + h * 128 + d0);
            vq[qd][u] = ld_nt(a.vc + tok * 4096 + h * 128 + d0);
        }
    };
    auto cq = [&](int qd) {
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += use(kq[qd][u]) + use(vq[qd][u]);
    };
    const int q0 = 48 * b + wave;   // rows q0 + 8 i, i < 6
    lrow(r[0], a.wqkv, q0);
    lrow(r[1], a.wqkv, q0 + 8);
    lrow(r[2], a.wqkv, q0 + 16);
    lrow(r[3], a.wqkv, q0 + 24);
    crow(r[0]); lrow(r[0], a.wqkv, q0 + 32);
    crow(r[1]); lrow(r[1], a.wqkv, q0 + 40);
    crow(r[2]); lq(0);
    crow(r[3]); lq(1);
    crow(r[0]); lq(2);
    crow(r[1]); lq(3);
    cq(0); lrow(r[2], a.wo, 16 * b + wave);
    cq(1); lrow(r[3], a.wo, 16 * b + 8 + wave);
    cq(2); cq(3);
    crow(r[2]); crow(r[3]);
    a.out[b * 1024 + threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { a.st[b * 2] = t0; a.st[b * 2 + 1] = __builtin_amdgcn_s_memrealtime(); }
}



// ---- C2: single rows (ring 4) but the two 256-token tiles of today (16 loads each) ----
__global__ __launch_bounds__(512, 2) void k_skel8_rows1_tiles2(Args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.x;
    const int h = (b & 7) * 4 + (b >> 6), j = (b >> 3) & 7;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float acc = 0.f;
    h16x8 r[4][8], ka[8], va[8], kb[8], vb[8];
    auto lrow = [&](h16x8 (&t)[8], const h16* base, int row) {
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = ld_nt(base + (size_t)row * 4096 + (i * 64 + lane) * 8);
    };
    auto crow = [&](const h16x8 (&t)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += use(t[i]);
    };
    const int gid = wave * 4 + (lane >> 4), d0 = (lane & 15) * 8;
    auto ltile = [&](h16x8 (&k)[8], h16x8 (&v)[8], int tb) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const size_t tok = (size_t)(512 * j + tb + u * 32 + gid);
            k[u] = ld_nt(a.kc + tok * 4096 + h * 128 + d0);
            v[u] = ld_nt(a.vc + tok * 4096 + h * 128 + d0);
        }
    };
    const int q0 = 48 * b + wave;
    lrow(r[0], a.wqkv, q0);
    lrow(r[1], a.wqkv, q0 + 8);
    lrow(r[2], a.wqkv, q0 + 16);
    lrow(r[3], a.wqkv, q0 + 24);
    crow(r[0]); lrow(r[0], a.wqkv, q0 + 32);
    crow(r[1]); lrow(r[1], a.wqkv, q0 + 40);
    crow(r[2]); crow(r[3]); ltile(ka, va, 0);
    crow(r[0]); crow(r[1]); ltile(kb, vb, 256);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += use(ka[u]) + use(va[u]);
    lrow(r[2], a.wo, 16 * b + wave);
    lrow(r[3], a.wo, 16 * b + 8 + wave);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += use(kb[u]) + use(vb[u]);
    crow(r[2]); crow(r[3]);
    a.out[b * 1024 + threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { a.st[b * 2] = t0; a.st[b * 2 + 1] = __builtin_amdgcn_s_memrealtime(); }
}
// ---- C3: row pairs x2 (today) but quarter tiles ----
__global__ __launch_bounds__(512, 2) void k_skel8_rows2_tilesq(Args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.x;
    const int h = (b & 7) * 4 + (b >> 6), j = (b >> 3) & 7;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float acc = 0.f;
    h16x8 ga[16], gb[16], kq[4][4], vq[4][4];
    auto lrow2 = [&](h16x8 (&t)[16], const h16* base, int pair) {
#pragma unroll
        for (int i = 0; i < 16; ++i) t[i] = ld_nt(base + (size_t)pair * 8192 + (i * 64 + lane) * 8);
    };
    auto crow2 = [&](const h16x8 (&t)[16]) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += use(t[i]);
    };
    const int gid = wave * 4 + (lane >> 4), d0 = (lane & 15) * 8;
    auto lq = [&](int qd) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t tok = (size_t)(512 * j + qd * 128 + u * 32 + gid);
            kq[qd][u] = ld_nt(a.kc + tok * 4096 + h * 128 + d0);
            vq[qd][u] = ld_nt(a.vc + tok * 4096 + h * 128 + d0);
        }
    };
    auto cq = [&](int qd) {
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += use(kq[qd][u]) + use(vq[qd][u]);
    };
    const int p0 = 24 * b + wave;
    lrow2(ga, a.wqkv, p0);
    lrow2(gb, a.wqkv, p0 + 8);
    crow2(ga);
    lrow2(ga, a.wqkv, p0 + 16);
    crow2(gb);
    lq(0); lq(1);
    crow2(ga);
    lq(2); lq(3);
    cq(0); cq(1);
    lrow2(gb, a.wo, 8 * b + wave);
    cq(2); cq(3);
    crow2(gb);
    a.out[b * 1024 + threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { a.st[b * 2] = t0; a.st[b * 2 + 1] = __builtin_amdgcn_s_memrealtime(); }
}


// ---- F: 8 waves, 1-KB granularity: every consumed 1-KB chunk is re-requested at once (32 loads in flight per wavefront,
//      a steady request stream); K/V and Wo chunks take over the same slots ------------------------------------------------
template <int NBUF>
__global__ __launch_bounds__(512, 2) void k_skel8_chunk(Args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.x;
    const int h = (b & 7) * 4 + (b >> 6), j = (b >> 3) & 7;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float acc = 0.f;
    constexpr int N = NBUF * 8;
    h16x8 r[N];
    const int gid = wave * 4 + (lane >> 4), d0 = (lane & 15) * 8;
    const int q0 = 48 * b + wave;
    // chunk sequence of this wavefront: 48 Wqkv chunks (6 rows x 8), 32 K/V chunks (16 K + 16 V interleaved), 16 Wo chunks
    auto chunk_ptr = [&](int c) -> const h16* {
        if (c < 48) return a.wqkv + (size_t)(q0 + 8 * (c >> 3)) * 4096 + ((c & 7) * 64 + lane) * 8;
        if (c < 80) {
            const int u = (c - 48) >> 1;
            const size_t tok = (size_t)(512 * j + u * 32 + gid);
            return ((c & 1) ? a.vc : a.kc) + tok * 4096 + h * 128 + d0;
        }
        const int cc = c - 80;
        return a.wo + (size_t)(16 * b + wave + 8 * (cc >> 3)) * 4096 + ((cc & 7) * 64 + lane) * 8;
    };
#pragma unroll
    for (int c = 0; c < N; ++c) r[c] = ld_nt(chunk_ptr(c));
#pragma unroll
    for (int c = 0; c < 96; ++c) {
        acc += use(r[c % N]);
        if (c + N < 96) r[c % N] = ld_nt(chunk_ptr(c + N));
        __builtin_amdgcn_sched_barrier(0);
    }
    a.out[b * 1024 + threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { a.st[b * 2] = t0; a.st[b * 2 + 1] = __builtin_amdgcn_s_memrealtime(); }
}
// ---- G: 16 waves, 1-KB granularity, ring of NBUF rows (8 chunks each) per wavefront ----
template <int NBUF>
__global__ __launch_bounds__(1024, 4) void k_skel16_chunk(Args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.x;
    const int h = (b & 7) * 4 + (b >> 6), j = (b >> 3) & 7;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float acc = 0.f;
    constexpr int N = NBUF * 8;
    h16x8 r[N];
    const int gid = wave * 4 + (lane >> 4), d0 = (lane & 15) * 8;   // 64 lane groups
    const int q0 = 48 * b + wave;
    // 24 Wqkv chunks (3 rows), 16 K/V chunks, 8 Wo chunks
    auto chunk_ptr = [&](int c) -> const h16* {
        if (c < 24) return a.wqkv + (size_t)(q0 + 16 * (c >> 3)) * 4096 + ((c & 7) * 64 + lane) * 8;
        if (c < 40) {
            const int u = (c - 24) >> 1;
            const size_t tok = (size_t)(512 * j + u * 64 + gid);
            return ((c & 1) ? a.vc : a.kc) + tok * 4096 + h * 128 + d0;
        }
        const int cc = c - 40;
        return a.wo + (size_t)(16 * b + wave) * 4096 + ((cc & 7) * 64 + lane) * 8;
    };
#pragma unroll
    for (int c = 0; c < N; ++c) r[c] = ld_nt(chunk_ptr(c));
#pragma unroll
    for (int c = 0; c < 48; ++c) {
        acc += use(r[c % N]);
        if (c + N < 48) r[c % N] = ld_nt(chunk_ptr(c + N));
        __builtin_amdgcn_sched_barrier(0);
    }
    a.out[b * 1024 + threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { a.st[b * 2] = t0; a.st[b * 2 + 1] = __builtin_amdgcn_s_memrealtime(); }
}


// ---- H: as A, but K/V as 1-KB pieces: a workgroup serves the 4 heads of its XCD group for 1/32 of the tokens
//      (one token's 4-head strip per wavefront instruction) instead of one head for 1/8 of the tokens ------------------
__global__ __launch_bounds__(512, 2) void k_skel8_kv4(Args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.x;
    const int hg = b & 7, jj = b >> 3;      // head group (4 heads = 1 KB per token), token slice of 128
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float acc = 0.f;
    h16x8 ga[16], gb[16], ka[8], va[8], kb[8], vb[8], go[16];
    auto lrow2 = [&](h16x8 (&t)[16], const h16* base, int pair) {
#pragma unroll
        for (int i = 0; i < 16; ++i) t[i] = ld_nt(base + (size_t)pair * 8192 + (i * 64 + lane) * 8);
    };
    auto crow2 = [&](const h16x8 (&t)[16]) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += use(t[i]);
    };
    auto ltile = [&](h16x8 (&k)[8], h16x8 (&v)[8], int tb) {   // 8 tokens per wavefront and tile: token = 128 jj + tb + 8 u + wave
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const size_t tok = (size_t)(128 * jj + tb + u * 8 + wave);
            k[u] = ld_nt(a.kc + tok * 4096 + hg * 512 + lane * 8);
            v[u] = ld_nt(a.vc + tok * 4096 + hg * 512 + lane * 8);
        }
    };
    auto ctile = [&](const h16x8 (&k)[8], const h16x8 (&v)[8]) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += use(k[u]) + use(v[u]);
    };
    const int p0 = 24 * b + wave;
    lrow2(ga, a.wqkv, p0);
    lrow2(gb, a.wqkv, p0 + 8);
    crow2(ga);
    lrow2(ga, a.wqkv, p0 + 16);
    crow2(gb);
    ltile(ka, va, 0);
    crow2(ga);
    ltile(kb, vb, 64);
    ctile(ka, va);
    lrow2(go, a.wo, 8 * b + wave);
    ctile(kb, vb);
    crow2(go);
    a.out[b * 1024 + threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { a.st[b * 2] = t0; a.st[b * 2 + 1] = __builtin_amdgcn_s_memrealtime(); }
}

// ---- D: as A, but the 24 row pairs per workgroup are not static: wavefronts claim row pairs from per-XCD pools (768 pairs
//      each, atomic counters), one claim always a full group ahead; when a pool is dry the buffer is refilled with a K/V tile
//      instead.  The first pair of every wavefront is static (no atomic round trip before the first request). ------------
__device__ __forceinline__ unsigned xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 7u;
}
template <int STEAL>
__global__ __launch_bounds__(512, 2) void k_skel8dyn(Args a, unsigned* ctr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.x;
    const int h = (b & 7) * 4 + (b >> 6), j = (b >> 3) & 7;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float acc = 0.f;
    h16x8 bufA[16], bufB[16], go[16];
    const unsigned xcd = xcc_id();
    // pool x: pairs x*768 .. x*768+767; the first 256 of each pool are the static first pairs of the XCD's 256 wavefronts
    unsigned* myctr = ctr + xcd * 32;
    // scalar atomic on the XCD's own counter (L2 of this XCD; SMEM path, lgkmcnt -- it does not queue with the vector loads)
    auto claim = [&]() -> unsigned {
        unsigned v = 1;
        asm volatile("s_atomic_add %0, %1, 0x0 glc" : "+s"(v) : "s"(myctr) : "memory");
        return v;
    };
    auto landed = [&](unsigned v) -> unsigned {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v) :: "memory");
        return v;
    };
    auto lrow2 = [&](h16x8 (&t)[16], const h16* base, unsigned pair) {
#pragma unroll
        for (int i = 0; i < 16; ++i) t[i] = ld_nt(base + (size_t)pair * 8192 + (i * 64 + lane) * 8);
    };
    auto crow2 = [&](const h16x8 (&t)[16]) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += use(t[i]);
    };
    const int gid = wave * 4 + (lane >> 4), d0 = (lane & 15) * 8;
    auto ltile = [&](h16x8 (&t)[16], int tb) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const size_t tok = (size_t)(512 * j + tb + u * 32 + gid);
            t[u] = ld_nt(a.kc + tok * 4096 + h * 128 + d0);
            t[8 + u] = ld_nt(a.vc + tok * 4096 + h * 128 + d0);
        }
    };
    const unsigned POOL = 768, STATIC = 256;
    const unsigned widx = (unsigned)((b >> 3) * 8 + wave);      // wavefront index within the XCD (b % 8 == xcd assumed for speed only)
    unsigned c_next = claim();                                  // for bufB's first fill
    lrow2(bufA, a.wqkv, xcd * POOL + widx);
    unsigned c1 = STATIC + landed(c_next);
    unsigned c_after = claim();
    bool tileA = false, tileB = false;
    if (c1 < POOL) lrow2(bufB, a.wqkv, xcd * POOL + c1); else { ltile(bufB, 0); tileA = true; }   // (never at start)
    // steady state: consume A, refill A; consume B, refill B
    int ntile = tileA ? 1 : 0;
    bool a_is_tile = false, b_is_tile = tileA;
    while (true) {
        // ---- buffer A
        crow2(bufA);
        {
            const unsigned c = STATIC + landed(c_after);
            c_after = claim();
            if (c < POOL) lrow2(bufA, a.wqkv, xcd * POOL + c);
            else { ltile(bufA, ntile * 256); ++ntile; a_is_tile = true; }
        }
        if (b_is_tile) break;
        crow2(bufB);
        {
            const unsigned c = STATIC + landed(c_after);
            c_after = claim();
            if (c < POOL) lrow2(bufB, a.wqkv, xcd * POOL + c);
            else { ltile(bufB, ntile * 256); ++ntile; b_is_tile = true; }
        }
        if (a_is_tile) break;
    }
    // here one buffer holds tile 0 and was requested first; the other still holds rows or tile 1
    if (ntile < 2) {   // the other buffer has rows: consume them, then request tile 1 into it
        if (a_is_tile) { crow2(bufB); ltile(bufB, 256); } else { crow2(bufA); ltile(bufA, 256); }
    }
    crow2(bufA);
    lrow2(go, a.wo, 8 * b + wave);
    crow2(bufB);
    crow2(go);
    a.out[b * 1024 + threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { a.st[b * 2] = t0; a.st[b * 2 + 1] = __builtin_amdgcn_s_memrealtime(); }
}

__global__ void k_fill(unsigned* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned x = (unsigned)i * 2654435761u + 12345u;
        x ^= x >> 13; x *= 0x5bd1e995u; x ^= x >> 15;
        p[i] = (x & 0x83ff83ffu) | 0x38003800u;
    }
}

template <class K>
void run(const char* name, K kern, int threads, h16* buf, size_t nsets, unsigned long long* st, float* out, unsigned* ctr = nullptr) {
    const int blocks = 256;
    std::vector<unsigned long long> h(blocks * 2);
    std::vector<double> walls, spans;
    const size_t set = (size_t)201326592 / 2;   // elements per layer set
    for (int rep = 0; rep < 45; ++rep) {
        h16* base = buf + (size_t)(rep % nsets) * set;
        Args a{base, base + (size_t)12288 * 4096, base + (size_t)16384 * 4096, base + (size_t)20480 * 4096, st, out};
        if constexpr (std::is_invocable_v<K, Args, unsigned*>) { hipMemsetAsync(ctr, 0, 8 * 128, 0); hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, a, ctr); }
        else hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, a);
        if (hipDeviceSynchronize() != hipSuccess) { printf("%s FAILED\n", name); return; }
        hipMemcpy(h.data(), st, blocks * 16, hipMemcpyDeviceToHost);
        if (rep < 5) continue;
        unsigned long long t0 = ~0ull, t1 = 0, e0 = ~0ull;
        for (int b = 0; b < blocks; ++b) { t0 = std::min(t0, h[b * 2]); t1 = std::max(t1, h[b * 2 + 1]); e0 = std::min(e0, h[b * 2 + 1]); }
        walls.push_back((t1 - t0) / 100.0);
        spans.push_back((t1 - e0) / 100.0);
    }
    if (name[0] == 'A') {
        std::vector<unsigned long long> tr(blocks * 8);
        hipMemcpy(tr.data(), st + 4096, blocks * 64, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull;
        for (int b = 0; b < blocks; ++b) t0 = std::min(t0, h[b * 2]);
        const char* nm[6] = {"pair0 consumed", "pair1 consumed", "pair2 consumed (P1 done)", "tile A consumed", "tile B consumed", "Wo consumed"};
        for (int i = 0; i < 6; ++i) {
            std::vector<double> v;
            for (int b = 0; b < blocks; ++b) v.push_back((tr[b * 8 + i] - t0) / 100.0);
            std::sort(v.begin(), v.end());
            printf("    wave 0 of each WG, last launch: %-26s min %.2f med %.2f p90 %.2f max %.2f\n", nm[i], v[0], v[128], v[230], v[255]);
        }
    }
    std::sort(walls.begin(), walls.end());
    std::sort(spans.begin(), spans.end());
    const double med = walls[walls.size() / 2];
    printf("%-52s wall med %.2f us (min %.2f p90 %.2f) -> %.0f GB/s   first-end..last-end %.2f us\n", name, med, walls.front(),
           walls[walls.size() * 9 / 10], 201326592.0 / med / 1e3, spans[spans.size() / 2]);
    fflush(stdout);
}

int main() {
    const size_t nsets = 10;
    const size_t bytes = nsets * (size_t)201326592;
    h16* w; float* out; unsigned long long* st;
    hipMalloc(&w, bytes); hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&st, 8192 * 16 + 4096 * 8); unsigned* ctr; hipMalloc(&ctr, 8 * 128);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (unsigned*)w, bytes / 4);
    hipDeviceSynchronize();
    for (int pass = 0; pass < 2; ++pass) {
        run("A  8 waves, row pairs x2, 2 tiles, 2 Wo rows (today)", k_skel8, 512, w, nsets, st, out);
        run("C  8 waves, single rows ring 4, quarter tiles", k_skel8f, 512, w, nsets, st, out);
        run("H  as A, K/V as 1-KB pieces (4 heads x 1/32 tokens)", k_skel8_kv4, 512, w, nsets, st, out);
    }
    return 0;
}
