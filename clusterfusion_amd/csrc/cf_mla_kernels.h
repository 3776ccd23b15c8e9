// cf_mla_kernels.h -- DeepSeek-V2-Lite MLA decoder-layer (attention block) decode, batch 1, gfx950.
//
// Reference: include/H100/deepseek/kernel.cuh:9-697 (one 4-CTA cluster per head, every projection a
// TMA-fed GEMV, the absorbed query q_abs = q_nope W_uk attends the 512-wide latent cache).  The same math
// here is three launches on one stream (the op has ~27 MB of weights and 4.7 MB of cache: it is bound by the
// chain of dependent stages, not by bytes); inside a launch, dependent stages hand small vectors over through
// tagged granules ({epoch, fp32} in one 8-byte write-through store, cf_fused_kernel.h: the data is the flag,
// no fences), consumers having requested their weights BEFORE they wait:
//
//   k_mla_ab    A  RMSNorm(x) . [W_q_nope | W_kv | W_q_pe | W_k_pe]   split-K partials -> granules
//               B  q_abs[h] = q_nope[h] . W_uk[:, h]  -> fp16 B operand of the scores;
//                  one extra workgroup: ckv RMSNorm, RoPE of q_pe / k_pe, the new token's latent row
//   k_mla_attn  C  scores[16 heads x tokens] and O[16 heads x 512] on the matrix cores: the 16 query heads
//                  share ONE latent cache (MQA with 16 query rows = exactly one MFMA tile edge);
//                  64 tokens per workgroup and step, partial (m, l, O) per workgroup
//   k_mla_de    D  merge of the partials (in the operand loader) . W_uv[:, h]   split-K partials -> granules
//               E  o_h . W_o; the workgroup of a column strip's last K-slice sums the 8 slices in slice order
//                  and writes fp16 `out` (kernel.cuh:680,695 uses fp16 atomics instead)
//
// Consumers always have higher workgroup ids than their producers, so in-order dispatch cannot starve a
// producer; every wait is a bounded spin that raises the workspace's error word instead of hanging.
//
// All weights are [in,out] (deepseek_kernel_dispatch.cu:55-206): a GEMV reads whole 128-byte row segments of a
// 64-column strip, 8 rows per wavefront instruction (lane l: row l / 8, columns 8 (l % 8) .. + 7), the
// 8 row groups meet by cross-lane adds, the 8 wavefronts (K sub-slices) in LDS -- every sum in a fixed order.
#pragma once
#include "cf_device.h"

namespace cf {

constexpr int MLA_HID = 2048, MLA_H = 16, MLA_NOPE = 128, MLA_ROPE = 64, MLA_L = 512, MLA_LAT = 576;
// stage A output columns: [0, 2048) q_nope | [2048, 2560) ckv | [2560, 3584) q_pe | [3584, 3648) k_pe
constexpr int MLA_A_CKV = 2048, MLA_A_QPE = 2560, MLA_A_KPE = 3584, MLA_A_COLS = 3648;
constexpr int MLA_A_KS = 8, MLA_D_KS = 4, MLA_E_KS = 8;
constexpr int MLA_NSPLIT_MAX = 256;

typedef h16 h16x4_t __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));

typedef unsigned long long u64;
constexpr unsigned MLA_SPIN_LIMIT = 400000u;

__device__ __forceinline__ void mla_granule_store(u64* p, unsigned epoch, float v) {
    __hip_atomic_store(p, ((u64)epoch << 32) | (u64)__builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
// plain (write-back) store that stops in this XCD's L2: visible to agent-scope loads of workgroups on the SAME XCD only
// (cf_fused_kernel.h granule_store_to); used when the published XCC ids say producer and consumers share the XCD
__device__ __forceinline__ void mla_granule_store_to(u64* p, unsigned epoch, float v, bool xcd_local) {
    const u64 g = ((u64)epoch << 32) | (u64)__builtin_bit_cast(unsigned, v);
    if (xcd_local) __hip_atomic_store(p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// a failed exchange: the workspace's error word + the host-mapped word the library reads at its next call
// (cf_fused_kernel.h flag_exchange_error; state[4..5] = device address of that word, set by cf_workspace_init)
__device__ __forceinline__ void mla_flag_error(unsigned* err /* = state + 1 */, unsigned code) {
    atomicCAS(err, 0u, code);
    unsigned* host = *reinterpret_cast<unsigned* const*>(err + 3);
    if (host) __hip_atomic_store(host, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// sum of NP granules g[p * stride] in index order once all carry this epoch; `live` = this thread takes part
template <int NP>
__device__ __forceinline__ float mla_granule_sum(const u64* g, size_t stride, unsigned epoch, bool live,
                                                 unsigned* err, unsigned code) {
    float v = 0.f;
    if (!live) return v;
    for (unsigned spin = 0;; ++spin) {
        u64 x[NP];
        bool ok = true;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            x[p] = __hip_atomic_load(g + (size_t)p * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok &= (unsigned)(x[p] >> 32) == epoch;
        }
        if (ok) {
#pragma unroll
            for (int p = 0; p < NP; ++p) v += __builtin_bit_cast(float, (unsigned)x[p]);
            return v;
        }
        if (spin > MLA_SPIN_LIMIT) {
            mla_flag_error(err, code);
            return 0.f;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

// ---- 64-column strip x (8 NL)-row slice of an [in,out] matrix -------------------------------------------------
template <int NL>
struct ColTile {
    h16x8 w[NL];
    __device__ __forceinline__ void load(const h16* wp /* W + k0 * ld + col0 */, int ld, int lane) {
        const h16* p = wp + (size_t)(lane >> 3) * ld + (lane & 7) * 8;
#pragma unroll
        for (int i = 0; i < NL; ++i) w[i] = ld_stream(p + (size_t)(8 * i) * ld);
    }
    // xs: the 8 NL inputs of this slice (LDS, fp32); returns this lane's 8 columns summed over its row group
    __device__ __forceinline__ void fma(const float* xs, int lane, float (&acc)[8]) const {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const float xv = xs[8 * i + (lane >> 3)];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = __builtin_fmaf((float)w[i][e], xv, acc[e]);
        }
    }
};

// the 8 row groups of a wavefront (lane bits 3..5) -> lanes 0..7; then the 8 wavefronts through LDS.
// Returns, for tid < 64, the strip's column tid summed over the workgroup's whole K-slice.
__device__ __forceinline__ float strip_reduce(float (&acc)[8], float (*s_red)[64], int tid) {
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float v = acc[e];
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        acc[e] = v;
    }
    if (lane < 8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) s_red[wave][lane * 8 + e] = acc[e];
    }
    __syncthreads();
    float v = 0.f;
    if (tid < 64) {
#pragma unroll
        for (int w = 0; w < 8; ++w) v += s_red[w][tid];
    }
    return v;
}

// ---- A + B: input projections, absorbed query, the new token's latent ---------------------------------------------
struct MlaAbArgs {
    unsigned* state;          // [0] epoch of the last completed call, [1] first error code
    const h16 *x, *rms_w;
    float eps;
    const h16 *w_q_nope, *w_kv, *w_q_pe, *w_k_pe;
    int n_a;                  // workgroups of stage A = strips * MLA_A_KS (strips = 40 without the rope parts, 57 with)
    u64* g_a;                 // [MLA_A_KS][MLA_A_COLS] granules: split-K partials of stage A
    const h16* w_uk;          // [128, 16 * 512]
    const h16* rms_ckv_w;     // [512]
    const float *cos, *sin;   // [64] each
    int with_pe;              // q_pe / k_pe are projected
    h16* qlat;                // [16][576] fp16: q_abs | RoPE(q_pe)
    h16* latent_new;          // [576] fp16: RMSNorm(ckv) | RoPE(k_pe)
    h16* latent_out;          // same, caller's copy (or null)
};

// rotate-half over 64 dims (kernel.cuh:298-315): out[i] = v[i] cos[i] - v[i+32] sin[i+32] (i < 32),
//                                                out[i] = v[i] cos[i] + v[i-32] sin[i-32] (i >= 32)
__device__ __forceinline__ float mla_rope(const float* v64, int i, const float* cos, const float* sin) {
    return i < 32 ? v64[i] * cos[i] - v64[i + 32] * sin[i + 32] : v64[i] * cos[i] + v64[i - 32] * sin[i - 32];
}

// grid = n_a + 128 (head h = b / 8, columns 64 (b % 8) of its 512) + 1; 512 threads
__global__ __launch_bounds__(512) void k_mla_ab(MlaAbArgs a) {
    __shared__ float s_x[MLA_H * MLA_ROPE + MLA_ROPE];
    __shared__ float s_red[8][64];
    __shared__ float s_ss[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned epoch = a.state[0] + 1u;      // (advanced by k_mla_attn, the next launch on the stream)
    if ((int)blockIdx.x < a.n_a) {
        // ---- A: one 64-column strip x one 256-row K-slice
        const int strip = blockIdx.x / MLA_A_KS, ks = blockIdx.x % MLA_A_KS;
        const h16* W;
        int ld, col0;
        if (strip < 32) { W = a.w_q_nope; ld = MLA_H * MLA_NOPE; col0 = 64 * strip; }
        else if (strip < 40) { W = a.w_kv; ld = MLA_L; col0 = 64 * (strip - 32); }
        else if (strip < 56) { W = a.w_q_pe; ld = MLA_H * MLA_ROPE; col0 = 64 * (strip - 40); }
        else { W = a.w_k_pe; ld = MLA_ROPE; col0 = 0; }
        ColTile<4> t;
        t.load(W + (size_t)(ks * 256 + wave * 32) * ld + col0, ld, lane);      // the weight stream starts before the norm
        float ss = 0.f;
        {
            const h16x2* xp = reinterpret_cast<const h16x2*>(a.x) + tid * 2;      // 4 halves per thread
            const h16x2 v0 = xp[0], v1 = xp[1];
            const float x0 = (float)v0[0], x1 = (float)v0[1], x2 = (float)v1[0], x3 = (float)v1[1];
            ss = x0 * x0 + x1 * x1 + x2 * x2 + x3 * x3;
        }
        float xk = 0.f, wk = 0.f;
        if (tid < 256) {
            xk = (float)a.x[ks * 256 + tid];
            wk = (float)a.rms_w[ks * 256 + tid];
        }
        ss = sum64(ss);
        if (lane == 0) s_ss[wave] = ss;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) tot += s_ss[w];
        if (tid < 256) s_x[tid] = xk * __builtin_amdgcn_rsqf(tot / (float)MLA_HID + a.eps) * wk;
        __syncthreads();
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        t.fma(s_x + wave * 32, lane, acc);
        const float v = strip_reduce(acc, s_red, tid);
        if (tid < 64) mla_granule_store(a.g_a + (size_t)ks * MLA_A_COLS + 64 * strip + tid, epoch, v);
        return;
    }
    const int b = blockIdx.x - a.n_a;
    if (b < 128) {
        // ---- B: q_abs of head h, 64 of its 512 latent columns
        const int h = b >> 3, c0 = 64 * (b & 7);
        ColTile<2> t;
        t.load(a.w_uk + (size_t)(wave * 16) * (MLA_H * MLA_L) + h * MLA_L + c0, MLA_H * MLA_L, lane);
        const float q = mla_granule_sum<MLA_A_KS>(a.g_a + h * MLA_NOPE + tid, MLA_A_COLS, epoch, tid < MLA_NOPE, a.state + 1, 1u);
        if (tid < MLA_NOPE) s_x[tid] = q;
        __syncthreads();
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        t.fma(s_x + wave * 16, lane, acc);
        const float v = strip_reduce(acc, s_red, tid);
        if (tid < 64) a.qlat[h * MLA_LAT + c0 + tid] = (h16)v;
        return;
    }
    // ---- the extra workgroup: everything that needs a whole small vector
    const float rw = (float)a.rms_ckv_w[tid];
    const float ckv = mla_granule_sum<MLA_A_KS>(a.g_a + MLA_A_CKV + tid, MLA_A_COLS, epoch, true, a.state + 1, 2u);   // 512 threads = 512 dims
    float ss = sum64(ckv * ckv);
    if (lane == 0) s_ss[wave] = ss;
    if (a.with_pe) {
        s_x[tid] = mla_granule_sum<MLA_A_KS>(a.g_a + MLA_A_QPE + tid, MLA_A_COLS, epoch, true, a.state + 1, 2u);
        s_x[512 + tid] = mla_granule_sum<MLA_A_KS>(a.g_a + MLA_A_QPE + 512 + tid, MLA_A_COLS, epoch, true, a.state + 1, 2u);
        const float kp = mla_granule_sum<MLA_A_KS>(a.g_a + MLA_A_KPE + tid, MLA_A_COLS, epoch, tid < MLA_ROPE, a.state + 1, 2u);
        if (tid < MLA_ROPE) s_x[1024 + tid] = kp;
    }
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) tot += s_ss[w];
    const h16 cn = (h16)(ckv * __builtin_amdgcn_rsqf(tot / (float)MLA_L + a.eps) * rw);
    a.latent_new[tid] = cn;
    if (a.latent_out) a.latent_out[tid] = cn;
    if (a.with_pe) {
        for (int i = tid; i < MLA_H * MLA_ROPE; i += 512) {
            const int h = i >> 6, d = i & 63;
            a.qlat[h * MLA_LAT + MLA_L + d] = (h16)mla_rope(s_x + h * 64, d, a.cos, a.sin);
        }
        if (tid < MLA_ROPE) {
            const h16 kp = (h16)mla_rope(s_x + 1024, tid, a.cos, a.sin);
            a.latent_new[MLA_L + tid] = kp;
            if (a.latent_out) a.latent_out[MLA_L + tid] = kp;
        }
    } else if (tid < MLA_ROPE) {
        a.latent_new[MLA_L + tid] = (h16)0.f;
        if (a.latent_out) a.latent_out[MLA_L + tid] = (h16)0.f;
    }
}

// ---- C: attention over the latent cache -----------------------------------------------------------------------------
struct MlaAttnArgs {
    unsigned* state;          // [0] epoch counter of the workspace
    const h16* qlat;          // [16][576]
    const h16* cache;         // [n_tok][576]; row n_tok - 1 is the new token's slot and is NOT read
    const h16* latent_new;    // [576]
    int n_tok;                // attended entries = cache rows 0 .. n_tok - 2 and the new token
    int iters;                // 64-token steps per workgroup
    float scale_log2e;
    float* part_o;            // [nsplit][16][512]
    float* part_ml;           // [nsplit][16][2]
};

// grid = nsplit; 256 threads (4 wavefronts, one 16-token tile each per step).
// scores: D[token][head] = sum_j mfma_16x16x32(A = latent rows (lane l: token l % 16, k = 32 j + 8 (l / 16) ..),
//                                               B = q (lane l: head l % 16, same k))
// O:      D[head][col]  += mfma_16x16x16(A = P (lane l: head l % 16, tokens 4 (l / 16) + r -- the layout the score
//                                        accumulators already have), B = V[token][col] read transposed from LDS)
// PE: the 64 rope columns enter the scores (the reference's kernel never reads them, kernel.cuh:407-408)
template <bool PE>
__global__ __launch_bounds__(256) void k_mla_attn(MlaAttnArgs a) {
    constexpr int NJ = PE ? 18 : 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    h16* s_v = reinterpret_cast<h16*>(smem);                                   // [4 tiles][32 blocks][16 tok][16 col]
    h16x4_t* s_p = reinterpret_cast<h16x4_t*>(smem + 4 * 16384);               // [4 tiles][64 lanes]
    float* s_m = reinterpret_cast<float*>(smem + 4 * 16384 + 2048);            // [4][16] tile max
    float* s_l = s_m + 64;                                                     // [4][16] tile sum
    float* s_al = s_l + 64;                                                    // [4 waves][16] rescale of this step
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t16 = lane & 15, kq = lane >> 4;
    const float NEG = -3.0e38f;

    h16x8 qb[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) qb[j] = ld_h8(a.qlat + t16 * MLA_LAT + 32 * j + 8 * kq);

    f32x4 acc[8];
#pragma unroll
    for (int cb = 0; cb < 8; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = NEG, l_run = 0.f;       // of head t16 (identical in every wavefront)

    for (int it = 0; it < a.iters; ++it) {
        const int tb = (blockIdx.x * a.iters + it) * 64 + wave * 16;
        const int t = tb + t16;
        // (rows past the end are masked below; they read the new token's row, which is always finite)
        const h16* rowp = t < a.n_tok - 1 ? a.cache + (size_t)t * MLA_LAT : a.latent_new;
        h16x8 av[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) av[j] = ld_stream(rowp + 32 * j + 8 * kq);
        if (it) __syncthreads();          // the previous step's V image / P have been consumed
        f32x4 sc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NJ; ++j) sc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[j], qb[j], sc, 0, 0, 0);
        // V image of this tile: lane holds token t16, columns 32 j + 8 kq .. + 7 -> block 2 j + kq / 2
#pragma unroll
        for (int j = 0; j < 16; ++j)
            *reinterpret_cast<h16x8*>(s_v + wave * 8192 + (2 * j + (kq >> 1)) * 256 + t16 * 16 + 8 * (kq & 1)) = av[j];
        float s2[4], mx = NEG;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s2[r] = (tb + 4 * kq + r) < a.n_tok ? sc[r] * a.scale_log2e : NEG;
            mx = fmaxf(mx, s2[r]);
        }
        mx = xmax32(xmax16(mx));
        if (kq == 0) s_m[wave * 16 + t16] = mx;
        __syncthreads();
        const float m_new = fmaxf(fmaxf(fmaxf(s_m[t16], s_m[16 + t16]), fmaxf(s_m[32 + t16], s_m[48 + t16])), m_run);
        const float alpha = fast_exp2(m_run - m_new);       // first step: exp2(-huge) = 0 and acc = 0
        h16x4_t pa;
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float p = (tb + 4 * kq + r) < a.n_tok ? fast_exp2(s2[r] - m_new) : 0.f;
            ps += p;
            pa[r] = (h16)p;
        }
        ps = xsum32(xsum16(ps));
        s_p[wave * 64 + lane] = pa;
        if (kq == 0) {
            s_l[wave * 16 + t16] = ps;
            s_al[wave * 16 + t16] = alpha;
        }
        __syncthreads();
        l_run = l_run * alpha + ((s_l[t16] + s_l[16 + t16]) + (s_l[32 + t16] + s_l[48 + t16]));
        m_run = m_new;
        float al[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) al[r] = s_al[wave * 16 + 4 * kq + r];
#pragma unroll
        for (int cb = 0; cb < 8; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[cb][r] *= al[r];
        // this wavefront's 128 output columns over the 4 tiles of the step
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const h16x4_t pt = s_p[tt * 64 + lane];
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) {
                const fp16x4_t vt = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                    (__attribute__((address_space(3))) fp16x4_t*)(s_v + tt * 8192 + (8 * wave + cb) * 256 + t16 * 4 + kq * 64));
                acc[cb] = __builtin_amdgcn_mfma_f32_16x16x16f16(pt, __builtin_bit_cast(h16x4_t, vt), acc[cb], 0, 0, 0);
            }
        }
    }
    float* po = a.part_o + (size_t)blockIdx.x * (MLA_H * MLA_L);
#pragma unroll
    for (int cb = 0; cb < 8; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) po[(4 * kq + r) * MLA_L + 128 * wave + 16 * cb + t16] = acc[cb][r];
    if (blockIdx.x == 0 && tid == 0) a.state[0] += 1u;       // the exchanges of k_mla_ab (before) and k_mla_de (after) use it
    if (wave == 0 && kq == 0) {
        a.part_ml[(size_t)blockIdx.x * 32 + 2 * t16] = m_run;
        a.part_ml[(size_t)blockIdx.x * 32 + 2 * t16 + 1] = l_run;
    }
}

// ---- D + E: merge the partials, . W_uv, . W_o ---------------------------------------------------------------------------
struct MlaDeArgs {
    unsigned* state;          // [0] this call's epoch (k_mla_attn advanced it), [1] first error code
    const float *part_o, *part_ml;
    int nsplit;
    const h16* w_uv;          // [512, 16 * 128]
    u64* g_d;                 // [MLA_D_KS][2048] granules: split-K partials of o_h
    const h16* w_o;           // [2048, 2048]
    u64* g_e;                 // [MLA_E_KS][2048] granules: split-K partials of out
    h16* out;                 // [2048]
};

constexpr int MLA_D_WGS = MLA_H * 2 * MLA_D_KS;     // 128
constexpr int MLA_E_WGS = 32 * MLA_E_KS;            // 256

// grid = 128 (D: head b / 8, strip (b / 4) % 2, K-slice b % 4) + 256 (E: strip b / 8, K-slice b % 8); 512 threads
__global__ __launch_bounds__(512) void k_mla_de(MlaDeArgs a) {
    __shared__ float s_w[MLA_NSPLIT_MAX];
    __shared__ __attribute__((aligned(16))) float s_xp[16][128];
    __shared__ float s_x[256];
    __shared__ float s_red[8][64];
    __shared__ float s_r8[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned epoch = a.state[0];
    if (blockIdx.x < MLA_D_WGS) {
        const int h = blockIdx.x >> 3, c2 = (blockIdx.x >> 2) & 1, ks = blockIdx.x & 3;
        const int k0 = ks * 128;
        ColTile<2> t;
        t.load(a.w_uv + (size_t)(k0 + wave * 16) * (MLA_H * MLA_NOPE) + h * MLA_NOPE + 64 * c2, MLA_H * MLA_NOPE, lane);
        // weights of the partials: w_s = exp2(m_s - M), denominator sum_s w_s l_s
        const float NEG = -3.0e38f;
        float m = NEG, l = 0.f;
        if (tid < a.nsplit) {
            m = a.part_ml[(size_t)tid * 32 + 2 * h];
            l = a.part_ml[(size_t)tid * 32 + 2 * h + 1];
        }
        float mx = m;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        if (lane == 0) s_r8[wave] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(fmaxf(s_r8[0], s_r8[1]), fmaxf(s_r8[2], s_r8[3])), fmaxf(fmaxf(s_r8[4], s_r8[5]), fmaxf(s_r8[6], s_r8[7])));
        const float w = tid < a.nsplit ? fast_exp2(m - mx) : 0.f;
        if (tid < MLA_NSPLIT_MAX) s_w[tid] = w;
        float den = sum64(w * l);
        __syncthreads();                       // s_r8 read by everyone before it is rewritten; s_w visible
        if (lane == 0) s_r8[wave] = den;
        {   // x[k0 + 4 kk ..] = sum_s w_s O_s[h][..]: 16 thread groups stride the partials (group q: s = q, q + 16, ..),
            // 4 loads in flight per thread; the groups meet in a fixed order below
            const int kk = tid & 31, q = tid >> 5;
            const float* po = a.part_o + (size_t)h * MLA_L + k0 + 4 * kk;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int s0 = q; s0 < a.nsplit; s0 += 64) {
                f32x4 o[4];
                float ws[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int s = s0 + 16 * u;
                    const bool live = s < a.nsplit;
                    o[u] = ld_f4(po + (size_t)(live ? s : s0) * (MLA_H * MLA_L));
                    ws[u] = live ? s_w[s] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(ws[u], o[u][e], v[e]);
            }
            *reinterpret_cast<f32x4*>(&s_xp[q][4 * kk]) = v;
        }
        __syncthreads();
        if (tid < 128) {
            den = ((s_r8[0] + s_r8[1]) + (s_r8[2] + s_r8[3])) + ((s_r8[4] + s_r8[5]) + (s_r8[6] + s_r8[7]));
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) v += s_xp[q][tid];
            s_x[tid] = v / den;
        }
        __syncthreads();
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        t.fma(s_x + wave * 16, lane, acc);
        const float v = strip_reduce(acc, s_red, tid);
        if (tid < 64) mla_granule_store(a.g_d + (size_t)ks * (MLA_H * MLA_NOPE) + h * MLA_NOPE + 64 * c2 + tid, epoch, v);
        return;
    }
    // ---- E
    const int be = blockIdx.x - MLA_D_WGS;
    const int strip = be / MLA_E_KS, ks = be % MLA_E_KS;
    ColTile<4> t;
    t.load(a.w_o + (size_t)(ks * 256 + wave * 32) * MLA_HID + 64 * strip, MLA_HID, lane);
    const float xin = mla_granule_sum<MLA_D_KS>(a.g_d + ks * 256 + tid, MLA_H * MLA_NOPE, epoch, tid < 256, a.state + 1, 3u);
    if (tid < 256) s_x[tid] = xin;
    __syncthreads();
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    t.fma(s_x + wave * 32, lane, acc);
    const float v = strip_reduce(acc, s_red, tid);
    if (tid < 64) mla_granule_store(a.g_e + (size_t)ks * MLA_HID + 64 * strip + tid, epoch, v);
    // the strip's last K-slice sums all 8 in slice order (its producers have lower workgroup ids)
    if (ks == MLA_E_KS - 1 && tid < 64) {
        const float o = mla_granule_sum<MLA_E_KS>(a.g_e + 64 * strip + tid, MLA_HID, epoch, true, a.state + 1, 4u);
        a.out[64 * strip + tid] = (h16)o;
    }
}

}  // namespace cf
