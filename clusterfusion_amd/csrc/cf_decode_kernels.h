// cf_decode_kernels.h -- gfx950 kernels of the decode path (stage pipeline, v1).
//
// What the reference computes in ONE cluster-cooperative CUDA kernel
// (/root/reference/include/H100/llama/kernel.cuh:20-620, kernel_sglang.cuh, kernel_batch_sglang.cuh)
// is computed here by three (four for [in,out] weights) back-to-back kernels on one HIP stream:
//
//   stage 0  RMSNorm + QKV projection          k_qkv_rows  ([out,in])   | k_qkv_cols  ([in,out])
//            kernel.cuh:95-276                 -> raw q|k|v, fp32, (split-K partials for [in,out])
//   stage 1  RoPE + k/v export + split-KV flash-decode incl. the new token   k_attn_split
//            kernel.cuh:278-505                -> per (head, split) record (m, l, o[128]) fp32
//   stage 2  softmax merge of the split records k_attn_merge             kernel.cuh:507-568
//            + O projection                    k_oproj_rows ([out,in])  | k_oproj_cols ([in,out])
//            kernel.cuh:570-619                -> out fp16 (| per-head partials)
//   stage 3  ([in,out] only) cross-head sum    k_reduce_heads           (replaces fp16 atomicAdd,
//            kernel.cuh:600,618, by a fixed-order fp32 sum)
//
// Thread mappings are wave64-native: a wavefront instruction moves 1 KiB (64 lanes x 16 B):
// one 8 KiB weight row = 8 instructions ([out,in]), 512 output columns of one input row
// ([in,out]), or 4 token rows x one head's 256-B strip of the KV cache.
#pragma once
#include "cf_device.h"

namespace cf {

struct NormArgs {
    const h16* x;         // [batch, hidden]
    const h16* residual;  // nullable
    const h16* rms_w;     // [hidden]
    float eps;
    int hidden;
};

// ------------------------------------------------------------------------------------------------
// stage 0, [out,in] weights: every wavefront owns whole weight rows (no cross-workgroup reduce)
// ------------------------------------------------------------------------------------------------
template <int J>
__device__ __forceinline__ void load_norm_x(const NormArgs& na, int b, int lane, float (&xn)[J][8]) {
    const h16* x = na.x + (size_t)b * na.hidden;
    // no residual: read x twice and scale the second copy by 0 -- a branch around each load would
    // serialise them (one full wait per element)
    const h16* r = na.residual ? na.residual + (size_t)b * na.hidden : x;
    const float rs = na.residual ? 1.f : 0.f;
    h16x8 xv[J], rv[J], wv[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int idx = (j * WAVE + lane) * 8;
        xv[j] = ld_h8(x + idx);
        rv[j] = ld_h8(r + idx);
        wv[j] = ld_h8(na.rms_w + idx);
    }
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float h = __builtin_fmaf(rs, (float)rv[j][e], (float)xv[j][e]);
            xn[j][e] = h;
            ss = __builtin_fmaf(h, h, ss);
        }
    ss = sum64(ss);
    const float rcp = __builtin_amdgcn_rsqf(ss / (float)na.hidden + na.eps);
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) xn[j][e] = xn[j][e] * rcp * (float)wv[j][e];
}

template <int J, int R>
struct RowGroup {
    h16x8 w[R][J];
    __device__ __forceinline__ void load(const h16* W, int row0, int n_rows, int row_len, int lane) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int row = row0 + r;
            row = row < n_rows ? row : n_rows - 1;
            const h16* p = W + (size_t)row * row_len + lane * 8;
#pragma unroll
            for (int j = 0; j < J; ++j) w[r][j] = ld_stream(p + j * WAVE * 8);
        }
    }
    // activations already rounded to fp16 (the attention output, as the reference rounds it: kernel.cuh:553-559)
    __device__ __forceinline__ void dot_h(const h16x8 (&xh)[J], float (&res)[R]) const {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < J; ++j) acc = dot8h(w[r][j], xh[j], acc);
            res[r] = sum64_lane63(acc);
        }
    }
    __device__ __forceinline__ void dot(const float (&xn)[J][8], float (&res)[R]) const {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < J; ++j) acc = dot8(w[r][j], xn[j], acc);
            res[r] = sum64_lane63(acc);
        }
    }
};

template <int J, int R>
__global__ __launch_bounds__(256) void k_qkv_rows(NormArgs na, const h16* __restrict__ W, int n_rows,
                                                  int rows_per_wave, float* __restrict__ qkv_raw) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.y;
    const int row_base = (blockIdx.x * 4 + wave) * rows_per_wave;
    if (row_base >= n_rows) return;
    const int ngroups = (rows_per_wave + R - 1) / R;
    float xn[J][8];
    RowGroup<J, R> ga, gb;
    load_norm_x<J>(na, b, lane, xn);   // its loads are issued first, the weight stream right behind
    ga.load(W, row_base, n_rows, na.hidden, lane);
    float* dst = qkv_raw + (size_t)b * n_rows;
    for (int g = 0; g < ngroups; g += 2) {
        if (g + 1 < ngroups) gb.load(W, row_base + (g + 1) * R, n_rows, na.hidden, lane);
        float res[R];
        ga.dot(xn, res);
        if (lane == 63) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int row = row_base + g * R + r;
                if (row < n_rows && g * R + r < rows_per_wave) dst[row] = res[r];
            }
        }
        if (g + 2 < ngroups) ga.load(W, row_base + (g + 2) * R, n_rows, na.hidden, lane);
        if (g + 1 < ngroups) {
            gb.dot(xn, res);
            if (lane == 63) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int row = row_base + (g + 1) * R + r;
                    if (row < n_rows && (g + 1) * R + r < rows_per_wave) dst[row] = res[r];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// stage 0, [in,out] weights (chat/llama/model.py:317-320): W[(m*hidden + i), c], c = output column.
// Workgroup = (512-column strip, matrix m in q|k|v, K-split ks); the K-split partials are summed
// by stage 1 in a fixed order (replaces cluster_reduce<LINEAR>, dsm.cuh:20-134).
// ------------------------------------------------------------------------------------------------
constexpr int COLS_RK_MAX = 1024;

template <int UR>
struct ColGroup {
    h16x8 w[UR];
    __device__ __forceinline__ void load(const h16* p, size_t stride) {
#pragma unroll
        for (int u = 0; u < UR; ++u) w[u] = ld_stream(p + u * stride);
    }
    __device__ __forceinline__ void fma(const float* xs, float (&acc)[8]) const {
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const float xv = xs[u];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = __builtin_fmaf((float)w[u][e], xv, acc[e]);
        }
    }
};

// block-wide sum of squares of (x + residual) for batch row b; every thread returns rsqrt(mean+eps)
__device__ __forceinline__ float block_rms_rcp(const NormArgs& na, int b, float* s_ss /*[4]*/) {
    const int tid = threadIdx.x;
    const h16* x = na.x + (size_t)b * na.hidden;
    const h16* r = na.residual ? na.residual + (size_t)b * na.hidden : x;
    const float rs = na.residual ? 1.f : 0.f;
    float ss = 0.f;
    for (int i = tid * 8; i < na.hidden; i += 256 * 8) {
        const h16x8 xv = ld_h8(x + i), rv = ld_h8(r + i);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float h = __builtin_fmaf(rs, (float)rv[e], (float)xv[e]);
            ss = __builtin_fmaf(h, h, ss);
        }
    }
    ss = sum64(ss);
    if ((tid & 63) == 0) s_ss[tid >> 6] = ss;
    __syncthreads();
    ss = s_ss[0] + s_ss[1] + s_ss[2] + s_ss[3];
    return __builtin_amdgcn_rsqf(ss / (float)na.hidden + na.eps);
}

template <int UR>
__global__ __launch_bounds__(256) void k_qkv_cols(NormArgs na, const h16* __restrict__ W, int C, int ksplit,
                                                  float* __restrict__ qkv_raw) {
    __shared__ float s_xn[COLS_RK_MAX];
    __shared__ float s_red[4][512];
    __shared__ float s_ss[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.y;
    const int ncb = C / 512;
    const int cb = blockIdx.x % ncb, m = (blockIdx.x / ncb) % 3, ks = blockIdx.x / (3 * ncb);
    const int rk = na.hidden / ksplit;          // input rows of this workgroup
    const int rw = rk / 4;                      // ... of this wavefront
    const size_t stride = (size_t)C;
    const h16* wp = W + ((size_t)m * na.hidden + (size_t)ks * rk + (size_t)wave * rw) * stride + cb * 512 + lane * 8;

    const float rcp = block_rms_rcp(na, b, s_ss);
    ColGroup<UR> ga, gb;
    ga.load(wp, stride);
    {   // normalised activations of this K-slice -> LDS (fp32)
        const h16* x = na.x + (size_t)b * na.hidden + ks * rk;
        const h16* r = na.residual ? na.residual + (size_t)b * na.hidden + ks * rk : x;
        const float rs = na.residual ? 1.f : 0.f;
        const h16* w = na.rms_w + ks * rk;
        for (int i = tid * 8; i < rk; i += 256 * 8) {
            const h16x8 xv = ld_h8(x + i), wv = ld_h8(w + i), rv = ld_h8(r + i);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float h = __builtin_fmaf(rs, (float)rv[e], (float)xv[e]);
                s_xn[i + e] = h * rcp * (float)wv[e];
            }
        }
    }
    __syncthreads();
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const float* xs = s_xn + wave * rw;
    const int ngroups = rw / UR;
    for (int g = 0; g < ngroups; g += 2) {
        if (g + 1 < ngroups) gb.load(wp + (size_t)(g + 1) * UR * stride, stride);
        ga.fma(xs + g * UR, acc);
        if (g + 2 < ngroups) ga.load(wp + (size_t)(g + 2) * UR * stride, stride);
        if (g + 1 < ngroups) gb.fma(xs + (g + 1) * UR, acc);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) s_red[wave][lane * 8 + e] = acc[e];
    __syncthreads();
    float* dst = qkv_raw + ((size_t)b * ksplit + ks) * (3 * (size_t)C) + (size_t)m * C + cb * 512;
    for (int c = tid; c < 512; c += 256) dst[c] = (s_red[0][c] + s_red[1][c]) + (s_red[2][c] + s_red[3][c]);
}

// ------------------------------------------------------------------------------------------------
// stage 1: RoPE, k/v export, split-KV flash-decode
// ------------------------------------------------------------------------------------------------
struct AttnArgs {
    const float* qkv_raw;   // [batch][ksplit][qkv_dim]
    int ksplit, qkv_dim, Hq, Hkv;
    const h16* k_cache;
    const h16* v_cache;
    const uint64_t* kptrs;
    const uint64_t* vptrs;
    int layer_id;
    int seq_len;            // contiguous mode
    const int32_t* indptr;  // paged mode when non-null
    const int32_t* indices;
    const int32_t* seq_lens;
    int page_shift;
    const float* cos;
    const float* sin;
    const int64_t* positions;
    int64_t rope_stride;
    int rope_style;
    int nsplit, tokens_per_split;
    float* part_o;          // [batch][Hq][nsplit][128]
    float* part_ml;         // [batch][Hq][nsplit][2]
    float* attn_out;        // nsplit == 1: the normalised output is final -- written here directly (no merge launch)
    h16* attn_h16;          //   [batch][Hq*128] fp32 and (optional) fp16
    h16* k_new;             // [batch][Hkv][128] or null
    h16* v_new;
    int write_cache;
};

constexpr int ATTN_MAX_IDX = 2048;   // page-table entries one workgroup may stage (host guarantees)

// raw projection values for 8 consecutive dims, summed over the K-split partials (fixed order)
__device__ __forceinline__ void load_raw8(const float* raw, int ksplit, int qkv_dim, int off, float (&v)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    for (int s = 0; s < ksplit; ++s) {
        const f32x4 a = ld_f4(raw + (size_t)s * qkv_dim + off);
        const f32x4 c = ld_f4(raw + (size_t)s * qkv_dim + off + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += a[e]; v[4 + e] += c[e]; }
    }
}

// RoPE of dims [d0, d0+8) of one head (kernel.cuh:278-314 GPT-J; kernel_sglang.cuh:295-309 NEOX)
__device__ __forceinline__ void rope8(const float* raw, int ksplit, int qkv_dim, int head_off, int d0,
                                      const float* cosp, const float* sinp, int style, float (&out)[8]) {
    float self[8];
    load_raw8(raw, ksplit, qkv_dim, head_off + d0, self);
    if (style == 0) {   // NEOX: partner = d +- 64, angle index d % 64
        float part[8];
        load_raw8(raw, ksplit, qkv_dim, head_off + ((d0 + 64) & 127), part);
        const int a0 = d0 & 63;
        const f32x4 c0 = ld_f4(cosp + a0), c1 = ld_f4(cosp + a0 + 4);
        const f32x4 s0 = ld_f4(sinp + a0), s1 = ld_f4(sinp + a0 + 4);
        const float sgn = d0 < 64 ? -1.f : 1.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float c = e < 4 ? c0[e & 3] : c1[e & 3];
            const float s = e < 4 ? s0[e & 3] : s1[e & 3];
            out[e] = self[e] * c + sgn * (part[e] * s);
        }
    } else {            // GPT-J: pairs (2i, 2i+1), tables pair-duplicated over head_dim
        const f32x4 c0 = ld_f4(cosp + d0), c1 = ld_f4(cosp + d0 + 4);
        const f32x4 s0 = ld_f4(sinp + d0), s1 = ld_f4(sinp + d0 + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float c = e < 4 ? c0[e & 3] : c1[e & 3];
            const float s = e < 4 ? s0[e & 3] : s1[e & 3];
            out[e] = (e & 1) ? self[e] * c + self[e ^ 1] * s : self[e] * c - self[e ^ 1] * s;
        }
    }
}

template <int U>
struct KvTile {
    h16x8 k[U], v[U];
};

template <int G, int U>
__global__ __launch_bounds__(256) void k_attn_split(AttnArgs a) {
    __shared__ int s_idx[ATTN_MAX_IDX];
    __shared__ float s_o[G][17][HEAD_DIM];   // 16 lane-groups + the new token
    __shared__ float s_ml[G][17][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, gid = wave * 4 + (lane >> 4), d0 = l16 * 8;
    const int split = blockIdx.x / a.Hkv, kvh = blockIdx.x % a.Hkv, b = blockIdx.y;

    int S = a.seq_len, ent0 = 0;
    if (a.indptr) {
        ent0 = a.indptr[b];
        S = a.seq_lens ? a.seq_lens[b] : a.indptr[b + 1] - 1 - ent0;
    }
    const h16* kc = a.kptrs ? reinterpret_cast<const h16*>(a.kptrs[a.layer_id]) : a.k_cache;
    const h16* vc = a.vptrs ? reinterpret_cast<const h16*>(a.vptrs[a.layer_id]) : a.v_cache;
    const int ps = a.page_shift, pmask = (1 << ps) - 1;
    // tokens per split derive from THIS row's length (rows of a paged batch differ): >= 64,
    // multiple of 16 so a split starts on a page boundary for page sizes up to 16
    int tps = a.tokens_per_split;
    if (tps <= 0) {
        tps = ((S + a.nsplit - 1) / a.nsplit + 15) & ~15;
        tps = tps < 64 ? 64 : tps;
    }
    const int t0 = split * tps;
    int t1 = t0 + tps;
    t1 = t1 < S ? t1 : S;
    const int e0 = t0 >> ps;
    bool staged = false;   // page-table slice of this split staged in LDS (else read through L2)
    if (a.indptr && t1 > t0) {
        const int n = ((t1 - 1) >> ps) - e0 + 1;
        staged = n <= ATTN_MAX_IDX;
        if (staged)
            for (int i = tid; i < n; i += 256) s_idx[i] = a.indices[ent0 + e0 + i];
    }

    // q of this lane: G heads x dims [d0, d0+8), RoPE'd, pre-scaled by log2(e)/sqrt(d)
    const float* raw = a.qkv_raw + (size_t)b * a.ksplit * a.qkv_dim;
    const int64_t roff = a.positions ? a.positions[b] * a.rope_stride : 0;
    const float* cosp = a.cos + roff;
    const float* sinp = a.sin + roff;
    const float qscale = 1.44269504088896340736f * 0.08838834764831845f;   // log2(e) / sqrt(128)
    float q[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        rope8(raw, a.ksplit, a.qkv_dim, (kvh * G + g) * HEAD_DIM, d0, cosp, sinp, a.rope_style, q[g]);
#pragma unroll
        for (int e = 0; e < 8; ++e) q[g][e] *= qscale;
    }
    __syncthreads();

    const size_t kvstride = (size_t)a.Hkv * HEAD_DIM;
    const h16* kbase = kc + kvh * HEAD_DIM + d0;
    const h16* vbase = vc + kvh * HEAD_DIM + d0;
    float m[G], l[G], o[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        m[g] = NEG_BIG;
        l[g] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[g][e] = 0.f;
    }
    // Row (slot) numbers of a tile are resolved FIRST -- from the LDS-staged page-table slice, from
    // global memory, or directly -- in one wave-uniform branch, so the 2U streaming loads that
    // follow are issued back to back (a per-row "LDS or global" select would compile into flat
    // loads with a full wait in front of every row).
    auto load_tile = [&](KvTile<U>& t, int it) {
        size_t rows[U];
        int tok[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int tk = t0 + (it * U + j) * 16 + gid;
            tok[j] = tk < t1 ? tk : t1 - 1;
        }
        if (!a.indptr) {
#pragma unroll
            for (int j = 0; j < U; ++j) rows[j] = (size_t)tok[j];
        } else if (staged) {
#pragma unroll
            for (int j = 0; j < U; ++j)
                rows[j] = ((size_t)s_idx[(tok[j] >> ps) - e0] << ps) + (size_t)(tok[j] & pmask);
        } else {
            int ent[U];
#pragma unroll
            for (int j = 0; j < U; ++j) ent[j] = a.indices[ent0 + (tok[j] >> ps)];
#pragma unroll
            for (int j = 0; j < U; ++j) rows[j] = ((size_t)ent[j] << ps) + (size_t)(tok[j] & pmask);
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            t.k[j] = ld_stream(kbase + rows[j] * kvstride);
            t.v[j] = ld_stream(vbase + rows[j] * kvstride);
        }
    };
    auto compute_tile = [&](const KvTile<U>& t, int it) {
        bool valid[U];
#pragma unroll
        for (int j = 0; j < U; ++j) valid[j] = (t0 + (it * U + j) * 16 + gid) < t1;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float s[U];
            float mx = NEG_BIG;
#pragma unroll
            for (int j = 0; j < U; ++j) {
                s[j] = sum16(dot8(t.k[j], q[g], 0.f));
                s[j] = valid[j] ? s[j] : NEG_BIG;
                mx = fmaxf(mx, s[j]);
            }
            const float mnew = fmaxf(m[g], mx);
            const float alpha = fast_exp2(m[g] - mnew);
            float psum = 0.f;
#pragma unroll
            for (int j = 0; j < U; ++j) {
                s[j] = valid[j] ? fast_exp2(s[j] - mnew) : 0.f;
                psum += s[j];
            }
            l[g] = l[g] * alpha + psum;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float acc = o[g][e] * alpha;
#pragma unroll
                for (int j = 0; j < U; ++j) acc = __builtin_fmaf((float)t.v[j][e], s[j], acc);
                o[g][e] = acc;
            }
            m[g] = mnew;
        }
    };

    if (t1 > t0) {
        const int ntiles = (t1 - t0 + 16 * U - 1) / (16 * U);
        KvTile<U> ta, tb;
        load_tile(ta, 0);
        for (int it = 0; it < ntiles; it += 2) {
            if (it + 1 < ntiles) load_tile(tb, it + 1);
            compute_tile(ta, it);
            if (it + 2 < ntiles) load_tile(ta, it + 2);
            if (it + 1 < ntiles) compute_tile(tb, it + 1);
        }
    }

#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int e = 0; e < 8; ++e) s_o[g][gid][d0 + e] = o[g][e];
        if (l16 == 0) { s_ml[g][gid][0] = m[g]; s_ml[g][gid][1] = l[g]; }
    }

    // the new token: attended from registers, never from the cache (kernel.cuh:444-477); split 0
    // also exports k (post-RoPE) / v, and in paged mode stores them into the new token's slot
    // (kernel_batch_sglang.cuh:343-344)
    if (split == 0 && gid == 0) {
        const int kq = a.Hq * HEAD_DIM, kk = a.Hkv * HEAD_DIM;
        float kf[8], vf[8];
        rope8(raw, a.ksplit, a.qkv_dim, kq + kvh * HEAD_DIM, d0, cosp, sinp, a.rope_style, kf);
        load_raw8(raw, a.ksplit, a.qkv_dim, kq + kk + kvh * HEAD_DIM + d0, vf);
        h16x8 k16, v16;
#pragma unroll
        for (int e = 0; e < 8; ++e) { k16[e] = (h16)kf[e]; v16[e] = (h16)vf[e]; }
        const size_t ooff = ((size_t)b * a.Hkv + kvh) * HEAD_DIM + d0;
        if (a.k_new) st_h8(a.k_new + ooff, k16);
        if (a.v_new) st_h8(a.v_new + ooff, v16);
        if (a.indptr && a.write_cache) {
            const size_t slot = ((size_t)a.indices[ent0 + (S >> ps)] << ps) + (size_t)(S & pmask);
            st_h8(const_cast<h16*>(kc) + slot * kvstride + kvh * HEAD_DIM + d0, k16);
            st_h8(const_cast<h16*>(vc) + slot * kvstride + kvh * HEAD_DIM + d0, v16);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float sn = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) sn = __builtin_fmaf(q[g][e], kf[e], sn);
            sn = sum16(sn);
#pragma unroll
            for (int e = 0; e < 8; ++e) s_o[g][16][d0 + e] = vf[e];
            if (l16 == 0) { s_ml[g][16][0] = sn; s_ml[g][16][1] = 1.f; }
        }
    }
    __syncthreads();

    // merge the 16 (+1) lane-group states of this workgroup -> one record per (head, split)
    const int nst = split == 0 ? 17 : 16;
    for (int idx = tid; idx < G * HEAD_DIM; idx += 256) {
        const int g = idx >> 7, d = idx & 127;
        float M = NEG_BIG;
        for (int i = 0; i < nst; ++i) M = fmaxf(M, s_ml[g][i][0]);
        float acc = 0.f, L = 0.f;
        for (int i = 0; i < nst; ++i) {
            const float w = fast_exp2(s_ml[g][i][0] - M);
            acc = __builtin_fmaf(w, s_o[g][i][d], acc);
            L = __builtin_fmaf(w, s_ml[g][i][1], L);
        }
        if (a.nsplit == 1) {       // the only record of this head: already the merged result (k_attn_merge's arithmetic)
            const size_t at = ((size_t)b * a.Hq + kvh * G + g) * HEAD_DIM + d;
            a.attn_out[at] = acc / L;
            if (a.attn_h16) a.attn_h16[at] = (h16)(acc / L);
            continue;
        }
        const size_t rec = ((size_t)b * a.Hq + kvh * G + g) * a.nsplit + split;
        a.part_o[rec * HEAD_DIM + d] = acc;
        if (d == 0) { a.part_ml[rec * 2] = M; a.part_ml[rec * 2 + 1] = L; }
    }
}

// ------------------------------------------------------------------------------------------------
// stage 2: merge the split records (kernel.cuh:507-568) + O projection
// ------------------------------------------------------------------------------------------------
struct MergeArgs {
    const float* part_o;
    const float* part_ml;
    int nsplit, Hq;
    h16* attn_h16;   // optional: the output rounded to fp16 as well (operand of the batched MFMA O projection)
};
struct ResidualOut {
    const h16* x;
    const h16* residual;
    h16* residual_out;   // may alias residual; written by ONE workgroup of the last stage only
    int hidden;
};

// One thread per (head, dim): merge the nsplit records of the head (kernel.cuh:507-568's cluster
// max / sum / vector all-reduces) into the normalised attention output a[b][h*128 + d], fp32.
// All loads of a 16-record chunk are issued before the first use (one L2 round trip per chunk).
__global__ __launch_bounds__(256) void k_attn_merge(MergeArgs ma, float* __restrict__ attn_out) {
    const int b = blockIdx.y, idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= ma.Hq * HEAD_DIM) return;
    const int h = idx >> 7, d = idx & 127;
    const size_t base = ((size_t)b * ma.Hq + h) * ma.nsplit;
    const float* __restrict__ pml = ma.part_ml + base * 2;
    const float* __restrict__ po = ma.part_o + base * HEAD_DIM + d;
    float M = NEG_BIG, acc = 0.f, L = 0.f;
    for (int s0 = 0; s0 < ma.nsplit; s0 += 16) {
        float mv[16], lv[16], ov[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int s = s0 + i < ma.nsplit ? s0 + i : ma.nsplit - 1;
            mv[i] = pml[s * 2];
            lv[i] = pml[s * 2 + 1];
            ov[i] = po[(size_t)s * HEAD_DIM];
        }
        float mc = M;
#pragma unroll
        for (int i = 0; i < 16; ++i) mc = fmaxf(mc, s0 + i < ma.nsplit ? mv[i] : NEG_BIG);
        const float rescale = fast_exp2(M - mc);
        acc *= rescale;
        L *= rescale;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float w = s0 + i < ma.nsplit ? fast_exp2(mv[i] - mc) : 0.f;
            acc = __builtin_fmaf(w, ov[i], acc);
            L = __builtin_fmaf(w, lv[i], L);
        }
        M = mc;
    }
    attn_out[(size_t)b * ma.Hq * HEAD_DIM + idx] = acc / L;
    if (ma.attn_h16) ma.attn_h16[(size_t)b * ma.Hq * HEAD_DIM + idx] = (h16)(acc / L);
}

__device__ __forceinline__ void write_residual(const ResidualOut& ro, int b) {
    if (!ro.residual_out) return;
    const size_t off = (size_t)b * ro.hidden;
    for (int i = threadIdx.x * 8; i < ro.hidden; i += 256 * 8) {
        h16x8 xv = ld_h8(ro.x + off + i), rv = ld_h8(ro.residual + off + i), hv;
#pragma unroll
        for (int e = 0; e < 8; ++e) hv[e] = (h16)((float)xv[e] + (float)rv[e]);
        *reinterpret_cast<h16x8*>(ro.residual_out + off + i) = hv;
    }
}

template <int J, int R>
__global__ __launch_bounds__(256) void k_oproj_rows(const float* __restrict__ attn, const h16* __restrict__ Wo,
                                                    int n_rows, int rows_per_wave, h16* __restrict__ out,
                                                    ResidualOut ro) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.y;
    constexpr int L = J * 512;
    const int row_base = (blockIdx.x * 4 + wave) * rows_per_wave;
    const int ngroups = (rows_per_wave + R - 1) / R;
    if (blockIdx.x == 0) write_residual(ro, b);
    if (row_base >= n_rows) return;
    RowGroup<J, R> ga, gb;
    float av[J][8];
    const float* ap = attn + (size_t)b * L + lane * 8;
    f32x4 a0[J], a1[J];
#pragma unroll
    for (int j = 0; j < J; ++j) { a0[j] = ld_f4(ap + j * 512); a1[j] = ld_f4(ap + j * 512 + 4); }
    ga.load(Wo, row_base, n_rows, L, lane);
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) { av[j][e] = a0[j][e]; av[j][4 + e] = a1[j][e]; }
    h16* dst = out + (size_t)b * n_rows;
    for (int g = 0; g < ngroups; g += 2) {
        if (g + 1 < ngroups) gb.load(Wo, row_base + (g + 1) * R, n_rows, L, lane);
        float res[R];
        ga.dot(av, res);
        if (lane == 63) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int row = row_base + g * R + r;
                if (row < n_rows && g * R + r < rows_per_wave) dst[row] = (h16)res[r];
            }
        }
        if (g + 2 < ngroups) ga.load(Wo, row_base + (g + 2) * R, n_rows, L, lane);
        if (g + 1 < ngroups) {
            gb.dot(av, res);
            if (lane == 63) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int row = row_base + (g + 1) * R + r;
                    if (row < n_rows && (g + 1) * R + r < rows_per_wave) dst[row] = (h16)res[r];
                }
            }
        }
    }
}

// [in,out] O projection: Wo[(h*128 + d), n].  Workgroup = (512-column strip, head h) -> per-head
// partial outputs, summed over heads by k_reduce_heads in a fixed order.
template <int UR>
__global__ __launch_bounds__(256) void k_oproj_cols(const float* __restrict__ attn, int Hq,
                                                    const h16* __restrict__ Wo, int hidden,
                                                    float* __restrict__ opart) {
    __shared__ float s_a[HEAD_DIM];
    __shared__ float s_red[4][512];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.y;
    const int ncb = hidden / 512;
    const int cb = blockIdx.x % ncb, h = blockIdx.x / ncb;
    constexpr int rw = HEAD_DIM / 4;   // 32 input rows per wavefront
    const size_t stride = (size_t)hidden;
    const h16* wp = Wo + ((size_t)h * HEAD_DIM + wave * rw) * stride + cb * 512 + lane * 8;
    ColGroup<UR> ga, gb;
    ga.load(wp, stride);
    if (tid < HEAD_DIM) s_a[tid] = attn[((size_t)b * Hq + h) * HEAD_DIM + tid];
    __syncthreads();
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const float* xs = s_a + wave * rw;
    constexpr int ngroups = rw / UR;
#pragma unroll
    for (int g = 0; g < ngroups; g += 2) {
        if (g + 1 < ngroups) gb.load(wp + (size_t)(g + 1) * UR * stride, stride);
        ga.fma(xs + g * UR, acc);
        if (g + 2 < ngroups) ga.load(wp + (size_t)(g + 2) * UR * stride, stride);
        if (g + 1 < ngroups) gb.fma(xs + (g + 1) * UR, acc);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) s_red[wave][lane * 8 + e] = acc[e];
    __syncthreads();
    float* dst = opart + ((size_t)b * Hq + h) * hidden + cb * 512;
    for (int c = tid; c < 512; c += 256) dst[c] = (s_red[0][c] + s_red[1][c]) + (s_red[2][c] + s_red[3][c]);
}

__global__ __launch_bounds__(256) void k_reduce_heads(const float* __restrict__ opart, int Hq, int hidden,
                                                      h16* __restrict__ out, ResidualOut ro) {
    const int b = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x == 0) write_residual(ro, b);
    if (n >= hidden) return;
    const float* p = opart + (size_t)b * Hq * hidden + n;
    float acc = 0.f;
    for (int h = 0; h < Hq; ++h) acc += p[(size_t)h * hidden];
    out[(size_t)b * hidden + n] = (h16)acc;
}

}  // namespace cf
