// cf_fused_kernel_g.h -- the persistent fused decode kernel generalised over the head geometry:
// HKV local kv heads, G = q heads per kv head ([out,in] weights, hidden 4096, batch 1).
//
//   <8, 4>   Llama-3-8B (32 q / 8 kv heads, BASELINE config 4): 32 workgroups per kv head
//   <16, 1>  one rank of a 2-way head-parallel shard of Llama-2-7B: 16 workgroups per head
//   <8, 1>   ... of a 4-way shard: 32 workgroups per head
//   <4, 1>   ... of an 8-way shard (BASELINE config 5): 64 workgroups per head (a head spans 2 XCDs)
//
// Same structure as k_fused_decode_mha (cf_fused_kernel.h): 256 co-resident workgroups, three granule
// exchanges, K/V and Wo requested ahead of the exchanges.  What changes with the geometry:
//   * a kv head's (G+2)*128 projection rows (its G q heads, k, v) are spread over NS = 256/HKV
//     workgroups, 3 rows per wavefront;
//   * every cached K/V row is streamed once and scored against the G q heads that share it
//     (the reference has no GQA path; chat/llama/model.py:166-175 repeat_kv is the eager definition);
//   * X2 has G leaders per kv head (workgroup j < G merges q head g*G + j over its NS records).
#pragma once
#include "cf_fused_kernel.h"

namespace cf {

template <int HKV, int G>
struct FusedGeom {
    static constexpr int HQ = HKV * G;
    static constexpr int NS = FUSED_WGS / HKV;            // workgroups per kv head
    static constexpr int RG = (G + 2) * HEAD_DIM;         // projection rows of one kv-head group
    static constexpr int RPW = RG / NS;                   // ... per workgroup
    static constexpr int P1_WAVES = RPW / 3;              // wavefronts that stream projection rows (3 each)
    static constexpr int U = G == 4 ? 8 : (NS <= 16 ? 8 : (NS == 32 ? 4 : 2));   // token rows per lane-group of tile A
    static constexpr int JO = HQ * HEAD_DIM / 512;        // 1-KB pieces of one Wo row
    static constexpr int RECW = NS / 8;                   // records one wavefront of a leader sweeps
    // LDS carve
    static constexpr int L_QKV = 0;                                    // float[RG]
    static constexpr int L_A = L_QKV + RG * 4;                         // float[4096] (x, then attention out)
    static constexpr int L_O = L_A + 4096 * 4;                         // float[G][9][128]
    static constexpr int L_ML = L_O + G * 9 * HEAD_DIM * 4;            // float[G][9][2] (+pad)
    static constexpr int L_REC = L_ML + ((G * 9 * 2 * 4 + 15) & ~15);  // float[NS][FUSED_REC]
    static constexpr int L_IDX = L_REC + NS * FUSED_REC * 4;           // int[FUSED_MAX_IDX]
    static constexpr int L_CS = L_IDX + FUSED_MAX_IDX * 4;             // float[256]
    static constexpr int L_CTL = L_CS + 256 * 4;                       // int[32]
    static constexpr int L_END = L_CTL + 128;
    static constexpr int LDS_BYTES = L_END > 84 * 1024 ? L_END : 84 * 1024;
    static_assert(RPW % 3 == 0 && P1_WAVES >= 1 && P1_WAVES <= 8, "3 projection rows per streaming wavefront");
};

template <int HKV, int G, bool LONG>
__global__ __launch_bounds__(512, 2) void k_fused_decode_g(FusedArgs a) {
    using GM = FusedGeom<HKV, G>;
    constexpr int HQ = GM::HQ, NS = GM::NS, RG = GM::RG, JO = GM::JO, HID = 4096, U = GM::U;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_qkv = reinterpret_cast<float*>(smem + GM::L_QKV);
    float* s_a = reinterpret_cast<float*>(smem + GM::L_A);
    float(*s_o)[9][HEAD_DIM] = reinterpret_cast<float(*)[9][HEAD_DIM]>(smem + GM::L_O);
    float(*s_ml)[9][2] = reinterpret_cast<float(*)[9][2]>(smem + GM::L_ML);
    float* s_rec = reinterpret_cast<float*>(smem + GM::L_REC);
    int* s_idx = reinterpret_cast<int*>(smem + GM::L_IDX);
    float* s_cs = reinterpret_cast<float*>(smem + GM::L_CS);
    int* s_ctl = reinterpret_cast<int*>(smem + GM::L_CTL);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, gid = wave * 4 + (lane >> 4), d0 = l16 * 8;
    const int b = blockIdx.x;
    // the NS workgroups of a kv head share b % 8 (one XCD hosts 32 / NS whole groups); with 64
    // workgroups per head a head spans the XCD pair (2p, 2p+1)
    const int g = NS <= 32 ? (b & 7) * (32 / (NS <= 32 ? NS : 32)) + (b >> 3) / NS : (b & 7) >> 1;
    const int j = NS <= 32 ? (b >> 3) % NS : (b >> 3) + 32 * (b & 1);
    CF_TRACE(0);

    // ---- small first-level loads first (loads return in issue order) -------------------------------
    const h16* rp = a.na.residual ? a.na.residual : a.na.x;
    const float rs = a.na.residual ? 1.f : 0.f;
    const h16x8 xv = ld_h8(a.na.x + tid * 8), rv = ld_h8(rp + tid * 8), wv8 = ld_h8(a.na.rms_w + tid * 8);
    const unsigned epoch = a.state[0] + 1u;
    int S = a.seq_len, ent0 = 0;
    if (a.indptr) {
        ent0 = a.indptr[0];
        S = a.seq_lens ? a.seq_lens[0] : a.indptr[1] - 1 - ent0;
    }
    const int64_t roff = a.positions ? a.positions[0] * a.rope_stride : 0;
    const h16* kc = a.kptrs ? reinterpret_cast<const h16*>(a.kptrs[a.layer_id]) : a.k_cache;
    const h16* vc = a.vptrs ? reinterpret_cast<const h16*>(a.vptrs[a.layer_id]) : a.v_cache;

    // ---- phase-1 weight stream: 3 rows of the group's q|k|v row space per wavefront -----------------
    const int rr0 = GM::RPW * j + 3 * wave;          // first row (inside the group) of this wavefront
    auto global_row = [&](int rr) -> int {           // Wqkv rows: q of all heads | k | v
        if (rr < G * HEAD_DIM) return g * G * HEAD_DIM + rr;
        if (rr < (G + 1) * HEAD_DIM) return HQ * HEAD_DIM + g * HEAD_DIM + (rr - G * HEAD_DIM);
        return (HQ + HKV) * HEAD_DIM + g * HEAD_DIM + (rr - (G + 1) * HEAD_DIM);
    };
    constexpr int NROWS = (HQ + 2 * HKV) * HEAD_DIM;
    RowGroup<8, 1> r0, r1, r2;
    const bool p1w = __builtin_amdgcn_readfirstlane(tid >> 6) < GM::P1_WAVES;   // small shards: few rows per workgroup
    if (p1w) {
        r0.load(a.Wqkv, global_row(rr0), NROWS, HID, lane);
        r1.load(a.Wqkv, global_row(rr0 + 1), NROWS, HID, lane);
        r2.load(a.Wqkv, global_row(rr0 + 2), NROWS, HID, lane);
    }

    // ---- RMSNorm once per workgroup ------------------------------------------------------------------
    float hx[8];
    {
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            hx[e] = __builtin_fmaf(rs, (float)rv[e], (float)xv[e]);
            ss = __builtin_fmaf(hx[e], hx[e], ss);
        }
        ss = sum64_lane63(ss);
        if (lane == 63) s_rec[wave] = ss;     // s_rec is free until X2
    }

    // ---- second-level loads: registers now, LDS after the first rows have been consumed ---------------
    const int ps = a.page_shift, pmask = (1 << ps) - 1;
    int tps = ((S + NS - 1) / NS + 31) & ~31;
    tps = tps < 32 ? 32 : tps;
    const int t0 = j * tps;
    int t1 = t0 + tps;
    t1 = t1 < S ? t1 : S;
    const int e0 = t0 >> ps;
    int n_idx = 0;
    if (a.indptr && t1 > t0) {
        n_idx = ((t1 - 1) >> ps) - e0 + 1;
        if (n_idx > FUSED_MAX_IDX) {
            if (tid == 0) atomicCAS(a.state + 1, 0u, 4u);
            n_idx = FUSED_MAX_IDX;
        }
    }
    int idx_reg = 0, slot_reg = 0;
    if (tid < n_idx) idx_reg = a.indices[ent0 + e0 + tid];
    if (a.indptr && tid == 0) slot_reg = a.indices[ent0 + (S >> ps)];
    float cs_reg = 0.f;
    {
        const int n_ang = a.rope_style == 0 ? HEAD_DIM / 2 : HEAD_DIM;
        if (tid < n_ang) cs_reg = a.cos[roff + tid];
        else if (tid >= 128 && tid < 128 + n_ang) cs_reg = a.sin[roff + tid - 128];
    }

    lds_barrier();
    float xn[8][8];
    {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) tot += s_rec[w];
        const float rcp = __builtin_amdgcn_rsqf(tot / (float)HID + a.na.eps);
        float* s_xn = s_a;
        f32x4 lo, hi;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            lo[e] = hx[e] * rcp * (float)wv8[e];
            hi[e] = hx[4 + e] * rcp * (float)wv8[4 + e];
        }
        *reinterpret_cast<f32x4*>(&s_xn[tid * 8]) = lo;
        *reinterpret_cast<f32x4*>(&s_xn[tid * 8 + 4]) = hi;
        lds_barrier();
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const f32x4 p0 = *reinterpret_cast<const f32x4*>(&s_xn[(jj * WAVE + lane) * 8]);
            const f32x4 p1 = *reinterpret_cast<const f32x4*>(&s_xn[(jj * WAVE + lane) * 8 + 4]);
#pragma unroll
            for (int e = 0; e < 4; ++e) { xn[jj][e] = p0[e]; xn[jj][4 + e] = p1[e]; }
        }
    }

    // ---- phase 1 --------------------------------------------------------------------------------------
    u64* gq = a.g_qkv + (size_t)g * RG + rr0;
    if (p1w) {
        float res[1];
        r0.dot(xn, res);
        if (lane == 63) granule_store(gq, epoch, res[0]);
        r1.dot(xn, res);
        if (lane == 63) granule_store(gq + 1, epoch, res[0]);
    }
    if (tid < n_idx) s_idx[tid] = idx_reg;
    for (int i = tid + 512; i < n_idx; i += 512) s_idx[i] = a.indices[ent0 + e0 + i];
    if (tid < 256) s_cs[tid] = cs_reg;
    if (tid == 0) s_ctl[20] = slot_reg;
    lds_barrier();

    // ---- K/V tiles requested before q exists ----------------------------------------------------------
    const size_t kvstride = (size_t)HKV * HEAD_DIM;
    const h16* kbase = kc + g * HEAD_DIM + d0;
    const h16* vbase = vc + g * HEAD_DIM + d0;
    auto load_tile = [&](auto& t, int tbase) {
        constexpr int UU = sizeof(t.k) / sizeof(h16x8);
        size_t rows[UU];
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            int tk = tbase + u * 32 + gid;
            tk = tk < t1 ? tk : t1 - 1;
            if (!a.indptr) {
                rows[u] = (size_t)tk;
            } else {
                int ei = (tk >> ps) - e0;
                ei = ei < FUSED_MAX_IDX ? ei : FUSED_MAX_IDX - 1;
                rows[u] = ((size_t)s_idx[ei] << ps) + (size_t)(tk & pmask);
            }
        }
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            t.k[u] = ld_stream(kbase + rows[u] * kvstride);
            t.v[u] = ld_stream(vbase + rows[u] * kvstride);
        }
    };
    constexpr int TILE = 32 * U;
    constexpr int UL = 4, TILE_L = 32 * UL;
    const int ntiles = t1 > t0 ? (t1 - t0 + TILE - 1) / TILE : 0;
    KvTile32<U> ta;
    if (ntiles > 0) load_tile(ta, t0);
    if (p1w) {
        float res[1];
        r2.dot(xn, res);
        if (lane == 63) granule_store(gq + 2, epoch, res[0]);
    }

    CF_TRACE(1);
    // ---- X1: q (G heads) | k | v of this kv-head group --------------------------------------------------
    if (wave == 0) {
        const bool ok = sweep_granules<RG / 64>(a.g_qkv + (size_t)g * RG, RG, epoch, s_qkv, lane, a.state + 1, 1u);
        if (lane == 0) s_ctl[0] = ok;
    }
    lds_barrier();
    if (!s_ctl[0]) return;
    CF_TRACE(2);

    // ---- RoPE(q) for the G heads ----------------------------------------------------------------------
    const float qscale = 1.44269504088896340736f * 0.08838834764831845f;
    auto rope_lds = [&](const float* src, float (&dst)[8]) {
        if (a.rope_style == 0) {
            const float sgn = d0 < 64 ? -1.f : 1.f;
            const int a0 = d0 & 63, p0 = (d0 + 64) & 127;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                dst[e] = src[d0 + e] * s_cs[a0 + e] + sgn * (src[p0 + e] * s_cs[128 + a0 + e]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float c = s_cs[d0 + e], s = s_cs[128 + d0 + e];
                dst[e] = (e & 1) ? src[d0 + e] * c + src[d0 + (e ^ 1)] * s : src[d0 + e] * c - src[d0 + (e ^ 1)] * s;
            }
        }
    };
    float q[G][8];
#pragma unroll
    for (int hh = 0; hh < G; ++hh) {
        rope_lds(s_qkv + hh * HEAD_DIM, q[hh]);
#pragma unroll
        for (int e = 0; e < 8; ++e) q[hh][e] *= qscale;
    }

    // ---- phase 2: every K/V row is scored against the G q heads of its group ---------------------------
    float m[G], l[G], o[G][8];
#pragma unroll
    for (int hh = 0; hh < G; ++hh) {
        m[hh] = NEG_BIG;
        l[hh] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[hh][e] = 0.f;
    }
    auto compute_tile = [&](const auto& t, int tbase) {
        constexpr int UU = sizeof(t.k) / sizeof(h16x8);
        bool valid[UU];
#pragma unroll
        for (int u = 0; u < UU; ++u) valid[u] = (tbase + u * 32 + gid) < t1;
#pragma unroll
        for (int hh = 0; hh < G; ++hh) {
            float s[UU];
            float mx = NEG_BIG;
#pragma unroll
            for (int u = 0; u < UU; ++u) {
                s[u] = sum16(dot8(t.k[u], q[hh], 0.f));
                s[u] = valid[u] ? s[u] : NEG_BIG;
                mx = fmaxf(mx, s[u]);
            }
            const float mnew = fmaxf(m[hh], mx);
            const float alpha = fast_exp2(m[hh] - mnew);
            float psum = 0.f;
#pragma unroll
            for (int u = 0; u < UU; ++u) {
                s[u] = valid[u] ? fast_exp2(s[u] - mnew) : 0.f;
                psum += s[u];
            }
            l[hh] = l[hh] * alpha + psum;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float acc = o[hh][e] * alpha;
#pragma unroll
                for (int u = 0; u < UU; ++u) acc = __builtin_fmaf((float)t.v[u][e], s[u], acc);
                o[hh][e] = acc;
            }
            m[hh] = mnew;
        }
    };
    RowGroup<JO, 2> go;
    constexpr int LO = HQ * HEAD_DIM;
    CF_TRACE(7);
    if (ntiles > 0) compute_tile(ta, t0);
    CF_TRACE(8);
    if constexpr (LONG) {
        KvTile32<UL> la, lb;
        const int tl = t0 + TILE;
        if (tl < t1) load_tile(la, tl);
        for (int tt = tl; tt < t1; tt += 2 * TILE_L) {
            if (tt + TILE_L < t1) load_tile(lb, tt + TILE_L);
            compute_tile(la, tt);
            if (tt + 2 * TILE_L < t1) load_tile(la, tt + 2 * TILE_L);
            if (tt + TILE_L < t1) compute_tile(lb, tt + TILE_L);
        }
    }
    go.load(a.Wo, 16 * b + 2 * wave, HID, LO, lane);   // phase-3 rows: in flight through X2 / X3
    CF_TRACE(9);

    // merge the 4 lane-groups of a wavefront in registers, then 8 wavefront states (+ new token) in LDS
#pragma unroll
    for (int hh = 0; hh < G; ++hh) {
        const float mw = xmax32(xmax16(m[hh]));
        const float sc = fast_exp2(m[hh] - mw);
        const float lw = xsum32(xsum16(l[hh] * sc));
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = xsum32(xsum16(o[hh][e] * sc));
            if (lane < 16) s_o[hh][wave][d0 + e] = v;
        }
        if (lane == 0) { s_ml[hh][wave][0] = mw; s_ml[hh][wave][1] = lw; }
    }
    // the new token + k/v export: split 0 of the group
    if (j == 0 && gid == 0) {
        float kf[8], vf[8];
        rope_lds(s_qkv + G * HEAD_DIM, kf);
#pragma unroll
        for (int e = 0; e < 8; ++e) vf[e] = s_qkv[(G + 1) * HEAD_DIM + d0 + e];
        h16x8 k16, v16;
#pragma unroll
        for (int e = 0; e < 8; ++e) { k16[e] = (h16)kf[e]; v16[e] = (h16)vf[e]; }
        const size_t ooff = (size_t)g * HEAD_DIM + d0;
        if (a.k_new) st_h8(a.k_new + ooff, k16);
        if (a.v_new) st_h8(a.v_new + ooff, v16);
        if (a.indptr && a.write_cache) {
            const size_t slot = ((size_t)s_ctl[20] << ps) + (size_t)(S & pmask);
            st_h8(const_cast<h16*>(kc) + slot * kvstride + ooff, k16);
            st_h8(const_cast<h16*>(vc) + slot * kvstride + ooff, v16);
        }
#pragma unroll
        for (int hh = 0; hh < G; ++hh) {
            float sn = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) sn = __builtin_fmaf(q[hh][e], kf[e], sn);
            sn = sum16(sn);
#pragma unroll
            for (int e = 0; e < 8; ++e) s_o[hh][8][d0 + e] = vf[e];
            if (l16 == 0) { s_ml[hh][8][0] = sn; s_ml[hh][8][1] = 1.f; }
        }
    }
    CF_TRACE(12);
    lds_barrier();
    CF_TRACE(3);

    // ---- X2: G records per workgroup -> the q head's leader --------------------------------------------
    for (int t = tid; t < G * FUSED_REC; t += 512) {
        const int hh = t / FUSED_REC, i = t - hh * FUSED_REC;
        const int nst = j == 0 ? 9 : 8;
        float M = NEG_BIG;
#pragma unroll
        for (int w = 0; w < 9; ++w) M = fmaxf(M, w < nst ? s_ml[hh][w][0] : NEG_BIG);
        float val = 0.f;
        if (i < HEAD_DIM) {
#pragma unroll
            for (int w = 0; w < 9; ++w)
                if (w < nst) val = __builtin_fmaf(fast_exp2(s_ml[hh][w][0] - M), s_o[hh][w][i], val);
        } else if (i == HEAD_DIM) {
            val = M;
        } else if (i == HEAD_DIM + 1) {
#pragma unroll
            for (int w = 0; w < 9; ++w)
                if (w < nst) val = __builtin_fmaf(fast_exp2(s_ml[hh][w][0] - M), s_ml[hh][w][1], val);
        }
        granule_store(a.g_rec + (((size_t)g * G + hh) * NS + j) * FUSED_REC + i, epoch, val);
    }
    if (j < G) {   // leader of q head g*G + j: wavefront w gathers NS/8 records, then the softmax merge
        constexpr int CNT = GM::RECW * FUSED_REC;
        const bool ok = sweep_granules<(CNT + 63) / 64>(a.g_rec + (((size_t)g * G + j) * NS + wave * GM::RECW) * FUSED_REC,
                                                       CNT, epoch, s_rec + wave * CNT, lane, a.state + 1, 2u);
        if (lane == 0) s_ctl[1 + wave] = ok;
        lds_barrier();
        bool all_ok = true;
        for (int w = 0; w < 8; ++w) all_ok &= s_ctl[1 + w] != 0;
        if (!all_ok) return;
        if (tid < HEAD_DIM) {
            float M = NEG_BIG;
            for (int w = 0; w < NS; ++w) M = fmaxf(M, s_rec[w * FUSED_REC + HEAD_DIM]);
            float acc = 0.f, L = 0.f;
            for (int w = 0; w < NS; ++w) {
                const float wt = fast_exp2(s_rec[w * FUSED_REC + HEAD_DIM] - M);
                acc = __builtin_fmaf(wt, s_rec[w * FUSED_REC + tid], acc);
                L = __builtin_fmaf(wt, s_rec[w * FUSED_REC + HEAD_DIM + 1], L);
            }
            granule_store(a.g_attn + ((size_t)g * G + j) * HEAD_DIM + tid, epoch, acc / L);
        }
    }

    CF_TRACE(4);
    // ---- X3: every workgroup gathers the full attention output -------------------------------------------
    {
        constexpr int PER = HQ * HEAD_DIM / 8;
        const bool ok = sweep_granules<PER / 64>(a.g_attn + wave * PER, PER, epoch, s_a + wave * PER, lane, a.state + 1, 3u);
        if (lane == 0) s_ctl[9 + wave] = ok;
    }
    lds_barrier();
    {
        bool all_ok = true;
        for (int w = 0; w < 8; ++w) all_ok &= s_ctl[9 + w] != 0;
        if (!all_ok) return;
    }

    CF_TRACE(5);
    // ---- phase 3: 16 rows of Wo per workgroup -------------------------------------------------------------
    float av[JO][8];
#pragma unroll
    for (int jj = 0; jj < JO; ++jj) {
        const f32x4 p0 = *reinterpret_cast<const f32x4*>(&s_a[(jj * WAVE + lane) * 8]);
        const f32x4 p1 = *reinterpret_cast<const f32x4*>(&s_a[(jj * WAVE + lane) * 8 + 4]);
#pragma unroll
        for (int e = 0; e < 4; ++e) { av[jj][e] = p0[e]; av[jj][4 + e] = p1[e]; }
    }
    {
        float res[2];
        go.dot(av, res);
        if (lane == 63) {
            a.out[16 * b + 2 * wave] = (h16)res[0];
            a.out[16 * b + 2 * wave + 1] = (h16)res[1];
        }
    }
    if (a.residual_out && tid < 16) {
        const int i = 16 * b + tid;
        a.residual_out[i] = (h16)((float)a.na.x[i] + (float)a.na.residual[i]);
    }
    if (b == 0 && tid == 0) a.state[0] = epoch;
    CF_TRACE(6);
}

}  // namespace cf
