// cf_fused_kernel_b.h -- the persistent [out,in] MHA decode layer for a SMALL BATCH (2 .. 4 sequences) in one launch (gfx950).
//
// `llama_decoder_layer_batch_decode_sglang` with more than one sequence (reference: kernel_batch_sglang.cuh:63-64 runs its
// whole GEMV kernel once per sequence, re-reading every weight).  From 16 rows up the projections are real GEMMs and run on
// the matrix cores (cf_batch_kernels.h, five launches); for 2 .. 4 rows those five launches are mostly launch boundaries and
// latency (47 / 54 us for 2 / 4 rows against 32 us for one).  Here the B rows ride ONE weight stream of the persistent kernel:
//   * phase 1: every row pair of Wqkv is dotted with the B normalised activation vectors (fp16 in LDS, as the reference
//     rounds them, kernel.cuh:133-138) while it is in registers once;
//   * phase 2: the 8 workgroups of a head are dealt to the rows -- 8/NB per (row, head), NB = 2 or 4 row slots -- each
//     streaming its share of THAT row's K/V (per-row page table, length, RoPE position, cache slot of the new token);
//   * X1 / X2 / X3 as in k_fused_decode_mha, one instance per row (granule arrays indexed by row; X3 carries two fp16 values
//     per granule);
//   * phase 3: each pair of Wo rows is dotted with the B attention vectors.
// 3 rows run in the 4-slot kernel with the last slot idle.  Scope: hidden 4096, 32 q = 32 kv heads, paged KV, rows up to
// 512 * 8 / NB cached tokens run straight-line (two 256-token tiles per workgroup), longer rows continue in a plain loop;
// the host sends batches whose rows it knows to be much longer to the stage pipeline, which spreads one row over more CUs.
#pragma once
#include "cf_fused_kernel.h"

namespace cf {

template <int NB>
struct FusedBGeom {
    static constexpr int NSP = FUSED_SPLITS / NB;                     // workgroups per (row, head)
    static constexpr int MAX_TOKENS = 512 * NSP;                      // cached tokens per row the straight-line tiles cover
    static constexpr int L_QKV = 0;                                   // float[384]       q|k|v of (row, head)
    static constexpr int L_A = L_QKV + 384 * 4;                       // h16[NB][4096]    xn (phase 1) / attention vectors (phase 3)
    static constexpr int L_O = L_A + NB * 4096 * 2;                   // float[9][128]
    static constexpr int L_ML = L_O + 9 * 128 * 4;                    // float[9][2] (+pad)
    static constexpr int L_REC = L_ML + 80;                           // float[NSP][FUSED_REC]; first: float[NB][8] sums of squares
    static constexpr int L_IDX = L_REC + 8 * FUSED_REC * 4;           // int[512]
    static constexpr int L_CS = L_IDX + 512 * 4;                      // float[256]
    static constexpr int L_CTL = L_CS + 256 * 4;                      // int[32]
    static constexpr int L_END = L_CTL + 128;
    static constexpr int LDS_BYTES = L_END > 84 * 1024 ? L_END : 84 * 1024;   // one workgroup per CU
};

template <int NB>
__global__ __launch_bounds__(FUSED_THREADS, 2) void k_fused_decode_mhab(FusedArgs a, int batch) {
    using GM = FusedBGeom<NB>;
    constexpr int NSP = GM::NSP, HID = 4096;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_qkv = reinterpret_cast<float*>(smem + GM::L_QKV);
    h16* s_a = reinterpret_cast<h16*>(smem + GM::L_A);
    float(*s_o)[HEAD_DIM] = reinterpret_cast<float(*)[HEAD_DIM]>(smem + GM::L_O);
    float(*s_ml)[2] = reinterpret_cast<float(*)[2]>(smem + GM::L_ML);
    float(*s_rec)[FUSED_REC] = reinterpret_cast<float(*)[FUSED_REC]>(smem + GM::L_REC);
    float* s_ss = reinterpret_cast<float*>(smem + GM::L_REC);          // [NB][8] (before X2 uses s_rec)
    int* s_idx = reinterpret_cast<int*>(smem + GM::L_IDX);
    float* s_cs = reinterpret_cast<float*>(smem + GM::L_CS);
    int* s_ctl = reinterpret_cast<int*>(smem + GM::L_CTL);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, gid = wave * 4 + (lane >> 4), d0 = l16 * 8;
    const int b = blockIdx.x;
    const int h = (b & 7) * 4 + (b >> 6);          // the 8 workgroups of a head share b % 8 (one XCD: speed only)
    const int j = (b >> 3) & 7;
    const int row = j / NSP, js = j % NSP;         // batch row this workgroup serves in phase 2, its split of that row's tokens
    const bool row_live = row < batch;
    CF_TRACE(0);

    // ---- first-level loads: x / residual of all rows, rms_w; thread t owns elements [8t, 8t+8) of every row ----------
    const float rs = a.na.residual ? 1.f : 0.f;
    h16x8 xv[NB], rv[NB];
#pragma unroll
    for (int r = 0; r < NB; ++r) {
        const int rr = r < batch ? r : 0;
        xv[r] = ld_h8(a.na.x + (size_t)rr * HID + tid * 8);
        rv[r] = ld_h8((a.na.residual ? a.na.residual : a.na.x) + (size_t)rr * HID + tid * 8);
    }
    const h16x8 wv8 = ld_h8(a.na.rms_w + tid * 8);
    const unsigned epoch = scalar_load(a.state) + 1u;   // (written by the previous launch: the scalar cache is invalidated at every kernel start)
    const unsigned xcc = my_xcc_id();
    if (tid == 0) granule_store(a.g_xcc + b, epoch, __builtin_bit_cast(float, xcc));
    int S = 0, ent0 = 0;
    if (row_live) {
        ent0 = scalar_load(a.indptr + row);
        S = a.seq_lens ? scalar_load(a.seq_lens + row) : scalar_load(a.indptr + row + 1) - 1 - ent0;
    }
    const int64_t roff = (a.positions && row_live) ? scalar_load(a.positions + row) * a.rope_stride : 0;
    const h16* kc = a.kptrs ? reinterpret_cast<const h16*>(scalar_load(a.kptrs + a.layer_id)) : a.k_cache;
    const h16* vc = a.vptrs ? reinterpret_cast<const h16*>(scalar_load(a.vptrs + a.layer_id)) : a.v_cache;

    // ---- weight stream of phase 1 (as k_fused_decode_mha: row pairs by share, masked slots) ---------------------------
    RowGroup<8, 2> ga, gb;
    // Equal shares, dealt DENSE (cf_fused_kernel.h): slot s < 3 of wavefront w is pair 2048 s + 8 b + w; the fourth slot is a masked
    // request.  (CF_B_DENSE=0: 24 consecutive pairs per workgroup from the host's table, as until round 5.)
#ifndef CF_B_DENSE
#define CF_B_DENSE 1
#endif
    const int p_lo = CF_B_DENSE ? 0 : a.p1_start[b], p_hi = CF_B_DENSE ? 0 : a.p1_start[b + 1];
    auto p1_pair = [&](int slot) { return CF_B_DENSE ? 2048 * slot + 8 * b + wave : p_lo + wave + 8 * slot; };
    auto p1_mine = [&](int slot) { return CF_B_DENSE ? slot < 3 : p_lo + wave + 8 * slot < p_hi; };
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16*>(a.Wqkv), 0, 3 * HID * HID * 2, 0x00020000);
    auto p1_load = [&](RowGroup<8, 2>& t, int slot) {
        const int pair = p1_pair(slot);
        const int voff = p1_mine(slot) ? pair * (2 * HID * 2) + lane * 16 : 0x40000000;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                t.w[r][i] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, voff + r * (HID * 2) + i * (WAVE * 16), 0, 2 /* nt */));
    };
    p1_load(ga, 0);
    p1_load(gb, 1);

    // ---- RMSNorm of every row, once per workgroup ----------------------------------------------------------------------
    float hx[NB][8];
#pragma unroll
    for (int r = 0; r < NB; ++r) {
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            hx[r][e] = __builtin_fmaf(rs, (float)rv[r][e], (float)xv[r][e]);
            ss = __builtin_fmaf(hx[r][e], hx[r][e], ss);
        }
        ss = sum64_lane63(ss);
        if (lane == 63) s_ss[r * 8 + wave] = ss;
    }
    // second-level loads of this workgroup's row: page-table slice, new-token slot, RoPE row
    const int ps = a.page_shift, pmask = (1 << ps) - 1;
    int tps = ((S + NSP - 1) / NSP + 31) & ~31;
    tps = tps < 32 ? 32 : tps;
    const int t0 = js * tps;
    int t1 = t0 + tps;
    t1 = t1 < S ? t1 : S;
    const int e0 = t0 >> ps;
    int n_idx = 0;
    if (row_live && t1 > t0) {
        n_idx = ((t1 - 1) >> ps) - e0 + 1;
        n_idx = n_idx < 512 ? n_idx : 512;   // (the two pre-requested tiles; tokens behind them find their pages through L2)
    }
    int idx_reg = 0, slot_reg = 0;
    if (tid < n_idx) idx_reg = a.indices[ent0 + e0 + tid];
    if (row_live && tid == 0) slot_reg = a.indices[ent0 + (S >> ps)];
    float cs_reg = 0.f;
    {
        const int n_ang = a.rope_style == 0 ? HEAD_DIM / 2 : HEAD_DIM;
        if (tid < n_ang) cs_reg = a.cos[roff + tid];
        else if (tid >= 128 && tid < 128 + n_ang) cs_reg = a.sin[roff + tid - 128];
    }
    lds_barrier();
#pragma unroll
    for (int r = 0; r < NB; ++r) {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) tot += s_ss[r * 8 + w];
        const float rcp = __builtin_amdgcn_rsqf(tot / (float)HID + a.na.eps);
        h16x8 xo;
#pragma unroll
        for (int e = 0; e < 8; ++e) xo[e] = (h16)(hx[r][e] * rcp * (float)wv8[e]);
        *reinterpret_cast<h16x8*>(s_a + (size_t)r * HID + tid * 8) = xo;
    }
    if (tid < n_idx) s_idx[tid] = idx_reg;
    if (tid < 256) s_cs[tid] = cs_reg;
    if (tid == 0) s_ctl[20] = slot_reg;
    lds_barrier();

    // rows 2p, 2p+1 x every batch row -> granules g_qkv[row][head][q|k|v][i]
    auto p1_dot_publish = [&](const RowGroup<8, 2>& t, int slot) {
        float acc[2][NB];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) acc[r][bb] = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) {
                const h16x8 av = *reinterpret_cast<const h16x8*>(s_a + (size_t)bb * HID + (i * WAVE + lane) * 8);
                acc[0][bb] = dot8h(t.w[0][i], av, acc[0][bb]);
                acc[1][bb] = dot8h(t.w[1][i], av, acc[1][bb]);
            }
        const int pair = p1_pair(slot);
        const int r = 2 * pair;
        u64* gp = a.g_qkv + (size_t)((r & 4095) >> 7) * 384 + (r >> 12) * 128 + (r & 127);
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) {
            const float v0 = sum64_lane63(acc[0][bb]), v1 = sum64_lane63(acc[1][bb]);
            if (lane == 63 && p1_mine(slot) && bb < batch) {
                granule_store(gp + (size_t)bb * (FUSED_HEADS * 384), epoch, v0);
                granule_store(gp + (size_t)bb * (FUSED_HEADS * 384) + 1, epoch, v1);
            }
        }
    };

    // ---- K/V tiles of this workgroup's (row, head, split), requested before q exists -------------------------------------
    const size_t kvstride = (size_t)FUSED_HEADS * HEAD_DIM;
    const h16* kbase = kc + h * HEAD_DIM + d0;
    const h16* vbase = vc + h * HEAD_DIM + d0;
    const h16* dummy = a.na.rms_w + d0;
    auto load_tile = [&](KvTile32<8>& t, int tbase) {
        const bool live = tbase < t1;                      // workgroup-uniform
        const h16* kb = live ? kbase : dummy;
        const h16* vb = live ? vbase : dummy;
        const size_t st = live ? kvstride : 0;
        size_t rows[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            int tk = tbase + u * FUSED_GROUPS + gid;
            tk = tk < t1 ? tk : t1 - 1;
            tk = tk > t0 ? tk : t0;
            int ei = (tk >> ps) - e0;
            ei = ei < 511 ? ei : 511;
            ei = ei > 0 ? ei : 0;
            rows[u] = ((size_t)s_idx[ei] << ps) + (size_t)(tk & pmask);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            t.k[u] = ld_stream(kb + rows[u] * st);
            t.v[u] = ld_stream(vb + rows[u] * st);
        }
    };
    constexpr int TILE = FUSED_GROUPS * 8;
    KvTile32<8> ta, tb;
    p1_dot_publish(ga, 0);
    p1_load(ga, 2);
    p1_dot_publish(gb, 1);
    p1_load(gb, 3);
    p1_dot_publish(ga, 2);
    load_tile(ta, t0);
    p1_dot_publish(gb, 3);
    load_tile(tb, t0 + TILE);
    CF_TRACE(1);   // phase 1 done

    // ---- X1: q|k|v of (row, head) ------------------------------------------------------------------------------------
    if (wave == 0) {
        bool ok = true;
        if (row_live) ok = sweep_granules<6>(a.g_qkv + ((size_t)row * FUSED_HEADS + h) * 384, 384, epoch, s_qkv, lane, a.state + 1, 1u);
        if (lane == 0) s_ctl[0] = ok;
    }
    lds_barrier();
    if (!s_ctl[0]) CF_FAIL_RETURN();
    CF_TRACE(2);   // X1 resolved
    // leader of (row, head) = its split 0: block index with j = row * NSP
    const u64 lead_x = __hip_atomic_load(a.g_xcc + ((b & ~0x38) | ((row * NSP) << 3)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    const float qscale = 1.44269504088896340736f * 0.08838834764831845f;
    float q[8];
    auto rope_lds = [&](const float* src, float (&dst)[8]) {
        if (a.rope_style == 0) {
            const float sgn = d0 < 64 ? -1.f : 1.f;
            const int a0 = d0 & 63, p0 = (d0 + 64) & 127;
#pragma unroll
            for (int e = 0; e < 8; ++e) dst[e] = src[d0 + e] * s_cs[a0 + e] + sgn * (src[p0 + e] * s_cs[128 + a0 + e]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float c = s_cs[d0 + e], sn = s_cs[128 + d0 + e];
                dst[e] = (e & 1) ? src[d0 + e] * c + src[d0 + (e ^ 1)] * sn : src[d0 + e] * c - src[d0 + (e ^ 1)] * sn;
            }
        }
    };
    rope_lds(s_qkv, q);
    h16x8 qh;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        q[e] *= qscale;
        qh[e] = (h16)q[e];
    }

    // ---- phase 2 ------------------------------------------------------------------------------------------------------
    float m = NEG_BIG, l = 0.f, o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto compute_tile = [&](const auto& t, int tbase) {
        constexpr int UU = sizeof(t.k) / sizeof(h16x8);
        float s[UU];
        bool valid[UU];
        float mx = NEG_BIG;
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            valid[u] = (tbase + u * FUSED_GROUPS + gid) < t1;
            s[u] = sum16(dot8h(t.k[u], qh, 0.f));
            s[u] = valid[u] ? s[u] : NEG_BIG;
            mx = fmaxf(mx, s[u]);
        }
        const float mnew = fmaxf(m, mx);
        const float alpha = fast_exp2(m - mnew);
        float psum = 0.f;
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            s[u] = valid[u] ? fast_exp2(s[u] - mnew) : 0.f;
            psum += s[u];
        }
        l = l * alpha + psum;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float acc = o[e] * alpha;
#pragma unroll
            for (int u = 0; u < UU; ++u) acc = __builtin_fmaf((float)t.v[u][e], s[u], acc);
            o[e] = acc;
        }
        m = mnew;
    };
    RowGroup<8, 2> go;
    compute_tile(ta, t0);
    go.load(a.Wo, 16 * b + 2 * wave, HID, HID, lane);      // 2 output rows per wavefront
    compute_tile(tb, t0 + TILE);
    // A row longer than MAX_TOKENS (the host routes such batches here when it cannot know better, or when the stage pipeline
    // would be slower still): the rest of the slice in 128-token tiles, two in flight, page numbers read through L2.  Off the
    // fast path: the branch sits behind the last request of the straight-line code, so the wait counts before it stay exact.
    if (t0 + 2 * TILE < t1) {
        constexpr int UL = 4, TILE_L = FUSED_GROUPS * UL;
        auto load_far = [&](KvTile32<UL>& t, int tbase) {
            size_t rows[UL];
#pragma unroll
            for (int u = 0; u < UL; ++u) {
                int tk = tbase + u * FUSED_GROUPS + gid;
                tk = tk < t1 ? tk : t1 - 1;
                rows[u] = ((size_t)a.indices[ent0 + (tk >> ps)] << ps) + (size_t)(tk & pmask);
            }
#pragma unroll
            for (int u = 0; u < UL; ++u) {
                t.k[u] = ld_stream(kbase + rows[u] * kvstride);
                t.v[u] = ld_stream(vbase + rows[u] * kvstride);
            }
        };
        KvTile32<UL> la, lb;
        const int tl = t0 + 2 * TILE;
        load_far(la, tl);
        for (int tt = tl; tt < t1; tt += 2 * TILE_L) {
            load_far(lb, tt + TILE_L);
            compute_tile(la, tt);
            load_far(la, tt + 2 * TILE_L);
            compute_tile(lb, tt + TILE_L);
        }
    }
    {
        const float mw = xmax32(xmax16(m));
        const float sc = fast_exp2(m - mw);
        l = xsum32(xsum16(l * sc));
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] *= sc;
        float r0, r1;
        xsum_rows8(o, r0, r1);      // (row r of the wavefront ends up with dims d0 + xrow_e(r) and d0 + 4 + xrow_e(r))
        m = mw;
        const int e0 = xrow_e(lane >> 4);
        s_o[wave][d0 + e0] = r0;
        s_o[wave][d0 + 4 + e0] = r1;
        if (lane == 0) { s_ml[wave][0] = m; s_ml[wave][1] = l; }
    }
    // the new token of this row (attended from registers) + k/v export + cache write: split 0 of (row, head)
    if (row_live && js == 0 && gid == 0) {
        float kf[8], vf[8];
        rope_lds(s_qkv + HEAD_DIM, kf);
#pragma unroll
        for (int e = 0; e < 8; ++e) vf[e] = s_qkv[2 * HEAD_DIM + d0 + e];
        h16x8 k16, v16;
#pragma unroll
        for (int e = 0; e < 8; ++e) { k16[e] = (h16)kf[e]; v16[e] = (h16)vf[e]; }
        const size_t ooff = (size_t)h * HEAD_DIM + d0;
        if (a.k_new) st_h8(a.k_new + (size_t)row * kvstride + ooff, k16);
        if (a.v_new) st_h8(a.v_new + (size_t)row * kvstride + ooff, v16);
        if (a.write_cache) {
            const size_t slot = ((size_t)s_ctl[20] << ps) + (size_t)(S & pmask);
            st_h8(const_cast<h16*>(kc) + slot * kvstride + ooff, k16);
            st_h8(const_cast<h16*>(vc) + slot * kvstride + ooff, v16);
        }
        float sn = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) sn = __builtin_fmaf(q[e], kf[e], sn);
        sn = sum16(sn);
#pragma unroll
        for (int e = 0; e < 8; ++e) s_o[8][d0 + e] = vf[e];
        if (l16 == 0) { s_ml[8][0] = sn; s_ml[8][1] = 1.f; }
    }
    lds_barrier();
    CF_TRACE(3);   // phase 2 done

    // ---- X2: one record per workgroup -> the leader of (row, head) ------------------------------------------------------
    const bool rec_local = (unsigned)(lead_x >> 32) == epoch && (unsigned)lead_x == xcc;
    if (row_live && tid < HEAD_DIM + 2) {
        const int nst = js == 0 ? 9 : 8;
        float M = NEG_BIG;
#pragma unroll
        for (int i = 0; i < 9; ++i) M = fmaxf(M, i < nst ? s_ml[i][0] : NEG_BIG);
        float val;
        if (tid < HEAD_DIM) {
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < 9; ++i)
                if (i < nst) acc = __builtin_fmaf(fast_exp2(s_ml[i][0] - M), s_o[i][tid], acc);
            val = acc;
        } else if (tid == HEAD_DIM) {
            val = M;
        } else {
            float L = 0.f;
#pragma unroll
            for (int i = 0; i < 9; ++i)
                if (i < nst) L = __builtin_fmaf(fast_exp2(s_ml[i][0] - M), s_ml[i][1], L);
            val = L;
        }
        // records of (row, head): g_rec[head][8 slots][..], slot = j (row * NSP + js): distinct per row
        granule_store_to(a.g_rec + ((size_t)h * FUSED_SPLITS + j) * FUSED_REC_G + tid, epoch, val, rec_local);
    }
    if (row_live && js == 0) {   // leader: wavefront w < NSP gathers record w of its (row, head), then the softmax merge
        if (wave < NSP) {
            const bool ok = sweep_granules<3>(a.g_rec + ((size_t)h * FUSED_SPLITS + row * NSP + wave) * FUSED_REC_G, HEAD_DIM + 2, epoch,
                                              s_rec[wave], lane, a.state + 1, 2u);
            if (lane == 0) s_ctl[1 + wave] = ok;
        }
        lds_barrier();
        bool all_ok = true;
        for (int w = 0; w < NSP; ++w) all_ok &= s_ctl[1 + w] != 0;
        if (!all_ok) CF_FAIL_RETURN();
        if (tid < HEAD_DIM) {
            float M = NEG_BIG;
#pragma unroll
            for (int w = 0; w < NSP; ++w) M = fmaxf(M, s_rec[w][HEAD_DIM]);
            float acc = 0.f, L = 0.f;
#pragma unroll
            for (int w = 0; w < NSP; ++w) {
                const float wt = fast_exp2(s_rec[w][HEAD_DIM] - M);
                acc = __builtin_fmaf(wt, s_rec[w][tid], acc);
                L = __builtin_fmaf(wt, s_rec[w][HEAD_DIM + 1], L);
            }
            // two fp16 values per granule: phase 3 consumes the attention output in fp16 (the reference rounds it there too,
            // kernel.cuh:553-559), and X3 -- every workgroup gathers every row -- moves half the granules
            const float mine = acc / L, next = __shfl_down(mine, 1);
            h16x2 pr;
            pr[0] = (h16)mine;
            pr[1] = (h16)next;
            // layout [wavefront chunk = head / 4][row slot][4 heads x 64]: what one wavefront of a consumer gathers is contiguous
            if (!(tid & 1)) granule_store(a.g_attn + ((size_t)(h >> 2) * NB + row) * 256 + (h & 3) * 64 + (tid >> 1), epoch, __builtin_bit_cast(float, pr));
        }
    }
    CF_TRACE(4);

    // ---- X3: the attention outputs of all rows: batch x 2048 granules (fp16 pairs), 256 per wavefront and row ---------------------------
    {   // ONE polling loop over all rows (a loop per row would pay the round trip once per row)
        constexpr int NG = 4 * NB;
        const u64* g = a.g_attn + (size_t)wave * NB * 256;
        const int count = batch * 256;
        for (unsigned spin = 0; spin < FUSED_SPIN_LIMIT; ++spin) {   // hint: the last granule of each (row, head) of this chunk
            u64 x = (u64)epoch << 32;
            if (lane < 4 * batch) x = __hip_atomic_load(g + lane * 64 + 63, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__popcll(__ballot((unsigned)(x >> 32) != epoch)) <= 2) break;      // (the sweep takes over for the last two)
            __builtin_amdgcn_s_sleep(2);
        }
        // (one wavefront watching all 32 NB (row, head) slots while the others wait at a barrier, as k_fused_decode_mha does:
        //  4 rows 42.1 vs 41.8 us, 2 rows level -- not here)
        unsigned v[NG];
        bool ok = true;
        for (unsigned spin = 0;; ++spin) {
            bool good = true;
#pragma unroll
            for (int k = 0; k < NG; ++k) {
                const int i = lane + WAVE * k;
                u64 x = (u64)epoch << 32;
                if (i < count) x = __hip_atomic_load(g + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v[k] = (unsigned)x;
                good &= (unsigned)(x >> 32) == epoch;
            }
            if (__all(good)) break;
            if (spin > FUSED_SPIN_LIMIT) {
                if (lane == 0) flag_exchange_error(a.state + 1, 3u);
                ok = false;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        unsigned* dst = reinterpret_cast<unsigned*>(s_a) + wave * 256;
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            const int i = lane + WAVE * k;
            if (i < count) dst[(i >> 8) * (HID / 2) + (i & 255)] = v[k];
        }
        if (lane == 0) s_ctl[9 + wave] = ok;
    }
    lds_barrier();
    {
        bool all_ok = true;
        for (int w = 0; w < 8; ++w) all_ok &= s_ctl[9 + w] != 0;
        if (!all_ok) CF_FAIL_RETURN();
    }
    CF_TRACE(5);
    // ---- phase 3: 16 rows of Wo per workgroup x every batch row -----------------------------------------------------------
    {
        float acc[2][NB];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) acc[r][bb] = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) {
                const h16x8 av = *reinterpret_cast<const h16x8*>(s_a + (size_t)(bb < batch ? bb : 0) * HID + (i * WAVE + lane) * 8);
                acc[0][bb] = dot8h(go.w[0][i], av, acc[0][bb]);
                acc[1][bb] = dot8h(go.w[1][i], av, acc[1][bb]);
            }
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) {
            const float v0 = sum64_lane63(acc[0][bb]), v1 = sum64_lane63(acc[1][bb]);
            if (lane == 63 && bb < batch) {
                a.out[(size_t)bb * HID + 16 * b + 2 * wave] = (h16)v0;
                a.out[(size_t)bb * HID + 16 * b + 2 * wave + 1] = (h16)v1;
            }
        }
    }
    // residual_out may alias residual: every workgroup read residual before X3 could complete
    if (a.residual_out && tid < 16) {
        for (int r = 0; r < batch; ++r) {
            const size_t i = (size_t)r * HID + 16 * b + tid;
            a.residual_out[i] = (h16)((float)a.na.x[i] + (float)a.na.residual[i]);
        }
    }
    if (b == 0 && tid == 0) {
        a.state[0] = epoch;
        a.state[2] = 0u;      // (no length arm to report: cf_workspace_last_arm documents 0 after a multi-row / MLA kernel)
    }
    CF_TRACE(6);
}

}  // namespace cf
