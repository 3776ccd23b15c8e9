// cf_fused_kernel_s.h -- the persistent fused decode kernel of a SMALL head-parallel shard ([out,in] weights, hidden 4096,
// batch 1, one q head per kv head): HKV = 4 local heads (one rank of TP = 8, BASELINE config 5) or 8 (TP = 4).
//
// A shard this small is a chain of hand-offs, not a byte stream (25 MB: 4 us of HBM time, 13 us of kernel in the
// geometry-generic kernel, cf_fused_kernel_g.h: phase 1 -> X1 -> phase 2 -> records -> merged records -> phase 3, every arrow a
// ~2 us round trip through the fabric).  Here the workgroups take ROLES, so that the chain is one hand-off shorter and its first
// link overlaps the K/V stream (round-2 verdict, item 3):
//   * 16 ATTENTION workgroups per head (NA = 16 HKV of the 256): they request their slice of the cached K/V -- S / 16 tokens, one
//     256-token tile up to S = 4096 -- in their first microsecond, take no part in the projections, wait for q|k|v of their head
//     (X1) while the tile is on its way, and publish one softmax record (normalised o as 64 fp16 pairs, m, l: 66 granules);
//   * the other NP = 256 - NA PROJECTION workgroups normalise x and stream the shard's 384 HKV rows of Wqkv (1 row per wavefront
//     at HKV = 4, 3 at HKV = 8) -- nothing else is in their way;
//   * the records are gathered by the workgroups that run the O projection, each for itself (34 KB at HKV = 4): no leader, no
//     second record level, no X3.  HKV = 4 (round 4): these are the 64 ATTENTION workgroups -- 64 rows of Wo each, requested
//     behind their K/V tile, 8 rows per wavefront -- and the projection workgroups leave after phase 1: a gather by 64 readers
//     instead of 256 is 0.7 us shorter (12.91 -> 12.19 us per call, same-box alternation; guide price list "allgather": 256 ->
//     32 readers -1.9 us on a 16 KB sweep).  HKV = 8: every workgroup gathers (a cheap wait on the records' m granules first)
//     and runs 16 rows.
// Results are identical in form to the generic kernel's (fixed-order sums, bit-reproducible); the record format and the granule
// protocol are cf_fused_kernel_g.h's.  Reference: the fused path does not shard (chat/llama/model.py:306-311); the contract is
// fairscale's Column/RowParallel split of the eager path (model.py:208-235) -- see clusterfusion_amd/tp.py.
#pragma once
#include "cf_fused_kernel.h"

#ifndef CF_S_HINT_SLACK
#define CF_S_HINT_SLACK 24     // records that may still be missing when the cheap wait hands over to the sweep (0 / 8 / 24 / 40 / 64:
                               // 13.07 / 12.98 / 12.96 / 12.95 / 13.22 us per call, three alternations)
#endif
#ifndef CF_S_P3_ATTN
#define CF_S_P3_ATTN 1       // phase 3 on the attention workgroups only (64 rows each; 4-head shard): the projection workgroups leave after phase 1 (0: everybody)
#endif
#ifndef CF_S_P3_ATTN8
#define CF_S_P3_ATTN8 0      // (experiment) the same for the 8-head shard: 32 rows on each of its 128 attention workgroups
#endif
#ifndef CF_S_WO_WHEN
#define CF_S_WO_WHEN 0       // (CF_S_P3_ATTN) the attention workgroups' 64 Wo rows are requested 0: behind the K/V tile, 1: when X1 has resolved, 2: behind tile A's arithmetic
#endif
#ifndef CF_S_HINT_ALL
#define CF_S_HINT_ALL 0      // 1: the attention workgroups, too, wait on the records' m granules before they sweep
#endif

namespace cf {

template <int HKV>
struct ShardGeom {
    static constexpr int NSA = 16;                        // attention workgroups (KV splits) per head
    static constexpr int NA = NSA * HKV;                  // attention workgroups: b < NA, head b % HKV, split b / HKV
    static constexpr int NP = FUSED_WGS - NA;             // projection workgroups
    static constexpr int ROWS = 3 * HKV * HEAD_DIM;       // rows of the shard's Wqkv: q of the local heads | k | v
    static constexpr int RPW = ROWS / NP;                 // ... per projection workgroup
    static constexpr int NRW = RPW / 8;                   // ... per wavefront
    static constexpr int U = 8, TILE = FUSED_GROUPS * U;  // one 256-token tile requested before X1
    static constexpr int SHORT_TOKENS = NSA * TILE;       // the straight-line arm covers this (4096)
    static constexpr int JO = HKV * HEAD_DIM / 512;       // 1-KB pieces of one Wo row
    static constexpr int NREC = NA;                       // records every workgroup gathers
    // LDS carve
    static constexpr int L_QKV = 0;                                    // float[384]
    static constexpr int L_A = L_QKV + 384 * 4;                        // float[4096] (x); later h16[HKV * 128] attention out
    static constexpr int L_O = L_A + 4096 * 4;                         // float[9][128]
    static constexpr int L_ML = L_O + 9 * HEAD_DIM * 4;                // float[9][2] (+pad)
    static constexpr int L_W = L_ML + 80;                              // float[9] merge weights (+pad)
    static constexpr int L_REC = L_W + 48;                             // unsigned[NREC][FUSED_RECH]
    static constexpr int MAX_IDX = 8192;                               // page-table entries one attention workgroup stages
    static constexpr int L_IDX = L_REC + NREC * FUSED_RECH * 4;        // int[MAX_IDX]
    static constexpr int L_CS = L_IDX + MAX_IDX * 4;                   // float[256]
    static constexpr int L_CTL = L_CS + 256 * 4;                       // int[32]
    static constexpr int L_END = L_CTL + 128;
    static constexpr int LDS_BYTES = L_END > 84 * 1024 ? L_END : 84 * 1024;      // (more than half a CU: one workgroup per CU)
    static_assert(ROWS % NP == 0 && RPW % 8 == 0 && (NRW == 1 || NRW == 3), "whole rows per wavefront");
    static_assert(NREC % 8 == 0 && LDS_BYTES <= 160 * 1024, "records per wavefront / LDS carve");
};

template <int HKV>
__global__ __launch_bounds__(512, 2) void k_fused_decode_s(FusedArgs a) {
    using GM = ShardGeom<HKV>;
    constexpr int HID = 4096, U = GM::U, TILE = GM::TILE, NSA = GM::NSA, JO = GM::JO, RH = FUSED_RECH, RM = HEAD_DIM / 2, RL = HEAD_DIM / 2 + 1;
    constexpr int LO = HKV * HEAD_DIM;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_qkv = reinterpret_cast<float*>(smem + GM::L_QKV);
    float* s_a = reinterpret_cast<float*>(smem + GM::L_A);
    float(*s_o)[HEAD_DIM] = reinterpret_cast<float(*)[HEAD_DIM]>(smem + GM::L_O);
    float(*s_ml)[2] = reinterpret_cast<float(*)[2]>(smem + GM::L_ML);
    float* s_w = reinterpret_cast<float*>(smem + GM::L_W);
    unsigned* s_rec = reinterpret_cast<unsigned*>(smem + GM::L_REC);
    int* s_idx = reinterpret_cast<int*>(smem + GM::L_IDX);
    float* s_cs = reinterpret_cast<float*>(smem + GM::L_CS);
    int* s_ctl = reinterpret_cast<int*>(smem + GM::L_CTL);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, gid = wave * 4 + (lane >> 4), d0 = l16 * 8;
    const int b = blockIdx.x;
    const bool attn_role = b < GM::NA;      // (workgroup-uniform)
    CF_TRACE(0);
    const unsigned epoch = scalar_load(a.state) + 1u;
    const unsigned tp_epoch = tp_call_epoch(a);
    // phase-3 rows of this workgroup: requested by everybody before anything else is waited for (16 KB x HKV / 4 per workgroup)
    constexpr bool P3A = CF_S_P3_ATTN && (HKV == 4 || CF_S_P3_ATTN8);      // phase 3 on the attention workgroups only
    constexpr int P3R = 2 * FUSED_WGS / GM::NA;                              // ... rows of Wo per wavefront there (8 / 4)
    RowGroup<JO, 2> go;
    RowGroup<JO, P3R> go8;      // (P3A: 8 P3R rows per attention workgroup)
    int arm = FUSED_ARM_TWO;

    if (!attn_role) {
        // ================= projection workgroup: RMSNorm(x [+ residual]) . NRW rows of Wqkv per wavefront ======================
        const h16* rp = a.na.residual ? a.na.residual : a.na.x;
        const float rs = a.na.residual ? 1.f : 0.f;
        const h16x8 xv = ld_h8(a.na.x + tid * 8), rv = ld_h8(rp + tid * 8), wv8 = ld_h8(a.na.rms_w + tid * 8);
        const int rr0 = (b - GM::NA) * GM::RPW + wave * GM::NRW;      // first row of this wavefront in the shard's Wqkv
        RowGroup<8, 1> r[GM::NRW];
#pragma unroll
        for (int i = 0; i < GM::NRW; ++i) {
            const h16* p = a.Wqkv + (size_t)(rr0 + i) * HID + lane * 8;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) r[i].w[0][jj] = ld_stream(p + jj * WAVE * 8);
        }
        if constexpr (!P3A) go.load(a.Wo, 16 * b + 2 * wave, HID, LO, lane);
        float hx[8];
        {
            float ss = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                hx[e] = __builtin_fmaf(rs, (float)rv[e], (float)xv[e]);
                ss = __builtin_fmaf(hx[e], hx[e], ss);
            }
            ss = sum64_lane63(ss);
            if (lane == 63) s_w[wave] = ss;
        }
        lds_barrier();
        float xn[8][8];
        {
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) tot += s_w[w];
            const float rcp = __builtin_amdgcn_rsqf(tot / (float)HID + a.na.eps);
            f32x4 lo, hi;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                lo[e] = hx[e] * rcp * (float)wv8[e];
                hi[e] = hx[4 + e] * rcp * (float)wv8[4 + e];
            }
            *reinterpret_cast<f32x4*>(&s_a[tid * 8]) = lo;
            *reinterpret_cast<f32x4*>(&s_a[tid * 8 + 4]) = hi;
            lds_barrier();
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const f32x4 p0 = *reinterpret_cast<const f32x4*>(&s_a[(jj * WAVE + lane) * 8]);
                const f32x4 p1 = *reinterpret_cast<const f32x4*>(&s_a[(jj * WAVE + lane) * 8 + 4]);
#pragma unroll
                for (int e = 0; e < 4; ++e) { xn[jj][e] = p0[e]; xn[jj][4 + e] = p1[e]; }
            }
        }
#pragma unroll
        for (int i = 0; i < GM::NRW; ++i) {
            float res[1];
            r[i].dot(xn, res);
            // row rr of the shard's Wqkv: section rr / (HKV * 128) (q | k | v), head (rr / 128) % HKV, dim rr % 128 -> granule
            // [head][section * 128 + dim]: the attention workgroups of a head sweep its 384 granules
            const int rr = rr0 + i, sec = rr / (HKV * HEAD_DIM), hh = (rr >> 7) % HKV;
            if (lane == 63) granule_store(a.g_qkv + (size_t)hh * 384 + sec * HEAD_DIM + (rr & 127), epoch, res[0]);
        }
        CF_TRACE(1);
        if constexpr (P3A) {      // nothing left to do here: the attention workgroups run phase 3
            CF_TRACE(6);
            return;
        }
        lds_barrier();      // (s_a is reused for the attention output below)
    } else {
        // ================= attention workgroup: split j of head g ===============================================================
        const int g = b % HKV, j = b / HKV;
        int S = a.seq_len, ent0 = 0;
        if (a.indptr) {
            ent0 = scalar_load(a.indptr);
            S = a.seq_lens ? scalar_load(a.seq_lens) : scalar_load(a.indptr + 1) - 1 - ent0;
        }
        const int64_t roff = a.positions ? scalar_load(a.positions) * a.rope_stride : 0;
        const h16* kc = a.kptrs ? reinterpret_cast<const h16*>(scalar_load(a.kptrs + a.layer_id)) : a.k_cache;
        const h16* vc = a.vptrs ? reinterpret_cast<const h16*>(scalar_load(a.vptrs + a.layer_id)) : a.v_cache;
        const int ps = a.page_shift, pmask = (1 << ps) - 1;
        int tps = ((S + NSA - 1) / NSA + 31) & ~31;
        tps = tps < 32 ? 32 : tps;
        const int t0 = j * tps;
        int t1 = t0 + tps;
        t1 = t1 < S ? t1 : S;
        const int e0 = t0 >> ps;
        const int max_idx = (a.flags & 64) ? 512 : GM::MAX_IDX;   // (debug bit 64: stage only what the first tile needs)
        int n_idx = 0, n_need = 0;
        if (a.indptr && t1 > t0) {
            n_need = ((t1 - 1) >> ps) - e0 + 1;
            n_idx = n_need < max_idx ? n_need : max_idx;   // (a longer slice reads the rest through L2)
        }
        // second-level loads: page-table slice, new-token slot, RoPE row -- registers first, LDS behind them
        {
            int idx_reg = 0, slot_reg = 0;
            float cs_reg = 0.f;
            if (tid < n_idx) idx_reg = a.indices[ent0 + e0 + tid];
            if (a.indptr && tid == 0) slot_reg = a.indices[ent0 + (S >> ps)];
            const int n_ang = a.rope_style == 0 ? HEAD_DIM / 2 : HEAD_DIM;
            if (tid < n_ang) cs_reg = a.cos[roff + tid];
            else if (tid >= 128 && tid < 128 + n_ang) cs_reg = a.sin[roff + tid - 128];
            if (tid < n_idx) s_idx[tid] = idx_reg;
            for (int i = tid + 512; i < n_idx; i += 512) s_idx[i] = a.indices[ent0 + e0 + i];
            if (tid < 256) s_cs[tid] = cs_reg;
            if (tid == 0) s_ctl[20] = slot_reg;
        }
        lds_barrier();
        const size_t kvstride = (size_t)HKV * HEAD_DIM;
        const h16* kbase = kc + g * HEAD_DIM + d0;
        const h16* vbase = vc + g * HEAD_DIM + d0;
        const h16* dummy = a.na.rms_w + d0;
        constexpr FusedArm<0> NEAR{};
        constexpr FusedArm<1> FARIDX{};
        auto load_tile = [&](auto& t, int tbase, auto far_c) {   // unconditional; a tile behind the slice reads one dummy line
            constexpr int UU = sizeof(t.k) / sizeof(h16x8);
            constexpr bool FAR = decltype(far_c)::value != 0;
            const bool live = tbase < t1;
            const h16* kb = live ? kbase : dummy;
            const h16* vb = live ? vbase : dummy;
            const size_t st = live ? kvstride : 0;
            size_t rows[UU];
#pragma unroll
            for (int u = 0; u < UU; ++u) {
                int tk = tbase + u * 32 + gid;
                tk = tk < t1 ? tk : t1 - 1;
                tk = tk > t0 ? tk : t0;
                if (!a.indptr) {
                    rows[u] = (size_t)tk;
                } else if constexpr (FAR) {
                    rows[u] = ((size_t)a.indices[ent0 + (tk >> ps)] << ps) + (size_t)(tk & pmask);
                } else {
                    int ei = (tk >> ps) - e0;
                    ei = ei < GM::MAX_IDX ? ei : GM::MAX_IDX - 1;
                    rows[u] = ((size_t)s_idx[ei] << ps) + (size_t)(tk & pmask);
                }
            }
#pragma unroll
            for (int u = 0; u < UU; ++u) {
                t.k[u] = ld_stream(kb + rows[u] * st);
                t.v[u] = ld_stream(vb + rows[u] * st);
            }
        };
        KvTile32<U> ta;
        load_tile(ta, t0, NEAR);
        if constexpr (P3A) {
            if constexpr (CF_S_WO_WHEN == 0) go8.load(a.Wo, (8 * b + wave) * P3R, HID, LO, lane);
        } else go.load(a.Wo, 16 * b + 2 * wave, HID, LO, lane);
        CF_TRACE(1);
        // ---- X1: q | k | v of head g (written through by the projection workgroups) ---------------------------------------------
        if (wave == 0) {
            const bool ok = sweep_granules<6>(a.g_qkv + (size_t)g * 384, 384, epoch, s_qkv, lane, a.state + 1, 1u);
            if (lane == 0) s_ctl[0] = ok;
        }
        lds_barrier();
        if (!s_ctl[0]) CF_FAIL_RETURN();
        CF_TRACE(2);
        if constexpr (P3A && CF_S_WO_WHEN == 1) go8.load(a.Wo, (8 * b + wave) * P3R, HID, LO, lane);
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
        float q[8];
        rope_lds(s_qkv, q);
        h16x8 qh;      // q rounded to fp16, as the reference keeps it (kernel.cuh:299-314): q.k runs on v_dot2_f32_f16
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            q[e] *= qscale;
            qh[e] = (h16)q[e];
        }
        float m = NEG_BIG, l = 0.f, o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        auto compute_tile = [&](const auto& t, int tbase) {
            constexpr int UU = sizeof(t.k) / sizeof(h16x8);
            float s[UU];
            bool valid[UU];
            float mx = NEG_BIG;
#pragma unroll
            for (int u = 0; u < UU; ++u) {
                valid[u] = (tbase + u * 32 + gid) < t1;
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
        CF_TRACE(7);
        compute_tile(ta, t0);
        CF_TRACE(8);
        if constexpr (P3A && CF_S_WO_WHEN == 2) go8.load(a.Wo, (8 * b + wave) * P3R, HID, LO, lane);
        if (tps > TILE) {      // (workgroup-uniform) a slice longer than the tile requested before X1: 128-token tiles, two deep
            arm = FUSED_ARM_LONG;
            constexpr int UL = 4, TILE_L = 32 * UL;
            KvTile32<UL> la, lb;
            const int tl = t0 + TILE;
            if (n_need <= max_idx) {
                load_tile(la, tl, NEAR);
                for (int tt = tl; tt < t1; tt += 2 * TILE_L) {
                    load_tile(lb, tt + TILE_L, NEAR);
                    compute_tile(la, tt);
                    load_tile(la, tt + 2 * TILE_L, NEAR);
                    compute_tile(lb, tt + TILE_L);
                }
            } else {
                load_tile(la, tl, FARIDX);
                for (int tt = tl; tt < t1; tt += 2 * TILE_L) {
                    load_tile(lb, tt + TILE_L, FARIDX);
                    compute_tile(la, tt);
                    load_tile(la, tt + 2 * TILE_L, FARIDX);
                    compute_tile(lb, tt + TILE_L);
                }
            }
        }
        // the 4 lane-groups of a wavefront merge in registers, the 8 wavefront states (+ the new token) meet in LDS
        {
            const float mw = xmax32(xmax16(m));
            const float sc = fast_exp2(m - mw);
            const float lw = xsum32(xsum16(l * sc));
            float ov[8], r0, r1;
#pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] = o[e] * sc;
            xsum_rows8(ov, r0, r1);
            const int ex = xrow_e(lane >> 4);
            s_o[wave][d0 + ex] = r0;
            s_o[wave][d0 + 4 + ex] = r1;
            if (lane == 0) { s_ml[wave][0] = mw; s_ml[wave][1] = lw; }
        }
        // the new token (attended from registers, kernel.cuh:444-477) + k/v export: split 0 of the head
        if (j == 0 && gid == 0) {
            float kf[8], vf[8];
            rope_lds(s_qkv + HEAD_DIM, kf);
#pragma unroll
            for (int e = 0; e < 8; ++e) vf[e] = s_qkv[2 * HEAD_DIM + d0 + e];
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
            float sn = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) sn = __builtin_fmaf(q[e], kf[e], sn);
            sn = sum16(sn);
#pragma unroll
            for (int e = 0; e < 8; ++e) s_o[8][d0 + e] = vf[e];
            if (l16 == 0) { s_ml[8][0] = sn; s_ml[8][1] = 1.f; }
        }
        CF_TRACE(12);
        lds_barrier();
        CF_TRACE(3);
        // ---- the record of (head g, split j): normalised o as 64 fp16 pairs, then m and l ---------------------------------------
        {
            const int nst = j == 0 ? 9 : 8;          // the new token belongs to split 0
            u64* rec = a.g_rec + ((size_t)g * NSA + j) * RH;
            if (tid < 9) {
                float mv[9];
#pragma unroll
                for (int w = 0; w < 9; ++w) mv[w] = s_ml[w][0];
                float M = NEG_BIG;
#pragma unroll
                for (int w = 0; w < 9; ++w) M = fmaxf(M, w < nst ? mv[w] : NEG_BIG);
                float L = 0.f;
#pragma unroll
                for (int w = 0; w < 9; ++w)
                    if (w < nst) L = __builtin_fmaf(fast_exp2(mv[w] - M), s_ml[w][1], L);
                const float rL = L > 0.f ? 1.f / L : 0.f;      // (a split without tokens: o = 0, l = 0)
                s_w[tid] = tid < nst ? fast_exp2(mv[tid] - M) * rL : 0.f;
                if (tid == 0) {
                    granule_store(rec + RM, epoch, M);
                    granule_store(rec + RL, epoch, L);
                }
            }
            lds_barrier();
            if (tid < HEAD_DIM) {
                float val = 0.f;
#pragma unroll
                for (int w = 0; w < 9; ++w)   // (the new-token slot of splits > 0 is uninitialised LDS: 0 x NaN)
                    val = __builtin_fmaf(s_w[w], w < nst ? s_o[w][tid] : 0.f, val);
                const float next = __shfl_down(val, 1);
                h16x2 pr;
                pr[0] = (h16)val;
                pr[1] = (h16)next;
                if (!(tid & 1)) granule_store(rec + (tid >> 1), epoch, __builtin_bit_cast(float, pr));
            }
        }
    }
    CF_TRACE(4);

    // ================= everybody: gather the NREC records, finish the softmax merge locally ========================================
    {
        // a cheap wait first (lanes watch the m granule of one record each; a waiting workgroup must not sweep 34 KB per round)
        if (wave == 0 && (!attn_role || CF_S_HINT_ALL)) {
            for (unsigned spin = 0; spin < FUSED_SPIN_LIMIT; ++spin) {
                bool good = true;
#pragma unroll
                for (int k = 0; k < (GM::NREC + 63) / 64; ++k) {
                    const int rec = lane + 64 * k;
                    u64 x = (u64)epoch << 32;
                    if (rec < GM::NREC) x = __hip_atomic_load(a.g_rec + (size_t)rec * RH + RM, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    good &= (unsigned)(x >> 32) == epoch;
                }
                if (__popcll(__ballot(!good)) <= CF_S_HINT_SLACK) break;      // (the sweep below catches the stragglers one round trip sooner)
                __builtin_amdgcn_s_sleep(2);
            }
        }
        lds_barrier();
        constexpr int RPWV = GM::NREC / 8, CNT = RPWV * RH;      // records / granules per wavefront (contiguous in g_rec)
        const bool ok = sweep_granules_raw<(CNT + 63) / 64>(a.g_rec + (size_t)wave * CNT, CNT, epoch, s_rec + wave * CNT, lane, a.state + 1, 2u);
        if (lane == 0) s_ctl[1 + wave] = ok;
        lds_barrier();
        bool all_ok = true;
        for (int w = 0; w < 8; ++w) all_ok &= s_ctl[1 + w] != 0;
        if (!all_ok) CF_FAIL_RETURN();
        for (int t = tid; t < HKV * HEAD_DIM; t += 512) {      // (fp16, as the reference rounds the attention output)
            const unsigned* r = s_rec + (size_t)(t >> 7) * NSA * RH;
            const int d = t & 127;
            float M = NEG_BIG;
#pragma unroll
            for (int w = 0; w < NSA; ++w) M = fmaxf(M, __builtin_bit_cast(float, r[w * RH + RM]));
            float acc = 0.f, L = 0.f;
#pragma unroll
            for (int w = 0; w < NSA; ++w) {      // fixed order
                const float wt = fast_exp2(__builtin_bit_cast(float, r[w * RH + RM]) - M) * __builtin_bit_cast(float, r[w * RH + RL]);
                acc = __builtin_fmaf(wt, (float)__builtin_bit_cast(h16x2, r[w * RH + (d >> 1)])[d & 1], acc);
                L += wt;
            }
            reinterpret_cast<h16*>(s_a)[t] = (h16)(L > 0.f ? acc / L : 0.f);
        }
    }
    lds_barrier();
    CF_TRACE(5);
    // ---- phase 3: 16 rows of Wo per workgroup ---------------------------------------------------------------------------------------
    h16x8 av[JO];
#pragma unroll
    for (int jj = 0; jj < JO; ++jj) av[jj] = *reinterpret_cast<const h16x8*>(reinterpret_cast<const h16*>(s_a) + (jj * WAVE + lane) * 8);
    if constexpr (P3A) {
        float res[P3R];
        go8.dot_h(av, res);
        if (lane == 63) {
#pragma unroll
            for (int r = 0; r < P3R; ++r) a.out[(8 * b + wave) * P3R + r] = (h16)res[r];
        }
        if (a.tp_world > 0) tp_publish_wg<P3R / 2>(a, tp_epoch, 4 * P3R * b, res, reinterpret_cast<unsigned*>(s_qkv), lane, wave);      // (s_qkv: free since phase 2)
        if (a.residual_out && tid < 8 * P3R) {
            const int i = 8 * P3R * b + tid;
            a.residual_out[i] = (h16)((float)a.na.x[i] + (float)a.na.residual[i]);
        }
    } else {
        float res[2];
        go.dot_h(av, res);
        if (lane == 63) {
            a.out[16 * b + 2 * wave] = (h16)res[0];
            a.out[16 * b + 2 * wave + 1] = (h16)res[1];
        }
        if (a.tp_world > 0) tp_publish_wg<1>(a, tp_epoch, 8 * b, res, reinterpret_cast<unsigned*>(s_qkv), lane, wave);      // (s_qkv: free since phase 2)
        if (a.residual_out && tid < 16) {
            const int i = 16 * b + tid;
            a.residual_out[i] = (h16)((float)a.na.x[i] + (float)a.na.residual[i]);
        }
    }
    if (b == 0 && tid == 0) {
        a.state[0] = epoch;
        a.state[2] = arm;      // which arm this call took (cf_workspace_last_arm); b = 0 is an attention workgroup
    }
    CF_TRACE(6);
}

}  // namespace cf
