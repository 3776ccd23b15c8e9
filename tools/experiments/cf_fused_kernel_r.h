// cf_fused_kernel_r.h -- the grouped-query persistent decode kernel with ROLES: Llama-3-8B, 32 q / 8 kv heads ([out,in] weights,
// hidden 4096, batch 1; BASELINE config 4), for cached lengths up to ~10 k tokens.
//
// Why.  In k_fused_decode_g<8, 4> (cf_fused_kernel_g.h) every workgroup runs the same program: 24 projection rows + a 256-token
// K/V slice (324 KB, on chip by ~14 us at the CU's ~25 GB/s) -> X1 -> phase 2 -> record -> leader -> X3 -> 16 rows of Wo.  Only
// the 128 KB of Wo stream during the 10 us exchange chain behind X1 (timeline: profiles/r04_timelines.md), they cannot be
// requested before X1 without delaying it (the polls queue behind them in the CU's memory pipe: rounds 1-2), and every hop of
// the chain is polled from a CU whose pipe is full of Wo rows.  459 KB per CU need 18.4 us; the kernel takes 24.2 + boundary.
//
// Here the 32 workgroups of a kv-head group split into 16 ATTENTION and 16 PROJECTION workgroups (same XCD, interleaved):
//   * projection workgroup: 32 of the group's 768 q|k|v rows (4 per wavefront) and, requested right behind them as their
//     registers retire, 24 rows of Wo (3 per wavefront) -- 459 KB in ONE uninterrupted request stream.  It never waits for X1,
//     holds no K/V, publishes no record; its only hand-off is X3, polled after its stream has drained;
//   * attention workgroup: 16 projection rows (2 per wavefront), then a 512-token K/V slice as four 128-token MFMA tiles (all
//     four requested before X1: 128 registers per lane), 8 rows of Wo (1 per wavefront, requested when the first tile has been
//     consumed) -- 452 KB.  Its K/V lands while X1 resolves; the chain X1 -> phase 2 -> record -> leader -> X3 runs on CUs whose
//     memory pipe holds 64 KB of Wo instead of 128, and a leader merges 16 records instead of 32.
// Everything else is cf_fused_kernel_g.h's: phase 2 on the matrix cores (S = K q^T on v_mfma_f32_16x16x32_f16, online softmax
// on the accumulator layout, O += P V on v_mfma_f32_16x16x16_f16 with V through ds_read_b64_tr_b16), half-size fp16 records,
// XCD-local hand-offs when the published XCC ids confirm the placement, the device-side length with a straight-line arm (the
// slice fits the four tiles: S <= 8192) and a loop arm (any length: correct, but the balance above is sized for ~8 k tokens --
// the host routes longer caches to k_fused_decode_g<8, 4>, cf_api.hip).  Fixed-order fp32 merges: bit-reproducible.
// Reference: the reference kernels have no grouped-query path (chat/llama/model.py:166-175 repeat_kv is the eager definition);
// the phases are kernel.cuh:95-619's.
#pragma once
#include "cf_fused_kernel_g.h"

namespace cf {

struct RoleGeom {
    static constexpr int HKV = 8, G = 4, HQ = 32, NS = 32;
    static constexpr int NSA = 16;                        // attention workgroups per kv head (even j); the odd j project
    static constexpr int RG = (G + 2) * HEAD_DIM;         // 768 projection rows of one kv-head group
    static constexpr int A_RPW = 2, P_RPW = 4;            // projection rows per wavefront: attention / projection workgroup
    static constexpr int A_ROWS = NSA * 8 * A_RPW;        // rows [0, 256) of the group belong to its attention workgroups
    static constexpr int A_WO = 1, P_WO = 3;              // rows of Wo per wavefront
    static constexpr int P_WO_ROWS = 8 * HKV * (NS - NSA) * 8 * P_WO / 8;      // 3072: outputs [0, 3072) on the projection workgroups
    static constexpr int TILE = 128, NT = 4;              // four 128-token MFMA tiles requested before X1
    static constexpr int SHORT_TOKENS = NSA * NT * TILE;  // 8192: the straight-line arm
    static constexpr int JO = HQ * HEAD_DIM / 512;        // 8 1-KB pieces of one Wo row
    static constexpr int NST = 9;                         // softmax states per q head: 8 wavefronts + the new token
    static constexpr int KT_ROW = 136;
    static constexpr int MAX_IDX = 8192;
    // LDS carve (the grouped-query kernel's, with 16 records per head)
    static constexpr int L_QKV = 0;                                    // float[768]
    static constexpr int L_A = L_QKV + RG * 4;                         // float[4096] (x, then attention out)
    static constexpr int O_BYTES = G * NST * HEAD_DIM * 4, REC_BYTES = NSA * FUSED_RECH * 4;
    static constexpr int L_O = L_A + 4096 * 4;                         // float[G][NST][128]; later the leader's gathered records
    static constexpr int L_ML = L_O + (O_BYTES > REC_BYTES ? O_BYTES : REC_BYTES);
    static constexpr int L_W = L_ML + ((G * NST * 2 * 4 + 15) & ~15);
    static constexpr int L_QH = L_W + ((G * NST * 4 + 15) & ~15);      // h16[G][128] RoPE'd, scaled q
    static constexpr int L_VT = L_QH + G * HEAD_DIM * 2;               // h16[8 wavefronts][8][16][16] V images
    static constexpr int L_KT = L_VT + 8 * 4096;                       // h16[8 wavefronts][16 tokens][KT_ROW] K images
    static constexpr int L_IDX = L_KT + 8 * 16 * KT_ROW * 2;           // int[MAX_IDX]
    static constexpr int L_CS = L_IDX + MAX_IDX * 4;                   // float[256]
    static constexpr int L_CTL = L_CS + 256 * 4;                       // int[32]
    static constexpr int L_END = L_CTL + 128;
    static constexpr int LDS_BYTES = L_END > 84 * 1024 ? L_END : 84 * 1024;
    static_assert(P_WO_ROWS + HKV * NSA * 8 * A_WO == 4096, "every output row has one owner");
    static_assert(A_ROWS + (NS - NSA) * 8 * P_RPW == RG, "every projection row has one owner");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS carve exceeds a CU");
};

__global__ __launch_bounds__(512, 2) void k_fused_decode_r(FusedArgs a) {
    using GM = RoleGeom;
    constexpr int HKV = GM::HKV, G = GM::G, HQ = GM::HQ, NS = GM::NS, NSA = GM::NSA, RG = GM::RG, JO = GM::JO, HID = 4096, NST = GM::NST;
    constexpr int TILE = GM::TILE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_qkv = reinterpret_cast<float*>(smem + GM::L_QKV);
    float* s_a = reinterpret_cast<float*>(smem + GM::L_A);
    float(*s_o)[NST][HEAD_DIM] = reinterpret_cast<float(*)[NST][HEAD_DIM]>(smem + GM::L_O);
    float(*s_ml)[NST][2] = reinterpret_cast<float(*)[NST][2]>(smem + GM::L_ML);
    float(*s_w)[NST] = reinterpret_cast<float(*)[NST]>(smem + GM::L_W);
    float* s_rec = reinterpret_cast<float*>(smem + GM::L_O);
    int* s_idx = reinterpret_cast<int*>(smem + GM::L_IDX);
    float* s_cs = reinterpret_cast<float*>(smem + GM::L_CS);
    int* s_ctl = reinterpret_cast<int*>(smem + GM::L_CTL);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, gid = wave * 4 + (lane >> 4), d0 = l16 * 8;
    const int b = blockIdx.x;
    // XCD x (= b % 8) hosts kv head x ^ 1 (the slow-address heads 1 and 5 on the XCDs whose X1 resolves first: cf_fused_kernel_g.h)
    const int g = (b & 7) ^ 1, j = b >> 3;
    const bool att = (j & 1) == 0;      // (workgroup-uniform)
    const int jr = j >> 1;              // index among the group's attention / projection workgroups
    CF_TRACE(0);
    const unsigned xcc = my_xcc_id();

    // ---- small first-level loads first (loads return in issue order) ---------------------------------------------------------
    const h16* rp = a.na.residual ? a.na.residual : a.na.x;
    const float rs = a.na.residual ? 1.f : 0.f;
    const h16x8 xv = ld_h8(a.na.x + tid * 8), rv = ld_h8(rp + tid * 8), wv8 = ld_h8(a.na.rms_w + tid * 8);
    const unsigned epoch = scalar_load(a.state) + 1u;
    if (tid == 0) granule_store(a.g_xcc + b, epoch, __builtin_bit_cast(float, xcc));

    auto global_row = [&](int rr) -> int {           // Wqkv rows: q of all heads | k | v
        if (rr < G * HEAD_DIM) return g * G * HEAD_DIM + rr;
        if (rr < (G + 1) * HEAD_DIM) return HQ * HEAD_DIM + g * HEAD_DIM + (rr - G * HEAD_DIM);
        return (HQ + HKV) * HEAD_DIM + g * HEAD_DIM + (rr - (G + 1) * HEAD_DIM);
    };
    auto row_load = [&](RowGroup<8, 1>& t, const h16* W, int row) {
        const h16* p = W + (size_t)row * HID + lane * 8;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) t.w[0][jj] = ld_stream(p + jj * WAVE * 8);
    };
    // the normalised activations of this lane's 8 x 8 elements, through LDS (RMSNorm once per workgroup)
    float hx[8];
    auto norm_partial = [&]() {
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            hx[e] = __builtin_fmaf(rs, (float)rv[e], (float)xv[e]);
            ss = __builtin_fmaf(hx[e], hx[e], ss);
        }
        ss = sum64_lane63(ss);
        if (lane == 63) s_rec[wave] = ss;     // s_rec is free until X2
    };
    auto norm_finish = [&](float (&xn)[8][8]) {      // after an LDS barrier behind norm_partial
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) tot += s_rec[w];
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
    };
    // the XCC ids of the group's 32 workgroups (lane i % 32: member i): requested right behind the first row
    auto members = [&]() -> u64 {
        return __hip_atomic_load(a.g_xcc + (((lane & 31) << 3) | (b & 7)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // X3 + phase 3: every workgroup gathers the attention output (fp16 pairs) and runs its rows of Wo
    auto x3_gather = [&]() -> bool {
        constexpr int PER = HQ * HEAD_DIM / 16;      // 256 granules per wavefront = 4 heads
        wait_hint(a.g_attn + wave * PER + 63, 4, HEAD_DIM / 2, epoch, lane, 2);
        const bool ok = sweep_granules_raw<PER / 64>(a.g_attn + wave * PER, PER, epoch, reinterpret_cast<unsigned*>(s_a) + wave * PER, lane,
                                                     a.state + 1, 3u);
        if (lane == 0) s_ctl[9 + wave] = ok;
        lds_barrier();
        bool all_ok = true;
        for (int w = 0; w < 8; ++w) all_ok &= s_ctl[9 + w] != 0;
        return all_ok;
    };
    auto attn_vector = [&](h16x8 (&av)[JO]) {
#pragma unroll
        for (int jj = 0; jj < JO; ++jj) av[jj] = *reinterpret_cast<const h16x8*>(reinterpret_cast<const h16*>(s_a) + (jj * WAVE + lane) * 8);
    };

    if (!att) {
        // ================= projection workgroup: 4 rows of the group's q|k|v per wavefront, then 3 rows of Wo ====================
        const int rr0 = GM::A_ROWS + (jr * 8 + wave) * GM::P_RPW;
        const int orow0 = ((g * (NS - NSA) + jr) * 8 + wave) * GM::P_WO;
        RowGroup<8, 1> r0, r1, r2, r3;
        row_load(r0, a.Wqkv, global_row(rr0));
        const u64 member_x = members();
        row_load(r1, a.Wqkv, global_row(rr0 + 1));
        row_load(r2, a.Wqkv, global_row(rr0 + 2));
        row_load(r3, a.Wqkv, global_row(rr0 + 3));
        norm_partial();
        lds_barrier();
        float xn[8][8];
        norm_finish(xn);
        u64* gq = a.g_qkv + (size_t)g * RG + rr0;
        float res[1];
        r0.dot(xn, res);
        const bool grp_local = __all((unsigned)(member_x >> 32) == epoch && (unsigned)member_x == xcc);
        if (lane == 63) granule_store_to(gq, epoch, res[0], grp_local);
        RowGroup<JO, 1> w0, w1, w2;      // (the phase-3 rows take the registers the projection rows retire)
        row_load(w0, a.Wo, orow0);
        r1.dot(xn, res);
        if (lane == 63) granule_store_to(gq + 1, epoch, res[0], grp_local);
        row_load(w1, a.Wo, orow0 + 1);
        r2.dot(xn, res);
        if (lane == 63) granule_store_to(gq + 2, epoch, res[0], grp_local);
        row_load(w2, a.Wo, orow0 + 2);
        r3.dot(xn, res);
        if (lane == 63) granule_store_to(gq + 3, epoch, res[0], grp_local);
        CF_TRACE(1);
        lds_barrier();      // (s_a is reused for the attention output)
        if (!x3_gather()) CF_FAIL_RETURN();
        CF_TRACE(5);
        h16x8 av[JO];
        attn_vector(av);
        float o0[1], o1[1], o2[1];
        w0.dot_h(av, o0);
        w1.dot_h(av, o1);
        w2.dot_h(av, o2);
        if (lane == 63) {
            a.out[orow0] = (h16)o0[0];
            a.out[orow0 + 1] = (h16)o1[0];
            a.out[orow0 + 2] = (h16)o2[0];
        }
        if (a.residual_out && tid < 8 * GM::P_WO) {
            const int i = orow0 - wave * GM::P_WO + tid;
            a.residual_out[i] = (h16)((float)a.na.x[i] + (float)a.na.residual[i]);
        }
        CF_TRACE(6);
        return;
    }

    // ================= attention workgroup: split jr of kv head g ================================================================
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
    const int t0 = jr * tps;
    int t1 = t0 + tps;
    t1 = t1 < S ? t1 : S;
    const int e0 = t0 >> ps;
    const int max_idx = (a.flags & 64) ? 512 : GM::MAX_IDX;
    int n_idx = 0, n_need = 0;
    if (a.indptr && t1 > t0) {
        n_need = ((t1 - 1) >> ps) - e0 + 1;
        n_idx = n_need < max_idx ? n_need : max_idx;
    }
    int idx_reg = 0, slot_reg = 0;
    float cs_reg = 0.f;
    {
        if (tid < n_idx) idx_reg = a.indices[ent0 + e0 + tid];
        if (a.indptr && tid == 0) slot_reg = a.indices[ent0 + (S >> ps)];
        const int n_ang = a.rope_style == 0 ? HEAD_DIM / 2 : HEAD_DIM;
        if (tid < n_ang) cs_reg = a.cos[roff + tid];
        else if (tid >= 128 && tid < 128 + n_ang) cs_reg = a.sin[roff + tid - 128];
    }
    const size_t kvstride = (size_t)HKV * HEAD_DIM;
    const h16* kbase = kc + g * HEAD_DIM + d0;
    const h16* vbase = vc + g * HEAD_DIM + d0;
    const h16* dummy = a.na.rms_w + d0;
    typedef KvTile32<4> Tile;
    auto load_tile = [&](Tile& t, int tbase, auto far_c) {   // unconditional; a tile behind the slice reads one dummy line
        constexpr bool FAR = decltype(far_c)::value != 0;
        const bool live = tbase < t1;
        const h16* kb = live ? kbase : dummy;
        const h16* vb = live ? vbase : dummy;
        const size_t st = live ? kvstride : 0;
        size_t rows[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
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
        for (int u = 0; u < 4; ++u) {
            t.k[u] = ld_stream(kb + rows[u] * st);
            t.v[u] = ld_stream(vb + rows[u] * st);
        }
    };
    constexpr FusedArm<0> NEAR{};
    constexpr FusedArm<1> FARIDX{};

    // ---- phase 1: 2 rows per wavefront; the four K/V tiles go out around them ------------------------------------------------
    const int rr0 = (jr * 8 + wave) * GM::A_RPW;
    RowGroup<8, 1> r0, r1;
    row_load(r0, a.Wqkv, global_row(rr0));
    const u64 member_x = members();
    row_load(r1, a.Wqkv, global_row(rr0 + 1));
    norm_partial();
    if (tid < n_idx) s_idx[tid] = idx_reg;
    for (int i = tid + 512; i < n_idx; i += 512) s_idx[i] = a.indices[ent0 + e0 + i];
    if (tid < 256) s_cs[tid] = cs_reg;
    if (tid == 0) s_ctl[20] = slot_reg;
    lds_barrier();
    Tile ta, tb, tc, td;
    load_tile(ta, t0, NEAR);
    load_tile(tb, t0 + TILE, NEAR);
    float xn[8][8];
    norm_finish(xn);
    bool grp_local;
    {
        u64* gq = a.g_qkv + (size_t)g * RG + rr0;
        float res[1];
        r0.dot(xn, res);
        grp_local = __all((unsigned)(member_x >> 32) == epoch && (unsigned)member_x == xcc);
        if (lane == 63) granule_store_to(gq, epoch, res[0], grp_local);
        r1.dot(xn, res);
        if (lane == 63) granule_store_to(gq + 1, epoch, res[0], grp_local);
    }
    const int ai = g * NSA + jr;                       // index among the 128 attention workgroups
    const int orow = GM::P_WO_ROWS + ai * 8 + wave;    // this wavefront's row of Wo

    // ================= from here on: one straight copy per arm ===================================================================
    // (the arms part BEFORE tiles C and D are requested: with four tiles in flight across the branch the register allocator spilled
    //  freshly loaded tiles behind s_waitcnt vmcnt(0).  The loop arm keeps two tiles across X1, as cf_fused_kernel_g.h does.)
    auto rest = [&](auto long_c) {
    constexpr bool LONG = decltype(long_c)::value != 0;
    if constexpr (!LONG) {
        __builtin_amdgcn_sched_barrier(0);      // (tiles C and D take the registers of the rows and of xn: not before the dots)
        load_tile(tc, t0 + 2 * TILE, NEAR);
        load_tile(td, t0 + 3 * TILE, NEAR);
    }
    CF_TRACE(1);

    // ---- X1: q (4 heads) | k | v of this kv-head group ---------------------------------------------------------------------------
    if (wave == 0) {
        const bool ok = sweep_granules<RG / 64>(a.g_qkv + (size_t)g * RG, RG, epoch, s_qkv, lane, a.state + 1, 1u);
        if (lane == 0) s_ctl[0] = ok;
    }
    lds_barrier();
    if (!s_ctl[0]) CF_FAIL_RETURN();
    CF_TRACE(2);

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
    const float qscale = 1.44269504088896340736f * 0.08838834764831845f;
    typedef h16 h16x4 __attribute__((ext_vector_type(4)));
    typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
    h16x8 qb[4];
    f32x4 oacc[8];
    float mfM = NEG_BIG, mfL = 0.f;
    h16* s_qh = reinterpret_cast<h16*>(smem + GM::L_QH);
    h16* s_vt = reinterpret_cast<h16*>(smem + GM::L_VT) + wave * 2048;
    h16* s_kt = reinterpret_cast<h16*>(smem + GM::L_KT) + wave * 16 * GM::KT_ROW;
    {   // RoPE'd, scaled q of the 4 heads -> fp16 in LDS (one element per thread), then the B operand of q.k
        const int hh = tid >> 7, d = tid & 127;
        const float* src = s_qkv + hh * HEAD_DIM;
        float v;
        if (a.rope_style == 0) {
            const int a0 = d & 63;
            v = src[d] * s_cs[a0] + (d < 64 ? -1.f : 1.f) * (src[(d + 64) & 127] * s_cs[128 + a0]);
        } else {
            const float c = s_cs[d], sn = s_cs[128 + d];
            v = (d & 1) ? src[d] * c + src[d ^ 1] * sn : src[d] * c - src[d ^ 1] * sn;
        }
        s_qh[tid] = (h16)(v * qscale);
    }
    lds_barrier();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const h16x8 v = *reinterpret_cast<const h16x8*>(s_qh + (l16 < G ? l16 : 0) * HEAD_DIM + 32 * u + (lane >> 4) * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) qb[u][e] = l16 < G ? v[e] : (h16)0.f;
    }
#pragma unroll
    for (int jb = 0; jb < 8; ++jb) oacc[jb] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- phase 2 on the matrix cores (cf_fused_kernel_g.h compute_tile, MF) ------------------------------------------------------
    auto compute_tile = [&](const Tile& t, int tbase) {
        const int lg = lane >> 4;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            *reinterpret_cast<h16x8*>(s_kt + (4 * u + lg) * GM::KT_ROW + l16 * 8) = t.k[u];
            *reinterpret_cast<h16x8*>(s_vt + (l16 >> 1) * 256 + (4 * u + lg) * 16 + (l16 & 1) * 8) = t.v[u];
        }
        asm volatile("" ::: "memory");
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const h16x8 ka = *reinterpret_cast<const h16x8*>(s_kt + l16 * GM::KT_ROW + 32 * u + lg * 8);
            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(ka, qb[u], d, 0, 0, 0);
        }
        const int tok0 = tbase + lg * 32 + wave * 4;
        float mx = NEG_BIG;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            d[r] = (tok0 + r) < t1 ? d[r] : NEG_BIG;
            mx = fmaxf(mx, d[r]);
        }
        mx = xmax32(xmax16(mx));
        const float mnew = fmaxf(mfM, mx);
        const float alpha = fast_exp2(mfM - mnew);
        h16x4 pa;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float pr = (tok0 + r) < t1 ? fast_exp2(d[r] - mnew) : 0.f;
            psum += pr;
            pa[r] = (h16)pr;
        }
        mfL = mfL * alpha + psum;
        mfM = mnew;
        float al[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) al[r] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, alpha), r));
#pragma unroll
        for (int jb = 0; jb < 8; ++jb) {
            const fp16x4_t vt = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                (__attribute__((address_space(3))) fp16x4_t*)(s_vt + jb * 256 + l16 * 4 + (lane >> 4) * 64));
#pragma unroll
            for (int r = 0; r < 4; ++r) oacc[jb][r] *= al[r];
            oacc[jb] = __builtin_amdgcn_mfma_f32_16x16x16f16(pa, __builtin_bit_cast(h16x4, vt), oacc[jb], 0, 0, 0);
        }
        asm volatile("" ::: "memory");
    };
    CF_TRACE(7);
    compute_tile(ta, t0);
    CF_TRACE(8);
    RowGroup<JO, 1> go;
    if constexpr (LONG) {
        // (the host routes long caches to k_fused_decode_g<8, 4>: this arm is the correctness path of a sequence that outgrew its
        //  hint.)  128-token tiles two deep behind the two requested before X1; Wo last.
        const int tl = t0 + 2 * TILE;
        Tile la, lb;
        if (n_need <= max_idx) {
            load_tile(la, tl, NEAR);
            compute_tile(tb, t0 + TILE);
            for (int tt = tl; tt < t1; tt += 2 * TILE) {
                load_tile(lb, tt + TILE, NEAR);
                compute_tile(la, tt);
                load_tile(la, tt + 2 * TILE, NEAR);
                compute_tile(lb, tt + TILE);
            }
        } else {
            load_tile(la, tl, FARIDX);
            compute_tile(tb, t0 + TILE);
            for (int tt = tl; tt < t1; tt += 2 * TILE) {
                load_tile(lb, tt + TILE, FARIDX);
                compute_tile(la, tt);
                load_tile(la, tt + 2 * TILE, FARIDX);
                compute_tile(lb, tt + TILE);
            }
        }
        row_load(go, a.Wo, orow);
    } else {
        row_load(go, a.Wo, orow);
        compute_tile(tb, t0 + TILE);
        compute_tile(tc, t0 + 2 * TILE);
        compute_tile(td, t0 + 3 * TILE);
    }
    CF_TRACE(9);

    {   // one state per wavefront and head
        const float lw = xsum32(xsum16(mfL));
        if (lane < G) { s_ml[lane][wave][0] = mfM; s_ml[lane][wave][1] = lw; }
        if (lane < 16) {
#pragma unroll
            for (int jb = 0; jb < 8; ++jb)
#pragma unroll
                for (int r = 0; r < G; ++r) s_o[r][wave][16 * jb + lane] = oacc[jb][r];
        }
    }
    // the new token + k/v export: split 0 of the group
    if (jr == 0 && gid == 0) {
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
            for (int e = 0; e < 8; ++e) sn = __builtin_fmaf((float)s_qh[hh * HEAD_DIM + d0 + e], kf[e], sn);
            sn = sum16(sn);
#pragma unroll
            for (int e = 0; e < 8; ++e) s_o[hh][NST - 1][d0 + e] = vf[e];
            if (l16 == 0) { s_ml[hh][NST - 1][0] = sn; s_ml[hh][NST - 1][1] = 1.f; }
        }
    }
    CF_TRACE(12);
    lds_barrier();
    CF_TRACE(3);

    // ---- X2: 4 records per attention workgroup -> the q head's leader (attention workgroup jr = head index in the group) -----------
    constexpr int RH = FUSED_RECH, RM = HEAD_DIM / 2, RL = HEAD_DIM / 2 + 1;
    auto rec_o = [](const unsigned* r, int d) -> float {
        return (float)__builtin_bit_cast(h16x2, r[d >> 1])[d & 1];
    };
    {
        const int nst = jr == 0 ? NST : NST - 1;
        if (tid < G * NST) {
            const int hh = tid / NST, i = tid - hh * NST;
            float mv[NST];
#pragma unroll
            for (int w = 0; w < NST; ++w) mv[w] = s_ml[hh][w][0];
            float M = NEG_BIG;
#pragma unroll
            for (int w = 0; w < NST; ++w) M = fmaxf(M, w < nst ? mv[w] : NEG_BIG);
            float L = 0.f;
#pragma unroll
            for (int w = 0; w < NST; ++w)
                if (w < nst) L = __builtin_fmaf(fast_exp2(mv[w] - M), s_ml[hh][w][1], L);
            const float rL = L > 0.f ? 1.f / L : 0.f;
            s_w[hh][i] = i < nst ? fast_exp2(mv[i] - M) * rL : 0.f;
            if (i == 0) {
                u64* rec = a.g_rec + (((size_t)g * G + hh) * NSA + jr) * RH;
                granule_store_to(rec + RM, epoch, M, grp_local);
                granule_store_to(rec + RL, epoch, L, grp_local);
            }
        }
        lds_barrier();
        {
            const int hh = tid >> 7, d = tid & 127;
            float val = 0.f;
#pragma unroll
            for (int w = 0; w < NST; ++w)   // (the new-token slot of splits > 0 is uninitialised LDS: 0 x NaN)
                val = __builtin_fmaf(s_w[hh][w], w < nst ? s_o[hh][w][d] : 0.f, val);
            const float next = __shfl_down(val, 1);
            h16x2 pr;
            pr[0] = (h16)val;
            pr[1] = (h16)next;
            if (!(d & 1)) granule_store_to(a.g_rec + (((size_t)g * G + hh) * NSA + jr) * RH + (d >> 1), epoch, __builtin_bit_cast(float, pr), grp_local);
        }
    }
    if (jr < G) {   // leader of q head g * G + jr: wavefront w gathers records 2 w, 2 w + 1, then the softmax merge
        unsigned* s_recu = reinterpret_cast<unsigned*>(s_rec);
        lds_barrier();   // s_rec reuses s_o: every wavefront is done reading the states
        constexpr int CNT = (NSA / 8) * RH;
        const bool ok = sweep_granules_raw<(CNT + 63) / 64>(a.g_rec + (((size_t)g * G + jr) * NSA + wave * (NSA / 8)) * RH,
                                                           CNT, epoch, s_recu + wave * CNT, lane, a.state + 1, 2u);
        if (lane == 0) s_ctl[1 + wave] = ok;
        lds_barrier();
        bool all_ok = true;
        for (int w = 0; w < 8; ++w) all_ok &= s_ctl[1 + w] != 0;
        if (!all_ok) CF_FAIL_RETURN();
        if (tid < HEAD_DIM) {
            float M = NEG_BIG;
#pragma unroll
            for (int w = 0; w < NSA; ++w) M = fmaxf(M, __builtin_bit_cast(float, s_recu[w * RH + RM]));
            float acc = 0.f, L = 0.f;
#pragma unroll
            for (int w = 0; w < NSA; ++w) {
                const float wt = fast_exp2(__builtin_bit_cast(float, s_recu[w * RH + RM]) - M) * __builtin_bit_cast(float, s_recu[w * RH + RL]);
                acc = __builtin_fmaf(wt, rec_o(s_recu + w * RH, tid), acc);
                L += wt;
            }
            const float mine = L > 0.f ? acc / L : 0.f;
            const float next = __shfl_down(mine, 1);
            h16x2 pr;
            pr[0] = (h16)mine;
            pr[1] = (h16)next;
            if (!(tid & 1)) granule_store(a.g_attn + ((size_t)g * G + jr) * (HEAD_DIM / 2) + (tid >> 1), epoch, __builtin_bit_cast(float, pr));
        }
    }
    CF_TRACE(4);
    // ---- X3 + phase 3: one row of Wo per wavefront ----------------------------------------------------------------------------------
    if (!x3_gather()) CF_FAIL_RETURN();
    CF_TRACE(5);
    h16x8 av[JO];
    attn_vector(av);
    float res[1];
    go.dot_h(av, res);
    if (lane == 63) a.out[orow] = (h16)res[0];
    if (a.residual_out && tid < 8) {
        const int i = orow - wave + tid;
        a.residual_out[i] = (h16)((float)a.na.x[i] + (float)a.na.residual[i]);
    }
    if (b == 0 && tid == 0) {       // (b = 0: j = 0, an attention workgroup)
        a.state[0] = epoch;
        a.state[2] = LONG ? FUSED_ARM_LONG : FUSED_ARM_TWO;
    }
    CF_TRACE(6);
    };   // rest
    if (tps <= GM::NT * TILE) rest(FusedArm<0>{});
    else rest(FusedArm<1>{});
}

}  // namespace cf
