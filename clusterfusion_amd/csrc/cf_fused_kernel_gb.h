// cf_fused_kernel_gb.h -- the grouped-query layer (32 q / 8 kv heads: Llama-3-8B, BASELINE config 4) for a SMALL BATCH of 2 .. 4
// sequences in ONE persistent launch (gfx950): configs 4 and f2 composed (VERDICT r5 missing #3).
//
// The reference's batched entry is one launch for every batch size (llama_kernel_batch_sglang_dispatch.cu:89: grid =
// HEAD_NUM * CLUSTER_SIZE * batch_size) and re-reads every weight per sequence (kernel_batch_sglang.cuh:63-64); it has no
// grouped-query path at all.  Until round 6 a 32q/8kv batch took the five-launch stage pipeline here.  This kernel is
// k_fused_decode_g<8, 4> (cf_fused_kernel_g.h) with the rows riding one weight stream, the way k_fused_decode_mhab
// (cf_fused_kernel_b.h) does it for the 32q/32kv model:
//   * phase 1: the 24 q|k|v rows of a workgroup are dotted with the NB normalised activation vectors (fp16 in LDS) while they
//     are in registers once -- Wqkv is streamed once per batch;
//   * phase 2: the 32 workgroups of a kv-head group are dealt to the rows, NSP = 32 / NB per (row, kv head); each streams its
//     token slice of THAT row's K/V (own page table, length, RoPE position, cache slot) once for the 4 q heads that share it, on
//     the matrix cores (S = K q^T and O += P V as in k_fused_decode_g's MF path);
//   * X1 / X2 / X3 per row: q|k|v granules [row][kv head][768], records [row][q head][NSP], attention vectors [row][2048 pairs];
//   * phase 3: the workgroup's 16 rows of Wo are dotted with the NB attention vectors.
// 3 rows run in the 4-slot kernel with one slot idle.  Scope: hidden 4096, 32 q / 8 kv heads, [out,in] weights, paged KV; rows
// up to 256 NSP cached tokens run on the two tiles requested before X1, longer rows continue in a loop (any length).
#pragma once
#include "cf_fused_kernel_g.h"

namespace cf {

template <int NB>
struct FusedGBGeom {
    static constexpr int HKV = 8, G = 4, HQ = 32, NS = 32;
    static constexpr int NSP = NS / NB;                                // workgroups per (row, kv head)
    static constexpr int RG = (G + 2) * HEAD_DIM;                      // 768 projection rows per kv-head group
    static constexpr int RPW = RG / NS, RPWV = 3;                      // 24 rows per workgroup, 3 per wavefront
    static constexpr int TILE = 128;                                   // tokens per MFMA tile (16 per wavefront)
    static constexpr int MAX_TOKENS = 2 * TILE * NSP;                  // cached tokens per row the two pre-requested tiles cover
    static constexpr int NST = 9;                                      // softmax states per q head: 8 wavefronts + the new token
    static constexpr int RECW = NSP / 8;                               // records one wavefront of a leader sweeps
    static constexpr int KT_ROW = 136;
    static constexpr int MAX_IDX = 4096;                               // page-table entries one workgroup stages
    static constexpr int L_QKV = 0;                                    // float[768]         q (4 heads) | k | v of (row, kv head)
    static constexpr int L_A = L_QKV + RG * 4;                         // h16[NB][4096]      xn (phase 1) / attention vectors (phase 3)
    static constexpr int L_O = L_A + NB * 4096 * 2;                    // float[G][NST][128]; later unsigned[NSP][FUSED_RECH]
    static constexpr int O_BYTES = G * NST * HEAD_DIM * 4, REC_BYTES = NSP * FUSED_RECH * 4;
    static constexpr int L_ML = L_O + (O_BYTES > REC_BYTES ? O_BYTES : REC_BYTES);   // float[G][NST][2]
    static constexpr int L_W = L_ML + ((G * NST * 2 * 4 + 15) & ~15);  // float[G][NST] merge weights
    static constexpr int L_QH = L_W + ((G * NST * 4 + 15) & ~15);      // h16[G][128] RoPE'd, scaled q
    static constexpr int L_VT = L_QH + G * HEAD_DIM * 2;               // h16[8 wavefronts][8][16][16] V images
    static constexpr int L_KT = L_VT + 8 * 4096;                       // h16[8 wavefronts][16][KT_ROW] K images
    static constexpr int L_IDX = L_KT + 8 * 16 * KT_ROW * 2;           // int[MAX_IDX]
    static constexpr int L_CS = L_IDX + MAX_IDX * 4;                   // float[256]
    static constexpr int L_SS = L_CS + 256 * 4;                        // float[NB][8] sums of squares
    static constexpr int L_CTL = L_SS + NB * 8 * 4;                    // int[32]
    static constexpr int L_END = L_CTL + 128;
    static constexpr int LDS_BYTES = L_END > 84 * 1024 ? L_END : 84 * 1024;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS carve exceeds a CU");
    static_assert(NSP >= 8 && NSP % 8 == 0, "a leader's 8 wavefronts sweep NSP / 8 records each");
};

template <int NB>
__global__ __launch_bounds__(FUSED_THREADS, 2) void k_fused_decode_gb(FusedArgs a, int batch) {
    using GM = FusedGBGeom<NB>;
    constexpr int HKV = GM::HKV, G = GM::G, HQ = GM::HQ, NS = GM::NS, NSP = GM::NSP, RG = GM::RG, HID = 4096, NST = GM::NST, TILE = GM::TILE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_qkv = reinterpret_cast<float*>(smem + GM::L_QKV);
    h16* s_a = reinterpret_cast<h16*>(smem + GM::L_A);
    float(*s_o)[NST][HEAD_DIM] = reinterpret_cast<float(*)[NST][HEAD_DIM]>(smem + GM::L_O);
    float(*s_ml)[NST][2] = reinterpret_cast<float(*)[NST][2]>(smem + GM::L_ML);
    float(*s_w)[NST] = reinterpret_cast<float(*)[NST]>(smem + GM::L_W);
    unsigned* s_recu = reinterpret_cast<unsigned*>(smem + GM::L_O);
    h16* s_qh = reinterpret_cast<h16*>(smem + GM::L_QH);
    int* s_idx = reinterpret_cast<int*>(smem + GM::L_IDX);
    float* s_cs = reinterpret_cast<float*>(smem + GM::L_CS);
    float* s_ss = reinterpret_cast<float*>(smem + GM::L_SS);
    int* s_ctl = reinterpret_cast<int*>(smem + GM::L_CTL);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, gid = wave * 4 + (lane >> 4), d0 = l16 * 8;
    const int b = blockIdx.x;
    const int g = (b & 7) ^ 1;                     // kv-head group: its 32 workgroups share b % 8 (one XCD: speed only; map of k_fused_decode_g)
    const int j = (b >> 3) % NS;
    const int row = j / NSP, js = j % NSP;         // the batch row this workgroup serves in phase 2, its split of that row's tokens
    const bool row_live = row < batch;
    h16* s_vt = reinterpret_cast<h16*>(smem + GM::L_VT) + wave * 2048;
    h16* s_kt = reinterpret_cast<h16*>(smem + GM::L_KT) + wave * 16 * GM::KT_ROW;
    CF_TRACE(0);
    const unsigned xcc = my_xcc_id();

    // ---- first-level loads: x / residual of all rows, rms_w; thread t owns elements [8t, 8t+8) of every row -------------------
    const float rs = a.na.residual ? 1.f : 0.f;
    h16x8 xv[NB], rv[NB];
#pragma unroll
    for (int r = 0; r < NB; ++r) {
        const int rr = r < batch ? r : 0;
        xv[r] = ld_h8(a.na.x + (size_t)rr * HID + tid * 8);
        rv[r] = ld_h8((a.na.residual ? a.na.residual : a.na.x) + (size_t)rr * HID + tid * 8);
    }
    const h16x8 wv8 = ld_h8(a.na.rms_w + tid * 8);
    // start values in one batch of unconditional scalar loads (a dead slot reads row 0's: always mapped, discarded)
    const int rq = row_live ? row : 0;
    const uint32_t* stp = a.state;
    const int32_t* slp = a.seq_lens ? a.seq_lens + rq : reinterpret_cast<const int32_t*>(stp);
    const int64_t* pop = a.positions ? a.positions + rq : reinterpret_cast<const int64_t*>(stp);
    const uint64_t* kpp = a.kptrs ? a.kptrs + a.layer_id : reinterpret_cast<const uint64_t*>(stp);
    const uint64_t* vpp = a.vptrs ? a.vptrs + a.layer_id : reinterpret_cast<const uint64_t*>(stp);
    const unsigned ep0 = scalar_load(stp);
    const int ip0 = scalar_load(a.indptr + rq), ip1 = scalar_load(a.indptr + rq + 1), sl0 = scalar_load(slp);
    const int64_t po0 = scalar_load(pop);
    const uint64_t kp0 = scalar_load(kpp), vp0 = scalar_load(vpp);
    const unsigned epoch = ep0 + 1u;
    const int ent0 = ip0;
    int S = a.seq_lens ? sl0 : ip1 - 1 - ent0;
    S = row_live ? S : 0;
    const int64_t roff = a.positions ? po0 * a.rope_stride : 0;
    const h16* kc = a.kptrs ? reinterpret_cast<const h16*>(kp0) : a.k_cache;
    const h16* vc = a.vptrs ? reinterpret_cast<const h16*>(vp0) : a.v_cache;
    if (tid == 0) granule_store(a.g_xcc + b, epoch, __builtin_bit_cast(float, xcc));

    // ---- second-level loads (page-table slice, new-token slot, RoPE row): registers first, LDS later -----------------------------
    const int ps = a.page_shift, pmask = (1 << ps) - 1;
    int tps = ((S + NSP - 1) / NSP + 31) & ~31;
    tps = tps < 32 ? 32 : tps;
    const int t0 = js * tps;
    int t1 = t0 + tps;
    t1 = t1 < S ? t1 : S;
    t1 = t1 > t0 ? t1 : t0;
    const int e0 = t0 >> ps;
    const int max_idx = (a.flags & 64) ? 512 : GM::MAX_IDX;   // (debug bit 64: stage only what the pre-requested tiles need -- the through-L2 path at test sizes)
    int n_idx = 0, n_need = 0;
    if (t1 > t0) {
        n_need = ((t1 - 1) >> ps) - e0 + 1;
        n_idx = n_need < max_idx ? n_need : max_idx;
    }
    int idx_reg = 0, slot_reg = 0;
    float cs_reg = 0.f;
    if (tid < n_idx) idx_reg = a.indices[ent0 + e0 + tid];
    if (row_live && tid == 0) slot_reg = a.indices[ent0 + (S >> ps)];
    {
        const int n_ang = a.rope_style == 0 ? HEAD_DIM / 2 : HEAD_DIM;
        if (tid < n_ang) cs_reg = a.cos[roff + tid];
        else if (tid >= 128 && tid < 128 + n_ang) cs_reg = a.sin[roff + tid - 128];
    }

    // ---- phase-1 weight stream: 3 rows of the group's q|k|v row space per wavefront -------------------------------------------
    const int rr0 = GM::RPW * j + GM::RPWV * wave;
    auto global_row = [&](int rr) -> int {           // Wqkv rows: q of all heads | k | v
        if (rr < G * HEAD_DIM) return g * G * HEAD_DIM + rr;
        if (rr < (G + 1) * HEAD_DIM) return HQ * HEAD_DIM + g * HEAD_DIM + (rr - G * HEAD_DIM);
        return (HQ + HKV) * HEAD_DIM + g * HEAD_DIM + (rr - (G + 1) * HEAD_DIM);
    };
    RowGroup<8, 1> r0, r1, r2;
    auto p1_load = [&](RowGroup<8, 1>& t, int rr) {
        const h16* p = a.Wqkv + (size_t)global_row(rr) * HID + lane * 8;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) t.w[0][jj] = ld_stream(p + jj * WAVE * 8);
    };
    p1_load(r0, rr0);
    // the ids of the group's 32 workgroups (lane i: member i % 32), requested right behind the first row (see k_fused_decode_g)
    const int mb = ((((b >> 3) / NS) * NS + (lane % NS)) << 3) | (b & 7);
    const u64 member_x = __hip_atomic_load(a.g_xcc + mb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    p1_load(r1, rr0 + 1);
    p1_load(r2, rr0 + 2);

    // ---- RMSNorm of every row, once per workgroup -> fp16 activation vectors in LDS ----------------------------------------------
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
    if (tid < n_idx) s_idx[tid] = idx_reg;
    for (int i = tid + 512; i < n_idx; i += 512) s_idx[i] = a.indices[ent0 + e0 + i];
    if (tid < 256) s_cs[tid] = cs_reg;
    if (tid == 0) s_ctl[20] = slot_reg;
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

    // ---- K/V tiles of this workgroup's (row, kv head, split), requested before q exists ------------------------------------------
    const size_t kvstride = (size_t)HKV * HEAD_DIM;
    const h16* kbase = kc + g * HEAD_DIM + d0;
    const h16* vbase = vc + g * HEAD_DIM + d0;
    const h16* dummy = a.na.rms_w + d0;
    auto load_tile = [&](KvTile32<4>& t, int tbase, auto far_c) {   // unconditional; a tile behind the slice reads one dummy line
        constexpr bool FAR = decltype(far_c)::value != 0;             // page numbers through L2 instead of the staged slice
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
            if constexpr (FAR) {
                rows[u] = ((size_t)a.indices[ent0 + (tk >> ps)] << ps) + (size_t)(tk & pmask);
            } else {
                int ei = (tk >> ps) - e0;
                ei = ei < GM::MAX_IDX ? ei : GM::MAX_IDX - 1;
                ei = ei > 0 ? ei : 0;
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
    lds_barrier();                      // xn of every row and the staged page numbers are in LDS
    // (4 row slots: the second tile goes out behind the dots -- three rows, two tiles, their 16 addresses and a chunk of four
    //  activation vectors do not fit 256 registers together: the allocator spilled freshly loaded tile data behind s_waitcnt vmcnt(0))
    constexpr bool TB_EARLY = NB <= 2;
    KvTile32<4> ta, tb;
    load_tile(ta, t0, NEAR);
    if constexpr (TB_EARLY) load_tile(tb, t0 + TILE, NEAR);

    // ---- phase 1: each row of Wqkv x every batch row -> granules g_qkv[row][kv head][768] -------------------------------------------
    const bool grp_local = __all((unsigned)(member_x >> 32) == epoch && (unsigned)member_x == xcc);
    {
        float acc[3][NB];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) acc[r][bb] = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) {
                const h16x8 av = *reinterpret_cast<const h16x8*>(s_a + (size_t)bb * HID + (i * WAVE + lane) * 8);
                acc[0][bb] = dot8h(r0.w[0][i], av, acc[0][bb]);
                acc[1][bb] = dot8h(r1.w[0][i], av, acc[1][bb]);
                acc[2][bb] = dot8h(r2.w[0][i], av, acc[2][bb]);
            }
            // (one chunk's activation vectors at a time: left alone the scheduler hoists all 8 x NB LDS reads -- 128 registers
            //  at 4 rows -- over the dots and spills the K/V tiles that are in flight)
            if constexpr (NB > 2) asm volatile("" ::: "memory");
        }
        u64* gq = a.g_qkv + (size_t)g * RG + rr0;
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) {
            const float v0 = sum64_lane63(acc[0][bb]), v1 = sum64_lane63(acc[1][bb]), v2 = sum64_lane63(acc[2][bb]);
            if (lane == 63 && bb < batch) {
                u64* p = gq + (size_t)bb * (HKV * RG);
                granule_store_to(p, epoch, v0, grp_local);
                granule_store_to(p + 1, epoch, v1, grp_local);
                granule_store_to(p + 2, epoch, v2, grp_local);
            }
        }
    }
    if constexpr (!TB_EARLY) load_tile(tb, t0 + TILE, NEAR);
    CF_TRACE(1);

    // ---- X1: q (4 heads) | k | v of (row, kv head) -----------------------------------------------------------------------------
    if (wave == 0) {
        bool ok = true;
        if (row_live) ok = sweep_granules<RG / 64>(a.g_qkv + ((size_t)row * HKV + g) * RG, RG, epoch, s_qkv, lane, a.state + 1, 1u);
        if (lane == 0) s_ctl[0] = ok;
    }
    lds_barrier();
    if (!s_ctl[0]) CF_FAIL_RETURN();
    CF_TRACE(2);

    // ---- RoPE'd, scaled q of the 4 heads -> fp16 in LDS, then the B operand of q.k ---------------------------------------------------
    const float qscale = 1.44269504088896340736f * 0.08838834764831845f;
    auto rope_lds = [&](const float* src, float (&dst)[8]) {
        if (a.rope_style == 0) {
            const float sgn = d0 < 64 ? -1.f : 1.f;
            const int a0 = d0 & 63, p0 = (d0 + 64) & 127;
#pragma unroll
            for (int e = 0; e < 8; ++e) dst[e] = src[d0 + e] * s_cs[a0 + e] + sgn * (src[p0 + e] * s_cs[128 + a0 + e]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float c = s_cs[d0 + e], s = s_cs[128 + d0 + e];
                dst[e] = (e & 1) ? src[d0 + e] * c + src[d0 + (e ^ 1)] * s : src[d0 + e] * c - src[d0 + (e ^ 1)] * s;
            }
        }
    };
    typedef h16 h16x4 __attribute__((ext_vector_type(4)));
    typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
    h16x8 qb[4];
    f32x4 oacc[8];
    float mfM = NEG_BIG, mfL = 0.f;
    {
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

    // ---- phase 2 on the matrix cores (the MF path of k_fused_decode_g: S = K q^T, softmax on the accumulator layout, O += P V) -----
    auto compute_tile = [&](const KvTile32<4>& t, int tbase) {
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
    compute_tile(ta, t0);
    RowGroup<8, 2> go;
    if (wave < 4) {      // (the two wavefronts of a SIMD do not stand at the request instructions together: k_fused_decode_g)
        go.load(a.Wo, 16 * b + 2 * wave, HID, HID, lane);
        compute_tile(tb, t0 + TILE);
    } else {
        compute_tile(tb, t0 + TILE);
        go.load(a.Wo, 16 * b + 2 * wave, HID, HID, lane);
    }
    // a row longer than MAX_TOKENS: the rest of the slice in 128-token tiles, two in flight (off the fast path: the branch sits
    // behind the last request of the straight-line code, so the wait counts before it stay exact)
    if (t0 + 2 * TILE < t1) {
        KvTile32<4> la, lb;
        const int tl = t0 + 2 * TILE;
        if (n_need <= max_idx) {
            load_tile(la, tl, NEAR);
            for (int tt = tl; tt < t1; tt += 2 * TILE) {
                load_tile(lb, tt + TILE, NEAR);
                compute_tile(la, tt);
                load_tile(la, tt + 2 * TILE, NEAR);
                compute_tile(lb, tt + TILE);
            }
        } else {
            load_tile(la, tl, FARIDX);
            for (int tt = tl; tt < t1; tt += 2 * TILE) {
                load_tile(lb, tt + TILE, FARIDX);
                compute_tile(la, tt);
                load_tile(la, tt + 2 * TILE, FARIDX);
                compute_tile(lb, tt + TILE);
            }
        }
    }
    {   // one state per wavefront and head: M uniform over the 4 token groups, L their sum, O in accumulator rows r = head of lanes 0..15
        const float lw = xsum32(xsum16(mfL));
        if (lane < G) { s_ml[lane][wave][0] = mfM; s_ml[lane][wave][1] = lw; }
        if (lane < 16) {
#pragma unroll
            for (int jb = 0; jb < 8; ++jb)
#pragma unroll
                for (int r = 0; r < G; ++r) s_o[r][wave][16 * jb + lane] = oacc[jb][r];
        }
    }
    // the new token of this row (attended from registers) + k/v export + cache write: split 0 of (row, kv head)
    if (row_live && js == 0 && gid == 0) {
        float kf[8], vf[8];
        rope_lds(s_qkv + G * HEAD_DIM, kf);
#pragma unroll
        for (int e = 0; e < 8; ++e) vf[e] = s_qkv[(G + 1) * HEAD_DIM + d0 + e];
        h16x8 k16, v16;
#pragma unroll
        for (int e = 0; e < 8; ++e) { k16[e] = (h16)kf[e]; v16[e] = (h16)vf[e]; }
        const size_t ooff = (size_t)g * HEAD_DIM + d0;
        if (a.k_new) st_h8(a.k_new + (size_t)row * kvstride + ooff, k16);
        if (a.v_new) st_h8(a.v_new + (size_t)row * kvstride + ooff, v16);
        if (a.write_cache) {
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
    lds_barrier();
    CF_TRACE(3);

    // ---- X2: 4 records per workgroup (one per q head) -> the leader of (row, q head): split js = head index within the group --------
    constexpr int RH = FUSED_RECH, RM = HEAD_DIM / 2, RL = HEAD_DIM / 2 + 1;
    auto rec_of = [&](int hh, int split) -> u64* { return a.g_rec + ((((size_t)row * HKV + g) * G + hh) * NSP + split) * RH; };
    if (row_live) {
        const int nst = js == 0 ? NST : NST - 1;
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
                granule_store_to(rec_of(hh, js) + RM, epoch, M, grp_local);
                granule_store_to(rec_of(hh, js) + RL, epoch, L, grp_local);
            }
        }
    }
    lds_barrier();
    if (row_live) {
        const int nst = js == 0 ? NST : NST - 1;
        for (int t = tid; t < G * HEAD_DIM; t += 512) {
            const int hh = t >> 7, d = t & 127;
            float val = 0.f;
#pragma unroll
            for (int w = 0; w < NST; ++w) val = __builtin_fmaf(s_w[hh][w], w < nst ? s_o[hh][w][d] : 0.f, val);
            const float next = __shfl_down(val, 1);
            h16x2 pr;
            pr[0] = (h16)val;
            pr[1] = (h16)next;
            if (!(d & 1)) granule_store_to(rec_of(hh, js) + (d >> 1), epoch, __builtin_bit_cast(float, pr), grp_local);
        }
    }
    if (row_live && js < G) {   // leader of q head g * 4 + js of this row: wavefront w gathers NSP / 8 records, then the softmax merge
        lds_barrier();          // s_recu reuses s_o: every wavefront is done reading the states
        constexpr int CNT = GM::RECW * RH;
        const bool ok = sweep_granules_raw<(CNT + 63) / 64>(rec_of(js, wave * GM::RECW), CNT, epoch, s_recu + wave * CNT, lane, a.state + 1, 2u);
        if (lane == 0) s_ctl[1 + wave] = ok;
        lds_barrier();
        bool all_ok = true;
        for (int w = 0; w < 8; ++w) all_ok &= s_ctl[1 + w] != 0;
        if (!all_ok) CF_FAIL_RETURN();
        if (tid < HEAD_DIM) {
            float M = NEG_BIG;
#pragma unroll
            for (int w = 0; w < NSP; ++w) M = fmaxf(M, __builtin_bit_cast(float, s_recu[w * RH + RM]));
            float acc = 0.f, L = 0.f;
#pragma unroll
            for (int w = 0; w < NSP; ++w) {
                const float wt = fast_exp2(__builtin_bit_cast(float, s_recu[w * RH + RM]) - M) * __builtin_bit_cast(float, s_recu[w * RH + RL]);
                acc = __builtin_fmaf(wt, (float)__builtin_bit_cast(h16x2, s_recu[w * RH + (tid >> 1)])[tid & 1], acc);
                L += wt;
            }
            const float mine = L > 0.f ? acc / L : 0.f, next = __shfl_down(mine, 1);
            h16x2 pr;
            pr[0] = (h16)mine;
            pr[1] = (h16)next;
            // layout [wavefront chunk = kv head][row slot][4 heads x 64]: what one wavefront of a consumer gathers is contiguous
            if (!(tid & 1)) granule_store(a.g_attn + ((size_t)g * NB + row) * 256 + js * 64 + (tid >> 1), epoch, __builtin_bit_cast(float, pr));
        }
    }
    CF_TRACE(4);

    // ---- X3: the attention outputs of all rows: batch x 2048 granules (fp16 pairs), 256 per wavefront and row -------------------------
    {
        constexpr int NG = 4 * NB;
        const u64* gp = a.g_attn + (size_t)wave * NB * 256;
        const int count = batch * 256;
        for (unsigned spin = 0; spin < FUSED_SPIN_LIMIT; ++spin) {   // hint: the last granule of each (row, head) of this chunk
            u64 x = (u64)epoch << 32;
            if (lane < 4 * batch) x = __hip_atomic_load(gp + lane * 64 + 63, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__popcll(__ballot((unsigned)(x >> 32) != epoch)) <= 2) break;
            __builtin_amdgcn_s_sleep(2);
        }
        unsigned v[NG];
        bool ok = true;
        for (unsigned spin = 0;; ++spin) {
            bool good = true;
#pragma unroll
            for (int k = 0; k < NG; ++k) {
                const int i = lane + WAVE * k;
                u64 x = (u64)epoch << 32;
                if (i < count) x = __hip_atomic_load(gp + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
        // (phase 1's activation vectors in s_a are dead since the X1 barrier)
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
    // ---- phase 3: 16 rows of Wo per workgroup x every batch row -------------------------------------------------------------------
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
        a.state[2] = 0u;      // (no length arm to report)
    }
    CF_TRACE(6);
}

}  // namespace cf
