// cf_fused_kernel_g.h -- the persistent fused decode kernel generalised over the head geometry:
// HKV local kv heads, G = q heads per kv head ([out,in] weights, hidden 4096, batch 1).
//
//   <8, 4>   Llama-3-8B (32 q / 8 kv heads, BASELINE config 4): 32 workgroups per kv head
//   <16, 1>  one rank of a 2-way head-parallel shard of Llama-2-7B: 16 workgroups per head
//   <8, 1>   ... of a 4-way shard: 32 workgroups per head
//   <4, 1>   ... of an 8-way shard (BASELINE config 5): 64 workgroups per head (a head spans 2 XCDs)
//   <4, 4>, <2, 4>, <1, 4>   one rank of a 2 / 4 / 8-way head-parallel shard of Llama-3-8B (16q/4kv, 8q/2kv, 4q/1kv: configs 4
//            and 5 composed): 64 / 128 / 256 workgroups per kv head (a group spans 2 / 4 / 8 XCDs), two-level record merge
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

#ifndef CF_G_P3ATT
#define CF_G_P3ATT 1
#endif
#ifndef CF_G_P3ATT_MIN_AS
#define CF_G_P3ATT_MIN_AS 4
#endif
#ifndef CF_G_X1_ALL
#define CF_G_X1_ALL 0
#endif
#ifndef CF_G_ATT_STRIDE
#define CF_G_ATT_STRIDE 1
#endif
#ifndef CF_G_LEADERLESS_MAX
#define CF_G_LEADERLESS_MAX 32     // merged records every workgroup gathers itself instead of waiting for a leader's X3 (see X2);
                                   // 64 (the 8q/2kv shard leaderless, too) measured 17.9 vs 16.1-17.2 us: not kept
#endif

namespace cf {

template <int HKV, int G>
struct FusedGeom {
    static constexpr int HQ = HKV * G;
    static constexpr int NS = FUSED_WGS / HKV;            // workgroups per kv head
    static constexpr int RG = (G + 2) * HEAD_DIM;         // projection rows of one kv-head group
    static constexpr int RPW = RG / NS;                   // ... per workgroup
    static constexpr int RPWV = 3;                        // projection rows per streaming wavefront.  (One row per wavefront where a
                                                          // workgroup has <= 6 rows -- RPWV = 1 -- ends phase 1 0.6 us earlier and X1
                                                          // 0.7 us LATER: more wavefronts with rows in flight in front of the poll.
                                                          // 4q/1kv shard 14.95 -> 15.5 us, 8q/2kv 15.9 -> 16.7: round 4, not kept.)
    static constexpr int P1_WAVES = RPW / RPWV;           // wavefronts that stream projection rows
    static constexpr bool TWO = G > 1 && NS <= 64;        // grouped-query: two half tiles, so the Wo rows can be requested
                                                          // between them (their issue overlaps nothing otherwise); with 128 / 256
                                                          // workgroups per kv head a slice is <= 128 tokens up to S = 16 k / 32 k:
                                                          // a second pre-requested tile would be clamped duplicates (64 KB per CU)
    static constexpr int U = G == 4 ? 4 : (NS <= 16 ? 8 : (NS == 32 ? 4 : 2));   // token rows per lane-group of a tile
    static constexpr int SHORT_TOKENS = (CF_G_ATT_STRIDE && NS >= 128 ? 64 : NS) * 32 * U * (TWO ? 2 : 1);   // straight-line variant covers this
    static constexpr int JO = HQ * HEAD_DIM / 512;        // 1-KB pieces of one Wo row
    static constexpr int RECW = NS / 8;                   // records one wavefront of a leader sweeps
    // LDS carve
    static constexpr int L_QKV = 0;                                    // float[RG]
    static constexpr int L_A = L_QKV + RG * 4;                         // float[4096] (x, then attention out)
    static constexpr bool MF = G == 4;                                 // phase 2 on the matrix cores (see compute_tile_mf)
    static constexpr int NST = 9;                                      // softmax states per q head: 8 wavefronts + the new token
    // two-level record merge (NS >= 64): sub-groups of SG = NS / 8 consecutive workgroups (always inside one XCD), i.e. 8
    // merged records per q head whatever the geometry
    // NSA of the NS workgroups of a kv-head group take part in the attention (token slices, records): all of them up to 64; with
    // 128 / 256 workgroups per group every AS-th one (a 32-token slice in a 128-token MFMA tile is three quarters padding, and
    // a sub-leader that sweeps 32 records waits for the slowest of 32)
    static constexpr int NSA = CF_G_ATT_STRIDE && NS >= 128 ? 64 : NS, AS = NS / NSA;
    static constexpr int SG = NSA >= 64 ? NSA / 8 : 8, NSG = NSA / SG;
    static constexpr bool TREE = NSA >= 64, LEADERLESS = TREE && HQ * NSG <= CF_G_LEADERLESS_MAX;      // (see X2 in the kernel)
    // records a workgroup gathers into LDS: a flat leader all NS of its head; with the tree SG level-1 records + the NSG merged
    // ones of a head (leaderless: the merged records of ALL heads).  (NS = 64 keeps its round-3 size.)
    static constexpr int REC_N = !TREE || NS == 64 ? NS : (LEADERLESS && HQ * NSG > SG + NSG ? HQ * NSG : SG + NSG);
    static_assert(REC_N >= (TREE ? SG + NSG : NS), "record area");
    static constexpr int O_BYTES = G * NST * HEAD_DIM * 4, REC_BYTES = REC_N * FUSED_RECH * 4;
    static constexpr int L_O = L_A + 4096 * 4;                         // float[G][NST][128]; later unsigned[NS][FUSED_RECH]
    static constexpr int L_REC = L_O;                                  //   (the leader's gathered records reuse it)
    static constexpr int L_ML = L_O + (O_BYTES > REC_BYTES ? O_BYTES : REC_BYTES);   // float[G][NST][2]
    static constexpr int L_W = L_ML + ((G * NST * 2 * 4 + 15) & ~15);  // float[G][NST] merge weights
    static constexpr int L_QH = L_W + ((G * NST * 4 + 15) & ~15);      // MF: h16[G][128] RoPE'd, scaled q
    static constexpr int L_VT = L_QH + (MF ? G * HEAD_DIM * 2 : 0);    // MF: h16[8 wavefronts][8][16][16] V tiles for tr reads
    static constexpr int KT_ROW = 136;                                 // MF: K image row = 128 dims + 8 pad (272 B: conflict-free)
    static constexpr int L_KT = L_VT + (MF ? 8 * 4096 : 0);            // MF: h16[8 wavefronts][16 tokens][KT_ROW]
    static constexpr int L_IDX = L_KT + (MF ? 8 * 16 * KT_ROW * 2 : 0);   // int[MAX_IDX]
    static constexpr int MAX_IDX = MF ? 8192 : FUSED_MAX_IDX;          // page-table entries one workgroup stages
    static constexpr int L_CS = L_IDX + MAX_IDX * 4;                   // float[256]
    static constexpr int L_CTL = L_CS + 256 * 4;                       // int[32]
    static constexpr int L_END = L_CTL + 128;
    static constexpr int LDS_BYTES = L_END > 84 * 1024 ? L_END : 84 * 1024;
    static_assert(RPW % RPWV == 0 && P1_WAVES >= 1 && P1_WAVES <= 8, "1 or 3 projection rows per streaming wavefront");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS carve exceeds a CU");
};

// One kernel per geometry for every cached length (as k_fused_decode_mha): the tiles requested before X1 are common; after
// the first of them is consumed a wave-uniform branch on the device-side length picks the straight-line rest (the slice fits
// the pre-requested tiles; Wo rows staggered around tile B) or the loop rest (further 128-token tiles two deep, Wo after the
// loop; page numbers beyond the staged part of the table through L2).  The two copies never join: exact wait counts in both.
template <int HKV, int G>
__global__ __launch_bounds__(512, 2) void k_fused_decode_g(FusedArgs a) {
    using GM = FusedGeom<HKV, G>;
    constexpr int HQ = GM::HQ, NS = GM::NS, RG = GM::RG, JO = GM::JO, HID = 4096, U = GM::U;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_qkv = reinterpret_cast<float*>(smem + GM::L_QKV);
    float* s_a = reinterpret_cast<float*>(smem + GM::L_A);
    constexpr int NST = GM::NST;
    float(*s_o)[NST][HEAD_DIM] = reinterpret_cast<float(*)[NST][HEAD_DIM]>(smem + GM::L_O);
    float(*s_ml)[NST][2] = reinterpret_cast<float(*)[NST][2]>(smem + GM::L_ML);
    float(*s_w)[NST] = reinterpret_cast<float(*)[NST]>(smem + GM::L_W);
    float* s_rec = reinterpret_cast<float*>(smem + GM::L_REC);
    int* s_idx = reinterpret_cast<int*>(smem + GM::L_IDX);
    float* s_cs = reinterpret_cast<float*>(smem + GM::L_CS);
    int* s_ctl = reinterpret_cast<int*>(smem + GM::L_CTL);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wavefront-uniform
    const int l16 = lane & 15, gid = wave * 4 + (lane >> 4), d0 = l16 * 8;
    const int b = blockIdx.x;
    // the NS workgroups of a kv head share b % 8 (one XCD hosts 32 / NS whole groups); with 64
    // workgroups per head a head spans the XCD pair (2p, 2p+1)
    // One group per XCD (NS = 32): XCD x hosts kv head x ^ 1.  Two systematic lags exist -- odd XCDs stream their phase-1 rows
    // ~1.2 us late (a position effect), and the kv heads 1 and 5, whose 256-byte K/V pieces sit at (address >> 8) & 3 = 1, stream
    // ~17 % slower (tools/ubench/kv_bw.hip) -- and with head x on XCD x they met on XCDs 1 and 5.  Swapped, the slow heads start
    // phase 2 on the XCDs whose X1 resolves first.  (CF_G_XCD_SWAP=0: the round-2 map, for A/B.)
#ifndef CF_G_XCD_SWAP
#define CF_G_XCD_SWAP 1
#endif
    constexpr int XPG = NS <= 32 ? 1 : NS / 32;      // XCDs one kv-head group spans (64 / 128 / 256 workgroups per group: 2 / 4 / 8)
    const int g = NS == 32 ? ((b & 7) ^ CF_G_XCD_SWAP)
                : NS < 32  ? (b & 7) * (32 / (NS <= 32 ? NS : 32)) + (b >> 3) / NS
                           : (b & 7) / XPG;
    const int j = NS <= 32 ? (b >> 3) % NS : (b >> 3) + 32 * ((b & 7) % XPG);
    CF_TRACE(0);
    // Where this workgroup runs.  With NS <= 32 all workgroups of a kv-head group are meant to share an XCD (b % 8):
    // when the published ids confirm it, the group's hand-offs X1 and X2 stay inside that XCD's L2 (plain stores)
    // instead of being written through to memory -- the fabric round trips queue behind the weight streams.
    const unsigned xcc = my_xcc_id();
    constexpr bool CAN_LOCAL = NS <= 32;

    // ---- small first-level loads first (loads return in issue order) -------------------------------
    const h16* rp = a.na.residual ? a.na.residual : a.na.x;
    const float rs = a.na.residual ? 1.f : 0.f;
    const h16x8 xv = ld_h8(a.na.x + tid * 8), rv = ld_h8(rp + tid * 8), wv8 = ld_h8(a.na.rms_w + tid * 8);
#ifndef CF_G_SCALAR_BATCH
#define CF_G_SCALAR_BATCH 1      // the start values in ONE batch of unconditional scalar loads (cf_fused_kernel.h; 0: one load behind every `if (pointer)`)
#endif
    unsigned epoch, tp_epoch;
    int S = a.seq_len, ent0 = 0;
    int64_t roff = 0;
    const h16 *kc = a.k_cache, *vc = a.v_cache;
    if constexpr (CF_G_SCALAR_BATCH) {
        const uint32_t* stp = a.state;      // (an absent table is read from the workspace's state words: always mapped, the value is discarded)
        const int32_t* ipp = a.indptr ? a.indptr : reinterpret_cast<const int32_t*>(stp);
        const int32_t* slp = a.seq_lens ? a.seq_lens : reinterpret_cast<const int32_t*>(stp);
        const int64_t* pop = a.positions ? a.positions : reinterpret_cast<const int64_t*>(stp);
        const uint64_t* kpp = a.kptrs ? a.kptrs + a.layer_id : reinterpret_cast<const uint64_t*>(stp);
        const uint64_t* vpp = a.vptrs ? a.vptrs + a.layer_id : reinterpret_cast<const uint64_t*>(stp);
        const uint32_t* tpp = a.tp_world > 0 ? reinterpret_cast<const uint32_t*>(a.tp_areas[a.tp_rank]) : stp;
        const unsigned ep0 = scalar_load(stp), tp0 = scalar_load(tpp);
        const int ip0 = scalar_load(ipp), ip1 = scalar_load(ipp + 1), sl0 = scalar_load(slp);
        const int64_t po0 = scalar_load(pop);
        const uint64_t kp0 = scalar_load(kpp), vp0 = scalar_load(vpp);
        epoch = ep0 + 1u;      // (written by the previous launch: the scalar cache is invalidated at every kernel start)
        tp_epoch = a.tp_world > 0 ? tp0 + 1u : 0u;
        if (a.indptr) {
            ent0 = ip0;
            S = a.seq_lens ? sl0 : ip1 - 1 - ent0;
        }
        if (a.positions) roff = po0 * a.rope_stride;
        if (a.kptrs) kc = reinterpret_cast<const h16*>(kp0);
        if (a.vptrs) vc = reinterpret_cast<const h16*>(vp0);
    } else {
        epoch = scalar_load(a.state) + 1u;
        tp_epoch = tp_call_epoch(a);
        if (a.indptr) {
            ent0 = scalar_load(a.indptr);
            S = a.seq_lens ? scalar_load(a.seq_lens) : scalar_load(a.indptr + 1) - 1 - ent0;
        }
        if (a.positions) roff = scalar_load(a.positions) * a.rope_stride;
        if (a.kptrs) kc = reinterpret_cast<const h16*>(scalar_load(a.kptrs + a.layer_id));
        if (a.vptrs) vc = reinterpret_cast<const h16*>(scalar_load(a.vptrs + a.layer_id));
    }
    if (tid == 0) granule_store(a.g_xcc + b, epoch, __builtin_bit_cast(float, xcc));

    // ---- second-level loads (page-table slice, new-token slot, RoPE row): registers first, LDS later ----------
    const int ps = a.page_shift, pmask = (1 << ps) - 1;
    constexpr int NSA = GM::NSA, AS = GM::AS;
    const bool att = AS == 1 || (j % AS) == 0;      // (workgroup-uniform) holds a token slice and publishes records
    const int ja = j / AS;
    int tps = ((S + NSA - 1) / NSA + 31) & ~31;
    tps = tps < 32 ? 32 : tps;
    const int t0 = att ? ja * tps : 0;
    int t1 = t0 + tps;
    t1 = t1 < S ? t1 : S;
    t1 = att ? t1 : t0;                             // (the others: an empty slice -- their tile requests read one dummy line)
    const int e0 = t0 >> ps;
    const int max_idx = (a.flags & 64) ? 512 : GM::MAX_IDX;   // (debug bit 64: stage only what the pre-requested tiles need)
    int n_idx = 0, n_need = 0;     // page-table entries of this slice: all of them / those staged in LDS
    if (a.indptr && t1 > t0) {
        n_need = ((t1 - 1) >> ps) - e0 + 1;
        n_idx = n_need < max_idx ? n_need : max_idx;   // (a longer slice reads the rest through L2)
    }
    int idx_reg = 0, slot_reg = 0;
    float cs_reg = 0.f;
    auto second_level_loads = [&]() {
        if (tid < n_idx) idx_reg = a.indices[ent0 + e0 + tid];
        if (a.indptr && tid == 0) slot_reg = a.indices[ent0 + (S >> ps)];
        const int n_ang = a.rope_style == 0 ? HEAD_DIM / 2 : HEAD_DIM;
        if (tid < n_ang) cs_reg = a.cos[roff + tid];
        else if (tid >= 128 && tid < 128 + n_ang) cs_reg = a.sin[roff + tid - 128];
    };
    // ---- K/V tiles requested before q exists ----------------------------------------------------------
    const size_t kvstride = (size_t)HKV * HEAD_DIM;
    const h16* kbase = kc + g * HEAD_DIM + d0;
    const h16* vbase = vc + g * HEAD_DIM + d0;
    const h16* dummy = a.na.rms_w + d0;
    constexpr bool MF = GM::MF;
    // (MF loads K/V exactly like the VALU path -- 16 lanes x 16 B = one token's 256-B strip, 4 tokens per
    //  instruction; requesting them in MFMA operand shape, 16 tokens x 64 B per instruction, made the K/V
    //  stream land 2 us later -- and re-shapes them through LDS)
    auto load_tile = [&](auto& t, int tbase, auto far_c) {   // unconditional; a tile behind the slice reads one dummy line
        constexpr int UU = sizeof(t.k) / sizeof(h16x8);
        constexpr bool FAR = decltype(far_c)::value != 0;  // page numbers through L2 instead of the staged slice
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
    constexpr int TILE = MF ? 128 : 32 * U;             // MF: 16 tokens per wavefront and tile
    constexpr int UL = 4, TILE_L = MF ? 128 : 32 * UL;
    constexpr bool TWO = GM::TWO;
    constexpr FusedArm<0> NEAR{};
    constexpr FusedArm<1> FARIDX{};
    // The page table is requested ahead of the projection rows and the K/V tiles go out as soon as it is staged -- before the
    // RMSNorm -- instead of after phase 1: the CU's request stream never has to wait for a row to be consumed before the
    // next bytes are asked for.  On the small shards (a latency chain, not a byte stream) X1 then waits for the slowest
    // producer only, not for tiles queued in front of its polling loads: TP-8 shard -0.8 us, TP-2 -0.6 us, GQA -0.25 us.
    constexpr bool EARLY_KV = true;
    KvTile32<U> ta;
    KvTile32<GM::TWO ? U : 1> tb;
    if constexpr (EARLY_KV) second_level_loads();
    // ---- phase-1 weight stream: 3 rows of the group's q|k|v row space per wavefront -----------------
    const int rr0 = GM::RPW * j + GM::RPWV * wave;   // first row (inside the group) of this wavefront
    auto global_row = [&](int rr) -> int {           // Wqkv rows: q of all heads | k | v
        if (rr < G * HEAD_DIM) return g * G * HEAD_DIM + rr;
        if (rr < (G + 1) * HEAD_DIM) return HQ * HEAD_DIM + g * HEAD_DIM + (rr - G * HEAD_DIM);
        return (HQ + HKV) * HEAD_DIM + g * HEAD_DIM + (rr - (G + 1) * HEAD_DIM);
    };
    constexpr int NROWS = (HQ + 2 * HKV) * HEAD_DIM;
    RowGroup<8, 1> r0, r1, r2;
    const bool p1w = wave < GM::P1_WAVES;   // small shards: few rows per workgroup
    // (unconditional requests: the other wavefronts read one dummy line -- a branch around the loads would
    //  put a control-flow join before the next use and make the compiler wait for everything in flight)
    auto p1_load = [&](RowGroup<8, 1>& t, int rr) {
        if constexpr (GM::P1_WAVES == 8) {      // every wavefront streams rows
            const h16* p = a.Wqkv + (size_t)global_row(rr) * HID + lane * 8;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) t.w[0][jj] = ld_stream(p + jj * WAVE * 8);
        } else if (p1w) {                       // small shards: the idle wavefronts skip even the dummy lines
            const h16* p = a.Wqkv + (size_t)global_row(rr) * HID + lane * 8;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) t.w[0][jj] = ld_stream(p + jj * WAVE * 8);
        } else {
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) t.w[0][jj] = h16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    };
    // the ids of the group's NS workgroups (lane i: member i) are requested right behind the FIRST row (loads return in
    // issue order): back in time for the first publish, yet late enough (~2 us into the kernel) that every member's id --
    // published in its first instructions -- is visible under graph replay.  A wavefront that still misses one writes
    // its results through.
    p1_load(r0, rr0);
    u64 member_x = 0;
    if constexpr (CAN_LOCAL) {
        const int mb = ((((b >> 3) / NS) * NS + (lane % NS)) << 3) | (b & 7);
        member_x = __hip_atomic_load(a.g_xcc + mb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {   // >= 64 workgroups per head (2 .. 8 XCDs): only the SG workgroups of a merge sub-group (consecutive j) share one
        const int jm = ((ja & ~(GM::SG - 1)) | (lane & (GM::SG - 1))) * AS;
        member_x = __hip_atomic_load(a.g_xcc + (((jm & 31) << 3) | (g * XPG + (jm >> 5))), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if constexpr (GM::RPWV == 3) {
        p1_load(r1, rr0 + 1);
        p1_load(r2, rr0 + 2);
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

    auto stage_second_level = [&]() {
        if (tid < n_idx) s_idx[tid] = idx_reg;
        for (int i = tid + 512; i < n_idx; i += 512) s_idx[i] = a.indices[ent0 + e0 + i];
        if (tid < 256) s_cs[tid] = cs_reg;
        if (tid == 0) s_ctl[20] = slot_reg;
    };
    if constexpr (!EARLY_KV) second_level_loads();
    else stage_second_level();      // (visible after the barrier below, together with the partial sums of squares)
    lds_barrier();
    if constexpr (EARLY_KV) {
        load_tile(ta, t0, NEAR);
        if constexpr (GM::TWO) load_tile(tb, t0 + TILE, NEAR);
    }
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
    bool grp_local = false;    // (per wavefront; needs every member's id of THIS call)
    bool rec_local = false;
    {
        float res[1];
        r0.dot(xn, res);
        const bool members_here = __all((unsigned)(member_x >> 32) == epoch && (unsigned)member_x == xcc);
        if constexpr (CAN_LOCAL) grp_local = members_here;
        rec_local = members_here;      // (the level-1 records of X2 stay inside the sub-group)
        if (p1w && lane == 63) granule_store_to(gq, epoch, res[0], grp_local);
        if constexpr (GM::RPWV == 3) {
            r1.dot(xn, res);
            if (p1w && lane == 63) granule_store_to(gq + 1, epoch, res[0], grp_local);
        }
    }
    if constexpr (!EARLY_KV) {
        stage_second_level();
        lds_barrier();
    }

    if constexpr (!EARLY_KV) load_tile(ta, t0, NEAR);
    if constexpr (GM::RPWV == 3) {
        float res[1];
        r2.dot(xn, res);
        if (p1w && lane == 63) granule_store_to(gq + 2, epoch, res[0], grp_local);
    }
    if constexpr (TWO && !EARLY_KV) load_tile(tb, t0 + TILE, NEAR);   // (both half tiles land before X1 can resolve: it waits ~2 us
                                                   //  for the slowest producer's rows to become visible anyway)

    CF_TRACE(1);
    // Groups where only every 4th workgroup holds a token slice (4q/1kv): those workgroups also run the O projection (64 rows
    // each), and the others are done here -- a workgroup that idles into a poll loop beside the chain costs the chain microseconds
    // (X1 sweeps by the idle ones: 4q/1kv 15.75 vs 14.25 us; the role-split shard kernel: 12.9 vs 12.2).  Measured A/B on one
    // box: 4q/1kv 14.18-14.27 vs 14.29-14.32 us (kept); 8q/2kv (AS = 2, 32 rows each) 16.2 vs 15.2-15.8 us: the longer phase 3
    // of half the workgroups costs more than the idle half's polls -- there every workgroup keeps its 16 rows.
    constexpr bool P3ATT = AS >= CF_G_P3ATT_MIN_AS && CF_G_P3ATT;
    constexpr int P3R = P3ATT ? HID / (NSA * HKV) / 8 : 2;      // rows of Wo per wavefront
    if constexpr (P3ATT) {
        if (!att) {
            CF_TRACE(6);
            return;
        }
    }
    const int ai = P3ATT ? g * NSA + ja : b;                    // index of this workgroup among those that run phase 3
    // ---- X1: q (G heads) | k | v of this kv-head group --------------------------------------------------
    if (wave == 0) {
        // (a workgroup without a token slice needs no q|k|v: it does not poll -- CF_G_X1_ALL = 1: everybody sweeps, for A/B)
        const bool ok = (att || CF_G_X1_ALL) ? sweep_granules<RG / 64>(a.g_qkv + (size_t)g * RG, RG, epoch, s_qkv, lane, a.state + 1, 1u) : true;
        if (lane == 0) s_ctl[0] = ok;
    }
    lds_barrier();
    if (!s_ctl[0]) CF_FAIL_RETURN();
    CF_TRACE(2);
    constexpr int LO = HQ * HEAD_DIM;

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
    h16x8 qh[G];   // q rounded to fp16 (as the reference keeps it)
    float m[G], l[G], o[G][8];
    // MF state: per lane (n = lane % 16: q head n if n < G; kq = lane / 16) the running max / sum of head n over
    // the tokens 4 kq .. 4 kq + 3 of every tile; O[head][dim] in the MFMA accumulators of lanes 0..15
    typedef h16 h16x4 __attribute__((ext_vector_type(4)));
    typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
    h16x8 qb[4];
    f32x4 oacc[8];
    float mfM = NEG_BIG, mfL = 0.f;
    h16* s_qh = reinterpret_cast<h16*>(smem + GM::L_QH);
    h16* s_vt = reinterpret_cast<h16*>(smem + GM::L_VT) + wave * 2048;     // this wavefront's [8][16][16] V image
    h16* s_kt = reinterpret_cast<h16*>(smem + GM::L_KT) + wave * 16 * GM::KT_ROW;   // ... and its [16][128 + pad] K image
    if constexpr (MF) {
        // RoPE'd, scaled q of the G heads -> fp16 in LDS (one element per thread), then the B operand of q.k:
        // lane (n, kq): q[head n][32 s + 8 kq .. + 8), zero columns for n >= G
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
    } else {
#pragma unroll
        for (int hh = 0; hh < G; ++hh) {
            rope_lds(s_qkv + hh * HEAD_DIM, q[hh]);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                q[hh][e] *= qscale;
                qh[hh][e] = (h16)q[hh][e];
            }
            m[hh] = NEG_BIG;
            l[hh] = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[hh][e] = 0.f;
        }
    }

    // ---- phase 2 ----------------------------------------------------------------------------------------
    // MF: S = K q^T on v_mfma_f32_16x16x32_f16 (16 tokens x 16 columns, G real), softmax on the accumulator
    //     layout (lane (n, kq): tokens 4 kq + r of head n), P straight back in as the A operand of
    //     O += P V on v_mfma_f32_16x16x16_f16; V needs k = tokens contiguous per lane, so its tile goes
    //     through LDS as 8 x [16 tokens][16 dims] images read with ds_read_b64_tr_b16.
    // else: every K/V row is scored against the q head on the VALU (memory-bound shards).
    auto compute_tile = [&](const auto& t, int tbase) {
        constexpr int UU = sizeof(t.k) / sizeof(h16x8);
        if constexpr (MF) {
            // the wavefront's 16 tokens: MFMA row 4 u + lg <-> token tbase + 32 u + 4 wave + lg (as loaded)
            // K tile -> LDS [16 tokens][128 dims (+pad)], V tile -> LDS [jb = dim / 16][token][dim % 16]
            const int lg = lane >> 4;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                *reinterpret_cast<h16x8*>(s_kt + (4 * u + lg) * GM::KT_ROW + l16 * 8) = t.k[u];
                *reinterpret_cast<h16x8*>(s_vt + (l16 >> 1) * 256 + (4 * u + lg) * 16 + (l16 & 1) * 8) = t.v[u];
            }
            asm volatile("" ::: "memory");                               // images written before they are read back
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 4; ++u) {      // A operand: lane (row l16, k-group lg): K[row][32 u + 8 lg .. + 8)
                const h16x8 ka = *reinterpret_cast<const h16x8*>(s_kt + l16 * GM::KT_ROW + 32 * u + lg * 8);
                d = __builtin_amdgcn_mfma_f32_16x16x32_f16(ka, qb[u], d, 0, 0, 0);
            }
            const int tok0 = tbase + lg * 32 + wave * 4;                 // accumulator row 4 lg + r <-> token tok0 + r
            float mx = NEG_BIG;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                d[r] = (tok0 + r) < t1 ? d[r] : NEG_BIG;
                mx = fmaxf(mx, d[r]);
            }
            mx = xmax32(xmax16(mx));                                     // the 4 token groups of head n
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
            // rescale O: accumulator row r of lanes 0..15 is head r
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
            asm volatile("" ::: "memory");                               // ... and read before the next tile overwrites them
            return;
        }
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
    CF_TRACE(7);
    compute_tile(ta, t0);       // (a tile behind the slice is all-masked: state unchanged)
    CF_TRACE(8);
    // ================= from here on: one straight copy per arm =================================================================
    auto rest = [&](auto long_c) {
    constexpr bool LONG = decltype(long_c)::value != 0;
    RowGroup<JO, P3R> go;
    if constexpr (LONG) {
        KvTile32<UL> la, lb;
        const int tl = t0 + (TWO ? 2 : 1) * TILE;
        if (n_need <= max_idx) {            // (workgroup-uniform) the whole slice of the page table is staged in LDS
            load_tile(la, tl, NEAR);
            if constexpr (TWO) compute_tile(tb, t0 + TILE);
            for (int tt = tl; tt < t1; tt += 2 * TILE_L) {
                load_tile(lb, tt + TILE_L, NEAR);
                compute_tile(la, tt);
                load_tile(la, tt + 2 * TILE_L, NEAR);
                compute_tile(lb, tt + TILE_L);
            }
        } else {
            load_tile(la, tl, FARIDX);
            if constexpr (TWO) compute_tile(tb, t0 + TILE);
            for (int tt = tl; tt < t1; tt += 2 * TILE_L) {
                load_tile(lb, tt + TILE_L, FARIDX);
                compute_tile(la, tt);
                load_tile(la, tt + 2 * TILE_L, FARIDX);
                compute_tile(lb, tt + TILE_L);
            }
        }
        go.load(a.Wo, (8 * ai + wave) * P3R, HID, LO, lane);
    } else {
        // phase-3 rows: in flight through X2 / X3.  (Grouped-query: their issue -- 16 KB per wavefront through
        // a 64 B/clk address path -- overlaps the latency of tile B instead of delaying tile A's arithmetic.)
        // Two straight-line copies: wavefronts 0-3 request before tile B, 4-7 after it -- the two wavefronts of a SIMD do not
        // stand at the (slow) request instructions together (see k_fused_decode_mha).
        if (!TWO || wave < 4) {
            go.load(a.Wo, (8 * ai + wave) * P3R, HID, LO, lane);
            if constexpr (TWO) compute_tile(tb, t0 + TILE);
        } else {
            if constexpr (TWO) compute_tile(tb, t0 + TILE);
            go.load(a.Wo, (8 * ai + wave) * P3R, HID, LO, lane);
        }
    }
    CF_TRACE(9);

    if constexpr (MF) {
        // one state per wavefront and head: M is uniform over the 4 token groups, L is their sum, O sits in the
        // accumulator rows r = head of lanes 0..15 (dims 16 jb + lane)
        const float lw = xsum32(xsum16(mfL));
        if (lane < G) { s_ml[lane][wave][0] = mfM; s_ml[lane][wave][1] = lw; }
        if (lane < 16) {
#pragma unroll
            for (int jb = 0; jb < 8; ++jb)
#pragma unroll
                for (int r = 0; r < G; ++r) s_o[r][wave][16 * jb + lane] = oacc[jb][r];
        }
    } else {   // one q head (memory-bound shards): the 4 lane-groups merge in registers, 8 wavefront states in LDS
        const float mw = xmax32(xmax16(m[0]));
        const float sc = fast_exp2(m[0] - mw);
        const float lw = xsum32(xsum16(l[0] * sc));
        float ov[8], r0, r1;
#pragma unroll
        for (int e = 0; e < 8; ++e) ov[e] = o[0][e] * sc;
        xsum_rows8(ov, r0, r1);      // (row r of the wavefront ends up with dims d0 + xrow_e(r) and d0 + 4 + xrow_e(r))
        const int e0 = xrow_e(lane >> 4);
        s_o[0][wave][d0 + e0] = r0;
        s_o[0][wave][d0 + 4 + e0] = r1;
        if (lane == 0) { s_ml[0][wave][0] = mw; s_ml[0][wave][1] = lw; }
    }
    // the new token + k/v export: split 0 of the group
    if (j == 0 && gid == 0) {      // (j = 0 is an attention workgroup in every geometry)
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
            for (int e = 0; e < 8; ++e) sn = __builtin_fmaf(MF ? (float)s_qh[hh * HEAD_DIM + d0 + e] : q[hh][e], kf[e], sn);
            sn = sum16(sn);
#pragma unroll
            for (int e = 0; e < 8; ++e) s_o[hh][NST - 1][d0 + e] = vf[e];
            if (l16 == 0) { s_ml[hh][NST - 1][0] = sn; s_ml[hh][NST - 1][1] = 1.f; }
        }
    }
    CF_TRACE(12);
    lds_barrier();
    CF_TRACE(3);

    // ---- X2: G records per workgroup -> the q head's leader --------------------------------------------
    // A record = the unit's softmax state, NORMALISED: 64 granules of fp16 pairs o[2i], o[2i+1] (o / l: a convex combination
    // of V rows, bounded by max |v| whatever the unit's token count), then m and l as fp32 -- FUSED_RECH = 66 granules
    // instead of 130: half the lines every merge sweeps.  A merge weights record s with exp2(m_s - M) l_s.
    constexpr int RH = FUSED_RECH, RM = HEAD_DIM / 2, RL = HEAD_DIM / 2 + 1;
    auto rec_o = [](const unsigned* r, int d) -> float {      // dim d of a record gathered into LDS
        return (float)__builtin_bit_cast(h16x2, r[d >> 1])[d & 1];
    };
    {
        const int nst = j == 0 ? NST : NST - 1;          // the new token belongs to split 0
        if (tid < G * NST) {                             // merge weights of the states (already divided by L); M, L of the record
            const int hh = tid / NST, i = tid - hh * NST;
            float mv[NST];
#pragma unroll
            for (int w = 0; w < NST; ++w) mv[w] = s_ml[hh][w][0];      // all reads in flight together
            float M = NEG_BIG;
#pragma unroll
            for (int w = 0; w < NST; ++w) M = fmaxf(M, w < nst ? mv[w] : NEG_BIG);
            float L = 0.f;
#pragma unroll
            for (int w = 0; w < NST; ++w)
                if (w < nst) L = __builtin_fmaf(fast_exp2(mv[w] - M), s_ml[hh][w][1], L);
            const float rL = L > 0.f ? 1.f / L : 0.f;      // (a unit without tokens: o = 0, l = 0)
            s_w[hh][i] = i < nst ? fast_exp2(mv[i] - M) * rL : 0.f;
            if (i == 0 && att) {
                u64* rec = a.g_rec + (((size_t)g * G + hh) * NSA + ja) * RH;
                granule_store_to(rec + RM, epoch, M, rec_local);
                granule_store_to(rec + RL, epoch, L, rec_local);
            }
        }
        lds_barrier();
        for (int t = tid; t < G * HEAD_DIM; t += 512) {
            const int hh = t >> 7, d = t & 127;
            float val = 0.f;
#pragma unroll
            for (int w = 0; w < NST; ++w)   // (the new-token slot of splits > 0 is uninitialised LDS: 0 x NaN)
                val = __builtin_fmaf(s_w[hh][w], w < nst ? s_o[hh][w][d] : 0.f, val);
            const float next = __shfl_down(val, 1);
            h16x2 pr;
            pr[0] = (h16)val;
            pr[1] = (h16)next;
            if (!(d & 1) && att) granule_store_to(a.g_rec + (((size_t)g * G + hh) * NSA + ja) * RH + (d >> 1), epoch, __builtin_bit_cast(float, pr), rec_local);
        }
    }
    unsigned* s_recu = reinterpret_cast<unsigned*>(s_rec);
    // merge of `n` gathered records at r[0], r[RH], ..: thread `t` < 128 gets dim t of sum_s w_s o_s / L; *Mo, *Lo = the merged
    // state.  (every thread recomputes the n weights: n <= 32 exps beside LDS reads, fixed order)
    auto merge_records = [&](const unsigned* r, auto n_c, int t, float& Mo, float& Lo) -> float {
        constexpr int N = decltype(n_c)::value;
        float M = NEG_BIG;
#pragma unroll
        for (int w = 0; w < N; ++w) M = fmaxf(M, __builtin_bit_cast(float, r[w * RH + RM]));
        float acc = 0.f, L = 0.f;
#pragma unroll
        for (int w = 0; w < N; ++w) {
            const float wt = fast_exp2(__builtin_bit_cast(float, r[w * RH + RM]) - M) * __builtin_bit_cast(float, r[w * RH + RL]);
            acc = __builtin_fmaf(wt, rec_o(r + w * RH, t), acc);
            L += wt;
        }
        Mo = M;
        Lo = L;
        return L > 0.f ? acc / L : 0.f;
    };
    // two fp16 values per granule (phase 3 consumes fp16): thread t < 128 holds dim t
    auto publish_pair = [&](u64* dst, float mine, int t, bool local) {
        const float next = __shfl_down(mine, 1);
        h16x2 pr;
        pr[0] = (h16)mine;
        pr[1] = (h16)next;
        if (!(t & 1)) granule_store_to(dst + (t >> 1), epoch, __builtin_bit_cast(float, pr), local);
    };
    // two merge levels (8 records -> a sub-leader, NS / 8 merged records -> the leader) only with 64 workgroups per head: with
    // the half-size records one leader sweeps the 32 records of a head itself -- one hop less (same-box A/B: config 4 26.17 ->
    // 26.02 us, TP-4 shard 16.6 -> 16.0 us; round 2's full-size records needed the tree)
    constexpr bool TREE = GM::TREE;
    // Few q heads (the small shards): EVERY workgroup gathers the HQ * NS / 8 merged records itself and finishes the softmax
    // merge locally -- 8.5 KB per workgroup while the memory system is idle -- instead of waiting for a leader to merge,
    // publish the attention vector and for X3 to carry it back: one hand-off less on a chain that is all hand-offs.
    constexpr bool LEADERLESS = GM::LEADERLESS;
    if constexpr (TREE) {
        // Level 1: SG = NS / 8 consecutive workgroups of the group form a sub-group (consecutive j share an XCD); its member
        // jj < G merges the sub-group's SG records of q head g*G + jj -- SG / 8 records per wavefront -- and publishes one
        // record of the same format.  Level 2: the 8 merged records of a head go to its leader (j < G), or -- few q heads --
        // to everybody.  One leader sweeping all NS records waited for the slowest of them and then paid the whole sweep on
        // the critical path.
        constexpr int NSG = GM::NSG, SG = GM::SG, R1 = SG / 8;      // (R1 level-1 records per sweeping wavefront)
        constexpr int L2H = (NSG * RH + 15) & ~15;      // merged records of one head: whole 128-B lines (heads of different XCDs never share one)
        const int sg = ja / SG, jj = ja % SG;
        u64* lvl2 = a.g_qkv_io;            // [HQ][L2H] (the [in,out] kernels' split-K area: unused by this layout)
        if (att && jj < G) {
            lds_barrier();   // s_rec reuses s_o: every wavefront is done reading the states
            const bool ok = sweep_granules_raw<(R1 * RH + 63) / 64>(a.g_rec + (((size_t)g * G + jj) * NSA + SG * sg + wave * R1) * RH, R1 * RH, epoch,
                                                                    s_recu + wave * R1 * RH, lane, a.state + 1, 2u);
            if (lane == 0) s_ctl[1 + wave] = ok;
            lds_barrier();
            bool all_ok = true;
            for (int w = 0; w < 8; ++w) all_ok &= s_ctl[1 + w] != 0;
            if (!all_ok) CF_FAIL_RETURN();
            if (tid < HEAD_DIM) {
                float M, L;
                const float val = merge_records(s_recu, FusedArm<SG>{}, tid, M, L);
                u64* dst = lvl2 + ((size_t)g * G + jj) * L2H + sg * RH;
                const bool loc = LEADERLESS ? false : grp_local;
                publish_pair(dst, val, tid, loc);
                if (tid == 0) {
                    granule_store_to(dst + RM, epoch, M, loc);
                    granule_store_to(dst + RL, epoch, L, loc);
                }
            }
        }
        if constexpr (LEADERLESS) {
            constexpr int NREC = HQ * NSG, RPWV = NREC / 8, CNT = RPWV * RH, NL = (CNT + 63) / 64;
            static_assert(NREC % 8 == 0 && NREC * RH * 4 <= GM::REC_BYTES, "merged records of all heads fit the record area");
            lds_barrier();   // s_rec reuses s_o (and a sub-leader's level-1 records): everybody is done with them
            unsigned v[NL];
            bool ok = true;
            for (unsigned spin = 0;; ++spin) {
                bool good = true;
#pragma unroll
                for (int k = 0; k < NL; ++k) {
                    const int i = lane + WAVE * k, rec = wave * RPWV + i / RH, off = i % RH;
                    u64 x = (u64)epoch << 32;
                    if (i < CNT) x = __hip_atomic_load(lvl2 + (size_t)(rec / NSG) * L2H + (rec % NSG) * RH + off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    v[k] = (unsigned)x;
                    good &= (unsigned)(x >> 32) == epoch;
                }
                if (__all(good)) break;
                if (spin > FUSED_SPIN_LIMIT) {
                    if (lane == 0) flag_exchange_error(a.state + 1, 2u);
                    ok = false;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int k = 0; k < NL; ++k) {
                const int i = lane + WAVE * k;
                if (i < CNT) s_recu[wave * RPWV * RH + i] = v[k];
            }
            if (lane == 0) s_ctl[21 + wave] = ok;
            lds_barrier();
            bool all_ok = true;
            for (int w = 0; w < 8; ++w) all_ok &= s_ctl[21 + w] != 0;
            if (!all_ok) CF_FAIL_RETURN();
            CF_TRACE(4);
            for (int t = tid; t < HQ * HEAD_DIM; t += 512) {      // (fp16, as the reference rounds the attention output)
                float M, L;
                reinterpret_cast<h16*>(s_a)[t] = (h16)merge_records(s_recu + (size_t)(t >> 7) * NSG * RH, FusedArm<NSG>{}, t & 127, M, L);
            }
        } else
        if (att && ja < G) {   // leader of q head g*G + ja (it was the sub-leader of sub-group 0 for the same head)
            unsigned* s_rec2 = s_recu + SG * RH;
            constexpr int R2 = NSG <= 8 ? 1 : NSG / 8, W2 = NSG / R2;      // merged records per sweeping wavefront; wavefronts that sweep
            if (wave < W2) {
                const bool ok = sweep_granules_raw<(R2 * RH + 63) / 64>(lvl2 + ((size_t)g * G + ja) * L2H + wave * R2 * RH, R2 * RH, epoch,
                                                                        s_rec2 + wave * R2 * RH, lane, a.state + 1, 2u);
                if (lane == 0) s_ctl[21 + wave] = ok;      // (own slots: a slow wavefront may still be reading level 1's)
            }
            lds_barrier();
            bool all_ok = true;
            for (int w = 0; w < W2; ++w) all_ok &= s_ctl[21 + w] != 0;
            if (!all_ok) CF_FAIL_RETURN();
            if (tid < HEAD_DIM) {
                float M, L;
                const float mine = merge_records(s_rec2, FusedArm<NSG>{}, tid, M, L);
                publish_pair(a.g_attn + ((size_t)g * G + ja) * (HEAD_DIM / 2), mine, tid, false);
            }
        }
    } else
    if (j < G) {   // leader of q head g*G + j: wavefront w gathers NS/8 records, then the softmax merge
        lds_barrier();   // s_rec reuses s_o: every wavefront is done reading the states
        constexpr int CNT = GM::RECW * RH;
        const bool ok = sweep_granules_raw<(CNT + 63) / 64>(a.g_rec + (((size_t)g * G + j) * NS + wave * GM::RECW) * RH,
                                                           CNT, epoch, s_recu + wave * CNT, lane, a.state + 1, 2u);
        if (lane == 0) s_ctl[1 + wave] = ok;
        lds_barrier();
        bool all_ok = true;
        for (int w = 0; w < 8; ++w) all_ok &= s_ctl[1 + w] != 0;
        if (!all_ok) CF_FAIL_RETURN();
        if (tid < HEAD_DIM) {
            float M, L;
            const float mine = merge_records(s_recu, FusedArm<NS>{}, tid, M, L);
            publish_pair(a.g_attn + ((size_t)g * G + j) * (HEAD_DIM / 2), mine, tid, false);
        }
    }

    if constexpr (!LEADERLESS) CF_TRACE(4);
    // ---- X3: every workgroup gathers the full attention output -------------------------------------------
    if constexpr (!LEADERLESS) {
        constexpr int PER = HQ * HEAD_DIM / 16;                     // granules (fp16 pairs) per wavefront: PER / 64 heads
        constexpr int NH = PER >= 64 ? PER / 64 : 1, LAST = PER >= 64 ? 63 : PER - 1;
        // cheap wait (until half of the heads are there), then the checked sweep.  One q head per kv head (the MHA shards): ONE
        // wavefront watches all heads and the others wait at the LDS barrier (8q/8kv 15.37 -> 15.22 us, 16q/16kv 21.38 -> 21.28;
        // with G = 4 the same measured neutral (32q/8kv) or +0.15 us (16q/4kv): there every wavefront watches its own heads)
        if constexpr (G == 1 && CF_X3_ONE_POLLER) {
            if (wave == 0) wait_hint(a.g_attn + HEAD_DIM / 2 - 1, HQ, HEAD_DIM / 2, epoch, lane, HQ / 2);
            lds_barrier();
        } else {
            wait_hint(a.g_attn + wave * PER + LAST, NH, HEAD_DIM / 2, epoch, lane, NH / 2);
        }
        const bool ok = sweep_granules_raw<(PER + 63) / 64>(a.g_attn + wave * PER, PER, epoch, reinterpret_cast<unsigned*>(s_a) + wave * PER, lane,
                                                            a.state + 1, 3u);
        if (lane == 0) s_ctl[9 + wave] = ok;
    }
    lds_barrier();
    if constexpr (!LEADERLESS) {
        bool all_ok = true;
        for (int w = 0; w < 8; ++w) all_ok &= s_ctl[9 + w] != 0;
        if (!all_ok) CF_FAIL_RETURN();
    }

    CF_TRACE(5);
    // ---- phase 3: 16 rows of Wo per workgroup -------------------------------------------------------------
    h16x8 av[JO];
#pragma unroll
    for (int jj = 0; jj < JO; ++jj) av[jj] = *reinterpret_cast<const h16x8*>(reinterpret_cast<const h16*>(s_a) + (jj * WAVE + lane) * 8);
    {
        float res[P3R];
        go.dot_h(av, res);
        if (lane == 63) {
#pragma unroll
            for (int r = 0; r < P3R; ++r) a.out[(8 * ai + wave) * P3R + r] = (h16)res[r];
        }
        if (a.tp_world > 0) tp_publish_wg<P3R / 2>(a, tp_epoch, 4 * P3R * ai, res, reinterpret_cast<unsigned*>(s_qkv), lane, wave);      // (s_qkv: free since phase 2)
    }
    if (a.residual_out && tid < 8 * P3R) {
        const int i = 8 * P3R * ai + tid;
        a.residual_out[i] = (h16)((float)a.na.x[i] + (float)a.na.residual[i]);
    }
    if (b == 0 && tid == 0) {
        a.state[0] = epoch;
        a.state[2] = LONG ? FUSED_ARM_LONG : FUSED_ARM_TWO;      // which arm this call took (cf_workspace_last_arm)
    }
    CF_TRACE(6);
    };   // rest
    if (tps <= (TWO ? 2 : 1) * TILE) rest(FusedArm<0>{});      // the slice fits the tiles requested before X1 (S <= SHORT_TOKENS)
    else rest(FusedArm<1>{});
}

}  // namespace cf
