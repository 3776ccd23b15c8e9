// cf_fused_kernel_q.h -- the persistent [out,in] MHA decode layer for 5 .. 16 sequences in ONE launch, projections on the
// matrix cores (gfx950 v_mfma_f32_16x16x32_f16).
//
// `llama_decoder_layer_batch_decode_sglang` (reference: one launch for every batch size, grid = HEAD_NUM * CLUSTER_SIZE *
// batch_size, llama_kernel_batch_sglang_dispatch.cu:89; its kernel is run once per sequence and re-reads every weight,
// kernel_batch_sglang.cuh:63-64).  Up to 4 rows ride the VALU weight stream of k_fused_decode_mhab; from 5 rows a byte of a
// weight meets 5 .. 16 MACs and the projections belong on the MFMA units.  The five-launch path (cf_batch_kernels.h: norm,
// QKV GEMM, attention, merge, O GEMM) pays two ~5-us latency launches, four launch boundaries and streams its GEMMs at
// 4 TB/s: 66.6 / 86.9 us at 8 / 16 rows of 1024 tokens where the bytes need 41 / 62 us.  Here the whole layer is one
// persistent launch of 256 co-resident workgroups (one per CU, 8 wavefronts):
//   * phase 0 / X0: row r is normalised once, by workgroup 17 r, and handed to everybody as 8 KB of fp16 behind a flag; every
//     workgroup loads the B rows straight into the MFMA B operand of its wavefronts' K-slices (wavefront w owns columns
//     [512 w, 512 w + 512) of every weight row it multiplies: split-K inside the workgroup);
//   * phase 1: workgroup b owns a share of the Wqkv rows (48 on average, up to four 16-row tiles); a wavefront requests ONE row's 1-KB
//     slice per instruction (the GEMV kernels' coalescing: tools/ubench/opl_bw.hip, 4.2 vs 2.8 TB/s for operand-layout
//     requests), turns the 16 rows of a tile into operand layout through a wavefront-private LDS image (k_proj_rows_lds),
//     16 MFMAs per tile; the 8 K-slices meet in LDS in fixed order; q|k|v of every (row, head) leave as tagged granules (X1);
//   * phase 2: the 8 workgroups of a head are dealt to the rows: workgroup (head h, j) serves rows j and j + 8 WHOLE -- a
//     (row, head) is never split over workgroups, so there are no split records, no leader and no X2; K/V tiles of 128
//     tokens stream two deep through registers exactly as in k_fused_decode_mha (the first two are requested before X1
//     resolves), per-row page table (first 2048 entries staged in LDS, the rest through L2), length, RoPE position and
//     cache slot read on the device;
//   * X3: the normalised attention output of (row, head) leaves as 256 bytes of fp16 (write-through stores) behind one flag
//     granule (guide G16 "R1": tagged granules would double the bytes every workgroup gathers -- B x 8 KB is already as much
//     as its weights at 16 rows); wavefront w of every workgroup waits for the flags of heads 4 w .. 4 w + 3 of all rows and
//     loads them straight into the MFMA B operand of its K-slice of phase 3;
//   * phase 3: one 16-row tile of Wo per workgroup (rows [16 b, 16 b + 16)), requested behind X1 and parked in the LDS images
//     through phase 2.
// Scope: hidden 4096, 32 q = 32 kv heads, paged KV, 5 <= B <= 16 (fewer rows: k_fused_decode_mha / _mhab).  Deterministic:
// fixed-order fp32 sums, no atomics on data.
#pragma once
#include "cf_batch_kernels.h"
#include "cf_fused_kernel.h"

#ifndef CF_Q_UP
#define CF_Q_UP 8      // token rows per lane-group of the tile requested before X1 (8: 256 tokens)
#endif

namespace cf {

struct FusedQGeom {
    static constexpr int MAX_ROWS = 16;
    static constexpr int L_IMG = 0;                                   // h16[8][PROJ_LDS_WAVE]   wavefront-private weight images
    static constexpr int L_PART = L_IMG + 8 * PROJ_LDS_WAVE * 2;      // float[8][256]           split-K partial blocks
    static constexpr int L_SS = L_PART + 8 * 256 * 4;                 // float[16][8]            sums of squares per (row, wavefront)
    static constexpr int MAX_IDX = 2048;                              // page-table entries staged per row slot
    static constexpr int L_IDX = L_SS + 16 * 8 * 4;                   // int[2][MAX_IDX]
    static constexpr int L_CS = L_IDX + 2 * MAX_IDX * 4;              // float[2][256]           cos | sin of the two row slots
    static constexpr int L_CTL = L_CS + 2 * 256 * 4;                  // int[64]
    static constexpr int L_END = L_CTL + 256;
    // phase 2's scratch lives in the split-K block area (idle between the projections; the images hold phase 3's Wo tile by then)
    static constexpr int L_QKV = L_PART;                              // float[2][384]
    static constexpr int L_O = L_QKV + 2 * 384 * 4;                   // float[9][128]
    static constexpr int L_ML = L_O + 9 * 128 * 4;                    // float[9][2] (+pad)
    static constexpr int L_P2_END = L_ML + 80;
    static constexpr int LDS_BYTES = L_END;
    static_assert(L_P2_END <= L_SS, "phase-2 scratch stays inside the split-K block area");
    static_assert(LDS_BYTES <= 160 * 1024 && LDS_BYTES > 80 * 1024, "one workgroup per CU");
};

__global__ __launch_bounds__(FUSED_THREADS, 2) void k_fused_decode_mhaq(FusedArgs a, int batch) {
    using GM = FusedQGeom;
    constexpr int HID = 4096;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    h16* s_img = reinterpret_cast<h16*>(smem + GM::L_IMG) + wave * PROJ_LDS_WAVE;
    float* s_part = reinterpret_cast<float*>(smem + GM::L_PART);              // [8][256]
    float* s_ss = reinterpret_cast<float*>(smem + GM::L_SS);                  // [16][8]
    int* s_idx = reinterpret_cast<int*>(smem + GM::L_IDX);                    // [2][MAX_IDX]
    float* s_cs = reinterpret_cast<float*>(smem + GM::L_CS);                  // [2][256]
    int* s_ctl = reinterpret_cast<int*>(smem + GM::L_CTL);
    float* s_qkv = reinterpret_cast<float*>(smem + GM::L_QKV);                // [2][384]
    float(*s_o)[HEAD_DIM] = reinterpret_cast<float(*)[HEAD_DIM]>(smem + GM::L_O);
    float(*s_ml)[2] = reinterpret_cast<float(*)[2]>(smem + GM::L_ML);

    const int r16 = lane & 15, kq = lane >> 4;            // MFMA operand coordinates: row / batch column, k-group
    const int l16 = r16, gid = wave * 4 + kq, d0 = l16 * 8;   // phase 2: 16 lanes x 8 dims per token row, 32 token rows per step
    const int b = blockIdx.x;
    const int h = (b & 7) * 4 + (b >> 6);                 // the 8 workgroups of a head share b % 8 (one XCD: speed only)
    const int j = (b >> 3) & 7;
    CF_TRACE(0);

    const unsigned epoch = scalar_load(a.state) + 1u;
    // the two row slots of this workgroup in phase 2: rows j and j + 8
    const bool live0 = j < batch, live1 = j + 8 < batch;
    int S0 = 0, S1 = 0, ent00 = 0, ent01 = 0;
    if (live0) {
        ent00 = scalar_load(a.indptr + j);
        S0 = a.seq_lens ? scalar_load(a.seq_lens + j) : scalar_load(a.indptr + j + 1) - 1 - ent00;
    }
    if (live1) {
        ent01 = scalar_load(a.indptr + j + 8);
        S1 = a.seq_lens ? scalar_load(a.seq_lens + j + 8) : scalar_load(a.indptr + j + 9) - 1 - ent01;
    }
    const int64_t roff0 = (a.positions && live0) ? scalar_load(a.positions + j) * a.rope_stride : 0;
    const int64_t roff1 = (a.positions && live1) ? scalar_load(a.positions + j + 8) * a.rope_stride : 0;
    const h16* kc = a.kptrs ? reinterpret_cast<const h16*>(scalar_load(a.kptrs + a.layer_id)) : a.k_cache;
    const h16* vc = a.vptrs ? reinterpret_cast<const h16*>(scalar_load(a.vptrs + a.layer_id)) : a.v_cache;
    const int ps = a.page_shift, pmask = (1 << ps) - 1;

    // ---- weight tiles: 16 rows of this workgroup's 48 Wqkv rows, one 1-KB row slice per instruction (lane l: 16 bytes at
    //      column 512 w + 8 l) ---------------------------------------------------------------------------------------------
    const int kw = wave * 512;
    h16x8 wa[16], wb[16];
    auto load_w = [&](h16x8 (&t)[16], const h16* W, int row0) {
        const h16* p = W + (size_t)row0 * HID + kw + lane * 8;
#pragma unroll
        for (int i = 0; i < 16; ++i) t[i] = ld_stream(p + (size_t)i * HID);
    };
    // Phase-1 shares: workgroup b owns the Wqkv rows [p1_start[b], p1_start[b + 1]) -- any workgroup can produce any row (the
    // X1 consumers find q|k|v by granule address), so the split is a free load-balancing knob filled in by the host: the
    // workgroups whose phase 2 streams slower (heads h = 1 mod 4, odd XCDs: DESIGN 3.1) get fewer rows.  Up to four 16-row
    // tiles; rows come through a buffer resource, a row beyond the share gets an offset beyond the buffer: the instruction
    // still issues (one code path, exact wait counts) but touches no memory and returns zeros.
    const int r_lo = a.p1_start[b], r_hi = a.p1_start[b + 1];
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16*>(a.Wqkv), 0, 3 * HID * HID * 2, 0x00020000);
    auto load_p1 = [&](h16x8 (&t)[16], int tile) {
        const int row0 = r_lo + 16 * tile;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int voff = row0 + i < r_hi ? (row0 + i) * (HID * 2) + (kw + lane * 8) * 2 : 0x40000000;
            t[i] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, voff, 0, 2 /* nt */));
        }
    };

    // ---- second-level loads of the two row slots (page-table slices, new-token slots, RoPE rows): registers now, LDS below ------
    const bool nlive = r16 < batch;
    int idx_reg[2][GM::MAX_IDX / FUSED_THREADS], slot_reg = 0;
    float cs_reg = 0.f;
    {
#pragma unroll
        for (int rs2 = 0; rs2 < 2; ++rs2) {
            const bool lv = rs2 ? live1 : live0;
            const int Sr = rs2 ? S1 : S0, e0 = rs2 ? ent01 : ent00;
            const int n_idx = lv ? (Sr >> ps) + 1 : 0;                 // entries of the row incl. the new token's page
#pragma unroll
            for (int c = 0; c < GM::MAX_IDX / FUSED_THREADS; ++c) {
                const int i = c * FUSED_THREADS + tid;
                idx_reg[rs2][c] = i < n_idx ? a.indices[e0 + i] : 0;
            }
        }
        // lanes 0 / 1 of wavefront 0: the page of the new token of slot 0 / 1
        if (tid < 2) {
            const bool lv = tid ? live1 : live0;
            const int Sr = tid ? S1 : S0, e0 = tid ? ent01 : ent00;
            slot_reg = lv ? a.indices[e0 + (Sr >> ps)] : 0;
        }
        // RoPE rows: threads 0..255 slot 0, 256..511 slot 1; [0,128) cos, [128,256) sin (NEOX reads 64 of each)
        const int t = tid & 255;
        const int n_ang = a.rope_style == 0 ? HEAD_DIM / 2 : HEAD_DIM;
        const int64_t ro = tid < 256 ? roff0 : roff1;
        if (t < n_ang) cs_reg = a.cos[ro + t];
        else if (t >= 128 && t < 128 + n_ang) cs_reg = a.sin[ro + t - 128];
    }
    auto stage_second_level = [&]() {
#pragma unroll
        for (int rs2 = 0; rs2 < 2; ++rs2)
#pragma unroll
            for (int c = 0; c < GM::MAX_IDX / FUSED_THREADS; ++c) s_idx[rs2 * GM::MAX_IDX + c * FUSED_THREADS + tid] = idx_reg[rs2][c];
        s_cs[tid] = cs_reg;
        if (tid < 2) s_ctl[20 + tid] = slot_reg;
    };

    // ---- phase 0 / X0: the fused add + RMSNorm of row r is computed ONCE, by workgroup 17 r (the producers sit on different
    //      XCDs), rounded once to fp16 (kernel.cuh:133-138) and handed to everybody as 8 KB of write-through stores behind one
    //      flag (guide G16 "R1"); the producer also writes the row of residual_out.  Every workgroup normalising all B rows
    //      itself (the first version) pulled B x 16 KB of x and residual through its request pipe -- 128 KB at 8 rows, a quarter
    //      of its weight stream -- where the normalised rows are B x 8 KB; the hop hides behind the first two weight tiles, which
    //      every consumer requests before it waits.  (A producer requests its tiles after it has published: a drained queue is
    //      what orders the flag behind the payload.  The host gives the producers a smaller phase-1 share.) ------------------------
    const __amdgpu_buffer_rsrc_t xn_rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(a.g_qkv_io), 0, batch * HID * 2, 0x00020000);
    u64* xn_flags = a.g_qkv_io + (size_t)batch * (HID * 2 / 8);
    const int prow = b / 17;
    const bool producer = b == 17 * prow && prow < batch;      // (workgroup-uniform)
    if (producer) {
        const size_t xo = (size_t)prow * HID + tid * 8;
        const h16x8 xv = ld_h8(a.na.x + xo), rv = ld_h8((a.na.residual ? a.na.residual : a.na.x) + xo), wv = ld_h8(a.na.rms_w + tid * 8);
        const float rs = a.na.residual ? 1.f : 0.f;
        float hx[8], ss = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            hx[e] = __builtin_fmaf(rs, (float)rv[e], (float)xv[e]);
            ss = __builtin_fmaf(hx[e], hx[e], ss);
        }
        ss = sum64_lane63(ss);
        if (lane == 63) s_ss[wave] = ss;
        stage_second_level();
        lds_barrier();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) tot += s_ss[w];
        const float rcp = __builtin_amdgcn_rsqf(tot / (float)HID + a.na.eps);
        h16x8 xo16, ho16;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            xo16[e] = (h16)(hx[e] * rcp * (float)wv[e]);
            ho16[e] = (h16)hx[e];
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, xo16), xn_rsrc, (prow * HID + tid * 8) * 2, 0, 16 /* sc1 */);
        // residual_out may alias residual: this workgroup is the only reader of the row, and it has read it
        if (a.residual_out) st_h8(a.residual_out + xo, ho16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wavefront drains (inline asm: the compiler cannot drop it)
        lds_barrier();
        if (tid == 0) granule_store(xn_flags + prow, epoch, 0.f);
        load_p1(wa, 0);
    } else {
        load_p1(wa, 0);
        stage_second_level();
        lds_barrier();
    }
    // the B operand of this wavefront's K-slice: bx[s] = xn[row r16][512 w + 32 s + 8 kq .. + 8); rows >= batch are zero.
    // (One weight tile is in flight while the flags are awaited -- loads return in issue order, so the poll comes back behind it,
    //  ~5 us into the kernel, when the rows are long published; the second tile is requested behind the operand loads.)
    h16x8 bx[16];
    {
        bool ok = false;
        for (unsigned spin = 0; spin < FUSED_SPIN_LIMIT; ++spin) {
            u64 x = (u64)epoch << 32;
            if (lane < batch) x = __hip_atomic_load(xn_flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all((unsigned)(x >> 32) == epoch)) { ok = true; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (!ok && lane == 0) flag_exchange_error(a.state + 1, 6u);
        const int off = ((nlive ? r16 : 0) * HID + kw + kq * 8) * 2;
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xn_rsrc, off + 64 * s2, 0, 16 /* sc1 */);
            bx[s2] = nlive ? __builtin_bit_cast(h16x8, v) : h16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
        if (lane == 0) s_ctl[40 + wave] = ok;
    }
    load_p1(wb, 1);      // (requesting it ahead of the poll instead measured the same: 53.3 us at 8 rows either way)

    CF_TRACE(14);   // operand ready

    // ---- phase 1: three 16-row tiles; tile -> image -> 16 MFMAs -> the 8 K-slices meet in LDS -> granules of (row, head) ----
    auto to_image = [&](const h16x8 (&t)[16]) {
#pragma unroll
        for (int i = 0; i < 16; ++i) *reinterpret_cast<h16x8*>(s_img + i * PROJ_LDS_ROW + lane * 8) = t[i];
    };
    auto mfma_tile = [&](const h16x8 (&bop)[16]) -> f32x4_t {
        f32x4_t d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const h16x8 av = *reinterpret_cast<const h16x8*>(s_img + r16 * PROJ_LDS_ROW + 32 * s + 8 * kq);
            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bop[s], d, 0, 0, 0);
        }
        return d;       // lane l: weight rows m = 4 (l / 16) + i, batch column n = l % 16
    };
    auto publish_qkv = [&](f32x4_t d, int tile) {
        *reinterpret_cast<f32x4_t*>(&s_part[wave * 256 + lane * 4]) = d;
        lds_only_barrier();
        if (tid < 256) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += s_part[w * 256 + tid];      // fixed order
            const int l = tid >> 2, i = tid & 3, n = l & 15, m = 4 * (l >> 4) + i;
            const int row = r_lo + 16 * tile + m;                        // Wqkv row: q of all heads | k | v
            if (n < batch && row < r_hi)
                granule_store(a.g_qkv + ((size_t)n * FUSED_HEADS + ((row & 4095) >> 7)) * 384 + (row >> 12) * 128 + (row & 127), epoch, v);
        }
        lds_only_barrier();
    };
    // K/V tiles of phase 2 (requested below, before q exists): `q` = tile number inside row slot `rs`
    const size_t kvstride = (size_t)FUSED_HEADS * HEAD_DIM;
    const h16* kbase = kc + h * HEAD_DIM + d0;
    const h16* vbase = vc + h * HEAD_DIM + d0;
    constexpr int UT = 4, TILE = FUSED_GROUPS * UT;      // 128 tokens per tile of the loop, two tiles in flight
    constexpr int UP = CF_Q_UP, PRE = FUSED_GROUPS * UP;       // row slot 0: a 256-token tile + the first loop tile are requested before X1
    auto load_tile = [&](auto& t, int rs2, int tbase) {
        constexpr int UT = sizeof(t.k) / sizeof(h16x8), TILE = FUSED_GROUPS * UT;
        const int Sr = rs2 ? S1 : S0, e0 = rs2 ? ent01 : ent00;
        size_t rows[UT];
        if (((tbase + TILE - 1) >> ps) < GM::MAX_IDX) {          // (workgroup-uniform) pages of this tile are staged in LDS
#pragma unroll
            for (int u = 0; u < UT; ++u) {
                int tk = tbase + u * FUSED_GROUPS + gid;
                tk = tk < Sr ? tk : Sr - 1;
                tk = tk > 0 ? tk : 0;
                rows[u] = ((size_t)s_idx[rs2 * GM::MAX_IDX + (tk >> ps)] << ps) + (size_t)(tk & pmask);
            }
        } else {                                                 // a longer row: page numbers through L2
#pragma unroll
            for (int u = 0; u < UT; ++u) {
                int tk = tbase + u * FUSED_GROUPS + gid;
                tk = tk < Sr ? tk : Sr - 1;
                tk = tk > 0 ? tk : 0;
                rows[u] = ((size_t)a.indices[e0 + (tk >> ps)] << ps) + (size_t)(tk & pmask);
            }
        }
#pragma unroll
        for (int u = 0; u < UT; ++u) {
            t.k[u] = ld_stream(kbase + rows[u] * kvstride);
            t.v[u] = ld_stream(vbase + rows[u] * kvstride);
        }
    };
    KvTile32<UP> pa;          // row slot 0, tokens [0, 256): in flight across X1 together with ta (192 KB per CU keep the stream busy)
    KvTile32<UT> ta, tb;
    {
        to_image(wa);
        load_p1(wa, 2);
        const f32x4_t d0v = mfma_tile(bx);
        publish_qkv(d0v, 0);
        to_image(wb);
        load_p1(wb, 3);
        const f32x4_t d1v = mfma_tile(bx);
        publish_qkv(d1v, 1);
        to_image(wa);
        load_tile(pa, 0, 0);      // (unconditional: a workgroup without a row reads slot 0 -- a branch here would join two
                                  //  different queue depths and make the next image wait for these tiles)
        const f32x4_t d2v = mfma_tile(bx);
        publish_qkv(d2v, 2);
        to_image(wb);
        load_tile(ta, 0, PRE);
        const f32x4_t d3v = mfma_tile(bx);
        publish_qkv(d3v, 3);       // (ends with a barrier: the split-K blocks are read -- phase 2's scratch reuses the area)
    }
    {   // (X0 gave up somewhere: the publishes above ended with barriers, the flags are visible)
        bool all_ok = true;
        for (int w = 0; w < 8; ++w) all_ok &= s_ctl[40 + w] != 0;
        if (!all_ok) CF_FAIL_RETURN();
    }
    CF_TRACE(1);   // phase 1 done

    // ---- phase 3's weights: this workgroup's 16 rows of Wo are requested right behind X1 and parked in the wavefronts' LDS images
    //      (idle during phase 2) as soon as they arrive: 64 registers for a microsecond instead of across the tile loops, nothing in
    //      flight when the last row is published behind a drained queue (payload, s_waitcnt vmcnt(0), flag), phase 3 never waits ----
    h16x8 go[16];

    // ---- phase 2: row slots 0 and 1, each a whole (row, head) ------------------------------------------------------------------
    const float qscale = 1.44269504088896340736f * 0.08838834764831845f;
    auto rope_lds = [&](const float* src, const float* cs, float (&dst)[8]) {
        if (a.rope_style == 0) {
            const float sgn = d0 < 64 ? -1.f : 1.f;
            const int a0 = d0 & 63, p0 = (d0 + 64) & 127;
#pragma unroll
            for (int e = 0; e < 8; ++e) dst[e] = src[d0 + e] * cs[a0 + e] + sgn * (src[p0 + e] * cs[128 + a0 + e]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float c = cs[d0 + e], sn = cs[128 + d0 + e];
                dst[e] = (e & 1) ? src[d0 + e] * c + src[d0 + (e ^ 1)] * sn : src[d0 + e] * c - src[d0 + (e ^ 1)] * sn;
            }
        }
    };
    // X3 area: [rows][4096] fp16 payload, then [rows][32 heads] flag granules
    const __amdgpu_buffer_rsrc_t x3_rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(a.g_attn), 0, batch * HID * 2, 0x00020000);
    u64* x3_flags = a.g_attn + (size_t)batch * (HID * 2 / 8);
    bool failed = false;
    // RS = row slot; LAST = the workgroup's last row (straight copies per combination: a run-time branch around the requests
    // below would join different queue depths and make the last tile's arithmetic wait for the requests behind it)
    auto row_slot = [&](auto rs_c, auto last_c) {
        constexpr int RS = decltype(rs_c)::value;
        constexpr bool last_row = decltype(last_c)::value != 0;
        const int row = j + 8 * RS;
        const int Sr = RS ? S1 : S0;
        // ---- X1: q|k|v of (row, head) -----------------------------------------------------------------------------------
        if (wave == 0) {
            const bool ok = sweep_granules<6>(a.g_qkv + ((size_t)row * FUSED_HEADS + h) * 384, 384, epoch, s_qkv + RS * 384, lane, a.state + 1, 1u);
            if (lane == 0) s_ctl[RS] = ok;
        }
        lds_barrier();
        if (!s_ctl[RS]) { failed = true; return; }
        if constexpr (RS == 0) CF_TRACE(2);   // X1 resolved
        h16x8 qh;      // (the fp32 q is recomputed for the new token at the end: 8 registers less across the tile loop)
        {
            float q[8];
            rope_lds(s_qkv + RS * 384, s_cs + RS * 256, q);
#pragma unroll
            for (int e = 0; e < 8; ++e) qh[e] = (h16)(q[e] * qscale);
        }
        float m = NEG_BIG, l = 0.f, o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        auto compute_tile = [&](const auto& t, int tbase) {
            constexpr int UT = sizeof(t.k) / sizeof(h16x8);
            float s[UT];
            bool valid[UT];
            float mx = NEG_BIG;
#pragma unroll
            for (int u = 0; u < UT; ++u) {
                valid[u] = (tbase + u * FUSED_GROUPS + gid) < Sr;
                s[u] = sum16(dot8h(t.k[u], qh, 0.f));
                s[u] = valid[u] ? s[u] : NEG_BIG;
                mx = fmaxf(mx, s[u]);
            }
            const float mnew = fmaxf(m, mx);
            const float alpha = fast_exp2(m - mnew);
            float psum = 0.f;
#pragma unroll
            for (int u = 0; u < UT; ++u) {
                s[u] = valid[u] ? fast_exp2(s[u] - mnew) : 0.f;
                psum += s[u];
            }
            l = l * alpha + psum;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float acc = o[e] * alpha;
#pragma unroll
                for (int u = 0; u < UT; ++u) acc = __builtin_fmaf((float)t.v[u][e], s[u], acc);
                o[e] = acc;
            }
            m = mnew;
        };
        // loop tiles: pairs of 128-token tiles from token t_first; tiles behind the row are all-masked (clamped duplicate rows)
        constexpr int t_first = RS == 0 ? PRE : 0;
        int t_end = t_first + ((Sr > t_first ? Sr - t_first : 0) + 2 * TILE - 1) / (2 * TILE) * (2 * TILE);
        t_end = t_end < t_first + 2 * TILE ? t_first + 2 * TILE : t_end;
        if constexpr (RS == 0) {      // the wide tile requested before X1; the Wo tile and the second loop tile go out behind it
            load_w(go, a.Wo, 16 * b);
            load_tile(tb, 0, t_first + TILE);
            compute_tile(pa, 0);
            __builtin_amdgcn_sched_barrier(0);
            to_image(go);
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int tt = t_first; tt + 2 * TILE < t_end; tt += 2 * TILE) {
            compute_tile(ta, tt);
            load_tile(ta, RS, tt + 2 * TILE);
            compute_tile(tb, tt + TILE);
            load_tile(tb, RS, tt + 3 * TILE);
        }
        // last pair of this row: the next requests are the other row slot's first tiles (straight copies per combination)
        compute_tile(ta, t_end - 2 * TILE);
        if constexpr (RS == 0) CF_TRACE(8);   // (first row: all but the last tile consumed)
        if constexpr (!last_row) load_tile(ta, 1, 0);
        compute_tile(tb, t_end - TILE);
        if constexpr (!last_row) load_tile(tb, 1, TILE);
        // merge the 4 lane-groups of this wavefront in registers, then 8 wavefront states (+ the new token) meet in LDS
        {
            const float mw = xmax32(xmax16(m));
            const float sc = fast_exp2(m - mw);
            l = xsum32(xsum16(l * sc));
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] *= sc;
            float r0, r1;
            xsum_rows8(o, r0, r1);      // (row r of the wavefront ends up with dims d0 + xrow_e(r) and d0 + 4 + xrow_e(r))
            const int e0 = xrow_e(lane >> 4);
            s_o[wave][d0 + e0] = r0;
            s_o[wave][d0 + 4 + e0] = r1;
            if (lane == 0) { s_ml[wave][0] = mw; s_ml[wave][1] = l; }
        }
        // the new token of this row (attended from registers, kernel.cuh:444-477) + k/v export + cache write
        if (gid == 0) {
            float kf[8], vf[8];
            rope_lds(s_qkv + RS * 384 + HEAD_DIM, s_cs + RS * 256, kf);
#pragma unroll
            for (int e = 0; e < 8; ++e) vf[e] = s_qkv[RS * 384 + 2 * HEAD_DIM + d0 + e];
            h16x8 k16, v16;
#pragma unroll
            for (int e = 0; e < 8; ++e) { k16[e] = (h16)kf[e]; v16[e] = (h16)vf[e]; }
            const size_t ooff = (size_t)h * HEAD_DIM + d0;
            if (a.k_new) st_h8(a.k_new + (size_t)row * kvstride + ooff, k16);
            if (a.v_new) st_h8(a.v_new + (size_t)row * kvstride + ooff, v16);
            if (a.write_cache) {
                const size_t slot = ((size_t)s_ctl[20 + RS] << ps) + (size_t)(Sr & pmask);
                st_h8(const_cast<h16*>(kc) + slot * kvstride + ooff, k16);
                st_h8(const_cast<h16*>(vc) + slot * kvstride + ooff, v16);
            }
            float q[8], sn = 0.f;
            rope_lds(s_qkv + RS * 384, s_cs + RS * 256, q);
#pragma unroll
            for (int e = 0; e < 8; ++e) sn = __builtin_fmaf(q[e] * qscale, kf[e], sn);
            sn = sum16(sn);
#pragma unroll
            for (int e = 0; e < 8; ++e) s_o[8][d0 + e] = vf[e];
            if (l16 == 0) { s_ml[8][0] = sn; s_ml[8][1] = 1.f; }
        }
        lds_barrier();
        // ---- the row's attention output, normalised, as fp16 pairs (phase 3 consumes fp16: kernel.cuh:553-559) -> X3 ------
        if (tid < HEAD_DIM) {
            float M = NEG_BIG;
#pragma unroll
            for (int i = 0; i < 9; ++i) M = fmaxf(M, s_ml[i][0]);
            float acc = 0.f, L = 0.f;
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const float wt = fast_exp2(s_ml[i][0] - M);
                acc = __builtin_fmaf(wt, s_o[i][tid], acc);
                L = __builtin_fmaf(wt, s_ml[i][1], L);
            }
            // payload: 256 bytes of fp16 per (row, head), sixteen 16-byte WRITE-THROUGH (sc1) stores (guide G16 "R1")
            const float mine = acc / L;
            h16x8 pk;
            pk[0] = (h16)mine;
#pragma unroll
            for (int e = 1; e < 8; ++e) pk[e] = (h16)__shfl_down(mine, e);
            if (!(tid & 7))
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pk), x3_rsrc, (row * HID + h * HEAD_DIM + tid) * 2, 0, 16 /* sc1 */);
            // ... and the flag behind the drained stores.  A row that is not the workgroup's last one keeps its flag back (nobody
            // can use it before every row is out, and tiles of the next row are in flight in this queue): it goes out with the last.
            if constexpr (last_row) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wavefront (inline asm: the compiler cannot drop it)
        }
        lds_barrier();      // the states are read: the next row slot (or phase 3's images) may overwrite them; both storing wavefronts drained
        if constexpr (last_row) {
            if (tid < 2 && (tid == 0 ? true : RS == 1)) {      // lane 0: this row; lane 1: the row of slot 0 that was kept back
                const int frow = tid == 0 ? row : j;
                granule_store(x3_flags + (size_t)frow * FUSED_HEADS + h, epoch, 0.f);
            }
        }
        if constexpr (RS == 0) CF_TRACE(4);   // first row's attention output published
    };
    if (live1) {
        row_slot(FusedArm<0>{}, FusedArm<0>{});
        if (failed) CF_FAIL_RETURN();
        row_slot(FusedArm<1>{}, FusedArm<1>{});
        if (failed) CF_FAIL_RETURN();
    } else if (live0) {
        row_slot(FusedArm<0>{}, FusedArm<1>{});
        if (failed) CF_FAIL_RETURN();
    } else {
        load_w(go, a.Wo, 16 * b);       // (a workgroup without a row: only the two projections)
        to_image(go);
    }
    CF_TRACE(3);   // phase 2 done, attention outputs published

    // ---- X3: heads 4 w .. 4 w + 3 of every row, straight into the B operand of this wavefront's K-slice: lane (n = r16, kq)
    //      watches the flag of (row n, head 4 w + kq); then sixteen 16-byte sc1 loads per lane, no tags to check, one round -----------
    h16x8 ax[16];
    {
        bool ok = false;
        for (unsigned spin = 0; spin < FUSED_SPIN_LIMIT; ++spin) {
            u64 x = (u64)epoch << 32;
            if (nlive) x = __hip_atomic_load(x3_flags + (size_t)r16 * FUSED_HEADS + 4 * wave + kq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all((unsigned)(x >> 32) == epoch)) { ok = true; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (!ok && lane == 0) flag_exchange_error(a.state + 1, 3u);
        const int off = ((nlive ? r16 : 0) * HID + kw + kq * 8) * 2;
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(x3_rsrc, off + 64 * s2, 0, 16 /* sc1: producer wrote through, L1 bypassed */);
            ax[s2] = nlive ? __builtin_bit_cast(h16x8, v) : h16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
        if (lane == 0) s_ctl[9 + wave] = ok;
    }
    lds_barrier();
    {
        bool all_ok = true;
        for (int w = 0; w < 8; ++w) all_ok &= s_ctl[9 + w] != 0;
        if (!all_ok) CF_FAIL_RETURN();
    }
    CF_TRACE(5);   // X3 resolved

    // ---- phase 3: out[n][16 b + m] = sum_k attn[n][k] Wo[16 b + m][k]; the Wo tile waits in the images since phase 2 began -----------
    {
        const f32x4_t d = mfma_tile(ax);
        *reinterpret_cast<f32x4_t*>(&s_part[wave * 256 + lane * 4]) = d;
        lds_only_barrier();
        if (tid < 256) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += s_part[w * 256 + tid];      // fixed order
            const int l = tid >> 2, i = tid & 3, n = l & 15, m = 4 * (l >> 4) + i;
            if (n < batch) a.out[(size_t)n * HID + 16 * b + m] = (h16)v;
        }
    }
    if (b == 0 && tid == 0) a.state[0] = epoch;
    CF_TRACE(6);
}

}  // namespace cf
