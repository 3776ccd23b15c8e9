// cf_fused_kernel_q.h -- the persistent [out,in] MHA decode layer for 5 .. 32 sequences in ONE launch, projections on the
// matrix cores (gfx950 v_mfma_f32_16x16x32_f16).
//
// `llama_decoder_layer_batch_decode_sglang` (reference: one launch for every batch size, grid = HEAD_NUM * CLUSTER_SIZE *
// batch_size, llama_kernel_batch_sglang_dispatch.cu:89; its kernel is run once per sequence and re-reads every weight,
// kernel_batch_sglang.cuh:63-64).  Up to 4 rows ride the VALU weight stream of k_fused_decode_mhab; from 5 rows a byte of a
// weight meets 5 .. 32 MACs and the projections belong on the MFMA units.  The five-launch path (cf_batch_kernels.h: norm,
// QKV GEMM, attention, merge, O GEMM) pays two ~5-us latency launches, four launch boundaries and streams its GEMMs at
// 4 TB/s: 66.6 / 86.9 us at 8 / 16 rows of 1024 tokens where the bytes need 41 / 62 us.  Here the whole layer is one
// persistent launch of 256 co-resident workgroups (one per CU, 8 wavefronts):
//   * phase 0 / X0: row r is normalised once, by workgroup 17 r, and handed to everybody as 8 KB of fp16 behind a flag; every
//     workgroup loads the B rows straight into the MFMA B operand of its wavefronts' K-slices (wavefront w owns columns
//     [512 w, 512 w + 512) of every weight row it multiplies: split-K inside the workgroup);
//   * phase 1: workgroup b owns a share of the Wqkv rows (48 on average, up to four 16-row tiles); a wavefront requests ONE
//     row's 1-KB slice per instruction (the GEMV kernels' coalescing: tools/ubench/opl_bw.hip, 4.2 vs 2.8 TB/s for
//     operand-layout requests), turns the 16 rows of a tile into operand layout through a wavefront-private LDS image
//     (k_proj_rows_lds), 16 MFMAs per tile; the 8 K-slices meet in LDS in fixed order; q|k|v of every (row, head) leave as
//     tagged granules (X1);
//   * phase 2: the cached tokens of ALL rows form one sequence per head (row after row: T tokens); the 8 workgroups of a head
//     take equal ranges of it, whatever the rows' lengths (decode batches are ragged: with one workgroup per (row, head) a
//     batch {4000, 300, 1200, 50, 2500, 800, 100, 3000} took 94 us where its bytes need 60).  A range covers whole rows --
//     their attention output is final in this workgroup, no records, no leader -- plus at most a leading part of a row that
//     began in an earlier range (its softmax state leaves as ONE half-size record) and a trailing row that continues in
//     later ranges (the workgroup where a row begins owns it: it gathers the later parts' records when its own part is done).
//     With equal lengths that divide evenly (8 or 16 rows) no row is split at all.  K/V tiles of 128 tokens stream two deep
//     through registers as in k_fused_decode_mha (the first two are requested before X1 resolves); page numbers come from
//     the page table through L2, requested one tile ahead of the tile they address (the first two tiles' from LDS);
//   * X3: the normalised attention output of (row, head) leaves as 256 bytes of fp16 (write-through stores) behind one flag
//     granule (guide G16 "R1": tagged granules would double the bytes every workgroup gathers -- B x 8 KB is already as much
//     as its weights at 16 rows); wavefront w of every workgroup waits for the flags of heads 4 w .. 4 w + 3 of all rows and
//     loads them straight into the MFMA B operand of its K-slice of phase 3;
//   * phase 3: one 16-row tile of Wo per workgroup (rows [16 b, 16 b + 16)), requested when the range is streamed.
// Scope: hidden 4096, 32 q = 32 kv heads, paged KV, 5 <= B <= 32 (fewer rows: k_fused_decode_mha / _mhab; BT below).  Deterministic:
// fixed-order fp32 sums, no atomics on data.
#pragma once
#include "cf_batch_kernels.h"
#include "cf_fused_kernel.h"

#ifndef CF_Q_WO_BEHIND_FLAGS
#define CF_Q_WO_BEHIND_FLAGS 1      // every wavefront requests its Wo rows behind the workgroup's X3 flags (0: the wavefronts 2 .. 7 behind the range)
#endif
#ifndef CF_Q_UT
#define CF_Q_UT 4      // token rows per lane-group of a K/V tile (4: 128 tokens, two tiles = 16 KB per wavefront in flight)
#endif

namespace cf {

// BT = batch tiles of 16 rows in the MFMA B operand: 1 serves 5 .. 16 sequences, 2 serves 17 .. 32 (round 4: the reference's
// batched entry is one launch for every batch size, llama_kernel_batch_sglang_dispatch.cu:89).  With two tiles the activation
// operand of a wavefront's K-slice is 128 registers, so only ONE weight tile is in flight beside it (the next one is
// requested as soon as the current one sits in the LDS image), where the one-tile kernel keeps two.
template <int BT>
struct FusedQGeomT {
    static constexpr int MAX_ROWS = 16 * BT, R = MAX_ROWS;
    static constexpr int L_IMG = 0;                                   // h16[8][PROJ_LDS_WAVE]   wavefront-private weight images
    static constexpr int L_PART = L_IMG + 8 * PROJ_LDS_WAVE * 2;      // float[8][BT][256]       split-K partial blocks
    static constexpr int L_SS = L_PART + 8 * BT * 256 * 4;            // float[16]               sums of squares (producer)
    static constexpr int L_TAB = L_SS + 64;                           // int S[R] | (32 spare) | ent[R] | int64 roff[R]
    static constexpr int L_PRE = L_TAB + R * 4 + 32 * 4 + R * 4 + R * 8;      // int[512]  cache rows of the range's first 512 tokens
    static constexpr int L_CTL = L_PRE + 512 * 4;                     // int[64]
    static constexpr int L_END = L_CTL + 256;
    // phase 2 lives in the image area (idle between the projections)
    static constexpr int L_QKV = 0;                                   // float[R][384]   q|k|v of the rows this workgroup touches
    static constexpr int L_CS = L_QKV + R * 384 * 4;                  // float[R][256]   their RoPE rows (cos | sin)
    static constexpr int L_O = L_CS + R * 256 * 4;                    // float[9][128]
    static constexpr int L_ML = L_O + 9 * 128 * 4;                    // float[9][2] (+pad)
    static constexpr int L_ST = L_ML + 128;                           // float[132]       own state of the row that continues in later ranges
    static constexpr int L_REC = L_ST + 132 * 4;                      // unsigned[7][FUSED_RECH]  the later parts' records
    static constexpr int L_P2_END = L_REC + 7 * FUSED_RECH * 4;
    static constexpr int LDS_BYTES = L_END > 84 * 1024 ? L_END : 84 * 1024;
    static_assert(L_P2_END <= L_PART, "phase-2 scratch stays inside the image area");
    static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
};
typedef FusedQGeomT<1> FusedQGeom;

// which workgroup normalises row r (X0): one tile 17 r (r < 16), two tiles 8 r + r % 8 (r < 32) -- consecutive rows on different XCDs
__host__ __device__ constexpr int fused_q_producer(int bt, int r) { return bt == 1 ? 17 * r : 8 * r + (r & 7); }

template <int BT>
__global__ __launch_bounds__(FUSED_THREADS, 2) void k_fused_decode_mhaq(FusedArgs a, int batch) {
    using GM = FusedQGeomT<BT>;
    constexpr int R = GM::R;
    constexpr int HID = 4096;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    h16* s_img = reinterpret_cast<h16*>(smem + GM::L_IMG) + wave * PROJ_LDS_WAVE;
    float* s_part = reinterpret_cast<float*>(smem + GM::L_PART);              // [8][BT][256]
    float* s_ss = reinterpret_cast<float*>(smem + GM::L_SS);
    int* s_S = reinterpret_cast<int*>(smem + GM::L_TAB);                      // cached tokens of every row
    int* s_ent = s_S + R + 32;                                                // first page-table entry of every row
    int64_t* s_roff = reinterpret_cast<int64_t*>(s_ent + R);                  // RoPE row offset of every row
    int* s_pre = reinterpret_cast<int*>(smem + GM::L_PRE);
    int* s_ctl = reinterpret_cast<int*>(smem + GM::L_CTL);
    float* s_qkv = reinterpret_cast<float*>(smem + GM::L_QKV);                // [R][384]
    float* s_cs = reinterpret_cast<float*>(smem + GM::L_CS);                  // [R][256]
    float(*s_o)[HEAD_DIM] = reinterpret_cast<float(*)[HEAD_DIM]>(smem + GM::L_O);
    float(*s_ml)[2] = reinterpret_cast<float(*)[2]>(smem + GM::L_ML);
    float* s_st = reinterpret_cast<float*>(smem + GM::L_ST);
    unsigned* s_rec = reinterpret_cast<unsigned*>(smem + GM::L_REC);

    const int r16 = lane & 15, kq = lane >> 4;            // MFMA operand coordinates: row / batch column, k-group
    const int l16 = r16, gid = wave * 4 + kq, d0 = l16 * 8;   // phase 2: 16 lanes x 8 dims per token row, 32 token rows per step
    const int b = blockIdx.x;
    const int h = (b & 7) * 4 + (b >> 6);                 // the 8 workgroups of a head share b % 8 (one XCD: speed only)
    const int j = (b >> 3) & 7;
    CF_TRACE(0);

    const unsigned epoch = scalar_load(a.state) + 1u;
    const h16* kc = a.kptrs ? reinterpret_cast<const h16*>(scalar_load(a.kptrs + a.layer_id)) : a.k_cache;
    const h16* vc = a.vptrs ? reinterpret_cast<const h16*>(scalar_load(a.vptrs + a.layer_id)) : a.v_cache;
    const int ps = a.page_shift, pmask = (1 << ps) - 1;
    // every row's length, first page-table entry and RoPE row (threads 0 .. 15: one small load each, ahead of everything else)
    int tS = 0, tE = 0;
    int64_t tR = 0;
    if (tid < batch) {
        tE = a.indptr[tid];
        tS = a.seq_lens ? a.seq_lens[tid] : a.indptr[tid + 1] - 1 - tE;
        tR = a.positions ? a.positions[tid] * a.rope_stride : 0;
    }

    // ---- weight tiles: 16 rows of this workgroup's Wqkv share, one 1-KB row slice per instruction (lane l: 16 bytes at
    //      column 512 w + 8 l) ---------------------------------------------------------------------------------------------
    const int kw = wave * 512;
    h16x8 wa[16], wb[BT == 1 ? 16 : 1];      // (two batch tiles: one weight tile in flight)
    auto load_w = [&](h16x8 (&t)[16], const h16* W, int row0) {
        const h16* p = W + (size_t)row0 * HID + kw + lane * 8;
#pragma unroll
        for (int i = 0; i < 16; ++i) t[i] = ld_stream(p + (size_t)i * HID);
    };
    // Phase-1 shares: workgroup b owns the Wqkv rows [p1_start[b], p1_start[b + 1]) -- any workgroup can produce any row (the
    // X1 consumers find q|k|v by granule address), so the split is a free load-balancing knob filled in by the host.  Up to
    // four 16-row tiles; rows come through a buffer resource, a row beyond the share gets an offset beyond the buffer: the
    // instruction still issues (one code path, exact wait counts) but touches no memory and returns zeros.
    const int r_lo = a.p1_start[b], r_hi = a.p1_start[b + 1];
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16*>(a.Wqkv), 0, 3 * HID * HID * 2, 0x00020000);
    // 17 .. 32 rows (BT = 2): the Wqkv rows are dealt DENSE (cf_fused_kernel.h) -- tile t < 3 of workgroup b is rows 4096 t + 16 b ..
    // + 16, equal shares, the chip reads one third of the matrix front to back -- instead of by the host's share table: 93.9 / 143.0 ->
    // 93.2 / 141.5 us at 17 / 32 rows of 1024 tokens.  With one batch tile (5 .. 16 rows) the table stays: dense measured +0.4 us at
    // 5 / 8 rows of 1024 tokens and +2.4 at 8 rows of 4096 (profiles/r05_experiments.md section 12).
#ifndef CF_Q_DENSE2
#define CF_Q_DENSE2 1
#endif
    constexpr bool QD = CF_Q_DENSE2 && BT == 2;
    auto p1_row0 = [&](int tile) { return QD ? 4096 * tile + 16 * b : r_lo + 16 * tile; };
    auto p1_live = [&](int tile, int i) { return QD ? tile < 3 : r_lo + 16 * tile + i < r_hi; };
    auto load_p1 = [&](auto& t, int tile) {
        const int row0 = p1_row0(tile);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int voff = p1_live(tile, i) ? (row0 + i) * (HID * 2) + (kw + lane * 8) * 2 : 0x40000000;
            t[i] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, voff, 0, 2 /* nt */));
        }
    };
    auto stage_row_table = [&]() __attribute__((always_inline)) {
        if (tid < R) {
            s_S[tid] = tS;
            s_ent[tid] = tE;
            s_roff[tid] = tR;
        }
    };

    // ---- phase 0 / X0: the fused add + RMSNorm of row r is computed ONCE, by workgroup 17 r (the producers sit on different
    //      XCDs), rounded once to fp16 (kernel.cuh:133-138) and handed to everybody as 8 KB of write-through stores behind one
    //      flag (guide G16 "R1"); the producer also writes the row of residual_out.  Every workgroup normalising all B rows
    //      itself (the first version) pulled B x 16 KB of x and residual through its request pipe -- 128 KB at 8 rows, a quarter
    //      of its weight stream -- where the normalised rows are B x 8 KB; the hop hides behind the first weight tile, which
    //      every consumer requests before it waits.  (A producer requests its tiles after it has published: a drained queue is
    //      what orders the flag behind the payload.  The host gives the producers a smaller phase-1 share.) ------------------------
    const __amdgpu_buffer_rsrc_t xn_rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(a.g_qkv_io), 0, batch * HID * 2, 0x00020000);
    u64* xn_flags = a.g_qkv_io + (size_t)batch * (HID * 2 / 8);
    const int prow = BT == 1 ? b / 17 : b >> 3;
    const bool producer = b == fused_q_producer(BT, prow) && prow < batch;      // (workgroup-uniform)
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
        stage_row_table();
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
        stage_row_table();
        lds_barrier();
    }

    // ---- this workgroup's range of the head's token sequence (every value below is workgroup-uniform) -----------------------------
    //   P[r] = tokens of the rows before r; T = all of them; range [lo, hi) = [j c, (j + 1) c), c = a multiple of 128 >= T / 8
    auto S_of = [&](int r) __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(s_S[r]); };
    int T = 0;
    for (int r = 0; r < batch; ++r) T += S_of(r);
    int cpw = (((T + 7) >> 3) + 127) & ~127;
    cpw = cpw < 128 ? 128 : cpw;
    const int lo = j * cpw, hi = lo + cpw < T ? lo + cpw : T;      // (lo >= T: nothing to stream)
    auto owner_of = [&](int p_r) __attribute__((always_inline)) { const int ow = p_r / cpw; return ow < 7 ? ow : 7; };      // the workgroup where a row begins
    // first row with tokens in the range, and the rows whose q|k|v this workgroup needs (owned rows, and the row its range
    // starts inside): [r_base, r_end)
    int r_first = batch, p_first = 0, r_base = batch, r_end = 0;
    {
        int p = 0;
        for (int r = 0; r < batch; ++r) {
            const int s = S_of(r);
            const bool inside = s > 0 && p < hi && p + s > lo;
            if (inside && r_first == batch) { r_first = r; p_first = p; }
            if (inside || owner_of(p) == j) {
                r_base = r < r_base ? r : r_base;
                r_end = r + 1;
            }
            p += s;
        }
    }
    // the cache rows of the range's first 512 tokens (thread i: token lo + i), requested now, staged in LDS below: the two
    // tiles that go out before X1 must not wait for page numbers in the middle of the weight stream
    int pre_reg = 0;
    if (lo + tid < hi) {
        int p = 0, r = 0;
        for (; r < batch - 1; ++r) {
            const int s = s_S[r];
            if (lo + tid < p + s) break;
            p += s;
        }
        const int tok = lo + tid - p;
        pre_reg = (a.indices[s_ent[r] + (tok >> ps)] << ps) + (tok & pmask);
    }

    // the B operand of this wavefront's K-slice: bx[s] = xn[row r16][512 w + 32 s + 8 kq .. + 8); rows >= batch are zero.
    // (One weight tile is in flight while the flags are awaited -- loads return in issue order, so the poll comes back behind it,
    //  ~5 us into the kernel, when the rows are long published; the second tile is requested behind the operand loads.)
    h16x8 bx[BT][16];
    {
        bool ok = false;
        for (unsigned spin = 0; spin < FUSED_SPIN_LIMIT; ++spin) {
            u64 x = (u64)epoch << 32;
            if (lane < batch) x = __hip_atomic_load(xn_flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all((unsigned)(x >> 32) == epoch)) { ok = true; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (!ok && lane == 0) flag_exchange_error(a.state + 1, 6u);
        // (the payload loads below are sc1 -- they bypass this CU's L1 and are issued, in program order, behind the poll that saw
        //  the flag -- so no cache maintenance is needed; the barrier keeps the COMPILER from hoisting them above the spin loop)
        asm volatile("" ::: "memory");
#pragma unroll
        for (int t = 0; t < BT; ++t) {
            const bool nlive = 16 * t + r16 < batch;
            const int off = ((nlive ? 16 * t + r16 : 0) * HID + kw + kq * 8) * 2;
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xn_rsrc, off, 64 * s2, 16 /* sc1 */);
                bx[t][s2] = nlive ? __builtin_bit_cast(h16x8, v) : h16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
        if (lane == 0) s_ctl[40 + wave] = ok;
    }
    if constexpr (BT == 1) load_p1(wb, 1);      // (requesting it ahead of the poll instead measured the same: 53.3 us at 8 rows either way)
    CF_TRACE(14);   // operand ready

    // ---- phase 1: four 16-row tiles; tile -> image -> 16 MFMAs -> the 8 K-slices meet in LDS -> granules of (row, head) -----
    auto to_image = [&](const auto& t) {
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
    auto publish_qkv = [&](const f32x4_t (&d)[BT], int tile) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < BT; ++t) *reinterpret_cast<f32x4_t*>(&s_part[(wave * BT + t) * 256 + lane * 4]) = d[t];
        lds_only_barrier();
        if (tid < 256 * BT) {
            const int bt = tid >> 8, e = tid & 255;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += s_part[(w * BT + bt) * 256 + e];      // fixed order
            const int l = e >> 2, i = e & 3, n = 16 * bt + (l & 15), m = 4 * (l >> 4) + i;
            const int row = p1_row0(tile) + m;                           // Wqkv row: q of all heads | k | v
            if (n < batch && p1_live(tile, m))
                granule_store(a.g_qkv + ((size_t)n * FUSED_HEADS + ((row & 4095) >> 7)) * 384 + (row >> 12) * 128 + (row & 127), epoch, v);
        }
        lds_only_barrier();
    };

    // ---- the K/V stream of phase 2: tiles of 128 tokens of one row, in range order -------------------------------------------------
    const size_t kvstride = (size_t)FUSED_HEADS * HEAD_DIM;
    const h16* kbase = kc + h * HEAD_DIM + d0;
    const h16* vbase = vc + h * HEAD_DIM + d0;
    constexpr int UT = CF_Q_UT, TILE = FUSED_GROUPS * UT;      // two tiles in flight per wavefront
    typedef KvTile32<UT> Tile;
    // tile cursor: row (-1: past the end of the range), row-local first token of the tile, end of the row's part inside the
    // range, P[row]
    struct Cur { int row, t, e, p; };
    auto seg_cur = [&](int r, int p_r) __attribute__((always_inline)) -> Cur {      // first tile of the part of row r inside the range
        const int s = S_of(r);
        Cur c;
        c.row = r;
        c.p = p_r;
        c.t = lo > p_r ? lo - p_r : 0;
        c.e = hi - p_r < s ? hi - p_r : s;
        return c;
    };
    auto advance = [&](Cur c) __attribute__((always_inline)) -> Cur {
        if (c.row < 0) return c;
        if (c.t + TILE < c.e) { c.t += TILE; return c; }
        int p = c.p + S_of(c.row);
        for (int r = c.row + 1; r < batch; ++r) {
            const int s = S_of(r);
            if (s > 0 && p < hi) return seg_cur(r, p);
            p += s;
        }
        return Cur{-1, 0, 0, 0};
    };
    // page numbers of a tile's tokens (this lane-group's UT tokens) from the page table through L2: REQUESTED here, turned into
    // cache rows only where they are used (arithmetic on the loaded value here would wait for the request -- and, loads
    // returning in order, for every K/V tile in flight)
    auto tile_pages = [&](const Cur& c, int (&pages)[UT]) {
        const int ent = c.row < 0 ? 0 : __builtin_amdgcn_readfirstlane(s_ent[c.row]);
#pragma unroll
        for (int u = 0; u < UT; ++u) {
            int tk = c.t + u * FUSED_GROUPS + gid;
            tk = tk < c.e ? tk : c.e - 1;
            tk = c.row < 0 ? 0 : tk;
            pages[u] = a.indices[ent + (tk >> ps)];      // (a cursor past the end reads entry 0 of the table: harmless, never used)
        }
    };
    auto pages_to_rows = [&](const Cur& c, const int (&pages)[UT], int (&rows)[UT]) {
#pragma unroll
        for (int u = 0; u < UT; ++u) {
            int tk = c.t + u * FUSED_GROUPS + gid;
            tk = tk < c.e ? tk : c.e - 1;
            rows[u] = c.row < 0 ? 0 : (pages[u] << ps) + (tk & pmask);
        }
    };
    auto staged_rows = [&](const Cur& c, int (&rows)[UT]) {      // ... of one of the range's first 256 tokens: from LDS
#pragma unroll
        for (int u = 0; u < UT; ++u) {
            int tk = c.t + u * FUSED_GROUPS + gid;
            tk = tk < c.e ? tk : c.e - 1;
            int ri = c.row < 0 ? 0 : c.p + tk - lo;
            ri = ri < 511 ? ri : 511;
            rows[u] = s_pre[ri > 0 ? ri : 0];
        }
    };
    auto load_kv = [&](Tile& t, const int (&rows)[UT]) {      // (a cursor past the end reads cache row 0: harmless, never used)
#pragma unroll
        for (int u = 0; u < UT; ++u) {
            t.k[u] = ld_stream(kbase + (size_t)rows[u] * kvstride);
            t.v[u] = ld_stream(vbase + (size_t)rows[u] * kvstride);
        }
    };
    Tile ta, tb;
    Cur cA = r_first < batch ? seg_cur(r_first, p_first) : Cur{-1, 0, 0, 0};
    Cur cB = advance(cA), cN = advance(cB);
    int npages[UT];     // page numbers of tile cN: requested one issue ahead of the tile they address
    auto mfma_tiles = [&](f32x4_t (&d)[BT]) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < BT; ++t) d[t] = mfma_tile(bx[t]);
    };
    if constexpr (BT == 1) {
        f32x4_t d[1];
        to_image(wa);
        load_p1(wa, 2);
        mfma_tiles(d);
        s_pre[tid] = pre_reg;      // (visible behind the barriers of the publish below)
        publish_qkv(d, 0);
        to_image(wb);
        load_p1(wb, 3);
        mfma_tiles(d);
        publish_qkv(d, 1);
        to_image(wa);
        {   // (unconditional requests: a branch here would join two different queue depths and make the next image wait for the tiles)
            int rows[UT];
            staged_rows(cA, rows);
            load_kv(ta, rows);
        }
        mfma_tiles(d);
        publish_qkv(d, 2);
        to_image(wb);
        {   // (the second tile ends at most 2 x TILE <= 512 tokens into the range: staged as well)
            int rows[UT];
            staged_rows(cB, rows);
            load_kv(tb, rows);
        }
        tile_pages(cN, npages);
        mfma_tiles(d);
        publish_qkv(d, 3);       // (ends with a barrier: every wavefront is done with its image -- phase 2 reuses the area)
    } else {
        // two batch tiles: ONE weight tile in flight -- requested as soon as the previous one sits in the image -- beside the
        // 128 registers of the activation operand; the K/V tiles go out behind the last weight tile
        f32x4_t d[BT];
        to_image(wa);
        load_p1(wa, 1);
        mfma_tiles(d);
        s_pre[tid] = pre_reg;
        publish_qkv(d, 0);
        to_image(wa);
        load_p1(wa, 2);
        mfma_tiles(d);
        publish_qkv(d, 1);
        to_image(wa);
        load_p1(wa, 3);
        mfma_tiles(d);
        publish_qkv(d, 2);
        to_image(wa);
        {
            int rows[UT];
            staged_rows(cA, rows);
            load_kv(ta, rows);
            staged_rows(cB, rows);
            load_kv(tb, rows);
        }
        tile_pages(cN, npages);
        mfma_tiles(d);
        publish_qkv(d, 3);
    }
    {   // (X0 gave up somewhere: the publishes above ended with barriers, the flags are visible)
        bool all_ok = true;
        for (int w = 0; w < 8; ++w) all_ok &= s_ctl[40 + w] != 0;
        if (!all_ok) CF_FAIL_RETURN();
    }
    CF_TRACE(1);   // phase 1 done

    // ---- X1 + RoPE rows of every row this workgroup touches (one parallel round: wavefront w takes rows r_base + w, + 8) ----------
    {
        // RoPE rows: [0,128) cos, [128,256) sin per row (NEOX reads 64 of each)
        const int n_ang = a.rope_style == 0 ? HEAD_DIM / 2 : HEAD_DIM;
        float cs_reg[8 * BT];
#pragma unroll
        for (int i = 0; i < 8 * BT; ++i) {
            const int idx = i * FUSED_THREADS + tid, slot = idx >> 8, t = idx & 255, r = r_base + slot;
            cs_reg[i] = 0.f;
            if (r < r_end) {
                const int64_t ro = s_roff[r];
                if (t < n_ang) cs_reg[i] = a.cos[ro + t];
                else if (t >= 128 && t < 128 + n_ang) cs_reg[i] = a.sin[ro + t - 128];
            }
        }
        int newtok_page = 0;      // the page of every touched row's new token (no request may sit behind a branch inside phase 2)
        if (tid < R && r_base + tid < r_end) newtok_page = a.indices[s_ent[r_base + tid] + (s_S[r_base + tid] >> ps)];
        bool ok = true;
        for (int rr = r_base + wave; rr < r_end; rr += 8)
            ok &= sweep_granules<6>(a.g_qkv + ((size_t)rr * FUSED_HEADS + h) * 384, 384, epoch, s_qkv + (rr - r_base) * 384, lane, a.state + 1, 1u);
        if (lane == 0) s_ctl[wave] = ok;
#pragma unroll
        for (int i = 0; i < 8 * BT; ++i) s_cs[i * FUSED_THREADS + tid] = cs_reg[i];
        if (tid < R) s_pre[tid] = newtok_page;      // (the staged page numbers of the first two tiles are consumed: the area is free)
        lds_barrier();
        bool all_ok = true;
        for (int w = 0; w < 8; ++w) all_ok &= s_ctl[w] != 0;
        if (!all_ok) CF_FAIL_RETURN();
    }
    CF_TRACE(2);   // X1 resolved

    // ---- phase 2 ----------------------------------------------------------------------------------------------------------------
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
    h16x8 qh = {0, 0, 0, 0, 0, 0, 0, 0};
    float m = NEG_BIG, l = 0.f, o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int deferred = -1, deferred_last = 0;      // the owned row that continues in later ranges; the last workgroup that holds a part of it
    auto begin_row = [&](int r, int p_r) __attribute__((always_inline)) {
        float q[8];
        rope_lds(s_qkv + (r - r_base) * 384, s_cs + (r - r_base) * 256, q);
#pragma unroll
        for (int e = 0; e < 8; ++e) qh[e] = (h16)(q[e] * qscale);
        m = NEG_BIG;
        l = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = 0.f;
    };
    auto compute_tile = [&](const Tile& t, const Cur& c) __attribute__((always_inline)) {
        float s[UT];
        bool valid[UT];
        float mx = NEG_BIG;
#pragma unroll
        for (int u = 0; u < UT; ++u) {
            valid[u] = (c.t + u * FUSED_GROUPS + gid) < c.e;
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
    // the part of row r this workgroup streamed is done: its 8 wavefront states (+ the new token when this workgroup owns the
    // row) meet in LDS; then the row is final here (payload out), or leaves as a record, or waits for the later parts
    auto finish_row = [&](int r, int p_r) __attribute__((always_inline)) {
        const int Sr = S_of(r);
        const bool own = owner_of(p_r) == j;
        const bool ends_here = p_r + Sr <= hi || Sr == 0;      // no later range holds a part of it
        {
            const float mw = xmax32(xmax16(m));
            const float sc = fast_exp2(m - mw);
            const float lw = xsum32(xsum16(l * sc));
            float ov[8], r0, r1;
#pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] = o[e] * sc;
            xsum_rows8(ov, r0, r1);      // (row x of the wavefront ends up with dims d0 + xrow_e(x) and d0 + 4 + xrow_e(x))
            const int e0 = xrow_e(lane >> 4);
            s_o[wave][d0 + e0] = r0;
            s_o[wave][d0 + 4 + e0] = r1;
            if (lane == 0) { s_ml[wave][0] = mw; s_ml[wave][1] = lw; }
        }
        // the new token of the row (attended from registers, kernel.cuh:444-477) + k/v export + cache write: the owner
        if (own && gid == 0) {
            const float* qkv = s_qkv + (r - r_base) * 384;
            const float* cs = s_cs + (r - r_base) * 256;
            float kf[8], vf[8], q[8];
            rope_lds(qkv + HEAD_DIM, cs, kf);
            rope_lds(qkv, cs, q);
#pragma unroll
            for (int e = 0; e < 8; ++e) vf[e] = qkv[2 * HEAD_DIM + d0 + e];
            h16x8 k16, v16;
#pragma unroll
            for (int e = 0; e < 8; ++e) { k16[e] = (h16)kf[e]; v16[e] = (h16)vf[e]; }
            const size_t ooff = (size_t)h * HEAD_DIM + d0;
            if (a.k_new) st_h8(a.k_new + (size_t)r * kvstride + ooff, k16);
            if (a.v_new) st_h8(a.v_new + (size_t)r * kvstride + ooff, v16);
            if (a.write_cache) {
                const size_t slot = ((size_t)s_pre[r - r_base] << ps) + (size_t)(Sr & pmask);      // (the new token's page: staged with X1)
                st_h8(const_cast<h16*>(kc) + slot * kvstride + ooff, k16);
                st_h8(const_cast<h16*>(vc) + slot * kvstride + ooff, v16);
            }
            float sn = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) sn = __builtin_fmaf(q[e] * qscale, kf[e], sn);
            sn = sum16(sn);
#pragma unroll
            for (int e = 0; e < 8; ++e) s_o[8][d0 + e] = vf[e];
            if (l16 == 0) { s_ml[8][0] = sn; s_ml[8][1] = 1.f; }
        }
        lds_barrier();
        if (tid < HEAD_DIM) {
            const int nst = own ? 9 : 8;
            float M = NEG_BIG;
#pragma unroll
            for (int i = 0; i < 9; ++i) M = fmaxf(M, i < nst ? s_ml[i][0] : NEG_BIG);
            float acc = 0.f, L = 0.f;
#pragma unroll
            for (int i = 0; i < 9; ++i)
                if (i < nst) {
                    const float wt = fast_exp2(s_ml[i][0] - M);
                    acc = __builtin_fmaf(wt, s_o[i][tid], acc);
                    L = __builtin_fmaf(wt, s_ml[i][1], L);
                }
            const float mine = L > 0.f ? acc / L : 0.f;      // normalised: a convex combination of V rows
            if (own && ends_here) {
                // payload: 256 bytes of fp16 per (row, head), sixteen 16-byte WRITE-THROUGH (sc1) stores (guide G16 "R1"); the
                // flag goes out at the end of phase 2, behind a drained queue
                h16x8 pk;
                pk[0] = (h16)mine;
#pragma unroll
                for (int e = 1; e < 8; ++e) pk[e] = (h16)__shfl_down(mine, e);
                if (!(tid & 7))
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pk), x3_rsrc, (r * HID + h * HEAD_DIM + tid) * 2, 0, 16 /* sc1 */);
            } else if (own) {      // continues in later ranges: keep the state, the later parts' records are gathered at the end
                s_st[tid] = mine;
                if (tid == 0) { s_st[HEAD_DIM] = M; s_st[HEAD_DIM + 1] = L; }
            } else {               // a part of a row that began in an earlier range: ONE record (fp16 pairs of o / l, then m, l)
                u64* rec = a.g_rec + (((size_t)r * FUSED_HEADS + h) * FUSED_SPLITS + j) * FUSED_RECH;
                const float next = __shfl_down(mine, 1);
                h16x2 pr;
                pr[0] = (h16)mine;
                pr[1] = (h16)next;
                if (!(tid & 1)) granule_store(rec + (tid >> 1), epoch, __builtin_bit_cast(float, pr));
                if (tid == 0) {
                    granule_store(rec + HEAD_DIM / 2, epoch, M);
                    granule_store(rec + HEAD_DIM / 2 + 1, epoch, L);
                }
            }
        }
        if (own && !ends_here) {
            deferred = r;
            const int jl = (p_r + Sr - 1) / cpw;
            deferred_last = jl < 7 ? jl : 7;
        }
        lds_barrier();      // the states are read: the next row may overwrite them
    };
    // one issue: the buffer whose tile was just consumed takes the tile after next of the range.  Its page numbers were
    // requested one issue ago; the following tile's are requested now, AHEAD of this tile's K/V requests: loads return in issue
    // order, so the next issue waits for nothing younger than them.
    auto issue = [&](Tile& t, Cur& c) __attribute__((always_inline)) {
        int rows[UT];
        pages_to_rows(cN, npages, rows);
        c = cN;
        cN = advance(cN);
        tile_pages(cN, npages);
        load_kv(t, rows);
    };
    // one row's part inside the range: its tiles alternate between the two buffers, starting with X.  The pair loop has no
    // request or wait behind a branch (a join would merge different queue depths and make every tile wait for everything in
    // flight); the row boundaries -- q of the row, the merge of its states -- sit outside it.  Returns whether the row had an odd
    // number of tiles (the next row then starts in Y).
    auto process_row = [&](Tile& X, Cur& cX, Tile& Y, Cur& cY) __attribute__((always_inline)) -> bool {
        const int row = cX.row, p_r = cX.p;
        int n = (cX.e - cX.t + TILE - 1) / TILE;
        begin_row(row, p_r);
        for (; n >= 2; n -= 2) {
            compute_tile(X, cX);
            issue(X, cX);
            compute_tile(Y, cY);
            issue(Y, cY);
        }
        const bool odd = n == 1;
        if (odd) {
            compute_tile(X, cX);
            issue(X, cX);
        }
        finish_row(row, p_r);
        return odd;
    };
    // phase 3's weights: this workgroup's 16 rows of Wo.  Requested behind the workgroup's X3 flags by every wavefront: the flags
    // need the two publishing wavefronts' queues drained and a barrier of all eight, and a wavefront that requests its rows in
    // front of that barrier stands ~4 us at the request instructions (the CU admits ~25 GB/s) while the flags wait for it
    // (same-box alternations against "wavefronts 2 .. 7 behind the range": 8 rows of 1024 tokens 54.6-55.0 vs 55.8-56.0 us, of
    // 128 tokens 38.3-38.5 vs 39.7-40.1; 16 rows 85.0-85.4 vs 84.9-85.1 and 47.4-47.9 vs 49.0-49.3; 12 rows 72.7-73.1 vs
    // 71.8-72.4; S = 4096 121.4 / 212.4-213.1 vs 121.5-122.9 / 213.2-214.5).  (Requested HERE they would keep the request pipe full while the two-deep tile loop
    // ramps up -- the loop alone holds 16 KB per wavefront in flight and streams at ~19 GB/s per CU, the version that parked
    // them in LDS at this point reached 25 -- but 64 more registers across the loop do not fit: CF_Q_EARLY_WO spills 48 of them,
    // 256-token tiles spill 192; the LDS images that could park them hold the rows' q|k|v and RoPE rows now.)
    h16x8 go[16];
#ifdef CF_Q_EARLY_WO
    load_w(go, a.Wo, 16 * b);
#endif
    CF_TRACE(7);
    {
        bool in_b = false;      // which buffer holds the next tile to consume
        for (;;) {
            if ((in_b ? cB.row : cA.row) < 0) break;
            const bool odd = in_b ? process_row(tb, cB, ta, cA) : process_row(ta, cA, tb, cB);
            in_b ^= odd;
        }
    }
    CF_TRACE(8);   // the range is streamed
#if !defined(CF_Q_EARLY_WO) && !CF_Q_WO_BEHIND_FLAGS
    if (wave >= 2) load_w(go, a.Wo, 16 * b);      // (the two publishing wavefronts: behind their flags)
#endif
    // owned rows without cached tokens: the new token alone
    {
        int p = 0;
        for (int r = 0; r < batch; ++r) {
            const int s = S_of(r);
            if (s == 0 && owner_of(p) == j) {
                begin_row(r, p);
                finish_row(r, p);
            }
            p += s;
        }
    }
    // the owned row that continued in later ranges: their records (wavefronts 0 and 1 gather: nothing else is in flight there)
    if (deferred >= 0) {
        const int nrec = deferred_last - j;      // 1 .. 7
        if (wave < 2) {
            bool ok = true;
            for (int i = wave; i < nrec; i += 2)
                ok &= sweep_granules_raw<2>(a.g_rec + (((size_t)deferred * FUSED_HEADS + h) * FUSED_SPLITS + j + 1 + i) * FUSED_RECH, FUSED_RECH, epoch,
                                            s_rec + i * FUSED_RECH, lane, a.state + 1, 2u);
            if (lane == 0) s_ctl[8 + wave] = ok;
        }
        lds_barrier();
        if (!s_ctl[8] || !s_ctl[9]) CF_FAIL_RETURN();
        if (tid < HEAD_DIM) {
            float M = s_st[HEAD_DIM];
            for (int i = 0; i < nrec; ++i) M = fmaxf(M, __builtin_bit_cast(float, s_rec[i * FUSED_RECH + HEAD_DIM / 2]));
            float wt = fast_exp2(s_st[HEAD_DIM] - M) * s_st[HEAD_DIM + 1];
            float acc = wt * s_st[tid], L = wt;
            for (int i = 0; i < nrec; ++i) {      // fixed order
                const unsigned* rc = s_rec + i * FUSED_RECH;
                wt = fast_exp2(__builtin_bit_cast(float, rc[HEAD_DIM / 2]) - M) * __builtin_bit_cast(float, rc[HEAD_DIM / 2 + 1]);
                acc = __builtin_fmaf(wt, (float)__builtin_bit_cast(h16x2, rc[tid >> 1])[tid & 1], acc);
                L += wt;
            }
            const float mine = L > 0.f ? acc / L : 0.f;
            h16x8 pk;
            pk[0] = (h16)mine;
#pragma unroll
            for (int e = 1; e < 8; ++e) pk[e] = (h16)__shfl_down(mine, e);
            if (!(tid & 7))
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pk), x3_rsrc, (deferred * HID + h * HEAD_DIM + tid) * 2, 0, 16 /* sc1 */);
        }
    }
    // flags of every owned row, behind the drained payload stores of the two storing wavefronts
    if (wave < 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (inline asm: the compiler cannot drop it)
    lds_barrier();
    if (tid < batch) {
        int p = 0;
        for (int r = 0; r < tid; ++r) p += s_S[r];
        if (owner_of(p) == j) granule_store(x3_flags + (size_t)tid * FUSED_HEADS + h, epoch, 0.f);
    }
#ifndef CF_Q_EARLY_WO
    if (wave < 2 || CF_Q_WO_BEHIND_FLAGS) load_w(go, a.Wo, 16 * b);
#endif
    CF_TRACE(3);   // phase 2 done, attention outputs published

    // ---- X3: heads 4 w .. 4 w + 3 of every row, straight into the B operand of this wavefront's K-slice: lane (n = r16, kq)
    //      watches the flag of (row n, head 4 w + kq); then sixteen 16-byte sc1 loads per lane, no tags to check, one round -----------
    h16x8 ax[BT][16];
    {
        bool ok = false;
        for (unsigned spin = 0; spin < FUSED_SPIN_LIMIT; ++spin) {
            bool here = true;
#pragma unroll
            for (int t = 0; t < BT; ++t) {
                u64 x = (u64)epoch << 32;
                if (16 * t + r16 < batch) x = __hip_atomic_load(x3_flags + (size_t)(16 * t + r16) * FUSED_HEADS + 4 * wave + kq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                here &= (unsigned)(x >> 32) == epoch;
            }
            if (__all(here)) { ok = true; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (!ok && lane == 0) flag_exchange_error(a.state + 1, 3u);
        asm volatile("" ::: "memory");      // (as at X0: the sc1 payload loads stay behind the poll)
#pragma unroll
        for (int t = 0; t < BT; ++t) {
            const bool nlive = 16 * t + r16 < batch;
            const int off = ((nlive ? 16 * t + r16 : 0) * HID + kw + kq * 8) * 2;
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) {
                // (the constant goes into the instruction's scalar offset: as part of the vector offset the two-tile kernel computed
                //  all 32 addresses up front and spilled 13 of them)
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(x3_rsrc, off, 64 * s2, 16 /* sc1: producer wrote through, L1 bypassed */);
                ax[t][s2] = nlive ? __builtin_bit_cast(h16x8, v) : h16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
        if (lane == 0) s_ctl[17 + wave] = ok;
    }
    lds_barrier();
    {
        bool all_ok = true;
        for (int w = 0; w < 8; ++w) all_ok &= s_ctl[17 + w] != 0;
        if (!all_ok) CF_FAIL_RETURN();
    }
    CF_TRACE(5);   // X3 resolved

    // ---- phase 3: out[n][16 b + m] = sum_k attn[n][k] Wo[16 b + m][k] -----------------------------------------------------------
    to_image(go);
    {
#pragma unroll
        for (int t = 0; t < BT; ++t) {
            const f32x4_t d = mfma_tile(ax[t]);
            *reinterpret_cast<f32x4_t*>(&s_part[(wave * BT + t) * 256 + lane * 4]) = d;
        }
        lds_only_barrier();
        if (tid < 256 * BT) {
            const int bt = tid >> 8, e = tid & 255;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += s_part[(w * BT + bt) * 256 + e];      // fixed order
            const int l = e >> 2, i = e & 3, n = 16 * bt + (l & 15), m2 = 4 * (l >> 4) + i;
            if (n < batch) a.out[(size_t)n * HID + 16 * b + m2] = (h16)v;
        }
    }
    if (b == 0 && tid == 0) {
        a.state[0] = epoch;
        a.state[2] = 0u;      // (no length arm to report: cf_workspace_last_arm documents 0 after a multi-row / MLA kernel)
    }
    CF_TRACE(6);
}

}  // namespace cf
