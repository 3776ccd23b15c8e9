// cf_mla_fused.h -- the whole DeepSeek-V2-Lite MLA decode layer in ONE persistent launch (gfx950, 256 CUs).
//
// Same five stages as cf_mla_kernels.h (A input projections, B absorbed query, C attention over the latent cache,
// D W_uv, E W_o), but as roles of 256 co-resident workgroups (one per CU) instead of separate launches:
//
//   workgroup b:   A units b, b + NA, b + 2 NA  (b < NA = 256 - #C workgroups, at least 192)   64-col strip x 256-row K-slice
//                  B unit b - 64            (64 <= b < 192)                  q_abs: head, 64 of 512 latent columns
//                  the small-vector role    (b == 192)                       ckv RMSNorm, RoPE(q_pe), RoPE(k_pe)
//                  C unit (range r, column half) on workgroup 255 - c'      128 tokens per step, 8 wavefronts; the two
//                                                                            halves of a range sit on the same XCD
//                  D unit (b < 128)         head 2 (b % 8) + b / 64, strip, K-slice   merge + W_uv; on the XCD (= b % 8) of
//                                           the E units that consume it: that hand-off stays in the XCD's L2
//                  E unit b                                                  W_o: strip b / 8, K-slice b % 8
//
// A workgroup requests EVERY weight tile of all its roles (and its first latent-cache tile) at kernel start --
// the layer's 27 MB are all in flight within the first microsecond and stream at the full rate while nothing is
// computed yet -- and then walks its roles in dependency order.  Stages meet through tagged granules
// ({epoch, fp32} in one 8-byte write-through store, cf_fused_kernel.h): the data is the flag, every wait is a
// bounded spin on the granules themselves.  All 256 workgroups are resident, and every workgroup runs its roles in
// the stages' topological order, so no wait can depend on a workgroup that has not started.
#pragma once
#include "cf_mla_kernels.h"

namespace cf {

constexpr int MLAF_WGS = 256;
constexpr int MLAF_B_FIRST = 64, MLAF_X_WG = 192;
// dynamic LDS (bytes)
constexpr int MLAF_L_V = 0;                                   // h16 [8 tiles][32 blocks][16 tok][16 col] (C); role scratch (others)
constexpr int MLAF_L_Q = 8 * 16384;                           // h16 [16][MLAF_QROW]   q operand
constexpr int MLAF_QROW = 584;                                //   (576 + 8 pad: rows 16 B apart mod 128)
constexpr int MLAF_L_LAT = MLAF_L_Q + 16 * MLAF_QROW * 2;     // h16 [576]   the new token's latent row
constexpr int MLAF_L_P = MLAF_L_LAT + 1152;                   // h16x4 [8][64]
constexpr int MLAF_L_ML = MLAF_L_P + 8 * 64 * 8;              // float [3][8][16]  tile max | tile sum | rescale
constexpr int MLAF_LDS = MLAF_L_ML + 3 * 128 * 4;             // 156544 B
// role scratch inside the (then idle) V image
constexpr int MLAF_S_X = 0;        // float [3][256]   operand slices
constexpr int MLAF_S_RED = 3072;   // float [8][64]
constexpr int MLAF_S_R8 = 5120;    // float [16]
constexpr int MLAF_S_W = 5248;     // float [256]
constexpr int MLAF_S_XP = 6272;    // float [4][128]
constexpr int MLAF_S_BIG = 8320;   // float [1088]

struct MlaFusedArgs {
    unsigned* state;          // [0] epoch of the last completed call, [1] first error code
    // A
    const h16 *x, *rms_w;
    float eps;
    const h16 *w_q_nope, *w_kv, *w_q_pe, *w_k_pe;
    int n_a;                  // 320, or 456 with the rope parts
    int na_wgs;               // workgroups that take A units (the C workgroups of a short cache take none)
    u64* g_a;                 // [8][3648]
    // B
    const h16* w_uk;
    const h16* rms_ckv_w;
    const float *cos, *sin;
    int with_pe;
    u64* g_q;                 // fp16 pairs: [16][288] q_abs | RoPE(q_pe), then [288] the new token's latent row
    h16* latent_out;
    // C
    const h16* cache;
    const h16* zeros;         // [576] finite filler for rows that are masked or replaced
    int n_tok, iters, nsplit;
    float scale_log2e;
    u64* g_po;                // [nsplit][16][256] fp16 pairs
    u64* g_ml;                // [nsplit][32]
    // D, E
    const h16* w_uv;
    u64* g_d;                 // [4][2048]
    u64* g_xcc;               // [256] XCC id every workgroup runs on
    const h16* w_o;
    u64* g_e;                 // [8][2048]
    h16* out;
    u64* trace;               // debug: [256][16] wall-clock stamps (100 MHz) per workgroup, or null
};

#define MLAF_TRACE(slot)                                                                                  \
    do {                                                                                                  \
        if (a.trace && tid == 0) a.trace[(size_t)blockIdx.x * 16 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)

// two fp16 values in one granule payload (the attention role consumes q in fp16, the matrix cores' operand type: the producer
// rounds, and every attention workgroup sweeps half the granules)
__device__ __forceinline__ float mla_pack2(float lo, float hi) {
    h16x2 pr;
    pr[0] = (h16)lo;
    pr[1] = (h16)hi;
    return __builtin_bit_cast(float, pr);
}
template <bool PE>
__global__ __launch_bounds__(512) void k_mla_fused(MlaFusedArgs a) {
    constexpr int NJ = PE ? 18 : 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    h16* s_v = reinterpret_cast<h16*>(smem + MLAF_L_V);
    h16* s_q = reinterpret_cast<h16*>(smem + MLAF_L_Q);
    h16* s_lat = reinterpret_cast<h16*>(smem + MLAF_L_LAT);
    h16x4_t* s_p = reinterpret_cast<h16x4_t*>(smem + MLAF_L_P);
    float* s_m = reinterpret_cast<float*>(smem + MLAF_L_ML);
    float* s_l = s_m + 128;
    float* s_al = s_l + 128;
    float* s_x = reinterpret_cast<float*>(smem + MLAF_S_X);
    float (*s_red)[64] = reinterpret_cast<float (*)[64]>(smem + MLAF_S_RED);
    float* s_r8 = reinterpret_cast<float*>(smem + MLAF_S_R8);
    float* s_w = reinterpret_cast<float*>(smem + MLAF_S_W);
    float (*s_xp)[128] = reinterpret_cast<float (*)[128]>(smem + MLAF_S_XP);
    float* s_big = reinterpret_cast<float*>(smem + MLAF_S_BIG);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    const int t16 = lane & 15, kq = lane >> 4;
    const unsigned epoch = a.state[0] + 1u;
    unsigned* err = a.state + 1;

    MLAF_TRACE(0);
    // where this workgroup runs (decides whether the D -> E hand-off may stay inside the XCD's L2)
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 15u;
    if (tid == 0) mla_granule_store(a.g_xcc + b, epoch, __builtin_bit_cast(float, xcc));
    // ---- roles of this workgroup ------------------------------------------------------------------------------------
    const bool has_a1 = b < a.na_wgs, has_a2 = has_a1 && b + a.na_wgs < a.n_a, has_a3 = has_a1 && b + 2 * a.na_wgs < a.n_a;
    const bool has_b = b >= MLAF_B_FIRST && b < MLAF_B_FIRST + 128;
    const bool has_x = b == MLAF_X_WG;
    const int cc = MLAF_WGS - 1 - b;                                  // c' = (r / 8) * 16 + half * 8 + r % 8
    const int c_unit = (cc >> 4) * 8 + (cc & 7), c_half = (cc >> 3) & 1;
    const bool has_c = c_unit < a.nsplit;
    const bool has_d = b < 128;

    // ---- the input vector first (loads return in issue order: the norm must not queue behind the tiles) -------------
    float ss_x, xk, wk;
    {
        const h16x2* xp = reinterpret_cast<const h16x2*>(a.x) + tid * 2;      // 4 halves per thread
        const h16x2 v0 = xp[0], v1 = xp[1];
        const float x0 = (float)v0[0], x1 = (float)v0[1], x2 = (float)v1[0], x3 = (float)v1[1];
        ss_x = x0 * x0 + x1 * x1 + x2 * x2 + x3 * x3;
    }
    // the operand slices of this workgroup's A units (x and the norm weights at the units' K-slices): requested NOW -- behind
    // the tile requests below they would come back with the last tile (loads return in issue order) and hold the norm up
    float xk3 = 0.f, wk3 = 0.f;
    {
        const int u = tid < 256 ? b : b + a.na_wgs;
        const int k = (u & 7) * 256 + (tid & 255);
        const bool live = has_a1 && (tid < 256 || has_a2);
        xk = live ? (float)a.x[k] : 0.f;
        wk = live ? (float)a.rms_w[k] : 0.f;
        if (has_a3 && tid < 256) {
            const int k3 = ((b + 2 * a.na_wgs) & 7) * 256 + tid;
            xk3 = (float)a.x[k3];
            wk3 = (float)a.rms_w[k3];
        }
    }
    // ---- every tile this workgroup will ever need is requested now -------------------------------------------------
    ColTile<4> ta1, ta2, ta3, te;
    ColTile<2> tb2, td;
    auto a_tile = [&](ColTile<4>& t, int u) {
        const int strip = u >> 3, ks = u & 7;
        const h16* W;
        int ld, col0;
        if (strip < 32) { W = a.w_q_nope; ld = MLA_H * MLA_NOPE; col0 = 64 * strip; }
        else if (strip < 40) { W = a.w_kv; ld = MLA_L; col0 = 64 * (strip - 32); }
        else if (strip < 56) { W = a.w_q_pe; ld = MLA_H * MLA_ROPE; col0 = 64 * (strip - 40); }
        else { W = a.w_k_pe; ld = MLA_ROPE; col0 = 0; }
        t.load(W + (size_t)(ks * 256 + wave * 32) * ld + col0, ld, lane);
    };
    h16x8 av[NJ];
    auto c_tile = [&](int it) {
        const int t = (c_unit * a.iters + it) * 128 + wave * 16 + t16;
        const h16* rowp = t < a.n_tok - 1 ? a.cache + (size_t)t * MLA_LAT : a.zeros;     // new / masked rows: filler
#pragma unroll
        for (int j = 0; j < NJ; ++j) av[j] = ld_stream(rowp + 32 * j + 8 * kq);
    };
    if (has_a1) a_tile(ta1, b);
    if (has_a2) a_tile(ta2, b + a.na_wgs);
    if (has_a3) a_tile(ta3, b + 2 * a.na_wgs);
    const int bh = (b - MLAF_B_FIRST) >> 3, bc0 = 64 * ((b - MLAF_B_FIRST) & 7);
    // D unit of workgroup b < 128: the two heads 2 x, 2 x + 1 of XCD x = b % 8 -- exactly what the E units (strip, K-slice x)
    // of that XCD consume
    const int du = b >> 3;                                            // 0..15
    const int dh = 2 * (b & 7) + (du >> 3), dc2 = (du >> 2) & 1, dks = du & 3;
    const int estrip = b >> 3, eks = b & 7;
    u64 member_x = 0;
    // The tiles of the later roles go out behind the A tiles: workgroups with an A role request them after the norm's barrier
    // (the first hand-off waits for the slowest A unit; -0.2 us per layer).  Delaying the attention workgroups' requests by
    // the clock as well (1 / 2 us) changed nothing: the A phase is paced by each CU's own admission rate, not by queue order.
    auto later_tiles = [&]() {
        if (has_b) tb2.load(a.w_uk + (size_t)(wave * 16) * (MLA_H * MLA_L) + bh * MLA_L + bc0, MLA_H * MLA_L, lane);
        if (has_c) c_tile(0);
        if (has_d)
            td.load(a.w_uv + (size_t)(dks * 128 + wave * 16) * (MLA_H * MLA_NOPE) + dh * MLA_NOPE + 64 * dc2, MLA_H * MLA_NOPE, lane);
        te.load(a.w_o + (size_t)(eks * 256 + wave * 32) * MLA_HID + 64 * estrip, MLA_HID, lane);
        // ids of the 32 workgroups of this XCD position (lane i % 32: workgroup 8 i + b % 8), behind every tile request
        member_x = __hip_atomic_load(a.g_xcc + (((lane & 31) << 3) | (b & 7)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    if (!has_a1) later_tiles();

    MLAF_TRACE(1);   // all tiles requested
    // ---- A ------------------------------------------------------------------------------------------------------------------
    if (has_a1) {
        float ss = sum64(ss_x);
        if (lane == 0) s_r8[wave] = ss;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) tot += s_r8[w];
        const float rcp = __builtin_amdgcn_rsqf(tot / (float)MLA_HID + a.eps);
        // operand slices of the (up to) three units: threads 0..255 -> s_x[0..256), 256..511 -> unit 2; unit 3 second pass
        s_x[tid] = xk * rcp * wk;
        if (has_a3 && tid < 256) s_x[512 + tid] = xk3 * rcp * wk3;
        later_tiles();
        __syncthreads();
        MLAF_TRACE(2);   // norm done
        auto a_unit = [&](const ColTile<4>& t, int u, const float* xs) {
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            t.fma(xs + wave * 32, lane, acc);
            const float v = strip_reduce(acc, s_red, tid);
            if (tid < 64) mla_granule_store(a.g_a + (size_t)(u & 7) * MLA_A_COLS + 64 * (u >> 3) + tid, epoch, v);
            __syncthreads();                                        // s_red free again
        };
        a_unit(ta1, b, s_x);
        MLAF_TRACE(3);   // first A unit published
        if (has_a2) a_unit(ta2, b + a.na_wgs, s_x + 256);
        if (has_a3) a_unit(ta3, b + 2 * a.na_wgs, s_x + 512);
    }

    MLAF_TRACE(4);   // A published
    // ---- B ------------------------------------------------------------------------------------------------------------------
    if (has_b) {
        const float q = mla_granule_sum<MLA_A_KS>(a.g_a + bh * MLA_NOPE + tid, MLA_A_COLS, epoch, tid < MLA_NOPE, err, 1u);
        if (tid < MLA_NOPE) s_x[tid] = q;
        __syncthreads();
        MLAF_TRACE(5);   // q_nope arrived
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        tb2.fma(s_x + wave * 16, lane, acc);
        const float v = strip_reduce(acc, s_red, tid);
        {   // g_q holds fp16 PAIRS: [16][288] q_abs | RoPE(q_pe), then [288] the new token's latent row
            const float vn = __shfl_down(v, 1);
            if (tid < 64 && !(tid & 1)) mla_granule_store(a.g_q + ((bh * MLA_LAT + bc0 + tid) >> 1), epoch, mla_pack2(v, vn));
        }
        __syncthreads();
    }
    if (has_x) {
        const float rw = (float)a.rms_ckv_w[tid];
        const float ckv = mla_granule_sum<MLA_A_KS>(a.g_a + MLA_A_CKV + tid, MLA_A_COLS, epoch, true, err, 2u);   // 512 threads = 512 dims
        float ss = sum64(ckv * ckv);
        if (lane == 0) s_r8[wave] = ss;
        if (a.with_pe) {
            s_big[tid] = mla_granule_sum<MLA_A_KS>(a.g_a + MLA_A_QPE + tid, MLA_A_COLS, epoch, true, err, 2u);
            s_big[512 + tid] = mla_granule_sum<MLA_A_KS>(a.g_a + MLA_A_QPE + 512 + tid, MLA_A_COLS, epoch, true, err, 2u);
            const float kp = mla_granule_sum<MLA_A_KS>(a.g_a + MLA_A_KPE + tid, MLA_A_COLS, epoch, tid < MLA_ROPE, err, 2u);
            if (tid < MLA_ROPE) s_big[1024 + tid] = kp;
        }
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) tot += s_r8[w];
        const h16 cn = (h16)(ckv * __builtin_amdgcn_rsqf(tot / (float)MLA_L + a.eps) * rw);
        {
            const float cf = (float)cn, cfn = __shfl_down(cf, 1);
            if (!(tid & 1)) mla_granule_store(a.g_q + ((MLA_H * MLA_LAT + tid) >> 1), epoch, mla_pack2(cf, cfn));
        }
        if (a.latent_out) a.latent_out[tid] = cn;
        if (a.with_pe) {
            for (int i = tid; i < MLA_H * MLA_ROPE; i += 512) {
                const int h = i >> 6, d = i & 63;
                const float rv = mla_rope(s_big + h * 64, d, a.cos, a.sin), rvn = __shfl_down(rv, 1);
                if (!(d & 1)) mla_granule_store(a.g_q + ((h * MLA_LAT + MLA_L + d) >> 1), epoch, mla_pack2(rv, rvn));
            }
        }
        if (tid < MLA_ROPE) {
            const h16 kp = a.with_pe ? (h16)mla_rope(s_big + 1024, tid, a.cos, a.sin) : (h16)0.f;
            const float kf = (float)kp, kfn = __shfl_down(kf, 1);
            if (!(tid & 1)) mla_granule_store(a.g_q + ((MLA_H * MLA_LAT + MLA_L + tid) >> 1), epoch, mla_pack2(kf, kfn));
            if (a.latent_out) a.latent_out[MLA_L + tid] = kp;
        }
        __syncthreads();
    }

    MLAF_TRACE(6);   // B published
    // ---- C ------------------------------------------------------------------------------------------------------------------
    if (has_c) {
        const float NEG = -3.0e38f;
        // q (and the new token's row): granules -> fp16 LDS.  Without the rope parts columns 512..575 are not produced.
        {
            constexpr int QP = (PE ? MLA_LAT : MLA_L) / 2;                    // fp16 pairs per head
            const bool need_new = (c_unit + 1) * a.iters * 128 >= a.n_tok;      // this unit's range holds entry n_tok - 1
            const int total = MLA_H * QP + (need_new ? MLA_LAT / 2 : 0);
            constexpr int NG = (MLA_H * QP + MLA_LAT / 2 + 511) / 512;          // granules per thread, all in flight at once
            int idx[NG];
            unsigned val[NG];
#pragma unroll
            for (int u = 0; u < NG; ++u) {
                const int i = u * 512 + tid;
                // i < 16 QP: pair (h = i / QP, k = i % QP); beyond: the latent row; -1: nothing
                idx[u] = i >= total ? -1 : (i < MLA_H * QP ? (i / QP) * (MLA_LAT / 2) + i % QP : MLA_H * (MLA_LAT / 2) + (i - MLA_H * QP));
            }
            for (unsigned spin = 0;; ++spin) {
                bool ok = true;
#pragma unroll
                for (int u = 0; u < NG; ++u) {
                    u64 g = (u64)epoch << 32;
                    if (idx[u] >= 0) g = __hip_atomic_load(a.g_q + idx[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok &= (unsigned)(g >> 32) == epoch;
                    val[u] = (unsigned)g;
                }
                if (ok) break;
                if (spin > MLA_SPIN_LIMIT) {
                    mla_flag_error(err, 3u);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
#pragma unroll
            for (int u = 0; u < NG; ++u) {
                const int i = idx[u];
                if (i < 0) continue;
                if (i < MLA_H * (MLA_LAT / 2)) reinterpret_cast<unsigned*>(s_q)[(i / (MLA_LAT / 2)) * (MLAF_QROW / 2) + i % (MLA_LAT / 2)] = val[u];
                else reinterpret_cast<unsigned*>(s_lat)[i - MLA_H * (MLA_LAT / 2)] = val[u];
            }
        }
        __syncthreads();
        MLAF_TRACE(7);   // q arrived
        h16x8 qb[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) qb[j] = *reinterpret_cast<const h16x8*>(s_q + t16 * MLAF_QROW + 32 * j + 8 * kq);
        f32x4 acc[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
        float m_run = NEG, l_run = 0.f;       // of head t16 (identical in every wavefront)
        for (int it = 0; it < a.iters; ++it) {
            const int tb = (c_unit * a.iters + it) * 128 + wave * 16;
            if (it) {
                c_tile(it);
                __syncthreads();              // the previous step's V image / P have been consumed
            }
            if (tb + t16 == a.n_tok - 1) {    // the new token stands in for the last row (kernel.cuh:470-473)
#pragma unroll
                for (int j = 0; j < NJ; ++j) av[j] = *reinterpret_cast<const h16x8*>(s_lat + 32 * j + 8 * kq);
            }
            f32x4 sc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NJ; ++j) sc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[j], qb[j], sc, 0, 0, 0);
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
            float m_new = m_run;
#pragma unroll
            for (int w = 0; w < 8; ++w) m_new = fmaxf(m_new, s_m[w * 16 + t16]);
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
            float lsum = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) lsum += s_l[w * 16 + t16];
            l_run = l_run * alpha + lsum;
            m_run = m_new;
            float al[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) al[r] = s_al[wave * 16 + 4 * kq + r];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[cb][r] *= al[r];
            // this wavefront's 32 output columns (of this workgroup's 256) over the 8 tiles of the step
#pragma unroll
            for (int tt = 0; tt < 8; ++tt) {
                const h16x4_t pt = s_p[tt * 64 + lane];
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const fp16x4_t vt = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                        (__attribute__((address_space(3))) fp16x4_t*)(s_v + tt * 8192 + (16 * c_half + 2 * wave + cb) * 256 + t16 * 4 + kq * 64));
                    acc[cb] = __builtin_amdgcn_mfma_f32_16x16x16f16(pt, __builtin_bit_cast(h16x4_t, vt), acc[cb], 0, 0, 0);
                }
            }
        }
        MLAF_TRACE(8);   // attention computed
        u64* po = a.g_po + (((size_t)c_unit * (MLA_H * MLA_L) + 256 * c_half) >> 1);
        // fp16 pairs along the columns (lanes t16, t16 + 1 hold neighbouring columns): half the granules to publish here and
        // to gather in D.  Published NORMALISED per unit, O_s / l_s: a convex combination of latent rows, |value| <= max |v|
        // whatever the number of tokens a unit sums (un-normalised sums of 1024+ tokens of a long cache could leave fp16's
        // range); D weights partial s with w_s l_s.  The rounding (2^-11 relative per partial) is of the size of the
        // reference's own fp16 roundings of the same path
        float rl[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float lh = __shfl(l_run, 4 * kq + r, 64);       // (lane t16 holds head t16's running sum)
            rl[r] = lh > 0.f ? 1.f / lh : 0.f;
        }
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float vv = acc[cb][r] * rl[r];
                const float vn = __shfl_down(vv, 1);
                if (!(t16 & 1)) mla_granule_store(po + (((4 * kq + r) * MLA_L + 32 * wave + 16 * cb + t16) >> 1), epoch, mla_pack2(vv, vn));
            }
        if (c_half == 0 && wave == 0 && kq == 0) {
            mla_granule_store(a.g_ml + (size_t)c_unit * 32 + 2 * t16, epoch, m_run);
            mla_granule_store(a.g_ml + (size_t)c_unit * 32 + 2 * t16 + 1, epoch, l_run);
        }
        __syncthreads();                      // the V image becomes role scratch again
    }

    MLAF_TRACE(9);   // C published
    // ---- D ------------------------------------------------------------------------------------------------------------------
    if (has_d) {
        const float NEG = -3.0e38f;
        const int k0 = dks * 128;
        // One round trip for everything this thread needs first: its 4 granules (fp16 pairs) of the first 32 partials (thread
        // (kp = tid % 64, q = tid / 64) takes columns 2 kp, 2 kp + 1 of partials q, q + 8, ..) and, for tid < nsplit, (m, l) of partial tid.
        const int kp = tid & 63, q = tid >> 6;
        const u64* po = a.g_po + (((size_t)dh * MLA_L + k0) >> 1) + kp;
        constexpr size_t PSTRIDE = (size_t)MLA_H * MLA_L / 2;
        float m = NEG, l = 0.f;
        h16x2 o1[4];
        for (unsigned spin = 0;; ++spin) {
            bool ok = true;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s = q + 8 * u;
                u64 g = (u64)epoch << 32;
                if (s < a.nsplit) g = __hip_atomic_load(po + (size_t)s * PSTRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok &= (unsigned)(g >> 32) == epoch;
                o1[u] = __builtin_bit_cast(h16x2, (unsigned)g);
            }
            if (tid < a.nsplit) {
                const u64 gm = __hip_atomic_load(a.g_ml + (size_t)tid * 32 + 2 * dh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const u64 gl = __hip_atomic_load(a.g_ml + (size_t)tid * 32 + 2 * dh + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok &= (unsigned)(gm >> 32) == epoch && (unsigned)(gl >> 32) == epoch;
                m = __builtin_bit_cast(float, (unsigned)gm);
                l = __builtin_bit_cast(float, (unsigned)gl);
            }
            if (ok) break;
            if (spin > MLA_SPIN_LIMIT) {
                mla_flag_error(err, 4u);
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
        // weights of the partials: w_s = exp2(m_s - M), denominator sum_s w_s l_s
        float mx = m;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        if (lane == 0) s_r8[wave] = mx;
        __syncthreads();
        MLAF_TRACE(10);  // (m, l) of all partials arrived
        mx = fmaxf(fmaxf(fmaxf(s_r8[0], s_r8[1]), fmaxf(s_r8[2], s_r8[3])), fmaxf(fmaxf(s_r8[4], s_r8[5]), fmaxf(s_r8[6], s_r8[7])));
        const float w = tid < a.nsplit ? fast_exp2(m - mx) * l : 0.f;      // (the partials arrive normalised by their l_s)
        if (tid < MLA_NSPLIT_MAX) s_w[tid] = w;
        float den = sum64(w);
        __syncthreads();                       // s_r8 read by everyone before it is rewritten; s_w visible
        if (lane == 0) s_r8[8 + wave] = den;
        {   // x[k0 + k] = sum_s w_s O_s[h][k0 + k]; the 8 thread groups meet in a fixed order below
            float (*s_xp8)[128] = s_xp;          // [8][128]: runs into s_big, which only the small-vector role (never a D unit) uses
            float v0 = 0.f, v1 = 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s = q + 8 * u;
                const float w_s = s < a.nsplit ? s_w[s] : 0.f;
                v0 = __builtin_fmaf(w_s, (float)o1[u][0], v0);
                v1 = __builtin_fmaf(w_s, (float)o1[u][1], v1);
            }
            for (int s0 = q + 32; s0 < a.nsplit; s0 += 32) {     // longer caches: the remaining partials
                h16x2 o[4];
                for (unsigned spin = 0;; ++spin) {
                    bool ok = true;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int s = s0 + 8 * u;
                        u64 g = (u64)epoch << 32;
                        if (s < a.nsplit) g = __hip_atomic_load(po + (size_t)s * PSTRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok &= (unsigned)(g >> 32) == epoch;
                        o[u] = __builtin_bit_cast(h16x2, (unsigned)g);
                    }
                    if (ok) break;
                    if (spin > MLA_SPIN_LIMIT) {
                        mla_flag_error(err, 4u);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int s = s0 + 8 * u;
                    const float w_s = s < a.nsplit ? s_w[s] : 0.f;
                    v0 = __builtin_fmaf(w_s, (float)o[u][0], v0);
                    v1 = __builtin_fmaf(w_s, (float)o[u][1], v1);
                }
            }
            s_xp8[q][2 * kp] = v0;
            s_xp8[q][2 * kp + 1] = v1;
        }
        __syncthreads();
        if (tid < 128) {
            float (*s_xp8)[128] = s_xp;
            den = ((s_r8[8] + s_r8[9]) + (s_r8[10] + s_r8[11])) + ((s_r8[12] + s_r8[13]) + (s_r8[14] + s_r8[15]));
            s_x[tid] = (((s_xp8[0][tid] + s_xp8[1][tid]) + (s_xp8[2][tid] + s_xp8[3][tid])) +
                        ((s_xp8[4][tid] + s_xp8[5][tid]) + (s_xp8[6][tid] + s_xp8[7][tid]))) / den;
        }
        __syncthreads();
        MLAF_TRACE(11);  // partials merged
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        td.fma(s_x + wave * 16, lane, acc);
        const float v = strip_reduce(acc, s_red, tid);
        // (consumers: the E units of K-slice b % 8, all on this XCD position)
        const bool d_local = __all((unsigned)(member_x >> 32) == epoch && (unsigned)member_x == xcc);
        if (tid < 64) mla_granule_store_to(a.g_d + (size_t)dks * (MLA_H * MLA_NOPE) + dh * MLA_NOPE + 64 * dc2 + tid, epoch, v, d_local);
        __syncthreads();
    }

    MLAF_TRACE(12);  // D published
    // ---- E ------------------------------------------------------------------------------------------------------------------
    {
        const float xin = mla_granule_sum<MLA_D_KS>(a.g_d + eks * 256 + tid, MLA_H * MLA_NOPE, epoch, tid < 256, err, 5u);
        if (tid < 256) s_x[tid] = xin;
        __syncthreads();
        MLAF_TRACE(13);  // o_h arrived
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        te.fma(s_x + wave * 32, lane, acc);
        const float v = strip_reduce(acc, s_red, tid);
        if (tid < 64) mla_granule_store(a.g_e + (size_t)eks * MLA_HID + 64 * estrip + tid, epoch, v);
        // the strip's last K-slice sums all 8 in slice order
        if (eks == MLA_E_KS - 1 && tid < 64) {
            const float o = mla_granule_sum<MLA_E_KS>(a.g_e + 64 * estrip + tid, MLA_HID, epoch, true, err, 6u);
            a.out[64 * estrip + tid] = (h16)o;
        }
    }
    MLAF_TRACE(14);  // done
    // Workgroup 0 has consumed stage D, which depends (through C and B) on every workgroup's stage A: everybody has
    // read state[0] long ago.
    if (b == 0 && tid == 0) {
        a.state[0] = epoch;
        a.state[2] = 0u;      // (no length arm to report: cf_workspace_last_arm documents 0 after a multi-row / MLA kernel)
    }
}

}  // namespace cf
