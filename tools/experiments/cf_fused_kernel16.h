// NOT PART OF THE LIBRARY (never compiled by clusterfusion_amd/build.py): the 16-wavefront variant of k_fused_decode_mha measured in
// round 2 (profiles/r02_experiments.md section 2): 40.3-41.7 us per layer against 36.5 -- phase 1 done at 15.1 us instead of 20.0, but the
// K/V tiles, requested in one burst behind it, arrive 13.8 us later (X1 at 28.9 instead of 25.4).  This is the LAST version tried (Wo row
// requested before X1 + scalar X1, K/V through buffer resources): it does not fit 128 VGPRs -- the measured, spill-free one requested
// the Wo row after the scores (`go.load` behind CF_TRACE(8)) and gathered X1 with sweep_granules<6> on wavefront 0.
// cf_fused_kernel16.h -- k_fused_decode_mha with SIXTEEN wavefronts per workgroup (gfx950).
//
// Same decomposition, exchanges and arithmetic as k_fused_decode_mha<false, false> (cf_fused_kernel.h; reference:
// /root/reference/include/H100/llama/kernel.cuh:20-620 + include/dsm.cuh:20-171); what changes is how a CU's request pipe is
// fed.  That pipe -- not HBM -- paces a streaming decode kernel (~25 GB/s per CU), and it runs a few per cent
// faster when sixteen wavefronts keep one 8-KB row each in flight than when eight keep two row pairs
// (tools/ubench/dma_bw.hip: 6.47 vs 6.20 TB/s on rows; skel_bw.hip: 32.1 vs 33.1 us for this kernel's byte stream).
// Sixteen wavefronts share the 512-register file four to a SIMD: 128 VGPRs each.  To fit:
//   * the normalised activations live in LDS as fp16 (the reference rounds them there too, kernel.cuh:133-138) and are read
//     per 1-KB chunk, not held in 64 registers;
//   * phase 1 streams single rows (two in flight per wavefront), phase 2 one 512-token tile per workgroup (8 K + 8 V
//     requests per wavefront: S <= 4096, longer sequences take the 8-wavefront kernel), phase 3 one Wo row per wavefront.
#pragma once
#include "cf_fused_kernel.h"

namespace cf {

constexpr int F16_THREADS = 1024, F16_WAVES = 16;
constexpr int F16_QKV = 0;                                  // float[384]
constexpr int F16_A = F16_QKV + 384 * 4;                    // h16[4096]: xn (phase 1), attention vector (phase 3)
constexpr int F16_O = F16_A + 4096 * 2;                     // float[17][128]: 16 wavefront states + the new token
constexpr int F16_ML = F16_O + 17 * 128 * 4;                // float[17][2] (+pad)
constexpr int F16_REC = F16_ML + 144;                       // float[8][FUSED_REC]
constexpr int F16_IDX = F16_REC + 8 * FUSED_REC * 4;        // int[1024]: page-table slice (S <= 4096: <= 512 entries)
constexpr int F16_CS = F16_IDX + 1024 * 4;                  // float[256]
constexpr int F16_CTL = F16_CS + 256 * 4;                   // int[64]
constexpr int F16_END = F16_CTL + 256;
constexpr int F16_LDS_BYTES = F16_END > 84 * 1024 ? F16_END : 84 * 1024;   // > half a CU's LDS: one workgroup per CU

// 16 consecutive granules through the SCALAR memory path (two s_load_dwordx16 ... glc): SMEM does not queue behind the CU's
// vector requests (tools/ubench/hop_scalar*.hip), so X1 resolves when q exists, not when the tiles and the Wo row requested
// before it have arrived.  Bounded: the caller falls back to the vector sweep.
typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ bool scalar_sweep8(const u64* g /* wave-uniform */, unsigned epoch, float* dst /* 8 floats */, int lane) {
    for (unsigned spin = 0; spin < 8192u; ++spin) {
        u32x16 va;
        asm volatile("s_load_dwordx16 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=&s"(va) : "s"(g) : "memory");
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 8; ++i) ok &= va[2 * i + 1] == epoch;
        if (ok) {
            unsigned v = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) v = lane == i ? va[2 * i] : v;
            if (lane < 8) dst[lane] = __builtin_bit_cast(float, v);
            return true;
        }
        __builtin_amdgcn_s_sleep(2);
    }
    return false;
}
__device__ __forceinline__ bool scalar_sweep16(const u64* g, unsigned epoch, float* dst, int lane) {
    return scalar_sweep8(g, epoch, dst, lane) && scalar_sweep8(g + 8, epoch, dst + 8, lane);
}

__global__ __launch_bounds__(F16_THREADS) void k_fused_decode_mha16(FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_qkv = reinterpret_cast<float*>(smem + F16_QKV);
    h16* s_a = reinterpret_cast<h16*>(smem + F16_A);
    float(*s_o)[HEAD_DIM] = reinterpret_cast<float(*)[HEAD_DIM]>(smem + F16_O);
    float(*s_ml)[2] = reinterpret_cast<float(*)[2]>(smem + F16_ML);
    float(*s_rec)[FUSED_REC] = reinterpret_cast<float(*)[FUSED_REC]>(smem + F16_REC);
    int* s_idx = reinterpret_cast<int*>(smem + F16_IDX);
    float* s_cs = reinterpret_cast<float*>(smem + F16_CS);
    int* s_ctl = reinterpret_cast<int*>(smem + F16_CTL);

    constexpr int HID = 4096;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, gid = wave * 4 + (lane >> 4), d0 = l16 * 8;      // 64 lane groups
    const int b = blockIdx.x;
    if (a.trace && tid == 0) {
        unsigned hw, xc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xc));
        a.trace[(size_t)blockIdx.x * 16 + 13] = ((u64)xc << 32) | hw;
    }
    const int h = (b & 7) * 4 + (b >> 6);         // the 8 workgroups of a head share b % 8 (one XCD: speed only)
    const int j = (b >> 3) & 7;
    CF_TRACE(0);

    // ---- small first-level loads first: thread t owns elements [4t, 4t + 4) of x / residual / rms_w ----
    typedef h16 h16x4 __attribute__((ext_vector_type(4)));
    const h16* rp = a.na.residual ? a.na.residual : a.na.x;
    const float rs = a.na.residual ? 1.f : 0.f;
    const h16x4 xv = *(const CF_GLOBAL h16x4*)(a.na.x + tid * 4), rv = *(const CF_GLOBAL h16x4*)(rp + tid * 4),
                wv4 = *(const CF_GLOBAL h16x4*)(a.na.rms_w + tid * 4);
    const unsigned epoch = a.state[0] + 1u;
    const unsigned xcc = my_xcc_id();
    if (tid == 0) granule_store(a.g_xcc + b, epoch, __builtin_bit_cast(float, xcc));
    int S = a.seq_len, ent0 = 0;
    if (a.indptr) {
        ent0 = a.indptr[0];
        S = a.seq_lens ? a.seq_lens[0] : a.indptr[1] - 1 - ent0;
    }
    const int64_t roff = a.positions ? a.positions[0] * a.rope_stride : 0;
    const h16* kc = a.kptrs ? reinterpret_cast<const h16*>(a.kptrs[a.layer_id]) : a.k_cache;
    const h16* vc = a.vptrs ? reinterpret_cast<const h16*>(a.vptrs[a.layer_id]) : a.v_cache;

    // ---- weight stream of phase 1: rows 2 p_lo .. 2 p_hi - 1 of Wqkv; wavefront w takes rows w, w + 16, .. : up to four row
    //      slots, a slot beyond the share gets an offset beyond the buffer (the load issues, touches no memory, returns zeros)
    const int r_lo = 2 * a.p1_start[b], r_hi = 2 * a.p1_start[b + 1];
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16*>(a.Wqkv), 0, 3 * HID * HID * 2, 0x00020000);
    RowGroup<8, 1> r0, r1;
    auto p1_load = [&](RowGroup<8, 1>& t, int slot) {
        const int row = r_lo + wave + F16_WAVES * slot;
        const int voff = row < r_hi ? row * (HID * 2) + lane * 16 : 0x40000000;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            t.w[0][i] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, voff + i * (WAVE * 16), 0, 2 /* nt */));
    };
    p1_load(r0, 0);
    p1_load(r1, 1);

    // ---- RMSNorm once per workgroup -> fp16 activations in LDS -------------------------------------
    float hx[4];
    {
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            hx[e] = __builtin_fmaf(rs, (float)rv[e], (float)xv[e]);
            ss = __builtin_fmaf(hx[e], hx[e], ss);
        }
        ss = sum64_lane63(ss);
        if (lane == 63) s_rec[0][wave] = ss;     // s_rec is free until X2
    }
    // second-level loads: page-table slice, new-token slot, RoPE row
    const int ps = a.page_shift, pmask = (1 << ps) - 1;
    int tps = ((S + FUSED_SPLITS - 1) / FUSED_SPLITS + 63) & ~63;   // multiple of 64 (one token per lane-group row)
    tps = tps < 64 ? 64 : tps;
    const int t0 = j * tps;
    int t1 = t0 + tps;
    t1 = t1 < S ? t1 : S;
    const int e0 = t0 >> ps;
    int n_idx = 0;
    if (a.indptr && t1 > t0) {
        n_idx = ((t1 - 1) >> ps) - e0 + 1;
        if (n_idx > 1024) {   // (host guard: S <= 4096)
            if (tid == 0) flag_exchange_error(a.state + 1, 4u);
            n_idx = 1024;
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
    {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < F16_WAVES; ++w) tot += s_rec[0][w];
        const float rcp = __builtin_amdgcn_rsqf(tot / (float)HID + a.na.eps);
        h16x4 xo;
#pragma unroll
        for (int e = 0; e < 4; ++e) xo[e] = (h16)(hx[e] * rcp * (float)wv4[e]);
        *reinterpret_cast<h16x4*>(s_a + tid * 4) = xo;
        if (tid < n_idx) s_idx[tid] = idx_reg;
        if (tid < 256) s_cs[tid] = cs_reg;
        if (tid == 0) s_ctl[20] = slot_reg;
        lds_barrier();
    }

    // row r = m*4096 + head*128 + i -> the granule of (head, q|k|v, index); activations from LDS, one 1-KB chunk at a time
    auto p1_dot_publish = [&](const RowGroup<8, 1>& t, int slot) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc = dot8h(t.w[0][i], *reinterpret_cast<const h16x8*>(s_a + (i * WAVE + lane) * 8), acc);
        acc = sum64_lane63(acc);
        const int r = r_lo + wave + F16_WAVES * slot;
        if (lane == 63 && r < r_hi)
            granule_store(a.g_qkv + (size_t)((r & 4095) >> 7) * 384 + (r >> 12) * 128 + (r & 127), epoch, acc);
    };

    // ---- the K/V tile of phase 2: token t0 + gid + 64 u, u < 8 (the workgroup's whole slice), requested BEFORE q exists;
    //      unconditional requests (a slice that is empty reads one dummy line) ----------------------------------------------
    const size_t kvstride = (size_t)FUSED_HEADS * HEAD_DIM;
    const h16* kbase = kc + h * HEAD_DIM + d0;
    const h16* vbase = vc + h * HEAD_DIM + d0;
    const h16* dummy = a.na.rms_w + d0;
    const bool live = t0 < t1;
    int krow[8];      // cache row (slot) of this lane group's token u; 32-bit to spare registers (128 per wavefront)
    auto tile_rows = [&]() {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            int tk = t0 + u * 64 + gid;
            tk = tk < t1 ? tk : t1 - 1;
            tk = tk > t0 ? tk : t0;
            int row = tk;
            if (a.indptr) {
                int ei = (tk >> ps) - e0;
                ei = ei < 1023 ? ei : 1023;
                row = (s_idx[ei] << ps) + (tk & pmask);
            }
            krow[u] = live ? row : 0;
        }
    };
    h16x8 kt[8], vt[8];
    p1_dot_publish(r0, 0);
    CF_TRACE(14);
    p1_load(r0, 2);
    p1_dot_publish(r1, 1);
    CF_TRACE(15);
    p1_load(r1, 3);
    tile_rows();
    p1_dot_publish(r0, 2);
    // (K/V through buffer resources with 32-bit offsets: 24 64-bit addresses in flight would not fit 128 registers.
    //  EXPERIMENT: caches up to 4 GB)
    const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16*>(live ? kc : a.na.rms_w), 0, 0xFFFFFFFFu, 0x00020000);
    const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16*>(live ? vc : a.na.rms_w), 0, 0xFFFFFFFFu, 0x00020000);
    const unsigned kv_col = live ? (unsigned)(h * HEAD_DIM + d0) * 2u : (unsigned)d0 * 2u;
    const unsigned kv_rowb = live ? (unsigned)kvstride * 2u : 0u;
    {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            kt[u] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(k_rsrc, (unsigned)krow[u] * kv_rowb + kv_col, 0, 2));
    }
    p1_dot_publish(r1, 3);
    {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            vt[u] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(v_rsrc, (unsigned)krow[u] * kv_rowb + kv_col, 0, 2));
    }
    // the Wo row of this wavefront goes out right behind the tiles (the registers of the two row buffers are free)
    RowGroup<8, 1> go;
    {
        const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<h16*>(a.Wo), 0, HID * HID * 2, 0x00020000);
        const int voff = (16 * b + wave) * (HID * 2) + lane * 16;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            go.w[0][i] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(o_rsrc, voff + i * (WAVE * 16), 0, 2));
    }
    CF_TRACE(1);   // phase 1 done (all rows published, everything requested)

    // ---- X1: gather q (split 0: q|k|v) of this head through the scalar path, 16 granules per wavefront and chunk -------
    {
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        const int nch = j == 0 ? 24 : 8;
        bool ok = true;
        for (int c = wv; c < nch; c += F16_WAVES) {
            const u64* gq = a.g_qkv + (size_t)h * 384 + 16 * c;
            bool got = scalar_sweep16(gq, epoch, s_qkv + 16 * c, lane);
            if (!got) {
                if (lane == 0) atomicAdd(a.state + 2, 1u);
                got = sweep_granules<1>(gq, 16, epoch, s_qkv + 16 * c, lane, a.state + 1, 1u);
            }
            ok &= got;
        }
        if (lane == 0) s_ctl[32 + wave] = ok;
    }
    lds_barrier();
    {
        bool all_ok = true;
        for (int w = 0; w < F16_WAVES; ++w) all_ok &= s_ctl[32 + w] != 0;
        if (!all_ok) CF_FAIL_RETURN();
    }
    CF_TRACE(2);   // X1 resolved
    const u64 lead_x = __hip_atomic_load(a.g_xcc + (b & ~0x38), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // ---- RoPE(q), scaled for base-2 softmax ----------------------------------------------------------
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
    CF_TRACE(7);

    // ---- phase 2: scores of the 8 token rows, then the Wo row goes out into the K registers, then the V accumulation ----
    float m, l, o[8];
    {
        float s[8];
        float mx = NEG_BIG;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool valid = (t0 + u * 64 + gid) < t1;
            const float sv = sum16(dot8h(kt[u], qh, 0.f));
            s[u] = valid ? sv : NEG_BIG;
            mx = fmaxf(mx, s[u]);
        }
        CF_TRACE(8);   // K consumed
        m = mx;
        float psum = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s[u] = (t0 + u * 64 + gid) < t1 ? fast_exp2(s[u] - mx) : 0.f;
            psum += s[u];
        }
        l = psum;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float acc = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_fmaf((float)vt[u][e], s[u], acc);
            o[e] = acc;
        }
        CF_TRACE(10);  // V consumed
    }
    {   // merge the 4 lane groups of this wavefront in registers
        const float mw = xmax32(xmax16(m));
        const float sc = fast_exp2(m - mw);
        l = xsum32(xsum16(l * sc));
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = xsum32(xsum16(o[e] * sc));
        m = mw;
    }
    CF_TRACE(11);
    if (lane < 16) {
#pragma unroll
        for (int e = 0; e < 8; ++e) s_o[wave][d0 + e] = o[e];
        if (lane == 0) { s_ml[wave][0] = m; s_ml[wave][1] = l; }
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
        const size_t ooff = (size_t)h * HEAD_DIM + d0;
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
        for (int e = 0; e < 8; ++e) s_o[F16_WAVES][d0 + e] = vf[e];
        if (l16 == 0) { s_ml[F16_WAVES][0] = sn; s_ml[F16_WAVES][1] = 1.f; }
    }
    CF_TRACE(12);
    lds_barrier();
    CF_TRACE(3);   // phase 2 done

    // ---- X2: one record per workgroup -> the head's leader ---------------------------------------
    const bool rec_local = (unsigned)(lead_x >> 32) == epoch && (unsigned)lead_x == xcc;
    if (tid < HEAD_DIM + 2) {
        const int nst = j == 0 ? F16_WAVES + 1 : F16_WAVES;
        float M = NEG_BIG;
#pragma unroll
        for (int i = 0; i < F16_WAVES + 1; ++i) M = fmaxf(M, i < nst ? s_ml[i][0] : NEG_BIG);
        float val;
        if (tid < HEAD_DIM) {
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < F16_WAVES + 1; ++i)
                if (i < nst) acc = __builtin_fmaf(fast_exp2(s_ml[i][0] - M), s_o[i][tid], acc);
            val = acc;
        } else if (tid == HEAD_DIM) {
            val = M;
        } else {
            float L = 0.f;
#pragma unroll
            for (int i = 0; i < F16_WAVES + 1; ++i)
                if (i < nst) L = __builtin_fmaf(fast_exp2(s_ml[i][0] - M), s_ml[i][1], L);
            val = L;
        }
        granule_store_to(a.g_rec + ((size_t)h * FUSED_SPLITS + j) * FUSED_REC_G + tid, epoch, val, rec_local);
    }
    if (j == 0) {   // leader: wavefront w < 8 gathers record w, then the head's softmax merge
        if (wave < 8) {
            const bool ok = sweep_granules<3>(a.g_rec + ((size_t)h * FUSED_SPLITS + wave) * FUSED_REC_G, HEAD_DIM + 2, epoch,
                                              s_rec[wave], lane, a.state + 1, 2u);
            if (lane == 0) s_ctl[1 + wave] = ok;
        }
        lds_barrier();
        bool all_ok = true;
        for (int w = 0; w < 8; ++w) all_ok &= s_ctl[1 + w] != 0;
        if (!all_ok) CF_FAIL_RETURN();
        if (tid < HEAD_DIM) {
            float M = NEG_BIG;
#pragma unroll
            for (int w = 0; w < FUSED_SPLITS; ++w) M = fmaxf(M, s_rec[w][HEAD_DIM]);
            float acc = 0.f, L = 0.f;
#pragma unroll
            for (int w = 0; w < FUSED_SPLITS; ++w) {
                const float wt = fast_exp2(s_rec[w][HEAD_DIM] - M);
                acc = __builtin_fmaf(wt, s_rec[w][tid], acc);
                L = __builtin_fmaf(wt, s_rec[w][HEAD_DIM + 1], L);
            }
            granule_store(a.g_attn + (size_t)h * HEAD_DIM + tid, epoch, acc / L);
        }
    }
    CF_TRACE(4);   // record published (leader: head merged + published)

    // ---- X3: every workgroup gathers the full attention output: 256 granules (2 heads) per wavefront ----
    {
        wait_hint(a.g_attn + wave * 256 + HEAD_DIM - 1, 2, HEAD_DIM, epoch, lane);
        const bool ok = sweep_granules<4>(a.g_attn + wave * 256, 256, epoch, s_a + wave * 256, lane, a.state + 1, 3u);
        if (lane == 0) s_ctl[32 + wave] = ok;
    }
    lds_barrier();
    {
        bool all_ok = true;
        for (int w = 0; w < F16_WAVES; ++w) all_ok &= s_ctl[32 + w] != 0;
        if (!all_ok) CF_FAIL_RETURN();
    }
    CF_TRACE(5);   // X3 resolved
    // ---- phase 3: one row of Wo per wavefront -------------------------------------------------------
    {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc = dot8h(go.w[0][i], *reinterpret_cast<const h16x8*>(s_a + (i * WAVE + lane) * 8), acc);
        acc = sum64_lane63(acc);
        if (lane == 63) a.out[16 * b + wave] = (h16)acc;
    }
    if (a.residual_out && tid < 16) {
        const int i = 16 * b + tid;
        a.residual_out[i] = (h16)((float)a.na.x[i] + (float)a.na.residual[i]);
    }
    if (b == 0 && tid == 0) a.state[0] = epoch;
    CF_TRACE(6);
}

}  // namespace cf
