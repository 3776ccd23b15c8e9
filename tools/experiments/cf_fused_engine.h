// NOT PART OF THE LIBRARY (never compiled by clusterfusion_amd/build.py): the loader / consumer engine measured in round 2,
// kept as the source behind profiles/r02_experiments.md section 2 ("loader / consumer engine": 43.6 us per layer against
// 36.5 us of the shipped register-stream kernel).  To build it again: copy next to cf_fused_kernel.h, include it from cf_api.hip and
// launch k_fused_engine<NL> with EG_LDS_BYTES of dynamic LDS in place of k_fused_decode_mha<false, false, 0>; it needs the
// scalar_sweep16 helper and the g_nowait debug word quoted at the end of this file.
// cf_fused_engine.h -- the persistent [out,in] MHA decode layer as a LOADER / CONSUMER engine (gfx950).
//
// Same contract, exchanges and arithmetic as k_fused_decode_mha (cf_fused_kernel.h; reference:
// /root/reference/include/H100/llama/kernel.cuh:20-620 + include/dsm.cuh:20-171), different use of the wavefronts:
//
//   k_fused_decode_mha   every wavefront requests its own rows / tiles into registers and consumes them.  A CU admits
//                        vector-memory requests only as fast as it retires them (~25 GB/s): a wavefront that has to put
//                        16 KB of Wo rows behind 32 KB of K/V tiles sits in the issue stage for microseconds -- in the
//                        middle of phase 2, on the critical path of the exchange chain -- and every poll or granule store
//                        of the CU queues behind whatever its eight wavefronts have requested (tools/ubench/hop_scalar3.hip:
//                        2.1 us per hop between two streaming CUs, 0.55 us through the scalar path).
//   k_fused_engine       NL loader wavefronts do nothing but request: the workgroup's whole byte stream, in the order the
//                        phases need it (Wqkv rows -> this head's K/V slice -> Wo rows), straight into a ring of 8-KB LDS
//                        slots (global_load_lds_dwordx4 ... nt: no register return path), as far ahead as the ring allows.
//                        8 - NL consumer wavefronts never request bulk data: they take landed slots, do the arithmetic
//                        and run the exchanges -- small hand-offs through the scalar memory path (s_store / s_load glc),
//                        which does not share the vector queue.  The memory pipe of the CU never waits for a dependency
//                        and no dependency ever waits for the memory pipe to take a request.
//
// Ring protocol (all in LDS): slot sequence numbers s = 0.. are dealt to the loaders round-robin (loader q: s % NL == q) and
// to ring position s % NSLOT.  A loader may refill a position once `done[pos]` says the previous tenant (s - NSLOT) has been
// read; it announces landed slots through `landed[q]` (its own vmcnt, data first).  Items are handed to consumers
// round-robin per phase; a consumer takes its items in sequence order, so positions retire in (nearly) ring order.
#pragma once
#include "cf_fused_kernel.h"

namespace cf {

constexpr int EG_NSLOT = 16;                      // 8-KB ring slots (the 16 rows of Wo of a workgroup fit at once)
constexpr int EG_SLOT = 8192;
constexpr int EG_MAX_IDX = 3072;                  // page-table entries one workgroup stages
constexpr int EG_RING = 0;
constexpr int EG_QKV = EG_RING + EG_NSLOT * EG_SLOT;      // float[384]
constexpr int EG_A = EG_QKV + 384 * 4;                    // h16[4096]   attention vector
constexpr int EG_O = EG_A + 4096 * 2;                     // float[9][128]
constexpr int EG_ML = EG_O + 9 * 128 * 4;                 // float[9][2] (+pad)
constexpr int EG_REC = EG_ML + 128;                       // float[8][FUSED_REC]
constexpr int EG_CS = EG_REC + 8 * FUSED_REC * 4;         // float[256] cos|sin
constexpr int EG_CTL = EG_CS + 256 * 4;                   // int[64]: landed[2], done[16], sync counters, flags
constexpr int EG_IDX = EG_CTL + 256;                      // int[EG_MAX_IDX]
constexpr int EG_END = EG_IDX + EG_MAX_IDX * 4;
constexpr int EG_LDS_BYTES = EG_END;
static_assert(EG_END <= 160 * 1024, "LDS budget");

// control words (int index into s_ctl)
constexpr int EC_LANDED = 0;      // [2]  slots landed per loader
constexpr int EC_DONE = 4;        // [16] per ring position: sequence number + 1 of the last slot read out of it
constexpr int EC_SYNC = 24;       // [8]  consumer meeting points (monotonic counters)
constexpr int EC_FAIL = 40;       // an exchange gave up
constexpr int EC_SLOT = 41;       // new token's cache slot
constexpr int EC_IDXREADY = 42;   // page table staged

// one 1-KB piece: lane l -> 16 B at gp, LDS destination lds_addr + 16 l
__device__ __forceinline__ void dma1(const void* gp, unsigned lds_addr) {
    unsigned keep;   // (M0 is compiler-reserved: saved and restored inside the statement that uses it)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gp), "s"(lds_addr) : "memory");
}
// four consecutive 1-KB pieces of one row (global and LDS address both advance by the instruction offset)
__device__ __forceinline__ void dma4(const void* gp, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off nt\n\t"
                 "global_load_lds_dwordx4 %1, off offset:1024 nt\n\t"
                 "global_load_lds_dwordx4 %1, off offset:2048 nt\n\t"
                 "global_load_lds_dwordx4 %1, off offset:3072 nt\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gp), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ int lds_ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

// scalar-path publish of ONE granule
__device__ __forceinline__ void scalar_publish1(u64* p /* wave-uniform */, unsigned epoch, float v) {
    const u64 g = ((u64)epoch << 32) | (u64)__builtin_bit_cast(unsigned, v);
    asm volatile("s_store_dwordx2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)\n\ts_dcache_wb" :: "s"(g), "s"(p) : "memory");
}

template <int NL>
__global__ __launch_bounds__(FUSED_THREADS, 2) void k_fused_engine(FusedArgs a) {
    constexpr int NC = 8 - NL;                    // consumer wavefronts
    constexpr int KFLY = NL <= 2 ? 7 : NL == 3 ? 5 : 4;   // slots in flight per loader (8 requests each; vmcnt counts to 63; NL * KFLY <= ring)
    constexpr int HID = 4096;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_qkv = reinterpret_cast<float*>(smem + EG_QKV);
    h16* s_a = reinterpret_cast<h16*>(smem + EG_A);
    float(*s_o)[HEAD_DIM] = reinterpret_cast<float(*)[HEAD_DIM]>(smem + EG_O);
    float(*s_ml)[2] = reinterpret_cast<float(*)[2]>(smem + EG_ML);
    float(*s_rec)[FUSED_REC] = reinterpret_cast<float(*)[FUSED_REC]>(smem + EG_REC);
    float* s_cs = reinterpret_cast<float*>(smem + EG_CS);
    int* s_ctl = reinterpret_cast<int*>(smem + EG_CTL);
    int* s_idx = reinterpret_cast<int*>(smem + EG_IDX);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, g4 = lane >> 4, d0 = l16 * 8;
    const int b = blockIdx.x;
    const int h = (b & 7) * 4 + (b >> 6);         // the 8 workgroups of a head share b % 8 (one XCD: speed only)
    const int j = (b >> 3) & 7;
    const unsigned ring_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(smem + EG_RING);

    // control words start at zero: the first wavefront to run clears them, nobody reads them before the barrier
    if (tid < 64) s_ctl[tid] = 0;
    const unsigned epoch = a.state[0] + 1u;
    const bool nowait = g_nowait != 0;
    int S = a.seq_len, ent0 = 0;
    if (a.indptr) {
        ent0 = a.indptr[0];
        S = a.seq_lens ? a.seq_lens[0] : a.indptr[1] - 1 - ent0;
    }
    const h16* kc = a.kptrs ? reinterpret_cast<const h16*>(a.kptrs[a.layer_id]) : a.k_cache;
    const h16* vc = a.vptrs ? reinterpret_cast<const h16*>(a.vptrs[a.layer_id]) : a.v_cache;
    const int ps = a.page_shift, pmask = (1 << ps) - 1;
    int tps = ((S + FUSED_SPLITS - 1) / FUSED_SPLITS + 31) & ~31;   // multiple of 32 (one slot = 32 tokens of K or of V)
    tps = tps < 32 ? 32 : tps;
    const int t0 = j * tps;
    int t1 = t0 + tps;
    t1 = t1 < S ? t1 : S;
    const int ng = t1 > t0 ? (t1 - t0 + 31) >> 5 : 0;              // 32-token groups of this workgroup
    const int e0 = t0 >> ps;
    const int n_idx = (a.indptr && t1 > t0) ? ((t1 - 1) >> ps) - e0 + 1 : 0;   // (host guarantees <= EG_MAX_IDX)
    const int p_lo = a.p1_start[b], p_hi = a.p1_start[b + 1];
    const int nrows = 2 * (p_hi - p_lo);
    const int s_kv = nrows, s_wo = nrows + 2 * ng, s_end = s_wo + 16;
    const size_t kvstride = (size_t)FUSED_HEADS * HEAD_DIM;
    lds_barrier();

    if (wave < NL) {
        // =============================== LOADER ===============================================================
        const int q = wave;
        int issued = 0, announced = 0;      // slots this loader has requested / told the consumers about
        bool idx_ok = !a.indptr;
        for (int s = q; s < s_end; s += NL) {
            const int pos = s & (EG_NSLOT - 1);
            if (s >= EG_NSLOT && lds_ld(&s_ctl[EC_DONE + pos]) < s - EG_NSLOT + 1) {
                // the ring is full (the consumers are behind, or wait for an exchange): nothing to request, so everything
                // requested so far is waited for and announced -- a consumer never waits for a slot that sits in LDS
                // unannounced -- then the previous tenant of this position must have been read
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (announced < issued) { announced = issued; lds_st(&s_ctl[EC_LANDED + q], announced); }
                for (unsigned spin = 0; lds_ld(&s_ctl[EC_DONE + pos]) < s - EG_NSLOT + 1 && spin < 4000000u; ++spin) __builtin_amdgcn_s_sleep(1);
            }
            const unsigned dst = ring_addr + pos * EG_SLOT;
            if (s < s_kv) {                        // a row of Wqkv
                const h16* p = a.Wqkv + (size_t)(2 * p_lo + s) * HID + lane * 8;
                dma4(p, dst);
                dma4(p + 2048, dst + 4096);
            } else if (s < s_wo) {                 // 32 tokens of K (even) or V (odd): 4 tokens per request
                if (!idx_ok) {   // first K/V slot of this loader: the page table must be staged
                    for (unsigned spin = 0; lds_ld(&s_ctl[EC_IDXREADY]) == 0 && spin < 4000000u; ++spin) __builtin_amdgcn_s_sleep(1);
                    idx_ok = true;
                }
                const int kvi = s - s_kv;
                const h16* base = ((kvi & 1) ? vc : kc) + h * HEAD_DIM + d0;
                const int tg = t0 + (kvi >> 1) * 32 + g4;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    int tk = tg + 4 * i;
                    tk = tk < t1 ? tk : t1 - 1;
                    size_t row = (size_t)tk;
                    if (a.indptr) row = ((size_t)s_idx[(tk >> ps) - e0] << ps) + (size_t)(tk & pmask);
                    dma1(base + row * kvstride, dst + i * 1024);
                }
            } else {                               // a row of Wo
                const h16* p = a.Wo + (size_t)(16 * b + (s - s_wo)) * HID + lane * 8;
                dma4(p, dst);
                dma4(p + 2048, dst + 4096);
            }
            ++issued;
            if (issued - announced > KFLY) {       // at most KFLY slots outstanding: the older ones have landed
                if constexpr (KFLY == 7) asm volatile("s_waitcnt vmcnt(56)" ::: "memory");
                else if constexpr (KFLY == 5) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
                announced = issued - KFLY;
                lds_st(&s_ctl[EC_LANDED + q], announced);
            }
        }
        // drain
#pragma unroll
        for (int k = KFLY - 1; k >= 0; --k) {
            switch (k) {
                case 6: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
                case 5: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
                case 4: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
                case 1: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            }
            if (issued - k > announced) { announced = issued - k; lds_st(&s_ctl[EC_LANDED + q], announced); }
        }
        return;
    }

    // =================================== CONSUMERS =============================================================
    const int cw = wave - NL;                     // consumer index 0 .. NC-1
    const int ct = tid - NL * 64;                 // consumer thread index 0 .. NC*64-1
    auto wait_landed = [&](int s) {               // slot s is in LDS
        const int q = s % NL, need = s / NL + 1;
        for (unsigned spin = 0; lds_ld(&s_ctl[EC_LANDED + q]) < need && spin < 4000000u; ++spin) __builtin_amdgcn_s_sleep(1);
    };
    auto release = [&](int s) {                   // this wavefront has read slot s
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) lds_st(&s_ctl[EC_DONE + (s & (EG_NSLOT - 1))], s + 1);
    };
    int sync_gen = 0;
    auto csync = [&](int k) {                     // all consumer wavefronts meet (LDS counter; the loaders are elsewhere)
        ++sync_gen;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(&s_ctl[EC_SYNC + k], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        for (unsigned spin = 0; lds_ld(&s_ctl[EC_SYNC + k]) < NC && spin < 4000000u; ++spin) __builtin_amdgcn_s_sleep(1);
    };
    if (a.trace && ct == 0) {
        unsigned hw, xc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xc));
        a.trace[(size_t)b * 16 + 13] = ((u64)xc << 32) | hw;
    }
#define EG_TRACE(slot)                                                                              \
    do {                                                                                            \
        if (a.trace && ct == 0) a.trace[(size_t)b * 16 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
    EG_TRACE(0);
    const unsigned xcc = my_xcc_id();
    if (ct == 0) granule_store(a.g_xcc + b, epoch, __builtin_bit_cast(float, xcc));   // where this workgroup runs

    // ---- RMSNorm: every consumer wavefront normalises the whole row for itself (lane l, i: elements (64 i + l) 8 ..) ----
    float xn[8][8];
    load_norm_x<8>(a.na, 0, lane, xn);
    // ---- page-table slice, new-token slot, RoPE row -> LDS ------------------------------------------------------
    for (int i = ct; i < n_idx; i += NC * 64) s_idx[i] = a.indices[ent0 + e0 + i];
    {
        const int64_t roff = a.positions ? a.positions[0] * a.rope_stride : 0;
        const int n_ang = a.rope_style == 0 ? HEAD_DIM / 2 : HEAD_DIM;
        if (ct < n_ang) s_cs[ct] = a.cos[roff + ct];
        else if (ct >= 128 && ct < 128 + n_ang) s_cs[ct] = a.sin[roff + ct - 128];
        if (a.indptr && ct == 0) s_ctl[EC_SLOT] = a.indices[ent0 + (S >> ps)];
    }
    csync(0);
    if (ct == 0) lds_st(&s_ctl[EC_IDXREADY], 1);

    // ---- phase 1: rows i = cw, cw + NC, .. of this workgroup's share; row r = m*4096 + head*128 + i ---------------
    for (int i = cw; i < nrows; i += NC) {
        wait_landed(i);
        const h16x8* src = reinterpret_cast<const h16x8*>(smem + EG_RING + (i & (EG_NSLOT - 1)) * EG_SLOT) + lane;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc = dot8(src[k * 64], xn[k], acc);
        release(i);
        acc = sum64_lane63(acc);
        const float res = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, acc), 63));
        const int r = 2 * p_lo + i;
        scalar_publish1(a.g_qkv + (size_t)((r & 4095) >> 7) * 384 + (r >> 12) * 128 + (r & 127), epoch, res);
    }
    EG_TRACE(1);   // phase 1 done

    // ---- X1: q | k | v of this head through the scalar path: 24 chunks of 16 granules over the consumers ----------
    {
        bool ok = true;
        const int nch = j == 0 ? 24 : 8;          // (k | v of the new token: split 0 only)
        for (int c = cw; c < nch; c += NC) {
            const u64* gq = a.g_qkv + (size_t)h * 384 + 16 * c;
            bool got = scalar_sweep16(gq, epoch, s_qkv + 16 * c, lane, nowait);
            if (!got) {
                if (lane == 0) atomicAdd(a.state + 2, 1u);
                got = sweep_granules<1>(gq, 16, epoch, s_qkv + 16 * c, lane, a.state + 1, 1u);
            }
            ok &= got;
        }
        if (!ok && lane == 0) s_ctl[EC_FAIL] = 1;
    }
    csync(1);
    EG_TRACE(2);   // X1 resolved
    const u64 lead_x = __hip_atomic_load(a.g_xcc + (b & ~0x38), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // ---- RoPE(q), scaled for base-2 softmax ---------------------------------------------------------------------
    const float qscale = 1.44269504088896340736f * 0.08838834764831845f;
    float qf[8];
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
    rope_lds(s_qkv, qf);
    h16x8 qh;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        qf[e] *= qscale;
        qh[e] = (h16)qf[e];
    }

    // ---- phase 2: 32-token groups g = cw, cw + NC, ..: K in slot s_kv + 2g, V in the next; lane group g4 takes tokens 4u + g4 ----
    float m = NEG_BIG, l = 0.f, o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int g = cw; g < ng; g += NC) {
        const int sk = s_kv + 2 * g;
        wait_landed(sk);
        const h16x8* kp = reinterpret_cast<const h16x8*>(smem + EG_RING + (sk & (EG_NSLOT - 1)) * EG_SLOT) + lane;
        float sc[8];
        float mx = NEG_BIG;
        const int tb = t0 + 32 * g + g4;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float sv = sum16(dot8h(kp[u * 64], qh, 0.f));
            sc[u] = (tb + 4 * u) < t1 ? sv : NEG_BIG;
            mx = fmaxf(mx, sc[u]);
        }
        release(sk);
        const float mnew = fmaxf(m, mx);
        const float alpha = fast_exp2(m - mnew);
        float psum = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            sc[u] = (tb + 4 * u) < t1 ? fast_exp2(sc[u] - mnew) : 0.f;
            psum += sc[u];
        }
        l = l * alpha + psum;
        wait_landed(sk + 1);
        const h16x8* vp = reinterpret_cast<const h16x8*>(smem + EG_RING + ((sk + 1) & (EG_NSLOT - 1)) * EG_SLOT) + lane;
        h16x8 vv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) vv[u] = vp[u * 64];
        release(sk + 1);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float acc = o[e] * alpha;
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_fmaf((float)vv[u][e], sc[u], acc);
            o[e] = acc;
        }
        m = mnew;
    }
    {   // the 4 lane groups of the wavefront hold the same dims for different tokens: merge in registers
        const float mw = xmax32(xmax16(m));
        const float scl = fast_exp2(m - mw);
        l = xsum32(xsum16(l * scl));
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = xsum32(xsum16(o[e] * scl));
        m = mw;
    }
    if (lane < 16) {
#pragma unroll
        for (int e = 0; e < 8; ++e) s_o[cw][d0 + e] = o[e];
        if (lane == 0) { s_ml[cw][0] = m; s_ml[cw][1] = l; }
    }
    // the new token (attended from registers, kernel.cuh:444-477) + k/v export: split 0, first consumer, lanes 0..15
    if (j == 0 && cw == 0 && g4 == 0) {
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
            const size_t slot = ((size_t)s_ctl[EC_SLOT] << ps) + (size_t)(S & pmask);
            st_h8(const_cast<h16*>(kc) + slot * kvstride + ooff, k16);
            st_h8(const_cast<h16*>(vc) + slot * kvstride + ooff, v16);
        }
        float sn = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) sn = __builtin_fmaf(qf[e], kf[e], sn);
        sn = sum16(sn);
#pragma unroll
        for (int e = 0; e < 8; ++e) s_o[NC][d0 + e] = vf[e];
        if (l16 == 0) { s_ml[NC][0] = sn; s_ml[NC][1] = 1.f; }
    }
    csync(2);
    EG_TRACE(3);   // phase 2 done

    // ---- X2: one record per workgroup -> the head's leader (split 0) ------------------------------------------------
    const bool rec_local = (unsigned)(lead_x >> 32) == epoch && (unsigned)lead_x == xcc;
    if (ct < HEAD_DIM + 2) {
        const int nst = j == 0 ? NC + 1 : NC;
        float M = NEG_BIG;
#pragma unroll
        for (int i = 0; i < NC + 1; ++i) M = fmaxf(M, i < nst ? s_ml[i][0] : NEG_BIG);
        float val;
        if (ct < HEAD_DIM) {
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < NC + 1; ++i)
                if (i < nst) acc = __builtin_fmaf(fast_exp2(s_ml[i][0] - M), s_o[i][ct], acc);
            val = acc;
        } else if (ct == HEAD_DIM) {
            val = M;
        } else {
            float L = 0.f;
#pragma unroll
            for (int i = 0; i < NC + 1; ++i)
                if (i < nst) L = __builtin_fmaf(fast_exp2(s_ml[i][0] - M), s_ml[i][1], L);
            val = L;
        }
        granule_store_to(a.g_rec + ((size_t)h * FUSED_SPLITS + j) * FUSED_REC_G + ct, epoch, val, rec_local);
    }
    if (j == 0) {   // leader: the 8 records over the consumers, then the head's softmax merge
        bool ok = true;
        for (int r = cw; r < FUSED_SPLITS; r += NC)
            ok &= sweep_granules<3>(a.g_rec + ((size_t)h * FUSED_SPLITS + r) * FUSED_REC_G, HEAD_DIM + 2, epoch, s_rec[r], lane,
                                    a.state + 1, 2u);
        if (!ok && lane == 0) s_ctl[EC_FAIL] = 1;
        csync(3);
        if (ct < HEAD_DIM) {
            float M = NEG_BIG;
#pragma unroll
            for (int w = 0; w < FUSED_SPLITS; ++w) M = fmaxf(M, s_rec[w][HEAD_DIM]);
            float acc = 0.f, L = 0.f;
#pragma unroll
            for (int w = 0; w < FUSED_SPLITS; ++w) {
                const float wt = fast_exp2(s_rec[w][HEAD_DIM] - M);
                acc = __builtin_fmaf(wt, s_rec[w][ct], acc);
                L = __builtin_fmaf(wt, s_rec[w][HEAD_DIM + 1], L);
            }
            granule_store(a.g_attn + (size_t)h * HEAD_DIM + ct, epoch, acc / L);
        }
    }
    EG_TRACE(4);   // record published (leader: head merged + published)

    // ---- X3: the full attention output (fp16 into LDS, as the reference rounds it, kernel.cuh:553-559): an even share of the
    //      4096 granules per consumer ---------------------------------------------------------------------------------
    {
        constexpr int G3 = ((4096 + NC - 1) / NC + 63) & ~63;      // granules per consumer (640 for 7, 704 for 6)
        wait_hint(a.g_attn + HEAD_DIM - 1, 32, HEAD_DIM, epoch, lane);   // last element of every head
        const int g0 = cw * G3;
        const int cnt = 4096 - g0 < G3 ? 4096 - g0 : G3;
        const bool ok = sweep_granules<G3 / 64>(a.g_attn + g0, cnt, epoch, s_a + g0, lane, a.state + 1, 3u);
        if (!ok && lane == 0) s_ctl[EC_FAIL] = 1;
    }
    csync(4);
    EG_TRACE(5);   // X3 resolved
    if (s_ctl[EC_FAIL]) return;     // (the loaders finish on their own: every slot they wait for has been released)

    // ---- phase 3: rows 16 b + i, i = cw, cw + NC, .. (slot s_wo + i) ---------------------------------------------------
    {
        h16x8 av[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) av[k] = *reinterpret_cast<const h16x8*>(s_a + (k * WAVE + lane) * 8);
        for (int i = cw; i < 16; i += NC) {
            wait_landed(s_wo + i);
            const h16x8* src = reinterpret_cast<const h16x8*>(smem + EG_RING + ((s_wo + i) & (EG_NSLOT - 1)) * EG_SLOT) + lane;
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) acc = dot8h(src[k * 64], av[k], acc);
            acc = sum64_lane63(acc);
            if (lane == 63) a.out[16 * b + i] = (h16)acc;
        }
    }
    // residual_out may alias residual: every workgroup read residual before X3 could complete
    if (a.residual_out && ct < 16) {
        const int i = 16 * b + ct;
        a.residual_out[i] = (h16)((float)a.na.x[i] + (float)a.na.residual[i]);
    }
    if (b == 0 && ct == 0) a.state[0] = epoch;
    EG_TRACE(6);
#undef EG_TRACE
}

}  // namespace cf

/* ---- helpers the engine expects in cf_fused_kernel.h (round-2 experiment versions) ------------------------------------
__device__ int g_nowait = 0;      // debug: exchanges do not wait
constexpr unsigned FUSED_SCALAR_TRIES = 8192u;
typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ bool scalar_sweep16(const u64* g, unsigned epoch, float* dst, int lane, bool nowait) {
    for (unsigned spin = 0; spin < FUSED_SCALAR_TRIES; ++spin) {
        u32x16 va, vb;
        asm volatile("s_load_dwordx16 %0, %2, 0x0 glc\n\ts_load_dwordx16 %1, %2, 0x40 glc\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(va), "=&s"(vb) : "s"(g) : "memory");
        bool ok = true;
        for (int i = 0; i < 8; ++i) ok &= va[2 * i + 1] == epoch && vb[2 * i + 1] == epoch;
        if (ok || nowait) {
            unsigned v = 0;
            for (int i = 0; i < 8; ++i) { v = lane == i ? va[2 * i] : v; v = lane == 8 + i ? vb[2 * i] : v; }
            if (lane < 16) dst[lane] = __builtin_bit_cast(float, v);
            return true;
        }
        __builtin_amdgcn_s_sleep(2);
    }
    return false;
}
*/
