// cf_fused_kernel.h -- ONE persistent launch per decoder layer (gfx950), the CDNA4 answer to the
// reference's thread-block-cluster kernel (/root/reference/include/H100/llama/kernel.cuh:20-620 +
// include/dsm.cuh:20-171).
//
// Hopper: 4 CTAs per head form a cluster and all-reduce through distributed shared memory.
// CDNA4 has no clusters and per-XCD L2s that are not coherent, so the collective is rebuilt as:
//   * 256 co-resident workgroups (one per CU, 8 wavefronts each); the 8 workgroups of one head sit
//     on one XCD (block b -> XCD b % 8 is how the dispatcher places them; used for speed only);
//   * DPP lane permutes inside a wavefront, LDS staging across the wavefronts of a workgroup;
//   * between workgroups: 8-byte {epoch tag, fp32 payload} granules written by ONE write-through
//     (sc1) store each and swept with relaxed agent-scope loads until every tag carries this call's
//     epoch (guide G16 "R2": the data IS the flag; no fences, no separate flag, placement
//     independent).  Three exchanges per layer:
//       X1  q|k|v of a head         8 producers -> the same 8 consumers      384 granules / head
//       X2  split-KV softmax records 8 producers -> the head's leader          8 x 130 granules
//       X3  normalised attention out 32 leaders  -> all 256 workgroups         4096 granules
//   * only x carries a dependency: Wqkv, the KV cache and Wo do not depend on earlier phases, so
//     every workgroup requests its KV tiles before X1 resolves and its Wo rows before X2/X3
//     resolve -- the HBM stream of a CU never waits for a hand-off.
//
// Per workgroup (h = head, j = 0..7) the byte stream is 48 Wqkv rows (384 KB) -> 1/8 of the
// head's K and V (S=4096: 256 KB) -> 16 Wo rows (128 KB).
//
// Scope of this kernel: [out,in] weights, hidden 4096, 32 q heads = 32 kv heads, batch 1,
// contiguous or paged KV.  Everything else takes the stage pipeline (cf_decode_kernels.h).
#pragma once
#include "cf_decode_kernels.h"

namespace cf {

typedef unsigned long long u64;

constexpr int FUSED_WGS_C = 256;
struct FusedArgs {
    NormArgs na;
    const h16* Wqkv;
    const h16* Wo;
    const h16* k_cache;
    const h16* v_cache;
    const uint64_t* kptrs;
    const uint64_t* vptrs;
    int layer_id;
    int seq_len;
    const int32_t* indptr;
    const int32_t* indices;
    const int32_t* seq_lens;
    int page_shift;
    const float* cos;
    const float* sin;
    const int64_t* positions;
    int64_t rope_stride;
    int rope_style;
    h16* out;
    h16* residual_out;
    h16* k_new;
    h16* v_new;
    int write_cache;
    // persistent exchange state (zero-initialised once, then owned by the kernel)
    uint32_t* state;   // [0] epoch of the last completed call, [1] first error code (0 = none)
    u64* g_qkv;        // [32][384]
    u64* g_rec;        // [32][8][FUSED_REC]
    u64* g_attn;       // [4096]
    u64* g_xcc;        // [256]          XCC id each workgroup runs on (decides XCD-local hand-offs)
    u64* g_qkv_io;     // [32][8][384]   [in,out] weights: split-K partials of q|k|v per workgroup
    u64* g_part;       // [32][4096]     [in,out] weights: per-head partial outputs of the O projection
    // [out,in] phase 1 of k_fused_decode_mha: workgroup b produces the Wqkv row pairs [p_lo, p_lo + share), share = byte b % 8 of
    // p1_tab[b / 64], p_lo = p1_grp[b / 64] + ((b / 8) % 8) p1_rowsum[b / 64] + byte b % 8 of p1_pre[b / 64] -- dwords and
    // qwords of the kernarg segment, i.e. scalar loads and scalar arithmetic: a 257-entry table of shorts is read with a VECTOR
    // load, whose result comes back behind x / residual / rms_w (loads return in issue order) -- the first weight rows were
    // requested 0.76 us after the workgroup's start (profiles/r05_experiments.md section 12)
    unsigned long long p1_tab[4], p1_pre[4];
    unsigned int p1_rowsum[4], p1_grp[4];
    unsigned short p1_start[FUSED_WGS_C + 1];   // the multi-row kernels (cf_fused_kernel_b.h, _q.h): workgroup b produces the Wqkv rows / row pairs [p1_start[b], p1_start[b+1])
    int flags;         // debug/tuning bits (cf_debug_set_flags)
    u64* trace;        // debug: [256][16] wall-clock stamps (100 MHz) per workgroup, or null
    // head-parallel TP with the collective's publish folded into phase 3 (cf_layer_args.tp_*; protocol: cf_tp_kernels.h):
    // tp_world > 0 -> the shard kernels also write their 16 output values per workgroup as {epoch, fp16 x 2} granules into slot
    // tp_rank of every rank's receive area
    u64* tp_areas[8];
    int tp_rank, tp_world;
};

constexpr int TP_MAX_WORLD = 8;
constexpr int TP_HDR_GRANULES = 32;      // 256-byte header in front of the slots of a receive area

// The TP epoch of this call = the epoch word of this rank's own receive area + 1 (advanced by the gather that consumes the call:
// cf_tp_gather / cf_rmsnorm_tp_gather, stream-ordered behind this kernel).  Read once at kernel start through the scalar cache.
__device__ __forceinline__ unsigned tp_call_epoch(const FusedArgs& a) {
    return a.tp_world > 0 ? *reinterpret_cast<const __attribute__((address_space(4))) uint32_t*>(reinterpret_cast<uintptr_t>(a.tp_areas[a.tp_rank])) + 1u : 0u;
}
// The output values of a workgroup -- 2 NPW per wavefront, valid in lane 63 (sum64_lane63), 16 NPW consecutive values of `out` per
// workgroup -- go into slot tp_rank of every rank's area as 8 NPW consecutive {epoch, fp16 x 2} granules = ONE 64 NPW-byte piece
// per area: the wavefronts drop their pairs into LDS, wavefront 0 stores them -- 8 NPW consecutive lanes cover one area -- so a
// workgroup costs the fabric 8 write transactions (8 B from lane 63 of every wavefront = 64 per workgroup measured +0.9 us on
// the shard kernel whether issued as one instruction or eight).  Remote traffic is write-only (one xGMI link latency, no hop
// depends on another); SYSTEM scope because the reader is another GPU.  Called by every thread of the workgroup; `s_pub` =
// 8 NPW words of LDS nobody reads any more; `g0` = the first granule (out index / 2) of the workgroup.
template <int NPW>
__device__ __forceinline__ void tp_publish_wg(const FusedArgs& a, unsigned tp_epoch, int g0, const float (&v)[2 * NPW], unsigned* s_pub, int lane, int wave) {
    static_assert(NPW == 1 || NPW == 2 || NPW == 4, "8, 16 or 32 granules per workgroup");
    if (lane == 63) {
#pragma unroll
        for (int k = 0; k < NPW; ++k) {
            h16x2 pr;
            pr[0] = (h16)v[2 * k];
            pr[1] = (h16)v[2 * k + 1];
            s_pub[wave * NPW + k] = __builtin_bit_cast(unsigned, pr);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wave != 0) return;
    constexpr int G = 8 * NPW, APP = 64 / G;      // granules per workgroup; areas one store instruction covers
    const int k = lane % G;
    const u64 gran = ((u64)tp_epoch << 32) | (u64)s_pub[k];
    const int ng = 4096 / 2;
    const size_t at = TP_HDR_GRANULES + (size_t)(tp_epoch & 1u) * a.tp_world * ng + (size_t)a.tp_rank * ng + g0 + k;
#pragma unroll
    for (int pass = 0; pass < TP_MAX_WORLD / APP; ++pass) {
        const int p = pass * APP + lane / G;
        u64* dst = a.tp_areas[0];
#pragma unroll
        for (int q = 1; q < TP_MAX_WORLD; ++q) dst = p == q ? a.tp_areas[q] : dst;
        if (p < a.tp_world) __hip_atomic_store(dst + at, gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

#define CF_TRACE(slot)                                                                         \
    do {                                                                                       \
        if (a.trace && tid == 0) a.trace[(size_t)blockIdx.x * 16 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)

constexpr int FUSED_WGS = 256;
constexpr int FUSED_THREADS = 512;
constexpr int FUSED_HEADS = 32;
constexpr int FUSED_SPLITS = 8;          // workgroups per head
constexpr int FUSED_REC = 132;           // granules per record: o[128], m, l (+2 pad)
constexpr int FUSED_RECH = 66;           // the grouped-query / shard kernels' record: 64 fp16 pairs of o / l, then m, l
constexpr int FUSED_REC_G = 144;         // record stride in the workspace of k_fused_decode_mha: whole 128-B
                                         // lines, so XCD-local and write-through producers never share a line
constexpr int FUSED_GROUPS = 32;         // 16-lane groups per workgroup
constexpr unsigned FUSED_SPIN_LIMIT = 400000u;   // bounded spins: give up instead of hanging the GPU

// LDS carve (bytes, all 16-B aligned)
constexpr int FL_QKV = 0;                                 // float[384]
constexpr int FL_A = FL_QKV + 384 * 4;                    // float[4096]
constexpr int FL_O = FL_A + 4096 * 4;                     // float[9][128] (8 wavefront states + new token)
constexpr int FL_ML = FL_O + 33 * 128 * 4;                // float[33][2] (+pad)
constexpr int FL_REC = FL_ML + 272;                       // float[8][FUSED_REC]
constexpr int FL_IDX = FL_REC + 8 * FUSED_REC * 4;        // int[FUSED_MAX_IDX]
constexpr int FUSED_MAX_IDX = 16384;     // page-table entries one workgroup stages (64 KB)
constexpr int FL_CS = FL_IDX + FUSED_MAX_IDX * 4;         // float[256] cos|sin
constexpr int FL_CTL = FL_CS + 256 * 4;                   // int[32]
constexpr int FL_PART = FL_CTL + 128;                      // float[8][384]  [in,out]: wavefront partials of q|k|v
constexpr int FL_X1 = FL_PART + 8 * 384 * 4;              // float[8][384]  [in,out]: the 8 workgroups' partials (X1)
constexpr int FL_END = FL_X1 + 8 * 384 * 4;               // (FL_PART..FL_END = 24 KB doubles as float[8][512] in phase 3)
#ifndef CF_X3_ONE_POLLER
#define CF_X3_ONE_POLLER 1      // X3's cheap wait by one wavefront per workgroup (0: every wavefront watches its own heads)
#endif
// ask for more than half a CU's LDS so exactly one workgroup lands on each CU
constexpr int FUSED_LDS_BYTES = FL_END > 84 * 1024 ? FL_END : 84 * 1024;

__device__ __forceinline__ void granule_store(u64* p, unsigned epoch, float v) {
    __hip_atomic_store(p, ((u64)epoch << 32) | (u64)__builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

// XCD-local variant: a plain (write-back) store that stops in this XCD's L2 instead of being written
// through to memory.  Visible to agent-scope loads of workgroups on the SAME XCD only, but it does not
// queue behind the weight streams in the fabric (tools/ubench/hop_lat.hip: 0.2 us vs 0.43 us idle; the
// loaded difference is larger).  A producer uses it only after it has seen, through the ordinary
// write-through path, that its consumer runs on the same XCD (g_xcc).
__device__ __forceinline__ void granule_store_to(u64* p, unsigned epoch, float v, bool xcd_local) {
    const u64 g = ((u64)epoch << 32) | (u64)__builtin_bit_cast(unsigned, v);
    if (xcd_local) __hip_atomic_store(p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned my_xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 15u;
}

// A failed exchange is reported twice: in the workspace (state[1], cf_workspace_status) and, when the workspace was set up by
// cf_workspace_init, in a host-mapped word the library reads at the start of every later call (state[4..5] = its device
// address): the call AFTER a failed one returns CF_ELAUNCH even if nobody polls the workspace.
__device__ __forceinline__ void flag_exchange_error(uint32_t* err /* = state + 1 */, unsigned code) {
    atomicCAS(err, 0u, code);
    uint32_t* host = *reinterpret_cast<uint32_t* const*>(err + 3);
    if (host) __hip_atomic_store(host, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// Every failure path ends a workgroup through this: the epoch must advance even when a call fails, or the next call would
// take the granules the failed one left behind for its own.
#define CF_FAIL_RETURN()                                   \
    do {                                                   \
        if (b == 0 && tid == 0) a.state[0] = epoch;        \
        return;                                            \
    } while (0)

// LDS-only barrier: does not drain the vector-memory queue, so register prefetches stay in flight
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Cheap wait before a wide sweep: lanes < n watch ONE granule each (g[lane * stride]) until all n carry
// this epoch.  Only a hint -- the sweep that follows still checks every tag -- but while a workgroup waits
// it polls n granules instead of the whole block: 256 waiting workgroups re-reading 32 KB each per round
// cost about as much fabric bandwidth as their weight streams did.
// `missing_ok` > 0: return already when all but that many of the watched granules have arrived -- the sweep then polls the
// whole block for the straggler(s) and sees them one round trip sooner than hint-then-sweep would.
__device__ __forceinline__ void wait_hint(const u64* g, int n, int stride, unsigned epoch, int lane, int missing_ok = 0) {
    for (unsigned spin = 0; spin < FUSED_SPIN_LIMIT; ++spin) {
        u64 x = (u64)epoch << 32;
        if (lane < n) x = __hip_atomic_load(g + (size_t)lane * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (missing_ok == 0) {
            if (__all((unsigned)(x >> 32) == epoch)) break;
        } else if (__popcll(__ballot((unsigned)(x >> 32) != epoch)) <= missing_ok) break;
        __builtin_amdgcn_s_sleep(2);
    }
}

// ONE wavefront re-reads its granules until every tag == epoch, then drops the payloads in LDS.
template <int N, class T = float>
__device__ __forceinline__ bool sweep_granules(const u64* g, int count, unsigned epoch, T* dst, int lane,
                                               uint32_t* err, unsigned code) {
    unsigned v[N];
    for (unsigned spin = 0;; ++spin) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const int i = lane + WAVE * k;
            u64 x = (u64)epoch << 32;
            if (i < count) x = __hip_atomic_load(g + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v[k] = (unsigned)x;
            ok &= (unsigned)(x >> 32) == epoch;
        }
        if (__all(ok)) break;
        if (spin > FUSED_SPIN_LIMIT) {
            if (lane == 0) flag_exchange_error(err, code);
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int i = lane + WAVE * k;
        if (i < count) dst[i] = (T)__builtin_bit_cast(float, v[k]);
    }
    return true;
}

// sweep_granules with the 32-bit payloads stored as they are (here: two fp16 values per granule)
template <int N>
__device__ __forceinline__ bool sweep_granules_raw(const u64* g, int count, unsigned epoch, unsigned* dst, int lane, uint32_t* err, unsigned code) {
    unsigned v[N];
    for (unsigned spin = 0;; ++spin) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const int i = lane + WAVE * k;
            u64 x = (u64)epoch << 32;
            if (i < count) x = __hip_atomic_load(g + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v[k] = (unsigned)x;
            ok &= (unsigned)(x >> 32) == epoch;
        }
        if (__all(ok)) break;
        if (spin > FUSED_SPIN_LIMIT) {
            if (lane == 0) flag_exchange_error(err, code);
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int i = lane + WAVE * k;
        if (i < count) dst[i] = v[k];
    }
    return true;
}

// A wave-uniform read-only value through the scalar cache: lands in SGPRs (no vector register, no vmcnt wait).  Only for
// data no kernel in flight writes (page-table bounds, positions, cache base pointers).
template <class T>
__device__ __forceinline__ T scalar_load(const T* p) {
    return *reinterpret_cast<const __attribute__((address_space(4))) T*>(reinterpret_cast<uintptr_t>(p));
}

template <int U>
struct KvTile32 {
    h16x8 k[U], v[U];
};

// ONE kernel for every cached length: the length is read on the device (kernel_batch_sglang.cuh:118-122 reads it there too), and
// after the common phase-1 prologue a wave-uniform branch picks one of four straight copies of the rest of the kernel.  The copies
// never join again, so each keeps exact wait counts (a join with the tile loop would make the short path wait for freshly requested
// Wo rows before unrelated LDS traffic); one hipGraph captured once serves a sequence that grows through all of them.
//   arm 2 (S <= 1024): one 128-token tile per workgroup (4 rows per lane-group), flat phase-1 shares; up to 128 tokens the head's
//                      split 0 holds them all and waits for no record (no X2);
//   arm 3 (S <= 2048): one 256-token tile;
//   arm 1 (S <= 4096): two 256-token tiles requested before X1;
//   arm 4 (longer)   : those two tiles, then 128-token tiles streamed two deep in a loop, Wo requested after the loop; page
//                      numbers beyond the FUSED_MAX_IDX staged in LDS are read through L2 (any length works, no host bound).
//   At short sequences the larger / second tile would be mostly clamped duplicate rows or dummy lines that still cost issue
//   slots and L2 traffic.  state[2] records the arm the last call took (cf_workspace_last_arm; tests assert it).
// IO = true: weights in the reference's plain [in,out] orientation (chat/llama/model.py:317-322):
//   phase 1 streams this head's 256-B column strips of 512 input rows per workgroup (split-K: X1 sums 8
//   partials in fixed order), phase 3 streams head h's 128 input rows x a 512-column strip of Wo and a
//   fourth exchange X4 sums the 32 per-head partials of each output column in fixed order.
template <int V>
struct FusedArm { static constexpr int value = V; };
constexpr int FUSED_ARM_TWO = 1, FUSED_ARM_TILE128 = 2, FUSED_ARM_TILE256 = 3, FUSED_ARM_LONG = 4;

template <bool IO>
__global__ __launch_bounds__(FUSED_THREADS, 2) void k_fused_decode_mha(FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_qkv = reinterpret_cast<float*>(smem + FL_QKV);
    float* s_a = reinterpret_cast<float*>(smem + FL_A);
    float(*s_o)[HEAD_DIM] = reinterpret_cast<float(*)[HEAD_DIM]>(smem + FL_O);
    float(*s_ml)[2] = reinterpret_cast<float(*)[2]>(smem + FL_ML);
    float(*s_rec)[FUSED_REC] = reinterpret_cast<float(*)[FUSED_REC]>(smem + FL_REC);
    int* s_idx = reinterpret_cast<int*>(smem + FL_IDX);
    float* s_cs = reinterpret_cast<float*>(smem + FL_CS);
    int* s_ctl = reinterpret_cast<int*>(smem + FL_CTL);
    float(*s_part)[384] = reinterpret_cast<float(*)[384]>(smem + FL_PART);
    float(*s_x1)[384] = reinterpret_cast<float(*)[384]>(smem + FL_X1);
    float(*s_red)[512] = reinterpret_cast<float(*)[512]>(smem + FL_PART);
    float* s_x4 = reinterpret_cast<float*>(smem + FL_X1);   // float[32][16] view (X4)

    constexpr int HID = 4096;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, gid = wave * 4 + (lane >> 4), d0 = l16 * 8;
    // debug bits 1/2 permute the block -> work map (bit 1: swap the row groups 0-7 <-> 8-15 of every XCD,
    // bit 2: swap neighbouring XCDs) to tell position effects from data effects in the timeline
    const int b = blockIdx.x ^ ((a.flags & 2) ? 64 : 0) ^ ((a.flags & 4) ? 1 : 0);
    // Rows come through a buffer resource: a slot this wavefront does not own gets an offset beyond the buffer --
    // the instruction still issues (same code path and same wait counts for every wavefront), touches no memory
    // and returns zeros.
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<h16*>(a.Wqkv), 0, IO ? 0 : 3 * 4096 * 4096 * 2, 0x00020000);
    auto p1_load_pair = [&](RowGroup<8, 2>& t, int pair, bool mine) {
        const int voff = mine ? pair * (2 * 4096 * 2) + lane * 16 : 0x40000000;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj)
                t.w[r][jj] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, voff + r * (4096 * 2) + jj * (WAVE * 16), 0, 2 /* nt */));
    };
#ifndef CF_EARLY_CHUNKS
#define CF_EARLY_CHUNKS 4      // 1-KB chunks of the first row of slot 0 requested here, before the start values are read (0: none)
#endif
    // Slot 0 is pair 8 b + w in every form of the deal (below): its address needs the weight pointer only.  The first chunks of its
    // first row go out HERE -- the start values' scalar loads and the arm branch put 0.7 us between a workgroup's start and the
    // requests behind them, and 2 KB per wavefront in flight is what a CU moves in that time.  (The whole pair carried across the
    // arm branch makes the allocator spill: profiles/r05_experiments.md section 12.)
    h16x8 early[CF_EARLY_CHUNKS > 0 ? CF_EARLY_CHUNKS : 1];
    if constexpr (!IO && CF_EARLY_CHUNKS > 0) {
        const int voff = (8 * b + wave) * (2 * 4096 * 2) + lane * 16;
#pragma unroll
        for (int jj = 0; jj < CF_EARLY_CHUNKS; ++jj)
            early[jj] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, voff + jj * (WAVE * 16), 0, 2 /* nt */));
    }
#ifndef CF_IO_EARLY
#define CF_IO_EARLY 4      // [in,out]: this many of the first half batch's 8 loads (256-B strips of 4 input rows each) requested here as well
#endif
    h16x8 io_early[CF_IO_EARLY > 0 ? CF_IO_EARLY : 1];
    if constexpr (IO && CF_IO_EARLY > 0) {
        const int hh = (a.flags & 1) ? (b >> 6) * 8 + (b & 7) : (b & 7) * 4 + (b >> 6), jj0 = (b >> 3) & 7;
        const h16* p = a.Wqkv + ((size_t)(512 * jj0 + 64 * wave + (lane >> 4))) * 4096 + hh * HEAD_DIM + (lane & 15) * 8;
#pragma unroll
        for (int u = 0; u < CF_IO_EARLY; ++u) io_early[u] = ld_stream(p + (size_t)u * 4 * 4096);
    }
    if (a.trace && tid == 0) {   // where this workgroup runs: HW_ID (se.sh.cu) | XCC_ID << 32
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        a.trace[(size_t)blockIdx.x * 16 + 13] = ((u64)xcc << 32) | hw;
    }
    // the 8 workgroups of a head share b % 8 (one XCD); flag bit 0 interleaves heads over XCDs
    const int h = (a.flags & 1) ? (b >> 6) * 8 + (b & 7) : (b & 7) * 4 + (b >> 6);
    const int j = (b >> 3) & 7;
    CF_TRACE(0);

    // ---- small first-level loads go out FIRST (loads return in issue order: behind the weight
    //      stream they would come back microseconds later) ------------------------------------------
    const h16* rp = a.na.residual ? a.na.residual : a.na.x;
    const float rs = a.na.residual ? 1.f : 0.f;
    const h16x8 xv = ld_h8(a.na.x + tid * 8), rv = ld_h8(rp + tid * 8), wv8 = ld_h8(a.na.rms_w + tid * 8);
    // wave-uniform start values.  [out,in]: ONE batch of scalar loads behind one wait (a load behind `if (pointer)` is a branch, and
    // the compiler waits for everything outstanding at every branch: seven dependent round trips).  An absent table is read from
    // the workspace's state words instead -- always mapped, the value is discarded.  (state[0] was written by the previous
    // launch: the scalar cache is invalidated at every kernel start.)  The [in,out] kernel keeps the plain form: the batch costs
    // it two spilled registers.
    unsigned epoch;
    int S = a.seq_len, ent0 = 0;
    int64_t roff = 0;
    const h16 *kc = a.k_cache, *vc = a.v_cache;
    int p1_lo_tab = 0, p1_hi_tab = 0;
    if constexpr (!IO) {
        const uint32_t* stp = a.state;
        const int32_t* ipp = a.indptr ? a.indptr : reinterpret_cast<const int32_t*>(stp);
        const int32_t* slp = a.seq_lens ? a.seq_lens : reinterpret_cast<const int32_t*>(stp);
        const int64_t* pop = a.positions ? a.positions : reinterpret_cast<const int64_t*>(stp);
        const uint64_t* kpp = a.kptrs ? a.kptrs + a.layer_id : reinterpret_cast<const uint64_t*>(stp);
        const uint64_t* vpp = a.vptrs ? a.vptrs + a.layer_id : reinterpret_cast<const uint64_t*>(stp);
        const unsigned ep0 = scalar_load(stp);
        const int ip0 = scalar_load(ipp), ip1 = scalar_load(ipp + 1), sl0 = scalar_load(slp);
        const int64_t po0 = scalar_load(pop);
        const uint64_t kp0 = scalar_load(kpp), vp0 = scalar_load(vpp);
        // phase-1 share of this workgroup (FusedArgs::p1_tab): kernarg words, scalar arithmetic
        const int s4 = b >> 6, sh = 8 * (b & 7);
        p1_lo_tab = (int)a.p1_grp[s4] + ((b >> 3) & 7) * (int)a.p1_rowsum[s4] + (int)((a.p1_pre[s4] >> sh) & 0xffu);
        p1_hi_tab = p1_lo_tab + (int)((a.p1_tab[s4] >> sh) & 0xffu);
        epoch = ep0 + 1u;
        if (a.indptr) {
            ent0 = ip0;
            S = a.seq_lens ? sl0 : ip1 - 1 - ent0;
        }
        if (a.positions) roff = po0 * a.rope_stride;
        if (a.kptrs) kc = reinterpret_cast<const h16*>(kp0);
        if (a.vptrs) vc = reinterpret_cast<const h16*>(vp0);
    } else {
        epoch = scalar_load(a.state) + 1u;
        if (a.indptr) {
            ent0 = scalar_load(a.indptr);
            S = a.seq_lens ? scalar_load(a.seq_lens) : scalar_load(a.indptr + 1) - 1 - ent0;
        }
        if (a.positions) roff = scalar_load(a.positions) * a.rope_stride;
        if (a.kptrs) kc = reinterpret_cast<const h16*>(scalar_load(a.kptrs + a.layer_id));
        if (a.vptrs) vc = reinterpret_cast<const h16*>(scalar_load(a.vptrs + a.layer_id));
    }
    const unsigned xcc = my_xcc_id();
    if (tid == 0) granule_store(a.g_xcc + b, epoch, __builtin_bit_cast(float, xcc));   // where this workgroup runs

    // ================= from here on: one straight copy per arm (see the kernel's header comment) =================================
    auto rest = [&](auto arm_c, auto deal_c) {
    constexpr int ARM = decltype(arm_c)::value;
    constexpr bool P1_DENSE = decltype(deal_c)::value != 0;      // how phase 1 deals the Wqkv rows (below)
    constexpr bool LONG = ARM == FUSED_ARM_LONG, TINY = ARM == FUSED_ARM_TILE128 || ARM == FUSED_ARM_TILE256;
    constexpr int U = ARM == FUSED_ARM_TILE128 ? 4 : 8;
    constexpr int TILE = FUSED_GROUPS * U;             // 256 tokens: the two tiles requested before X1
    // ---- weight stream of phase 1 ------------------------------------------------------------------
    RowGroup<8, 2> ga, gb;
    // [in,out]: a batch = 64 input rows (16 iterations x 4 lane-groups) of one matrix, 256 B per row
    const int irow = 512 * j + 64 * wave + (lane >> 4);    // first input row of this lane-group
    // half batch hb = 2*m + half: 32 input rows (8 iterations x 4 lane-groups) of matrix m; three in flight
    h16x8 ca[8], cb[8], cc[8];
    auto io_load = [&](h16x8 (&t)[8], int hb) {
        const h16* p = a.Wqkv + ((size_t)(hb >> 1) * HID + irow + (hb & 1) * 32) * HID + h * HEAD_DIM + l16 * 8;
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = ld_stream(p + (size_t)u * 4 * HID);
    };
    // [out,in]: the 6144 row pairs of Wqkv (rows 2p, 2p+1 of the [12288, 4096] matrix) are dealt to the wavefronts' four slots; any
    // workgroup can produce any row (the X1 consumers find q|k|v of their head by granule address), so the deal is a pure
    // performance knob.  Two forms (profiles/r05_experiments.md section 12):
    //   DENSE: slot s < 3 of wavefront w of workgroup b is pair 2048 s + 8 b + w -- equal shares, and the chip reads one third of
    //     Wqkv (q, then k, then v: 32 MB) front to back at a time.  Until round 5 an equal deal was 24 CONSECUTIVE pairs per
    //     workgroup: the dense form is 0.9-1.3 us faster at every length up to 2048 (S = 1024: 28.6 -> 27.4 us).
    //   TABLE: slot 0 dense, the pairs 2048 .. 6143 dealt in index order by the host's share table (cf_api.hip fill_p1_table:
    //     16..30 pairs per workgroup incl. the 8 of slot 0; odd XCDs and the workgroups 64..127 get fewer), slots 1 .. 3 of
    //     wavefront w: p_lo + w + 8 (slot - 1) < p_hi.  The table corrects lags that build up with a long phase 2 inside the
    //     two-tile arm; it was tuned at S = 4096 and pays from ~3600 cached tokens (S = 4096: 34.8 us with it, 35.6 dense;
    //     S = 3072: 32.8 / 32.4; S = 2304: 31.8 / 30.9); the loop arm measured level to 0.4 us better dense.
    // The device-side length picks the form with the arm (the two-tile arm exists in both forms: a run-time choice of the
    // slots' base and stride inside one copy measured 0.25-0.45 us slower at S = 3072 / 3584 than the straight dense copy): one graph
    // serves a growing sequence.
    int p_lo = 0, p_hi = 0;
    if constexpr (!IO && !P1_DENSE) {
        p_lo = p1_lo_tab;
        p_hi = p1_hi_tab;
    }
    auto p1_pair = [&](int slot) {
        if constexpr (P1_DENSE) return 2048 * slot + 8 * b + wave;
        else return slot == 0 ? 8 * b + wave : p_lo + wave + 8 * (slot - 1);
    };
    auto p1_mine = [&](int slot) {
        if constexpr (P1_DENSE) return slot < 3;
        else return slot == 0 || p_lo + wave + 8 * (slot - 1) < p_hi;
    };
    auto p1_load = [&](RowGroup<8, 2>& t, int slot) { p1_load_pair(t, p1_pair(slot), p1_mine(slot)); };
    if constexpr (!IO) {
        if constexpr (CF_EARLY_CHUNKS > 0) {      // slot 0: the chunks requested at the top, then the rest of the pair
            const int voff = (8 * b + wave) * (2 * 4096 * 2) + lane * 16;
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    if (r == 0 && jj < CF_EARLY_CHUNKS) ga.w[0][jj] = early[jj];
                    else ga.w[r][jj] = __builtin_bit_cast(h16x8, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, voff + r * (4096 * 2) + jj * (WAVE * 16), 0, 2 /* nt */));
                }
        } else p1_load(ga, 0);
        p1_load(gb, 1);
    } else {
        if constexpr (CF_IO_EARLY > 0) {      // half batch 0: the loads requested at the top of the kernel, then the rest
            const h16* p = a.Wqkv + ((size_t)irow) * HID + h * HEAD_DIM + l16 * 8;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (u < CF_IO_EARLY) ca[u] = io_early[u < CF_IO_EARLY ? u : 0];
                else ca[u] = ld_stream(p + (size_t)u * 4 * HID);
            }
        } else io_load(ca, 0);
        io_load(cb, 1);
        io_load(cc, 2);
    }
    // [in,out]: X1 is a hand-off among the head's own 8 workgroups, which are meant to share an XCD: their published
    // ids (lane i % 8: member i), requested behind the first rows, decide whether the partials may stay in that L2.
    u64 member_x = 0;
    if constexpr (IO)
        member_x = __hip_atomic_load(a.g_xcc + ((b & ~0x38) | ((lane & 7) << 3)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // ---- RMSNorm ONCE per workgroup: thread t owns elements [8t, 8t+8) -----------------------------
    float hx[8];
    {
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            hx[e] = __builtin_fmaf(rs, (float)rv[e], (float)xv[e]);
            ss = __builtin_fmaf(hx[e], hx[e], ss);
        }
        ss = sum64_lane63(ss);
        if (lane == 63) s_rec[0][wave] = ss;     // s_rec is free until X2
    }

    // ---- second-level loads (page-table slice, new-token slot, RoPE row): into registers now, into
    //      LDS after the first weight rows have been consumed -----------------------------------------
    const int ps = a.page_shift, pmask = (1 << ps) - 1;
    int tps = ((S + FUSED_SPLITS - 1) / FUSED_SPLITS + 31) & ~31;   // multiple of 32 (one token per lane-group row)
    tps = tps < 32 ? 32 : tps;
    // Up to 128 cached tokens the whole sequence is one tile of the head's split 0: that workgroup IS the head -- the other seven
    // hold an empty slice and publish no record, it waits for none: X2, a hand-off of ~1.5 us on an otherwise finished chip,
    // drops out of the chain (wave-uniform, inside the 128-token-tile arm only).
#ifndef CF_ONE_TILE
#define CF_ONE_TILE 1
#endif
    const bool one = CF_ONE_TILE && !IO && ARM == FUSED_ARM_TILE128 && S <= 128;
    if (one) tps = 128;
    const int t0 = j * tps;
    int t1 = t0 + tps;
    t1 = t1 < S ? t1 : S;
    const int e0 = t0 >> ps;
    const int max_idx = (a.flags & 64) ? 512 : FUSED_MAX_IDX;   // (debug bit 64: stage only what the pre-requested tiles need)
    int n_idx = 0, n_need = 0;     // page-table entries of this slice: all of them / those staged in LDS
    if (a.indptr && t1 > t0) {
        n_need = ((t1 - 1) >> ps) - e0 + 1;
        n_idx = n_need < max_idx ? n_need : max_idx;   // (a longer slice reads the rest through L2: arm 4)
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

    lds_barrier();   // partial sums of squares visible
    float xn[8][8];
    {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) tot += s_rec[0][w];
        const float rcp = __builtin_amdgcn_rsqf(tot / (float)HID + a.na.eps);
        float* s_xn = s_a;                        // s_a is free until X3
        f32x4 lo, hi;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            lo[e] = hx[e] * rcp * (float)wv8[e];
            hi[e] = hx[4 + e] * rcp * (float)wv8[4 + e];
        }
        *reinterpret_cast<f32x4*>(&s_xn[tid * 8]) = lo;
        *reinterpret_cast<f32x4*>(&s_xn[tid * 8 + 4]) = hi;
        lds_barrier();
        if constexpr (!IO) {
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const f32x4 p0 = *reinterpret_cast<const f32x4*>(&s_xn[(jj * WAVE + lane) * 8]);
                const f32x4 p1 = *reinterpret_cast<const f32x4*>(&s_xn[(jj * WAVE + lane) * 8 + 4]);
#pragma unroll
                for (int e = 0; e < 4; ++e) { xn[jj][e] = p0[e]; xn[jj][4 + e] = p1[e]; }
            }
        }
    }

    // ---- KV tiles of phase 2 are requested BEFORE q exists ----------------------------------------
    const size_t kvstride = (size_t)FUSED_HEADS * HEAD_DIM;
    const h16* kbase = kc + h * HEAD_DIM + d0;
    const h16* vbase = vc + h * HEAD_DIM + d0;
    // slot numbers first (one wave-uniform branch), then 2UU streaming loads back to back;
    // `tbase` = first token of the tile, a tile covers FUSED_GROUPS * UU tokens.
    // The loads are UNCONDITIONAL: a tile that lies wholly behind the slice reads one dummy line
    // instead.  A conditional request would put a control-flow join between the request and the next
    // use of the weight rows requested before it, and the compiler's wait count at a join is the
    // smaller of the two paths' -- i.e. the v rows of phase 1 would also wait for this whole tile
    // (measured: X1 resolved 5 us later than it had to).
    const h16* dummy = a.na.rms_w + d0;
    auto load_tile = [&](auto& t, int tbase, auto far_c) {
        constexpr int UU = sizeof(t.k) / sizeof(h16x8);
        constexpr bool FAR = decltype(far_c)::value != 0;  // page numbers through L2 instead of the staged slice
        const bool live = tbase < t1;                      // workgroup-uniform
        const h16* kb = live ? kbase : dummy;
        const h16* vb = live ? vbase : dummy;
        const size_t st = live ? kvstride : 0;
        size_t rows[UU];
        int tok[UU];
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            int tk = tbase + u * FUSED_GROUPS + gid;
            tk = tk < t1 ? tk : t1 - 1;
            tok[u] = tk > t0 ? tk : t0;
        }
        if (!a.indptr) {
#pragma unroll
            for (int u = 0; u < UU; ++u) rows[u] = (size_t)tok[u];
        } else if constexpr (FAR) {
#pragma unroll
            for (int u = 0; u < UU; ++u)
                rows[u] = ((size_t)a.indices[ent0 + (tok[u] >> ps)] << ps) + (size_t)(tok[u] & pmask);
        } else {
#pragma unroll
            for (int u = 0; u < UU; ++u) {
                int ei = (tok[u] >> ps) - e0;
                ei = ei < FUSED_MAX_IDX ? ei : FUSED_MAX_IDX - 1;
                rows[u] = ((size_t)s_idx[ei] << ps) + (size_t)(tok[u] & pmask);
            }
        }
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            t.k[u] = ld_stream(kb + rows[u] * st);
            t.v[u] = ld_stream(vb + rows[u] * st);
        }
    };
    constexpr int UL = 4, TILE_L = FUSED_GROUPS * UL;   // 128 tokens: tiles of the long-sequence loop
    constexpr FusedArm<0> NEAR{};
    constexpr FusedArm<1> FARIDX{};
    // ---- phase 1: this workgroup's share of the Wqkv rows ----------------------------------------
    float pacc[3][8];      // [in,out]: this lane's 8 columns of q|k|v over its 16 input rows
    auto io_fma = [&](const h16x8 (&t)[8], int hb) {
        const float* xs = s_a + irow + (hb & 1) * 32;   // normalised activations (LDS), one per input row
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float xv1 = xs[u * 4];
#pragma unroll
            for (int e = 0; e < 8; ++e) pacc[hb >> 1][e] = __builtin_fmaf((float)t[u][e], xv1, pacc[hb >> 1][e]);
        }
    };
    // second-level values -> LDS (they came back ahead of the weight rows: they were requested first)
    auto stage_second_level = [&]() {
        if (tid < n_idx) s_idx[tid] = idx_reg;
        for (int i = tid + FUSED_THREADS; i < n_idx; i += FUSED_THREADS) s_idx[i] = a.indices[ent0 + e0 + i];
        if (tid < 256) s_cs[tid] = cs_reg;
        if (tid == 0) s_ctl[20] = slot_reg;
        lds_barrier();   // s_idx / s_cs / slot visible
    };
    // rows 2p, 2p+1 -> granules of (head, q|k|v, index): row r = m*4096 + head*128 + i
    auto p1_dot_publish = [&](const RowGroup<8, 2>& t, int slot) {
        float res[2];
        t.dot(xn, res);
        if (lane == 63 && p1_mine(slot)) {
            const int r = 2 * p1_pair(slot);
            u64* gp = a.g_qkv + (size_t)((r & 4095) >> 7) * 384 + (r >> 12) * 128 + (r & 127);
            granule_store(gp, epoch, res[0]);
            granule_store(gp + 1, epoch, res[1]);
        }
    };
    if constexpr (!IO) {
        p1_dot_publish(ga, 0);
        CF_TRACE(14);
        p1_load(ga, 2);
        p1_dot_publish(gb, 1);
        CF_TRACE(15);
        p1_load(gb, 3);
        stage_second_level();
    } else {
#pragma unroll
        for (int mm = 0; mm < 3; ++mm)
#pragma unroll
            for (int e = 0; e < 8; ++e) pacc[mm][e] = 0.f;
        io_fma(ca, 0);
        io_load(ca, 3);
        io_fma(cb, 1);
        io_load(cb, 4);
        io_fma(cc, 2);
        io_load(cc, 5);
        stage_second_level();
    }
    if constexpr (IO) {
        io_fma(ca, 3);
        io_fma(cb, 4);
        __builtin_amdgcn_sched_barrier(0);
        io_fma(cc, 5);
        // 4 lane-groups (different input rows, same columns) -> lanes 0..15; 8 wavefronts -> LDS
#pragma unroll
        for (int mm = 0; mm < 3; ++mm)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                pacc[mm][e] = xsum32(xsum16(pacc[mm][e]));
            }
        if (lane < 16) {
#pragma unroll
            for (int mm = 0; mm < 3; ++mm)
#pragma unroll
                for (int e = 0; e < 8; ++e) s_part[wave][mm * HEAD_DIM + d0 + e] = pacc[mm][e];
        }
        __builtin_amdgcn_sched_barrier(0);
        // both K/V tiles are requested once the partial sums have left the registers (requesting tile A
        // earlier, as the [out,in] variant does, makes the allocator spill it straight back to scratch)
        lds_barrier();
        if (tid < 384) {   // this workgroup's split-K partial of q|k|v (fixed-order sum over wavefronts)
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += s_part[w][tid];
            // (per wavefront: one that does not see all eight ids yet writes through)
            const bool x1_local = __all((unsigned)(member_x >> 32) == epoch && (unsigned)member_x == xcc);
            granule_store_to(a.g_qkv_io + ((size_t)h * FUSED_SPLITS + j) * 384 + tid, epoch, v, x1_local);
        }
        CF_TRACE(1);   // phase 1 done (partial published)
    }
    KvTile32<U> ta;
    KvTile32<TINY ? 1 : U> tb;
    if constexpr (!IO) {
        // the K/V tiles of phase 2 are requested before q exists, as early as the registers allow
        p1_dot_publish(ga, 2);
        load_tile(ta, t0, NEAR);
        p1_dot_publish(gb, 3);
        if constexpr (!TINY) load_tile(tb, t0 + TILE, NEAR);
    }
    // Wo rows are requested as soon as tile A's registers retire and stay in flight through X2/X3.
    RowGroup<8, 2> go;
    auto load_wo = [&](RowGroup<8, 2>& t) {
        if constexpr (!IO) {      // [out,in]: 2 output rows per wavefront
            t.load(a.Wo, 16 * b + 2 * wave, HID, HID, lane);
        } else {   // [in,out]: 16 of head h's input rows per wavefront x this workgroup's 512-column strip
            const h16* p = a.Wo + ((size_t)h * HEAD_DIM + 16 * wave) * HID + 512 * j + lane * 8;
#pragma unroll
            for (int u = 0; u < 16; ++u) t.w[u >> 3][u & 7] = ld_stream(p + (size_t)u * HID);
        }
    };
    if constexpr (!IO) {
        CF_TRACE(1);   // phase 1 done (all rows published)

        // ---- X1: gather q|k|v of this head -------------------------------------------------------
        if (wave == 0) {
            const bool ok = sweep_granules<6>(a.g_qkv + (size_t)h * 384, 384, epoch, s_qkv, lane, a.state + 1, 1u);
            if (lane == 0) s_ctl[0] = ok;
        }
        lds_barrier();
        if (!s_ctl[0]) CF_FAIL_RETURN();
    } else {
        // (the partial is published BEFORE the tiles are requested: their 32 loads per wavefront enter a
        //  saturated queue slowly, and the other 7 workgroups of the head wait for this partial)
        load_tile(ta, t0, NEAR);
        if constexpr (!TINY) load_tile(tb, t0 + TILE, NEAR);

        // ---- X1: the head's 8 split-K partials, summed in fixed order (replaces cluster_reduce<LINEAR>,
        //      dsm.cuh:20-134) ------------------------------------------------------------------------
        {
            const bool ok = sweep_granules<6>(a.g_qkv_io + ((size_t)h * FUSED_SPLITS + wave) * 384, 384, epoch,
                                              s_x1[wave], lane, a.state + 1, 1u);
            if (lane == 0) s_ctl[1 + wave] = ok;
        }
        lds_barrier();
        {
            bool all_ok = true;
            for (int w = 0; w < 8; ++w) all_ok &= s_ctl[1 + w] != 0;
            if (!all_ok) CF_FAIL_RETURN();
        }
        if (tid < 384) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += s_x1[w][tid];
            s_qkv[tid] = v;
        }
        lds_barrier();
    }
    CF_TRACE(2);   // X1 resolved
    // does the head's leader (split 0) run on this XCD?  (requested now, looked at when the record is
    // published; written by the leader at its start, so long visible -- if not, take the slow path)
    const u64 lead_x = __hip_atomic_load(a.g_xcc + (b & ~0x38), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // ---- RoPE(q), scaled for base-2 softmax --------------------------------------------------------
    const float qscale = 1.44269504088896340736f * 0.08838834764831845f;
    float q[8];
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
    rope_lds(s_qkv, q);
    h16x8 qh;   // q rounded to fp16, as the reference keeps it (kernel.cuh:299-314): q.k runs on v_dot2_f32_f16.
                // Phase 2 starts when X1 resolves, with both tiles already on chip: it is VALU time, two
                // wavefronts per SIMD, on the critical path.
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        q[e] *= qscale;
        qh[e] = (h16)q[e];
    }
    CF_TRACE(7);   // q ready

    // ---- phase 2: flash-decode over this workgroup's token slice ------------------------------------
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
    compute_tile(ta, t0);       // (a tile behind the slice is all-masked: state unchanged)
    CF_TRACE(8);   // tile A consumed
    if constexpr (!LONG) {
        // Requesting the 16 Wo rows takes a wavefront ~2 us (the CU admits requests at ~25 GB/s; the instruction stream waits
        // at each one), and the two wavefronts of a SIMD would stand there together: wavefronts 0-3 request before tile B,
        // 4-7 after it, so one computes while the other one's requests go out (-0.2 us per call).  Two straight-line copies:
        // a join between request and use would cost the exact wait counts.
        if (TINY || wave < 4) {
            load_wo(go);
            CF_TRACE(9);   // Wo requested
            if constexpr (!TINY) compute_tile(tb, t0 + TILE);
        } else {
            if constexpr (!TINY) compute_tile(tb, t0 + TILE);
            load_wo(go);
        }
        CF_TRACE(10);  // tile B consumed
    } else {
        // continue in 128-token tiles (half the registers, still two tiles in flight)
        KvTile32<UL> la, lb;
        const int tl = t0 + 2 * TILE;
        if (n_need <= max_idx) {            // (workgroup-uniform) the whole slice of the page table is staged in LDS
            load_tile(la, tl, NEAR);
            compute_tile(tb, t0 + TILE);
            for (int tt = tl; tt < t1; tt += 2 * TILE_L) {
                load_tile(lb, tt + TILE_L, NEAR);
                compute_tile(la, tt);
                load_tile(la, tt + 2 * TILE_L, NEAR);
                compute_tile(lb, tt + TILE_L);
            }
        } else {                            // a slice longer than the staged part: page numbers through L2
            load_tile(la, tl, FARIDX);
            compute_tile(tb, t0 + TILE);
            for (int tt = tl; tt < t1; tt += 2 * TILE_L) {
                load_tile(lb, tt + TILE_L, FARIDX);
                compute_tile(la, tt);
                load_tile(la, tt + 2 * TILE_L, FARIDX);
                compute_tile(lb, tt + TILE_L);
            }
        }
        load_wo(go);
    }

    // merge the 4 lane-groups of this wavefront in registers (lanes l, l+16, l+32, l+48 hold the same
    // dims for different tokens), then 8 wavefront states (+ the new token) meet in LDS
    {
        const float mw = xmax32(xmax16(m));
        const float sc = fast_exp2(m - mw);
        l = xsum32(xsum16(l * sc));
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] *= sc;
        float r0, r1;
        xsum_rows8(o, r0, r1);      // (row r of the wavefront ends up with dims d0 + xrow_e(r) and d0 + 4 + xrow_e(r))
        m = mw;
        CF_TRACE(11);  // wavefront merge done
        const int e0 = xrow_e(lane >> 4);
        s_o[wave][d0 + e0] = r0;
        s_o[wave][d0 + 4 + e0] = r1;
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
        for (int e = 0; e < 8; ++e) s_o[8][d0 + e] = vf[e];
        if (l16 == 0) { s_ml[8][0] = sn; s_ml[8][1] = 1.f; }
    }
    CF_TRACE(12);  // before the phase-2 barrier (wavefront 0)
    lds_barrier();
    CF_TRACE(3);   // phase 2 done

    // ---- X2: one record per workgroup -> the head's leader ---------------------------------------
    const bool rec_local = (unsigned)(lead_x >> 32) == epoch && (unsigned)lead_x == xcc;
    if (tid < HEAD_DIM + 2) {
        const int nst = j == 0 ? 9 : 8;
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
        if (one) {      // the head's only record stays in its workgroup, the seven empty ones are written beside it
            if (j == 0) {
                s_rec[0][tid] = val;
#pragma unroll
                for (int w = 1; w < FUSED_SPLITS; ++w) s_rec[w][tid] = tid == HEAD_DIM ? NEG_BIG : 0.f;
            }
        } else granule_store_to(a.g_rec + ((size_t)h * FUSED_SPLITS + j) * FUSED_REC_G + tid, epoch, val, rec_local);
    }
    if (j == 0) {   // leader: wavefront w gathers record w, then the head's softmax merge
        if (one) {
            lds_barrier();
        } else {
            const bool ok = sweep_granules<3>(a.g_rec + ((size_t)h * FUSED_SPLITS + wave) * FUSED_REC_G, HEAD_DIM + 2, epoch,
                                              s_rec[wave], lane, a.state + 1, 2u);
            if (lane == 0) s_ctl[1 + wave] = ok;
            lds_barrier();
            bool all_ok = true;
            for (int w = 0; w < 8; ++w) all_ok &= s_ctl[1 + w] != 0;
            if (!all_ok) CF_FAIL_RETURN();
        }
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
            if constexpr (IO) {
                granule_store(a.g_attn + (size_t)h * HEAD_DIM + tid, epoch, acc / L);
            } else {
                // two fp16 values per granule: phase 3 consumes the attention output in fp16 (the reference rounds it there too,
                // kernel.cuh:553-559), and X3 -- every workgroup gathers all of it -- moves half the granules
                const float mine = acc / L, next = __shfl_down(mine, 1);
                h16x2 pr;
                pr[0] = (h16)mine;
                pr[1] = (h16)next;
                if (!(tid & 1)) granule_store(a.g_attn + (size_t)h * (HEAD_DIM / 2) + (tid >> 1), epoch, __builtin_bit_cast(float, pr));
            }
        }
    }

    CF_TRACE(4);   // record published (leader: head merged + published)
    if constexpr (!IO) {
        // ---- X3: every workgroup gathers the full attention output --------------------------------
        {
            // (last pair of 4 heads; with two of them there the sweep takes over: -0.3 us per call against waiting for all four)
            // ONE wavefront watches the last pair of the 32 heads (until half of them are there), the others wait at the LDS
            // barrier: 8 x fewer pollers on the lines the leaders are about to write (every wavefront watching its own 4 heads:
            // +0.10 us at S = 4096, +0.11 at 2048, +0.06 at 8192, +0.05 at 1024; same-process alternation, 10 rounds, sd 0.01-0.03)
#if CF_X3_ONE_POLLER
            if (wave == 0) wait_hint(a.g_attn + HEAD_DIM / 2 - 1, 32, HEAD_DIM / 2, epoch, lane, 16);
            lds_barrier();
#else
            wait_hint(a.g_attn + wave * 256 + HEAD_DIM / 2 - 1, 4, HEAD_DIM / 2, epoch, lane, 2);
#endif
            // (fp16 pairs: phase 3 reads half the LDS bytes and runs on v_dot2_f32_f16)
            const bool ok = sweep_granules_raw<4>(a.g_attn + wave * 256, 256, epoch, reinterpret_cast<unsigned*>(s_a) + wave * 256, lane,
                                                  a.state + 1, 3u);
            if (lane == 0) s_ctl[9 + wave] = ok;   // own slots: a slow wavefront may still be reading X2's
        }
        lds_barrier();
        {
            bool all_ok = true;
            for (int w = 0; w < 8; ++w) all_ok &= s_ctl[9 + w] != 0;
            if (!all_ok) CF_FAIL_RETURN();
        }
        CF_TRACE(5);   // X3 resolved
        // ---- phase 3: 16 rows of Wo per workgroup -----------------------------------------------------
        h16x8 av[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) av[jj] = *reinterpret_cast<const h16x8*>(reinterpret_cast<const h16*>(s_a) + (jj * WAVE + lane) * 8);
        float res[2];
        go.dot_h(av, res);
        if (lane == 63) {
            a.out[16 * b + 2 * wave] = (h16)res[0];
            a.out[16 * b + 2 * wave + 1] = (h16)res[1];
        }
    } else {
        // ---- X3: the head's own attention output is all this workgroup needs ------------------------
        if (wave == 0) {
            const bool ok = sweep_granules<2>(a.g_attn + (size_t)h * HEAD_DIM, HEAD_DIM, epoch, s_a, lane, a.state + 1, 3u);
            if (lane == 0) s_ctl[9] = ok;
        }
        lds_barrier();
        if (!s_ctl[9]) CF_FAIL_RETURN();
        CF_TRACE(5);   // X3 resolved
        // ---- phase 3: head h's 128 input rows x a 512-column strip of Wo -> per-head partial outputs --
        {
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const float av1 = s_a[16 * wave + u];
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = __builtin_fmaf((float)go.w[u >> 3][u & 7][e], av1, acc[e]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) s_red[wave][lane * 8 + e] = acc[e];
        }
        lds_barrier();
        {   // 512 columns, one per thread: fixed-order sum over the 8 wavefronts
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += s_red[w][tid];
            granule_store(a.g_part + (size_t)h * HID + 512 * j + tid, epoch, v);
        }
        // ---- X4: cross-head sum of this workgroup's 16 output columns (replaces the fp16 atomicAdd of
        //      kernel.cuh:600,618 by a fixed-order fp32 sum) ------------------------------------------
        {
            const int hh = 4 * wave + (lane >> 4), c = lane & 15;
            const u64* g = a.g_part + (size_t)hh * HID + 16 * b + c;
            unsigned v = 0;
            bool ok = false;
            for (unsigned spin = 0; spin <= FUSED_SPIN_LIMIT; ++spin) {
                const u64 x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v = (unsigned)x;
                if (__all((unsigned)(x >> 32) == epoch)) { ok = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (!ok && lane == 0) flag_exchange_error(a.state + 1, 5u);
            s_x4[hh * 16 + c] = __builtin_bit_cast(float, v);
            if (lane == 0) s_ctl[17 + wave] = ok;
        }
        lds_barrier();
        {
            bool all_ok = true;
            for (int w = 0; w < 8; ++w) all_ok &= s_ctl[17 + w] != 0;
            if (!all_ok) CF_FAIL_RETURN();
        }
        if (tid < 16) {
            float v = 0.f;
#pragma unroll
            for (int hh = 0; hh < FUSED_HEADS; ++hh) v += s_x4[hh * 16 + tid];
            a.out[16 * b + tid] = (h16)v;
        }
    }
    // residual_out may alias residual: every workgroup read residual in phase 1, and X3 completing
    // means all of them are past phase 1
    if (a.residual_out && tid < 16) {
        const int i = 16 * b + tid;
        a.residual_out[i] = (h16)((float)a.na.x[i] + (float)a.na.residual[i]);
    }
    if (b == 0 && tid == 0) {
        a.state[0] = epoch;
        a.state[2] = (uint32_t)ARM;      // which arm this call took (cf_workspace_last_arm)
    }
    CF_TRACE(6);
    };   // rest
#ifndef CF_P1_TABLE_FROM
#define CF_P1_TABLE_FROM 3585      // the two-tile arm deals phase 1 by the share table from this many cached tokens, dense below
#endif
    constexpr FusedArm<1> DENSE{};
    constexpr FusedArm<0> TABLE{};
    if (S <= 8 * 128) rest(FusedArm<FUSED_ARM_TILE128>{}, DENSE);
    else if (S <= 8 * 256) rest(FusedArm<FUSED_ARM_TILE256>{}, DENSE);
    else if (S <= 8 * 2 * 256) {
        if (IO || S < CF_P1_TABLE_FROM) rest(FusedArm<FUSED_ARM_TWO>{}, DENSE);
        else rest(FusedArm<FUSED_ARM_TWO>{}, TABLE);
    } else rest(FusedArm<FUSED_ARM_LONG>{}, DENSE);
}

}  // namespace cf
