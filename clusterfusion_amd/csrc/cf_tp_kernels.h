// cf_tp_kernels.h -- the collective of head-parallel TP at batch 1: a ONE-SHOT all-reduce of the [hidden] fp16 O-projection
// partial over peer-mapped receive areas (xGMI point-to-point writes), the same tagged-granule protocol the persistent
// kernels use between workgroups (cf_fused_kernel.h), one level up.
//
// Why: at batch 1 the message is 8 KB -- pure latency.  A ring or tree all-reduce is a chain of dependent hops and host-side
// bookkeeping per call; here every rank WRITES its partial straight into slot `rank` of every peer's receive area (7 remote
// stores per granule over the 7 xGMI links of a GPU, no hop depends on another), then POLLS ITS OWN memory until the `world`
// slots carry this call's epoch and sums them in rank order in fp32: one link latency, identical bits on every rank
// (fixed order), no atomics, graph-capturable (the epoch lives in the area and is advanced by the kernel).
//   receive area of a rank: [0] state word (epoch of the last completed call) | ... | granules [2][world][n / 2]: {epoch, fp16 x 2}
// Two slot sets, used by epoch parity: a peer that has finished call e may already write call e + 1 while this rank still reads
// call e (it cannot reach e + 2 before this rank has published e + 1), so one call of lead must not touch what is being read.
// Remote traffic is write-only; every poll is local.  Stores and polls are SYSTEM scope (the writer is another GPU).
// Reference: the contract is the RowParallelLinear all-reduce of /root/reference/chat/llama/model.py:208-235; the reference's
// fused path does not shard (model.py:306-311).  N > 1 over xGMI is UNMEASURED here (no multi-GPU box): on one GPU the
// protocol runs with virtual ranks (tests/test_parity_gpu.py::test_tp_oneshot_*), and between two processes sharing a device.
#pragma once
#include "cf_fused_kernel.h"

namespace cf {

// (TP_MAX_WORLD, TP_HDR_GRANULES and the publish half used by the layer kernels -- tp_publish_pair -- live in cf_fused_kernel.h)

struct TpOneShotArgs {
    const h16* partial;                  // [n] this rank's partial
    h16* out;                            // [n] the sum (may alias partial)
    u64* areas[TP_MAX_WORLD];            // every rank's receive area as mapped into THIS process (areas[rank] = own)
    int n, rank, world;
    int flags;                           // bit 0: publish only (test hook: a virtual rank that does not gather)
                                         // bit 1: gather only -- the partial was published by the layer kernel's phase 3
};

// A gather that gave up is reported twice, like a failed exchange of the layer kernels: in the area's error word (word 1,
// cf_tp_area_status) and in the device's host-mapped sticky word (its device address sits in words 4..5 of the area when the
// area was set up by cf_tp_area_alloc): the next layer call of the process on this device returns CF_ELAUNCH.
__device__ __forceinline__ void tp_flag_error(u64* own) {
    uint32_t* w = reinterpret_cast<uint32_t*>(own);
    atomicCAS(w + 1, 0u, 7u);
    uint32_t* host = *reinterpret_cast<uint32_t* const*>(w + 4);
    if (host) __hip_atomic_store(host, 7u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(256) void k_tp_oneshot_allreduce(TpOneShotArgs a) {
    const int tid = threadIdx.x, g = blockIdx.x * 256 + tid, ng = a.n / 2;
    u64* own = a.areas[a.rank];
    const unsigned epoch = scalar_load(reinterpret_cast<const uint32_t*>(own)) + 1u;
    const bool live = g < ng;
    unsigned mine = 0;
    if (live && !(a.flags & 2)) mine = *((const CF_GLOBAL unsigned*)(reinterpret_cast<const unsigned*>(a.partial) + g));
    const u64 gran = ((u64)epoch << 32) | mine;
    const size_t set = (size_t)(epoch & 1u) * a.world * ng;      // slot set of this call
    if (live && !(a.flags & 2)) {
        for (int p = 0; p < a.world; ++p)      // remote write-only traffic: slot `rank` of every rank's area (own included)
            __hip_atomic_store(a.areas[p] + TP_HDR_GRANULES + set + (size_t)a.rank * ng + g, gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    float s0 = 0.f, s1 = 0.f;
    bool ok = true;
    if (live && !(a.flags & 1)) {
        // every rank's slot is polled in the SAME round (one system-scope round trip for the whole gather when the peers have
        // published; a loop of per-rank spins was `world` dependent round trips: round 6), the sum still runs in rank order
        u64 x[TP_MAX_WORLD];
        for (unsigned spin = 0;; ++spin) {
            bool good = true;
#pragma unroll
            for (int p = 0; p < TP_MAX_WORLD; ++p) {
                x[p] = (u64)epoch << 32;
                if (p < a.world) x[p] = __hip_atomic_load(own + TP_HDR_GRANULES + set + (size_t)p * ng + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                good &= (unsigned)(x[p] >> 32) == epoch;
            }
            if (good) break;
            if (spin > 4u * FUSED_SPIN_LIMIT) { ok = false; break; }      // bounded (~2 s): a lost peer must not hang the GPU
            __builtin_amdgcn_s_sleep(2);
        }
#pragma unroll
        for (int p = 0; p < TP_MAX_WORLD; ++p) {      // fixed order: the same bits on every rank
            if (p < a.world) {
                const h16x2 v = __builtin_bit_cast(h16x2, (unsigned)x[p]);
                s0 += (float)v[0];
                s1 += (float)v[1];
            }
        }
        h16x2 r;
        r[0] = (h16)s0;
        r[1] = (h16)s1;
        // a slot that never arrived: the sum would be built from stale bits -- poison it (NaN) so that it cannot pass for a result
        reinterpret_cast<unsigned*>(a.out)[g] = ok ? __builtin_bit_cast(unsigned, r) : 0x7e007e00u;
    }
    if (!ok) tp_flag_error(own);
    // the epoch advances when the LAST workgroup of this launch is done (a later call must not reuse it while one still polls)
    __syncthreads();
    if (tid == 0) {
        const unsigned done = atomicAdd(reinterpret_cast<uint32_t*>(own) + 2, 1u) + 1u;
        if (done == gridDim.x) {
            reinterpret_cast<uint32_t*>(own)[2] = 0u;
            __hip_atomic_store(reinterpret_cast<uint32_t*>(own), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// The gather half folded into the op that consumes the all-reduced attention output: the fused add + RMSNorm between the
// attention block and the FFN (cf_rmsnorm's residual form; chat/llama/model.py:492,519).  One workgroup (batch 1): thread t owns
// the 8 output pairs t, t + 256, .. of the row; it polls the `world` slots of each (fixed rank order: the same bits on every
// rank), rounds the sum to fp16 as the all-reduce would, adds the residual, and the row is normalised from registers (the
// arithmetic of cf_tp_gather followed by cf_rmsnorm; the sum of squares meets in another order: <= 1 ulp apart).
struct TpNormArgs {
    u64* areas[TP_MAX_WORLD];
    int rank, world, hidden;
    const h16* residual;      // nullable
    const h16* weight;
    float eps;
    h16* out;                 // RMSNorm(sum + residual) * weight
    h16* residual_out;        // fp16(sum + residual), nullable
    h16* sum_out;             // the all-reduced vector itself, nullable
};
template <int PAIRS>          // output pairs per thread: hidden / 512
__global__ __launch_bounds__(256) void k_rmsnorm_tp_gather(TpNormArgs a) {
    __shared__ float s_part[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ng = a.hidden / 2;
    u64* own = a.areas[a.rank];
    const unsigned epoch = scalar_load(reinterpret_cast<const uint32_t*>(own)) + 1u;
    const size_t set = (size_t)(epoch & 1u) * a.world * ng;
    float h[PAIRS][2];
    bool ok = true;
    // the operands that do not depend on the peers go out first (they used to be requested behind the polls and behind the barrier:
    // two more dependent round trips in a kernel that is nothing but latency)
    h16x2 res[PAIRS], wgt[PAIRS];
#pragma unroll
    for (int k = 0; k < PAIRS; ++k) {
        const int i = 2 * (tid + 256 * k);
        res[k] = a.residual ? *reinterpret_cast<const h16x2*>(a.residual + i) : h16x2{(h16)0.f, (h16)0.f};
        wgt[k] = *reinterpret_cast<const h16x2*>(a.weight + i);
    }
#pragma unroll
    for (int k = 0; k < PAIRS; ++k) h[k][0] = h[k][1] = 0.f;
    // RPR ranks' slots per round (RPR x PAIRS polls in flight: what the registers hold -- every rank at once up to hidden 2048, four
    // ranks at 4096, two beyond); a loop of per-rank spins was `world` dependent system-scope round trips (round 6).  The sum runs in
    // rank order whatever arrives first: the same bits as before and on every rank.
    constexpr int RPR = PAIRS <= 4 ? 8 : (PAIRS <= 8 ? 4 : 2);
    for (int p0 = 0; p0 < a.world; p0 += RPR) {
        u64 x[RPR][PAIRS];
        for (unsigned spin = 0;; ++spin) {
            bool good = true;
#pragma unroll
            for (int q = 0; q < RPR; ++q)
#pragma unroll
                for (int k = 0; k < PAIRS; ++k) {
                    x[q][k] = (u64)epoch << 32;
                    if (p0 + q < a.world) x[q][k] = __hip_atomic_load(own + TP_HDR_GRANULES + set + (size_t)(p0 + q) * ng + tid + 256 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    good &= (unsigned)(x[q][k] >> 32) == epoch;
                }
            if (good) break;
            if (spin > 4u * FUSED_SPIN_LIMIT) { ok = false; break; }
            __builtin_amdgcn_s_sleep(2);
        }
#pragma unroll
        for (int q = 0; q < RPR; ++q)
#pragma unroll
            for (int k = 0; k < PAIRS; ++k) {
                if (p0 + q < a.world) {
                    const h16x2 v = __builtin_bit_cast(h16x2, (unsigned)x[q][k]);
                    h[k][0] += (float)v[0];
                    h[k][1] += (float)v[1];
                }
            }
    }
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < PAIRS; ++k) {
        const int i = 2 * (tid + 256 * k);
        h16x2 sum;
        sum[0] = (h16)h[k][0];                 // the all-reduce's rounding
        sum[1] = (h16)h[k][1];
        if (!ok) sum = __builtin_bit_cast(h16x2, 0x7e007e00u);
        if (a.sum_out) *reinterpret_cast<h16x2*>(a.sum_out + i) = sum;
        float v0 = (float)sum[0], v1 = (float)sum[1];
        if (a.residual) {
            const h16x2 r = res[k];
            v0 += (float)r[0];
            v1 += (float)r[1];
            h16x2 ro;
            ro[0] = (h16)v0;
            ro[1] = (h16)v1;
            if (a.residual_out) *reinterpret_cast<h16x2*>(a.residual_out + i) = ro;      // (normalised below: the fp32 sum, as k_norm_rows does)
        }
        h[k][0] = v0;
        h[k][1] = v1;
        ss = __builtin_fmaf(v0, v0, __builtin_fmaf(v1, v1, ss));
    }
    ss = sum64_lane63(ss);
    if (lane == 63) s_part[wave] = ss;
    __syncthreads();
    const float rcp = __builtin_amdgcn_rsqf((s_part[0] + s_part[1] + s_part[2] + s_part[3]) / (float)a.hidden + a.eps);
#pragma unroll
    for (int k = 0; k < PAIRS; ++k) {
        const int i = 2 * (tid + 256 * k);
        const h16x2 w = wgt[k];
        h16x2 o;
        o[0] = (h16)(h[k][0] * rcp * (float)w[0]);
        o[1] = (h16)(h[k][1] * rcp * (float)w[1]);
        *reinterpret_cast<h16x2*>(a.out + i) = o;
    }
    if (!ok) tp_flag_error(own);
    if (tid == 0) __hip_atomic_store(reinterpret_cast<uint32_t*>(own), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (one workgroup: nobody else polls)
}

// The same op on EIGHT workgroups (hidden 4096 / 8192): with 8 ranks one workgroup has 128 KB of slots to poll through one CU --
// 5.2 us more than a plain fused add + RMSNorm on this GPU, slower than gather and norm launched separately (round 6,
// tools/tp_gather_bench.py).  Here workgroup w gathers, sums and adds the residual for its eighth of the row, publishes the partial
// sum of squares of that eighth as one {epoch, fp32} granule in the area's header (granules 16 .. 23, device scope: the readers are
// this launch's other workgroups), sweeps the eight partials -- one more round trip -- and normalises its own eighth: nothing funnels
// through one CU and nothing is read twice.  The partials meet in workgroup order on every workgroup and every rank: identical
// bits everywhere (the order differs from the one-workgroup kernel's: <= 1 ulp apart).  The epoch advances when the last
// workgroup is done, as in k_tp_oneshot_allreduce.
constexpr int TP_NORM_WGS = 8, TP_HDR_PARTIALS = 16;
template <int PAIRS>          // output pairs per thread: hidden / (512 x 8)
__global__ __launch_bounds__(256) void k_rmsnorm_tp_gather_mw(TpNormArgs a) {
    __shared__ float s_part[4];
    __shared__ float s_all[TP_NORM_WGS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ng = a.hidden / 2, w = blockIdx.x;
    u64* own = a.areas[a.rank];
    const int g0 = w * (ng / TP_NORM_WGS);      // first pair of this workgroup's eighth
    h16x2 res[PAIRS], wgt[PAIRS];
#pragma unroll
    for (int k = 0; k < PAIRS; ++k) {
        const int i = 2 * (g0 + tid + 256 * k);
        res[k] = a.residual ? *reinterpret_cast<const h16x2*>(a.residual + i) : h16x2{(h16)0.f, (h16)0.f};
        wgt[k] = *reinterpret_cast<const h16x2*>(a.weight + i);
    }
    const unsigned epoch = scalar_load(reinterpret_cast<const uint32_t*>(own)) + 1u;
    const size_t set = (size_t)(epoch & 1u) * a.world * ng;
    bool ok = true;
    u64 x[TP_MAX_WORLD][PAIRS];      // every rank's slots of this eighth in one round
    for (unsigned spin = 0;; ++spin) {
        bool good = true;
#pragma unroll
        for (int p = 0; p < TP_MAX_WORLD; ++p)
#pragma unroll
            for (int k = 0; k < PAIRS; ++k) {
                x[p][k] = (u64)epoch << 32;
                if (p < a.world) x[p][k] = __hip_atomic_load(own + TP_HDR_GRANULES + set + (size_t)p * ng + g0 + tid + 256 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                good &= (unsigned)(x[p][k] >> 32) == epoch;
            }
        if (good) break;
        if (spin > 4u * FUSED_SPIN_LIMIT) { ok = false; break; }
        __builtin_amdgcn_s_sleep(2);
    }
    float h[PAIRS][2], ss = 0.f;
#pragma unroll
    for (int k = 0; k < PAIRS; ++k) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int p = 0; p < TP_MAX_WORLD; ++p)      // fixed rank order
            if (p < a.world) {
                const h16x2 v = __builtin_bit_cast(h16x2, (unsigned)x[p][k]);
                s0 += (float)v[0];
                s1 += (float)v[1];
            }
        const int i = 2 * (g0 + tid + 256 * k);
        h16x2 sum;
        sum[0] = (h16)s0;                      // the all-reduce's rounding
        sum[1] = (h16)s1;
        if (!ok) sum = __builtin_bit_cast(h16x2, 0x7e007e00u);
        if (a.sum_out) *reinterpret_cast<h16x2*>(a.sum_out + i) = sum;
        float v0 = (float)sum[0], v1 = (float)sum[1];
        if (a.residual) {
            v0 += (float)res[k][0];
            v1 += (float)res[k][1];
            h16x2 ro;
            ro[0] = (h16)v0;
            ro[1] = (h16)v1;
            if (a.residual_out) *reinterpret_cast<h16x2*>(a.residual_out + i) = ro;
        }
        h[k][0] = v0;
        h[k][1] = v1;
        ss = __builtin_fmaf(v0, v0, __builtin_fmaf(v1, v1, ss));
    }
    ss = sum64_lane63(ss);
    if (lane == 63) s_part[wave] = ss;
    __syncthreads();
    // the eight partial sums of squares: publish mine, sweep all (wavefront 0), fixed order
    if (wave == 0) {
        if (lane == 0) granule_store(own + TP_HDR_PARTIALS + w, epoch, (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]));
        u64 y = (u64)epoch << 32;
        for (unsigned spin = 0;; ++spin) {
            if (lane < TP_NORM_WGS) y = __hip_atomic_load(own + TP_HDR_PARTIALS + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all((unsigned)(y >> 32) == epoch)) break;
            if (spin > 4u * FUSED_SPIN_LIMIT) { ok = false; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (lane < TP_NORM_WGS) s_all[lane] = __builtin_bit_cast(float, (unsigned)y);
    }
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < TP_NORM_WGS; ++q) tot += s_all[q];
    const float rcp = __builtin_amdgcn_rsqf(tot / (float)a.hidden + a.eps);
#pragma unroll
    for (int k = 0; k < PAIRS; ++k) {
        const int i = 2 * (g0 + tid + 256 * k);
        h16x2 o;
        o[0] = (h16)(h[k][0] * rcp * (float)wgt[k][0]);
        o[1] = (h16)(h[k][1] * rcp * (float)wgt[k][1]);
        *reinterpret_cast<h16x2*>(a.out + i) = o;
    }
    if (!ok && tid == 0) tp_flag_error(own);
    __syncthreads();
    if (tid == 0) {      // the epoch advances when the LAST workgroup is done (a workgroup that starts late must still read the old one)
        const unsigned done = atomicAdd(reinterpret_cast<uint32_t*>(own) + 2, 1u) + 1u;
        if (done == gridDim.x) {
            reinterpret_cast<uint32_t*>(own)[2] = 0u;
            __hip_atomic_store(reinterpret_cast<uint32_t*>(own), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

}  // namespace cf
