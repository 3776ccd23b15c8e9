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

constexpr int TP_MAX_WORLD = 8;
constexpr int TP_HDR_GRANULES = 32;      // 256-byte header in front of the slots

struct TpOneShotArgs {
    const h16* partial;                  // [n] this rank's partial
    h16* out;                            // [n] the sum (may alias partial)
    u64* areas[TP_MAX_WORLD];            // every rank's receive area as mapped into THIS process (areas[rank] = own)
    int n, rank, world;
    int flags;                           // bit 0: publish only (test hook: a virtual rank that does not gather)
};

__global__ __launch_bounds__(256) void k_tp_oneshot_allreduce(TpOneShotArgs a) {
    const int tid = threadIdx.x, g = blockIdx.x * 256 + tid, ng = a.n / 2;
    u64* own = a.areas[a.rank];
    const unsigned epoch = scalar_load(reinterpret_cast<const uint32_t*>(own)) + 1u;
    const bool live = g < ng;
    unsigned mine = 0;
    if (live) mine = *((const CF_GLOBAL unsigned*)(reinterpret_cast<const unsigned*>(a.partial) + g));
    const u64 gran = ((u64)epoch << 32) | mine;
    const size_t set = (size_t)(epoch & 1u) * a.world * ng;      // slot set of this call
    if (live) {
        for (int p = 0; p < a.world; ++p)      // remote write-only traffic: slot `rank` of every rank's area (own included)
            __hip_atomic_store(a.areas[p] + TP_HDR_GRANULES + set + (size_t)a.rank * ng + g, gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    float s0 = 0.f, s1 = 0.f;
    bool ok = true;
    if (live && !(a.flags & 1)) {
        for (int p = 0; p < a.world; ++p) {      // fixed order: the same bits on every rank
            const u64* src = own + TP_HDR_GRANULES + set + (size_t)p * ng + g;
            u64 x = 0;
            unsigned spin = 0;
            for (;; ++spin) {
                x = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if ((unsigned)(x >> 32) == epoch) break;
                if (spin > 4u * FUSED_SPIN_LIMIT) { ok = false; break; }      // bounded (~2 s): a lost peer must not hang the GPU
                __builtin_amdgcn_s_sleep(2);
            }
            const h16x2 v = __builtin_bit_cast(h16x2, (unsigned)x);
            s0 += (float)v[0];
            s1 += (float)v[1];
        }
        h16x2 r;
        r[0] = (h16)s0;
        r[1] = (h16)s1;
        reinterpret_cast<unsigned*>(a.out)[g] = __builtin_bit_cast(unsigned, r);
    }
    if (!ok) atomicCAS(reinterpret_cast<uint32_t*>(own) + 1, 0u, 7u);      // error word of the area
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

}  // namespace cf
