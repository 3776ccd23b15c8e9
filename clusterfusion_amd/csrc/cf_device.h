// cf_device.h -- wave64 / gfx950 device helpers shared by the decode kernels.
//
// The reference's on-chip collective is Hopper's thread-block-cluster all-reduce over
// distributed shared memory (/root/reference/include/dsm.cuh:20-171).  CDNA4 has no clusters;
// the replacement is built from (a) DPP lane permutes inside a wavefront (no LDS traffic),
// (b) LDS staging across the wavefronts of a workgroup and (c) fp32 partial records in global
// memory handed from one kernel stage to the next (deterministic order, no atomics).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cf {

typedef _Float16 h16;
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
typedef h16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int WAVE = 64;
constexpr int HEAD_DIM = 128;          // the only head size of the path (config.h:11)
constexpr float NEG_BIG = -1.0e30f;    // finite "-inf" for online softmax (no NaN on empty splits)

// ---- 16-byte streaming loads ------------------------------------------------------------------
// Weights and cached K/V are read exactly once per call: non-temporal so they do not displace
// the small re-read vectors (x, partial records) from L2 (guide: nt-weights row).
// Every pointer on this path is a GLOBAL pointer; saying so matters when it was read from a
// device-side pointer table (kernel_batch_sglang.cuh:118-119 k_cache_ptrs[layer_id]): a generic
// pointer compiles to flat_load, which counts on BOTH the vector-memory and the LDS counter, so
// every LDS wait would also drain the prefetched K/V loads.
#define CF_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ h16x8 ld_stream(const h16* p) {
    return __builtin_nontemporal_load((const CF_GLOBAL h16x8*)p);
}
__device__ __forceinline__ h16x8 ld_h8(const h16* p) {
    return *(const CF_GLOBAL h16x8*)p;
}
__device__ __forceinline__ f32x4 ld_f4(const float* p) {
    return *(const CF_GLOBAL f32x4*)p;
}
__device__ __forceinline__ void st_h8(h16* p, h16x8 v) {
    *(CF_GLOBAL h16x8*)p = v;
}

// ---- DPP lane permutes ---------------------------------------------------------------------------
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
// sum over each aligned group of 16 lanes; every lane of the group receives the sum
__device__ __forceinline__ float sum16(float v) {
    v += dpp_f<0xB1>(v);    // quad_perm [1,0,3,2]  (lane ^ 1)
    v += dpp_f<0x4E>(v);    // quad_perm [2,3,0,1]  (lane ^ 2)
    v += dpp_f<0x141>(v);   // row_half_mirror      (adds the other quad of the 8)
    v += dpp_f<0x140>(v);   // row_mirror           (adds the other 8 of the 16)
    return v;
}
// sum over the 64 lanes; the result is valid in lane 63 only
__device__ __forceinline__ float sum64_lane63(float v) {
    v = sum16(v);
    v += dpp_f<0x142, 0xA>(v);   // row_bcast15 into rows 1,3
    v += dpp_f<0x143, 0xC>(v);   // row_bcast31 into rows 2,3
    return v;
}
// sum over the 64 lanes, wave-uniform result
__device__ __forceinline__ float sum64(float v) {
    v = sum64_lane63(v);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Cross-row exchanges on the VALU (gfx950 v_permlane16_swap / v_permlane32_swap) instead of
// ds_bpermute through the LDS crossbar.  swap(a = v, b = v) leaves a = v of the even row of each row
// pair (16) / of lanes 0..31 (32) and b = v of the odd row / of lanes 32..63, in every lane of the
// pair -- so a (+) b is the xor-16 / xor-32 butterfly step.  Inline asm on purpose: the ROCm 7.2
// builtin __builtin_amdgcn_permlane{16,32}_swap drops the second result (both results come back as
// the first; tools/ubench/permlane_test.hip checks both forms on the device).  The s_nop states are
// the VALU-write -> v_permlane read hazard (guide 5.7 item 2: nothing inside an asm string is padded).
#define CF_PERMLANE_SWAP(WIDTH, a, b) \
    asm volatile("s_nop 1\n\tv_permlane" #WIDTH "_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b))
__device__ __forceinline__ float xsum16(float v) {
    float a = v, b = v;
    CF_PERMLANE_SWAP(16, a, b);
    return a + b;
}
__device__ __forceinline__ float xsum32(float v) {
    float a = v, b = v;
    CF_PERMLANE_SWAP(32, a, b);
    return a + b;
}
// Sum of EIGHT values over the 4 rows (lanes l, l+16, l+32, l+48) in 6 swaps instead of 16: a swap of two DIFFERENT values
// reduces both at once (each ends up in one half / one row pair).  Result: r0 holds the sum of v[xrow_e(row)], r1 that of
// v[4 + xrow_e(row)], row = lane / 16, xrow_e = {0, 2, 1, 3}.
__device__ __forceinline__ int xrow_e(int row) { return ((row & 1) << 1) | (row >> 1); }
__device__ __forceinline__ void xsum_rows8(float (&v)[8], float& r0, float& r1) {
    asm volatile(
        "s_nop 1\n\t"
        "v_permlane32_swap_b32 %0, %1\n\t"
        "v_permlane32_swap_b32 %2, %3\n\t"
        "v_permlane32_swap_b32 %4, %5\n\t"
        "v_permlane32_swap_b32 %6, %7\n\t"
        "s_nop 1"
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
    float a = v[0] + v[1], b = v[2] + v[3], c = v[4] + v[5], d = v[6] + v[7];   // lanes 0-31: v0 / v2 / v4 / v6, lanes 32-63: v1 / v3 / v5 / v7
    asm volatile(
        "s_nop 1\n\t"
        "v_permlane16_swap_b32 %0, %1\n\t"
        "v_permlane16_swap_b32 %2, %3\n\t"
        "s_nop 1"
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    r0 = a + b;      // rows: v0, v2, v1, v3
    r1 = c + d;      // rows: v4, v6, v5, v7
}
__device__ __forceinline__ float xmax16(float v) {
    float a = v, b = v;
    CF_PERMLANE_SWAP(16, a, b);
    return fmaxf(a, b);
}
__device__ __forceinline__ float xmax32(float v) {
    float a = v, b = v;
    CF_PERMLANE_SWAP(32, a, b);
    return fmaxf(a, b);
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// 8 fp16 weights x 8 fp32 activations, fp32 accumulate (lowers to v_fma_mix_f32)
__device__ __forceinline__ float dot8(const h16x8 w, const float (&x)[8], float acc) {
#pragma unroll
    for (int e = 0; e < 8; ++e) acc = __builtin_fmaf((float)w[e], x[e], acc);
    return acc;
}

// 8 fp16 weights x 8 fp16 activations, fp32 accumulate: 4 x v_dot2_f32_f16 (two MACs per instruction)
__device__ __forceinline__ float dot8h(const h16x8 w, const h16x8 x, float acc) {
    acc = __builtin_amdgcn_fdot2(__builtin_shufflevector(w, w, 0, 1), __builtin_shufflevector(x, x, 0, 1), acc, false);
    acc = __builtin_amdgcn_fdot2(__builtin_shufflevector(w, w, 2, 3), __builtin_shufflevector(x, x, 2, 3), acc, false);
    acc = __builtin_amdgcn_fdot2(__builtin_shufflevector(w, w, 4, 5), __builtin_shufflevector(x, x, 4, 5), acc, false);
    acc = __builtin_amdgcn_fdot2(__builtin_shufflevector(w, w, 6, 7), __builtin_shufflevector(x, x, 6, 7), acc, false);
    return acc;
}

}  // namespace cf
