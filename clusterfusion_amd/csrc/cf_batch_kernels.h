// cf_batch_kernels.h -- batch > 1: the two projections of the decode path as weight-streaming GEMMs on the
// matrix cores (gfx950 v_mfma_f32_16x16x32_f16).
//
// With one sequence the projections are GEMVs (1 flop per weight byte, VALU, cf_decode_kernels.h /
// cf_fused_kernel*.h).  With B sequences (llama_decoder_layer_batch_decode_sglang,
// kernel_batch_sglang.cuh:63 batch_id = cluster_rank / 32: the reference simply runs its whole GEMV kernel
// once per sequence and re-reads every weight B times) they are [B, K] x [K, N] GEMMs: every weight byte
// is streamed from HBM ONCE and multiplied against all B activation rows; at B = 16 that is 16 MACs per
// fp16 weight, which the VALU cannot do at HBM speed and one MFMA per 1-KB load does for free.
//
//   k_norm_rows        xn[b][:] = fp16(RMSNorm(x[b] + residual[b]) * w)      (kernel.cuh:95-139; rounded once)
//   k_proj_rows_mfma   Y[b][n] = sum_k A[b][k] * W[n][k]     A fp16 [B, K]; W fp16 [N, K] ([out,in]),
//                                                             N % 16 == 0, K % 256 == 0
//
// Mapping of the GEMM (no LDS staging of either operand):
//   * workgroup = 8 wavefronts, one per CU; row tile = 16 weight rows; tile t is served by workgroup
//     t % gridDim.x; wavefront w owns the K-slice [w K/8, (w+1) K/8) of every tile of its workgroup
//     (split-K inside the workgroup, fixed-order sum of the 8 partial 16x16 blocks through LDS);
//   * MFMA 16x16x32: lane l holds, for row m = l % 16 (weights) / batch row n = l % 16 (activations), the 8
//     consecutive k of group l / 16 -- ONE 16-byte load per lane and k-block for either operand, straight
//     from global memory in operand layout (a wavefront instruction reads 64 B of each of 16 rows);
//   * the activation operand of a wavefront's K-slice (B x K/8 values) lives in registers for the whole
//     kernel and is reused by every row tile; batch rows beyond B are zero; 16 rows per batch tile, BT tiles;
//   * weight rows are requested DEPTH tiles ahead (2 when the registers allow), so the LDS reduction of one
//     tile overlaps the stream of the next; requests past the last tile read one dummy line.
#pragma once
#include "cf_decode_kernels.h"

namespace cf {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

// barrier that orders LDS only: the weight rows requested ahead stay in flight (__syncthreads would drain them)
__device__ __forceinline__ void lds_only_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// fused add + RMSNorm of every batch row, rounded once to fp16 (the reference keeps its normalised
// activations in fp16 shared memory, kernel.cuh:132-139).  One workgroup per row; K <= 8192.
__global__ __launch_bounds__(256) void k_norm_rows(NormArgs na, h16* xn_out, h16* res_out /* fp16(x + residual) or null; may alias */) {
    __shared__ float s_ss[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    const int K = na.hidden;
    const h16* x = na.x + (size_t)b * K;
    const h16* r = na.residual ? na.residual + (size_t)b * K : x;
    const float rs = na.residual ? 1.f : 0.f;
    float h[4][8];
    h16x8 wv[4];          // the weight is requested with x and the residual: one global round trip, not two
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int i = (c * 256 + tid) * 8;
        if (i < K) {
            const h16x8 xv = ld_h8(x + i), rv = ld_h8(r + i);
            wv[c] = ld_h8(na.rms_w + i);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                h[c][e] = __builtin_fmaf(rs, (float)rv[e], (float)xv[e]);
                ss = __builtin_fmaf(h[c][e], h[c][e], ss);
            }
        }
    }
    ss = sum64_lane63(ss);
    if (lane == 63) s_ss[wave] = ss;
    __syncthreads();
    const float rcp = __builtin_amdgcn_rsqf((s_ss[0] + s_ss[1] + s_ss[2] + s_ss[3]) / (float)K + na.eps);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int i = (c * 256 + tid) * 8;
        if (i < K) {
            h16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (h16)(h[c][e] * rcp * (float)wv[c][e]);
            if (res_out) {   // (every element is read and written by the same thread: in-place is safe)
                h16x8 ho;
#pragma unroll
                for (int e = 0; e < 8; ++e) ho[e] = (h16)h[c][e];
                st_h8(res_out + (size_t)b * K + i, ho);
            }
            st_h8(xn_out + (size_t)b * K + i, o);
        }
    }
}

struct ProjArgs {
    const h16* W;        // [n_rows, K]
    int n_rows, K, batch;
    const h16* in;       // [batch, K] fp16 activations
    float* out_f32;      // [batch, n_rows]   (one of the two)
    h16* out_h16;        // [batch, n_rows]
};

// NB = k-blocks (of 32) per wavefront slice = K / 256; BT = batch tiles of 16 rows; DEPTH = tiles in flight
template <int NB, int BT, int DEPTH>
__global__ __launch_bounds__(512, 2) void k_proj_rows_mfma(ProjArgs a, ResidualOut ro) {
    __shared__ __attribute__((aligned(16))) float s_part[8][BT][256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const int K = a.K, ks0 = wave * (K / 8) + kq * 8;   // this lane's first k inside a row
    const int ntiles = a.n_rows / 16, G = gridDim.x;

    // ---- the weight stream starts first: DEPTH row tiles of this workgroup ---------------------------------
    h16x8 wt[DEPTH][NB];
    auto load_tile = [&](h16x8 (&t)[NB], int tile) {
        const bool live = tile < ntiles;                                  // workgroup-uniform
        const h16* p = live ? a.W + (size_t)(16 * tile + r16) * K + ks0 : a.W;   // past the end: one dummy line
        const int js = live ? 32 : 0;
#pragma unroll
        for (int j = 0; j < NB; ++j) t[j] = ld_stream(p + j * js);
    };
#pragma unroll
    for (int dd = 0; dd < DEPTH; ++dd) load_tile(wt[dd], blockIdx.x + dd * G);

    // ---- activation operand: bx[bt][j] = A[16 bt + r16][ks0 + 32 j .. + 8), zero rows beyond the batch ------
    h16x8 bx[BT][NB];
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) {
        const int n = 16 * bt + r16;
        const bool live = n < a.batch;
        const h16* ip = a.in + (size_t)(live ? n : 0) * K + ks0;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const h16x8 v = ld_h8(ip + 32 * j);
#pragma unroll
            for (int e = 0; e < 8; ++e) bx[bt][j][e] = live ? v[e] : (h16)0.f;
        }
    }

    // ---- row tiles: one MFMA per 1-KB load; the 8 K-slices meet in LDS ---------------------------------------
    for (int t0 = blockIdx.x; t0 < ntiles; t0 += DEPTH * G) {
#pragma unroll
        for (int dd = 0; dd < DEPTH; ++dd) {
            const int tile = t0 + dd * G;
            if (tile < ntiles) {                 // (workgroup-uniform; a slot past the last tile does no arithmetic)
                f32x4_t d[BT];
#pragma unroll
                for (int bt = 0; bt < BT; ++bt) d[bt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int bt = 0; bt < BT; ++bt)
                        d[bt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wt[dd][j], bx[bt][j], d[bt], 0, 0, 0);
                // D: lane l holds rows m = 4 (l / 16) + i, batch column n = l % 16
#pragma unroll
                for (int bt = 0; bt < BT; ++bt) *reinterpret_cast<f32x4_t*>(&s_part[wave][bt][lane * 4]) = d[bt];
            }
            load_tile(wt[dd], tile + DEPTH * G);
            if (tile < ntiles) {
                lds_only_barrier();
                for (int o = tid; o < BT * 256; o += 512) {
                    const int bt = o >> 8, l = (o >> 2) & 63, i = o & 3;
                    float v = 0.f;
#pragma unroll
                    for (int w = 0; w < 8; ++w) v += s_part[w][bt][o & 255];      // fixed order
                    const int n = 16 * bt + (l & 15), m = 4 * (l >> 4) + i;
                    if (n < a.batch) {
                        const size_t at = (size_t)n * a.n_rows + 16 * tile + m;
                        if (a.out_f32) a.out_f32[at] = v;
                        else a.out_h16[at] = (h16)v;
                    }
                }
                lds_only_barrier();
            }
        }
    }
    // last stage only (every reader of `residual` is done): workgroup b writes row b of fp16(x + residual)
    if (ro.residual_out) {
        for (int b = blockIdx.x; b < a.batch; b += G) write_residual(ro, b);
    }
}

// Same GEMM, weight rows streamed as whole 1-KB pieces.  k_proj_rows_mfma above loads the weights straight in MFMA
// operand layout, i.e. sixteen 64-byte row pieces per wavefront instruction; that pattern streams at 3.7 TB/s here
// against ~6 TB/s for the GEMV kernels' 1-KB-contiguous instructions.  This variant requests ONE row's 1-KB slice per
// instruction (lane l: 16 bytes at 16 l) and turns the 16 rows of a tile into operand layout through a
// wavefront-private LDS image: row stride 1040 bytes, so that the sixteen 16-byte bank groups are each hit by
// exactly four lanes of every operand read (row r, k-group q -> group (r + q + 4 j) mod 16), writes are contiguous.
// DEPTH tiles in flight in registers while the previous one is multiplied out of LDS; K-split over the 8 wavefronts
// and the fixed-order reduction are unchanged.  K must be 4096 (NB = 16: 512 columns = 1 KB per wavefront).
constexpr int PROJ_LDS_ROW = 520;                                    // halves per image row (512 + 8)
constexpr int PROJ_LDS_WAVE = 16 * PROJ_LDS_ROW;                     // halves per wavefront image
template <int BT>
constexpr int proj_lds_bytes() { return 8 * PROJ_LDS_WAVE * 2 + 8 * BT * 256 * 4; }

template <int BT, int DEPTH>
__global__ __launch_bounds__(512) void k_proj_rows_lds(ProjArgs a, ResidualOut ro) {
    constexpr int NB = 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_p[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    h16* s_img = reinterpret_cast<h16*>(smem_p) + wave * PROJ_LDS_WAVE;
    float (*s_part)[BT][256] = reinterpret_cast<float (*)[BT][256]>(smem_p + 8 * PROJ_LDS_WAVE * 2);
    const int r16 = lane & 15, kq = lane >> 4;
    const int K = a.K, kw = wave * (K / 8);                            // this wavefront's first column
    const int ntiles = a.n_rows / 16, G = gridDim.x;

    // ---- the weight stream starts first: one row slice per instruction, DEPTH tiles ahead ------------------------------
    h16x8 wt[DEPTH][16];
    auto load_tile = [&](h16x8 (&t)[16], int tile) {
        const bool live = tile < ntiles;                              // workgroup-uniform
        const h16* p = live ? a.W + (size_t)(16 * tile) * K + kw + lane * 8 : a.W + lane * 8;   // past the end: one dummy row
        const size_t rs = live ? (size_t)K : 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) t[i] = ld_stream(p + i * rs);
    };
#pragma unroll
    for (int dd = 0; dd < DEPTH; ++dd) load_tile(wt[dd], blockIdx.x + dd * G);

    // ---- activation operand: bx[bt][j] = A[16 bt + r16][kw + 32 j + 8 kq .. + 8), zero rows beyond the batch -----------
    h16x8 bx[BT][NB];
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) {
        const int n = 16 * bt + r16;
        const bool live = n < a.batch;
        const h16* ip = a.in + (size_t)(live ? n : 0) * K + kw + kq * 8;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const h16x8 v = ld_h8(ip + 32 * j);
#pragma unroll
            for (int e = 0; e < 8; ++e) bx[bt][j][e] = live ? v[e] : (h16)0.f;
        }
    }

    for (int t0 = blockIdx.x; t0 < ntiles; t0 += DEPTH * G) {
#pragma unroll
        for (int dd = 0; dd < DEPTH; ++dd) {
            const int tile = t0 + dd * G;
            if (tile >= ntiles) break;                                // (workgroup-uniform)
            // rows -> the wavefront's LDS image (contiguous 1-KB writes), registers free for the tile DEPTH ahead
#pragma unroll
            for (int i = 0; i < 16; ++i) *reinterpret_cast<h16x8*>(s_img + i * PROJ_LDS_ROW + lane * 8) = wt[dd][i];
            load_tile(wt[dd], tile + DEPTH * G);
            f32x4_t d[BT];
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) d[bt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const h16x8 av = *reinterpret_cast<const h16x8*>(s_img + r16 * PROJ_LDS_ROW + 32 * j + 8 * kq);
#pragma unroll
                for (int bt = 0; bt < BT; ++bt) d[bt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bx[bt][j], d[bt], 0, 0, 0);
            }
            // D: lane l holds rows m = 4 (l / 16) + i, batch column n = l % 16
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) *reinterpret_cast<f32x4_t*>(&s_part[wave][bt][lane * 4]) = d[bt];
            lds_only_barrier();
            for (int o = tid; o < BT * 256; o += 512) {
                const int bt = o >> 8, l = (o >> 2) & 63, i = o & 3;
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) v += s_part[w][bt][o & 255];      // fixed order
                const int n = 16 * bt + (l & 15), m = 4 * (l >> 4) + i;
                if (n < a.batch) {
                    const size_t at = (size_t)n * a.n_rows + 16 * tile + m;
                    if (a.out_f32) a.out_f32[at] = v;
                    else a.out_h16[at] = (h16)v;
                }
            }
            lds_only_barrier();
        }
    }
    // last stage only (every reader of `residual` is done): workgroup b writes row b of fp16(x + residual)
    if (ro.residual_out) {
        for (int b = blockIdx.x; b < a.batch; b += G) write_residual(ro, b);
    }
}

// More than 32 batch rows (round 4).  The two kernels above keep the activation operand of a wavefront's K-slice in registers
// (64 VGPRs per 16-row batch tile: two tiles at most), so the host ran a batch of B rows as ceil(B / 32) launches and every one
// of them streamed all the weights again: 60 / 112 us for the QKV projection at 64 / 128 rows where the bytes need 17.  Here a
// workgroup owns MT 16-row tiles of W (48 rows of Wqkv: 256 workgroups) and ALL batch rows (<= 128): K runs in chunks of 256
// columns, both operands go through LDS -- the weight chunk (MT x 16 rows x 512 B, from HBM, requested two chunks ahead through
// registers) and the activation chunk (128 rows x 512 B, L2-resident: every workgroup reads the same 1 MB) -- and wavefront w
// takes batch tile w % 4 of every group of 64 batch rows (one or two groups) and half w / 4 of the chunk's k-steps: a weight
// operand read from LDS serves both groups (5 operand reads per 6 MFMAs); the two halves' accumulators meet in LDS in fixed order.  Every weight byte crosses the CU once per launch
// whatever the batch.  Image rows are 528 bytes apart (a 16-byte operand read of row r, k-group q lands in bank group
// (r + q + 4 s) mod 16: four lanes per group, no conflict beyond the 4 passes a 1-KB read takes anyway).
constexpr int BIG_MAX_ROWS = 128;                        // batch rows per launch
#ifndef CF_BIG_KC1
#define CF_BIG_KC1 256                                   // columns per chunk with one group of batch rows (512 fits LDS there: QKV +3 us at 33 / 64 rows)
#endif
template <int NG>
constexpr int big_kc() { return NG == 1 ? CF_BIG_KC1 : 256; }
template <int MT, int NG>
constexpr int proj_big_lds_bytes() { return (16 * MT + 64 * NG) * (big_kc<NG>() + 8) * 2; }

template <int MT, int NG>      // MT row tiles per workgroup; NG groups of 64 batch rows (1: <= 64 rows, 2: <= 128)
__global__ __launch_bounds__(512) void k_proj_rows_big(ProjArgs a, ResidualOut ro) {
    constexpr int BIG_KC = big_kc<NG>(), BIG_ROW = BIG_KC + 8;       // columns per chunk; halves per LDS image row (+16 bytes)
    constexpr int PPR = BIG_KC / 8, PSH = PPR == 32 ? 5 : 6;         // 16-byte pieces per image row
    constexpr int NS = BIG_KC / 32 / 2;                              // k-steps of a chunk per wavefront (the chunk's two halves: w / 4)
    constexpr int AP = MT * 16 * PPR / 512, BP = NG * 64 * PPR / 512;   // 16-byte pieces per thread and chunk: weights, activations
    static_assert(MT * 16 * PPR % 512 == 0 && (PPR == 32 || PPR == 64), "whole pieces per thread");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    h16* s_w = reinterpret_cast<h16*>(smem_b);                       // [16 MT][BIG_ROW]
    h16* s_x = s_w + 16 * MT * BIG_ROW;                              // [64 NG][BIG_ROW]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kq = lane >> 4;
    const int K = a.K, nchunk = K / BIG_KC;
    const int row0 = blockIdx.x * 16 * MT;
    const int bt = wave & 3, kh = wave >> 2;      // batch tile (of every group) and k-half of this wavefront

    // piece p of a chunk: image row p / PPR, 16-byte column p % PPR
    h16x8 wa[2][AP], xb[BP];
    auto load_w = [&](h16x8 (&t)[AP], int c) {
        const bool live = c < nchunk;                                 // (uniform) past the end: one dummy line
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int p = tid + 512 * i;
            t[i] = ld_stream(live ? a.W + (size_t)(row0 + (p >> PSH)) * K + c * BIG_KC + (p & (PPR - 1)) * 8 : a.W);
        }
    };
    auto load_x = [&](int c) {
        const bool live = c < nchunk;
#pragma unroll
        for (int i = 0; i < BP; ++i) {
            const int p = tid + 512 * i, n = p >> PSH;
            const h16x8 v = ld_h8(live && n < a.batch ? a.in + (size_t)n * K + c * BIG_KC + (p & (PPR - 1)) * 8 : a.in);
#pragma unroll
            for (int e = 0; e < 8; ++e) xb[i][e] = n < a.batch ? v[e] : (h16)0.f;
        }
    };
    load_w(wa[0], 0);
    load_x(0);
    load_w(wa[1], 1);
    f32x4_t d[NG][MT];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int m = 0; m < MT; ++m) d[g][m] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    auto chunk = [&](h16x8 (&cur)[AP], int c) {
        // registers -> images (the previous chunk's readers are behind the barrier at the end of the last round)
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int p = tid + 512 * i;
            *reinterpret_cast<h16x8*>(s_w + (p >> PSH) * BIG_ROW + (p & (PPR - 1)) * 8) = cur[i];
        }
#pragma unroll
        for (int i = 0; i < BP; ++i) {
            const int p = tid + 512 * i;
            *reinterpret_cast<h16x8*>(s_x + (p >> PSH) * BIG_ROW + (p & (PPR - 1)) * 8) = xb[i];
        }
        load_x(c + 1);              // (L2-resident; needed at the top of the next round)
        load_w(cur, c + 2);         // the weight stream: two chunks ahead (four chunks ahead for the weights and two for the
                                    //  activations measured the same up to 64 rows and spilled at 128: 435 -> 470 us per call)
        lds_only_barrier();
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int k0 = 32 * (kh * NS + s) + 8 * kq;
            h16x8 av[MT];           // a weight operand read serves the batch tiles of both groups
#pragma unroll
            for (int m = 0; m < MT; ++m) av[m] = *reinterpret_cast<const h16x8*>(s_w + (16 * m + r16) * BIG_ROW + k0);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const h16x8 bv = *reinterpret_cast<const h16x8*>(s_x + (64 * g + 16 * bt + r16) * BIG_ROW + k0);
#pragma unroll
                for (int m = 0; m < MT; ++m) d[g][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[m], bv, d[g][m], 0, 0, 0);
            }
        }
        lds_only_barrier();
    };
    for (int c = 0; c < nchunk; c += 2) {      // (an even number of chunks: K % 1024 == 0 is checked by the host)
        chunk(wa[0], c);
        chunk(wa[1], c + 1);
    }

    // D: lane l holds weight rows m = 4 (l / 16) + i of the tile, batch row n = 64 g + 16 bt + l % 16.  The upper k-half's
    // accumulators go through LDS and are added second (fixed order).
    float* s_red = reinterpret_cast<float*>(smem_b);      // [4][NG][MT][256] (the images are dead)
    if (kh == 1) {
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int m = 0; m < MT; ++m) *reinterpret_cast<f32x4_t*>(&s_red[(((bt * NG + g) * MT + m) * 64 + lane) * 4]) = d[g][m];
    }
    lds_only_barrier();
    if (kh == 0) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int n = 64 * g + 16 * bt + r16;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const f32x4_t v = d[g][m] + *reinterpret_cast<const f32x4_t*>(&s_red[(((bt * NG + g) * MT + m) * 64 + lane) * 4]);
                if (n < a.batch) {
                    const size_t at = (size_t)n * a.n_rows + row0 + 16 * m + 4 * kq;
                    if (a.out_f32) {
                        *reinterpret_cast<f32x4_t*>(a.out_f32 + at) = v;
                    } else {
                        typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
                        h16x4 o;
#pragma unroll
                        for (int i = 0; i < 4; ++i) o[i] = (h16)v[i];
                        *reinterpret_cast<h16x4*>(a.out_h16 + at) = o;
                    }
                }
            }
        }
    }
    // last stage only (every reader of `residual` is done): workgroup b writes row b of fp16(x + residual)
    if (ro.residual_out) {
        for (int b = blockIdx.x; b < a.batch; b += gridDim.x) write_residual(ro, b);
    }
}

}  // namespace cf
