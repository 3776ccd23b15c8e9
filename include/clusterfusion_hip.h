/*
 * clusterfusion_hip.h -- C-ABI of libclusterfusion_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for ONE hot path of xinhao-luo/ClusterFusion: the fused Llama
 * decoder-layer (attention block) decode op.  Every entry point below replaces one
 * binding of the reference's pybind module (citations: /root/reference/...):
 *
 *   cf_llama_decoder_layer               <- include/pybind.cpp:3-12,110
 *                                           (llama_decoder_layer_sm90,
 *                                            include/H100/llama/llama_kernel_dispatch.cu:4-146)
 *   cf_llama_decoder_layer_sglang        <- include/pybind.cpp:14-25,111
 *                                           (include/H100/llama/llama_kernel_sglang_dispatch.cu:4-151)
 *   cf_llama_decoder_layer_batch_decode_sglang
 *                                        <- include/pybind.cpp:27-43,112
 *                                           (include/H100/llama/llama_kernel_batch_sglang_dispatch.cu:6-111)
 *   cf_decoder_layer_ex                  -- superset used by the three above; adds what the
 *                                           reference hard-codes in include/H100/llama/config.h:2-11
 *                                           as run-time dims (GQA, head-parallel TP shards) and a
 *                                           KV page size > 1.
 *
 * Conventions
 *   - plain pointers and sizes only; all tensor pointers are DEVICE pointers on the current HIP
 *     device, fp16 unless noted, dense row-major, 16-byte aligned;
 *   - every call is asynchronous on `stream` (a hipStream_t; NULL = default stream); no device
 *     synchronisation, no allocation: the caller owns outputs and the workspace
 *     (cf_workspace_bytes);
 *   - return 0 on success, a negative CF_E* code otherwise; cf_last_error() gives the text
 *     (thread-local).  Nothing is launched when an error is returned.
 */
#ifndef CLUSTERFUSION_HIP_H
#define CLUSTERFUSION_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CF_ABI_VERSION 2      /* 2: cf_layer_args grew tp_areas / tp_rank / tp_world; cf_tp_gather, cf_rmsnorm_tp_gather */

enum cf_status {
    CF_OK = 0,
    CF_EINVAL = -1,      /* bad argument (NULL pointer, unsupported dims, ...) */
    CF_EWORKSPACE = -2,  /* workspace too small */
    CF_ELAUNCH = -3,     /* HIP reported a launch error */
    CF_EUNSUPPORTED = -4 /* shape outside the compiled specialisations */
};

/* weight orientation: the reference's plain entry takes [in,out] (chat/llama/model.py:317-322),
 * its sglang entries take torch-Linear [out,in] (tests/test_llama.py:124-125). */
enum cf_weight_layout { CF_W_OUT_IN = 0, CF_W_IN_OUT = 1 };
/* RoPE pairing: NEOX rotate-half (kernel_sglang.cuh:12,295-309) or GPT-J interleaved
 * (kernel.cuh:299-314). */
enum cf_rope_style { CF_ROPE_NEOX = 0, CF_ROPE_GPTJ = 1 };

/* Model dims of the (possibly TP-sharded) layer this rank computes.
 * Reference: fixed HIDDEN_DIM 4096 / HEAD_NUM 32 / HEAD_DIM 128 (config.h:2-11). */
typedef struct cf_dims {
    int32_t hidden;      /* model width (input of QKV proj, output of O proj) */
    int32_t n_q_heads;   /* query heads on this rank */
    int32_t n_kv_heads;  /* key/value heads on this rank (== n_q_heads for MHA) */
    int32_t head_dim;    /* must be 128 */
} cf_dims;

/* Superset argument block (cf_decoder_layer_ex). */
typedef struct cf_layer_args {
    cf_dims dims;
    int32_t batch;          /* rows of x */
    int32_t weight_layout;  /* enum cf_weight_layout */
    int32_t rope_style;     /* enum cf_rope_style */
    float eps;              /* RMSNorm epsilon */

    const void* x;          /* [batch, hidden] un-normalised hidden state */
    const void* residual;   /* [batch, hidden] or NULL; h = x + residual */
    const void* weight_qkv; /* OUT_IN: [(Hq+2Hkv)*d, hidden]; IN_OUT: [3*hidden, Hq*d] (MHA) */
    const void* weight_o;   /* OUT_IN: [hidden, Hq*d];         IN_OUT: [Hq*d, hidden] */
    const void* rms_weight; /* [hidden] */

    /* KV cache.  Contiguous mode (kv_indptr == NULL): k_cache/v_cache = [seq_len, Hkv*d].
     * Paged mode: k_cache/v_cache = slot arrays [num_slots, Hkv*d]; if kv_cache_ptrs_k/_v are
     * non-NULL they are DEVICE arrays of uint64 device pointers and the cache of this layer is
     * ptrs[layer_id] (kernel_batch_sglang.cuh:118-119). */
    const void* k_cache;
    const void* v_cache;
    const uint64_t* kv_cache_ptrs_k;
    const uint64_t* kv_cache_ptrs_v;
    int32_t layer_id;
    int32_t page_size;          /* 1 = reference semantics (indices are token slots) */
    int64_t seq_len;            /* contiguous mode: cached tokens (>= 0) */
    const int32_t* kv_indptr;   /* [batch+1] */
    const int32_t* kv_indices;  /* page ids / token slots */
    const int32_t* kv_seq_lens; /* [batch] cached tokens per row; required when page_size > 1;
                                   page_size == 1: NULL -> indptr[b+1]-1-indptr[b] (:120-122) */
    int64_t max_seq_len;        /* paged mode: a HINT for host-side planning (split count of the stage
                                   pipeline, routing of 2..4-row batches); 0 = unknown.  Never a
                                   correctness bound: every kernel reads the lengths on the device, a
                                   stale or too-small value cannot drop tokens */

    /* RoPE tables, fp32.  cos/sin point at the row for batch row 0; row b is at
     * + positions[b] * rope_row_stride floats when positions != NULL, else all rows share row 0.
     * NEOX reads head_dim/2 values, GPT-J head_dim (pair-duplicated). */
    const float* cos;
    const float* sin;
    const int64_t* positions;   /* [batch] or NULL */
    int64_t rope_row_stride;

    void* out;            /* [batch, hidden] attention-block output, NO residual add */
    void* residual_out;   /* [batch, hidden] fp16(x + residual); may alias `residual`; NULL ok */
    void* k_new;          /* [batch, Hkv, d] post-RoPE key of the new token, or NULL */
    void* v_new;          /* [batch, Hkv, d] or NULL */
    int32_t write_kv_to_cache; /* paged mode: also store k/v of the new token into its slot */

    void* workspace;
    size_t workspace_bytes;
    void* stream;         /* hipStream_t */

    /* Head-parallel TP with the all-reduce's publish folded into the layer (ABI 2; zero = off).  tp_world > 0: phase 3 of the
     * shard's persistent kernel writes its output ALSO as {epoch, fp16 x 2} granules into slot tp_rank of every rank's receive
     * area (tp_areas[0 .. tp_world): host array of device pointers as mapped into this process, cf_tp_area_*), i.e. the publish
     * half of cf_tp_oneshot_allreduce without a launch of its own and without re-reading `out`.  The call must be followed, on
     * the same stream, by exactly one gather on the same areas: cf_tp_gather or cf_rmsnorm_tp_gather.  That gather's n
     * (cf_tp_gather) / hidden (cf_rmsnorm_tp_gather) must equal dims.hidden, because the slot offsets inside an area depend on
     * n: a gather of another size returns CF_EINVAL and launches nothing.  Scope of that check: host side, per process, at CALL
     * time -- the layer call leaves a note under tp_areas[tp_rank], the next gather call on that area consumes it (match or
     * mismatch).  Calls made while a stream is capturing are checked like eager calls; replays of a captured graph make no calls
     * and are not checked.  Needs batch 1, hidden 4096 and a geometry with a persistent kernel (16 / 8 / 4 heads, 16q/4kv,
     * 8q/2kv, 4q/1kv, 32q/8kv); CF_EUNSUPPORTED otherwise. */
    void* const* tp_areas;
    int32_t tp_rank;
    int32_t tp_world;
} cf_layer_args;

int cf_abi_version(void);
const char* cf_last_error(void);

/* Upper bound of scratch bytes any call with these dims/batch needs (independent of seq_len);
 * 0 for dims the library does not support.
 * The workspace carries the persistent kernel's exchange state (epoch counter + tagged granules):
 * set it up ONCE with cf_workspace_init before its first use (zeroes it and records where the
 * kernels report failures), never write to it afterwards, and do not share one workspace between
 * streams that may run concurrently.
 *
 * Co-residency contract of the persistent kernels (the reference gets it from its cluster launch,
 * llama_kernel_dispatch.cu:123-127): their 256 workgroups must be resident together, one per CU.
 * The library checks the occupancy query once per device and kernel and takes the stage pipeline
 * when the device cannot hold them; while a persistent launch runs, no other stream or process may
 * hold CUs of the device.  If that is violated the launch does not hang and does not fail silently:
 * every wait is bounded (~0.5 s), the exchange that gave up is recorded -- in the workspace
 * (cf_workspace_status) and in a host-mapped word -- the epoch still advances (the workspace stays
 * usable), and the NEXT layer call of the process on that device returns CF_ELAUNCH once, launching
 * nothing: the outputs of the failed call are invalid. */
size_t cf_workspace_bytes(const cf_dims* dims, int32_t batch);
int cf_workspace_init(void* workspace, size_t workspace_bytes, void* stream);
/* Synchronises `stream` and reports the device-side error word of the workspace (0 = none).
 * Llama kernels: 1 = X1 (q|k|v gather), 2 = X2 (split records), 3 = X3 (attention output), 5 = X4
 * ([in,out] head sum) gave up after its bounded spin (4 is retired: page-table slices longer than
 * the part a workgroup stages in LDS are read through L2); 6 = X0 of the 5 .. 32-row kernel (the
 * normalised rows); 7 = a TP gather whose peer never published (cf_tp_gather, reported in the receive
 * area and in the sticky word, not here).
 * cf_deepseek_decoder_layer: 1..6 = its hand-offs in pipeline order.  The word is cleared by
 * cf_workspace_init only. */
int cf_workspace_status(const void* workspace, void* stream, uint32_t* error_code);
/* Synchronises `stream` and reports which arm of the persistent kernel the LAST completed call on this
 * workspace took -- chosen on the device from the cached length (kernel_batch_sglang.cuh:118-122 reads the
 * length there too), so one captured graph serves a growing sequence: 1 = two pre-requested tiles per
 * workgroup, straight-line (Llama-2-7B: 2048 < S <= 4096); 2 / 3 = one 128- / 256-token tile (S <= 1024 /
 * <= 2048); 4 = tile loop (longer).  0 = no persistent call completed yet (or a multi-row / MLA kernel). */
int cf_workspace_last_arm(const void* workspace, void* stream, uint32_t* arm);
/* Reads and clears the host-mapped failure word of the current device (0 = no persistent launch failed since the last
 * read); for callers that poll cf_workspace_status themselves and do not want the next layer call to report it again. */
uint32_t cf_take_sticky_error(void);

/* Algorithmic bytes one call must move (every weight / cached K,V byte once + vectors). */
uint64_t cf_algorithmic_bytes(const cf_dims* dims, int32_t batch, int64_t seq_len, int32_t has_residual);

int cf_decoder_layer_ex(const cf_layer_args* args);

/* replaces pybind `llama_decoder_layer(input, weight_qkv, weight_o, k_cache, v_cache,
 * rms_input_weight, cos, sin) -> (o, k, v)`  (include/pybind.cpp:3-12,110).
 * Llama-2-7B dims, [in,out] weights, GPT-J RoPE (cos/sin fp32[128]), eps = 1e-6
 * (kernel.cuh:58).  out [1,4096], k_new/v_new [1,32,128]. */
int cf_llama_decoder_layer(const void* input, const void* weight_qkv, const void* weight_o,
                           const void* k_cache, const void* v_cache, int64_t seq_len,
                           const void* rms_input_weight, const float* cos, const float* sin,
                           void* out, void* k_new, void* v_new,
                           void* workspace, size_t workspace_bytes, void* stream);

/* The same op for a caller that re-laid its weights out ONCE at load time (cf_relayout_weights: weight_qkv_out_in
 * [3*4096, 4096], weight_o_out_in [4096, 4096]): identical contract and results (GPT-J RoPE, eps 1e-6, no residual), but it
 * runs the kernel whose first phase streams whole 8-KB weight rows -- 29 instead of 32.5 us per layer at S = 1024, 35.4
 * instead of 40 at S = 4096 -- at the price of the caller keeping the weights in this orientation (no second copy if it
 * drops the [in,out] originals).  This is what a pybind maintainer binds behind `llama_decoder_layer`. */
int cf_llama_decoder_layer_out_in(const void* input, const void* weight_qkv_out_in, const void* weight_o_out_in,
                                  const void* k_cache, const void* v_cache, int64_t seq_len,
                                  const void* rms_input_weight, const float* cos, const float* sin,
                                  void* out, void* k_new, void* v_new,
                                  void* workspace, size_t workspace_bytes, void* stream);

/* replaces pybind `llama_decoder_layer_sglang(input, residual, weight_qkv, weight_o, k_cache,
 * v_cache, rms_input_weight, eps, cos, sin) -> (o, residual, k, v)` (include/pybind.cpp:14-25,111).
 * [out,in] weights, NEOX RoPE (first 64 floats of cos/sin are read), residual updated IN PLACE
 * (kernel_sglang.cuh:99-105). */
int cf_llama_decoder_layer_sglang(const void* input, void* residual, const void* weight_qkv,
                                  const void* weight_o, const void* k_cache, const void* v_cache,
                                  int64_t seq_len, const void* rms_input_weight, float eps,
                                  const float* cos, const float* sin,
                                  void* out, void* k_new, void* v_new,
                                  void* workspace, size_t workspace_bytes, void* stream);

/* replaces pybind `llama_decoder_layer_batch_decode_sglang(output, residual_output, input,
 * residual, weight_qkv, weight_o, paged_kv_indptr, paged_kv_indices, k_cache_ptrs, v_cache_ptrs,
 * layer_id, rms_input_weight, eps, positions, cos_sin) -> None` (include/pybind.cpp:27-43,112).
 * Token-granular page table (page size 1): the LAST index of each sequence is the slot the new
 * token's K/V are written to (kernel_batch_sglang.cuh:120-122,343-344);
 * cos_sin fp32 [max_pos,128], row = cat(cos[64], sin[64]) (:322-323). */
int cf_llama_decoder_layer_batch_decode_sglang(
    void* output, void* residual_output, const void* input, const void* residual,
    const void* weight_qkv, const void* weight_o, const int32_t* paged_kv_indptr,
    const int32_t* paged_kv_indices, const uint64_t* k_cache_ptrs, const uint64_t* v_cache_ptrs,
    int32_t layer_id, const void* rms_input_weight, float eps, const int64_t* positions,
    const float* cos_sin, int32_t batch, int64_t max_seq_len,
    void* workspace, size_t workspace_bytes, void* stream);

/* replaces pybind `rmsnorm(input, weight) -> out` (include/pybind.cpp:60-63,113; include/H100/norm/kernel.cuh:8-76;
 * tests/test_norm.py: input [64, 8192]).  out[r] = fp16(x[r] * rsqrt(mean(x[r]^2) + eps) * weight), fp32 math, one
 * rounding.  With `residual` != NULL it is the fused add + RMSNorm the decoder layers use between the attention
 * block and the FFN: x := input + residual, and fp16(x) is stored to `residual_out` when given (may alias
 * `residual` or `input`).  hidden: multiple of 8, <= 8192. */
int cf_rmsnorm(const void* input, const void* residual, const void* weight, float eps, int32_t rows, int32_t hidden,
               void* out, void* residual_out, void* stream);

/* Head-parallel TP at batch 1 (chat/llama/model.py:208-235: RowParallelLinear's all-reduce of the O projection): a ONE-SHOT
 * all-reduce of n fp16 values (n = hidden) over peer-mapped receive areas.  Every rank allocates one area of
 * cf_tp_oneshot_bytes(world, n) bytes (256-byte aligned, zeroed once), maps every peer's area into its process (hipIpc /
 * symmetric memory) and passes the `world` device pointers (areas[rank] = its own).  A call writes this rank's partial into slot
 * `rank` of EVERY area (remote traffic is write-only, one xGMI link latency, no dependent hops), polls its OWN area until
 * all slots carry this call's epoch, and sums them in rank order in fp32: the same bits on every rank.  The epoch lives in the
 * area and is advanced by the kernel: graph-capturable, no per-call memset.  flags bit 0 = publish only (test hook: a
 * virtual rank).  Word 1 of the area = error (7: a peer's slot never arrived within the bounded spin).  `out` may alias
 * `partial`.  Exercised on ONE GPU only (virtual ranks, and two processes sharing a device through the hipIpc path below);
 * N > 1 over xGMI is unmeasured. */
size_t cf_tp_oneshot_bytes(int32_t world, int32_t n);
/* Receive areas.  Peers on OTHER GPUs write into an area while its owner's kernel polls it, so it must be FINE-GRAINED device
 * memory (hipDeviceMallocFinegrained: coherent across agents during a kernel; ordinary hipMalloc / torch memory is coherent
 * only at kernel boundaries -- good enough for ranks sharing one GPU, not for xGMI).  cf_tp_area_alloc allocates one on the current
 * device, zeroed, 256-byte aligned; _export / _import move it between processes (hipIpc, 64-byte handle; the importer needs
 * peer access to the exporter's device: one xGMI hop); _unmap / _free undo them; _status copies the area's error word
 * (0 = fine, 7 = a peer's slot never arrived) after synchronising `stream`. */
#define CF_TP_HANDLE_BYTES 64
int cf_tp_area_alloc(size_t bytes, void** area);
int cf_tp_area_free(void* area);
int cf_tp_area_export(void* area, void* handle);
int cf_tp_area_import(const void* handle, void** mapped);
int cf_tp_area_unmap(void* mapped);
int cf_tp_area_status(const void* area, void* stream, uint32_t* code);
int cf_tp_oneshot_allreduce(const void* partial, void* out, int32_t n, int32_t rank, int32_t world, void* const* areas,
                            int32_t flags, void* stream);
/* The gather half alone, for partials published by the layer kernel itself (cf_layer_args.tp_areas): polls this rank's area
 * until the `world` slots carry the call's epoch, writes their fp32 sum in rank order to `out` (fp16; the same bits on every
 * rank and the same bits cf_tp_oneshot_allreduce produces) and advances the area's epoch.  A slot that never arrives within the
 * bounded spin: `out` is filled with NaN where it is missing, the area's error word becomes 7 and the device's sticky failure word
 * is raised (the next layer call returns CF_ELAUNCH) -- never a silently wrong sum. */
int cf_tp_gather(void* out, int32_t n, int32_t rank, int32_t world, void* const* areas, void* stream);
/* ... and the gather folded into the op that consumes the all-reduced attention output at batch 1: the fused add + RMSNorm
 * between the attention block and the FFN (cf_rmsnorm's residual form; chat/llama/model.py:492,519).  sum = fp16(sum over ranks
 * of the published partials) (-> sum_out when given), h = sum + residual (-> residual_out, fp16, when given; may alias
 * residual), out = fp16(h * rsqrt(mean(h^2) + eps) * weight).  One launch instead of gather + norm, no `out` round trip.
 * hidden: 512 .. 8192, multiple of 512.  Same failure behaviour as cf_tp_gather.  With more than one rank and hidden 4096 / 8192 the
 * op runs on eight workgroups (an eighth of the row each; the eight partial sums of squares meet through granules 16 .. 23 of the
 * area's header, in workgroup order on every rank): out differs from the one-workgroup form by <= 1 fp16 ulp, sum_out and
 * residual_out not at all.  The eight workgroups wait for each other (bounded): like every kernel of this library that does, they
 * must be able to run together -- eight workgroups of 256 threads without LDS. */
int cf_rmsnorm_tp_gather(void* const* areas, int32_t rank, int32_t world, const void* residual, const void* weight, float eps,
                         int32_t hidden, void* out, void* residual_out, void* sum_out, void* stream);
/* Clears the error word of an area (after the caller has dealt with a failed gather). */
int cf_tp_area_clear_error(void* area, void* stream);

/* replaces pybind `deepseek_decoder_layer(input, weight_q_nope, weight_q_pe, weight_uk, weight_kv_nope, weight_k_pe,
 * weight_uv, weight_o, ckv_cache, rms_input_weight, rms_ckv_weight, cos, sin) -> o`  (include/pybind.cpp:45-59,113;
 * include/H100/deepseek/deepseek_kernel_dispatch.cu:4-242, kernel.cuh:9-697).  DeepSeek-V2-Lite MLA dims
 * (config.h:2-9: hidden 2048, 16 heads, nope 128, rope 64, kv_lora 512), batch 1, all weights [in,out] fp16:
 *   weight_q_nope [2048, 16*128]  weight_q_pe [2048, 16*64]  weight_uk [128, 16*512]  weight_kv_nope [2048, 512]
 *   weight_k_pe [2048, 64]  weight_uv [512, 16*128]  weight_o [2048, 2048]  rms_* fp16  cos/sin fp32 [64]
 *   ckv_cache [seq_len, 576] = kv_lora latent | rope key.  As in the reference (kernel.cuh:470-473, SEQ_LEN fixed
 *   4096 there) the LAST row is the new token's slot: it is never read, the new token's normalised latent is
 *   attended in its place; seq_len >= 1.
 * out [1, 2048] fp16.  Extensions (0 / NULL = the reference's behaviour):
 *   rope_scores != 0  scores also include RoPE(q_pe) . k_pe (cache columns 512..575), i.e. complete MLA; the
 *                     reference computes RoPE(q_pe), RoPE(k_pe) but never uses them (kernel.cuh:298-315, 407-408);
 *   latent_out        [576] fp16 = RMSNorm(ckv) | RoPE(k_pe): the row a serving stack appends to the cache.
 * weight_q_pe, weight_k_pe, cos, sin may be NULL when neither extension is used (they cannot influence `out`).
 * Execution: one persistent launch (256 co-resident workgroups, needs the whole chip like the fused Llama kernel)
 * when cf_set_path is AUTO and seq_len <= 4096 or when it is FUSED; otherwise three launches (CF_PATH_PIPELINE).
 * workspace: cf_deepseek_workspace_bytes(), zeroed once with cf_workspace_init, one per concurrent stream;
 * cf_workspace_status reports its sticky error word (an in-kernel hand-off that gave up after its bounded spin). */
size_t cf_deepseek_workspace_bytes(void);
uint64_t cf_deepseek_algorithmic_bytes(int64_t seq_len, int32_t rope_scores);
int cf_deepseek_decoder_layer(const void* input, const void* weight_q_nope, const void* weight_q_pe,
                              const void* weight_uk, const void* weight_kv_nope, const void* weight_k_pe,
                              const void* weight_uv, const void* weight_o, const void* ckv_cache, int64_t seq_len,
                              const void* rms_input_weight, const void* rms_ckv_weight, const float* cos,
                              const float* sin, float eps, int32_t rope_scores, void* out, void* latent_out,
                              void* workspace, size_t workspace_bytes, void* stream);
/* per-stage hipEvent timing of cf_deepseek_decoder_layer on this thread (synchronises every call while on):
 * 0 = input projections + absorbed query (or the whole persistent launch), 1 = attention, 2 = W_uv + W_o. */
#define CF_MLA_STAGES 3
int cf_deepseek_profile_enable(int32_t on);
int cf_deepseek_profile_read(double* stage_ms /*[CF_MLA_STAGES]*/, int64_t* n_calls, int32_t reset);

/* Measurement hook (bench.py): when enabled, every subsequent cf_* layer call on this thread
 * records hipEvents around each of its kernels; cf_profile_read() synchronises and returns the
 * accumulated per-stage milliseconds and call count since the last reset.
 * stage order: 0 = QKV projection, 1 = attention (split-KV), 2 = O projection, 3 = reduce. */
#define CF_PROFILE_STAGES 4
int cf_profile_enable(int32_t on);
int cf_profile_read(double* stage_ms /*[CF_PROFILE_STAGES]*/, int64_t* n_calls, int32_t reset);

/* Tuning knobs (0 = default): KV splits per head; >0 forces that split count. */
int cf_set_tuning(int32_t kv_splits);
/* Execution path: 0 = auto (the persistent fused kernel when the shape qualifies:
 * hidden 4096, batch 1, >= 256 CUs and one of: 32 q = 32 kv heads (either weight layout); [out,in] weights
 * with 32 q / 8 kv heads or a 16 / 8 / 4-head shard; else the stage pipeline), 1 = always the
 * stage pipeline, 2 = require the fused kernel (CF_EUNSUPPORTED when the shape does not qualify). */
enum cf_path { CF_PATH_AUTO = 0, CF_PATH_PIPELINE = 1, CF_PATH_FUSED = 2 };
int cf_set_path(int32_t path);
/* Which path the last layer call on this thread took: CF_PATH_PIPELINE or CF_PATH_FUSED (0 = none yet). */
int cf_last_path(void);
/* ... and which kernel: e.g. "k_fused_decode_mha<IO=false>", "k_fused_decode_g<8, 4>" (kv heads, q heads per kv head),
 * "k_fused_decode_mhab<4>" or "stage pipeline".  Static string, valid forever.  (Which length arm of a persistent
 * kernel ran is decided on the device: cf_workspace_last_arm.) */
const char* cf_last_variant(void);
/* One-time weight re-layout for callers that hold the reference's plain orientation (`clusterfusion.llama_decoder_layer`:
 * weight_qkv = three [hidden, q_dim] matrices, weight_o = [q_dim, hidden]; /root/reference/chat/llama/model.py:317-322):
 * writes weight_qkv_out_in [3 * q_dim, hidden] and weight_o_out_in [hidden, q_dim], the orientation whose first phase
 * streams whole rows (Llama-2-7B: 29 instead of 33 us per layer at S = 1024, 35 instead of 40 at S = 4096).  Call it once per
 * layer at load time, then `cf_decoder_layer_ex` with CF_W_OUT_IN (and the entry's RoPE style).  Needs n_q_heads ==
 * n_kv_heads; hidden and q_dim multiples of 64.  Destinations must not alias the sources.  Asynchronous on `stream`. */
int cf_relayout_weights(const cf_dims* dims, const void* weight_qkv_in_out, const void* weight_o_in_out,
                        void* weight_qkv_out_in, void* weight_o_out_in, void* stream);
/* Debug: when non-NULL, the persistent kernel writes [256 workgroups][16] uint64 wall-clock stamps
 * (100 MHz s_memrealtime) at its phase boundaries into this device buffer. */
int cf_debug_set_trace(void* device_buffer);
/* Debug / experiment bits: 1 = heads interleaved over the XCDs, 2 / 4 = permute the block -> work map (timeline tool,
 * placement-independence tests); 16 = batch > 1 projections through the operand-layout kernel (A/B against the LDS one);
 * 32 = batches of 2 .. 4 sequences skip their persistent kernel (k_fused_decode_mhab) and take the five-launch MFMA path
 * (A/B and parity of that path at small batch); 64 = the persistent kernels stage only the first 512 page-table entries
 * of a workgroup's slice in LDS and read the rest through L2 (exercises the long-table path at test-sized sequences);
 * 128 = the 4-head shard (one rank of TP = 8) through the geometry-generic kernel k_fused_decode_g<4, 1> instead of the
 * role-split k_fused_decode_s<4>; 256 = the 8-head shard through k_fused_decode_s<8> instead of k_fused_decode_g<8, 1> (A/B);
 * 2048 = more than 32 rows through the chunked projection launches; 4096 = cf_rmsnorm_tp_gather on one workgroup at any world size. */
int cf_debug_set_flags(int32_t flags);
/* Test hook for the co-residency contract above: launches `blocks` workgroups (64 threads, `lds_bytes` of LDS each) that
 * hold their CUs for `microseconds` on `stream`. */
int cf_debug_occupy(void* stream, int32_t blocks, int32_t lds_bytes, int64_t microseconds);

#ifdef __cplusplus
}
#endif
#endif /* CLUSTERFUSION_HIP_H */
