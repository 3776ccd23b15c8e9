// cf_mla_api.hip -- C-ABI of the DeepSeek MLA decoder-layer op (include/clusterfusion_hip.h).
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cmath>
#include "clusterfusion_hip.h"
#include "cf_mla_kernels.h"
#include "cf_mla_fused.h"

namespace cf {
int api_fail(int code, const char* fmt, ...);      // cf_api.hip: sets the thread-local error text
int api_path();                                    // cf_set_path
void* api_trace();                                 // cf_debug_set_trace
void api_set_last_path(int p);                     // cf_last_path
int api_fail_sticky(uint32_t code);
uint32_t api_take_sticky_error();                  // host-mapped failure word of the persistent kernels
}

namespace {

using cf::h16;

inline size_t align256(size_t v) { return (v + 255) & ~size_t(255); }

struct MlaWorkspace {
    unsigned int* state;      // [0] epoch, [1] error   (first 256 B; zeroed once by cf_workspace_init; same layout
                              //                         as the Llama workspace, so cf_workspace_status reads it)
    unsigned long long* g_a;  // [8][3648] granules
    unsigned long long* g_d;  // [4][2048] granules
    unsigned long long* g_e;  // [8][2048] granules
    h16* qlat;                // [16][576]
    h16* latent_new;          // [576]
    float* part_o;            // [256][16][512]   (three-launch path)
    float* part_ml;           // [256][16][2]
    unsigned long long* g_q;  // [16][576] + [576] granules   (single-launch path)
    unsigned long long* g_po; // [256][16][512] granules
    unsigned long long* g_ml; // [256][32] granules
    h16* zeros;               // [576] never written after cf_workspace_init
    unsigned long long* g_xcc; // [256] granules
    size_t total;
};

MlaWorkspace carve(void* base) {
    MlaWorkspace w;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t at = off; off += align256(bytes); return reinterpret_cast<char*>(base) + at; };
    w.state = reinterpret_cast<unsigned int*>(take(256));
    w.g_a = reinterpret_cast<unsigned long long*>(take(8 * cf::MLA_A_KS * cf::MLA_A_COLS));
    w.g_d = reinterpret_cast<unsigned long long*>(take(8 * cf::MLA_D_KS * cf::MLA_HID));
    w.g_e = reinterpret_cast<unsigned long long*>(take(8 * cf::MLA_E_KS * cf::MLA_HID));
    w.qlat = reinterpret_cast<h16*>(take(sizeof(h16) * cf::MLA_H * cf::MLA_LAT));
    w.latent_new = reinterpret_cast<h16*>(take(sizeof(h16) * cf::MLA_LAT));
    w.part_o = reinterpret_cast<float*>(take(sizeof(float) * cf::MLA_NSPLIT_MAX * cf::MLA_H * cf::MLA_L));
    w.part_ml = reinterpret_cast<float*>(take(sizeof(float) * cf::MLA_NSPLIT_MAX * cf::MLA_H * 2));
    w.g_q = reinterpret_cast<unsigned long long*>(take(8 * (cf::MLA_H + 1) * cf::MLA_LAT));
    w.g_po = reinterpret_cast<unsigned long long*>(take(8 * (size_t)cf::MLA_NSPLIT_MAX * cf::MLA_H * cf::MLA_L));
    w.g_ml = reinterpret_cast<unsigned long long*>(take(8 * cf::MLA_NSPLIT_MAX * 32));
    w.zeros = reinterpret_cast<h16*>(take(sizeof(h16) * cf::MLA_LAT));
    w.g_xcc = reinterpret_cast<unsigned long long*>(take(8 * 256));
    w.total = off;
    return w;
}

constexpr int ATTN_LDS = 4 * 16384 + 2048 + 3 * 64 * 4;

thread_local bool g_mla_prof = false;
thread_local double g_mla_ms[CF_MLA_STAGES] = {0, 0, 0};
thread_local int64_t g_mla_calls = 0;

}  // namespace

extern "C" {

size_t cf_deepseek_workspace_bytes(void) { return carve(nullptr).total; }

uint64_t cf_deepseek_algorithmic_bytes(int64_t seq_len, int32_t rope_scores) {
    using namespace cf;
    uint64_t w = (uint64_t)MLA_HID * MLA_H * MLA_NOPE + (uint64_t)MLA_NOPE * MLA_H * MLA_L + (uint64_t)MLA_HID * MLA_L +
                 (uint64_t)MLA_L * MLA_H * MLA_NOPE + (uint64_t)MLA_H * MLA_NOPE * MLA_HID;
    if (rope_scores) w += (uint64_t)MLA_HID * MLA_H * MLA_ROPE + (uint64_t)MLA_HID * MLA_ROPE;
    const uint64_t cache = (uint64_t)(seq_len > 0 ? seq_len - 1 : 0) * (rope_scores ? MLA_LAT : MLA_L);
    const uint64_t vec = 3 * (uint64_t)MLA_HID + MLA_L;
    return 2 * (w + cache + vec) + 4 * 2 * MLA_ROPE;
}

int cf_deepseek_profile_enable(int32_t on) {
    g_mla_prof = on != 0;
    return CF_OK;
}

int cf_deepseek_profile_read(double* stage_ms, int64_t* n_calls, int32_t reset) {
    if (!stage_ms || !n_calls) return cf::api_fail(CF_EINVAL, "NULL argument");
    for (int i = 0; i < CF_MLA_STAGES; ++i) stage_ms[i] = g_mla_ms[i];
    *n_calls = g_mla_calls;
    if (reset) {
        for (int i = 0; i < CF_MLA_STAGES; ++i) g_mla_ms[i] = 0;
        g_mla_calls = 0;
    }
    return CF_OK;
}

int cf_deepseek_decoder_layer(const void* input, const void* weight_q_nope, const void* weight_q_pe,
                              const void* weight_uk, const void* weight_kv_nope, const void* weight_k_pe,
                              const void* weight_uv, const void* weight_o, const void* ckv_cache, int64_t seq_len,
                              const void* rms_input_weight, const void* rms_ckv_weight, const float* cos,
                              const float* sin, float eps, int32_t rope_scores, void* out, void* latent_out,
                              void* workspace, size_t workspace_bytes, void* stream) {
    using namespace cf;
    if (const uint32_t code = api_take_sticky_error()) return api_fail_sticky(code);
    if (!input || !weight_q_nope || !weight_uk || !weight_kv_nope || !weight_uv || !weight_o || !rms_input_weight ||
        !rms_ckv_weight || !out)
        return api_fail(CF_EINVAL, "NULL tensor argument");
    const bool with_pe = rope_scores != 0 || latent_out != nullptr;
    if (with_pe && (!weight_q_pe || !weight_k_pe || !cos || !sin))
        return api_fail(CF_EINVAL, "weight_q_pe / weight_k_pe / cos / sin are required when rope_scores or latent_out is set");
    if (seq_len < 1) return api_fail(CF_EINVAL, "seq_len %lld: ckv_cache needs at least the new token's row", (long long)seq_len);
    if (seq_len > 1 && !ckv_cache) return api_fail(CF_EINVAL, "NULL ckv_cache");
    if (seq_len > (int64_t)1 << 30) return api_fail(CF_EUNSUPPORTED, "seq_len %lld too large", (long long)seq_len);
    if (!(eps > 0.f)) return api_fail(CF_EINVAL, "eps must be positive");
    if (!workspace) return api_fail(CF_EINVAL, "NULL workspace");
    const MlaWorkspace w = carve(workspace);
    if (workspace_bytes < w.total)
        return api_fail(CF_EWORKSPACE, "workspace %zu B < %zu B (cf_deepseek_workspace_bytes)", workspace_bytes, w.total);
    hipStream_t st = static_cast<hipStream_t>(stream);

    {   // > 64 KB of LDS for the attention kernel: opt in once per device
        static thread_local unsigned long long attr_devs = 0;
        int dev = 0;
        hipGetDevice(&dev);
        if (dev < 64 && !((attr_devs >> dev) & 1ull)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mla_attn<false>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, ATTN_LDS);
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mla_attn<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, ATTN_LDS);
            if (e != hipSuccess) return api_fail(CF_ELAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
            attr_devs |= 1ull << dev;
        }
    }

    // ---- single persistent launch when the whole chip is there (256 co-resident workgroups) ----------------------
    int cus = 0;
    {
        static thread_local int cached_dev = -1, cached_cus = 0;
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev != cached_dev) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, dev) == hipSuccess) { cached_dev = dev; cached_cus = prop.multiProcessorCount; }
        }
        cus = cached_cus;
    }
    const int path = api_path();
    // auto: the single launch wins while the cache is short (measured: 20.2 vs 22.3 us at 1024 entries, 22.6 vs 23.9 at
    // 4096, 28.0 vs 26.2 at 8192: beyond ~4096 its attention role needs the workgroups its projections run on)
    const bool fused = path != CF_PATH_PIPELINE && cus >= MLAF_WGS && (path == CF_PATH_FUSED || seq_len <= 4096);
    if (path == CF_PATH_FUSED && !fused)
        return api_fail(CF_EUNSUPPORTED, "the persistent MLA kernel needs %d CUs (device has %d)", MLAF_WGS, cus);
    if (fused) {
        static thread_local unsigned long long attr_devs2 = 0;
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 64 && !((attr_devs2 >> dev) & 1ull)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mla_fused<false>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, MLAF_LDS);
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mla_fused<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, MLAF_LDS);
            if (e != hipSuccess) return api_fail(CF_ELAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
            attr_devs2 |= 1ull << dev;
        }
        const int n_tok = (int)seq_len;
        // token ranges of 128 x iters entries, two workgroups (column halves) per range: at most 128 ranges
        const int iters = (n_tok + 128 * 128 - 1) / (128 * 128);
        const int nsplit = (n_tok + 128 * iters - 1) / (128 * iters);
        MlaFusedArgs a;
        {
            const int c_wgs = ((nsplit + 7) / 8) * 16;       // c' of the last range, rounded to whole XCD groups
            a.na_wgs = MLAF_WGS - c_wgs < 192 ? 192 : MLAF_WGS - c_wgs;
        }
        a.state = w.state;
        a.x = (const h16*)input; a.rms_w = (const h16*)rms_input_weight; a.eps = eps;
        a.w_q_nope = (const h16*)weight_q_nope; a.w_kv = (const h16*)weight_kv_nope;
        a.w_q_pe = (const h16*)weight_q_pe; a.w_k_pe = (const h16*)weight_k_pe;
        a.n_a = (with_pe ? 57 : 40) * MLA_A_KS; a.g_a = w.g_a;
        a.w_uk = (const h16*)weight_uk; a.rms_ckv_w = (const h16*)rms_ckv_weight; a.cos = cos; a.sin = sin;
        a.with_pe = with_pe ? 1 : 0; a.g_q = w.g_q; a.latent_out = (h16*)latent_out;
        a.cache = (const h16*)ckv_cache; a.zeros = w.zeros; a.n_tok = n_tok; a.iters = iters; a.nsplit = nsplit;
        a.scale_log2e = 1.4426950408889634f / std::sqrt((float)(MLA_NOPE + MLA_ROPE));
        a.g_po = w.g_po; a.g_ml = w.g_ml;
        a.trace = (unsigned long long*)api_trace();
        a.w_uv = (const h16*)weight_uv; a.g_d = w.g_d; a.g_xcc = w.g_xcc; a.w_o = (const h16*)weight_o; a.g_e = w.g_e; a.out = (h16*)out;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (g_mla_prof) { (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventRecord(e0, st); }
        (void)hipGetLastError();
        if (rope_scores) hipLaunchKernelGGL(k_mla_fused<true>, dim3(MLAF_WGS), dim3(512), MLAF_LDS, st, a);
        else hipLaunchKernelGGL(k_mla_fused<false>, dim3(MLAF_WGS), dim3(512), MLAF_LDS, st, a);
        hipError_t e = hipGetLastError();
        if (g_mla_prof) {
            (void)hipEventRecord(e1, st);
            (void)hipEventSynchronize(e1);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            g_mla_ms[0] += ms;
            ++g_mla_calls;
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        }
        if (e != hipSuccess) return api_fail(CF_ELAUNCH, "launch: %s", hipGetErrorString(e));
        api_set_last_path(CF_PATH_FUSED);
        return CF_OK;
    }
    api_set_last_path(CF_PATH_PIPELINE);

    hipEvent_t ev[CF_MLA_STAGES + 1];
    if (g_mla_prof)
        for (auto& e : ev) hipEventCreate(&e);
    auto mark = [&](int i) { if (g_mla_prof) hipEventRecord(ev[i], st); };
    (void)hipGetLastError();
    mark(0);
    {
        const int strips = with_pe ? 57 : 40;
        MlaAbArgs a{w.state, (const h16*)input, (const h16*)rms_input_weight, eps, (const h16*)weight_q_nope,
                    (const h16*)weight_kv_nope, (const h16*)weight_q_pe, (const h16*)weight_k_pe, strips * MLA_A_KS, w.g_a,
                    (const h16*)weight_uk, (const h16*)rms_ckv_weight, cos, sin, with_pe ? 1 : 0,
                    w.qlat, w.latent_new, (h16*)latent_out};
        hipLaunchKernelGGL(k_mla_ab, dim3(a.n_a + 129), dim3(512), 0, st, a);
    }
    mark(1);
    const int n_tok = (int)seq_len;
    const int iters = (n_tok + 64 * MLA_NSPLIT_MAX - 1) / (64 * MLA_NSPLIT_MAX);
    const int nsplit = (n_tok + 64 * iters - 1) / (64 * iters);
    {
        // scores in base 2: exp2((s - m) * log2(e) / sqrt(192))   (softmax_scale = rsqrt(HEAD_DIM), kernel.cuh:47)
        MlaAttnArgs a{w.state, w.qlat, (const h16*)ckv_cache, w.latent_new, n_tok, iters,
                      1.4426950408889634f / std::sqrt((float)(MLA_NOPE + MLA_ROPE)), w.part_o, w.part_ml};
        if (rope_scores) hipLaunchKernelGGL(k_mla_attn<true>, dim3(nsplit), dim3(256), ATTN_LDS, st, a);
        else hipLaunchKernelGGL(k_mla_attn<false>, dim3(nsplit), dim3(256), ATTN_LDS, st, a);
    }
    mark(2);
    {
        MlaDeArgs a{w.state, w.part_o, w.part_ml, nsplit, (const h16*)weight_uv, w.g_d, (const h16*)weight_o, w.g_e, (h16*)out};
        hipLaunchKernelGGL(k_mla_de, dim3(MLA_D_WGS + MLA_E_WGS), dim3(512), 0, st, a);
    }
    mark(3);
    hipError_t e = hipGetLastError();
    if (g_mla_prof) {
        hipEventSynchronize(ev[CF_MLA_STAGES]);
        for (int i = 0; i < CF_MLA_STAGES; ++i) {
            float ms = 0.f;
            hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
            g_mla_ms[i] += ms;
        }
        ++g_mla_calls;
        for (auto& x : ev) hipEventDestroy(x);
    }
    if (e != hipSuccess) return api_fail(CF_ELAUNCH, "launch: %s", hipGetErrorString(e));
    return CF_OK;
}

}  // extern "C"
