// cf_mla_api.hip -- C-ABI of the DeepSeek MLA decoder-layer op (include/clusterfusion_hip.h).
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cmath>
#include "clusterfusion_hip.h"
#include "cf_mla_kernels.h"

namespace cf {
int api_fail(int code, const char* fmt, ...);      // cf_api.hip: sets the thread-local error text
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
    float* part_o;            // [256][16][512]
    float* part_ml;           // [256][16][2]
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
