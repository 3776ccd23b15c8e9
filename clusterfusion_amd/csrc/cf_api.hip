// cf_api.hip -- host side of libclusterfusion_hip.so: argument checks, launch planning, C-ABI.
// Interface contract and reference citations: include/clusterfusion_hip.h.
#include "clusterfusion_hip.h"

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "cf_decode_kernels.h"
#include "cf_fused_kernel.h"
#include "cf_fused_kernel_g.h"
#include "cf_fused_kernel_gb.h"
#include "cf_batch_kernels.h"
#include "cf_fused_kernel_b.h"
#include "cf_fused_kernel_q.h"
#include "cf_fused_kernel_s.h"
#include "cf_tp_kernels.h"

namespace {

thread_local char g_err[512] = "";
thread_local int g_kv_splits = 0;
thread_local int g_path = CF_PATH_AUTO;
thread_local void* g_trace = nullptr;
thread_local int g_flags = 0;
thread_local int g_last_path = 0;
thread_local const char* g_last_variant = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace

namespace cf {
// shared with the other translation units of the library (cf_mla_api.hip)
int api_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
int api_path() { return g_path; }

// ---- sticky failure word ----------------------------------------------------------------------------------------------
// One host-mapped (pinned) word per device.  A persistent kernel whose exchange gives up writes its code there (device
// side: flag_exchange_error) besides the workspace's error word; every later layer call of the process looks at it first
// -- a plain host read, no synchronisation -- and returns CF_ELAUNCH once: the call after a failed one raises even if the
// caller never polls cf_workspace_status.
struct HostWord { uint32_t* host; uint32_t* dev; };
static HostWord g_host_word[64] = {};
static std::mutex g_host_word_mu;
static HostWord* host_word_for_current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    HostWord& w = g_host_word[dev];
    if (!w.host) {
        std::lock_guard<std::mutex> lock(g_host_word_mu);
        if (!w.host) {
            void* h = nullptr;
            void* d = nullptr;
            if (hipHostMalloc(&h, 64, hipHostMallocMapped) != hipSuccess) return nullptr;
            memset(h, 0, 64);
            if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { hipHostFree(h); return nullptr; }
            w.dev = static_cast<uint32_t*>(d);
            w.host = static_cast<uint32_t*>(h);
        }
    }
    return &w;
}
// 0, or the exchange code of a persistent-kernel launch that failed since the last check (cleared by the read)
uint32_t api_take_sticky_error() {
    HostWord* w = host_word_for_current_device();
    if (!w) return 0;
    const uint32_t code = __atomic_load_n(w->host, __ATOMIC_RELAXED);
    if (code) __atomic_store_n(w->host, 0u, __ATOMIC_RELAXED);
    return code;
}
uint32_t* api_sticky_device_pointer() {
    HostWord* w = host_word_for_current_device();
    return w ? w->dev : nullptr;
}
void* api_trace() { return g_trace; }
void api_set_last_path(int p) { g_last_path = p; }
// What a raised sticky word means, for the CF_ELAUNCH text of the call that finds it (codes: cf_workspace_status).
int api_fail_sticky(uint32_t code) {
    if (code == 7)
        return api_fail(CF_ELAUNCH, "an earlier TP gather on this device timed out (code 7): a peer rank never published its partial "
                        "into this rank's receive area (see cf_tp_area_status); the gather's output was filled with NaN where slots "
                        "were missing and the areas' epochs are out of step -- re-create the receive areas. Nothing was launched "
                        "now; call again to continue");
    if (code == 6)
        return api_fail(CF_ELAUNCH, "an earlier 5 .. 32-row persistent launch on this device failed: the hand-off of the normalised rows "
                        "(X0, code 6) gave up (its workgroups were not co-resident -- was another stream or process using the GPU?); "
                        "the outputs of THAT call are invalid. Nothing was launched now; call again to continue");
    return api_fail(CF_ELAUNCH, "an earlier persistent-kernel launch on this device failed: exchange %u gave up (its workgroups "
                    "were not co-resident -- was another stream or process using the GPU?); the outputs of THAT call are "
                    "invalid. Nothing was launched now; call again to continue", code);
}
}  // namespace cf

namespace {
// ---- in-kernel TP publishes waiting for their gather ------------------------------------------------------------------
// tp_publish_wg (cf_fused_kernel.h) lays the granules out for n = hidden values; a gather of another n would poll other
// granules, spin to its bound and return NaN.  The layer call notes the n it published under this rank's own
// area; the gather that follows must ask for the same n (CF_EINVAL otherwise, nothing launched).
std::mutex g_tp_pending_mu;
std::unordered_map<const void*, int> g_tp_pending;
void tp_note_publish(const void* own_area, int n) {
    std::lock_guard<std::mutex> lock(g_tp_pending_mu);
    g_tp_pending[own_area] = n;
}
// 0 = fine (no publish pending, or the sizes agree); otherwise the n that was published.  The note is consumed EITHER way: a
// mismatch is reported once -- by the CF_EINVAL of the gather that found it, which launched nothing -- and must not poison every
// later gather of these areas.  Host side, call time: a note is left by an eager (or capturing) layer call and read by
// the next eager (or capturing) gather call of this process; replays of a captured graph make no calls and are not checked.
int tp_match_gather(const void* own_area, int n) {
    std::lock_guard<std::mutex> lock(g_tp_pending_mu);
    auto it = g_tp_pending.find(own_area);
    if (it == g_tp_pending.end()) return 0;
    const int published = it->second;
    g_tp_pending.erase(it);
    return published == n ? 0 : published;
}
void tp_forget(const void* area) {
    std::lock_guard<std::mutex> lock(g_tp_pending_mu);
    g_tp_pending.erase(area);
}
}  // namespace

namespace {

constexpr int NSPLIT_MAX = 64;
constexpr int KSPLIT_MAX = 64;
constexpr int CHIP_CUS = 256;

inline size_t align256(size_t v) { return (v + 255) & ~size_t(255); }

struct Workspace {
    uint32_t* state;          // persistent kernel: [0] epoch, [1] error     (first 256 B)
    unsigned long long* g_xcc;   // [256] XCC id of each workgroup of the persistent kernel
    unsigned long long* g_qkv;   // [Hkv][(G+2)*128] granules
    unsigned long long* g_rec;   // [Hq][8][FUSED_REC]
    unsigned long long* g_attn;  // [Hq*128]
    unsigned long long* g_qkv_io;   // [Hq][8][(G+2)*128]  [in,out] split-K partials
    unsigned long long* g_part;     // [Hq][hidden]        [in,out] per-head O-projection partials
    float* qkv_raw;   // [batch][KSPLIT_MAX][qkv_dim]   (only ksplit slices used)
    float* part_o;    // [batch][Hq][NSPLIT_MAX][128]
    float* part_ml;   // [batch][Hq][NSPLIT_MAX][2]
    float* opart;     // [batch][Hq][hidden]
    float* attn;      // [batch][Hq*128]  merged, normalised attention output
    cf::h16* xn16;    // [batch][hidden]  batch > 1: normalised activations, fp16 (MFMA operand)
    cf::h16* attn16;  // [batch][Hq*128]  batch > 1: attention output, fp16 (MFMA operand)
    unsigned long long* g_bqkv;    // [batch][Hq][384]    5 .. 16 rows in one persistent launch: q|k|v granules of every (row, head)
    unsigned long long* g_battn;   // [batch][Hq*64]      ... and the attention outputs (fp16 pairs)
    unsigned long long* g_brec;    // [batch][Hq][8][FUSED_RECH]   ... records of the rows that span several workgroups' token ranges
    unsigned long long* g_bxn;     // [batch][hidden] fp16 + [batch] flag granules   ... the normalised rows (X0)
    size_t total;
};

Workspace carve(const cf_dims& d, int batch, void* base) {
    const size_t qkv_dim = (size_t)(d.n_q_heads + 2 * d.n_kv_heads) * d.head_dim;
    Workspace w;
    size_t off = 0;
    char* p = static_cast<char*>(base);
    w.state = reinterpret_cast<uint32_t*>(p + off);
    off += 256;
    w.g_xcc = reinterpret_cast<unsigned long long*>(p + off);
    off += align256((size_t)cf::FUSED_WGS * 8);
    w.g_qkv = reinterpret_cast<unsigned long long*>(p + off);
    off += align256((size_t)qkv_dim * 8);
    w.g_rec = reinterpret_cast<unsigned long long*>(p + off);
    off += align256((size_t)d.n_q_heads * (cf::FUSED_WGS / d.n_kv_heads > 32 ? cf::FUSED_WGS / d.n_kv_heads : 32) * cf::FUSED_REC_G * 8);
    w.g_attn = reinterpret_cast<unsigned long long*>(p + off);
    off += align256((size_t)d.n_q_heads * cf::HEAD_DIM * 8);
    w.g_qkv_io = reinterpret_cast<unsigned long long*>(p + off);
    {   // ... doubles as the merged records of the two-level record merge: [Hq][8 records of FUSED_RECH granules, whole lines]
        const size_t splitk = (size_t)d.n_kv_heads * cf::FUSED_SPLITS * (d.n_q_heads / d.n_kv_heads + 2) * cf::HEAD_DIM * 8;
        const size_t nsg = d.n_kv_heads <= 4 ? 8 : 0;      // (two-level merge: 8 merged records per q head)
        const size_t merged = (size_t)d.n_q_heads * ((nsg * cf::FUSED_RECH + 15) & ~size_t(15)) * 8;
        off += align256(splitk > merged ? splitk : merged);
    }
    w.g_part = reinterpret_cast<unsigned long long*>(p + off);
    off += align256((size_t)d.n_q_heads * d.hidden * 8);
    w.qkv_raw = reinterpret_cast<float*>(p + off);
    off += align256((size_t)batch * KSPLIT_MAX * qkv_dim * 4);
    w.part_o = reinterpret_cast<float*>(p + off);
    off += align256((size_t)batch * d.n_q_heads * NSPLIT_MAX * cf::HEAD_DIM * 4);
    w.part_ml = reinterpret_cast<float*>(p + off);
    off += align256((size_t)batch * d.n_q_heads * NSPLIT_MAX * 2 * 4);
    w.opart = reinterpret_cast<float*>(p + off);
    off += align256((size_t)batch * d.n_q_heads * d.hidden * 4);
    w.attn = reinterpret_cast<float*>(p + off);
    off += align256((size_t)batch * d.n_q_heads * cf::HEAD_DIM * 4);
    w.xn16 = reinterpret_cast<cf::h16*>(p + off);
    off += align256((size_t)batch * d.hidden * 2);
    w.attn16 = reinterpret_cast<cf::h16*>(p + off);
    off += align256((size_t)batch * d.n_q_heads * cf::HEAD_DIM * 2);
    const bool rows_q = batch > 4 && batch <= cf::FusedQGeomT<2>::MAX_ROWS && d.n_q_heads == d.n_kv_heads;
    w.g_bqkv = reinterpret_cast<unsigned long long*>(p + off);
    off += rows_q ? align256((size_t)batch * d.n_q_heads * 384 * 8) : 0;
    w.g_battn = reinterpret_cast<unsigned long long*>(p + off);
    off += rows_q ? align256((size_t)batch * d.n_q_heads * (cf::HEAD_DIM / 2) * 8) : 0;
    w.g_brec = reinterpret_cast<unsigned long long*>(p + off);
    off += rows_q ? align256((size_t)batch * d.n_q_heads * cf::FUSED_SPLITS * cf::FUSED_RECH * 8) : 0;
    w.g_bxn = reinterpret_cast<unsigned long long*>(p + off);
    off += rows_q ? align256((size_t)batch * d.hidden * 2 + (size_t)batch * 8) : 0;
    w.total = off;
    return w;
}

int check_dims(const cf_dims& d) {
    if (d.head_dim != cf::HEAD_DIM) return fail(CF_EUNSUPPORTED, "head_dim %d unsupported (128 only)", d.head_dim);
    if (d.hidden <= 0 || d.hidden % 512) return fail(CF_EUNSUPPORTED, "hidden %d must be a multiple of 512", d.hidden);
    if (d.n_q_heads <= 0 || d.n_kv_heads <= 0 || d.n_q_heads % d.n_kv_heads)
        return fail(CF_EINVAL, "n_q_heads %d must be a positive multiple of n_kv_heads %d", d.n_q_heads, d.n_kv_heads);
    if (d.n_q_heads % 4) return fail(CF_EUNSUPPORTED, "n_q_heads %d must be a multiple of 4", d.n_q_heads);
    return CF_OK;
}

// ---- profiling hook -------------------------------------------------------------------------------
struct ProfRec { hipEvent_t ev[CF_PROFILE_STAGES + 1]; int n; };
thread_local bool g_prof_on = false;
thread_local std::vector<ProfRec> g_prof_pending;
thread_local double g_prof_ms[CF_PROFILE_STAGES] = {0, 0, 0, 0};
thread_local int64_t g_prof_calls = 0;

void prof_flush() {
    for (auto& r : g_prof_pending) {
        hipEventSynchronize(r.ev[r.n - 1]);
        for (int i = 0; i + 1 < r.n; ++i) {
            float ms = 0.f;
            hipEventElapsedTime(&ms, r.ev[i], r.ev[i + 1]);
            g_prof_ms[i] += ms;
        }
        for (int i = 0; i < r.n; ++i) hipEventDestroy(r.ev[i]);
        ++g_prof_calls;
    }
    g_prof_pending.clear();
}

struct ProfScope {
    ProfRec rec;
    hipStream_t st;
    bool on;
    explicit ProfScope(hipStream_t s) : st(s), on(g_prof_on) {
        rec.n = 0;
        if (on) mark();
    }
    void mark() {
        if (!on || rec.n > CF_PROFILE_STAGES) return;
        hipEventCreate(&rec.ev[rec.n]);
        hipEventRecord(rec.ev[rec.n], st);
        ++rec.n;
    }
    ~ProfScope() {
        if (!on) return;
        g_prof_pending.push_back(rec);
        if (g_prof_pending.size() >= 2048) prof_flush();
    }
};

// ---- phase-1 shares of the persistent [out,in] MHA kernel ------------------------------------------------
// Any workgroup can produce any row pair of Wqkv (consumers find q|k|v by granule address), so the split of the 6144
// pairs over the 256 workgroups is a pure load-balancing knob.  share(b) = P1_SHARE[b / 64][b % 2]: under hipGraph
// replay all workgroups start within 0.4 us, but odd XCDs stream ~8 % slower and the workgroups 64..127 (slot b / 64 = 1:
// the heads h = 1 mod 4, whose 256-B K/V pieces stream ~17 % slower, tools/ubench/kv_bw.hip) finish phase 2 about 2 us late
// (tools/fused_timeline.py, CF_TL_GRAPH=1 CF_TL_ABS=1; tools/tune_p1_shares.py iterates a 4 x 8 table from those
// stamps -- it converges to this pattern within the noise of the stamps).  Same-box A/B at S = 4096: equal shares
// 37.28 us, this table 36.3 us per layer.  Short caches (S <= 1024) are best with equal shares (28.5 vs 29.3 us at
// S = 512): `flat`.  CF_P1_TABLE="32 shares, [b / 64][b % 8]" overrides (tuning).
constexpr int P1_SHARE[4][2] = {{30, 23}, {22, 15}, {30, 23}, {28, 21}};     // x 32 workgroups each: 6144
void fill_p1_shares(unsigned short (&start)[cf::FUSED_WGS_C + 1], bool flat) {
    static int table[4][8];
    static bool have_table = false;
    static std::once_flag parsed_once;      // (the rest of the API keeps its state per thread; this table is process-wide)
    std::call_once(parsed_once, [] {
        if (const char* e = getenv("CF_P1_TABLE")) {
            int n = 0, sum = 0;
            bool ok = true;
            const char* q = e;
            while (n < 32 && *q) {
                char* end = nullptr;
                const long v = strtol(q, &end, 10);
                if (end == q) break;
                table[n / 8][n % 8] = (int)v;
                sum += (int)v;
                ok &= v >= 8 && v <= 32;      // (8 core pairs per workgroup are fixed; four slots of 8 pairs)
                ++n;
                q = *end == ',' ? end + 1 : end;
            }
            have_table = n == 32 && ok && sum * 8 == 6144;
            if (!have_table && *e) fprintf(stderr, "[clusterfusion] CF_P1_TABLE ignored (32 shares in 8..32, sum 768)\n");
        }
    });
    int at = 0;
    for (int b = 0; b < cf::FUSED_WGS_C; ++b) {
        start[b] = (unsigned short)at;
        at += flat ? 24 : have_table ? table[b >> 6][b & 7] : P1_SHARE[b >> 6][b & 1];
    }
    start[cf::FUSED_WGS_C] = (unsigned short)at;      // == 6144
}
// The same table in the form k_fused_decode_mha reads it (FusedArgs::p1_tab: scalar loads and scalar arithmetic only).  Every
// workgroup's share holds 8 core pairs (pair 8 b + w: slot 0 of wavefront w, requested before anything else is read); the
// table deals the OTHER pairs, 2048 .. 6143: share - 8 per workgroup.
void fill_p1_table(cf::FusedArgs& fa) {
    unsigned short start[cf::FUSED_WGS_C + 1];
    fill_p1_shares(start, /*flat=*/false);
    int at = 2048;
    for (int s = 0; s < 4; ++s) {
        unsigned long long tab = 0, pre = 0;
        int acc = 0;
        for (int x = 0; x < 8; ++x) {
            const int share = start[64 * s + x + 1] - start[64 * s + x] - 8;
            tab |= (unsigned long long)share << (8 * x);
            pre |= (unsigned long long)acc << (8 * x);
            acc += share;
        }
        fa.p1_tab[s] = tab;
        fa.p1_pre[s] = pre;
        fa.p1_rowsum[s] = (unsigned)acc;
        fa.p1_grp[s] = (unsigned)at;
        at += 8 * acc;
    }
}

// ---- phase-1 shares of the 5 .. 16-row persistent kernel (cf_fused_kernel_q.h), in Wqkv ROWS -----------------------------
// Equal shares, minus the X0 producers' (below).  The systematic stream-rate effects (heads h = 1 mod 4, odd XCDs: DESIGN 3.1)
// show in its timeline too (tools/fused_timeline.py 1024 0 b8, CF_TL_MAP=1: the workgroups 64..127 finish phase 2 ~4 us late),
// but skewed shares {slot 1 even, slot 1 odd, other even, other odd} = {38,34,54,50} ... {18,14,61,57} all measured within
// +-0.5 us of equal shares at 8 and 16 rows: the kernel is paced by its total request volume, not by the stragglers.
// CF_Q_SHARES="a,b,c,d" overrides (tuning).
void fill_q_shares(unsigned short (&start)[cf::FUSED_WGS_C + 1], int batch) {
    static int env[4] = {0, 0, 0, 0};
    static std::once_flag once;
    std::call_once(once, [] {
        if (const char* e = getenv("CF_Q_SHARES")) {
            int v[4];
            if (sscanf(e, "%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3]) == 4 && 32 * (v[0] + v[1]) + 96 * (v[2] + v[3]) == 12288 &&
                v[0] > 8 && v[1] > 8 && v[2] <= 60 && v[3] <= 60 && v[0] <= 60 && v[1] <= 60 && v[2] > 8 && v[3] > 8)
                memcpy(env, v, sizeof(v));
            else fprintf(stderr, "[clusterfusion] CF_Q_SHARES ignored (four shares in 9..60 with 32 (a + b) + 96 (c + d) = 12288)\n");
        }
    });
    static const int flat[4] = {48, 48, 48, 48};
    const int* sh = env[0] ? env : flat;
    // the workgroups 17 r, r < batch, normalise row r and publish it before they request their first tile (X0): 8 rows less
    // each, handed to the next two workgroups that are not producers themselves (a share is at most 64 rows: four tiles)
    int share[cf::FUSED_WGS_C];
    for (int b = 0; b < cf::FUSED_WGS_C; ++b) share[b] = sh[((b >> 6) == 1 ? 0 : 2) + (b & 1)];
    const int bt = batch > 16 ? 2 : 1;
    auto is_producer = [&](int b) {      // (cf_fused_kernel_q.h fused_q_producer)
        const int r = bt == 1 ? b / 17 : b >> 3;
        return r < batch && b == cf::fused_q_producer(bt, r);
    };
    for (int b = 0; b < cf::FUSED_WGS_C; ++b) {
        if (!is_producer(b)) continue;
        share[b] -= 8;
        int given = 0;
        for (int k = 1; k < cf::FUSED_WGS_C && given < 2; ++k) {
            const int t = (b + k) % cf::FUSED_WGS_C;
            if (!is_producer(t) && share[t] + 4 <= 64) { share[t] += 4; ++given; }
        }
    }
    int at = 0;
    for (int b = 0; b < cf::FUSED_WGS_C; ++b) {
        start[b] = (unsigned short)at;
        at += share[b];
    }
    start[cf::FUSED_WGS_C] = (unsigned short)at;      // == 12288
}

// ---- launch helpers ------------------------------------------------------------------------------
template <int J>
void launch_qkv_rows(const cf::NormArgs& na, const cf::h16* W, int n_rows, int batch, float* raw, hipStream_t st) {
    constexpr int R = J > 8 ? 1 : 2;
    // aim at ~2 workgroups per CU; whole R-groups per wavefront
    int rpw = (n_rows + CHIP_CUS * 2 * 4 - 1) / (CHIP_CUS * 2 * 4);
    rpw = ((rpw + R - 1) / R) * R;
    const int nwg = (n_rows + rpw * 4 - 1) / (rpw * 4);
    hipLaunchKernelGGL((cf::k_qkv_rows<J, R>), dim3(nwg, batch), dim3(256), 0, st, na, W, n_rows, rpw, raw);
}

template <int J>
void launch_oproj_rows(const float* ma, const cf::h16* Wo, int n_rows, int batch, cf::h16* out,
                       const cf::ResidualOut& ro, hipStream_t st) {
    constexpr int R = J > 8 ? 1 : 2;
    int rpw = (n_rows + CHIP_CUS * 4 - 1) / (CHIP_CUS * 4);
    rpw = ((rpw + R - 1) / R) * R;
    const int nwg = (n_rows + rpw * 4 - 1) / (rpw * 4);
    hipLaunchKernelGGL((cf::k_oproj_rows<J, R>), dim3(nwg, batch), dim3(256), 0, st, ma, Wo, n_rows, rpw, out, ro);
}

template <int G>
void launch_attn(const cf::AttnArgs& aa, int batch, hipStream_t st) {
    constexpr int U = G >= 8 ? 4 : 8;
    hipLaunchKernelGGL((cf::k_attn_split<G, U>), dim3(aa.nsplit * aa.Hkv, batch), dim3(256), 0, st, aa);
}

// batch > 1, [out,in] weights: projection as a weight-streaming MFMA GEMM (cf_batch_kernels.h).  Rows are
// processed in chunks of <= 32 (two 16-row batch tiles held in registers); K / 256 k-blocks per wavefront.
template <int NB, int BT, int DEPTH>
void launch_proj_one(const cf::ProjArgs& pa, const cf::ResidualOut& ro, hipStream_t st) {
    const int ntiles = pa.n_rows / 16;
    const int grid = ntiles < CHIP_CUS ? ntiles : CHIP_CUS;
    hipLaunchKernelGGL((cf::k_proj_rows_mfma<NB, BT, DEPTH>), dim3(grid), dim3(512), 0, st, pa, ro);
}
// K = 4096: rows streamed as 1-KB pieces through a per-wavefront LDS image (k_proj_rows_lds); > 64 KB of LDS
template <int BT, int DEPTH>
bool launch_proj_lds(const cf::ProjArgs& pa, const cf::ResidualOut& ro, hipStream_t st) {
    constexpr int LDS = cf::proj_lds_bytes<BT>();
    static thread_local unsigned long long attr_devs = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (dev >= 64 || !((attr_devs >> dev) & 1ull)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&cf::k_proj_rows_lds<BT, DEPTH>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        if (dev < 64) attr_devs |= 1ull << dev;
    }
    const int ntiles = pa.n_rows / 16;
    const int grid = ntiles < CHIP_CUS ? ntiles : CHIP_CUS;
    (void)hipGetLastError();      // (only THIS launch's status decides the fall-back: an older pending error is not ours to judge)
    hipLaunchKernelGGL((cf::k_proj_rows_lds<BT, DEPTH>), dim3(grid), dim3(512), LDS, st, pa, ro);
    return hipGetLastError() == hipSuccess;
}
// more than 32 rows: every weight byte once per launch of <= 128 rows (k_proj_rows_big)
template <int MT, int NG>
bool launch_proj_big_one(const cf::ProjArgs& pa, const cf::ResidualOut& ro, hipStream_t st) {
    constexpr int LDS = cf::proj_big_lds_bytes<MT, NG>();
    static thread_local unsigned long long attr_devs = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    // the > 64 KB opt-in is per device: remembered for devices 0..63, repeated on every call beyond
    if (dev >= 64 || !((attr_devs >> dev) & 1ull)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&cf::k_proj_rows_big<MT, NG>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        if (dev < 64) attr_devs |= 1ull << dev;
    }
    (void)hipGetLastError();      // (an older error pending on this thread must not read as a refusal of this launch: ADVICE r5)
    hipLaunchKernelGGL((cf::k_proj_rows_big<MT, NG>), dim3(pa.n_rows / (16 * MT)), dim3(512), LDS, st, pa, ro);
    // a launch that was refused (not: a kernel that failed later) hands the rows to the chunked launches
    return hipGetLastError() == hipSuccess;
}
bool launch_proj_big(const cf::ProjArgs& c, const cf::ResidualOut& ro, hipStream_t st) {
    const bool wide = c.batch > 64;
    // 48 rows per workgroup where that still gives every CU a workgroup (Wqkv of 32 heads: 256), else 32, else 16
    if (c.n_rows % 48 == 0 && c.n_rows / 48 >= CHIP_CUS) return wide ? launch_proj_big_one<3, 2>(c, ro, st) : launch_proj_big_one<3, 1>(c, ro, st);
    if (c.n_rows % 32 == 0 && c.n_rows / 32 >= CHIP_CUS) return wide ? launch_proj_big_one<2, 2>(c, ro, st) : launch_proj_big_one<2, 1>(c, ro, st);
    return wide ? launch_proj_big_one<1, 2>(c, ro, st) : launch_proj_big_one<1, 1>(c, ro, st);
}
bool launch_proj_mfma(cf::ProjArgs pa, const cf::ResidualOut& ro_last, bool is_last_stage, hipStream_t st) {
    const int nb = pa.K / 256;
    if (pa.batch > 32 && pa.K % 1024 == 0 && pa.n_rows % 16 == 0 && !(g_flags & 2048)) {      // (debug bit 2048: the chunked launches below)
        const int batch = pa.batch;
        bool ok = true;
        for (int b0 = 0; b0 < batch && ok; b0 += cf::BIG_MAX_ROWS) {
            cf::ProjArgs c = pa;
            c.batch = batch - b0 < cf::BIG_MAX_ROWS ? batch - b0 : cf::BIG_MAX_ROWS;
            c.in += (size_t)b0 * pa.K;
            if (c.out_f32) c.out_f32 += (size_t)b0 * pa.n_rows;
            if (c.out_h16) c.out_h16 += (size_t)b0 * pa.n_rows;
            cf::ResidualOut ro{nullptr, nullptr, nullptr, 0};
            if (is_last_stage && ro_last.residual_out) {
                ro = ro_last;
                ro.x += (size_t)b0 * ro.hidden;
                ro.residual += (size_t)b0 * ro.hidden;
                ro.residual_out += (size_t)b0 * ro.hidden;
            }
            ok = launch_proj_big(c, ro, st);
        }
        if (ok) return true;
    }
    const int chunk_max = nb > 16 ? 16 : 32;   // two batch tiles only while both operands fit the registers
    const int batch = pa.batch;
    const bool deep = pa.n_rows / 16 > CHIP_CUS;   // more than one tile per workgroup: keep two in flight
    for (int b0 = 0; b0 < batch; b0 += chunk_max) {
        cf::ProjArgs c = pa;
        c.batch = batch - b0 < chunk_max ? batch - b0 : chunk_max;
        c.in += (size_t)b0 * pa.K;
        if (c.out_f32) c.out_f32 += (size_t)b0 * pa.n_rows;
        if (c.out_h16) c.out_h16 += (size_t)b0 * pa.n_rows;
        cf::ResidualOut ro{nullptr, nullptr, nullptr, 0};
        if (is_last_stage && ro_last.residual_out) {   // the chunk writes the residual rows it owns
            ro = ro_last;
            ro.x += (size_t)b0 * ro.hidden;
            ro.residual += (size_t)b0 * ro.hidden;
            ro.residual_out += (size_t)b0 * ro.hidden;
        }
        const bool two = c.batch > 16;
        if (nb == 16 && !(g_flags & 16)) {     // (debug flag 16: the direct-operand-layout kernel, for comparison)
            // (one tile ahead: two measured 2 % slower -- 94.0 vs 92.3 us per call at batch 16)
            const bool ok = two ? launch_proj_lds<2, 1>(c, ro, st) : launch_proj_lds<1, 1>(c, ro, st);
            if (ok) continue;
        }
        // registers: activations BT x NB x 4 + weights DEPTH x NB x 4 VGPRs
#define CF_PROJ_CASE(N)                                                                  \
    case N:                                                                              \
        if (two) launch_proj_one<N, (N <= 16 ? 2 : 1), (N <= 8 ? 2 : 1)>(c, ro, st);     \
        else if (deep) launch_proj_one<N, 1, (N <= 16 ? 2 : 1)>(c, ro, st);              \
        else launch_proj_one<N, 1, 1>(c, ro, st);                                        \
        break;
        switch (nb) {
            CF_PROJ_CASE(2) CF_PROJ_CASE(4) CF_PROJ_CASE(8) CF_PROJ_CASE(16) CF_PROJ_CASE(20)
            default: return false;
        }
#undef CF_PROJ_CASE
    }
    return true;
}

int device_cus() {
    static thread_local int cached_dev = -1, cached_cus = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (dev != cached_dev) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        cached_dev = dev;
        cached_cus = prop.multiProcessorCount;
    }
    return cached_cus;
}

// which persistent-kernel specialisation serves this call: 0 = none (stage pipeline)
enum FusedKind { FK_NONE = 0, FK_MHA32 = 1, FK_GQA_32_8 = 2, FK_MHA16 = 3, FK_MHA8 = 4, FK_MHA4 = 5, FK_GQA_16_4 = 6, FK_GQA_8_2 = 7, FK_GQA_4_1 = 8 };
int fused_kind(const cf_layer_args* a) {
    const cf_dims& d = a->dims;
    if (a->batch != 1 || d.hidden != 4096 || d.head_dim != 128) return FK_NONE;
    if (d.n_q_heads == 32 && d.n_kv_heads == 32) return FK_MHA32;                       // either weight layout
    if (a->weight_layout != CF_W_OUT_IN) return FK_NONE;
    if (d.n_q_heads == 32 && d.n_kv_heads == 8) return FK_GQA_32_8;                      // Llama-3-8B
    if (d.n_q_heads == 16 && d.n_kv_heads == 16) return FK_MHA16;                        // Llama-2-7B, TP=2 shard
    if (d.n_q_heads == 8 && d.n_kv_heads == 8) return FK_MHA8;                           // ... TP=4
    if (d.n_q_heads == 4 && d.n_kv_heads == 4) return FK_MHA4;                           // ... TP=8
    if (d.n_q_heads == 16 && d.n_kv_heads == 4) return FK_GQA_16_4;                      // Llama-3-8B, TP=2 shard
    if (d.n_q_heads == 8 && d.n_kv_heads == 2) return FK_GQA_8_2;                        // ... TP=4
    if (d.n_q_heads == 4 && d.n_kv_heads == 1) return FK_GQA_4_1;                        // ... TP=8
    return FK_NONE;
}
bool fused_shape_ok(const cf_layer_args* a) { return fused_kind(a) != FK_NONE; }

template <class K>
hipError_t set_lds(K kern, int bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

// The persistent kernels assume their 256 workgroups are co-resident (one per CU).  A plain launch gives exactly the
// residency a cooperative launch would (guide: MI355X_MICROARCH.md "Residency and cooperative launch"; hipLaunchCooperative-
// Kernel costs +15-19 us of host time per launch, +17-20 us per graph replay, and buys only the launch-time size check), so the
// check is done here, once per (thread, device, kernel): the occupancy query must admit one workgroup per CU with this
// kernel's registers and LDS.  What the query cannot see -- CUs held by another stream or process -- is caught at run time:
// every spin is bounded, a failed exchange raises the sticky word (api_take_sticky_error) and the epoch still advances.
template <class K>
bool fused_resident(K kern, int lds_bytes) {
    // (K is the same function-pointer type for every instantiation: the verdicts are keyed by the kernel's address)
    struct Verdict { const void* kern; unsigned long long ok_devs, bad_devs; };
    static thread_local std::vector<Verdict> verdicts;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return false;
    const void* key = reinterpret_cast<const void*>(kern);
    Verdict* v = nullptr;
    for (auto& x : verdicts)
        if (x.kern == key) v = &x;
    if (!v) {
        verdicts.push_back(Verdict{key, 0, 0});
        v = &verdicts.back();
    }
    if ((v->ok_devs >> dev) & 1ull) return true;
    if ((v->bad_devs >> dev) & 1ull) return false;
    int per_cu = 0;
    const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, key, cf::FUSED_THREADS, lds_bytes);
    const bool ok = e == hipSuccess && per_cu >= 1 && device_cus() >= cf::FUSED_WGS;
    (ok ? v->ok_devs : v->bad_devs) |= 1ull << dev;
    return ok;
}

// out[c][r] = in[r][c] for `nmat` stacked [rows, cols] fp16 matrices: 64 x 64 tiles through LDS (one 128-B row piece per 8 lanes
// on both sides; the +8 pad keeps the column reads off one bank)
__global__ __launch_bounds__(256) void k_transpose_h16(const cf::h16* in, cf::h16* out, int rows, int cols) {
    __shared__ cf::h16 tile[64][72];
    const size_t mat = (size_t)blockIdx.z * rows * cols;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64, t = threadIdx.x;
    for (int i = t; i < 64 * 8; i += 256) {
        const int r = i >> 3, c8 = (i & 7) * 8;
        *reinterpret_cast<cf::h16x8*>(&tile[r][c8]) = *reinterpret_cast<const cf::h16x8*>(in + mat + (size_t)(r0 + r) * cols + c0 + c8);
    }
    __syncthreads();
    for (int i = t; i < 64 * 8; i += 256) {
        const int c = i >> 3, r8 = (i & 7) * 8;
        cf::h16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = tile[r8 + e][c];
        *reinterpret_cast<cf::h16x8*>(out + mat + (size_t)(c0 + c) * rows + r0 + r8) = v;
    }
}

// test hook: `blocks` workgroups of 64 threads holding `lds_bytes` of LDS each spin for `microseconds` on `stream`
__global__ void k_debug_occupy(long long ticks, unsigned* sink) {
    extern __shared__ char smem_occ[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned acc = 0;
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) {
        acc += smem_occ[threadIdx.x];
        __builtin_amdgcn_s_sleep(32);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int ilog2_exact(int v) {
    int s = 0;
    while ((1 << s) < v) ++s;
    return (1 << s) == v ? s : -1;
}

}  // namespace

extern "C" {

int cf_abi_version(void) { return CF_ABI_VERSION; }
const char* cf_last_error(void) { return g_err; }

size_t cf_workspace_bytes(const cf_dims* dims, int32_t batch) {
    if (!dims || batch <= 0 || check_dims(*dims) != CF_OK) return 0;     // (carve divides by the head counts)
    return carve(*dims, batch, nullptr).total;
}

uint64_t cf_algorithmic_bytes(const cf_dims* d, int32_t batch, int64_t seq_len, int32_t has_residual) {
    if (!d) return 0;
    const uint64_t hd = d->head_dim, D = d->hidden;
    const uint64_t qd = (uint64_t)d->n_q_heads * hd, kd = (uint64_t)d->n_kv_heads * hd;
    uint64_t w = 2 * D * (qd + 2 * kd) + 2 * qd * D;
    uint64_t kv = 2 * 2 * (uint64_t)seq_len * kd * batch;
    uint64_t small = (uint64_t)batch * (2 * D + 2 * D + 2 * 2 * kd + 2 * hd * 4) + 2 * D;
    if (has_residual) small += (uint64_t)batch * 4 * D;
    return w + kv + small;
}

int cf_set_path(int32_t path) {
    if (path < CF_PATH_AUTO || path > CF_PATH_FUSED) return fail(CF_EINVAL, "bad path %d", path);
    g_path = path;
    return CF_OK;
}

int cf_last_path(void) { return g_last_path; }
const char* cf_last_variant(void) { return g_last_variant; }
uint32_t cf_take_sticky_error(void) { return cf::api_take_sticky_error(); }

int cf_debug_set_flags(int32_t flags) {
    g_flags = flags;
    return CF_OK;
}

int cf_relayout_weights(const cf_dims* dims, const void* weight_qkv_in_out, const void* weight_o_in_out,
                        void* weight_qkv_out_in, void* weight_o_out_in, void* stream) {
    if (!dims || !weight_qkv_in_out || !weight_o_in_out || !weight_qkv_out_in || !weight_o_out_in)
        return fail(CF_EINVAL, "cf_relayout_weights: null argument");
    if (const int rc = check_dims(*dims)) return rc;
    const cf_dims& d = *dims;
    if (d.n_q_heads != d.n_kv_heads) return fail(CF_EUNSUPPORTED, "cf_relayout_weights: the [in,out] orientation is defined for n_q_heads == n_kv_heads");
    const int qd = d.n_q_heads * d.head_dim;
    if (d.hidden % 64 || qd % 64) return fail(CF_EUNSUPPORTED, "cf_relayout_weights: hidden and q_dim must be multiples of 64");
    if (weight_qkv_in_out == weight_qkv_out_in || weight_o_in_out == weight_o_out_in) return fail(CF_EINVAL, "cf_relayout_weights: in-place re-layout is not supported");
    hipStream_t st = static_cast<hipStream_t>(stream);
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_transpose_h16, dim3(qd / 64, d.hidden / 64, 3), dim3(256), 0, st, (const cf::h16*)weight_qkv_in_out,
                       (cf::h16*)weight_qkv_out_in, d.hidden, qd);
    hipLaunchKernelGGL(k_transpose_h16, dim3(d.hidden / 64, qd / 64, 1), dim3(256), 0, st, (const cf::h16*)weight_o_in_out,
                       (cf::h16*)weight_o_out_in, qd, d.hidden);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CF_ELAUNCH, "cf_relayout_weights: %s", hipGetErrorString(e));
    return CF_OK;
}

int cf_debug_occupy(void* stream, int32_t blocks, int32_t lds_bytes, int64_t microseconds) {
    if (blocks <= 0 || lds_bytes < 0 || lds_bytes > 160 * 1024 || microseconds < 0) return fail(CF_EINVAL, "cf_debug_occupy: bad argument");
    static thread_local bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_debug_occupy), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return fail(CF_ELAUNCH, "cf_debug_occupy: hipFuncSetAttribute failed");
        attr = true;
    }
    hipLaunchKernelGGL(k_debug_occupy, dim3(blocks), dim3(64), lds_bytes, static_cast<hipStream_t>(stream), (long long)microseconds * 100,
                       (unsigned*)nullptr);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CF_ELAUNCH, "cf_debug_occupy: %s", hipGetErrorString(e));
    return CF_OK;
}

int cf_debug_set_trace(void* device_buffer) {
    g_trace = device_buffer;
    return CF_OK;
}

int cf_workspace_init(void* workspace, size_t workspace_bytes, void* stream) {
    if (!workspace || !workspace_bytes) return fail(CF_EINVAL, "NULL workspace");
    if (workspace_bytes < 256) return fail(CF_EWORKSPACE, "workspace %zu B < 256 B", workspace_bytes);
    hipError_t e = hipMemsetAsync(workspace, 0, workspace_bytes, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(CF_ELAUNCH, "hipMemsetAsync: %s", hipGetErrorString(e));
    // state[4..5]: device address of this device's host-mapped failure word (persistent kernels report there too)
    static thread_local uint32_t* slot[64];
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
        slot[dev] = cf::api_sticky_device_pointer();
        if (slot[dev]) {
            e = hipMemcpyAsync(static_cast<char*>(workspace) + 16, &slot[dev], sizeof(uint32_t*), hipMemcpyHostToDevice,
                               static_cast<hipStream_t>(stream));
            if (e != hipSuccess) return fail(CF_ELAUNCH, "hipMemcpyAsync: %s", hipGetErrorString(e));
        }
    }
    return CF_OK;
}

int cf_workspace_status(const void* workspace, void* stream, uint32_t* error_code) {
    if (!workspace || !error_code) return fail(CF_EINVAL, "NULL argument");
    uint32_t st2[2] = {0, 0};
    hipError_t e = hipMemcpyAsync(st2, workspace, sizeof(st2), hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream));
    if (e == hipSuccess) e = hipStreamSynchronize(static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(CF_ELAUNCH, "status read: %s", hipGetErrorString(e));
    *error_code = st2[1];
    return CF_OK;
}

int cf_workspace_last_arm(const void* workspace, void* stream, uint32_t* arm) {
    if (!workspace || !arm) return fail(CF_EINVAL, "NULL argument");
    uint32_t st3[3] = {0, 0, 0};
    hipError_t e = hipMemcpyAsync(st3, workspace, sizeof(st3), hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream));
    if (e == hipSuccess) e = hipStreamSynchronize(static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(CF_ELAUNCH, "arm read: %s", hipGetErrorString(e));
    *arm = st3[2];
    return CF_OK;
}

int cf_set_tuning(int32_t kv_splits) {
    if (kv_splits < 0 || kv_splits > NSPLIT_MAX) return fail(CF_EINVAL, "kv_splits %d out of [0,%d]", kv_splits, NSPLIT_MAX);
    g_kv_splits = kv_splits;
    return CF_OK;
}

int cf_profile_enable(int32_t on) {
    if (!on) prof_flush();
    g_prof_on = on != 0;
    return CF_OK;
}

int cf_profile_read(double* stage_ms, int64_t* n_calls, int32_t reset) {
    prof_flush();
    if (stage_ms) memcpy(stage_ms, g_prof_ms, sizeof(g_prof_ms));
    if (n_calls) *n_calls = g_prof_calls;
    if (reset) {
        memset(g_prof_ms, 0, sizeof(g_prof_ms));
        g_prof_calls = 0;
    }
    return CF_OK;
}

int cf_decoder_layer_ex(const cf_layer_args* a) {
    if (!a) return fail(CF_EINVAL, "args is NULL");
    if (const uint32_t code = cf::api_take_sticky_error()) return cf::api_fail_sticky(code);
    (void)hipGetLastError();      // (only this call's launches decide its status: an error another call left pending -- a capture that
                                  //  was invalidated and abandoned, say -- is not this call's to report)
    const cf_dims& d = a->dims;
    if (int rc = check_dims(d)) return rc;
    if (a->batch <= 0 || a->batch > 65535) return fail(CF_EINVAL, "batch %d out of range", a->batch);
    if (!a->x || !a->weight_qkv || !a->weight_o || !a->rms_weight || !a->cos || !a->sin || !a->out)
        return fail(CF_EINVAL, "NULL required pointer (x/weights/rms/cos/sin/out)");
    {   // every kernel reads its operands with 16-byte vector loads
        const void* ptrs[] = {a->x, a->residual, a->weight_qkv, a->weight_o, a->rms_weight, a->k_cache, a->v_cache, a->out,
                              a->residual_out, a->k_new, a->v_new};
        for (const void* q : ptrs)
            if (reinterpret_cast<uintptr_t>(q) & 15) return fail(CF_EINVAL, "tensor pointers must be 16-byte aligned (%p)", q);
    }
    if (a->residual_out && !a->residual) return fail(CF_EINVAL, "residual_out given without residual");
    if (a->weight_layout != CF_W_OUT_IN && a->weight_layout != CF_W_IN_OUT) return fail(CF_EINVAL, "bad weight_layout %d", a->weight_layout);
    if (a->rope_style != CF_ROPE_NEOX && a->rope_style != CF_ROPE_GPTJ) return fail(CF_EINVAL, "bad rope_style %d", a->rope_style);
    if (a->weight_layout == CF_W_IN_OUT && d.n_q_heads != d.n_kv_heads)
        return fail(CF_EUNSUPPORTED, "[in,out] weights are MHA-only (q|k|v blocks share one column count)");
    const bool paged = a->kv_indptr != nullptr;
    int page_shift = 0;
    if (paged) {
        if (!a->kv_indices) return fail(CF_EINVAL, "kv_indptr without kv_indices");
        page_shift = ilog2_exact(a->page_size);
        if (a->page_size < 1 || page_shift < 0) return fail(CF_EUNSUPPORTED, "page_size %d must be a power of two", a->page_size);
        if (a->page_size > 1 && !a->kv_seq_lens) return fail(CF_EINVAL, "page_size > 1 needs kv_seq_lens");
        if (!(a->kv_cache_ptrs_k && a->kv_cache_ptrs_v) && !(a->k_cache && a->v_cache))
            return fail(CF_EINVAL, "paged mode needs k/v cache pointers");
    } else {
        if (a->seq_len < 0 || a->seq_len > (int64_t)1 << 30) return fail(CF_EINVAL, "seq_len %lld out of range", (long long)a->seq_len);
        if (a->seq_len > 0 && (!a->k_cache || !a->v_cache)) return fail(CF_EINVAL, "NULL k/v cache with seq_len > 0");
        if (a->batch != 1) return fail(CF_EUNSUPPORTED, "contiguous KV mode is single-sequence (batch 1)");
    }
    const int G = d.n_q_heads / d.n_kv_heads;
    if (G != 1 && G != 2 && G != 4 && G != 8) return fail(CF_EUNSUPPORTED, "q/kv head ratio %d unsupported (1,2,4,8)", G);
    const int J_in = d.hidden / 512, J_o = d.n_q_heads * d.head_dim / 512;
    auto okJ = [](int j) { return j == 1 || j == 2 || j == 4 || j == 8 || j == 10 || j == 16; };
    if (!okJ(J_in)) return fail(CF_EUNSUPPORTED, "hidden %d unsupported (512 x {1,2,4,8,10,16})", d.hidden);
    if (!okJ(J_o)) return fail(CF_EUNSUPPORTED, "n_q_heads %d unsupported (4 x {1,2,4,8,10,16})", d.n_q_heads);

    const bool tp_publish = a->tp_areas != nullptr && a->tp_world > 0;
    if (tp_publish) {
        if (a->tp_world > cf::TP_MAX_WORLD || a->tp_rank < 0 || a->tp_rank >= a->tp_world)
            return fail(CF_EINVAL, "tp_rank %d / tp_world %d (world <= %d)", a->tp_rank, a->tp_world, cf::TP_MAX_WORLD);
        for (int p = 0; p < a->tp_world; ++p)
            if (!a->tp_areas[p] || (reinterpret_cast<uintptr_t>(a->tp_areas[p]) & 255))
                return fail(CF_EINVAL, "tp_areas[%d] NULL or not 256-byte aligned", p);
        const int kind = fused_kind(a);
        if (kind == FK_NONE || kind == FK_MHA32 || g_path == CF_PATH_PIPELINE || device_cus() < cf::FUSED_WGS)
            return fail(CF_EUNSUPPORTED, "the in-kernel TP publish (tp_areas) needs a persistent shard kernel: batch 1, hidden 4096, "
                        "[out,in] weights, 16 / 8 / 4 heads or 32q/8kv, 16q/4kv, 8q/2kv, 4q/1kv, >= 256 CUs, path not PIPELINE");
    }
    const size_t need = carve(d, a->batch, nullptr).total;
    if (!a->workspace || a->workspace_bytes < need)
        return fail(CF_EWORKSPACE, "workspace %zu B < required %zu B", a->workspace_bytes, need);
    Workspace ws = carve(d, a->batch, a->workspace);
    hipStream_t st = static_cast<hipStream_t>(a->stream);
    const int qkv_dim = (d.n_q_heads + 2 * d.n_kv_heads) * d.head_dim;

    // ---- plan -----------------------------------------------------------------------------------
    int64_t s_plan = paged ? a->max_seq_len : a->seq_len;
    int nsplit = g_kv_splits;
    if (nsplit <= 0) {
        nsplit = (2 * CHIP_CUS + d.n_kv_heads * a->batch - 1) / (d.n_kv_heads * a->batch);   // ~2 WG / CU
        if (nsplit < 1) nsplit = 1;
        if (nsplit > NSPLIT_MAX) nsplit = NSPLIT_MAX;
        if (!paged || s_plan > 0) {
            int64_t cap = (s_plan + 63) / 64;   // >= 64 tokens per split
            if (cap < 1) cap = 1;
            if (nsplit > cap) nsplit = (int)cap;
        }
    }
    if (s_plan > 0) {   // keep one split's page-table slice stageable in LDS
        while (nsplit < NSPLIT_MAX && (s_plan + nsplit - 1) / nsplit > 16 * ((cf::ATTN_MAX_IDX - 2) / 16)) ++nsplit;
    }
    int ksplit = 1;
    if (a->weight_layout == CF_W_IN_OUT) {
        const int ncb = d.n_q_heads * d.head_dim / 512;
        ksplit = 16;
        while (ksplit < KSPLIT_MAX && 3 * ncb * ksplit < CHIP_CUS && d.hidden / (ksplit * 2) >= 64) ksplit *= 2;
        while (d.hidden / ksplit > cf::COLS_RK_MAX) ksplit *= 2;
        if (ksplit > KSPLIT_MAX || d.hidden % (ksplit * 64)) return fail(CF_EUNSUPPORTED, "cannot K-split hidden %d", d.hidden);
    }

    cf::NormArgs na{(const cf::h16*)a->x, (const cf::h16*)a->residual, (const cf::h16*)a->rms_weight, a->eps, d.hidden};

    auto fill_fused_args = [&](cf::FusedArgs& fa) {
        fa.na = na;
        fa.Wqkv = (const cf::h16*)a->weight_qkv;
        fa.Wo = (const cf::h16*)a->weight_o;
        fa.k_cache = (const cf::h16*)a->k_cache;
        fa.v_cache = (const cf::h16*)a->v_cache;
        fa.kptrs = paged ? a->kv_cache_ptrs_k : nullptr;
        fa.vptrs = paged ? a->kv_cache_ptrs_v : nullptr;
        fa.layer_id = a->layer_id;
        fa.seq_len = (int)a->seq_len;
        fa.indptr = a->kv_indptr;
        fa.indices = a->kv_indices;
        fa.seq_lens = a->kv_seq_lens;
        fa.page_shift = page_shift;
        fa.cos = a->cos;
        fa.sin = a->sin;
        fa.positions = a->positions;
        fa.rope_stride = a->rope_row_stride;
        fa.rope_style = a->rope_style;
        fa.out = (cf::h16*)a->out;
        fa.residual_out = (cf::h16*)a->residual_out;
        fa.k_new = (cf::h16*)a->k_new;
        fa.v_new = (cf::h16*)a->v_new;
        fa.write_cache = paged ? a->write_kv_to_cache : 0;
        fa.state = ws.state;
        fa.g_xcc = ws.g_xcc;
        fa.g_qkv = ws.g_qkv;
        fa.g_rec = ws.g_rec;
        fa.g_attn = ws.g_attn;
        fa.g_qkv_io = ws.g_qkv_io;
        fa.g_part = ws.g_part;
        fa.trace = static_cast<unsigned long long*>(g_trace);
        fa.flags = g_flags;
        fa.tp_rank = a->tp_rank;
        fa.tp_world = a->tp_areas ? a->tp_world : 0;
        for (int p = 0; p < cf::TP_MAX_WORLD; ++p)
            fa.tp_areas[p] = p < fa.tp_world ? static_cast<unsigned long long*>(a->tp_areas[p]) : nullptr;
    };

    // ---- persistent fused kernel -----------------------------------------------------------------
    bool fused = false;
    if (g_path != CF_PATH_PIPELINE && fused_shape_ok(a)) fused = device_cus() >= cf::FUSED_WGS;
    // (A workgroup of the persistent kernels stages its slice of the page table in LDS, up to FUSED_MAX_IDX entries -- half of it
    //  in the grouped-query geometry -- and reads what lies beyond through L2: any length, known to the host or not, qualifies.)
    const bool small_batch_shape = paged && a->batch >= 2 && a->batch <= 4 && d.hidden == 4096 && d.head_dim == 128 && d.n_q_heads == 32 &&
                                   d.n_kv_heads == 32 && a->weight_layout == CF_W_OUT_IN;
    const bool rows_q_shape = paged && a->batch >= 5 && a->batch <= cf::FusedQGeomT<2>::MAX_ROWS && d.hidden == 4096 && d.head_dim == 128 &&
                              d.n_q_heads == 32 && d.n_kv_heads == 32 && a->weight_layout == CF_W_OUT_IN;
    // 2 .. 4 sequences of the grouped-query model (32q/8kv, Llama-3-8B): k_fused_decode_gb (cf_fused_kernel_gb.h)
    const bool gqa_batch_shape = paged && a->batch >= 2 && a->batch <= 4 && d.hidden == 4096 && d.head_dim == 128 && d.n_q_heads == 32 &&
                                 d.n_kv_heads == 8 && a->weight_layout == CF_W_OUT_IN;
    if (g_path == CF_PATH_FUSED && !fused && !small_batch_shape && !rows_q_shape && !gqa_batch_shape)
        return fail(CF_EUNSUPPORTED, "fused path requested but shape/device does not qualify");
    if (g_path == CF_PATH_AUTO && !fused && a->batch == 1 && d.hidden == 4096 && d.head_dim == 128 && device_cus() >= cf::FUSED_WGS) {
        // a Llama-shaped layer that misses the persistent kernels' geometry list runs 1.5-2x slower through the stage pipeline:
        // say so once per geometry instead of leaving it to cf_last_path()
        static std::mutex mu;
        static std::vector<long> told;
        const long key = ((long)d.n_q_heads << 20) | ((long)d.n_kv_heads << 4) | a->weight_layout;
        std::lock_guard<std::mutex> lock(mu);
        bool seen = false;
        for (long k : told) seen |= k == key;
        if (!seen) {
            told.push_back(key);
            fprintf(stderr, "[clusterfusion] %dq/%dkv heads, %s weights: no persistent kernel for this geometry (have: 32/32 either "
                            "layout; [out,in] 32/8, 16/16, 8/8, 4/4, 16/4, 8/2, 4/1) -- running the multi-kernel stage pipeline\n",
                    d.n_q_heads, d.n_kv_heads, a->weight_layout == CF_W_OUT_IN ? "[out,in]" : "[in,out]");
        }
    }
    if (fused) {
        const int kind = fused_kind(a);
        // the > 64 KB dynamic-LDS opt-in is a per-device function attribute: set it once per (thread, device)
        static thread_local unsigned long long attr_devs = 0;
        int cur_dev = 0;
        if (hipGetDevice(&cur_dev) != hipSuccess || cur_dev < 0 || cur_dev > 63) cur_dev = 63;
        const bool attr_set = cur_dev != 63 && ((attr_devs >> cur_dev) & 1ull);
        if (!attr_set) {
            hipError_t e = set_lds(cf::k_fused_decode_mha<false>, cf::FUSED_LDS_BYTES);
            if (e == hipSuccess) e = set_lds(cf::k_fused_decode_mha<true>, cf::FUSED_LDS_BYTES);
            if (e == hipSuccess) e = set_lds(cf::k_fused_decode_g<8, 4>, cf::FusedGeom<8, 4>::LDS_BYTES);
            if (e == hipSuccess) e = set_lds(cf::k_fused_decode_g<16, 1>, cf::FusedGeom<16, 1>::LDS_BYTES);
            if (e == hipSuccess) e = set_lds(cf::k_fused_decode_g<8, 1>, cf::FusedGeom<8, 1>::LDS_BYTES);
            if (e == hipSuccess) e = set_lds(cf::k_fused_decode_g<4, 1>, cf::FusedGeom<4, 1>::LDS_BYTES);
            if (e == hipSuccess) e = set_lds(cf::k_fused_decode_g<4, 4>, cf::FusedGeom<4, 4>::LDS_BYTES);
            if (e == hipSuccess) e = set_lds(cf::k_fused_decode_g<2, 4>, cf::FusedGeom<2, 4>::LDS_BYTES);
            if (e == hipSuccess) e = set_lds(cf::k_fused_decode_g<1, 4>, cf::FusedGeom<1, 4>::LDS_BYTES);
            if (e == hipSuccess) e = set_lds(cf::k_fused_decode_s<4>, cf::ShardGeom<4>::LDS_BYTES);
            if (e == hipSuccess) e = set_lds(cf::k_fused_decode_s<8>, cf::ShardGeom<8>::LDS_BYTES);
            if (e != hipSuccess) return fail(CF_ELAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
            attr_devs |= 1ull << cur_dev;
        }
        // (no planning from the sequence length: the persistent kernels read it on the device and pick their straight-line or
        //  loop arm there -- a graph captured once serves a growing sequence, and a stale max_seq_len cannot drop tokens)
        cf::FusedArgs fa;
        fill_fused_args(fa);
        fill_p1_table(fa);      // (S <= 1024: the kernel deals equal shares itself, from the device-side length)
        g_last_path = CF_PATH_FUSED;
        ProfScope prof(st);
        const bool io = a->weight_layout == CF_W_IN_OUT;
        const dim3 grid(cf::FUSED_WGS), block(cf::FUSED_THREADS);
        bool launched = false;
        auto launch_fused = [&](auto kern, int lds, const char* name) {
            if (!fused_resident(kern, lds)) return false;      // cannot be co-resident on this device: stage pipeline
            hipLaunchKernelGGL(kern, grid, block, lds, st, fa);
            g_last_variant = name;
            return true;
        };
        if (kind == FK_GQA_32_8) {
            constexpr int LB = cf::FusedGeom<8, 4>::LDS_BYTES;
            launched = launch_fused(cf::k_fused_decode_g<8, 4>, LB, "k_fused_decode_g<8, 4>");
        } else if (kind == FK_GQA_16_4) {
            launched = launch_fused(cf::k_fused_decode_g<4, 4>, cf::FusedGeom<4, 4>::LDS_BYTES, "k_fused_decode_g<4, 4>");
        } else if (kind == FK_GQA_8_2) {
            launched = launch_fused(cf::k_fused_decode_g<2, 4>, cf::FusedGeom<2, 4>::LDS_BYTES, "k_fused_decode_g<2, 4>");
        } else if (kind == FK_GQA_4_1) {
            launched = launch_fused(cf::k_fused_decode_g<1, 4>, cf::FusedGeom<1, 4>::LDS_BYTES, "k_fused_decode_g<1, 4>");
        } else if (kind == FK_MHA16) {
            constexpr int LB = cf::FusedGeom<16, 1>::LDS_BYTES;
            launched = launch_fused(cf::k_fused_decode_g<16, 1>, LB, "k_fused_decode_g<16, 1>");
        } else if (kind == FK_MHA8) {
            // (debug bit 128 / 256: the geometry-generic kernel instead of the role-split shard kernel, cf_fused_kernel_s.h)
            if (g_flags & 256) launched = launch_fused(cf::k_fused_decode_s<8>, cf::ShardGeom<8>::LDS_BYTES, "k_fused_decode_s<8>");
            else launched = launch_fused(cf::k_fused_decode_g<8, 1>, cf::FusedGeom<8, 1>::LDS_BYTES, "k_fused_decode_g<8, 1>");
        } else if (kind == FK_MHA4) {
            // The role-split kernel streams the K/V of a head on 16 CUs: the better plan up to ~12 k cached tokens (12.9 vs 13.3-14.4 us
            // at 4096, 15.8 vs 17.1 at 8192, 22.0 vs 19.0 at 16384).  The length is a device-side value; the host only has a
            // bound (the exact length of a contiguous cache, the caller's max_seq_len hint of a paged one -- unknown = long): a
            // sequence that outgrows its hint stays correct (the kernel's loop arm), it just keeps the plan chosen here.
            const long long bound = paged ? (a->max_seq_len > 0 ? a->max_seq_len : (1ll << 40)) : a->seq_len;
            if ((g_flags & 128) || bound > 8192) launched = launch_fused(cf::k_fused_decode_g<4, 1>, cf::FusedGeom<4, 1>::LDS_BYTES, "k_fused_decode_g<4, 1>");
            else launched = launch_fused(cf::k_fused_decode_s<4>, cf::ShardGeom<4>::LDS_BYTES, "k_fused_decode_s<4>");
        } else if (io) launched = launch_fused(cf::k_fused_decode_mha<true>, cf::FUSED_LDS_BYTES, "k_fused_decode_mha<IO=true>");
        else launched = launch_fused(cf::k_fused_decode_mha<false>, cf::FUSED_LDS_BYTES, "k_fused_decode_mha<IO=false>");
        if (launched) {
            prof.mark();
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return fail(CF_ELAUNCH, "HIP launch failed: %s", hipGetErrorString(e));
            if (tp_publish) tp_note_publish(a->tp_areas[a->tp_rank], d.hidden);
            return CF_OK;
        }
        if (g_path == CF_PATH_FUSED || tp_publish) return fail(CF_EUNSUPPORTED, "fused path requested but 256 workgroups cannot be co-resident on this device");
        prof.on = false;      // (nothing was launched: the pipeline below records its own stages)
    }

    // ---- 2 .. 4 sequences: the rows ride one weight stream of the persistent kernel (cf_fused_kernel_b.h) -----------------
    if (g_path != CF_PATH_PIPELINE && !(g_flags & 32) && small_batch_shape &&
        // (rows up to MAX_TOKENS run straight-line, longer ones loop on 8 / rows CUs per head: beyond a few times that the stage
        //  pipeline, which spreads one row over more CUs, is the better plan; an unknown bound takes the kernel)
        (g_path == CF_PATH_FUSED || a->max_seq_len <= 16 * (a->batch == 2 ? cf::FusedBGeom<2>::MAX_TOKENS : cf::FusedBGeom<4>::MAX_TOKENS)) &&
        device_cus() >= cf::FUSED_WGS) {
        static thread_local unsigned long long attr_devs_b = 0;
        int cur_dev = 0;
        if (hipGetDevice(&cur_dev) != hipSuccess || cur_dev < 0 || cur_dev > 63) cur_dev = 63;
        if (cur_dev == 63 || !((attr_devs_b >> cur_dev) & 1ull)) {
            hipError_t e = set_lds(cf::k_fused_decode_mhab<2>, cf::FusedBGeom<2>::LDS_BYTES);
            if (e == hipSuccess) e = set_lds(cf::k_fused_decode_mhab<4>, cf::FusedBGeom<4>::LDS_BYTES);
            if (e != hipSuccess) return fail(CF_ELAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
            if (cur_dev != 63) attr_devs_b |= 1ull << cur_dev;
        }
        cf::FusedArgs fa;
        fill_fused_args(fa);
        fa.g_qkv = ws.g_qkv_io;      // [rows][32][384] granules: the [in,out] kernel's split-K area is free in this layout
        fa.g_attn = ws.g_part;       // [rows][4096]
        fill_p1_shares(fa.p1_start, /*flat=*/true);
        ProfScope prof(st);
        const dim3 grid(cf::FUSED_WGS), block(cf::FUSED_THREADS);
        bool launched = false;
        if (a->batch == 2) {
            if ((launched = fused_resident(cf::k_fused_decode_mhab<2>, cf::FusedBGeom<2>::LDS_BYTES))) {
                hipLaunchKernelGGL(cf::k_fused_decode_mhab<2>, grid, block, cf::FusedBGeom<2>::LDS_BYTES, st, fa, a->batch);
                g_last_variant = "k_fused_decode_mhab<2>";
            }
        } else if ((launched = fused_resident(cf::k_fused_decode_mhab<4>, cf::FusedBGeom<4>::LDS_BYTES))) {
            hipLaunchKernelGGL(cf::k_fused_decode_mhab<4>, grid, block, cf::FusedBGeom<4>::LDS_BYTES, st, fa, a->batch);
            g_last_variant = "k_fused_decode_mhab<4>";
        }
        if (launched) {
            g_last_path = CF_PATH_FUSED;
            prof.mark();
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return fail(CF_ELAUNCH, "HIP launch failed: %s", hipGetErrorString(e));
            return CF_OK;
        }
        prof.on = false;
    }
    // ---- 2 .. 4 sequences of the 32q/8kv model: the rows ride the weight stream of the grouped-query kernel (cf_fused_kernel_gb.h) ----
    if (g_path != CF_PATH_PIPELINE && !(g_flags & 32) && gqa_batch_shape &&
        (g_path == CF_PATH_FUSED || a->max_seq_len <= 16 * (a->batch == 2 ? cf::FusedGBGeom<2>::MAX_TOKENS : cf::FusedGBGeom<4>::MAX_TOKENS)) &&
        device_cus() >= cf::FUSED_WGS) {
        static thread_local unsigned long long attr_devs_gb = 0;
        int cur_dev = 0;
        if (hipGetDevice(&cur_dev) != hipSuccess || cur_dev < 0 || cur_dev > 63) cur_dev = 63;
        if (cur_dev == 63 || !((attr_devs_gb >> cur_dev) & 1ull)) {
            hipError_t e = set_lds(cf::k_fused_decode_gb<2>, cf::FusedGBGeom<2>::LDS_BYTES);
            if (e == hipSuccess) e = set_lds(cf::k_fused_decode_gb<4>, cf::FusedGBGeom<4>::LDS_BYTES);
            if (e != hipSuccess) return fail(CF_ELAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
            if (cur_dev != 63) attr_devs_gb |= 1ull << cur_dev;
        }
        cf::FusedArgs fa;
        fill_fused_args(fa);
        fa.g_qkv = ws.g_qkv_io;      // [rows][8 kv heads][768] granules (the [in,out] kernels' split-K area: free in this layout)
        fa.g_attn = ws.g_part;       // [8][rows][256] granules
        ProfScope prof(st);
        const dim3 grid(cf::FUSED_WGS), block(cf::FUSED_THREADS);
        bool launched = false;
        if (a->batch == 2) {
            if ((launched = fused_resident(cf::k_fused_decode_gb<2>, cf::FusedGBGeom<2>::LDS_BYTES))) {
                hipLaunchKernelGGL(cf::k_fused_decode_gb<2>, grid, block, cf::FusedGBGeom<2>::LDS_BYTES, st, fa, a->batch);
                g_last_variant = "k_fused_decode_gb<2>";
            }
        } else if ((launched = fused_resident(cf::k_fused_decode_gb<4>, cf::FusedGBGeom<4>::LDS_BYTES))) {
            hipLaunchKernelGGL(cf::k_fused_decode_gb<4>, grid, block, cf::FusedGBGeom<4>::LDS_BYTES, st, fa, a->batch);
            g_last_variant = "k_fused_decode_gb<4>";
        }
        if (launched) {
            g_last_path = CF_PATH_FUSED;
            prof.mark();
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return fail(CF_ELAUNCH, "HIP launch failed: %s", hipGetErrorString(e));
            return CF_OK;
        }
        prof.on = false;
    }
    // ---- 5 .. 32 sequences: one persistent launch, projections on the matrix cores (cf_fused_kernel_q.h) -------------------------
    // AUTO keeps the persistent kernel up to 29 rows: at 30 and 32 rows the five launches measured 1-3 % faster for every S = 512 ..
    // 4096 (16 .. 28 rows: the kernel by 1-17 %; profiles/r06_batch_routing.md).  CF_PATH_FUSED still takes it up to 32 rows.
    constexpr int ROWS_Q_AUTO_MAX = 29;
    if (g_path != CF_PATH_PIPELINE && !(g_flags & 32) && rows_q_shape && (g_path == CF_PATH_FUSED || a->batch <= ROWS_Q_AUTO_MAX) &&
        device_cus() >= cf::FUSED_WGS) {
        static thread_local unsigned long long attr_devs_q = 0;
        int cur_dev = 0;
        if (hipGetDevice(&cur_dev) != hipSuccess || cur_dev < 0 || cur_dev > 63) cur_dev = 63;
        if (cur_dev == 63 || !((attr_devs_q >> cur_dev) & 1ull)) {
            hipError_t e = set_lds(cf::k_fused_decode_mhaq<1>, cf::FusedQGeomT<1>::LDS_BYTES);
            if (e == hipSuccess) e = set_lds(cf::k_fused_decode_mhaq<2>, cf::FusedQGeomT<2>::LDS_BYTES);
            if (e != hipSuccess) return fail(CF_ELAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
            if (cur_dev != 63) attr_devs_q |= 1ull << cur_dev;
        }
        const bool two_tiles = a->batch > 16;      // 17 .. 32 rows: two 16-row batch tiles in the MFMA operand
        if (two_tiles ? fused_resident(cf::k_fused_decode_mhaq<2>, cf::FusedQGeomT<2>::LDS_BYTES)
                      : fused_resident(cf::k_fused_decode_mhaq<1>, cf::FusedQGeomT<1>::LDS_BYTES)) {
            cf::FusedArgs fa;
            fill_fused_args(fa);
            fill_q_shares(fa.p1_start, a->batch);
            fa.g_qkv = ws.g_bqkv;        // [rows][32][384] granules
            fa.g_attn = ws.g_battn;      // [rows][4096] fp16 payload + [rows][32] flag granules
            fa.g_rec = ws.g_brec;        // [rows][32][8][66] granules
            fa.g_qkv_io = ws.g_bxn;      // [rows][4096] fp16 + [rows] flag granules: X0 has its own region (it used to alias the [in,out] kernels' split-K area)
            ProfScope prof(st);
            if (two_tiles) hipLaunchKernelGGL(cf::k_fused_decode_mhaq<2>, dim3(cf::FUSED_WGS), dim3(cf::FUSED_THREADS), cf::FusedQGeomT<2>::LDS_BYTES, st, fa, a->batch);
            else hipLaunchKernelGGL(cf::k_fused_decode_mhaq<1>, dim3(cf::FUSED_WGS), dim3(cf::FUSED_THREADS), cf::FusedQGeomT<1>::LDS_BYTES, st, fa, a->batch);
            g_last_variant = "k_fused_decode_mhaq";
            g_last_path = CF_PATH_FUSED;
            prof.mark();
            const hipError_t e = hipGetLastError();
            if (e != hipSuccess) return fail(CF_ELAUNCH, "HIP launch failed: %s", hipGetErrorString(e));
            return CF_OK;
        }
    }
    if (g_path == CF_PATH_FUSED)
        return fail(CF_EUNSUPPORTED, "fused path requested but 256 workgroups cannot be co-resident on this device (or debug flag 32 is set)");

    g_last_path = CF_PATH_PIPELINE;
    g_last_variant = "stage pipeline";
    cf::ResidualOut ro{(const cf::h16*)a->x, (const cf::h16*)a->residual, (cf::h16*)a->residual_out, d.hidden};
    ProfScope prof(st);

    // ---- stage 0: RMSNorm + QKV projection --------------------------------------------------------
    // batch > 1 with [out,in] weights: weight-streaming MFMA GEMMs (each weight byte read once for all rows)
    const bool batch_mfma = a->batch >= 2 && a->weight_layout == CF_W_OUT_IN && d.hidden <= 5120 &&
                            d.n_q_heads * d.head_dim <= 5120;   // (wider: the per-row kernels below)
    if (batch_mfma) {
        hipLaunchKernelGGL(cf::k_norm_rows, dim3(a->batch), dim3(256), 0, st, na, ws.xn16, (cf::h16*)nullptr);
        cf::ProjArgs pa{};
        pa.W = (const cf::h16*)a->weight_qkv;
        pa.n_rows = qkv_dim;
        pa.K = d.hidden;
        pa.batch = a->batch;
        pa.in = ws.xn16;
        pa.out_f32 = ws.qkv_raw;
        if (!launch_proj_mfma(pa, ro, false, st))
            return fail(CF_EUNSUPPORTED, "hidden %d: no MFMA projection", d.hidden);
    } else if (a->weight_layout == CF_W_OUT_IN) {
        const cf::h16* W = (const cf::h16*)a->weight_qkv;
        switch (J_in) {
            case 1: launch_qkv_rows<1>(na, W, qkv_dim, a->batch, ws.qkv_raw, st); break;
            case 2: launch_qkv_rows<2>(na, W, qkv_dim, a->batch, ws.qkv_raw, st); break;
            case 4: launch_qkv_rows<4>(na, W, qkv_dim, a->batch, ws.qkv_raw, st); break;
            case 8: launch_qkv_rows<8>(na, W, qkv_dim, a->batch, ws.qkv_raw, st); break;
            case 10: launch_qkv_rows<10>(na, W, qkv_dim, a->batch, ws.qkv_raw, st); break;
            default: launch_qkv_rows<16>(na, W, qkv_dim, a->batch, ws.qkv_raw, st); break;
        }
    } else {
        const int C = d.n_q_heads * d.head_dim;
        hipLaunchKernelGGL((cf::k_qkv_cols<16>), dim3((C / 512) * 3 * ksplit, a->batch), dim3(256), 0, st, na,
                           (const cf::h16*)a->weight_qkv, C, ksplit, ws.qkv_raw);
    }
    prof.mark();

    // ---- stage 1: attention -----------------------------------------------------------------------
    cf::AttnArgs aa;
    aa.qkv_raw = ws.qkv_raw;
    aa.ksplit = ksplit;
    aa.qkv_dim = qkv_dim;
    aa.Hq = d.n_q_heads;
    aa.Hkv = d.n_kv_heads;
    aa.k_cache = (const cf::h16*)a->k_cache;
    aa.v_cache = (const cf::h16*)a->v_cache;
    aa.kptrs = paged ? a->kv_cache_ptrs_k : nullptr;
    aa.vptrs = paged ? a->kv_cache_ptrs_v : nullptr;
    aa.layer_id = a->layer_id;
    aa.seq_len = (int)a->seq_len;
    aa.indptr = a->kv_indptr;
    aa.indices = a->kv_indices;
    aa.seq_lens = a->kv_seq_lens;
    aa.page_shift = page_shift;
    aa.cos = a->cos;
    aa.sin = a->sin;
    aa.positions = a->positions;
    aa.rope_stride = a->rope_row_stride;
    aa.rope_style = a->rope_style;
    aa.nsplit = nsplit;
    aa.tokens_per_split = 0;   // derived on device from each row's length
    aa.part_o = ws.part_o;
    aa.part_ml = ws.part_ml;
    aa.attn_out = ws.attn;
    aa.attn_h16 = batch_mfma ? ws.attn16 : nullptr;
    aa.k_new = (cf::h16*)a->k_new;
    aa.v_new = (cf::h16*)a->v_new;
    aa.write_cache = paged ? a->write_kv_to_cache : 0;
    switch (G) {
        case 1: launch_attn<1>(aa, a->batch, st); break;
        case 2: launch_attn<2>(aa, a->batch, st); break;
        case 4: launch_attn<4>(aa, a->batch, st); break;
        default: launch_attn<8>(aa, a->batch, st); break;
    }
    prof.mark();

    // ---- stage 2 (+3): merge + O projection -------------------------------------------------------
    cf::MergeArgs mrg{ws.part_o, ws.part_ml, nsplit, d.n_q_heads, batch_mfma ? ws.attn16 : nullptr};
    if (nsplit > 1)            // (one split per head: the attention kernel wrote the merged output itself)
        hipLaunchKernelGGL(cf::k_attn_merge, dim3((d.n_q_heads * cf::HEAD_DIM + 255) / 256, a->batch), dim3(256), 0, st,
                           mrg, ws.attn);
    const float* ma = ws.attn;
    if (batch_mfma) {
        cf::ProjArgs pa{};
        pa.W = (const cf::h16*)a->weight_o;
        pa.n_rows = d.hidden;
        pa.K = d.n_q_heads * d.head_dim;
        pa.batch = a->batch;
        pa.in = ws.attn16;
        pa.out_h16 = (cf::h16*)a->out;
        if (!launch_proj_mfma(pa, ro, true, st)) return fail(CF_EUNSUPPORTED, "n_q_heads %d: no MFMA projection", d.n_q_heads);
        prof.mark();
    } else if (a->weight_layout == CF_W_OUT_IN) {
        const cf::h16* Wo = (const cf::h16*)a->weight_o;
        cf::h16* out = (cf::h16*)a->out;
        switch (J_o) {
            case 1: launch_oproj_rows<1>(ma, Wo, d.hidden, a->batch, out, ro, st); break;
            case 2: launch_oproj_rows<2>(ma, Wo, d.hidden, a->batch, out, ro, st); break;
            case 4: launch_oproj_rows<4>(ma, Wo, d.hidden, a->batch, out, ro, st); break;
            case 8: launch_oproj_rows<8>(ma, Wo, d.hidden, a->batch, out, ro, st); break;
            case 10: launch_oproj_rows<10>(ma, Wo, d.hidden, a->batch, out, ro, st); break;
            default: launch_oproj_rows<16>(ma, Wo, d.hidden, a->batch, out, ro, st); break;
        }
        prof.mark();
    } else {
        hipLaunchKernelGGL((cf::k_oproj_cols<16>), dim3((d.hidden / 512) * d.n_q_heads, a->batch), dim3(256), 0, st,
                           ma, d.n_q_heads, (const cf::h16*)a->weight_o, d.hidden, ws.opart);
        prof.mark();
        hipLaunchKernelGGL(cf::k_reduce_heads, dim3((d.hidden + 255) / 256, a->batch), dim3(256), 0, st, ws.opart,
                           d.n_q_heads, d.hidden, (cf::h16*)a->out, ro);
        prof.mark();
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CF_ELAUNCH, "HIP launch failed: %s", hipGetErrorString(e));
    return CF_OK;
}

size_t cf_tp_oneshot_bytes(int32_t world, int32_t n) {
    if (world < 1 || world > cf::TP_MAX_WORLD || n <= 0 || n % 2) return 0;
    return (size_t)(cf::TP_HDR_GRANULES + 2 * (size_t)world * (n / 2)) * 8;      // two slot sets (epoch parity)
}

int cf_tp_area_alloc(size_t bytes, void** area) {
    if (!area || bytes == 0) return fail(CF_EINVAL, "cf_tp_area_alloc: NULL / empty");
    void* p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) return fail(CF_ELAUNCH, "cf_tp_area_alloc: hipExtMallocWithFlags(finegrained, %zu): %s", bytes, hipGetErrorString(e));
    e = hipMemset(p, 0, bytes);
    if (e == hipSuccess && bytes >= 24) {      // words 4..5: device address of this device's host-mapped failure word (tp_flag_error)
        uint32_t* sticky = cf::api_sticky_device_pointer();
        if (sticky) e = hipMemcpy(static_cast<char*>(p) + 16, &sticky, sizeof(sticky), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        (void)hipFree(p);
        return fail(CF_ELAUNCH, "cf_tp_area_alloc: clearing the area: %s", hipGetErrorString(e));
    }
    *area = p;
    return CF_OK;
}

int cf_tp_area_free(void* area) {
    if (!area) return CF_OK;
    tp_forget(area);      // (an address the allocator hands out again must not inherit a pending publish)
    const hipError_t e = hipFree(area);
    return e == hipSuccess ? CF_OK : fail(CF_ELAUNCH, "cf_tp_area_free: %s", hipGetErrorString(e));
}

int cf_tp_area_export(void* area, void* handle) {
    static_assert(sizeof(hipIpcMemHandle_t) == CF_TP_HANDLE_BYTES, "CF_TP_HANDLE_BYTES");
    if (!area || !handle) return fail(CF_EINVAL, "cf_tp_area_export: NULL argument");
    hipIpcMemHandle_t h;
    const hipError_t e = hipIpcGetMemHandle(&h, area);
    if (e != hipSuccess) return fail(CF_ELAUNCH, "cf_tp_area_export: hipIpcGetMemHandle: %s", hipGetErrorString(e));
    memcpy(handle, &h, sizeof(h));
    return CF_OK;
}

int cf_tp_area_import(const void* handle, void** mapped) {
    if (!handle || !mapped) return fail(CF_EINVAL, "cf_tp_area_import: NULL argument");
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    void* p = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return fail(CF_ELAUNCH, "cf_tp_area_import: hipIpcOpenMemHandle: %s", hipGetErrorString(e));
    *mapped = p;
    return CF_OK;
}

int cf_tp_area_unmap(void* mapped) {
    if (!mapped) return CF_OK;
    const hipError_t e = hipIpcCloseMemHandle(mapped);
    return e == hipSuccess ? CF_OK : fail(CF_ELAUNCH, "cf_tp_area_unmap: %s", hipGetErrorString(e));
}

int cf_tp_area_status(const void* area, void* stream, uint32_t* code) {
    if (!area || !code) return fail(CF_EINVAL, "cf_tp_area_status: NULL argument");
    hipError_t e = hipStreamSynchronize(static_cast<hipStream_t>(stream));
    if (e == hipSuccess) e = hipMemcpy(code, static_cast<const uint32_t*>(area) + 1, sizeof(uint32_t), hipMemcpyDeviceToHost);
    return e == hipSuccess ? CF_OK : fail(CF_ELAUNCH, "cf_tp_area_status: %s", hipGetErrorString(e));
}

int cf_tp_oneshot_allreduce(const void* partial, void* out, int32_t n, int32_t rank, int32_t world, void* const* areas,
                            int32_t flags, void* stream) {
    if (!partial || !out || !areas) return fail(CF_EINVAL, "cf_tp_oneshot_allreduce: NULL argument");
    if (world < 1 || world > cf::TP_MAX_WORLD || rank < 0 || rank >= world) return fail(CF_EINVAL, "cf_tp_oneshot_allreduce: rank %d / world %d", rank, world);
    if (n <= 0 || n % 2 || n > (1 << 20)) return fail(CF_EINVAL, "cf_tp_oneshot_allreduce: n %d (even, <= 2^20)", n);
    cf::TpOneShotArgs a;
    memset(&a, 0, sizeof(a));
    a.partial = (const cf::h16*)partial;
    a.out = (cf::h16*)out;
    for (int p = 0; p < world; ++p) {
        if (!areas[p] || (reinterpret_cast<uintptr_t>(areas[p]) & 255)) return fail(CF_EINVAL, "cf_tp_oneshot_allreduce: area %d NULL or not 256-byte aligned", p);
        a.areas[p] = (unsigned long long*)areas[p];
    }
    a.n = n;
    a.rank = rank;
    a.world = world;
    a.flags = flags;
    tp_forget(areas[rank]);      // (an explicit publish supersedes whatever a layer kernel was noted to have published here)
    hipLaunchKernelGGL(cf::k_tp_oneshot_allreduce, dim3((n / 2 + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CF_ELAUNCH, "HIP launch failed: %s", hipGetErrorString(e));
    return CF_OK;
}

static int tp_check_areas(const char* who, int32_t rank, int32_t world, void* const* areas) {
    if (!areas) return fail(CF_EINVAL, "%s: NULL areas", who);
    if (world < 1 || world > cf::TP_MAX_WORLD || rank < 0 || rank >= world) return fail(CF_EINVAL, "%s: rank %d / world %d", who, rank, world);
    for (int p = 0; p < world; ++p)
        if (!areas[p] || (reinterpret_cast<uintptr_t>(areas[p]) & 255)) return fail(CF_EINVAL, "%s: area %d NULL or not 256-byte aligned", who, p);
    return CF_OK;
}

int cf_tp_gather(void* out, int32_t n, int32_t rank, int32_t world, void* const* areas, void* stream) {
    if (!out) return fail(CF_EINVAL, "cf_tp_gather: NULL out");
    if (const int rc = tp_check_areas("cf_tp_gather", rank, world, areas)) return rc;
    if (n <= 0 || n % 2 || n > (1 << 20)) return fail(CF_EINVAL, "cf_tp_gather: n %d (even, <= 2^20)", n);
    if (const int published = tp_match_gather(areas[rank], n))
        return fail(CF_EINVAL, "cf_tp_gather: n %d, but the layer kernel published n = %d values into these areas (the granule "
                    "layout depends on n); nothing was launched", n, published);
    cf::TpOneShotArgs a;
    memset(&a, 0, sizeof(a));
    a.partial = nullptr;
    a.out = (cf::h16*)out;
    for (int p = 0; p < world; ++p) a.areas[p] = (unsigned long long*)areas[p];
    a.n = n;
    a.rank = rank;
    a.world = world;
    a.flags = 2;      // gather only
    hipLaunchKernelGGL(cf::k_tp_oneshot_allreduce, dim3((n / 2 + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CF_ELAUNCH, "HIP launch failed: %s", hipGetErrorString(e));
    return CF_OK;
}

int cf_rmsnorm_tp_gather(void* const* areas, int32_t rank, int32_t world, const void* residual, const void* weight, float eps,
                         int32_t hidden, void* out, void* residual_out, void* sum_out, void* stream) {
    if (!weight || !out) return fail(CF_EINVAL, "cf_rmsnorm_tp_gather: NULL pointer");
    if (const int rc = tp_check_areas("cf_rmsnorm_tp_gather", rank, world, areas)) return rc;
    if (hidden < 512 || hidden > 8192 || hidden % 512) return fail(CF_EUNSUPPORTED, "cf_rmsnorm_tp_gather: hidden %d (512 .. 8192, multiple of 512)", hidden);
    if (residual_out && !residual) return fail(CF_EINVAL, "cf_rmsnorm_tp_gather: residual_out without residual");
    if (const int published = tp_match_gather(areas[rank], hidden))
        return fail(CF_EINVAL, "cf_rmsnorm_tp_gather: hidden %d, but the layer kernel published n = %d values into these areas (the "
                    "granule layout depends on n); nothing was launched", hidden, published);
    cf::TpNormArgs a;
    memset(&a, 0, sizeof(a));
    for (int p = 0; p < world; ++p) a.areas[p] = (unsigned long long*)areas[p];
    a.rank = rank;
    a.world = world;
    a.hidden = hidden;
    a.residual = (const cf::h16*)residual;
    a.weight = (const cf::h16*)weight;
    a.eps = eps;
    a.out = (cf::h16*)out;
    a.residual_out = (cf::h16*)residual_out;
    a.sum_out = (cf::h16*)sum_out;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // hidden 4096 / 8192 with more than one rank: eight workgroups, an eighth of the row each (cf_tp_kernels.h; one workgroup polling
    // world x hidden / 2 slots through one CU cost more than gather and norm launched separately).  Debug bit 4096: the one-workgroup kernel.
    if (world > 1 && (hidden == 4096 || hidden == 8192) && !(g_flags & 4096)) {
        if (hidden == 4096) hipLaunchKernelGGL(cf::k_rmsnorm_tp_gather_mw<1>, dim3(cf::TP_NORM_WGS), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(cf::k_rmsnorm_tp_gather_mw<2>, dim3(cf::TP_NORM_WGS), dim3(256), 0, st, a);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(CF_ELAUNCH, "HIP launch failed: %s", hipGetErrorString(e));
        return CF_OK;
    }
    switch (hidden / 512) {
#define CF_TPN(P) case P: hipLaunchKernelGGL(cf::k_rmsnorm_tp_gather<P>, dim3(1), dim3(256), 0, st, a); break;
        CF_TPN(1) CF_TPN(2) CF_TPN(3) CF_TPN(4) CF_TPN(5) CF_TPN(6) CF_TPN(7) CF_TPN(8) CF_TPN(9) CF_TPN(10) CF_TPN(11) CF_TPN(12)
        CF_TPN(13) CF_TPN(14) CF_TPN(15) CF_TPN(16)
#undef CF_TPN
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CF_ELAUNCH, "HIP launch failed: %s", hipGetErrorString(e));
    return CF_OK;
}

int cf_tp_area_clear_error(void* area, void* stream) {
    if (!area) return fail(CF_EINVAL, "cf_tp_area_clear_error: NULL area");
    tp_forget(area);
    const hipError_t e = hipMemsetAsync(static_cast<char*>(area) + 4, 0, 4, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? CF_OK : fail(CF_ELAUNCH, "cf_tp_area_clear_error: %s", hipGetErrorString(e));
}

int cf_rmsnorm(const void* input, const void* residual, const void* weight, float eps, int32_t rows, int32_t hidden,
               void* out, void* residual_out, void* stream) {
    if (!input || !weight || !out) return fail(CF_EINVAL, "cf_rmsnorm: NULL pointer");
    if (rows <= 0) return fail(CF_EINVAL, "cf_rmsnorm: rows %d", rows);
    if (hidden <= 0 || hidden % 8 || hidden > 8192) return fail(CF_EUNSUPPORTED, "cf_rmsnorm: hidden %d (multiple of 8, <= 8192)", hidden);
    if (residual_out && !residual) return fail(CF_EINVAL, "cf_rmsnorm: residual_out without residual");
    cf::NormArgs na{(const cf::h16*)input, (const cf::h16*)residual, (const cf::h16*)weight, eps, hidden};
    hipLaunchKernelGGL(cf::k_norm_rows, dim3(rows), dim3(256), 0, static_cast<hipStream_t>(stream), na, (cf::h16*)out,
                       (cf::h16*)residual_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(CF_ELAUNCH, "HIP launch failed: %s", hipGetErrorString(e));
    return CF_OK;
}

static const cf_dims kLlama2_7B = {4096, 32, 32, 128};

int cf_llama_decoder_layer(const void* input, const void* weight_qkv, const void* weight_o, const void* k_cache,
                           const void* v_cache, int64_t seq_len, const void* rms_input_weight, const float* cos,
                           const float* sin, void* out, void* k_new, void* v_new, void* workspace,
                           size_t workspace_bytes, void* stream) {
    cf_layer_args a;
    memset(&a, 0, sizeof(a));
    a.dims = kLlama2_7B;
    a.batch = 1;
    a.weight_layout = CF_W_IN_OUT;
    a.rope_style = CF_ROPE_GPTJ;
    a.eps = 1e-6f;   // hard-coded in the reference kernel (kernel.cuh:58)
    a.x = input;
    a.weight_qkv = weight_qkv;
    a.weight_o = weight_o;
    a.rms_weight = rms_input_weight;
    a.k_cache = k_cache;
    a.v_cache = v_cache;
    a.seq_len = seq_len;
    a.page_size = 1;
    a.cos = cos;
    a.sin = sin;
    a.out = out;
    a.k_new = k_new;
    a.v_new = v_new;
    a.workspace = workspace;
    a.workspace_bytes = workspace_bytes;
    a.stream = stream;
    return cf_decoder_layer_ex(&a);
}

int cf_llama_decoder_layer_out_in(const void* input, const void* weight_qkv_out_in, const void* weight_o_out_in, const void* k_cache,
                                  const void* v_cache, int64_t seq_len, const void* rms_input_weight, const float* cos,
                                  const float* sin, void* out, void* k_new, void* v_new, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    cf_layer_args a;
    memset(&a, 0, sizeof(a));
    a.dims = kLlama2_7B;
    a.batch = 1;
    a.weight_layout = CF_W_OUT_IN;      // (the only difference from cf_llama_decoder_layer: weights from cf_relayout_weights)
    a.rope_style = CF_ROPE_GPTJ;
    a.eps = 1e-6f;
    a.x = input;
    a.weight_qkv = weight_qkv_out_in;
    a.weight_o = weight_o_out_in;
    a.rms_weight = rms_input_weight;
    a.k_cache = k_cache;
    a.v_cache = v_cache;
    a.seq_len = seq_len;
    a.page_size = 1;
    a.cos = cos;
    a.sin = sin;
    a.out = out;
    a.k_new = k_new;
    a.v_new = v_new;
    a.workspace = workspace;
    a.workspace_bytes = workspace_bytes;
    a.stream = stream;
    return cf_decoder_layer_ex(&a);
}

int cf_llama_decoder_layer_sglang(const void* input, void* residual, const void* weight_qkv, const void* weight_o,
                                  const void* k_cache, const void* v_cache, int64_t seq_len,
                                  const void* rms_input_weight, float eps, const float* cos, const float* sin,
                                  void* out, void* k_new, void* v_new, void* workspace, size_t workspace_bytes,
                                  void* stream) {
    cf_layer_args a;
    memset(&a, 0, sizeof(a));
    a.dims = kLlama2_7B;
    a.batch = 1;
    a.weight_layout = CF_W_OUT_IN;
    a.rope_style = CF_ROPE_NEOX;
    a.eps = eps;
    a.x = input;
    a.residual = residual;
    a.residual_out = residual;   // in place (kernel_sglang.cuh:99-105), written race-free by the last stage
    a.weight_qkv = weight_qkv;
    a.weight_o = weight_o;
    a.rms_weight = rms_input_weight;
    a.k_cache = k_cache;
    a.v_cache = v_cache;
    a.seq_len = seq_len;
    a.page_size = 1;
    a.cos = cos;
    a.sin = sin;
    a.out = out;
    a.k_new = k_new;
    a.v_new = v_new;
    a.workspace = workspace;
    a.workspace_bytes = workspace_bytes;
    a.stream = stream;
    return cf_decoder_layer_ex(&a);
}

int cf_llama_decoder_layer_batch_decode_sglang(void* output, void* residual_output, const void* input,
                                               const void* residual, const void* weight_qkv, const void* weight_o,
                                               const int32_t* paged_kv_indptr, const int32_t* paged_kv_indices,
                                               const uint64_t* k_cache_ptrs, const uint64_t* v_cache_ptrs,
                                               int32_t layer_id, const void* rms_input_weight, float eps,
                                               const int64_t* positions, const float* cos_sin, int32_t batch,
                                               int64_t max_seq_len, void* workspace, size_t workspace_bytes,
                                               void* stream) {
    if (!cos_sin || !positions) return fail(CF_EINVAL, "cos_sin / positions is NULL");
    cf_layer_args a;
    memset(&a, 0, sizeof(a));
    a.dims = kLlama2_7B;
    a.batch = batch;
    a.weight_layout = CF_W_OUT_IN;
    a.rope_style = CF_ROPE_NEOX;
    a.eps = eps;
    a.x = input;
    a.residual = residual;
    a.residual_out = residual_output;
    a.weight_qkv = weight_qkv;
    a.weight_o = weight_o;
    a.rms_weight = rms_input_weight;
    a.kv_cache_ptrs_k = k_cache_ptrs;
    a.kv_cache_ptrs_v = v_cache_ptrs;
    a.layer_id = layer_id;
    a.page_size = 1;
    a.kv_indptr = paged_kv_indptr;
    a.kv_indices = paged_kv_indices;
    a.max_seq_len = max_seq_len;
    a.cos = cos_sin;                      // row p = cat(cos[64], sin[64])  (kernel_batch_sglang.cuh:322-323)
    a.sin = cos_sin + cf::HEAD_DIM / 2;
    a.positions = positions;
    a.rope_row_stride = cf::HEAD_DIM;
    a.out = output;
    a.write_kv_to_cache = 1;
    a.workspace = workspace;
    a.workspace_bytes = workspace_bytes;
    a.stream = stream;
    return cf_decoder_layer_ex(&a);
}

}  // extern "C"
