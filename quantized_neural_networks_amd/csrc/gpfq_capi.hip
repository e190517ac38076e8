// extern "C" boundary of libgpfq_hip.so (declared in include/gpfq.h).  Argument validation,
// path selection and error reporting; no allocation, no synchronisation.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <atomic>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "../../include/gpfq.h"
#include "gpfq_launch.hpp"

namespace {

thread_local char g_err[512] = "";
thread_local const char *g_dense_kernel = "";
thread_local hipEvent_t g_main_ev[2] = {nullptr, nullptr};

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(hipError_t e, const char *what)
{
    return fail(GPFQ_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
}

// The host alphabet as the by-value kernel arguments: A for up to 64 members (int8 indices, every kernel family), B beyond
// (65..GPFQ_MAX_ALPHABET members, int16 indices: the wavefront-per-neuron, wide, streaming and thread-per-neuron Gram kernels).
struct HostAlphabet {
    gpfq::AlphabetArg A;
    gpfq::AlphabetBig B;
    bool is_big = false;
    const gpfq::AlphabetBig *big() const { return is_big ? &B : nullptr; }
    size_t idx_bytes() const { return is_big ? 2 : 1; }
    // element offset into an index array of either width
    int8_t *at(void *qidx, int64_t elems) const { return qidx ? static_cast<int8_t *>(qidx) + elems * (int64_t)idx_bytes() : nullptr; }
};

// Copies the host alphabet into the kernel argument of its size class and classifies it.
int make_alphabet(const double *alphabet, int M, int zero_idx, HostAlphabet *H)
{
    if (!alphabet) return fail(GPFQ_ERR_INVALID_ARG, "alphabet is NULL");
    if (M < 1) return fail(GPFQ_ERR_INVALID_ARG, "alphabet size M=%d must be >= 1", M);
    if (M > GPFQ_MAX_ALPHABET)
        return fail(GPFQ_ERR_UNSUPPORTED, "alphabet size M=%d exceeds GPFQ_MAX_ALPHABET=%d", M, GPFQ_MAX_ALPHABET);
    if (zero_idx < -1 || zero_idx >= M) return fail(GPFQ_ERR_INVALID_ARG, "zero_idx=%d out of range", zero_idx);
    std::memset(&H->A, 0, sizeof(H->A));
    std::memset(&H->B, 0, sizeof(H->B));
    H->is_big = M > 64;
    bool asc = true;
    for (int k = 0; k < M; ++k) {
        if (H->is_big) H->B.a[k] = alphabet[k];
        else H->A.a[k] = alphabet[k];
        if (std::isnan(alphabet[k])) asc = false;
        if (k > 0 && !(alphabet[k - 1] <= alphabet[k])) asc = false;
    }
    H->A.M = H->B.M = M;
    H->A.zero_idx = H->B.zero_idx = zero_idx;
    H->A.ascending = H->B.ascending = asc ? 1 : 0;
    return GPFQ_OK;
}

}  // namespace

namespace gpfq {
void note_dense_kernel(const char *name) { g_dense_kernel = name; }
bool main_kernel_events(hipEvent_t *start, hipEvent_t *stop)
{
    *start = g_main_ev[0]; *stop = g_main_ev[1];
    return g_main_ev[0] != nullptr && g_main_ev[1] != nullptr;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, device): it is raised when a launch needs more than
// any launch before it on that device, not once per launch.
hipError_t ensure_dynamic_lds(const void *kernel, size_t bytes)
{
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, size_t> granted;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    size_t &have = granted[std::make_pair(kernel, dev)];
    if (bytes <= have) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) have = bytes;
    return e;
}
}

extern "C" {

int gpfq_version(void) { return 306; }   // 306: the two-pass median's workspace grew by its coarse histograms (gpfq_median_abs_workspace_bytes_for), gpfq_set_main_kernel_events' events ride on the kernel's dispatch, option blk_prep_norms (round 6); 305: device-resident layer alphabet, gpfq_quantize_dense_layer, gpfq_call_status, the block kernel's workspace grew by its alphabet block (round 6); 304: options blk_cluster / blk_cluster_nl / blk_cluster_map, larger workspaces for long rows (round 5); 303: gpfq_set_main_kernel_events (round 4); hip.load() checks it

const char *gpfq_last_dense_kernel(void) { return g_dense_kernel; }

int gpfq_set_main_kernel_events(void *start, void *stop)
{
    g_main_ev[0] = static_cast<hipEvent_t>(start);
    g_main_ev[1] = static_cast<hipEvent_t>(stop);
    return GPFQ_OK;
}

const char *gpfq_last_error(void) { return g_err; }

int gpfq_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    int ok = 0;
    for (int d = 0; d < n; ++d) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, d) == hipSuccess && std::strncmp(p.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
}

int gpfq_row_norms(const float *Xq, int64_t N, int64_t m, int64_t ld, float *nrm32, void *stream)
{
    if (N < 0 || m < 0) return fail(GPFQ_ERR_INVALID_ARG, "negative size N=%lld m=%lld", (long long)N, (long long)m);
    if (N == 0) return GPFQ_OK;
    if (!nrm32 || (!Xq && m > 0)) return fail(GPFQ_ERR_INVALID_ARG, "NULL pointer");
    if (ld < m) return fail(GPFQ_ERR_INVALID_ARG, "row pitch ld=%lld < m=%lld", (long long)ld, (long long)m);
    hipError_t e = gpfq::launch_row_norms(Xq, N, m, ld, nrm32, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? GPFQ_OK : hip_fail(e, "gpfq_row_norms");
}

static int resolve_path(int64_t m, int path)
{
    if (path == GPFQ_PATH_AUTO) return m <= GPFQ_ONCHIP_MAX_M ? GPFQ_PATH_ONCHIP : GPFQ_PATH_STREAM;
    return path;
}

// GPFQ_PATH_AUTO prefers the Gram path (N x N records + scalar recurrences, gpfq_quantize_neurons_gram) exactly where
// the Python binding's auto_path does: rows beyond GPFQ_GRAM_MIN_M samples (or half of that with many neurons), walks
// the records can hold, and no residual vectors requested.
static bool auto_wants_gram(int64_t N, int64_t m, int64_t C, bool want_u)
{
    const bool long_rows = m > GPFQ_GRAM_MIN_M || (m > GPFQ_GRAM_MIN_M / 2 && C * m >= 5000000);
    return !want_u && long_rows && N <= GPFQ_GRAM_MAX_N && m < (1LL << 30);
}

// on-chip workspace: [fallback counter, 64 B][RowStats x N][iteration records of the pipelined kernel]
static size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
static size_t onchip_stats_bytes(int64_t N) { return al256(64 + (size_t)N * sizeof(gpfq::RowStats)); }
static size_t onchip_workspace_bytes(int64_t N, int64_t m, int64_t C)
{
    const size_t p = gpfq::pipe_workspace_bytes(N, m), b = gpfq::blk_workspace_bytes(N, m, C);
    return onchip_stats_bytes(N) + (p > b ? p : b);
}

// AUTO -> Gram: [Gram workspace][uncertified flags i32 x C][streaming workspace of ONE neuron, for the rare reruns]
static size_t auto_gram_workspace_bytes(int64_t N, int64_t m, int64_t C)
{
    return al256(gpfq::gram_workspace_bytes(N, m, C)) + al256((size_t)C * sizeof(int32_t)) +
           gpfq::stream_workspace_bytes(N, m, 1, /*need_u=*/true);
}

size_t gpfq_workspace_bytes(int64_t N, int64_t m, int64_t C, int path)
{
    if (N < 0 || m < 0 || C < 0) return 0;
    size_t need = resolve_path(m, path) == GPFQ_PATH_ONCHIP ? onchip_workspace_bytes(N, m, C)
                                                             : gpfq::stream_workspace_bytes(N, m, C, /*need_u=*/true);
    if (path == GPFQ_PATH_AUTO && auto_wants_gram(N, m, C, false)) {
        const size_t g = auto_gram_workspace_bytes(N, m, C);
        if (g > need) need = g;
    }
    return need;
}

// Tuning / test hooks (process-wide, atomics: a call on another thread sees either the old or the new value of each).
// Results never depend on them.
static std::atomic<int> g_onchip_mode{1};      // 1 = certified (default), 0 = exact flow
static std::atomic<int> g_tile_steps{0};       // 0 = heuristic
static std::atomic<int> g_group_waves{0};      // 0 = heuristic
static std::atomic<int> g_lpn{0};              // 0 = heuristic, 1 = wave-per-neuron kernel, 16/32/64 = row-group kernel
static std::atomic<int> g_gram_slack_log2{0};  // Gram path: error bounds multiplied by 2^this (tests force the uncertified branch)
static std::atomic<int> g_wpn{0};              // wide kernel: wavefronts per neuron (0 = heuristic: only for rows > 2048)
static std::atomic<int> g_variant{0};          // bit 0: row-group kernel without the float64 copy of Xq in LDS; bit 1: wide kernel with LDS-staged rows
static std::atomic<int> g_pipe{-1};            // pipelined dense kernels: -1 = heuristic, 0 = never, 1 = one step per slot (gpfq_pipe.hip) whenever it
                                   // applies, 2 = blocks of steps per slot (gpfq_blk.hip) whenever it applies
static std::atomic<int> g_auto_gram{1};        // GPFQ_PATH_AUTO may take the Gram path (one stream synchronisation inside the call); 0: AUTO stays asynchronous
static std::atomic<int> g_conv_fused{1};       // conv channel loop: 3x3/stride-1 Gram matrices straight from the planes
static std::atomic<int> g_conv_planes_free{1}; // 7x7 / 2 layers read the NHWC activations themselves (gpfq_quantize_conv_channels_nhwc; 0: channel planes first)
static std::atomic<int> g_conv_nhwc{1};        // 3x3 / stride 1 / SAME layers straight from the NHWC activations (no channel-major copy)
static std::atomic<int> g_conv_strip{0};
static std::atomic<int> g_sync_errors{0};      // 1: gpfq_quantize_neurons / gpfq_quantize_dense_layer wait for their launches and return the call's status words as an error code
static std::atomic<int> g_conv_shift{1};    // fused 3x3 conv kernel with SAME padding: the shift form (0 = the per-output-position form)       // fused conv kernel: forced strip length (0 = heuristic)

int gpfq_set_option(const char *key, int value)
{
    if (!key) return fail(GPFQ_ERR_INVALID_ARG, "option key is NULL");
    if (!std::strcmp(key, "onchip_mode")) { g_onchip_mode = value ? 1 : 0; return GPFQ_OK; }
    if (!std::strcmp(key, "tile_steps")) {
        if (value < 0 || value > 64 || (value & (value - 1))) return fail(GPFQ_ERR_INVALID_ARG, "tile_steps must be 0 or a power of two <= 64");
        g_tile_steps = value; return GPFQ_OK;
    }
    if (!std::strcmp(key, "group_waves")) {
        if (value < 0 || value > 16) return fail(GPFQ_ERR_INVALID_ARG, "group_waves must be in [0, 16]");
        g_group_waves = value; return GPFQ_OK;
    }
    if (!std::strcmp(key, "variant")) { g_variant = value; return GPFQ_OK; }
    if (!std::strcmp(key, "pipe")) {
        if (value != -1 && value != 0 && value != 1 && value != 2)
            return fail(GPFQ_ERR_INVALID_ARG, "pipe must be -1, 0, 1 or 2");
        g_pipe = value; return GPFQ_OK;
    }
    if (!std::strcmp(key, "blk_four_groups")) { gpfq::blk_set_four_groups(value); return GPFQ_OK; }
    if (!std::strcmp(key, "blk_wide_groups")) { gpfq::blk_set_wide_groups(value); return GPFQ_OK; }
    if (!std::strcmp(key, "blk_pair_groups")) { gpfq::blk_set_pair_groups(value); return GPFQ_OK; }
    if (!std::strcmp(key, "blk_single_groups")) { gpfq::blk_set_single_groups(value); return GPFQ_OK; }
    if (!std::strcmp(key, "blk_quad_groups")) { gpfq::blk_set_quad_groups(value); return GPFQ_OK; }
    if (!std::strcmp(key, "blk_cluster_map")) { gpfq::blk_set_cluster_map(value); return GPFQ_OK; }
    if (!std::strcmp(key, "blk_cluster768")) { gpfq::blk_set_cluster768(value); return GPFQ_OK; }
    if (!std::strcmp(key, "blk_prep_run")) { gpfq::blk_set_prep_run(value); return GPFQ_OK; }
    if (!std::strcmp(key, "blk_prep_norms")) { gpfq::blk_set_prep_norms(value); return GPFQ_OK; }
    if (!std::strcmp(key, "blk_chip_ok")) { gpfq::blk_set_chip_ok(value); return GPFQ_OK; }
    if (!std::strcmp(key, "blk_cluster_fault")) { gpfq::blk_set_cluster_fault(value); return GPFQ_OK; }
    if (!std::strcmp(key, "blk_cluster_timeout_ms")) {
        if (value < 1 || value > 60000) return fail(GPFQ_ERR_INVALID_ARG, "blk_cluster_timeout_ms must be in [1, 60000]");
        gpfq::blk_set_cluster_timeout_ms(value); return GPFQ_OK;
    }
    if (!std::strcmp(key, "sync_errors")) { g_sync_errors = value ? 1 : 0; return GPFQ_OK; }
    if (!std::strcmp(key, "blk_cluster_nl")) { gpfq::blk_set_cluster_nl(value); return GPFQ_OK; }
    if (!std::strcmp(key, "blk_cluster")) {
        if (value < 0 || (value > 1 && value < 1024)) return fail(GPFQ_ERR_INVALID_ARG, "blk_cluster must be 0 (off), 1 (default: by row length and width) or a row length >= 1024");
        gpfq::blk_set_cluster(value); return GPFQ_OK;
    }
    if (!std::strcmp(key, "blk_quad_waves")) {
        if (value != 0 && value != 7 && value != 8) return fail(GPFQ_ERR_INVALID_ARG, "blk_quad_waves must be 0 (by shape), 7 or 8");
        gpfq::blk_set_quad_waves(value); return GPFQ_OK;
    }
    if (!std::strcmp(key, "blk_sweep_waves")) {
        if (value != 0 && value != 8 && value != 11) return fail(GPFQ_ERR_INVALID_ARG, "blk_sweep_waves must be 0 (by shape), 8 or 11");
        gpfq::blk_set_sweep_waves(value); return GPFQ_OK;
    }
    if (!std::strcmp(key, "waves_per_neuron")) {
        if (value < 0 || value > 16) return fail(GPFQ_ERR_INVALID_ARG, "waves_per_neuron must be in [0, 16]");
        g_wpn = value; return GPFQ_OK;
    }
    if (!std::strcmp(key, "gram_slack_log2")) { g_gram_slack_log2 = value; return GPFQ_OK; }
    if (!std::strcmp(key, "auto_gram")) { g_auto_gram = value ? 1 : 0; return GPFQ_OK; }
    if (!std::strcmp(key, "conv_fused")) { g_conv_fused = value ? 1 : 0; return GPFQ_OK; }
    if (!std::strcmp(key, "conv_nhwc")) { g_conv_nhwc = value ? 1 : 0; return GPFQ_OK; }
    if (!std::strcmp(key, "conv_planes_free")) { g_conv_planes_free = value ? 1 : 0; return GPFQ_OK; }
    if (!std::strcmp(key, "conv_s2")) { gpfq::conv_set_s2(value); return GPFQ_OK; }
    if (!std::strcmp(key, "conv_nhwc_slots")) { gpfq::image_set_nhwc_slots(value); return GPFQ_OK; }
    if (!std::strcmp(key, "conv_nhwc_halves")) { gpfq::image_set_nhwc_halves(value); return GPFQ_OK; }
    if (!std::strcmp(key, "conv_shift")) {
        if (value < 0 || value > 2) return fail(GPFQ_ERR_INVALID_ARG, "conv_shift must be 0, 1 or 2");
        g_conv_shift = value; return GPFQ_OK;
    }
    if (!std::strcmp(key, "conv_strip")) {
        if (value != 0 && value != 1 && value != 2 && value != 4)
            return fail(GPFQ_ERR_INVALID_ARG, "conv_strip must be 0, 1, 2 or 4");
        g_conv_strip = value; return GPFQ_OK;
    }
    if (!std::strcmp(key, "lanes_per_neuron")) {
        if (value != 0 && value != 1 && value != 16 && value != 32 && value != 64)
            return fail(GPFQ_ERR_INVALID_ARG, "lanes_per_neuron must be 0, 1, 16, 32 or 64");
        g_lpn = value; return GPFQ_OK;
    }
    return fail(GPFQ_ERR_INVALID_ARG, "unknown option '%s'", key);
}

int gpfq_call_status(const void *workspace, void *stream)
{
    if (!workspace) return fail(GPFQ_ERR_INVALID_ARG, "workspace is NULL");
    // (a pinned landing buffer per calling thread, allocated at the first use and kept: a 16-byte copy into pageable memory goes through
    //  the runtime's staging path and cost a 0.15 ms layer 0.05 ms)
    static thread_local int32_t *pinned = nullptr;
    int32_t stack_words[4] = {0, 0, 0, 0};
    if (!pinned && hipHostMalloc(reinterpret_cast<void **>(&pinned), 64, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); pinned = nullptr; }
    int32_t *w = pinned ? pinned : stack_words;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemcpyAsync(w, workspace, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return hip_fail(e, "gpfq_call_status");
    if (w[2] != 0)
        return fail(GPFQ_ERR_CLUSTER_TIMEOUT, "an exchange between the workgroups of the block kernel's cluster form timed out (a slice never arrived): "
                                              "the outputs of this call are invalid; rerun it with gpfq_set_option(\"blk_cluster\", 0)");
    if (w[3] != 0)
        return fail(GPFQ_ERR_ALPHABET, "the device-resident alphabet is not a strictly ascending arithmetic progression (radius zero, infinite or NaN): "
                                       "nothing was computed; rerun the layer through gpfq_quantize_neurons with a host alphabet");
    return GPFQ_OK;
}

int gpfq_quantize_neurons(const float *X, const float *Xq, int64_t ld, const float *nrm32,
                          const float *Wt, int64_t ldw,
                          const double *alphabet, int M, int zero_idx,
                          int64_t N, int64_t m, int64_t C,
                          void *qidx_v, float *Qt, double *resid, double *u_out,
                          void *workspace, size_t workspace_bytes, int path, void *stream)
{
    if (N < 0 || m < 0 || C < 0)
        return fail(GPFQ_ERR_INVALID_ARG, "negative size N=%lld m=%lld C=%lld", (long long)N, (long long)m, (long long)C);
    HostAlphabet H;
    int rc = make_alphabet(alphabet, M, zero_idx, &H);
    if (rc != GPFQ_OK) return rc;
    const gpfq::AlphabetArg &A = H.A;
    int8_t *qidx = static_cast<int8_t *>(qidx_v);          // int16 elements when H.is_big
    if (C == 0) return GPFQ_OK;
    if (N > 0 && (!Wt || !nrm32)) return fail(GPFQ_ERR_INVALID_ARG, "Wt/nrm32 is NULL");
    if (N > 0 && m > 0 && (!X || !Xq)) return fail(GPFQ_ERR_INVALID_ARG, "X/Xq is NULL");
    if (ld < m) return fail(GPFQ_ERR_INVALID_ARG, "row pitch ld=%lld < m=%lld", (long long)ld, (long long)m);
    if (ldw < N) return fail(GPFQ_ERR_INVALID_ARG, "weight pitch ldw=%lld < N=%lld", (long long)ldw, (long long)N);
    if (path != GPFQ_PATH_AUTO && path != GPFQ_PATH_ONCHIP && path != GPFQ_PATH_STREAM)
        return fail(GPFQ_ERR_INVALID_ARG, "unknown path %d", path);
    const int p = resolve_path(m, path);
    hipStream_t s = static_cast<hipStream_t>(stream);

    // (alphabets beyond 64 members have no wavefront-per-neuron chain for walks beyond 64 steps: those stay on the element-wise paths)
    if (path == GPFQ_PATH_AUTO && g_auto_gram && N > 0 && m > 0 && auto_wants_gram(N, m, C, u_out != nullptr) && !(H.is_big && N > 64) && workspace &&
        (uintptr_t)workspace % 16 == 0 && workspace_bytes >= auto_gram_workspace_bytes(N, m, C)) {
        // Long rows, short walks: Gram records once per layer, the recurrence on scalars with every decision certified,
        // uncertifiable chains repaired on the device; whatever is still flagged afterwards (practically never) is rerun
        // here through the streaming kernel, which costs this call ONE stream synchronisation (the flags cross to the host).
        char *ws = static_cast<char *>(workspace);
        void *gram_ws = ws;
        int32_t *unc = reinterpret_cast<int32_t *>(ws + al256(gpfq::gram_workspace_bytes(N, m, C)));
        char *stream_ws = reinterpret_cast<char *>(unc) + al256((size_t)C * sizeof(int32_t));
        gpfq::GramArgs g;
        g.X = X; g.Xq = Xq; g.ld = ld; g.nrm32 = nrm32; g.Wt = Wt; g.ldw = ldw; g.A = A; g.big = H.big();
        g.N = N; g.m = m; g.C = C; g.qidx = qidx; g.Qt = Qt; g.resid = resid; g.uncertified = unc;
        g.workspace = gram_ws;
        g.slack = std::ldexp(1.0, g_gram_slack_log2);
        g.variant = g_variant;
        gpfq::note_dense_kernel("gpfq_gram_* (Gram records + certified scalar recurrence), reruns through gpfq_stream_*");
        hipError_t e = gpfq::launch_gram(g, s);
        if (e != hipSuccess) return hip_fail(e, "gpfq_quantize_neurons(auto: gram)");
        std::vector<int32_t> flags((size_t)C);
        e = hipMemcpyAsync(flags.data(), unc, (size_t)C * sizeof(int32_t), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return hip_fail(e, "gpfq_quantize_neurons(auto: flags)");
        for (int64_t j = 0; j < C; ++j) {
            if (!flags[(size_t)j]) continue;
            gpfq::StreamArgs a;
            a.X = X; a.Xq = Xq; a.ld = ld; a.nrm32 = nrm32; a.Wt = Wt + j * ldw; a.ldw = ldw; a.A = A; a.big = H.big();
            a.N = N; a.m = m; a.C = 1; a.qidx = H.at(qidx, j * N); a.Qt = Qt ? Qt + j * N : nullptr;
            a.resid = resid ? resid + j : nullptr; a.u_out = nullptr;
            a.workspace = stream_ws; a.workspace_bytes = gpfq::stream_workspace_bytes(N, m, 1, true);
            e = gpfq::launch_stream(a, s);
            if (e != hipSuccess) return hip_fail(e, "gpfq_quantize_neurons(auto: rerun)");
        }
        return GPFQ_OK;
    }

    if (p == GPFQ_PATH_ONCHIP) {
        if (m > GPFQ_ONCHIP_MAX_M)
            return fail(GPFQ_ERR_UNSUPPORTED, "on-chip path needs m <= %d (got %lld)", GPFQ_ONCHIP_MAX_M, (long long)m);
        gpfq::OnchipArgs a;
        a.X = X; a.Xq = Xq; a.ld = ld; a.nrm32 = nrm32; a.Wt = Wt; a.ldw = ldw; a.A = A; a.big = H.big();
        a.N = N; a.m = m; a.C = C; a.qidx = qidx; a.Qt = Qt; a.resid = resid; a.u_out = u_out;
        a.ts_override = g_tile_steps; a.nw_override = g_group_waves;
        a.mode = g_onchip_mode;
        a.lpn = g_lpn;
        a.wpn = g_wpn;
        a.variant = g_variant;
        // certified mode needs the per-row statistics in the workspace; without one, run the exact flow
        const bool have_ws = workspace && workspace_bytes >= onchip_stats_bytes(N) && (uintptr_t)workspace % 16 == 0;
        bool counters_zeroed = false;
        auto zero_counters = [&]() -> hipError_t {             // the call's counter block: exact fallbacks, cluster timeout, alphabet word
            if (!have_ws || counters_zeroed) return hipSuccess;
            counters_zeroed = true;
            return hipMemsetAsync(workspace, 0, 64, s);
        };
        // the pipelined kernel (gpfq_pipe.hip): rows of up to 2048 samples in layers wide enough to fill the chip
        // with 16 neurons per workgroup; narrow layers keep the latency-oriented kernels below
        if (!H.is_big) {
            gpfq::PipeArgs pa;
            pa.X = X; pa.Xq = Xq; pa.ld = ld; pa.nrm32 = nrm32; pa.Wt = Wt; pa.ldw = ldw; pa.A = A;
            pa.N = N; pa.m = m; pa.C = C; pa.qidx = qidx; pa.Qt = Qt; pa.resid = resid; pa.u_out = u_out;
            pa.ts_override = g_tile_steps; pa.variant = g_variant >> 4;
            const bool forced_old = g_lpn != 0 || g_wpn != 0 || g_onchip_mode != 1 || g_pipe == 0;
            // measured (tools/blk_ab.sh shapes): the block-pipelined kernel is ahead of the row-group and wavefront-per-neuron kernels
            // for rows of 257..4096 samples (2049+: one step per slot; 4096 x 4096 x 4096: 16.6 vs 38 ms) whenever the layer has 512 neurons or more (4096 x 4096, m = 1024: 4.1 vs 5.4 ms;
            // m = 2048, 16 levels: 8.7 vs 10.0; m = 512: 3.2 vs 3.9; 4096 x 1024, m = 1536: 3.7 vs 6.8; 784 x 4096, m = 512: 0.68
            // vs 0.85).  Its time per step does not depend on the number of neurons up to one workgroup per CU; narrower layers
            // stay with the kernels that split a neuron over several wavefronts (4096 x 128, m = 512: 3.05 vs 3.16 ms; m = 2048:
            // 2.06 vs 2.20) -- except for rows of 769+ samples, where workgroups of 8 and of 4 neurons make it the fastest at any
            // width (4096 x 2048, m = 1024: 2.5 vs 5.4 ms; 4096 x 128, m = 2048: 2.9 vs 4.5; 2048 x 128, m = 4096: 2.3 vs 2.5).
            // Round 3: two-neuron workgroups (layers of at most 512 neurons, rows of up to 5120 samples) make it the fastest for narrow
            // layers as well (784 x 128, m = 512, 16 levels; 2048 x 128, m = 5008: tools/blk_ab.sh latency, profiles/r03/).
            // Round 5: rows of 5121..28672 samples too -- the cluster form (gpfq_blk.hip: slices of 1024 samples over several workgroups;
            // blk_supported() says no when the option blk_cluster switches it off)
            const bool fits = m > 256 && M <= 64 && m <= GPFQ_ONCHIP_MAX_M;
            const bool want = g_pipe == 1;
            if ((g_pipe == 2 || (g_pipe < 0 && !forced_old && fits)) && N > 0 && m > 0 && gpfq::blk_supported(pa) && workspace &&
                (uintptr_t)workspace % 16 == 0 && workspace_bytes >= onchip_workspace_bytes(N, m, C)) {
                pa.workspace = static_cast<char *>(workspace) + onchip_stats_bytes(N);
                pa.fallback_count = static_cast<unsigned long long *>(workspace);
                pa.zero_counters = 1;                              // (the kernel that stores the alphabet zeroes the counter block as well: one launch, no memset)
                gpfq::note_dense_kernel("gpfq_blk_kernel (4 to 11 sweep wavefronts + 1 decision wavefront per workgroup, blocks of steps per slot)");
                hipError_t e = gpfq::launch_blk(pa, s);
                if (e != hipSuccess) return hip_fail(e, "gpfq_quantize_neurons(block-pipelined)");
                return g_sync_errors ? gpfq_call_status(workspace, stream) : GPFQ_OK;
            }
            if (want && N > 0 && m > 0 && gpfq::pipe_supported(pa) && workspace && (uintptr_t)workspace % 16 == 0 &&
                workspace_bytes >= onchip_workspace_bytes(N, m, C)) {
                hipError_t ez = zero_counters();
                if (ez != hipSuccess) return hip_fail(ez, "gpfq_quantize_neurons(workspace)");
                pa.workspace = static_cast<char *>(workspace) + onchip_stats_bytes(N);
                pa.fallback_count = static_cast<unsigned long long *>(workspace);
                gpfq::note_dense_kernel("gpfq_pipe_kernel (8 sweep wavefronts over the sample axis + 1 decision wavefront per workgroup)");
                hipError_t e = gpfq::launch_pipe(pa, s);
                return e == hipSuccess ? GPFQ_OK : hip_fail(e, "gpfq_quantize_neurons(pipelined)");
            }
        }
        {
            hipError_t ez = zero_counters();
            if (ez != hipSuccess) return hip_fail(ez, "gpfq_quantize_neurons(workspace)");
        }
        if (a.mode == 1 && N > 0 && have_ws) {
            auto *stats = reinterpret_cast<gpfq::RowStats *>(static_cast<char *>(workspace) + 64);
            a.fallback_count = static_cast<unsigned long long *>(workspace);
            hipError_t e0 = gpfq::launch_row_stats(X, Xq, N, m, ld, nrm32, stats, s);
            if (e0 != hipSuccess) return hip_fail(e0, "gpfq_quantize_neurons(row stats)");
            a.stats = stats;
        } else {
            a.mode = 0;
        }
        hipError_t e = gpfq::launch_onchip(a, s);
        return e == hipSuccess ? GPFQ_OK : hip_fail(e, "gpfq_quantize_neurons(on-chip)");
    }

    const size_t need = gpfq::stream_workspace_bytes(N, m, C, u_out == nullptr);
    if (!workspace || workspace_bytes < need)
        return fail(GPFQ_ERR_WORKSPACE, "streaming path needs %zu workspace bytes, got %zu", need, workspace ? workspace_bytes : (size_t)0);
    if ((uintptr_t)workspace % 16 != 0) return fail(GPFQ_ERR_INVALID_ARG, "workspace must be 16-byte aligned");
    gpfq::StreamArgs a;
    a.X = X; a.Xq = Xq; a.ld = ld; a.nrm32 = nrm32; a.Wt = Wt; a.ldw = ldw; a.A = A; a.big = H.big();
    a.N = N; a.m = m; a.C = C; a.qidx = qidx; a.Qt = Qt; a.resid = resid; a.u_out = u_out;
    a.workspace = workspace; a.workspace_bytes = workspace_bytes;
    gpfq::note_dense_kernel("gpfq_stream_step_kernel + gpfq_stream_decide_kernel (residual in HBM)");
    hipError_t e = gpfq::launch_stream(a, s);
    return e == hipSuccess ? GPFQ_OK : hip_fail(e, "gpfq_quantize_neurons(stream)");
}


// ---- the Dense layer driver as one call, alphabet in device memory (round 6) ----------------------------------------------------------
static int unit_alphabet_arg(const double *unit_alphabet, int M, HostAlphabet *H)
{
    if (!unit_alphabet) return fail(GPFQ_ERR_INVALID_ARG, "unit_alphabet is NULL");
    int zero_idx = -1;
    for (int k = 0; k < M && k < GPFQ_MAX_ALPHABET; ++k)
        if (unit_alphabet[k] == 0.0) zero_idx = k;
    return make_alphabet(unit_alphabet, M, zero_idx, H);
}

int gpfq_layer_alphabet_device(const float *median32, double alphabet_scalar, const double *unit_alphabet, int M,
                               void *dev_alphabet, void *stream)
{
    HostAlphabet H;
    int rc = unit_alphabet_arg(unit_alphabet, M, &H);
    if (rc != GPFQ_OK) return rc;
    if (H.is_big) return fail(GPFQ_ERR_UNSUPPORTED, "device-resident alphabets hold up to 64 members (got %d)", M);
    if (!median32 || !dev_alphabet) return fail(GPFQ_ERR_INVALID_ARG, "NULL pointer");
    if ((uintptr_t)dev_alphabet % 16 != 0) return fail(GPFQ_ERR_INVALID_ARG, "dev_alphabet must be 16-byte aligned");
    hipError_t e = gpfq::launch_alphabet_device(median32, alphabet_scalar, H.A, dev_alphabet, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? GPFQ_OK : hip_fail(e, "gpfq_layer_alphabet_device");
}

// median(|W|) and the layer alphabet in one go: the last workgroup of the median's second pass forms the alphabet (no launch of its
// own between the two).  Needs the two-pass form: a 16-byte aligned kernel and gpfq_median_abs_workspace_bytes_for(n) of workspace.
int gpfq_layer_alphabet_from_kernel(const float *W, int64_t n, double alphabet_scalar, const double *unit_alphabet, int M,
                                    void *dev_alphabet, float *median_out, void *workspace, size_t workspace_bytes, void *stream)
{
    if (n <= 0) return fail(GPFQ_ERR_INVALID_ARG, "median of %lld elements", (long long)n);
    HostAlphabet H;
    int rc = unit_alphabet_arg(unit_alphabet, M, &H);
    if (rc != GPFQ_OK) return rc;
    if (H.is_big) return fail(GPFQ_ERR_UNSUPPORTED, "device-resident alphabets hold up to 64 members (got %d)", M);
    if (!W || !dev_alphabet) return fail(GPFQ_ERR_INVALID_ARG, "NULL pointer");
    if ((uintptr_t)dev_alphabet % 16 != 0) return fail(GPFQ_ERR_INVALID_ARG, "dev_alphabet must be 16-byte aligned");
    if ((uintptr_t)W % 16 != 0) return fail(GPFQ_ERR_UNSUPPORTED, "a kernel that is not 16-byte aligned: gpfq_median_abs + gpfq_layer_alphabet_device");
    if (!workspace || (uintptr_t)workspace % 16 != 0 || workspace_bytes < gpfq_median_abs_workspace_bytes_for(n))
        return fail(GPFQ_ERR_WORKSPACE, "gpfq_layer_alphabet_from_kernel needs %zu aligned workspace bytes", gpfq_median_abs_workspace_bytes_for(n));
    hipError_t e = gpfq::launch_median_abs(W, n, median_out, workspace, static_cast<hipStream_t>(stream), workspace_bytes,
                                           dev_alphabet, alphabet_scalar, &H.A, gpfq::blk_unit_wants_sym(H.A) ? 1 : 0);
    return e == hipSuccess ? GPFQ_OK : hip_fail(e, "gpfq_layer_alphabet_from_kernel");
}

static gpfq::PipeArgs dense_layer_probe(int64_t N, int64_t m, int64_t C, const HostAlphabet &H)
{
    gpfq::PipeArgs pa{};
    pa.A = H.A; pa.N = N; pa.m = m; pa.C = C;
    return pa;
}

int gpfq_dense_layer_supported(int64_t N, int64_t m, int64_t C, const double *unit_alphabet, int M)
{
    if (N < 1 || m < 1 || C < 1 || !unit_alphabet || M < 1 || M > 64) return 0;
    HostAlphabet H;
    if (unit_alphabet_arg(unit_alphabet, M, &H) != GPFQ_OK) return 0;
    // exactly the rows and alphabets gpfq_quantize_neurons gives the block-pipelined kernel by default
    const bool forced_old = g_lpn != 0 || g_wpn != 0 || g_onchip_mode != 1 || g_pipe == 0 || g_pipe == 1;
    if (forced_old || !(m > 256 && m <= GPFQ_ONCHIP_MAX_M)) return 0;
    return gpfq::blk_supported(dense_layer_probe(N, m, C, H)) ? 1 : 0;
}

int gpfq_dense_layer_keras_out_supported(int64_t N, int64_t m, int64_t C, const double *unit_alphabet, int M)
{
    return gpfq_dense_layer_supported(N, m, C, unit_alphabet, M) && gpfq::blk_keras_out_supported(m, C) ? 1 : 0;
}

size_t gpfq_dense_layer_workspace_bytes(int64_t N, int64_t m, int64_t C)
{
    if (N < 0 || m < 0 || C < 0) return 0;
    return onchip_workspace_bytes(N, m, C) + al256((size_t)N * sizeof(float));      // (+ the row norms when the caller passes none)
}

// phase 0: the whole layer call; 1: the alphabet-independent half (status block, row norms, record pre-pass); 2: the alphabet-dependent half
static int dense_layer_impl(int phase, const float *X, const float *Xq, int64_t ld, const float *nrm32,
                            const float *W, int64_t ldc, int64_t c_lo, int64_t C,
                            const void *dev_alphabet, const double *unit_alphabet, int M,
                            int64_t N, int64_t m,
                            int8_t *qidx, float *Q, int out_layout, int64_t ldo, double *resid,
                            void *workspace, size_t workspace_bytes, void *stream, const char *what)
{
    if (N < 1 || m < 1 || C < 0 || c_lo < 0)
        return fail(GPFQ_ERR_INVALID_ARG, "bad size N=%lld m=%lld C=%lld c_lo=%lld", (long long)N, (long long)m, (long long)C, (long long)c_lo);
    HostAlphabet H;
    int rc = unit_alphabet_arg(unit_alphabet, M, &H);
    if (rc != GPFQ_OK) return rc;
    if (C == 0) return GPFQ_OK;
    if (!X || !Xq) return fail(GPFQ_ERR_INVALID_ARG, "NULL pointer");
    if (phase != 1 && (!W || !dev_alphabet)) return fail(GPFQ_ERR_INVALID_ARG, "NULL pointer");
    if (ld < m) return fail(GPFQ_ERR_INVALID_ARG, "row pitch ld=%lld < m=%lld", (long long)ld, (long long)m);
    if (phase != 1) {
        if (ldc < c_lo + C) return fail(GPFQ_ERR_INVALID_ARG, "kernel pitch ldc=%lld < c_lo + C=%lld", (long long)ldc, (long long)(c_lo + C));
        if (out_layout != GPFQ_LAYOUT_NEURON_MAJOR && out_layout != GPFQ_LAYOUT_KERAS) return fail(GPFQ_ERR_INVALID_ARG, "unknown output layout %d", out_layout);
        if (out_layout == GPFQ_LAYOUT_KERAS && ldo < c_lo + C) return fail(GPFQ_ERR_INVALID_ARG, "output pitch ldo=%lld < c_lo + C=%lld", (long long)ldo, (long long)(c_lo + C));
        if (out_layout == GPFQ_LAYOUT_KERAS && ldo != 1 && !gpfq::blk_keras_out_supported(m, C))
            return fail(GPFQ_ERR_UNSUPPORTED, "the kernel of this shape (m=%lld, C=%lld) writes neuron-major outputs only (gpfq_dense_layer_keras_out_supported): "
                                              "ask for GPFQ_LAYOUT_NEURON_MAJOR and lay them out with gpfq_assemble_kernel_device", (long long)m, (long long)C);
    }
    if (!gpfq_dense_layer_supported(N, m, C, unit_alphabet, M))
        return fail(GPFQ_ERR_UNSUPPORTED, "no block-pipelined kernel for N=%lld m=%lld C=%lld M=%d (gpfq_dense_layer_supported): use gpfq_quantize_neurons with a host alphabet",
                    (long long)N, (long long)m, (long long)C, M);
    if (!workspace || (uintptr_t)workspace % 16 != 0 || workspace_bytes < gpfq_dense_layer_workspace_bytes(N, m, C))
        return fail(GPFQ_ERR_WORKSPACE, "%s needs %zu aligned workspace bytes", what, gpfq_dense_layer_workspace_bytes(N, m, C));
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipSuccess;
    float *n32_out = nullptr;
    if (phase != 2) {
        // the call's counter block (exact fallbacks, cluster timeout, alphabet word) is zeroed with the row norms when they are formed here
        // (launch_blk: inside the record pre-pass, or by the row-norm kernel in front of it)
        if (!nrm32) {
            n32_out = reinterpret_cast<float *>(static_cast<char *>(workspace) + onchip_workspace_bytes(N, m, C));
        } else {
            e = hipMemsetAsync(workspace, 0, 64, s);
            if (e != hipSuccess) return hip_fail(e, what);
        }
    }
    gpfq::PipeArgs pa{};
    pa.X = X; pa.Xq = Xq; pa.ld = ld; pa.nrm32 = nrm32; pa.nrm32_out = n32_out;
    pa.Wt = W ? W + c_lo : nullptr; pa.ldw = 1; pa.ldt = ldc;     // the Keras kernel itself: neuron j's weight of step t is W[t][c_lo + j]
    pa.A = H.A; pa.N = N; pa.m = m; pa.C = C;
    pa.resid = resid; pa.u_out = nullptr;
    pa.ts_override = g_tile_steps; pa.variant = g_variant >> 4;
    if (out_layout == GPFQ_LAYOUT_KERAS) {
        pa.qidx = qidx ? qidx + c_lo : nullptr; pa.Qt = Q ? Q + c_lo : nullptr;
        pa.o_sj = 1; pa.o_st = ldo;
        if (ldo == 1) { pa.o_sj = N; pa.o_st = 1; }               // (a one-neuron layer: the two layouts coincide)
    } else {
        pa.qidx = qidx; pa.Qt = Q;
    }
    pa.dev_alpha = static_cast<const gpfq::DevAlphabet *>(dev_alphabet);
    pa.phase = phase;
    pa.workspace = static_cast<char *>(workspace) + onchip_stats_bytes(N);
    pa.fallback_count = static_cast<unsigned long long *>(workspace);
    if (phase != 1) gpfq::note_dense_kernel("gpfq_blk_kernel (4 to 11 sweep wavefronts + 1 decision wavefront per workgroup, blocks of steps per slot)");
    e = gpfq::launch_blk(pa, s);
    if (e != hipSuccess) return hip_fail(e, what);
    return (g_sync_errors && phase != 1) ? gpfq_call_status(workspace, stream) : GPFQ_OK;
}

int gpfq_quantize_dense_layer(const float *X, const float *Xq, int64_t ld, const float *nrm32,
                              const float *W, int64_t ldc, int64_t c_lo, int64_t C,
                              const void *dev_alphabet, const double *unit_alphabet, int M,
                              int64_t N, int64_t m,
                              int8_t *qidx, float *Q, int out_layout, int64_t ldo, double *resid,
                              void *workspace, size_t workspace_bytes, void *stream)
{
    return dense_layer_impl(0, X, Xq, ld, nrm32, W, ldc, c_lo, C, dev_alphabet, unit_alphabet, M, N, m, qidx, Q, out_layout, ldo, resid,
                            workspace, workspace_bytes, stream, "gpfq_quantize_dense_layer");
}

int gpfq_dense_layer_prepare(const float *X, const float *Xq, int64_t ld, const float *nrm32, const double *unit_alphabet, int M,
                             int64_t N, int64_t m, int64_t C, void *workspace, size_t workspace_bytes, void *stream)
{
    return dense_layer_impl(1, X, Xq, ld, nrm32, nullptr, 0, 0, C, nullptr, unit_alphabet, M, N, m, nullptr, nullptr, GPFQ_LAYOUT_KERAS, 0, nullptr,
                            workspace, workspace_bytes, stream, "gpfq_dense_layer_prepare");
}

int gpfq_dense_layer_run(const float *X, const float *Xq, int64_t ld,
                         const float *W, int64_t ldc, int64_t c_lo, int64_t C,
                         const void *dev_alphabet, const double *unit_alphabet, int M,
                         int64_t N, int64_t m,
                         int8_t *qidx, float *Q, int out_layout, int64_t ldo, double *resid,
                         void *workspace, size_t workspace_bytes, void *stream)
{
    return dense_layer_impl(2, X, Xq, ld, nullptr, W, ldc, c_lo, C, dev_alphabet, unit_alphabet, M, N, m, qidx, Q, out_layout, ldo, resid,
                            workspace, workspace_bytes, stream, "gpfq_dense_layer_run");
}

int gpfq_assemble_kernel_device(const void *qidx, int bits, const void *dev_alphabet, int M, int64_t N, int64_t C,
                                float *Q, void *qidx_t, void *stream)
{
    if (N < 0 || C < 0) return fail(GPFQ_ERR_INVALID_ARG, "negative size");
    if (bits != 2 && bits != 4 && bits != 8) return fail(GPFQ_ERR_INVALID_ARG, "bits must be 2, 4 or 8 (device-resident alphabets hold up to 64 members)");
    if (M < 1 || M > 64) return fail(GPFQ_ERR_UNSUPPORTED, "device-resident alphabets hold 1..64 members (got %d)", M);
    if (bits < 8 && M + 1 > (1 << bits)) return fail(GPFQ_ERR_INVALID_ARG, "%d-bit codes cannot hold an alphabet of %d", bits, M);
    if (N == 0 || C == 0) return GPFQ_OK;
    if (!qidx || !dev_alphabet) return fail(GPFQ_ERR_INVALID_ARG, "NULL pointer");
    if (N > 2147483647LL * 32) return fail(GPFQ_ERR_UNSUPPORTED, "kernel too large to assemble in one call");
    gpfq::AlphabetArg A{};
    A.M = M;
    hipError_t e = gpfq::launch_assemble(static_cast<const int8_t *>(qidx), A, N, C, bits, Q, static_cast<int8_t *>(qidx_t),
                                         static_cast<hipStream_t>(stream), nullptr, static_cast<const gpfq::DevAlphabet *>(dev_alphabet));
    return e == hipSuccess ? GPFQ_OK : hip_fail(e, "gpfq_assemble_kernel_device");
}

size_t gpfq_gram_workspace_bytes(int64_t N, int64_t m, int64_t C)
{
    if (N < 0 || m < 0 || C < 0) return 0;
    return gpfq::gram_workspace_bytes(N, m, C);
}

int gpfq_quantize_neurons_gram(const float *X, const float *Xq, int64_t ld, float *nrm32, int compute_norms,
                               const float *Wt, int64_t ldw,
                               const double *alphabet, int M, int zero_idx,
                               int64_t N, int64_t m, int64_t C,
                               void *qidx, float *Qt, double *resid, int32_t *uncertified,
                               void *workspace, size_t workspace_bytes, void *stream)
{
    if (N < 0 || m < 0 || C < 0)
        return fail(GPFQ_ERR_INVALID_ARG, "negative size N=%lld m=%lld C=%lld", (long long)N, (long long)m, (long long)C);
    HostAlphabet H;
    int rc = make_alphabet(alphabet, M, zero_idx, &H);
    if (rc != GPFQ_OK) return rc;
    if (C == 0) return GPFQ_OK;
    if (N > GPFQ_GRAM_MAX_N) return fail(GPFQ_ERR_UNSUPPORTED, "Gram path needs N <= %d (got %lld)", GPFQ_GRAM_MAX_N, (long long)N);
    if (m >= (1LL << 30)) return fail(GPFQ_ERR_UNSUPPORTED, "Gram path needs m < 2^30 (error-bound derivation)");
    if (!uncertified) return fail(GPFQ_ERR_INVALID_ARG, "uncertified is NULL");
    if (N > 0 && (!Wt || !nrm32)) return fail(GPFQ_ERR_INVALID_ARG, "Wt/nrm32 is NULL");
    if (N > 0 && m > 0 && (!X || !Xq)) return fail(GPFQ_ERR_INVALID_ARG, "X/Xq is NULL");
    if (ld < m) return fail(GPFQ_ERR_INVALID_ARG, "row pitch ld=%lld < m=%lld", (long long)ld, (long long)m);
    if (ldw < N) return fail(GPFQ_ERR_INVALID_ARG, "weight pitch ldw=%lld < N=%lld", (long long)ldw, (long long)N);
    const size_t need = gpfq::gram_workspace_bytes(N, m, C);
    if (!workspace || workspace_bytes < need || (uintptr_t)workspace % 16 != 0)
        return fail(GPFQ_ERR_WORKSPACE, "Gram path needs %zu aligned workspace bytes", need);
    gpfq::GramArgs a;
    a.X = X; a.Xq = Xq; a.ld = ld; a.nrm32 = nrm32; a.Wt = Wt; a.ldw = ldw; a.A = H.A; a.big = H.big();
    a.N = N; a.m = m; a.C = C; a.qidx = static_cast<int8_t *>(qidx); a.Qt = Qt; a.resid = resid; a.uncertified = uncertified;
    a.workspace = workspace;
    a.nrm32_out = compute_norms ? nrm32 : nullptr;
    a.slack = std::ldexp(1.0, g_gram_slack_log2);
    a.variant = g_variant;
    hipError_t e = gpfq::launch_gram(a, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? GPFQ_OK : hip_fail(e, "gpfq_quantize_neurons_gram");
}

int gpfq_msq_round(const float *W, int64_t n, const double *alphabet, int M, float *Q, void *qidx, void *stream)
{
    if (n < 0) return fail(GPFQ_ERR_INVALID_ARG, "negative n");
    HostAlphabet H;
    int rc = make_alphabet(alphabet, M, -1, &H);
    if (rc != GPFQ_OK) return rc;
    if (n == 0) return GPFQ_OK;
    if (!W) return fail(GPFQ_ERR_INVALID_ARG, "W is NULL");
    hipError_t e = gpfq::launch_msq(W, n, H.A, Q, static_cast<int8_t *>(qidx), static_cast<hipStream_t>(stream), H.big());
    return e == hipSuccess ? GPFQ_OK : hip_fail(e, "gpfq_msq_round");
}

int gpfq_index_bits(int M)
{
    if (M < 1 || M > GPFQ_MAX_ALPHABET) return 0;
    return M <= 3 ? 2 : (M <= 15 ? 4 : (M <= 64 ? 8 : 16));   // packed codes 0..M (0 = literal zero); plain int8 / int16 indices
}

int gpfq_pack_indices(const int8_t *qidx, int64_t N, int64_t C, int bits, uint8_t *packed, void *stream)
{
    if (N < 0 || C < 0) return fail(GPFQ_ERR_INVALID_ARG, "negative size");
    if (bits != 2 && bits != 4) return fail(GPFQ_ERR_INVALID_ARG, "bits must be 2 or 4 (8-bit indices need no packing)");
    if (N == 0 || C == 0) return GPFQ_OK;
    if (!qidx || !packed) return fail(GPFQ_ERR_INVALID_ARG, "NULL pointer");
    hipError_t e = gpfq::launch_pack(qidx, N, C, bits, packed, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? GPFQ_OK : hip_fail(e, "gpfq_pack_indices");
}

int gpfq_assemble_kernel(const void *qidx, int bits, const double *alphabet, int M, int64_t N, int64_t C,
                         float *Q, void *qidx_t, void *stream)
{
    if (N < 0 || C < 0) return fail(GPFQ_ERR_INVALID_ARG, "negative size");
    if (bits != 2 && bits != 4 && bits != 8 && bits != 16) return fail(GPFQ_ERR_INVALID_ARG, "bits must be 2, 4, 8 or 16");
    HostAlphabet H;
    int rc = make_alphabet(alphabet, M, -1, &H);
    if (rc != GPFQ_OK) return rc;
    if (bits < 8 && M + 1 > (1 << bits)) return fail(GPFQ_ERR_INVALID_ARG, "%d-bit codes cannot hold an alphabet of %d", bits, M);
    if ((bits == 16) != H.is_big)
        return fail(GPFQ_ERR_INVALID_ARG, "alphabets of %d members have %s indices (gpfq_index_bits)", M, H.is_big ? "int16" : "int8 or packed");
    if (N == 0 || C == 0) return GPFQ_OK;
    if (!qidx) return fail(GPFQ_ERR_INVALID_ARG, "qidx is NULL");
    if (N > 2147483647LL * 32) return fail(GPFQ_ERR_UNSUPPORTED, "kernel too large to assemble in one call");
    hipError_t e = gpfq::launch_assemble(static_cast<const int8_t *>(qidx), H.A, N, C, bits, Q, static_cast<int8_t *>(qidx_t),
                                         static_cast<hipStream_t>(stream), H.big());
    return e == hipSuccess ? GPFQ_OK : hip_fail(e, "gpfq_assemble_kernel");
}

size_t gpfq_median_abs_workspace_bytes(void) { return gpfq::median_workspace_bytes() + 64; }
size_t gpfq_median_abs_workspace_bytes_for(int64_t n) { return n > 0 ? gpfq::median_workspace_bytes_fast(n) : gpfq::median_workspace_bytes() + 64; }

int gpfq_median_abs(const float *W, int64_t n, float *median_out, void *workspace, size_t workspace_bytes, void *stream)
{
    if (n <= 0) return fail(GPFQ_ERR_INVALID_ARG, "median of %lld elements", (long long)n);
    if (!W || !median_out) return fail(GPFQ_ERR_INVALID_ARG, "NULL pointer");
    if (!workspace || workspace_bytes < gpfq_median_abs_workspace_bytes() || (uintptr_t)workspace % 16 != 0)
        return fail(GPFQ_ERR_WORKSPACE, "median needs %zu aligned workspace bytes", gpfq_median_abs_workspace_bytes());
    hipError_t e = gpfq::launch_median_abs(W, n, median_out, workspace, static_cast<hipStream_t>(stream), workspace_bytes);
    return e == hipSuccess ? GPFQ_OK : hip_fail(e, "gpfq_median_abs");
}

static int median_ws_ok(void *workspace, size_t workspace_bytes)
{
    if (!workspace || workspace_bytes < gpfq::median_workspace_bytes() || (uintptr_t)workspace % 16 != 0)
        return fail(GPFQ_ERR_WORKSPACE, "median needs %zu aligned workspace bytes", gpfq::median_workspace_bytes());
    return GPFQ_OK;
}

int gpfq_median_abs_begin(int64_t n_total, void *workspace, size_t workspace_bytes, void *stream)
{
    if (n_total <= 0) return fail(GPFQ_ERR_INVALID_ARG, "median of %lld elements", (long long)n_total);
    int rc = median_ws_ok(workspace, workspace_bytes);
    if (rc != GPFQ_OK) return rc;
    hipError_t e = gpfq::launch_median_begin(n_total, workspace, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? GPFQ_OK : hip_fail(e, "gpfq_median_abs_begin");
}

int gpfq_median_abs_count(const float *W_local, int64_t n_local, int64_t n_total, int pass, void *workspace, void *stream)
{
    if (n_local < 0 || n_total <= 0 || n_local > n_total || pass < 0 || pass > 2) return fail(GPFQ_ERR_INVALID_ARG, "bad slice/pass");
    if ((n_local > 0 && !W_local) || !workspace) return fail(GPFQ_ERR_INVALID_ARG, "NULL pointer");
    hipError_t e = gpfq::launch_median_count(W_local, n_local, n_total, pass, workspace, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? GPFQ_OK : hip_fail(e, "gpfq_median_abs_count");
}

int gpfq_median_abs_pick(int64_t n_total, int pass, void *workspace, void *stream)
{
    if (n_total <= 0 || pass < 0 || pass > 2 || !workspace) return fail(GPFQ_ERR_INVALID_ARG, "bad pass / NULL workspace");
    hipError_t e = gpfq::launch_median_pick(n_total, pass, workspace, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? GPFQ_OK : hip_fail(e, "gpfq_median_abs_pick");
}

int gpfq_median_abs_end(int64_t n_total, void *workspace, float *median_out, void *stream)
{
    if (n_total <= 0 || !workspace || !median_out) return fail(GPFQ_ERR_INVALID_ARG, "NULL pointer");
    hipError_t e = gpfq::launch_median_end(n_total, workspace, median_out, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? GPFQ_OK : hip_fail(e, "gpfq_median_abs_end");
}

int64_t gpfq_patch_out_dim(int64_t in, int64_t k, int64_t stride, int64_t rate, int same_padding)
{
    if (in <= 0 || k <= 0 || stride <= 0 || rate <= 0) return 0;
    if (same_padding) return (in + stride - 1) / stride;                 // ceil(in / stride)
    const int64_t keff = k + (k - 1) * (rate - 1);
    const int64_t span = in - keff + 1;
    return span <= 0 ? 0 : (span + stride - 1) / stride;                 // ceil((in - k_eff + 1) / stride)
}

int gpfq_extract_patches(const float *act, int64_t n, int64_t H, int64_t W, int64_t Cin, int64_t c,
                         int kh, int kw, int sh, int sw, int rh, int rw, int same_padding,
                         float *P, int64_t ldp, void *stream)
{
    if (n < 0 || H <= 0 || W <= 0 || Cin <= 0 || c < 0 || c >= Cin)
        return fail(GPFQ_ERR_INVALID_ARG, "bad activation shape/channel");
    if (kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || rh <= 0 || rw <= 0)
        return fail(GPFQ_ERR_INVALID_ARG, "bad kernel/stride/rate");
    const int64_t oh = gpfq_patch_out_dim(H, kh, sh, rh, same_padding);
    const int64_t ow = gpfq_patch_out_dim(W, kw, sw, rw, same_padding);
    const int64_t cols = n * oh * ow;
    if (cols == 0) return GPFQ_OK;
    if (!act || !P) return fail(GPFQ_ERR_INVALID_ARG, "NULL pointer");
    if (ldp < cols) return fail(GPFQ_ERR_INVALID_ARG, "patch pitch ldp=%lld < n*oh*ow=%lld", (long long)ldp, (long long)cols);
    int pad_top = 0, pad_left = 0;
    if (same_padding) {
        // TF SAME: total = max((out-1)*stride + k_eff - in, 0), before = total / 2 (floor)
        const int64_t keh = kh + (int64_t)(kh - 1) * (rh - 1), kew = kw + (int64_t)(kw - 1) * (rw - 1);
        int64_t th = (oh - 1) * sh + keh - H; if (th < 0) th = 0;
        int64_t tw = (ow - 1) * sw + kew - W; if (tw < 0) tw = 0;
        pad_top = (int)(th / 2);
        pad_left = (int)(tw / 2);
    }
    hipError_t e = gpfq::launch_extract_patches(act, n, H, W, Cin, c, kh, kw, sh, sw, rh, rw, pad_top, pad_left,
                                                oh, ow, P, ldp, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? GPFQ_OK : hip_fail(e, "gpfq_extract_patches");
}

int gpfq_channel_planes(const float *act, int64_t npos, int64_t Cin, int64_t c_lo, int64_t nch, float *planes, void *stream)
{
    if (npos < 0 || Cin <= 0 || c_lo < 0 || nch < 0 || c_lo + nch > Cin)
        return fail(GPFQ_ERR_INVALID_ARG, "bad channel range [%lld, %lld) of %lld", (long long)c_lo, (long long)(c_lo + nch), (long long)Cin);
    if (npos == 0 || nch == 0) return GPFQ_OK;
    if (!act || !planes) return fail(GPFQ_ERR_INVALID_ARG, "NULL pointer");
    hipError_t e = gpfq::launch_channel_planes(act, npos, Cin, c_lo, nch, planes, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? GPFQ_OK : hip_fail(e, "gpfq_channel_planes");
}

size_t gpfq_channel_sumsq_workspace_bytes(int64_t Cin) { return Cin > 0 ? gpfq::channel_sumsq_workspace_bytes(Cin) : 0; }

int gpfq_channel_sumsq(const float *act, int64_t n, int64_t H, int64_t W, int64_t Cin, int sh, int sw, double *sumsq,
                       void *workspace, size_t workspace_bytes, void *stream)
{
    if (n < 0 || H <= 0 || W <= 0 || Cin <= 0 || sh <= 0 || sw <= 0) return fail(GPFQ_ERR_INVALID_ARG, "bad shape or stride");
    if (!sumsq) return fail(GPFQ_ERR_INVALID_ARG, "NULL output");
    if (workspace_bytes < gpfq_channel_sumsq_workspace_bytes(Cin) || !workspace) return fail(GPFQ_ERR_WORKSPACE, "workspace too small");
    if (n > 0 && !act) return fail(GPFQ_ERR_INVALID_ARG, "NULL activations");
    hipError_t e = gpfq::launch_channel_sumsq(act, n, H, W, Cin, sh, sw, sumsq, workspace, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? GPFQ_OK : hip_fail(e, "gpfq_channel_sumsq");
}

size_t gpfq_channel_dead_workspace_bytes(int64_t Cin) { return Cin > 0 ? gpfq::channel_dead_workspace_bytes(Cin) : 0; }

int gpfq_channel_dead(const float *act, int64_t n, int64_t H, int64_t W, int64_t Cin, int sh, int sw, int64_t prefix_positions,
                      int32_t *dead, void *workspace, size_t workspace_bytes, void *stream)
{
    if (n < 0 || H <= 0 || W <= 0 || Cin <= 0 || sh <= 0 || sw <= 0 || prefix_positions < 0) return fail(GPFQ_ERR_INVALID_ARG, "bad shape, stride or prefix");
    if (!dead) return fail(GPFQ_ERR_INVALID_ARG, "NULL output");
    if (workspace_bytes < gpfq_channel_dead_workspace_bytes(Cin) || !workspace || (uintptr_t)workspace % 16 != 0)
        return fail(GPFQ_ERR_WORKSPACE, "workspace too small or misaligned");
    if (n > 0 && !act) return fail(GPFQ_ERR_INVALID_ARG, "NULL activations");
    hipError_t e = gpfq::launch_channel_dead(act, n, H, W, Cin, sh, sw, dead, workspace, prefix_positions, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? GPFQ_OK : hip_fail(e, "gpfq_channel_dead");
}

size_t gpfq_conv1x1_workspace_bytes(int64_t Cin)
{
    return Cin > 0 ? gpfq_channel_dead_workspace_bytes(Cin) + (((size_t)Cin * sizeof(int32_t) + 255) & ~(size_t)255) : 0;
}

int gpfq_quantize_conv1x1(const float *act_q, int64_t n, int64_t H, int64_t W, int64_t Cin, int sh, int sw, const float *Wt, int64_t F,
                          const double *alphabet, int M, float *Q, void *qidx, void *workspace, size_t workspace_bytes, void *stream)
{
    if (n < 0 || H <= 0 || W <= 0 || Cin <= 0 || F < 0 || sh <= 0 || sw <= 0) return fail(GPFQ_ERR_INVALID_ARG, "bad shape or stride");
    HostAlphabet A;
    int rc = make_alphabet(alphabet, M, -1, &A);
    if (rc != GPFQ_OK) return rc;
    if (F == 0) return GPFQ_OK;
    if (!Wt || (!Q && !qidx)) return fail(GPFQ_ERR_INVALID_ARG, "NULL pointer");
    if (n > 0 && !act_q) return fail(GPFQ_ERR_INVALID_ARG, "NULL activations");
    if (workspace_bytes < gpfq_conv1x1_workspace_bytes(Cin) || !workspace || (uintptr_t)workspace % 16 != 0)
        return fail(GPFQ_ERR_WORKSPACE, "workspace too small or misaligned");
    int zero_idx = -1;
    for (int k = 0; k < M; ++k)
        if (alphabet[k] == 0.0) zero_idx = k;
    int32_t *dead = reinterpret_cast<int32_t *>(static_cast<char *>(workspace) + gpfq_channel_dead_workspace_bytes(Cin));
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e = gpfq::launch_channel_dead(act_q, n, H, W, Cin, sh, sw, dead, workspace, 0, st);
    if (e == hipSuccess) e = gpfq::launch_msq(Wt, Cin * F, A.A, Q, static_cast<int8_t *>(qidx), st, A.big(), dead, F, zero_idx);
    return e == hipSuccess ? GPFQ_OK : hip_fail(e, "gpfq_quantize_conv1x1");
}

static size_t al256c(size_t x) { return (x + 255) & ~(size_t)255; }

size_t gpfq_conv_channels_workspace_bytes(int64_t n, int64_t H, int64_t W, int64_t nch, int kh, int kw, int sh, int sw,
                                          int rh, int rw, int same_padding, int64_t F, int want_resid)
{
    const int64_t oh = gpfq_patch_out_dim(H, kh, sh, rh, same_padding), ow = gpfq_patch_out_dim(W, kw, sw, rw, same_padding);
    const int64_t cols = n * oh * ow;
    const int64_t K = (int64_t)kh * kw;
    if (cols <= 0 || K <= 0 || F < 0 || nch < 0) return 0;
    if (!want_resid && g_conv_fused && gpfq::gram_image_supported(n, H, W, kh, kw, sh, sw, rh, rw, same_padding))
        return gpfq::gram_image_workspace_bytes(nch, F);
    if (!want_resid && g_conv_fused && gpfq::gram_conv_supported(n, H, W, nch, kh, kw, oh, ow)) {
        size_t b = al256c(gpfq::gram_conv_workspace_bytes(K, nch, F, cols));
        if (!same_padding && gpfq::gram_s2_supported(n, H, W, kh, kw, sh, sw, rh, rw, 0, 0, nch)) b += gpfq::gram_s2_workspace_bytes(n, H, W, nch);
        return b;
    }
    const int64_t ldp = (cols + 3) & ~(int64_t)3;
    return 2 * al256c((size_t)K * ldp * sizeof(float)) + al256c((size_t)K * sizeof(float)) + gpfq::gram_workspace_bytes(K, cols, F);
}

// phase 0: the whole channel loop; 1: Gram records only (-> records, negflags); 2: decide from given records
static int conv_channels_impl(int phase, double *records, int32_t *negflags,
                              const float *act_w, const float *act_q, int64_t n, int64_t H, int64_t W, int64_t nch,
                              int kh, int kw, int sh, int sw, int rh, int rw, int same_padding,
                              const float *Wt, const double *alphabet, int M, int zero_idx, int64_t F,
                              void *qidx_v, float *Qt, double *resid, int32_t *uncertified,
                              void *workspace, size_t workspace_bytes, void *stream, int64_t pix = 1)
{
    if (n < 0 || H <= 0 || W <= 0 || nch < 0 || F < 0) return fail(GPFQ_ERR_INVALID_ARG, "bad shape");
    if (kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || rh <= 0 || rw <= 0) return fail(GPFQ_ERR_INVALID_ARG, "bad kernel/stride/rate");
    HostAlphabet HA;
    if (phase != 1) {
        int rc = make_alphabet(alphabet, M, zero_idx, &HA);
        if (rc != GPFQ_OK) return rc;
    }
    const gpfq::AlphabetArg &A = HA.A;
    int8_t *qidx = static_cast<int8_t *>(qidx_v);          // int16 elements when HA.is_big
    const int64_t oh = gpfq_patch_out_dim(H, kh, sh, rh, same_padding), ow = gpfq_patch_out_dim(W, kw, sw, rw, same_padding);
    const int64_t cols = n * oh * ow, K = (int64_t)kh * kw;
    if (phase && (!records || !negflags)) return fail(GPFQ_ERR_INVALID_ARG, "NULL records / negflags");
    if (nch == 0 || (F == 0 && phase != 1) || cols == 0) return GPFQ_OK;
    if (K > GPFQ_GRAM_MAX_N || cols >= (1LL << 30)) return fail(GPFQ_ERR_UNSUPPORTED, "needs kh*kw <= %d and n*oh*ow < 2^30", GPFQ_GRAM_MAX_N);
    if (!act_w || !act_q) return fail(GPFQ_ERR_INVALID_ARG, "NULL pointer");
    if (phase != 1 && (!Wt || !qidx || !Qt || !uncertified)) return fail(GPFQ_ERR_INVALID_ARG, "NULL pointer");
    if (phase && resid) return fail(GPFQ_ERR_UNSUPPORTED, "residual norms need the whole call");
    const size_t need = gpfq_conv_channels_workspace_bytes(n, H, W, nch, kh, kw, sh, sw, rh, rw, same_padding, F, resid != nullptr);
    if (!workspace || workspace_bytes < need || (uintptr_t)workspace % 16 != 0)
        return fail(GPFQ_ERR_WORKSPACE, "conv channel loop needs %zu aligned workspace bytes", need);
    if (pix > 1 && !(same_padding == 0 && !resid && !phase && g_conv_fused && gpfq::gram_conv_supported(n, H, W, nch, kh, kw, oh, ow) &&
                     gpfq::gram_s2_supported(n, H, W, kh, kw, sh, sw, rh, rw, 0, 0, nch)))
        return fail(GPFQ_ERR_UNSUPPORTED, "NHWC activations: only the 7x7 / stride 2 / VALID shift-sum form reads them (gpfq_conv_channels_nhwc_supported)");
    if (pix == 1 && !resid && g_conv_fused && gpfq::gram_image_supported(n, H, W, kh, kw, sh, sw, rh, rw, same_padding)) {
        // 3x3 / stride 1: Gram matrices of every channel straight from the planes, one batched decide launch
        gpfq::ImageGramArgs g;
        g.act_w = act_w; g.act_q = act_q; g.n = n; g.H = H; g.W = W; g.nch = nch; g.pad = same_padding ? 1 : 0;
        g.Wt = Wt; g.A = A; g.big = HA.big(); g.F = F; g.qidx = qidx; g.Qt = Qt; g.uncertified = uncertified;
        g.workspace = workspace;
        g.slack = std::ldexp(1.0, g_gram_slack_log2);
        g.variant = g_conv_strip;
        g.shift_form = g_conv_shift;
        g.phase = phase; g.records = records; g.negflags = negflags;
        hipError_t e = gpfq::launch_gram_image(g, static_cast<hipStream_t>(stream));
        return e == hipSuccess ? GPFQ_OK : hip_fail(e, "gpfq_quantize_conv_channels(fused)");
    }
    int pad_top = 0, pad_left = 0;
    if (same_padding) {
        const int64_t keh = kh + (int64_t)(kh - 1) * (rh - 1), kew = kw + (int64_t)(kw - 1) * (rw - 1);
        int64_t th = (oh - 1) * sh + keh - H; if (th < 0) th = 0;
        int64_t tw = (ow - 1) * sw + kew - W; if (tw < 0) tw = 0;
        pad_top = (int)(th / 2);
        pad_left = (int)(tw / 2);
    }
    if (!resid && g_conv_fused && gpfq::gram_conv_supported(n, H, W, nch, kh, kw, oh, ow)) {
        // any other shape: the tile kernel gathers its patch rows from the planes (implicit im2col)
        gpfq::ConvGramArgs g;
        g.act_w = act_w; g.act_q = act_q; g.n = n; g.H = H; g.W = W; g.nch = nch;
        g.kh = kh; g.kw = kw; g.sh = sh; g.sw = sw; g.rh = rh; g.rw = rw; g.pt = pad_top; g.pl = pad_left; g.oh = oh; g.ow = ow;
        g.Wt = Wt; g.A = A; g.big = HA.big(); g.F = F; g.qidx = qidx; g.Qt = Qt; g.uncertified = uncertified;
        g.workspace = workspace;
        g.slack = std::ldexp(1.0, g_gram_slack_log2);
        g.variant = g_variant;
        g.phase = phase; g.records = records; g.negflags = negflags; g.pix = pix;
        if (pad_top == 0 && pad_left == 0 && gpfq::gram_s2_supported(n, H, W, kh, kw, sh, sw, rh, rw, pad_top, pad_left, nch))
            g.s2_part = reinterpret_cast<double *>(static_cast<char *>(workspace) + al256c(gpfq::gram_conv_workspace_bytes(K, nch, F, cols)));
        hipError_t e = gpfq::launch_gram_conv(g, static_cast<hipStream_t>(stream));
        return e == hipSuccess ? GPFQ_OK : hip_fail(e, "gpfq_quantize_conv_channels(implicit)");
    }
    if (phase) return fail(GPFQ_ERR_UNSUPPORTED, "records in / out need a kernel shape the plane kernels take");
    const int64_t ldp = (cols + 3) & ~(int64_t)3;
    char *ws = static_cast<char *>(workspace);
    float *Pw = reinterpret_cast<float *>(ws);   ws += al256c((size_t)K * ldp * sizeof(float));
    float *Pq = reinterpret_cast<float *>(ws);   ws += al256c((size_t)K * ldp * sizeof(float));
    float *nrm = reinterpret_cast<float *>(ws);  ws += al256c((size_t)K * sizeof(float));
    const bool same_act = act_w == act_q;          // first layer: both networks see the raw data (:478-481)
    const int64_t plane = n * H * W;
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (int64_t c = 0; c < nch; ++c) {
        hipError_t e = gpfq::launch_extract_patches(act_w + c * plane, n, H, W, 1, 0, kh, kw, sh, sw, rh, rw, pad_top, pad_left,
                                                    oh, ow, Pw, ldp, s);
        if (e == hipSuccess && !same_act)
            e = gpfq::launch_extract_patches(act_q + c * plane, n, H, W, 1, 0, kh, kw, sh, sw, rh, rw, pad_top, pad_left,
                                             oh, ow, Pq, ldp, s);
        if (e != hipSuccess) return hip_fail(e, "gpfq_quantize_conv_channels(patches)");
        gpfq::GramArgs a;
        a.X = Pw; a.Xq = same_act ? Pw : Pq; a.ld = ldp; a.nrm32 = nrm; a.nrm32_out = nrm;
        a.Wt = Wt + c * F * K; a.ldw = K; a.A = A; a.big = HA.big(); a.N = K; a.m = cols; a.C = F;
        a.qidx = HA.at(qidx, c * F * K); a.Qt = Qt + c * F * K; a.resid = resid ? resid + c * F : nullptr;
        a.uncertified = uncertified + c * F;
        a.workspace = ws;
        a.slack = std::ldexp(1.0, g_gram_slack_log2);
        a.variant = g_variant;
        e = gpfq::launch_gram(a, s);
        if (e != hipSuccess) return hip_fail(e, "gpfq_quantize_conv_channels(gram)");
    }
    return GPFQ_OK;
}

int gpfq_conv3x3_nhwc_supported(int64_t n, int64_t H, int64_t W, int64_t nch)
{
    return g_conv_fused && g_conv_nhwc && gpfq::gram_image_nhwc_supported(n, H, W, nch) ? 1 : 0;
}

size_t gpfq_conv3x3_nhwc_workspace_bytes(int64_t n, int64_t H, int64_t W, int64_t nch, int64_t F)
{
    if (!gpfq::gram_image_nhwc_supported(n, H, W, nch) || F < 0) return 0;
    return gpfq::gram_image_nhwc_workspace_bytes(n, H, W, nch, F);
}

int gpfq_quantize_conv3x3_nhwc(const float *act_w, const float *act_q, int64_t n, int64_t H, int64_t W, int64_t Cin, int64_t c_lo, int64_t nch,
                               const float *Wt, const double *alphabet, int M, int zero_idx, int64_t F,
                               void *qidx, float *Qt, int32_t *uncertified, void *workspace, size_t workspace_bytes, void *stream)
{
    if (n <= 0 || H <= 0 || W <= 0 || Cin <= 0 || c_lo < 0 || nch < 0 || c_lo + nch > Cin || F < 0) return fail(GPFQ_ERR_INVALID_ARG, "bad shape");
    HostAlphabet HA;
    int rc = make_alphabet(alphabet, M, zero_idx, &HA);
    if (rc != GPFQ_OK) return rc;
    if (nch == 0 || F == 0) return GPFQ_OK;
    if (!gpfq::gram_image_nhwc_supported(n, H, W, nch)) return fail(GPFQ_ERR_UNSUPPORTED, "NHWC form needs images of 4 x 4 or more and 32+ channels");
    if (!act_w || !act_q || !Wt || !qidx || !Qt || !uncertified) return fail(GPFQ_ERR_INVALID_ARG, "NULL pointer");
    const size_t need = gpfq_conv3x3_nhwc_workspace_bytes(n, H, W, nch, F);
    if (!workspace || workspace_bytes < need || (uintptr_t)workspace % 16 != 0)
        return fail(GPFQ_ERR_WORKSPACE, "NHWC conv call needs %zu aligned workspace bytes", need);
    gpfq::ImageGramArgs g;
    g.act_w = act_w + c_lo; g.act_q = act_q + c_lo; g.nhwc_cin = Cin;
    g.n = n; g.H = H; g.W = W; g.nch = nch; g.pad = 1;
    g.Wt = Wt; g.A = HA.A; g.big = HA.big(); g.F = F; g.qidx = static_cast<int8_t *>(qidx); g.Qt = Qt; g.uncertified = uncertified;
    g.workspace = workspace;
    g.slack = std::ldexp(1.0, g_gram_slack_log2);
    hipError_t e = gpfq::launch_gram_image_nhwc(g, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? GPFQ_OK : hip_fail(e, "gpfq_quantize_conv3x3_nhwc");
}

int gpfq_conv_records_supported(int64_t n, int64_t H, int64_t W, int64_t nch, int kh, int kw, int sh, int sw, int rh, int rw,
                                int same_padding)
{
    if (n <= 0 || H <= 0 || W <= 0 || nch <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || rh <= 0 || rw <= 0) return 0;
    const int64_t oh = gpfq_patch_out_dim(H, kh, sh, rh, same_padding), ow = gpfq_patch_out_dim(W, kw, sw, rw, same_padding);
    const int64_t cols = n * oh * ow, K = (int64_t)kh * kw;
    if (cols <= 0 || K > GPFQ_GRAM_MAX_N || cols >= (1LL << 30) || !g_conv_fused) return 0;
    return (gpfq::gram_image_supported(n, H, W, kh, kw, sh, sw, rh, rw, same_padding) ||
            gpfq::gram_conv_supported(n, H, W, nch, kh, kw, oh, ow)) ? 1 : 0;
}

int gpfq_quantize_conv_channels(const float *act_w, const float *act_q, int64_t n, int64_t H, int64_t W, int64_t nch,
                                int kh, int kw, int sh, int sw, int rh, int rw, int same_padding,
                                const float *Wt, const double *alphabet, int M, int zero_idx, int64_t F,
                                void *qidx, float *Qt, double *resid, int32_t *uncertified,
                                void *workspace, size_t workspace_bytes, void *stream)
{
    return conv_channels_impl(0, nullptr, nullptr, act_w, act_q, n, H, W, nch, kh, kw, sh, sw, rh, rw, same_padding, Wt, alphabet, M,
                              zero_idx, F, qidx, Qt, resid, uncertified, workspace, workspace_bytes, stream);
}

int gpfq_conv_channels_nhwc_supported(int64_t n, int64_t H, int64_t W, int64_t nch, int kh, int kw, int sh, int sw, int rh, int rw, int same_padding)
{
    if (n <= 0 || H <= 0 || W <= 0 || nch <= 0 || same_padding || !g_conv_fused || !g_conv_planes_free) return 0;
    const int64_t oh = gpfq_patch_out_dim(H, kh, sh, rh, 0), ow = gpfq_patch_out_dim(W, kw, sw, rw, 0);
    if ((int64_t)kh * kw > GPFQ_GRAM_MAX_N || n * oh * ow >= (1LL << 30)) return 0;
    return gpfq::gram_conv_supported(n, H, W, nch, kh, kw, oh, ow) && gpfq::gram_s2_supported(n, H, W, kh, kw, sh, sw, rh, rw, 0, 0, nch) ? 1 : 0;
}

int gpfq_quantize_conv_channels_nhwc(const float *act_w, const float *act_q, int64_t n, int64_t H, int64_t W, int64_t Cin, int64_t c_lo,
                                     int64_t nch, int kh, int kw, int sh, int sw, int rh, int rw, int same_padding,
                                     const float *Wt, const double *alphabet, int M, int zero_idx, int64_t F,
                                     void *qidx, float *Qt, int32_t *uncertified, void *workspace, size_t workspace_bytes, void *stream)
{
    if (Cin <= 0 || c_lo < 0 || nch < 0 || c_lo + nch > Cin) return fail(GPFQ_ERR_INVALID_ARG, "bad channel range");
    if (!act_w || !act_q) return fail(GPFQ_ERR_INVALID_ARG, "NULL pointer");
    if (Cin == 1)                                                   // one channel: the tensor IS its plane
        return conv_channels_impl(0, nullptr, nullptr, act_w, act_q, n, H, W, nch, kh, kw, sh, sw, rh, rw, same_padding, Wt, alphabet, M,
                                  zero_idx, F, qidx, Qt, nullptr, uncertified, workspace, workspace_bytes, stream);
    return conv_channels_impl(0, nullptr, nullptr, act_w + c_lo, act_q + c_lo, n, H, W, nch, kh, kw, sh, sw, rh, rw, same_padding, Wt, alphabet,
                              M, zero_idx, F, qidx, Qt, nullptr, uncertified, workspace, workspace_bytes, stream, Cin);
}

int gpfq_conv_channel_records(const float *act_w, const float *act_q, int64_t n, int64_t H, int64_t W, int64_t nch,
                              int kh, int kw, int sh, int sw, int rh, int rw, int same_padding,
                              double *records, int32_t *negflags, void *workspace, size_t workspace_bytes, void *stream)
{
    return conv_channels_impl(1, records, negflags, act_w, act_q, n, H, W, nch, kh, kw, sh, sw, rh, rw, same_padding, nullptr, nullptr, 0,
                              -1, 0, nullptr, nullptr, nullptr, nullptr, workspace, workspace_bytes, stream);
}

int gpfq_quantize_conv_channels_from_records(const double *records, const int32_t *negflags,
                                             const float *act_w, const float *act_q, int64_t n, int64_t H, int64_t W, int64_t nch,
                                             int kh, int kw, int sh, int sw, int rh, int rw, int same_padding,
                                             const float *Wt, const double *alphabet, int M, int zero_idx, int64_t F,
                                             void *qidx, float *Qt, int32_t *uncertified,
                                             void *workspace, size_t workspace_bytes, void *stream)
{
    return conv_channels_impl(2, const_cast<double *>(records), const_cast<int32_t *>(negflags), act_w, act_q, n, H, W, nch, kh, kw, sh, sw,
                              rh, rw, same_padding, Wt, alphabet, M, zero_idx, F, qidx, Qt, nullptr, uncertified, workspace,
                              workspace_bytes, stream);
}

}  // extern "C"
