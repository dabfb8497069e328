// wf_ctx.hip — context, error reporting, scratch management.
#include <stdarg.h>
#include <stdio.h>

#include "wf_common.h"

static thread_local char g_err[512] = "";

void wf_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *wf_version(void) { return "waveforms-amd 0.1.0 (gfx950)"; }
extern "C" const char *wf_last_error_string(void) { return g_err; }

extern "C" int wf_ctx_create(int device, wf_ctx **out)
{
    WF_REQUIRE(out != nullptr, "wf_ctx_create: out is NULL");
    int ndev = 0;
    WF_HIP(hipGetDeviceCount(&ndev));
    WF_REQUIRE(device >= 0 && device < ndev, "wf_ctx_create: device %d of %d", device, ndev);
    WF_HIP(hipSetDevice(device));
    wf_ctx *c = new wf_ctx();
    c->device = device;
    (void)hipDeviceGetAttribute(&c->cus, hipDeviceAttributeMultiprocessorCount, device);
    if (c->cus < 1) c->cus = 256;
    WF_HIP(hipMalloc(&c->d_fault, 64));
    WF_HIP(hipMemset(c->d_fault, 0, 64));
    WF_HIP(hipHostMalloc(&c->h_fault, 64, hipHostMallocDefault));
    WF_HIP(hipMalloc(&c->d_tables, 4096));
    WF_HIP(hipMalloc(&c->d_vit_unmerged, 4 * sizeof(unsigned long long)));      // [0] chunks left unproven, [1] chunk repairs run, [2] ... that handed on to the next chunk
    WF_HIP(hipMemset(c->d_vit_unmerged, 0, 4 * sizeof(unsigned long long)));
    WF_HIP(hipMalloc(&c->d_small, 256));
    WF_HIP(hipHostMalloc(&c->h_small, 256, hipHostMallocDefault));
    *out = c;
    int rc = wf_ctx_reserve_scan(c, 1 << 16);
    if (rc) return rc;
    return wf_ctx_reserve_fsm(c, 1 << 16);
}

extern "C" int wf_ctx_destroy(wf_ctx *c)
{
    if (!c) return WF_OK;
    (void)hipSetDevice(c->device);
    (void)wf_iter_server_stop(c);           // a persistent iteration server (if any) retires before the device-wide wait
    (void)hipDeviceSynchronize();
    for (auto &kv : c->lfsr) {
        if (kv.second->dev) (void)hipFree(kv.second->dev);
        delete kv.second;
    }
    if (c->events) {
        for (int k = 0; k < WF_LINK_EVENT_SLOTS * (WF_LINK_STAGES + 1); ++k) (void)hipEventDestroy(c->events[k]);
        delete[] c->events;
    }
    if (c->pipe_front) (void)hipEventDestroy(c->pipe_front);
    for (int k = 0; k < 2; ++k)
        if (c->pipe_done[k]) (void)hipEventDestroy(c->pipe_done[k]);
    if (c->pipe_stream) (void)hipStreamDestroy(static_cast<hipStream_t>(c->pipe_stream));
    if (c->pipe_stream2) (void)hipStreamDestroy(static_cast<hipStream_t>(c->pipe_stream2));
    if (c->pipe_pro) (void)hipStreamDestroy(static_cast<hipStream_t>(c->pipe_pro));
    if (c->pipe_main) (void)hipStreamDestroy(static_cast<hipStream_t>(c->pipe_main));
    if (c->pipe_call) (void)hipEventDestroy(c->pipe_call);
    if (c->pipe_pro_done) (void)hipEventDestroy(c->pipe_pro_done);
    if (c->d_scan) (void)hipFree(c->d_scan);
    if (c->d_fsm_scratch) (void)hipFree(c->d_fsm_scratch);
    if (c->d_mod_scratch) (void)hipFree(c->d_mod_scratch);
    if (c->d_vit_edge) (void)hipFree(c->d_vit_edge);
    if (c->d_vit_unmerged) (void)hipFree(c->d_vit_unmerged);
    if (c->d_fault) (void)hipFree(c->d_fault);
    if (c->h_fault) (void)hipHostFree(c->h_fault);
    if (c->d_tables) (void)hipFree(c->d_tables);
    if (c->tables_event) (void)hipEventDestroy(c->tables_event);
    if (c->d_small) (void)hipFree(c->d_small);
    if (c->h_small) (void)hipHostFree(c->h_small);
    if (c->h_iter) (void)hipHostFree(c->h_iter);
    if (c->iter_stream) (void)hipStreamDestroy(static_cast<hipStream_t>(c->iter_stream));
    if (c->h_mailbox) (void)hipHostFree(c->h_mailbox);
    delete c;
    return WF_OK;
}

// Retire a context WITHOUT freeing it (interpreter exit): the persistent iteration server leaves the device, the side
// stream of pipelined links drains, and the handle stays valid — objects that cached it may still call wf_link_join,
// wf_viterbi4_iteration_quiesce or wf_ctx_destroy from their finalisers.  Idempotent.
extern "C" int wf_ctx_retire(wf_ctx *c)
{
    if (!c) return WF_OK;
    (void)hipSetDevice(c->device);
    (void)wf_iter_server_stop(c);
    if (c->pipe_stream) (void)hipStreamSynchronize(static_cast<hipStream_t>(c->pipe_stream));
    if (c->pipe_stream2) (void)hipStreamSynchronize(static_cast<hipStream_t>(c->pipe_stream2));
    if (c->pipe_pro) (void)hipStreamSynchronize(static_cast<hipStream_t>(c->pipe_pro));
    if (c->pipe_main) (void)hipStreamSynchronize(static_cast<hipStream_t>(c->pipe_main));
    return WF_OK;
}

int wf_link_join_internal(wf_ctx *c, void *stream)
{
    if (!c) return WF_OK;
    for (int k = 0; k < 2; ++k)
        if (c->pipe_done_valid[k]) WF_HIP(hipStreamWaitEvent(wf_stream(stream), c->pipe_done[k], 0));
    return WF_OK;
}

extern "C" int wf_link_join(wf_ctx *c, void *stream)
{
    WF_REQUIRE(c != nullptr, "wf_link_join: ctx is NULL");
    return wf_link_join_internal(c, stream);
}

extern "C" int wf_ctx_set_option(wf_ctx *c, int key, int64_t value)
{
    WF_REQUIRE(c != nullptr, "wf_ctx_set_option: ctx is NULL");
    WF_REQUIRE(key >= 0 && key < WF_OPT_COUNT, "wf_ctx_set_option: unknown option %d", key);
    bool ok = true;
    switch (key) {
    case WF_OPT_CPM_FORM: ok = value >= 0 && value <= 2; break;
    case WF_OPT_CPM_CHUNK_CALLS: ok = value >= 0 && value <= 8192; break;
    case WF_OPT_DET_REPAIR: case WF_OPT_DET_FINAL_VERIFY: case WF_OPT_ITERATION_SERVER: ok = value == 0 || value == 1; break;
    case WF_OPT_MCB_TAIL_PERMILLE: ok = value >= -1 && value <= 16000; break;
    case WF_OPT_PIPE_RESERVE_CUS: ok = value >= -1 && value <= 128; break;
    case WF_OPT_CPM_SAMPLES_MIN_CALLS: ok = value == 0 || (value >= 4096 && value <= ((int64_t)1 << 40)); break;
    }
    WF_REQUIRE(ok, "wf_ctx_set_option: value %lld outside the range of option %d", (long long)value, key);
    c->opt[key] = value;
    return WF_OK;
}

extern "C" int wf_ctx_get_option(wf_ctx *c, int key, int64_t *value)
{
    WF_REQUIRE(c != nullptr && value != nullptr, "wf_ctx_get_option: NULL argument");
    WF_REQUIRE(key >= 0 && key < WF_OPT_COUNT, "wf_ctx_get_option: unknown option %d", key);
    *value = c->opt[key];
    return WF_OK;
}

extern "C" int wf_ctx_check(wf_ctx *c, void *stream)
{
    WF_REQUIRE(c != nullptr, "wf_ctx_check: ctx is NULL");
    {
        const int rj = wf_link_join_internal(c, stream);      // (pipelined links: their back ends are part of what is being checked)
        if (rj) return rj;
    }
    WF_HIP(hipMemcpyAsync(c->h_fault, c->d_fault, sizeof(unsigned), hipMemcpyDeviceToHost,
                          wf_stream(stream)));
    WF_HIP(hipStreamSynchronize(wf_stream(stream)));
    unsigned f = *c->h_fault;
    if (f) {
        WF_HIP(hipMemsetAsync(c->d_fault, 0, sizeof(unsigned), wf_stream(stream)));
        wf_set_error("device fault word 0x%x (bit0 = scan hand-off timeout)", f);
        return WF_ERR_DEVICE;
    }
    return WF_OK;
}

// See wf_common.h.  Verified on first sight of (kind, device pointers, sizes) on this context, never inside a stream capture
// (callers warm a context up eagerly before they capture: an unverified promise inside a capture is refused).  The verdict is
// remembered per ADDRESS, so whoever frees or rewrites such a table says so: wf_ctx_forget_promises (the Python links call it
// when they are created — an allocator may hand a new link's tables the addresses of a dead one's; round-5 advice).  No
// periodic re-check: it put a stream synchronisation into a pipelined link every 4096 blocks and closed no hole that the
// explicit call leaves open.
int wf_promise_verified(wf_ctx *c, int kind, const void *const *d_ptrs, const size_t *nbytes, int nptrs, void *stream,
                        bool (*check)(const unsigned char *const *host, const size_t *nbytes, const void *arg), const void *arg, const char *what)
{
    uint64_t key = 1469598103934665603ull;
    auto mix = [&](uint64_t v) {
        for (int b = 0; b < 8; ++b) {
            key ^= (v >> (8 * b)) & 0xFFull;
            key *= 1099511628211ull;
        }
    };
    mix((uint64_t)kind);
    for (int i = 0; i < nptrs; ++i) {
        WF_REQUIRE(d_ptrs[i] != nullptr && nbytes[i] > 0 && nbytes[i] <= (1u << 16), "%s: operand %d is NULL or larger than 64 KB", what, i);
        mix((uint64_t)reinterpret_cast<uintptr_t>(d_ptrs[i]));
        mix((uint64_t)nbytes[i]);
    }
    std::lock_guard<std::mutex> guard(c->promises_lock);
    auto it = c->promises.find(key);
    if (it != c->promises.end()) {
        ++it->second;
        return WF_OK;
    }
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(wf_stream(stream), &cap);
    if (cap != hipStreamCaptureStatusNone) {
        WF_REQUIRE(it != c->promises.end(), "%s: cannot be verified inside a stream capture — run the configuration once eagerly first", what);
        return WF_OK;
    }
    std::vector<std::vector<unsigned char>> host(nptrs);
    std::vector<const unsigned char *> hp(nptrs);
    for (int i = 0; i < nptrs; ++i) {
        host[i].resize(nbytes[i]);
        WF_HIP(hipMemcpyAsync(host[i].data(), d_ptrs[i], nbytes[i], hipMemcpyDeviceToHost, wf_stream(stream)));
        hp[i] = host[i].data();
    }
    WF_HIP(hipStreamSynchronize(wf_stream(stream)));
    if (!check(hp.data(), nbytes, arg)) {
        if (it != c->promises.end()) c->promises.erase(it);
        wf_set_error("%s", what);
        return WF_ERR_VALUE;
    }
    c->promises[key] = 1;
    return WF_OK;
}

extern "C" int wf_ctx_forget_promises(wf_ctx *c)
{
    WF_REQUIRE(c != nullptr, "wf_ctx_forget_promises: ctx is NULL");
    std::lock_guard<std::mutex> guard(c->promises_lock);
    c->promises.clear();
    return WF_OK;
}

// Grow-only scratch.  Growing synchronises the device (not capturable) — callers
// that capture graphs warm the context up with their largest size first.
int wf_ctx_reserve_scan(wf_ctx *c, size_t words)
{
    if (words <= c->scan_words) return WF_OK;
    WF_HIP(hipDeviceSynchronize());
    if (c->d_scan) WF_HIP(hipFree(c->d_scan));
    c->d_scan = nullptr;
    size_t cap = c->scan_words ? c->scan_words : (1 << 16);
    while (cap < words) cap *= 2;
    WF_HIP(hipMalloc(&c->d_scan, cap * sizeof(uint64_t)));
    c->scan_words = cap;
    return WF_OK;
}

int wf_ctx_reserve_fsm(wf_ctx *c, size_t words)
{
    if (words <= c->fsm_scratch_words) return WF_OK;
    WF_HIP(hipDeviceSynchronize());
    if (c->d_fsm_scratch) WF_HIP(hipFree(c->d_fsm_scratch));
    c->d_fsm_scratch = nullptr;
    size_t cap = c->fsm_scratch_words ? c->fsm_scratch_words : (1 << 16);
    while (cap < words) cap *= 2;
    WF_HIP(hipMalloc(&c->d_fsm_scratch, cap * sizeof(uint64_t)));
    c->fsm_scratch_words = cap;
    return WF_OK;
}

int wf_ctx_reserve_mod(wf_ctx *c, size_t words)
{
    if (words <= c->mod_scratch_words) return WF_OK;
    WF_HIP(hipDeviceSynchronize());
    if (c->d_mod_scratch) WF_HIP(hipFree(c->d_mod_scratch));
    c->d_mod_scratch = nullptr;
    size_t cap = c->mod_scratch_words ? c->mod_scratch_words : (1 << 16);
    while (cap < words) cap *= 2;
    WF_HIP(hipMalloc(&c->d_mod_scratch, cap * sizeof(double)));
    c->mod_scratch_words = cap;
    return WF_OK;
}

int wf_ctx_reserve_vit(wf_ctx *c, size_t words)
{
    if (words <= c->vit_edge_words) return WF_OK;
    WF_HIP(hipDeviceSynchronize());
    if (c->d_vit_edge) WF_HIP(hipFree(c->d_vit_edge));
    c->d_vit_edge = nullptr;
    size_t cap = c->vit_edge_words ? c->vit_edge_words : (1 << 14);
    while (cap < words) cap *= 2;
    WF_HIP(hipMalloc(&c->d_vit_edge, cap * sizeof(double)));
    c->vit_edge_words = cap;
    return WF_OK;
}
