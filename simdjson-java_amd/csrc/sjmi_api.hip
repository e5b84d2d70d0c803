// sjmi_api.hip -- the C ABI of libsjmi.so (see include/sjmi.h).  Host-side HIP runtime code only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <thread>
#include <mutex>
#include <new>
#include <string>
#include <system_error>
#include <utility>
#include <vector>

#include "stage1.h"

struct sjmi_ctx {
    int device = 0;
    uint64_t capacity = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;       // sjmi_parse_document: downloads the string records while the walker runs
    hipEvent_t strings_ready = nullptr;
    uint8_t* d_in = nullptr;      // capacity + padding
    uint32_t* d_idx = nullptr;    // capacity + 1 entries (host-buffer path)
    void* d_ws = nullptr;         // tile-state workspace
    size_t ws_bytes = 0;
    void* d_ws_dev = nullptr;     // workspace for the device-resident path: TWO halves used alternately (every launch
    size_t ws_dev_bytes = 0;      // zeroes the half the next one will use), grown on demand
    size_t ws_dev_clean[2] = {0, 0};  // bytes of each half known to be zero
    int ws_dev_next = 0;              // half the next launch uses
    void* ws_dev_last = nullptr;      // half the last launch used (debug read-back)
    sjmi_stage1_result* h_res = nullptr;  // pinned
    void* h_pack = nullptr;               // pinned: {error index, stage-1 record, string record} of sjmi_stage1_unescape
    uint8_t* staging = nullptr;           // sjmi_set_input_staging: the caller's page-locked copy of the input
    uint64_t staging_bytes = 0;
    void* d_pack = nullptr;
    void* d_res_tmp = nullptr;            // device stage-1 record of the same call
    uint64_t last_len = 0, last_count = 0;  // document of the last sjmi_stage1 call (still on the device)
    bool last_valid = false;
    uint8_t* d_sb = nullptr;      // string buffer (host path), grown on demand
    size_t sb_bytes = 0;
    void* d_ws_strm = nullptr;    // workspace of the streaming string pass (strings.hip), grown on demand
    size_t ws_strm_bytes = 0;
    void* d_ws_par = nullptr;     // workspace of a parity-only stage-1 launch (strings of a document this context has not indexed)
    size_t ws_par_bytes = 0;
    unsigned long long* d_blkpar = nullptr;  // in-string parity of every 64-byte block, left by the last stage-1 launch
    size_t blkpar_bytes = 0;                 // ... over (par_buf, par_len): what the string pass starts from
    const void* par_buf = nullptr;
    uint64_t par_len = 0;
    bool par_valid = false;
    unsigned long long* d_err_index = nullptr;  // host forms: position in indexes[] of the first failing string
    // what the cooperative walker takes from the string pass: offset of every record by string ordinal, the ordinal of the
    // first string at or behind every 64-byte block, the ordinal of every document's first string
    uint32_t* d_soff = nullptr;
    size_t soff_bytes = 0;
    uint32_t* d_blk_ord = nullptr;
    size_t blk_ord_bytes = 0;
    unsigned long long* d_doc_ord = nullptr;
    size_t doc_ord_bytes = 0;
    const void* soff_idx = nullptr;  // the index array these belong to (the last batch string pass on this context)
    sjmi_unescape_result* d_ures_walk = nullptr;  // ... and a copy of that pass's result record (for a walk queued later)
    uint8_t* d_copy = nullptr;       // isolated batches: the sanitized copy the string pass runs on, and its block parities
    size_t copy_bytes = 0;
    unsigned long long* d_blkpar2 = nullptr;
    size_t blkpar2_bytes = 0;
    sjmi_unescape_result* d_ures = nullptr;
    unsigned long long* d_docoff = nullptr;  // batch: document offsets + index offsets (+ statuses), grown on demand
    uint32_t* d_doccnt = nullptr;            // isolated batch: per-document index counts
    size_t doccnt_bytes = 0;
    unsigned long long* d_docstr = nullptr;  // batch: per-document string-buffer offsets
    size_t docstr_bytes = 0;
    void* d_ws_walk = nullptr;               // batch walk: scratch tape, tape lengths, chunk sums
    size_t ws_walk_bytes = 0;
    void* d_masks = nullptr;                 // sjmi_stage1_masks: 6 x u64 per block + its workspace, grown on demand
    size_t masks_bytes = 0;
    void* d_ws_masks = nullptr;
    size_t ws_masks_bytes = 0;
    void* d_single = nullptr;                // sjmi_parse_document: delimiters, tape offsets, error and results of ONE document
    struct HostView { const void* host = nullptr; void* dev = nullptr; };
    uint64_t pin_epoch = 0;                  // g_pin_epoch when the views below were taken (a (un)register anywhere drops them)
    HostView views[6];                       // device views of caller buffers that are page-locked and device-visible (zero-copy outputs)
    int view_next = 0;
    const uint32_t* idx_last = nullptr;      // where the last host-form stage-1 call left its indexes (c->d_idx, or the caller's array)
    const void* idx_last_host = nullptr;     // ... the caller's array as the host sees it (nullptr: c->d_idx)
    void* h_res_dev = nullptr;               // h_res / h_pack as the device sees them
    void* h_pack_dev = nullptr;
    void* s1_zero2 = nullptr;                // one-shot extras of the next stage1_device_impl call (sjmi_parse_document):
    size_t s1_zero2_bytes = 0;               //   a second region its workers zero, the single-document setup its scanner writes
    sjmi::Stage1Single s1_single;
    bool s1_single_done = false;             // ... the last launch was FAST and did write it
    const void* zc_host = nullptr;           // sjmi_parse_document: the caller's tape if it is device-visible (registered / pinned)
    unsigned long long* zc_dev = nullptr;    // ... as the device sees it
    void* h_single_dev = nullptr;            // h_single as the device sees it
    uint32_t* d_blkidx = nullptr;            // fused batch pipeline: k_stage1's per-block side outputs (stage1.h Stage1Extras)
    size_t blkidx_bytes = 0;
    uint16_t* d_blkw = nullptr;
    size_t blkw_bytes = 0;
    bool batch_side = false;                 // ... wanted from the next stage1_device_impl call
    uint32_t* d_batch_flags = nullptr;       // the two words of a batch's optimistic plain stage-1 pass ([1] != 0: accepted)
    const void* accept_buf = nullptr;        // ... which batch they belong to (the string pass of the same batch reads the flag)
    uint64_t accept_len = 0;
    bool accept_valid = false;
    unsigned long long* d_tape = nullptr;    // ... and its tape, grown on demand
    size_t tape_bytes = 0;
    void* h_single = nullptr;                // pinned copy of the three result records
    uint64_t last_ndocs = 0;                 // documents of the last batch call (their index offsets are still on the device)
    bool last_batch = false;
    size_t docoff_bytes = 0;
    int forced_steps = 0;
    uint32_t dbg = 0;  // ablation flags (sjmi_debug_set_flags)
    bool ticket_mode = false;  // safe tile assignment (latched on after a look-back timeout in fast mode)
    bool auto_safe = false;    // sjmi_set_auto_safe: the *_device stage-1 entry point synchronises, checks and re-runs in SAFE mode
    hipEvent_t busy_event = nullptr;  // recorded behind this context's last stage-1 launch (is it still running?)
    std::atomic<bool> launched{false};
    bool profiling = false;  // bracket every stage-1 kernel with HIP events (bench.py roofline)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t events_used = 0;
    std::string err;
};

namespace {

// contexts alive in this process: with more than one, two persistent stage-1 kernels may be resident at the same
// time, each with only part of its grid -- the kernels then take every granule by ticket (FLAG_ALL_TICKETS), so that
// a workgroup that has not started holds nothing the others could wait for (costs ~6 % on a GPU it has to itself)
std::atomic<int> g_live_contexts{0};
// Contexts of this process, so that a launch can ask whether ANOTHER context's persistent stage-1 kernel may still be
// running: only then do two kernels compete for residency and only then is the static first granule given up
// (FLAG_ALL_TICKETS, ~6 % slower).  A context that exists but is idle (parseBatch's second feeder context between
// batches) costs nothing.  hipStreamQuery is a non-blocking poll of the other context's last launch stream.
std::mutex g_registry_mutex;
std::vector<sjmi_ctx*> g_registry;
bool another_context_busy(const sjmi_ctx* self) {
    if (g_live_contexts.load() <= 1) return false;
    std::lock_guard<std::mutex> g(g_registry_mutex);
    for (sjmi_ctx* o : g_registry)
        if (o != self && o->launched.load(std::memory_order_acquire) && o->device == self->device &&
            hipEventQuery(o->busy_event) == hipErrorNotReady) {
            (void)hipGetLastError();  // (hipErrorNotReady is an answer, not an error to leave behind)
            return true;
        }
    return false;
}
uint32_t launch_flags(const sjmi_ctx* c) {
    return c->dbg | (c->ticket_mode ? sjmi::FLAG_SAFE : 0u) | (another_context_busy(c) ? sjmi::FLAG_ALL_TICKETS : 0u);
}
// called right behind a stage-1 launch: a library-owned event on the launch stream is what other contexts poll (never a
// stream they do not own, which may be gone by then); under the registry lock, like the poll and like sjmi_destroy
void note_launch(sjmi_ctx* c, hipStream_t st) {
    std::lock_guard<std::mutex> g(g_registry_mutex);
    if (!c->busy_event && hipEventCreateWithFlags(&c->busy_event, hipEventDisableTiming) != hipSuccess) {
        c->busy_event = nullptr;
        return;
    }
    if (hipEventRecord(c->busy_event, st) == hipSuccess) c->launched.store(true, std::memory_order_release);
}

bool fail(sjmi_ctx* c, const char* what, hipError_t e) {
    if (e == hipSuccess) return false;
    char b[256];
    snprintf(b, sizeof b, "%s: %s", what, hipGetErrorString(e));
    c->err = b;
    return true;
}


}  // namespace

extern "C" {

const char* sjmi_version(void) { return "sjmi 0.1 (gfx950)"; }

int sjmi_create(sjmi_ctx** out, int device, uint64_t capacity_bytes) {
    if (!out) return SJMI_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return SJMI_ERR_NO_DEVICE;
    sjmi_ctx* c = new (std::nothrow) sjmi_ctx();
    if (!c) return SJMI_ERR_ARG;
    g_live_contexts.fetch_add(1);  // (sjmi_destroy, also on the failure paths below, takes it back)
    c->device = device;
    c->capacity = capacity_bytes;
    const size_t in_bytes = ((capacity_bytes + 63) / 64) * 64 + 2 * SJMI_PADDING;
    c->ws_bytes = sjmi::stage1_workspace_bytes(capacity_bytes, 1);
    if (fail(c, "hipSetDevice", hipSetDevice(device)) ||
        fail(c, "hipStreamCreate", hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) ||
        fail(c, "hipMalloc(in)", hipMalloc((void**)&c->d_in, in_bytes)) ||
        fail(c, "hipMalloc(idx)", hipMalloc((void**)&c->d_idx, (capacity_bytes + 2) * sizeof(uint32_t))) ||
        fail(c, "hipMalloc(ws)", hipMalloc(&c->d_ws, c->ws_bytes)) ||
        fail(c, "hipHostMalloc", hipHostMalloc((void**)&c->h_res, sizeof(sjmi_stage1_result)))) {
        fprintf(stderr, "sjmi_create: %s\n", c->err.c_str());
        sjmi_destroy(c);
        return SJMI_ERR_HIP;
    }
    const char* mode = getenv("SJMI_TILE_MODE");  // "ticket" = start in the safe tile-assignment mode
    c->ticket_mode = mode && strcmp(mode, "ticket") == 0;
    {
        std::lock_guard<std::mutex> g(g_registry_mutex);
        g_registry.push_back(c);
    }
    *out = c;
    return SJMI_OK;
}

void sjmi_destroy(sjmi_ctx* c) {
    if (!c) return;
    {
        std::lock_guard<std::mutex> g(g_registry_mutex);
        for (size_t i = 0; i < g_registry.size(); ++i)
            if (g_registry[i] == c) {
                g_registry.erase(g_registry.begin() + i);
                break;
            }
    }
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->busy_event) (void)hipEventDestroy(c->busy_event);
    if (c->d_in) (void)hipFree(c->d_in);
    if (c->d_idx) (void)hipFree(c->d_idx);
    if (c->d_ws) (void)hipFree(c->d_ws);
    if (c->d_ws_dev) (void)hipFree(c->d_ws_dev);
    if (c->d_sb) (void)hipFree(c->d_sb);
    if (c->d_ws_strm) (void)hipFree(c->d_ws_strm);
    if (c->d_ws_par) (void)hipFree(c->d_ws_par);
    if (c->d_blkpar) (void)hipFree(c->d_blkpar);
    if (c->d_err_index) (void)hipFree(c->d_err_index);
    if (c->d_soff) (void)hipFree(c->d_soff);
    if (c->d_ures_walk) (void)hipFree(c->d_ures_walk);
    if (c->d_blk_ord) (void)hipFree(c->d_blk_ord);
    if (c->d_doc_ord) (void)hipFree(c->d_doc_ord);
    if (c->d_copy) (void)hipFree(c->d_copy);
    if (c->d_blkpar2) (void)hipFree(c->d_blkpar2);
    if (c->d_ures) (void)hipFree(c->d_ures);
    if (c->d_docoff) (void)hipFree(c->d_docoff);
    if (c->d_doccnt) (void)hipFree(c->d_doccnt);
    if (c->d_docstr) (void)hipFree(c->d_docstr);
    if (c->d_ws_walk) (void)hipFree(c->d_ws_walk);
    if (c->d_single) (void)hipFree(c->d_single);
    if (c->d_batch_flags) (void)hipFree(c->d_batch_flags);
    if (c->d_blkidx) (void)hipFree(c->d_blkidx);
    if (c->d_blkw) (void)hipFree(c->d_blkw);
    if (c->d_tape) (void)hipFree(c->d_tape);
    if (c->h_single) (void)hipHostFree(c->h_single);
    if (c->d_masks) (void)hipFree(c->d_masks);
    if (c->d_ws_masks) (void)hipFree(c->d_ws_masks);
    if (c->h_res) (void)hipHostFree(c->h_res);
    if (c->h_pack) (void)hipHostFree(c->h_pack);
    if (c->d_pack) (void)hipFree(c->d_pack);
    if (c->d_res_tmp) (void)hipFree(c->d_res_tmp);
    for (auto& e : c->events) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->strings_ready) (void)hipEventDestroy(c->strings_ready);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    g_live_contexts.fetch_sub(1);
}

const char* sjmi_last_error(const sjmi_ctx* c) { return c ? c->err.c_str() : "null context"; }

#ifdef SJMI_TRACE
extern "C" int sjmi_debug_read_ws(sjmi_ctx* c, void* dst, uint64_t offset, uint64_t bytes) {
    if (!c || !c->ws_dev_last || offset + bytes > c->ws_dev_bytes / 2) return SJMI_ERR_ARG;
    (void)hipDeviceSynchronize();
    return hipMemcpy(dst, (uint8_t*)c->ws_dev_last + offset, bytes, hipMemcpyDeviceToHost) == hipSuccess ? SJMI_OK : SJMI_ERR_HIP;
}
#endif

int sjmi_set_tile_steps(sjmi_ctx* c, int steps) {
    if (!c || !(steps == 0 || steps == 1 || steps == 2 || steps == 4)) return SJMI_ERR_ARG;
    c->forced_steps = steps;
    return SJMI_OK;
}

namespace {
void* parity_out(sjmi_ctx* c, const void* d_buf, uint64_t len);
}

int sjmi_set_input_staging(sjmi_ctx* c, void* pinned, uint64_t bytes) {
    if (!c || (pinned && !bytes)) return SJMI_ERR_ARG;
    c->staging = static_cast<uint8_t*>(pinned);
    c->staging_bytes = pinned ? bytes : 0;
    return SJMI_OK;
}
// H2D of a single document's bytes into c->d_in, queued on c->stream.  Through the caller's staging buffer when one is set
// (include/sjmi.h): small documents in one piece, large ones in chunks -- the DMA of chunk i runs while the host copies
// chunk i + 1 (64 MiB: 2.7 ms of memcpy + 1.2 ms of PCIe become ~2.8 ms).
static bool upload_document(sjmi_ctx* c, const uint8_t* buf, uint64_t len) {
    if (!len) return true;
    if (c->staging && buf != c->staging && len <= c->staging_bytes) {
        const uint64_t chunk = len >= (4ull << 20) ? (2ull << 20) : len;
        // one thread copies ~25 GB/s, PCIe takes 57: from 16 MiB on a second thread copies (and queues) every other chunk (8 MiB: no gain, the thread costs what it saves)
        std::atomic<bool> ok{true};
        auto part = [&](uint64_t first, uint64_t stride, bool set_device) {
            if (set_device && hipSetDevice(c->device) != hipSuccess) {
                ok = false;
                return;
            }
            for (uint64_t off = first * chunk; off < len; off += stride * chunk) {
                const uint64_t n = len - off < chunk ? len - off : chunk;
                memcpy(c->staging + off, buf + off, n);
                if (hipMemcpyAsync((uint8_t*)c->d_in + off, c->staging + off, n, hipMemcpyHostToDevice, c->stream) != hipSuccess) ok = false;
            }
        };
        bool split = len >= (16ull << 20);
        if (split) {
            std::thread helper;
            try {  // (thread creation can fail -- EAGAIN: no exception may cross the C ABI into a JVM / ctypes caller)
                helper = std::thread(part, (uint64_t)1, (uint64_t)2, true);
            } catch (const std::system_error&) {
                split = false;
            }
            if (split) {
                part(0, 2, false);
                helper.join();
            }
        }
        if (!split) part(0, 1, false);
        if (!ok) {
            c->err = "H2D (staged upload) failed";
            return false;
        }
        return true;
    }
    return !fail(c, "H2D", hipMemcpyAsync(c->d_in, buf, len, hipMemcpyHostToDevice, c->stream));
}
// sjmi_set_auto_safe's per-launch check switched off for the lifetime of the object (a call that re-runs in SAFE mode itself)
struct AutoSafeOff {
    sjmi_ctx* c;
    bool keep;
    explicit AutoSafeOff(sjmi_ctx* ctx) : c(ctx), keep(ctx->auto_safe) { ctx->auto_safe = false; }
    ~AutoSafeOff() { c->auto_safe = keep; }
};
// The device's view of a caller buffer that is page-locked and device-visible (sjmi_host_register / hipHostMalloc), or nullptr
// (pageable memory: the staged path).  The kernels then write their outputs straight into it over PCIe -- no download, and no
// second host synchronisation whose only purpose was to learn how much to download.  Cached per pointer (a handful of buffers
// per parser).  SJMI_ZERO_COPY=0 switches every zero-copy path off.
static std::atomic<uint64_t> g_pin_epoch{1};  // bumped by sjmi_host_register / sjmi_host_unregister: cached views are stale
static void drop_stale_views(sjmi_ctx* c) {
    const uint64_t e = g_pin_epoch.load(std::memory_order_acquire);
    if (c->pin_epoch == e) return;
    c->pin_epoch = e;
    for (auto& v : c->views) v = {};
    c->zc_host = nullptr;
    c->zc_dev = nullptr;
    // "the indexes of the last call" may sit in a caller array through a view that is gone now (unregistered, maybe unmapped):
    // the two-call form (sjmi_unescape) must not dereference it -- they ask for a new stage-1 call instead
    if (c->idx_last && c->idx_last != c->d_idx) {
        void* dp = nullptr;
        const bool still = c->idx_last_host && hipHostGetDevicePointer(&dp, const_cast<void*>(c->idx_last_host), 0) == hipSuccess &&
                           dp == (const void*)c->idx_last;
        if (!still) {
            (void)hipGetLastError();
            c->idx_last = nullptr;
            c->idx_last_host = nullptr;
            c->last_valid = false;
        }
    }
}
static void* device_view(sjmi_ctx* c, const void* host) {
    static const bool off = getenv("SJMI_ZERO_COPY") && atoi(getenv("SJMI_ZERO_COPY")) == 0;
    if (off || !host) return nullptr;
    drop_stale_views(c);
    for (const auto& v : c->views)
        if (v.host == host) return v.dev;
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, const_cast<void*>(host), 0) != hipSuccess || !dp) {
        (void)hipGetLastError();  // (not device-visible: not an error -- and not cached: the buffer may be registered a moment later)
        return nullptr;
    }
    c->views[c->view_next].host = host;
    c->views[c->view_next].dev = dp;
    c->view_next = (c->view_next + 1) % 6;
    return dp;
}

static int stage1_device_impl(sjmi_ctx* c, const void* d_buf, uint64_t len, void* d_indexes, uint64_t index_capacity,
                              void* d_result, void* stream, uint32_t shard_flags);
int sjmi_stage1(sjmi_ctx* c, const uint8_t* buf, uint64_t len, uint32_t* indexes, uint64_t index_capacity,
                uint64_t* count, uint32_t* status) {
    if (!c || (!buf && len) || !indexes || !count || !status) return SJMI_ERR_ARG;
    if (len > c->capacity || len >= (1ull << 32)) {
        c->err = "document larger than the context capacity";
        return SJMI_ERR_CAPACITY;
    }
    if (index_capacity < 1) return SJMI_ERR_CAPACITY;
    if (fail(c, "hipSetDevice", hipSetDevice(c->device))) return SJMI_ERR_HIP;
    // padIfNeeded (SimdJsonParser.java:42-48): only buf[0,len) is ever read from the caller
    if (!upload_document(c, buf, len)) return SJMI_ERR_HIP;
    const uint64_t dev_cap = c->capacity + 2 < index_capacity ? c->capacity + 2 : index_capacity;
    if (!c->d_res_tmp && fail(c, "hipMalloc(result)", hipMalloc((void**)&c->d_res_tmp, 64))) return SJMI_ERR_HIP;
    const uint64_t spec_idx = len + 2 < dev_cap ? len + 2 : dev_cap;
    const bool small = len <= (4u << 10);
    const AutoSafeOff own_retry(c);  // (the retry below is this call's own)
    // zero-copy: the caller's index array is device-visible -> the kernel writes indexes[0..count] there, the scanner the record
    // into the pinned result page: one synchronisation, no download
    uint32_t* const zc_idx = ((uintptr_t)indexes & 15) ? nullptr : (uint32_t*)device_view(c, indexes);
    if (!c->h_res_dev) c->h_res_dev = device_view(c, c->h_res);
    if (zc_idx && c->h_res_dev) {
        for (int attempt = 0; attempt < 2; ++attempt) {
            const int rc = stage1_device_impl(c, c->d_in, len, zc_idx, dev_cap, c->h_res_dev, c->stream, 0);
            if (rc != SJMI_OK) return rc;
            if (fail(c, "sync", hipStreamSynchronize(c->stream))) return SJMI_ERR_HIP;
            if (!(c->h_res->status & SJMI_ST_INTERNAL) || c->ticket_mode) break;
            c->ticket_mode = true;
        }
        *status = c->h_res->status & 0xFFu;
        *count = c->h_res->count;
        if (c->h_res->status & SJMI_ST_INTERNAL) {
            c->err = "look-back timeout";
            return SJMI_ERR_INTERNAL;
        }
        if (c->h_res->status & SJMI_ST_CAPACITY) {
            c->err = "index_capacity too small";
            return SJMI_ERR_CAPACITY;
        }
        c->idx_last = zc_idx;
        c->idx_last_host = indexes;
        c->last_len = len;
        c->last_count = c->h_res->count;
        c->last_valid = true;
        c->last_batch = false;
        return SJMI_OK;
    }
    c->idx_last = c->d_idx;
    c->idx_last_host = nullptr;
    for (int attempt = 0; attempt < 2; ++attempt) {
        // (the device entry point: double-buffered workspace, nothing but the kernel is queued once the context is warm)
        const int rc = stage1_device_impl(c, c->d_in, len, c->d_idx, dev_cap, c->d_res_tmp, c->stream, 0);
        if (rc != SJMI_OK) return rc;
        // (a small document's indexes come back behind the same synchronisation, by their bound of one structural per byte)
        if (fail(c, "D2H(result)", hipMemcpyAsync(c->h_res, c->d_res_tmp, sizeof(sjmi_stage1_result), hipMemcpyDeviceToHost, c->stream)) ||
            (small && fail(c, "D2H(indexes)", hipMemcpyAsync(indexes, c->d_idx, spec_idx * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream))) ||
            fail(c, "sync", hipStreamSynchronize(c->stream)))
            return SJMI_ERR_HIP;
        if (!(c->h_res->status & SJMI_ST_INTERNAL) || c->ticket_mode) break;
        c->ticket_mode = true;  // fast-mode liveness assumption failed: latch the safe mode and run again
    }
    *status = c->h_res->status & 0xFFu;
    *count = c->h_res->count;
    if (c->h_res->status & SJMI_ST_INTERNAL) {
        c->err = "look-back timeout";
        return SJMI_ERR_INTERNAL;
    }
    if (c->h_res->status & SJMI_ST_CAPACITY) {
        c->err = "index_capacity too small";
        return SJMI_ERR_CAPACITY;
    }
    if (!(small && c->h_res->count + 1 <= spec_idx) &&
        (fail(c, "D2H(indexes)",
              hipMemcpyAsync(indexes, c->d_idx, (c->h_res->count + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost,
                             c->stream)) ||
         fail(c, "sync", hipStreamSynchronize(c->stream))))
        return SJMI_ERR_HIP;
    c->last_len = len;
    c->last_count = c->h_res->count;
    c->last_valid = true;
    c->last_batch = false;
    return SJMI_OK;
}

namespace {
// document offsets of a batch come across the public ABI: [0] == 0 (isolated mode), monotonic, [n] <= total_len
bool bad_offsets(sjmi_ctx* c, const uint64_t* offs, uint64_t n_docs, uint64_t total_len, bool first_is_zero) {
    bool bad = (first_is_zero && n_docs && offs[0] != 0) || offs[n_docs] > total_len;
    for (uint64_t k = 0; k < n_docs && !bad; ++k) bad = offs[k + 1] < offs[k];
    if (bad) c->err = "doc_offsets must be monotonic, start at 0 and end at or before total_len";
    return bad;
}
bool grow(sjmi_ctx* c, void** p, size_t* have, size_t need, const char* what) {
    if (need <= *have) return true;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *have = 0;
    if (fail(c, what, hipMalloc(p, need))) return false;
    *have = need;
    return true;
}
// Where a plain stage-1 launch over (d_buf, len) leaves its block parities (null: the allocation failed -- the string pass
// then makes its own).  Shards and the per-document passes of an isolated batch do not record any.
void* parity_out(sjmi_ctx* c, const void* d_buf, uint64_t len) {
    c->par_valid = false;
    c->accept_valid = false;  // (another stage-1 launch: a batch's acceptance flag no longer describes the parities)
    if (!grow(c, (void**)&c->d_blkpar, &c->blkpar_bytes, sjmi::strings_parity_words(len) * sizeof(unsigned long long), "hipMalloc(blkpar)"))
        return nullptr;
    c->par_buf = d_buf;
    c->par_len = len;
    c->par_valid = true;
    return c->d_blkpar;
}
// The block parities of (d_buf, len) for the string pass: those of the context's last stage-1 launch if it was over the same
// bytes, else a stage-1 launch that writes nothing but them (no indexes, no sentinel), queued on `st`.
const unsigned long long* parity_for(sjmi_ctx* c, const void* d_buf, uint64_t len, hipStream_t st) {
    if (c->par_valid && c->par_buf == d_buf && c->par_len == len) return c->d_blkpar;
    const int steps = c->forced_steps ? c->forced_steps : sjmi::stage1_pick_steps(len);
    if (!grow(c, &c->d_ws_par, &c->ws_par_bytes, sjmi::stage1_workspace_bytes(len, steps), "hipMalloc(ws_par)")) return nullptr;
    sjmi::Stage1Extras ex;
    ex.blkpar = parity_out(c, d_buf, len);
    if (!ex.blkpar) return nullptr;
    c->par_valid = false;  // (until the launch is queued)
    if (fail(c, "parity launch", sjmi::stage1_launch((const uint8_t*)d_buf, len, nullptr, 0, c->d_ws_par, steps, st, nullptr, nullptr,
                                                     (launch_flags(c) & ~sjmi::DBG_NO_LOOKBACK) | sjmi::DBG_NO_WRITE, ex)))
        return nullptr;
    note_launch(c, st);
    c->par_valid = true;
    return c->d_blkpar;
}
}  // namespace

// the streaming string pass over (d_buf, len); optional: record offsets by ordinal / ordinals by block
// d_result == nullptr: the record inside the pass's workspace (zeroed with it: no memset of its own), returned in *used
static int strings_device_impl(sjmi_ctx* c, const void* d_buf, uint64_t len, void* d_string_buffer, uint64_t string_capacity,
                               uint32_t* d_soff, uint64_t soff_cap, uint32_t* d_blk_ord, void* d_result, hipStream_t st,
                               sjmi::UnescapeResult** used = nullptr, bool ws_is_zero = false) {
    const unsigned long long* par = parity_for(c, d_buf, len, st);
    if (!par) return SJMI_ERR_HIP;
    if (!grow(c, &c->d_ws_strm, &c->ws_strm_bytes, sjmi::strings_workspace_bytes(len), "hipMalloc(ws_strm)")) return SJMI_ERR_HIP;
    const bool own = d_result == nullptr;
    if (own) d_result = sjmi::strings_workspace_result(c->d_ws_strm);
    if (used) *used = (sjmi::UnescapeResult*)d_result;
    if ((!own && fail(c, "memset(result)", hipMemsetAsync(d_result, 0, sizeof(sjmi_unescape_result), st))) ||
        fail(c, "strings launch",
             sjmi::strings_launch((const uint8_t*)d_buf, len, par, (uint8_t*)d_string_buffer, string_capacity, d_soff, soff_cap,
                                  d_blk_ord, c->d_ws_strm, (sjmi::UnescapeResult*)d_result, st, nullptr, nullptr, sjmi::StringsAlt(), ws_is_zero)))
        return SJMI_ERR_HIP;
    return SJMI_OK;
}

// The string pass of a BATCH with everything the walkers need: the record table (c->d_soff), the ordinal of every document's
// first string (c->d_doc_ord) and, optionally, the offset of its first record (d_doc_str_offsets, n_docs + 1 entries).
//   plain     the batch was indexed by one plain stage-1 launch of this context (its block parities are on the device) and
//             every document passed: the pass runs over the batch itself;
//   otherwise the documents were indexed one by one (isolated mode): it runs over the sanitized copy (strings.hip), whose
//             parities come from a stage-1 launch that writes nothing else;
//   d_accept  the fused pipeline's device flag: != 0 -> plain, == 0 -> the copy; decided on the device, everything is queued.
static int strings_batch_impl(sjmi_ctx* c, const void* d_buf, uint64_t total_len, const void* d_indexes, uint64_t count_bound,
                              const void* d_doc_offsets, const void* d_index_offsets, uint64_t n_docs, bool plain,
                              const uint32_t* d_accept, void* d_string_buffer, uint64_t string_capacity, void* d_doc_str_offsets,
                              void* d_result, hipStream_t st) {
    const uint64_t soff_cap = count_bound + 64;
    if (!grow(c, (void**)&c->d_soff, &c->soff_bytes, soff_cap * sizeof(uint32_t), "hipMalloc(soff)") ||
        !grow(c, (void**)&c->d_blk_ord, &c->blk_ord_bytes, (total_len / 64 + 2) * sizeof(uint32_t), "hipMalloc(blk_ord)") ||
        !grow(c, (void**)&c->d_doc_ord, &c->doc_ord_bytes, (n_docs + 2) * sizeof(unsigned long long), "hipMalloc(doc_ord)") ||
        !grow(c, &c->d_ws_strm, &c->ws_strm_bytes, sjmi::strings_workspace_bytes(total_len), "hipMalloc(ws_strm)"))
        return SJMI_ERR_HIP;
    c->soff_idx = nullptr;
    const uint8_t* buf0 = (const uint8_t*)d_buf;
    const unsigned long long* par0 = nullptr;
    sjmi::StringsAlt alt;
    if (plain || d_accept) {
        if (!(c->par_valid && c->par_buf == d_buf && c->par_len == total_len)) {
            c->err = "string pass of a plain batch: the block parities of its stage-1 launch are not on this context";
            return SJMI_ERR_ARG;
        }
        par0 = c->d_blkpar;
    }
    if (!plain || d_accept) {
        const int steps = c->forced_steps ? c->forced_steps : sjmi::stage1_pick_steps(total_len);
        if (!grow(c, (void**)&c->d_copy, &c->copy_bytes, total_len + 2 * SJMI_PADDING + 64, "hipMalloc(copy)") ||
            !grow(c, (void**)&c->d_blkpar2, &c->blkpar2_bytes, sjmi::strings_parity_words(total_len) * sizeof(unsigned long long), "hipMalloc(blkpar2)") ||
            !grow(c, &c->d_ws_par, &c->ws_par_bytes, sjmi::stage1_workspace_bytes(total_len, steps), "hipMalloc(ws_par)"))
            return SJMI_ERR_HIP;
        sjmi::Stage1Extras ex;
        ex.blkpar = c->d_blkpar2;
        ex.skip = d_accept;
        if (fail(c, "sanitize", sjmi::strings_sanitize_launch((const uint8_t*)d_buf, total_len, (const unsigned long long*)d_doc_offsets,
                                                              (const unsigned long long*)d_index_offsets, n_docs, c->d_copy, d_accept, st)) ||
            fail(c, "parity launch", sjmi::stage1_launch(c->d_copy, total_len, nullptr, 0, c->d_ws_par, steps, st, nullptr, nullptr,
                                                         (launch_flags(c) & ~sjmi::DBG_NO_LOOKBACK) | sjmi::DBG_NO_WRITE, ex)))
            return SJMI_ERR_HIP;
        note_launch(c, st);
        if (d_accept) {
            alt.d_sel = d_accept;
            alt.d_buf = c->d_copy;
            alt.d_blkpar = c->d_blkpar2;
        } else {
            buf0 = c->d_copy;
            par0 = c->d_blkpar2;
        }
    }
    if (fail(c, "memset(result)", hipMemsetAsync(d_result, 0, sizeof(sjmi_unescape_result), st)) ||
        fail(c, "strings launch",
             sjmi::strings_launch(buf0, total_len, par0, (uint8_t*)d_string_buffer, string_capacity, c->d_soff, soff_cap, c->d_blk_ord,
                                  c->d_ws_strm, (sjmi::UnescapeResult*)d_result, st, nullptr, nullptr, alt)) ||
        fail(c, "doc ordinals",
             sjmi::strings_doc_ordinals_launch(buf0, par0, alt, total_len, (const unsigned long long*)d_doc_offsets, n_docs, c->d_blk_ord,
                                               c->d_soff, (const sjmi::UnescapeResult*)d_result, c->d_doc_ord,
                                               (unsigned long long*)d_doc_str_offsets, st, nullptr)))
        return SJMI_ERR_HIP;
    if (!c->d_ures_walk && fail(c, "hipMalloc(ures_walk)", hipMalloc((void**)&c->d_ures_walk, sizeof(sjmi_unescape_result)))) return SJMI_ERR_HIP;
    if (fail(c, "D2D(ures)", hipMemcpyAsync(c->d_ures_walk, d_result, sizeof(sjmi_unescape_result), hipMemcpyDeviceToDevice, st))) return SJMI_ERR_HIP;
    c->soff_idx = d_indexes;
    return SJMI_OK;
}

static int unescape_device_impl(sjmi_ctx* c, const void* d_buf, uint64_t len, const void* d_indexes, uint64_t count,
                                void* d_string_buffer, uint64_t string_capacity, void* d_result, void* stream,
                                const sjmi::UnescapeBatch& batch) {
    if (!c || !d_buf || !d_indexes || !d_string_buffer || !d_result || len >= (1ull << 32)) return SJMI_ERR_ARG;
    if (fail(c, "hipSetDevice", hipSetDevice(c->device))) return SJMI_ERR_HIP;
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    if (!batch.d_doc_offsets) {
        if (((uintptr_t)d_buf & 15)) return SJMI_ERR_ARG;
        return strings_device_impl(c, d_buf, len, d_string_buffer, string_capacity, nullptr, 0, nullptr, d_result, st);
    }
    // a batch: over the batch itself if this context's last stage-1 launch was the plain pass over it, else (documents indexed
    // one by one) over the sanitized copy
    // (indexed by sjmi_stage1_batch_isolated*: whether the plain pass was accepted is a flag on the device)
    const bool flagged = c->accept_valid && c->accept_buf == d_buf && c->accept_len == len;
    const bool plain = !flagged && c->par_valid && c->par_buf == d_buf && c->par_len == len;
    return strings_batch_impl(c, d_buf, len, d_indexes, count, batch.d_doc_offsets, batch.d_index_offsets, batch.n_docs, plain,
                              flagged ? c->d_batch_flags + 1 : nullptr, d_string_buffer, string_capacity, batch.d_doc_str_offsets,
                              d_result, st);
}

int sjmi_unescape_device(sjmi_ctx* c, const void* d_buf, uint64_t len, const void* d_indexes, uint64_t count,
                         void* d_string_buffer, uint64_t string_capacity, void* d_result, void* stream) {
    return unescape_device_impl(c, d_buf, len, d_indexes, count, d_string_buffer, string_capacity, d_result, stream,
                                sjmi::UnescapeBatch());
}

int sjmi_unescape_batch_device(sjmi_ctx* c, const void* d_buf, uint64_t total_len, const void* d_indexes, uint64_t count,
                               const void* d_doc_offsets, const void* d_index_offsets, uint64_t n_docs,
                               void* d_string_buffer, uint64_t string_capacity, void* d_doc_string_offsets, void* d_result,
                               void* stream) {
    if (!d_doc_offsets || !d_index_offsets) return SJMI_ERR_ARG;
    sjmi::UnescapeBatch batch;
    batch.d_doc_offsets = (const unsigned long long*)d_doc_offsets;
    batch.d_index_offsets = (const unsigned long long*)d_index_offsets;
    batch.n_docs = n_docs;
    batch.d_doc_str_offsets = (unsigned long long*)d_doc_string_offsets;  // optional
    return unescape_device_impl(c, d_buf, total_len, d_indexes, count, d_string_buffer, string_capacity, d_result, stream, batch);
}

// host forms: the document / batch of the last stage-1 call on this context
static int unescape_host(sjmi_ctx* c, uint8_t* string_buffer, uint64_t string_capacity, uint64_t* doc_string_offsets,
                         uint64_t* total_bytes, uint64_t* first_error_index, uint32_t* first_error_code) {
    if (fail(c, "hipSetDevice", hipSetDevice(c->device))) return SJMI_ERR_HIP;
    const uint64_t n_docs = c->last_ndocs;
    const size_t need_sb = (size_t)c->last_len + 4 * (size_t)c->last_count + 64;  // sum(4+len_k) <= len + 4*#strings
    const size_t ob = (n_docs + 1) * sizeof(unsigned long long);
    if (!grow(c, (void**)&c->d_sb, &c->sb_bytes, need_sb, "hipMalloc(sb)")) return SJMI_ERR_HIP;
    if (doc_string_offsets && !grow(c, (void**)&c->d_docstr, &c->docstr_bytes, ob + 64, "hipMalloc(docstr)")) return SJMI_ERR_HIP;
    if (!c->d_ures && fail(c, "hipMalloc(ures)", hipMalloc((void**)&c->d_ures, sizeof(sjmi_unescape_result))))
        return SJMI_ERR_HIP;
    sjmi::UnescapeBatch batch;
    if (c->last_batch) {  // (also for a plain sjmi_unescape after a batch call: its strings end inside their documents)
        batch.d_doc_offsets = c->d_docoff;
        batch.d_index_offsets = c->d_docoff + (n_docs + 1);  // written by the batch call
        batch.n_docs = n_docs;
        batch.d_doc_str_offsets = doc_string_offsets ? c->d_docstr : nullptr;
    }
    const uint32_t* const idx_dev = c->idx_last ? c->idx_last : c->d_idx;
    const int rc = unescape_device_impl(c, c->d_in, c->last_len, idx_dev, c->last_count, c->d_sb, c->sb_bytes, c->d_ures,
                                        c->stream, batch);
    if (rc != SJMI_OK) return rc;
    sjmi_unescape_result r;
    unsigned long long err_index = ~0ull;
    const bool streamed = true;  // (the string pass reports the error's byte position: turn it into the string's index)
    if (streamed) {
        if (!c->d_err_index && fail(c, "hipMalloc(err_index)", hipMalloc((void**)&c->d_err_index, sizeof(unsigned long long))))
            return SJMI_ERR_HIP;
        if (fail(c, "error index", sjmi::strings_error_index_launch(idx_dev, c->last_count, nullptr, (const sjmi::UnescapeResult*)c->d_ures,
                                                                    c->d_err_index, c->stream)) ||
            fail(c, "D2H(err_index)", hipMemcpyAsync(&err_index, c->d_err_index, sizeof err_index, hipMemcpyDeviceToHost, c->stream)))
            return SJMI_ERR_HIP;
    }
    if (fail(c, "D2H(ures)", hipMemcpyAsync(&r, c->d_ures, sizeof r, hipMemcpyDeviceToHost, c->stream)) ||
        (doc_string_offsets &&
         fail(c, "D2H(docstr)", hipMemcpyAsync(doc_string_offsets, c->d_docstr, ob, hipMemcpyDeviceToHost, c->stream))) ||
        fail(c, "sync", hipStreamSynchronize(c->stream)))
        return SJMI_ERR_HIP;
    *total_bytes = r.total_bytes;
    if (r.first_error_inv) {
        const uint64_t v = ~r.first_error_inv;
        *first_error_index = streamed ? err_index : v >> 8;
        *first_error_code = (uint32_t)(v & 0xFF);
    } else {
        *first_error_index = ~0ull;
        *first_error_code = 0;
    }
    if (r.flags & 0xCu) {
        c->err = "string pass: engine fault";
        return SJMI_ERR_INTERNAL;
    }
    if (r.total_bytes > string_capacity || (r.flags & 1u)) {
        c->err = "string_capacity too small";
        return SJMI_ERR_CAPACITY;
    }
    if (r.total_bytes &&
        (fail(c, "D2H(sb)", hipMemcpyAsync(string_buffer, c->d_sb, r.total_bytes, hipMemcpyDeviceToHost, c->stream)) ||
         fail(c, "sync", hipStreamSynchronize(c->stream))))
        return SJMI_ERR_HIP;
    return SJMI_OK;
}

int sjmi_unescape(sjmi_ctx* c, uint8_t* string_buffer, uint64_t string_capacity, uint64_t* total_bytes,
                  uint64_t* first_error_index, uint32_t* first_error_code) {
    if (!c || !string_buffer || !total_bytes || !first_error_index || !first_error_code) return SJMI_ERR_ARG;
    drop_stale_views(c);  // (the last call's indexes may sit in a caller array that has been unregistered since)
    if (!c->last_valid) {
        c->err = "sjmi_unescape needs a preceding successful sjmi_stage1 on this context";
        return SJMI_ERR_ARG;
    }
    return unescape_host(c, string_buffer, string_capacity, nullptr, total_bytes, first_error_index, first_error_code);
}

int sjmi_unescape_batch(sjmi_ctx* c, uint8_t* string_buffer, uint64_t string_capacity, uint64_t* doc_string_offsets,
                        uint64_t* total_bytes, uint64_t* first_error_index, uint32_t* first_error_code) {
    if (!c || !string_buffer || !doc_string_offsets || !total_bytes || !first_error_index || !first_error_code)
        return SJMI_ERR_ARG;
    drop_stale_views(c);
    if (!c->last_valid || !c->last_batch) {
        c->err = "sjmi_unescape_batch needs a preceding successful sjmi_stage1_batch[_isolated] on this context";
        return SJMI_ERR_ARG;
    }
    return unescape_host(c, string_buffer, string_capacity, doc_string_offsets, total_bytes, first_error_index, first_error_code);
}

int sjmi_stage1_unescape(sjmi_ctx* c, const uint8_t* buf, uint64_t len, uint32_t* indexes, uint64_t index_capacity,
                         uint64_t* count, uint32_t* status, uint8_t* string_buffer, uint64_t string_capacity,
                         uint64_t* total_bytes, uint64_t* first_error_index, uint32_t* first_error_code) {
    if (!c || (!buf && len) || !indexes || !count || !status || !string_buffer || !total_bytes || !first_error_index ||
        !first_error_code)
        return SJMI_ERR_ARG;
    if (len > c->capacity || len >= (1ull << 32)) {
        c->err = "document larger than the context capacity";
        return SJMI_ERR_CAPACITY;
    }
    if (index_capacity < 1) return SJMI_ERR_CAPACITY;
    if (fail(c, "hipSetDevice", hipSetDevice(c->device))) return SJMI_ERR_HIP;
    // everything is queued before the first synchronisation: the unescape kernels take the structural count from the
    // stage-1 result on the device; grids and workspace are sized for the bound "one structural per byte"
    const uint64_t bound = len + 1;
    const size_t need_sb = (size_t)len + 4 * ((size_t)len / 2 + 2) + 64;  // sum(4+len_k): every string takes >= 2 source bytes
    (void)bound;
    if (!grow(c, (void**)&c->d_sb, &c->sb_bytes, need_sb, "hipMalloc(sb)")) return SJMI_ERR_HIP;
    if (!c->d_ures && fail(c, "hipMalloc(ures)", hipMalloc((void**)&c->d_ures, sizeof(sjmi_unescape_result))))
        return SJMI_ERR_HIP;
    if (!upload_document(c, buf, len)) return SJMI_ERR_HIP;
    const int steps = c->forced_steps ? c->forced_steps : sjmi::stage1_pick_steps(len);
    const uint64_t dev_cap = c->capacity + 2 < index_capacity ? c->capacity + 2 : index_capacity;
    const sjmi::Stage1Result* d_res1 = (const sjmi::Stage1Result*)((uint8_t*)c->d_ws + sjmi::WS_RESULT_OFFSET);
    // Latency path of the drop-in call: besides the two kernels only what cannot be avoided is queued -- stage 1 through the
    // device entry point (double-buffered workspace: no memset, the scanner writes the record), the string pass's record
    // inside its own workspace (zeroed with it), one 48-byte D2H of {error index, both records} instead of three copies.
    sjmi_unescape_result r;
    unsigned long long err_index = ~0ull;
    struct Pack {
        unsigned long long err_index;
        sjmi_stage1_result s1;
        sjmi_unescape_result u;
    };
    static_assert(sizeof(Pack) == 48, "ParsePack layout");
    if (sjmi::strings_parse_pack_bytes() != sizeof(Pack)) return SJMI_ERR_INTERNAL;
    if (!c->d_pack && fail(c, "hipMalloc(pack)", hipMalloc(&c->d_pack, 64))) return SJMI_ERR_HIP;
    if (!c->h_pack && fail(c, "hipHostMalloc(pack)", hipHostMalloc(&c->h_pack, 64))) return SJMI_ERR_HIP;
    if (!c->d_res_tmp && fail(c, "hipMalloc(result)", hipMalloc((void**)&c->d_res_tmp, 64))) return SJMI_ERR_HIP;
    const uint64_t spec_idx = len + 2 < dev_cap ? len + 2 : dev_cap;
    const uint64_t sb_bound = len + 4 * (len / 2 + 2);
    const uint64_t spec_sb = string_buffer ? (sb_bound < string_capacity ? sb_bound : string_capacity) : 0;
    const bool small = len <= (4u << 10);
    const AutoSafeOff own_retry(c);  // (the retry below is this call's own)
    // ZERO-COPY (both output arrays device-visible: a parser with page-locked buffers): stage 1 writes the indexes, the string
    // pass the records and the last small kernel the packed result records straight into the caller's / the context's pinned
    // memory -- no download, ONE host synchronisation (twitter.json: two copies of 221 + 440 KB and a round trip less)
    uint32_t* const zc_idx = ((uintptr_t)indexes & 15) ? nullptr : (uint32_t*)device_view(c, indexes);
    uint8_t* const zc_sb = zc_idx && !((uintptr_t)string_buffer & 15) ? (uint8_t*)device_view(c, string_buffer) : nullptr;
    if (!c->h_pack_dev) c->h_pack_dev = device_view(c, c->h_pack);
    const bool zero_copy = zc_idx && zc_sb && c->h_pack_dev;
    uint32_t* const idx_out = zero_copy ? zc_idx : c->d_idx;
    c->idx_last = idx_out;
    c->idx_last_host = zero_copy ? indexes : nullptr;
    for (int attempt = 0; attempt < 2; ++attempt) {
        const int s1rc = stage1_device_impl(c, c->d_in, len, idx_out, dev_cap, c->d_res_tmp, c->stream, 0);
        if (s1rc != SJMI_OK) return s1rc;
        if (!grow(c, &c->d_ws_strm, &c->ws_strm_bytes, sjmi::strings_workspace_bytes(len), "hipMalloc(ws_strm)")) return SJMI_ERR_HIP;
        const unsigned long long* par = parity_for(c, c->d_in, len, c->stream);
        if (!par) return SJMI_ERR_HIP;
        sjmi::UnescapeResult* d_u = sjmi::strings_workspace_result(c->d_ws_strm);
        if (fail(c, "strings launch", sjmi::strings_launch(c->d_in, len, par, zero_copy ? zc_sb : c->d_sb, zero_copy ? string_capacity : c->sb_bytes,
                                                           nullptr, 0, nullptr, c->d_ws_strm, d_u, c->stream)) ||
            fail(c, "error index", sjmi::strings_error_index_pack_launch(idx_out, (const sjmi::Stage1Result*)c->d_res_tmp, d_u,
                                                                         zero_copy ? c->h_pack_dev : c->d_pack, c->stream)) ||
            (!zero_copy && fail(c, "D2H(results)", hipMemcpyAsync(c->h_pack, c->d_pack, sizeof(Pack), hipMemcpyDeviceToHost, c->stream))))
            return SJMI_ERR_HIP;
        // A SMALL document's outputs are downloaded speculatively, by their bounds (one structural per byte; 3 record bytes per
        // 2 source bytes), behind the same synchronisation: the second host round trip costs more than the few KB it would save
        if (!zero_copy && small &&
            (fail(c, "D2H(indexes)", hipMemcpyAsync(indexes, c->d_idx, spec_idx * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream)) ||
             (spec_sb && fail(c, "D2H(sb)", hipMemcpyAsync(string_buffer, c->d_sb, spec_sb, hipMemcpyDeviceToHost, c->stream)))))
            return SJMI_ERR_HIP;
        if (fail(c, "sync", hipStreamSynchronize(c->stream))) return SJMI_ERR_HIP;
        const Pack* hp = static_cast<const Pack*>(c->h_pack);
        *c->h_res = hp->s1;
        r = hp->u;
        err_index = hp->err_index;
        if (!(c->h_res->status & SJMI_ST_INTERNAL) || c->ticket_mode) break;
        c->ticket_mode = true;  // fast-mode liveness assumption failed: latch the safe mode and run again
    }
    (void)d_res1;
    *status = c->h_res->status & 0xFFu;
    *count = c->h_res->count;
    *total_bytes = 0;
    *first_error_index = ~0ull;
    *first_error_code = 0;
    if (c->h_res->status & SJMI_ST_INTERNAL) {
        c->err = "look-back timeout";
        return SJMI_ERR_INTERNAL;
    }
    if (c->h_res->status & SJMI_ST_CAPACITY) {
        c->err = "index_capacity too small";
        return SJMI_ERR_CAPACITY;
    }
    c->last_len = len;
    c->last_count = c->h_res->count;
    c->last_valid = true;
    c->last_batch = false;
    const bool strings_ok = *status == 0;  // (a document that fails stage 1 has no meaningful strings: the caller throws)
    if (strings_ok) {
        *total_bytes = r.total_bytes;
        if (r.first_error_inv) {
            const uint64_t v = ~r.first_error_inv;
            *first_error_index = err_index;
            *first_error_code = (uint32_t)(v & 0xFF);
        }
        if (r.flags & 0xCu) {
            c->err = "string pass: engine fault";
            return SJMI_ERR_INTERNAL;
        }
        if (r.total_bytes > string_capacity || (r.flags & 1u)) {
            c->err = "string_capacity too small";
            return SJMI_ERR_CAPACITY;
        }
    }
    if (zero_copy) return SJMI_OK;  // (indexes and records are in the caller's arrays already)
    if (small && c->h_res->count + 1 <= spec_idx && (!strings_ok || r.total_bytes <= spec_sb)) return SJMI_OK;  // (already here)
    if (fail(c, "D2H(indexes)",
             hipMemcpyAsync(indexes, c->d_idx, (c->h_res->count + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream)) ||
        (strings_ok && r.total_bytes &&
         fail(c, "D2H(sb)", hipMemcpyAsync(string_buffer, c->d_sb, r.total_bytes, hipMemcpyDeviceToHost, c->stream))) ||
        fail(c, "sync", hipStreamSynchronize(c->stream)))
        return SJMI_ERR_HIP;
    return SJMI_OK;
}

int sjmi_walk_batch_device(sjmi_ctx* c, const void* d_buf, const void* d_doc_offsets, uint64_t n_docs, const void* d_indexes,
                           uint64_t count, const void* d_index_offsets, const void* d_doc_status,
                           const void* d_string_buffer, const void* d_doc_string_offsets, uint64_t string_base,
                           int max_depth, void* d_tape, uint64_t tape_capacity, void* d_tape_offsets, void* d_doc_errors,
                           void* d_result, void* stream) {
    if (!c || !d_buf || !d_doc_offsets || !d_indexes || !d_index_offsets || !d_doc_status || !d_string_buffer ||
        !d_doc_string_offsets || !d_tape || !d_tape_offsets || !d_doc_errors || !d_result || max_depth < 1)
        return SJMI_ERR_ARG;
    if (fail(c, "hipSetDevice", hipSetDevice(c->device))) return SJMI_ERR_HIP;
    if (!grow(c, &c->d_ws_walk, &c->ws_walk_bytes, sjmi::walk_workspace_bytes(count, n_docs), "hipMalloc(ws_walk)"))
        return SJMI_ERR_HIP;
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    // The cooperative walker takes the offsets of the string records from the record table the string pass of THESE indexes
    // left on this context (sjmi_unescape_batch_device), and the ordinal of every document's first string with it.
    if (!c->soff_idx || c->soff_idx != d_indexes) {
        c->err = "sjmi_walk_batch_device needs the sjmi_unescape_batch_device call of the same indexes on this context";
        return SJMI_ERR_ARG;
    }
    if (fail(c, "walk launch",
             sjmi::walk_launch((const uint8_t*)d_buf, (const unsigned long long*)d_doc_offsets, n_docs, (const uint32_t*)d_indexes,
                               count, (const unsigned long long*)d_index_offsets, (const uint32_t*)d_doc_status,
                               (const uint8_t*)d_string_buffer, c->d_doc_ord, string_base,
                               max_depth, (unsigned long long*)d_tape, tape_capacity, (unsigned long long*)d_tape_offsets,
                               (int32_t*)d_doc_errors, c->d_ws_walk, (sjmi::WalkResult*)d_result, st, nullptr,
                               (const sjmi::UnescapeResult*)c->d_ures_walk, c->d_soff)))
        return SJMI_ERR_HIP;
    return SJMI_OK;
}

static int stage1_device_impl(sjmi_ctx* c, const void* d_buf, uint64_t len, void* d_indexes, uint64_t index_capacity,
                              void* d_result, void* stream, uint32_t shard_flags);

int sjmi_stage1_device(sjmi_ctx* c, const void* d_buf, uint64_t len, void* d_indexes, uint64_t index_capacity,
                       void* d_result, void* stream) {
    return stage1_device_impl(c, d_buf, len, d_indexes, index_capacity, d_result, stream, 0);
}

int sjmi_stage1_shard_device2(sjmi_ctx* c, const void* d_buf, uint64_t len, uint64_t halo_bytes, int halo_from_document_start,
                              int is_last, int entry_parity, void* d_indexes, uint64_t index_capacity, void* d_result, void* stream);
int sjmi_stage1_shard_device(sjmi_ctx* c, const void* d_buf, uint64_t len, uint64_t halo_bytes, int is_last, int entry_parity,
                             void* d_indexes, uint64_t index_capacity, void* d_result, void* stream) {
    return sjmi_stage1_shard_device2(c, d_buf, len, halo_bytes, 0, is_last, entry_parity, d_indexes, index_capacity, d_result, stream);
}

int sjmi_stage1_shard_device2(sjmi_ctx* c, const void* d_buf, uint64_t len, uint64_t halo_bytes, int halo_from_document_start,
                              int is_last, int entry_parity, void* d_indexes, uint64_t index_capacity, void* d_result, void* stream) {
    // a shard that is not the last one ends on a block boundary (its successor owns what straddles it); the halo is
    // whole blocks so that the shard itself stays 16-byte aligned
    if ((halo_bytes & 63) || halo_bytes > 65535ull * 64 || (!is_last && (len & 63)) || (!is_last && len == 0)) return SJMI_ERR_ARG;
    const uint32_t flags = ((uint32_t)(halo_bytes / 64) << 16) | (is_last ? 0u : sjmi::FLAG_NO_TAIL) |
                           (entry_parity ? sjmi::FLAG_ENTRY_PARITY : 0u) | (halo_from_document_start ? sjmi::FLAG_HALO_FROM_START : 0u);
    return stage1_device_impl(c, d_buf, len, d_indexes, index_capacity, d_result, stream, flags);
}

static int stage1_device_impl(sjmi_ctx* c, const void* d_buf, uint64_t len, void* d_indexes, uint64_t index_capacity,
                              void* d_result, void* stream, uint32_t shard_flags) {
    if (!c || !d_buf || !d_indexes || !d_result) return SJMI_ERR_ARG;
    if (len >= (1ull << 32) || ((uintptr_t)d_buf & 15) || ((uintptr_t)d_indexes & 15)) return SJMI_ERR_ARG;
    if (fail(c, "hipSetDevice", hipSetDevice(c->device))) return SJMI_ERR_HIP;
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    const int steps = c->forced_steps ? c->forced_steps : sjmi::stage1_pick_steps(len);
    const size_t need = (sjmi::stage1_workspace_bytes(len, steps) + 255) & ~(size_t)255;
    if (2 * need > c->ws_dev_bytes) {  // grown outside any timed loop on first use of a given size
        if (c->d_ws_dev) (void)hipFree(c->d_ws_dev);
        c->d_ws_dev = nullptr;
        c->ws_dev_bytes = 0;
        if (fail(c, "hipMalloc(ws_dev)", hipMalloc(&c->d_ws_dev, 2 * need))) return SJMI_ERR_HIP;
        c->ws_dev_bytes = 2 * need;
        c->ws_dev_clean[0] = c->ws_dev_clean[1] = 0;
    }
    // Nothing but the kernel is queued once the context is warm: this launch finds its half of the workspace zeroed
    // by the previous one, zeroes the other half for the next one, and its last wave writes *d_result.
    // (One launch stream per context at a time: the halves are handed over in stream order.)
    const size_t half = c->ws_dev_bytes / 2;
    const int h = c->ws_dev_next;
    uint8_t* ws = (uint8_t*)c->d_ws_dev + (size_t)h * half;
    const bool fast = !(launch_flags(c) & (sjmi::FLAG_SAFE | sjmi::DBG_NO_LOOKBACK));  // (the scanner writes the result)
    sjmi::Stage1Extras ex;
    ex.workspace_is_zero = c->ws_dev_clean[h] >= need;
    ex.zero_next = (uint8_t*)c->d_ws_dev + (size_t)(1 - h) * half;
    ex.zero_bytes = need;
    ex.result_out = fast ? d_result : nullptr;
    ex.blkpar = shard_flags ? nullptr : parity_out(c, d_buf, len);
    if (shard_flags) c->par_valid = false;
    ex.zero2 = c->s1_zero2;
    ex.zero2_bytes = c->s1_zero2_bytes;
    if (fast) ex.single = c->s1_single;
    c->s1_single_done = fast && c->s1_single.index_offsets != nullptr;
    if (c->batch_side && !shard_flags) {  // (the fused batch pipeline's plain pass: per-block index positions and tape words)
        const size_t entries = sjmi::stage1_block_entries(len);
        if (!grow(c, (void**)&c->d_blkidx, &c->blkidx_bytes, entries * sizeof(uint32_t), "hipMalloc(blkidx)") ||
            !grow(c, (void**)&c->d_blkw, &c->blkw_bytes, entries * sizeof(uint16_t), "hipMalloc(blkw)"))
            return SJMI_ERR_HIP;
        ex.blkidx = c->d_blkidx;
        ex.blkw = c->d_blkw;
    }
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (c->profiling) {
        if (c->events_used == c->events.size()) {
            hipEvent_t a, b;
            if (fail(c, "hipEventCreate", hipEventCreate(&a)) || fail(c, "hipEventCreate", hipEventCreate(&b)))
                return SJMI_ERR_HIP;
            c->events.emplace_back(a, b);
        }
        ev0 = c->events[c->events_used].first;
        ev1 = c->events[c->events_used].second;
        ++c->events_used;
    }
    if (fail(c, "launch", sjmi::stage1_launch((const uint8_t*)d_buf, len, (uint32_t*)d_indexes, index_capacity, ws, steps,
                                              st, ev0, ev1, launch_flags(c) | shard_flags, ex)))
        return SJMI_ERR_HIP;
    note_launch(c, st);
    if (!fast && fail(c, "D2D(result)", hipMemcpyAsync(d_result, ws + sjmi::WS_RESULT_OFFSET, sizeof(sjmi_stage1_result),
                                                       hipMemcpyDeviceToDevice, st)))
        return SJMI_ERR_HIP;
    c->ws_dev_clean[h] = 0;
    c->ws_dev_clean[1 - h] = need;
    c->ws_dev_next = 1 - h;
    c->ws_dev_last = ws;
    if (c->auto_safe && !c->ticket_mode) {
        // opt-in (sjmi_set_auto_safe): the FAST kernel's liveness rests on its whole grid being resident; a caller that
        // shares the GPU with other kernels can ask for the check here -- one synchronisation per launch -- instead of
        // finding SJMI_ST_INTERNAL in its result record: on a tripped spin bound SAFE mode is latched and the launch repeated
        sjmi_stage1_result r;
        if (fail(c, "D2H(result)", hipMemcpyAsync(&r, d_result, sizeof r, hipMemcpyDeviceToHost, st)) ||
            fail(c, "sync", hipStreamSynchronize(st)))
            return SJMI_ERR_HIP;
        if (r.status & SJMI_ST_INTERNAL) {
            c->ticket_mode = true;
            return stage1_device_impl(c, d_buf, len, d_indexes, index_capacity, d_result, stream, shard_flags);
        }
    }
    return SJMI_OK;
}

int sjmi_stage1_masks_device(sjmi_ctx* c, const void* d_buf, uint64_t len, void* d_masks, uint64_t mask_capacity_blocks,
                             void* stream) {
    if (!c || !d_buf || !d_masks || len >= (1ull << 32) || ((uintptr_t)d_buf & 15) || ((uintptr_t)d_masks & 7)) return SJMI_ERR_ARG;
    if (mask_capacity_blocks < len / 64 + 1) return SJMI_ERR_CAPACITY;
    if (fail(c, "hipSetDevice", hipSetDevice(c->device))) return SJMI_ERR_HIP;
    if (!grow(c, &c->d_ws_masks, &c->ws_masks_bytes, sjmi::masks_workspace_bytes(len), "hipMalloc(ws_masks)")) return SJMI_ERR_HIP;
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    if (fail(c, "masks launch", sjmi::masks_launch((const uint8_t*)d_buf, len, (unsigned long long*)d_masks, c->d_ws_masks, st)))
        return SJMI_ERR_HIP;
    return SJMI_OK;
}

int sjmi_stage1_masks(sjmi_ctx* c, const uint8_t* buf, uint64_t len, uint64_t* masks, uint64_t mask_capacity_blocks,
                      uint64_t* n_blocks) {
    if (!c || (!buf && len) || !masks || !n_blocks) return SJMI_ERR_ARG;
    if (len > c->capacity || len >= (1ull << 32)) {
        c->err = "document larger than the context capacity";
        return SJMI_ERR_CAPACITY;
    }
    const uint64_t nblocks = len / 64 + 1;
    if (mask_capacity_blocks < nblocks) return SJMI_ERR_CAPACITY;
    if (fail(c, "hipSetDevice", hipSetDevice(c->device))) return SJMI_ERR_HIP;
    if (!grow(c, &c->d_masks, &c->masks_bytes, (size_t)nblocks * 48, "hipMalloc(masks)")) return SJMI_ERR_HIP;
    if (len && fail(c, "H2D", hipMemcpyAsync(c->d_in, buf, len, hipMemcpyHostToDevice, c->stream))) return SJMI_ERR_HIP;
    c->last_valid = false;  // (the context's document buffer now holds this document, without indexes)
    c->par_valid = false;   // (... and without block parities: new bytes under the same pointer)
    c->accept_valid = false;
    const int rc = sjmi_stage1_masks_device(c, c->d_in, len, c->d_masks, nblocks, c->stream);
    if (rc != SJMI_OK) return rc;
    if (fail(c, "D2H(masks)", hipMemcpyAsync(masks, c->d_masks, (size_t)nblocks * 48, hipMemcpyDeviceToHost, c->stream)) ||
        fail(c, "sync", hipStreamSynchronize(c->stream)))
        return SJMI_ERR_HIP;
    *n_blocks = nblocks;
    return SJMI_OK;
}

int sjmi_stage1_batch_device(sjmi_ctx* c, const void* d_buf, uint64_t total_len, const void* d_doc_offsets,
                             uint64_t n_docs, void* d_indexes, uint64_t index_capacity, void* d_index_offsets,
                             void* d_result, void* stream) {
    if (!c || !d_doc_offsets || !d_index_offsets) return SJMI_ERR_ARG;
    int rc = sjmi_stage1_device(c, d_buf, total_len, d_indexes, index_capacity, d_result, stream);
    if (rc != SJMI_OK) return rc;
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    if (fail(c, "split launch",
             sjmi::split_docs_launch((const uint32_t*)d_indexes, (const sjmi::Stage1Result*)d_result,
                                     (const unsigned long long*)d_doc_offsets, n_docs,
                                     (unsigned long long*)d_index_offsets, st)))
        return SJMI_ERR_HIP;
    return SJMI_OK;
}

int sjmi_stage1_batch(sjmi_ctx* c, const uint8_t* buf, uint64_t total_len, const uint64_t* doc_offsets, uint64_t n_docs,
                      uint32_t* indexes, uint64_t index_capacity, uint64_t* index_offsets, uint64_t* count,
                      uint32_t* status) {
    if (!c || (!buf && total_len) || !doc_offsets || !indexes || !index_offsets || !count || !status) return SJMI_ERR_ARG;
    if (total_len > c->capacity || total_len >= (1ull << 32)) {
        c->err = "batch larger than the context capacity";
        return SJMI_ERR_CAPACITY;
    }
    if (bad_offsets(c, doc_offsets, n_docs, total_len, false)) return SJMI_ERR_ARG;
    if (fail(c, "hipSetDevice", hipSetDevice(c->device))) return SJMI_ERR_HIP;
    const size_t ob = (n_docs + 1) * sizeof(unsigned long long);
    if (!grow(c, (void**)&c->d_docoff, &c->docoff_bytes, 2 * ob + 64, "hipMalloc(docoff)")) return SJMI_ERR_HIP;
    unsigned long long* d_io = c->d_docoff + (n_docs + 1);
    void* d_res = (uint8_t*)c->d_ws + sjmi::WS_RESULT_OFFSET;
    if ((total_len && fail(c, "H2D", hipMemcpyAsync(c->d_in, buf, total_len, hipMemcpyHostToDevice, c->stream))) ||
        fail(c, "H2D(offsets)", hipMemcpyAsync(c->d_docoff, doc_offsets, ob, hipMemcpyHostToDevice, c->stream)))
        return SJMI_ERR_HIP;
    const int steps = c->forced_steps ? c->forced_steps : sjmi::stage1_pick_steps(total_len);
    const uint64_t dev_cap = c->capacity + 2 < index_capacity ? c->capacity + 2 : index_capacity;
    sjmi::Stage1Extras ex1;
    ex1.blkpar = parity_out(c, c->d_in, total_len);
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (fail(c, "launch", sjmi::stage1_launch(c->d_in, total_len, c->d_idx, dev_cap, c->d_ws, steps, c->stream, nullptr,
                                                  nullptr, launch_flags(c), ex1)) ||
            fail(c, "split", sjmi::split_docs_launch(c->d_idx, (const sjmi::Stage1Result*)d_res, c->d_docoff, n_docs, d_io,
                                                     c->stream)) ||
            fail(c, "D2H(result)", hipMemcpyAsync(c->h_res, d_res, sizeof(sjmi_stage1_result), hipMemcpyDeviceToHost, c->stream)) ||
            fail(c, "D2H(io)", hipMemcpyAsync(index_offsets, d_io, ob, hipMemcpyDeviceToHost, c->stream)) ||
            fail(c, "sync", hipStreamSynchronize(c->stream)))
            return SJMI_ERR_HIP;
        if (!(c->h_res->status & SJMI_ST_INTERNAL) || c->ticket_mode) break;
        c->ticket_mode = true;  // fast-mode liveness assumption failed: latch the safe mode and run again
    }
    *status = c->h_res->status & 0xFFu;
    *count = c->h_res->count;
    if (c->h_res->status & SJMI_ST_INTERNAL) return SJMI_ERR_INTERNAL;
    if (c->h_res->status & SJMI_ST_CAPACITY) return SJMI_ERR_CAPACITY;
    if (fail(c, "D2H(indexes)", hipMemcpyAsync(indexes, c->d_idx, (c->h_res->count + 1) * sizeof(uint32_t),
                                               hipMemcpyDeviceToHost, c->stream)) ||
        fail(c, "sync", hipStreamSynchronize(c->stream)))
        return SJMI_ERR_HIP;
    c->idx_last = c->d_idx;  // (a batch call's indexes are always in the context's own array, never in a caller's zero-copy view)
    c->idx_last_host = nullptr;
    c->last_len = total_len;
    c->last_count = c->h_res->count;
    c->last_valid = true;
    c->last_ndocs = n_docs;
    c->last_batch = true;
    return SJMI_OK;
}

static int stage1_batch_isolated_device_impl(sjmi_ctx* c, const void* d_buf, uint64_t total_len, const void* d_doc_offsets,
                                             uint64_t n_docs, void* d_indexes, uint64_t index_capacity, void* d_index_offsets,
                                             void* d_doc_status, void* d_result, void* stream, const uint32_t* d_skip);
static int stage1_batch_isolated_device_impl(sjmi_ctx* c, const void* d_buf, uint64_t total_len, const void* d_doc_offsets,
                                             uint64_t n_docs, void* d_indexes, uint64_t index_capacity, void* d_index_offsets,
                                             void* d_doc_status, void* d_result, void* stream, const uint32_t* d_skip);
// Stage 1 of a batch with per-document verdicts.  Optimistic: ONE plain k_stage1 launch over the packed batch, accepted on the
// device when every document ends in a control-character separator, the documents cover the buffer exactly and the global
// verdict is clean (batch.hip: then it is exactly what the per-document passes give); the per-document passes are queued behind
// it and leave at once if it was accepted.  *d_skip_out = the device flag (!= 0: accepted) or nullptr (not tried); the string
// pass of the same batch on this context reads it (c->accept_*).  (SJMI_BATCH_OPTIMISTIC=0 switches the plain pass off.)
// (sjmi_parse_batch_device has its own ordering of the same idea: parse_batch_pipeline below.)
static int stage1_batch_optimistic(sjmi_ctx* c, const void* d_buf, uint64_t total_len, const void* d_doc_offsets, uint64_t n_docs,
                                   void* d_indexes, uint64_t index_capacity, void* d_index_offsets, void* d_doc_status,
                                   void* d_result, void* stream, const uint32_t** d_skip_out) {
    if (!c || !d_buf || !d_doc_offsets || !d_indexes || !d_index_offsets || !d_doc_status || !d_result) return SJMI_ERR_ARG;
    if (total_len >= (1ull << 32)) return SJMI_ERR_ARG;
    static const bool optimistic = !(getenv("SJMI_BATCH_OPTIMISTIC") && atoi(getenv("SJMI_BATCH_OPTIMISTIC")) == 0);
    hipStream_t st0 = stream ? (hipStream_t)stream : c->stream;
    const uint32_t* d_skip = nullptr;
    c->accept_valid = false;
    if (optimistic && n_docs && total_len && index_capacity >= 1 && !((uintptr_t)d_buf & 15) && !((uintptr_t)d_indexes & 15)) {
        if (fail(c, "hipSetDevice", hipSetDevice(c->device))) return SJMI_ERR_HIP;
        if (!c->d_batch_flags && fail(c, "hipMalloc(batch flags)", hipMalloc((void**)&c->d_batch_flags, 64))) return SJMI_ERR_HIP;
        if (fail(c, "separator check", sjmi::batch_plain_check_launch((const uint8_t*)d_buf, (const unsigned long long*)d_doc_offsets,
                                                                      n_docs, total_len, c->d_batch_flags, st0)))
            return SJMI_ERR_HIP;
        int rc0;
        {
            const AutoSafeOff plain_only(c);  // (a tripped liveness bound only rejects the plain pass: the per-document passes take over)
            rc0 = stage1_device_impl(c, d_buf, total_len, d_indexes, index_capacity, d_result, stream, 0);
        }
        if (rc0 != SJMI_OK) return rc0;
        // (a FAST launch leaves the scanner's per-granule prefixes in its half of the workspace: the split starts from them)
        sjmi::Stage1Prefixes hint;
        if (!(launch_flags(c) & (sjmi::FLAG_SAFE | sjmi::DBG_NO_LOOKBACK)) && c->ws_dev_last)
            hint = sjmi::stage1_prefixes(c->ws_dev_last, total_len, c->forced_steps ? c->forced_steps : sjmi::stage1_pick_steps(total_len));
        if (fail(c, "plain accept", sjmi::batch_plain_accept_launch((const uint32_t*)d_indexes, (const sjmi::Stage1Result*)d_result,
                                                                    (const unsigned long long*)d_doc_offsets, n_docs,
                                                                    (unsigned long long*)d_index_offsets, (uint32_t*)d_doc_status,
                                                                    c->d_batch_flags, st0, hint, true)))
            return SJMI_ERR_HIP;
        d_skip = c->d_batch_flags + 1;
    }
    const int rc = stage1_batch_isolated_device_impl(c, d_buf, total_len, d_doc_offsets, n_docs, d_indexes, index_capacity,
                                                     d_index_offsets, d_doc_status, d_result, stream, d_skip);
    if (rc != SJMI_OK) return rc;
    if (d_skip) {  // (set after stage1_device_impl's parity_out, which clears it)
        c->accept_buf = d_buf;
        c->accept_len = total_len;
        c->accept_valid = true;
    }
    *d_skip_out = d_skip;
    return SJMI_OK;
}

int sjmi_stage1_batch_isolated_device(sjmi_ctx* c, const void* d_buf, uint64_t total_len, const void* d_doc_offsets,
                                      uint64_t n_docs, void* d_indexes, uint64_t index_capacity, void* d_index_offsets,
                                      void* d_doc_status, void* d_result, void* stream) {
    const uint32_t* d_skip = nullptr;
    return stage1_batch_optimistic(c, d_buf, total_len, d_doc_offsets, n_docs, d_indexes, index_capacity, d_index_offsets, d_doc_status,
                                   d_result, stream, &d_skip);
}
static int stage1_batch_isolated_device_impl(sjmi_ctx* c, const void* d_buf, uint64_t total_len, const void* d_doc_offsets,
                                             uint64_t n_docs, void* d_indexes, uint64_t index_capacity, void* d_index_offsets,
                                             void* d_doc_status, void* d_result, void* stream, const uint32_t* d_skip) {
    if (!c || !d_buf || !d_doc_offsets || !d_indexes || !d_index_offsets || !d_doc_status || !d_result) return SJMI_ERR_ARG;
    if (total_len >= (1ull << 32)) return SJMI_ERR_ARG;
    if (fail(c, "hipSetDevice", hipSetDevice(c->device))) return SJMI_ERR_HIP;
    if (!grow(c, (void**)&c->d_doccnt, &c->doccnt_bytes, sjmi::batch_isolated_workspace_bytes(n_docs), "hipMalloc(doccnt)"))
        return SJMI_ERR_HIP;
    if (!d_skip) {
        // no plain pass in front of these per-document passes (switched off, misaligned buffers, an empty batch): block
        // parities / an acceptance flag recorded for the same pointer and length by an EARLIER launch describe other bytes;
        // the string pass of this batch must take the sanitized copy, not them
        c->par_valid = false;
        c->accept_valid = false;
    }
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    if (fail(c, "isolated batch launch",
             sjmi::batch_isolated_launch((const uint8_t*)d_buf, (const unsigned long long*)d_doc_offsets, n_docs,
                                         (uint32_t*)d_indexes, index_capacity, (unsigned long long*)d_index_offsets,
                                         (uint32_t*)d_doc_status, c->d_doccnt, (sjmi::Stage1Result*)d_result, st, total_len, d_skip)))
        return SJMI_ERR_HIP;
    return SJMI_OK;
}

int sjmi_stage1_batch_isolated(sjmi_ctx* c, const uint8_t* buf, uint64_t total_len, const uint64_t* doc_offsets,
                               uint64_t n_docs, uint32_t* indexes, uint64_t index_capacity, uint64_t* index_offsets,
                               uint32_t* doc_status, uint64_t* count, uint32_t* status) {
    if (!c || (!buf && total_len) || !doc_offsets || !indexes || !index_offsets || !doc_status || !count || !status)
        return SJMI_ERR_ARG;
    if (total_len > c->capacity || total_len >= (1ull << 32)) {
        c->err = "batch larger than the context capacity";
        return SJMI_ERR_CAPACITY;
    }
    if (bad_offsets(c, doc_offsets, n_docs, total_len, false)) return SJMI_ERR_ARG;
    if (fail(c, "hipSetDevice", hipSetDevice(c->device))) return SJMI_ERR_HIP;
    const size_t ob = (n_docs + 1) * sizeof(unsigned long long);
    // document offsets | index offsets | document statuses
    if (!grow(c, (void**)&c->d_docoff, &c->docoff_bytes, 2 * ob + (n_docs + 16) * sizeof(uint32_t) + 64, "hipMalloc(docoff)"))
        return SJMI_ERR_HIP;
    unsigned long long* d_io = c->d_docoff + (n_docs + 1);
    uint32_t* d_st = reinterpret_cast<uint32_t*>(c->d_docoff + 2 * (n_docs + 1));
    void* d_res = (uint8_t*)c->d_ws + sjmi::WS_RESULT_OFFSET;
    if ((total_len && fail(c, "H2D", hipMemcpyAsync(c->d_in, buf, total_len, hipMemcpyHostToDevice, c->stream))) ||
        fail(c, "H2D(offsets)", hipMemcpyAsync(c->d_docoff, doc_offsets, ob, hipMemcpyHostToDevice, c->stream)))
        return SJMI_ERR_HIP;
    const uint64_t dev_cap = c->capacity + 2 < index_capacity ? c->capacity + 2 : index_capacity;
    const int rc = sjmi_stage1_batch_isolated_device(c, c->d_in, total_len, c->d_docoff, n_docs, c->d_idx, dev_cap, d_io, d_st,
                                                     d_res, c->stream);
    if (rc != SJMI_OK) return rc;
    if (fail(c, "D2H(result)", hipMemcpyAsync(c->h_res, d_res, sizeof(sjmi_stage1_result), hipMemcpyDeviceToHost, c->stream)) ||
        fail(c, "D2H(io)", hipMemcpyAsync(index_offsets, d_io, ob, hipMemcpyDeviceToHost, c->stream)) ||
        (n_docs && fail(c, "D2H(status)", hipMemcpyAsync(doc_status, d_st, n_docs * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream))) ||
        fail(c, "sync", hipStreamSynchronize(c->stream)))
        return SJMI_ERR_HIP;
    *status = c->h_res->status & 0xFFu;
    *count = c->h_res->count;
    if (c->h_res->status & SJMI_ST_CAPACITY) return SJMI_ERR_CAPACITY;
    if (fail(c, "D2H(indexes)", hipMemcpyAsync(indexes, c->d_idx, (c->h_res->count + 1) * sizeof(uint32_t),
                                               hipMemcpyDeviceToHost, c->stream)) ||
        fail(c, "sync", hipStreamSynchronize(c->stream)))
        return SJMI_ERR_HIP;
    c->idx_last = c->d_idx;  // (a batch call's indexes are always in the context's own array, never in a caller's zero-copy view)
    c->idx_last_host = nullptr;
    c->last_len = total_len;
    c->last_count = c->h_res->count;
    c->last_valid = true;
    c->last_ndocs = n_docs;
    c->last_batch = true;
    return SJMI_OK;
}

// The fused batch pipeline (round 5: re-ordered so that an ACCEPTED batch costs eight queue entries; round 6: a REPAIR stage
// between the plain pass and the per-document passes).
//   stage A   k_stage1_batch   ONE plain launch over the packed batch, per-block side outputs; its workers zero the string pass's
//                              workspace on their way out, the acceptance flags sit in the header of its own (zeroed) workspace half
//             k_strings<true>  over the batch itself, its record inside that workspace (leaves at once behind a stage-1 verdict)
//             k_doc_prepare    per boundary: the separator check (ORed into flags[0]), index ranges, string ordinals, predicted tape lengths
//             k_batch_layout   decides: accepted?  zeroes the walk's records, hands the string record over, scans the lengths
//             k_tape_offsets
//   stage B   (round 6; every kernel leaves at once when A was accepted)  A batch A rejects -- ONE document in a million fails
//             stage 1, or the documents are not separated by control characters -- used to fall to stage C, 2.2 x the cost of an
//             accepted batch.  Now: k_doc_pass<false> gives every document ITS verdict (16 lanes per document, every carry from
//             zero: StructuralIndexer.java:196-303 per document), the sanitized copy blanks the failing documents, and the SAME
//             five kernels of stage A run over the copy -- exact there, because every surviving document begins and ends outside a
//             string whatever separates it from its neighbours; the only thing a missing separator can still do is let a scalar
//             run on across a boundary, which k_doc_prepare's relaxed rule rules out (else: stage C).  A failing document keeps
//             its verdict, has no structurals and a two-word tape slot.
//   stage C   (every kernel leaves at once when A or B was accepted)  the per-document stage-1 passes with their index arrays,
//             the string pass over the sanitized copy, ordinals, the walk into scratch tapes + packing.
//   k_tok_stream, k_coop_walk (list), k_slow_doubles: ONE set of walkers behind all three (a device flag says where the tapes go).
// PIPE_OPTIMISTIC: stage A and the walkers, nothing else; a batch that does not qualify comes back with SJMI_ST_REJECTED.
// PIPE_EXACT: A, B, C.  PIPE_REJECTED: B and the walkers -- the call to make after SJMI_ST_REJECTED; what B cannot take either comes
// back REJECTED again and is PIPE_EXACT's.  No host round trip inside any of them.
enum PipeMode { PIPE_OPTIMISTIC, PIPE_EXACT, PIPE_REJECTED };
static int parse_batch_pipeline(sjmi_ctx* c, const void* d_buf, uint64_t total_len, const void* d_doc_offsets, uint64_t n_docs,
                                void* d_indexes, uint64_t index_capacity, void* d_index_offsets, void* d_doc_status,
                                void* d_string_buffer, uint64_t string_capacity, void* d_doc_string_offsets, int max_depth,
                                void* d_tape, uint64_t tape_capacity, void* d_tape_offsets, void* d_doc_errors, void* d_result,
                                void* stream, PipeMode mode) {
    if (!c || !d_buf || !d_doc_offsets || !d_indexes || !d_index_offsets || !d_doc_status || !d_string_buffer ||
        !d_doc_string_offsets || !d_tape || !d_tape_offsets || !d_doc_errors || !d_result || max_depth < 1 || index_capacity < 1)
        return SJMI_ERR_ARG;
    if (total_len >= (1ull << 32)) return SJMI_ERR_ARG;
    if (fail(c, "hipSetDevice", hipSetDevice(c->device))) return SJMI_ERR_HIP;
    const bool optimistic_only = mode == PIPE_OPTIMISTIC;
    sjmi_batch_result* r = (sjmi_batch_result*)d_result;
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    const uint64_t bound = index_capacity - 1;
    const uint64_t soff_cap = bound + 64;
    static const bool optimistic = !(getenv("SJMI_BATCH_OPTIMISTIC") && atoi(getenv("SJMI_BATCH_OPTIMISTIC")) == 0);
    static const bool repair_on = !(getenv("SJMI_BATCH_REPAIR") && atoi(getenv("SJMI_BATCH_REPAIR")) == 0);
    // (a batch of ONE document has no token-walker path behind tapes laid out in advance -- walk_launch takes the cooperative
    //  walker for it -- so the optimistic entry rejects it like any other batch it cannot take; the exact entry serves it)
    const bool aligned = optimistic && n_docs && total_len && !((uintptr_t)d_buf & 15) && !((uintptr_t)d_indexes & 15);
    const bool try_plain = aligned && mode != PIPE_REJECTED && !(optimistic_only && n_docs < 2);
    // stage B needs the FAST kernel (its scanner writes the caller's record: a SAFE launch's record is copied out behind the
    // kernel whether it ran or not) and more than one document
    const bool try_repair = aligned && repair_on && !optimistic_only && n_docs > 1 && !(launch_flags(c) & (sjmi::FLAG_SAFE | sjmi::DBG_NO_LOOKBACK));
    // PIPE_REJECTED queues the repair stage and the walkers, nothing else: a batch stage B cannot take either (rare: a scalar running
    // on across a boundary without a separator, a document ending in a backslash) comes back with SJMI_ST_REJECTED once more and is
    // the exact call's -- stage C's ~20 queue entries (100 us of kernels that leave at once) stay off the repaired batch's path
    const bool repair_only = mode == PIPE_REJECTED && try_repair;
    if (!grow(c, &c->d_ws_walk, &c->ws_walk_bytes, sjmi::walk_workspace_bytes(bound, n_docs), "hipMalloc(ws_walk)")) return SJMI_ERR_HIP;
    c->accept_valid = false;
    if (!try_plain && !try_repair) {
        if (optimistic_only) {  // (nothing the optimistic pipeline could run on: say so in the record)
            if (fail(c, "reject", sjmi::batch_reject_launch((sjmi::Stage1Result*)&r->stage1, st))) return SJMI_ERR_HIP;
            return SJMI_OK;
        }
        // the per-document passes only (switched off, misaligned buffers, an empty batch), as the three separate calls would
        int rc = stage1_batch_isolated_device_impl(c, d_buf, total_len, d_doc_offsets, n_docs, d_indexes, index_capacity, d_index_offsets,
                                                   d_doc_status, &r->stage1, stream, nullptr);
        if (rc != SJMI_OK) return rc;
        rc = strings_batch_impl(c, d_buf, total_len, d_indexes, bound, d_doc_offsets, d_index_offsets, n_docs, false, nullptr, d_string_buffer,
                                string_capacity, d_doc_string_offsets, &r->strings, st);
        if (rc != SJMI_OK) return rc;
        if (fail(c, "walk launch",
                 sjmi::walk_launch((const uint8_t*)d_buf, (const unsigned long long*)d_doc_offsets, n_docs, (const uint32_t*)d_indexes,
                                   bound, (const unsigned long long*)d_index_offsets, (const uint32_t*)d_doc_status,
                                   (const uint8_t*)d_string_buffer, c->d_doc_ord, 0, max_depth, (unsigned long long*)d_tape, tape_capacity,
                                   (unsigned long long*)d_tape_offsets, (int32_t*)d_doc_errors, c->d_ws_walk, (sjmi::WalkResult*)&r->walk, st,
                                   (const sjmi::Stage1Result*)&r->stage1, (const sjmi::UnescapeResult*)&r->strings, c->d_soff)))
            return SJMI_ERR_HIP;
        return SJMI_OK;
    }
    const size_t strm_bytes = (sjmi::strings_workspace_bytes(total_len) + 15) & ~(size_t)15;
    if (!grow(c, (void**)&c->d_soff, &c->soff_bytes, soff_cap * sizeof(uint32_t), "hipMalloc(soff)") ||
        !grow(c, (void**)&c->d_blk_ord, &c->blk_ord_bytes, (total_len / 64 + 2) * sizeof(uint32_t), "hipMalloc(blk_ord)") ||
        !grow(c, (void**)&c->d_doc_ord, &c->doc_ord_bytes, (n_docs + 2) * sizeof(unsigned long long), "hipMalloc(doc_ord)") ||
        !grow(c, &c->d_ws_strm, &c->ws_strm_bytes, strm_bytes, "hipMalloc(ws_strm)"))
        return SJMI_ERR_HIP;
    if (!c->d_batch_flags && fail(c, "hipMalloc(batch flags)", hipMalloc((void**)&c->d_batch_flags, 64))) return SJMI_ERR_HIP;
    // the pipeline's stage flags: pf[0] != 0 = stage A accepted, pf[1] != 0 = A or B accepted (the tapes are laid out in advance)
    uint32_t* const pf = c->d_batch_flags + 8;
    c->soff_idx = nullptr;
    sjmi::UnescapeResult* const d_u = sjmi::strings_workspace_result(c->d_ws_strm);
    const sjmi::WalkPrepared wp = sjmi::walk_prepared(c->d_ws_walk, bound, n_docs);
    sjmi::DocPrepare pa;
    pa.idx = (const uint32_t*)d_indexes;
    pa.doc_offsets = (const unsigned long long*)d_doc_offsets;
    pa.n_docs = n_docs;
    pa.total_len = total_len;
    pa.blk_ord = c->d_blk_ord;
    pa.soff = c->d_soff;
    pa.strings = d_u;
    pa.stage1 = (const sjmi::Stage1Result*)&r->stage1;
    pa.index_offsets = (unsigned long long*)d_index_offsets;
    pa.doc_status = (uint32_t*)d_doc_status;
    pa.doc_ord = c->d_doc_ord;
    pa.doc_str_offsets = (unsigned long long*)d_doc_string_offsets;
    pa.lens = wp.lens;
    pa.chunk_sums = wp.chunk_sums;
    pa.metas = wp.metas;
    sjmi::BatchLayout bl = {};
    bl.stage1 = (const sjmi::Stage1Result*)&r->stage1;
    bl.stage1_out = (sjmi::Stage1Result*)&r->stage1;
    bl.strings_ws = d_u;
    bl.strings_out = (sjmi::UnescapeResult*)&r->strings;
    bl.walk = (sjmi::WalkResult*)&r->walk;
    bl.chunk_sums = wp.chunk_sums;
    bl.n_docs = n_docs;
    bl.tape_capacity = tape_capacity;
    bl.tape_offsets = (unsigned long long*)d_tape_offsets;
    bl.optimistic_only = optimistic_only;
    bl.pipe_flags = optimistic_only ? nullptr : pf;
    const uint32_t* d_laid_out = nullptr;  // device flag != 0: the tapes are laid out (what the walkers and the packing kernels ask)
    if (try_plain) {
        // ---- stage A (1): the plain pass ----
        int rc0;
        {
            const AutoSafeOff plain_only(c);  // (a tripped liveness bound only rejects the plain pass)
            c->batch_side = true;
            c->s1_zero2 = c->d_ws_strm;
            c->s1_zero2_bytes = strm_bytes;
            // (experiments: SJMI_BATCH_STEPS = granule of the pipeline's plain pass in units of 4 KiB)
            static const int batch_steps = getenv("SJMI_BATCH_STEPS") ? atoi(getenv("SJMI_BATCH_STEPS")) : 0;
            const int keep_steps = c->forced_steps;
            if (batch_steps == 1 || batch_steps == 2 || batch_steps == 4) c->forced_steps = batch_steps;
            rc0 = stage1_device_impl(c, d_buf, total_len, d_indexes, index_capacity, &r->stage1, stream, 0);
            c->forced_steps = keep_steps;
            c->batch_side = false;
            c->s1_zero2 = nullptr;
            c->s1_zero2_bytes = 0;
        }
        if (rc0 != SJMI_OK) return rc0;
        if (!(c->par_valid && c->par_buf == d_buf && c->par_len == total_len) || !c->ws_dev_last) {
            c->err = "batch pipeline: the plain pass left no block parities";
            return SJMI_ERR_HIP;
        }
        uint32_t* const flags = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(c->ws_dev_last) + sjmi::WS_BATCH_FLAGS_OFFSET);
        // ---- (2) the string pass over the batch itself (not behind a stage-1 verdict: nothing would use it);
        //      (3) one pass over the boundaries; (4) the decision and the tape layout ----
        if (fail(c, "strings launch",
                 sjmi::strings_launch((const uint8_t*)d_buf, total_len, c->d_blkpar, (uint8_t*)d_string_buffer, string_capacity, c->d_soff,
                                      soff_cap, c->d_blk_ord, c->d_ws_strm, d_u, st, nullptr, nullptr, sjmi::StringsAlt(), true,
                                      &r->stage1.status)))
            return SJMI_ERR_HIP;
        pa.buf = (const uint8_t*)d_buf;
        pa.blkidx = c->d_blkidx;
        pa.blkw = c->d_blkw;
        pa.blkpar = c->d_blkpar;
        pa.flags = flags;
        bl.flags = flags;
        bl.stage = 0;
        if (fail(c, "prepare launch", sjmi::batch_prepare_launch(pa, st)) ||
            fail(c, "layout launch", sjmi::batch_layout_launch(bl, c->d_ws_walk, bound, wp.lens, wp.metas, (int32_t*)d_doc_errors, st)))
            return SJMI_ERR_HIP;
        d_laid_out = optimistic_only ? flags + 1 : pf + 1;
    } else if (fail(c, "memset(pipe flags)", hipMemsetAsync(pf, 0, 8, st))) {  // (PIPE_REJECTED: stage A is known to fail)
        return SJMI_ERR_HIP;
    }
    if (!optimistic_only) {
        const int steps = c->forced_steps ? c->forced_steps : sjmi::stage1_pick_steps(total_len);
        if (!grow(c, (void**)&c->d_doccnt, &c->doccnt_bytes, sjmi::batch_isolated_workspace_bytes(n_docs), "hipMalloc(doccnt)") ||
            !grow(c, (void**)&c->d_copy, &c->copy_bytes, total_len + 2 * SJMI_PADDING + 64, "hipMalloc(copy)") ||
            !grow(c, (void**)&c->d_blkpar2, &c->blkpar2_bytes, sjmi::strings_parity_words(total_len) * sizeof(unsigned long long), "hipMalloc(blkpar2)") ||
            !grow(c, &c->d_ws_par, &c->ws_par_bytes, sjmi::stage1_workspace_bytes(total_len, steps), "hipMalloc(ws_par)"))
            return SJMI_ERR_HIP;
        const uint32_t* const skip_b = pf;      // stages B and the verdicts: not behind an accepted A
        const uint32_t* const skip_c = pf + 1;  // stage C: not behind an accepted A or B
        // ---- every document's own stage-1 verdict (B needs nothing else of the per-document passes; C the rest) ----
        //      with stage B behind it the same pass makes B's sanitized copy on the way: the documents' bytes, a failing one blank
        if (fail(c, "verdicts launch", sjmi::batch_verdicts_launch((const uint8_t*)d_buf, (const unsigned long long*)d_doc_offsets, n_docs,
                                                                   (uint32_t*)d_doc_status, c->d_doccnt, st, total_len, skip_b,
                                                                   try_repair ? c->d_copy : nullptr)))
            return SJMI_ERR_HIP;
        c->par_valid = false;  // (whose parities the context holds is decided on the device from here on)
        const unsigned long long* copy_par = c->d_blkpar2;  // the copy's block parities
        if (try_repair) {
            // ---- stage B: the plain pipeline over the copy.  Its stage-1 launch has a workspace of its own (a launch that may
            //      leave at once cannot take part in the context's two alternating halves) and its string pass a zeroed one ----
            if (!grow(c, (void**)&c->d_blkidx, &c->blkidx_bytes, sjmi::stage1_block_entries(total_len) * sizeof(uint32_t), "hipMalloc(blkidx)") ||
                !grow(c, (void**)&c->d_blkw, &c->blkw_bytes, sjmi::stage1_block_entries(total_len) * sizeof(uint16_t), "hipMalloc(blkw)"))
                return SJMI_ERR_HIP;
            sjmi::Stage1Extras ex;
            ex.blkpar = c->d_blkpar2;
            ex.skip = skip_b;
            ex.blkidx = c->d_blkidx;
            ex.blkw = c->d_blkw;
            ex.result_out = &r->stage1;
            const uint64_t dev_cap = index_capacity;
            if (fail(c, "repair: plain pass", sjmi::stage1_launch(c->d_copy, total_len, (uint32_t*)d_indexes, dev_cap, c->d_ws_par, steps, st,
                                                                  nullptr, nullptr, launch_flags(c), ex)))
                return SJMI_ERR_HIP;
            note_launch(c, st);
            uint32_t* const flags_b = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(c->d_ws_par) + sjmi::WS_BATCH_FLAGS_OFFSET);
            if (fail(c, "repair: strings launch",
                     sjmi::strings_launch(c->d_copy, total_len, c->d_blkpar2, (uint8_t*)d_string_buffer, string_capacity, c->d_soff, soff_cap,
                                          c->d_blk_ord, c->d_ws_strm, d_u, st, nullptr, nullptr, sjmi::StringsAlt(), false, skip_b)))
                return SJMI_ERR_HIP;
            sjmi::DocPrepare pb = pa;
            pb.buf = c->d_copy;
            pb.boundary_buf = (const uint8_t*)d_buf;
            pb.blkidx = c->d_blkidx;
            pb.blkw = c->d_blkw;
            pb.blkpar = c->d_blkpar2;
            pb.flags = flags_b;
            pb.gate = skip_b;
            pb.status_in = (const uint32_t*)d_doc_status;
            pb.relaxed = true;
            sjmi::BatchLayout lb = bl;
            lb.flags = flags_b;
            lb.stage = 1;
            lb.optimistic_only = repair_only;
            lb.gate = skip_b;
            lb.status_or = sjmi::batch_status_or(c->d_doccnt, n_docs);
            if (fail(c, "repair: prepare launch", sjmi::batch_prepare_launch(pb, st)) ||
                fail(c, "repair: layout launch", sjmi::batch_layout_launch(lb, c->d_ws_walk, bound, wp.lens, wp.metas, (int32_t*)d_doc_errors, st)))
                return SJMI_ERR_HIP;
            d_laid_out = pf + 1;
        }
        // ---- stage C, every kernel gated on pf[1] == 0: its OWN sanitized copy (the whole buffer, failing documents blank, an odd
        //      trailing backslash run shortened by one: a batch B does not take may have bytes between its documents and documents
        //      that end in a backslash) and the copy's block parities from a parity-only launch ----
        if (!repair_only) {
            sjmi::Stage1Extras ex;
            ex.blkpar = c->d_blkpar2;
            ex.skip = skip_c;
            if (fail(c, "sanitize", sjmi::strings_sanitize_launch((const uint8_t*)d_buf, total_len, (const unsigned long long*)d_doc_offsets,
                                                                  nullptr, n_docs, c->d_copy, skip_c, st, (const uint32_t*)d_doc_status)) ||
                fail(c, "parity launch", sjmi::stage1_launch(c->d_copy, total_len, nullptr, 0, c->d_ws_par, steps, st, nullptr, nullptr,
                                                             (launch_flags(c) & ~sjmi::DBG_NO_LOOKBACK) | sjmi::DBG_NO_WRITE, ex)))
                return SJMI_ERR_HIP;
            note_launch(c, st);
        }
        //      ... the per-document index arrays, the string pass over the copy (fills the record k_batch_layout zeroed), ordinals
        if (!repair_only &&
            (fail(c, "indexes launch", sjmi::batch_indexes_launch((const uint8_t*)d_buf, (const unsigned long long*)d_doc_offsets, n_docs,
                                                                 (uint32_t*)d_indexes, index_capacity, (unsigned long long*)d_index_offsets,
                                                                 (uint32_t*)d_doc_status, c->d_doccnt, (sjmi::Stage1Result*)&r->stage1, st,
                                                                 total_len, skip_c)) ||
            fail(c, "strings launch",
                 sjmi::strings_launch(c->d_copy, total_len, copy_par, (uint8_t*)d_string_buffer, string_capacity, c->d_soff, soff_cap,
                                      c->d_blk_ord, c->d_ws_strm, (sjmi::UnescapeResult*)&r->strings, st, nullptr, nullptr, sjmi::StringsAlt(),
                                      false, skip_c)) ||
            fail(c, "doc ordinals",
                 sjmi::strings_doc_ordinals_launch(c->d_copy, copy_par, sjmi::StringsAlt(), total_len,
                                                   (const unsigned long long*)d_doc_offsets, n_docs, c->d_blk_ord, c->d_soff,
                                                   (const sjmi::UnescapeResult*)&r->strings, c->d_doc_ord,
                                                   (unsigned long long*)d_doc_string_offsets, st, skip_c))))
            return SJMI_ERR_HIP;
    }
    const bool layout_done = try_plain || try_repair;
    if (fail(c, "walk launch",
             sjmi::walk_launch((const uint8_t*)d_buf, (const unsigned long long*)d_doc_offsets, n_docs, (const uint32_t*)d_indexes,
                               bound, (const unsigned long long*)d_index_offsets, (const uint32_t*)d_doc_status,
                               (const uint8_t*)d_string_buffer, c->d_doc_ord, 0, max_depth,
                               (unsigned long long*)d_tape, tape_capacity, (unsigned long long*)d_tape_offsets,
                               (int32_t*)d_doc_errors, c->d_ws_walk, (sjmi::WalkResult*)&r->walk, st,
                               (const sjmi::Stage1Result*)&r->stage1, (const sjmi::UnescapeResult*)&r->strings, c->d_soff, false, false,
                               sjmi::SingleDocTail(), layout_done ? d_laid_out : nullptr, layout_done, optimistic_only || repair_only)))
        return SJMI_ERR_HIP;
    return SJMI_OK;
}

int sjmi_parse_batch_device(sjmi_ctx* c, const void* d_buf, uint64_t total_len, const void* d_doc_offsets, uint64_t n_docs,
                            void* d_indexes, uint64_t index_capacity, void* d_index_offsets, void* d_doc_status,
                            void* d_string_buffer, uint64_t string_capacity, void* d_doc_string_offsets, int max_depth,
                            void* d_tape, uint64_t tape_capacity, void* d_tape_offsets, void* d_doc_errors, void* d_result,
                            void* stream) {
    return parse_batch_pipeline(c, d_buf, total_len, d_doc_offsets, n_docs, d_indexes, index_capacity, d_index_offsets, d_doc_status,
                                d_string_buffer, string_capacity, d_doc_string_offsets, max_depth, d_tape, tape_capacity, d_tape_offsets,
                                d_doc_errors, d_result, stream, PIPE_EXACT);
}

int sjmi_parse_batch_device_optimistic(sjmi_ctx* c, const void* d_buf, uint64_t total_len, const void* d_doc_offsets, uint64_t n_docs,
                                       void* d_indexes, uint64_t index_capacity, void* d_index_offsets, void* d_doc_status,
                                       void* d_string_buffer, uint64_t string_capacity, void* d_doc_string_offsets, int max_depth,
                                       void* d_tape, uint64_t tape_capacity, void* d_tape_offsets, void* d_doc_errors, void* d_result,
                                       void* stream) {
    return parse_batch_pipeline(c, d_buf, total_len, d_doc_offsets, n_docs, d_indexes, index_capacity, d_index_offsets, d_doc_status,
                                d_string_buffer, string_capacity, d_doc_string_offsets, max_depth, d_tape, tape_capacity, d_tape_offsets,
                                d_doc_errors, d_result, stream, PIPE_OPTIMISTIC);
}

int sjmi_parse_batch_device_rejected(sjmi_ctx* c, const void* d_buf, uint64_t total_len, const void* d_doc_offsets, uint64_t n_docs,
                                     void* d_indexes, uint64_t index_capacity, void* d_index_offsets, void* d_doc_status,
                                     void* d_string_buffer, uint64_t string_capacity, void* d_doc_string_offsets, int max_depth,
                                     void* d_tape, uint64_t tape_capacity, void* d_tape_offsets, void* d_doc_errors, void* d_result,
                                     void* stream) {
    return parse_batch_pipeline(c, d_buf, total_len, d_doc_offsets, n_docs, d_indexes, index_capacity, d_index_offsets, d_doc_status,
                                d_string_buffer, string_capacity, d_doc_string_offsets, max_depth, d_tape, tape_capacity, d_tape_offsets,
                                d_doc_errors, d_result, stream, PIPE_REJECTED);
}

// (the result records of the three stages come back in one place, sjmi::SingleDocPack, written by the walk's last launch)
using SingleDocResults = sjmi::SingleDocPack;
static_assert(sizeof(sjmi_stage1_result) == sizeof(sjmi::Stage1Result) && sizeof(sjmi_unescape_result) == sizeof(sjmi::UnescapeResult) &&
              sizeof(sjmi_walk_result) == sizeof(sjmi::WalkResult), "C ABI records mirror the device records");

// SimdJsonParser.parse(byte[], int) with ALL THREE stages on the GPU (SimdJsonParser.java:35-40): H2D of the document,
// stage 1, string records, the cooperative walker, D2H of the tape and the string buffer -- the structural indexes never
// leave the device.  Two host synchronisations (the result records, then the outputs).
int sjmi_parse_document(sjmi_ctx* c, const uint8_t* buf, uint64_t len, int max_depth, uint64_t* tape, uint64_t tape_capacity,
                        uint64_t* tape_len, uint8_t* string_buffer, uint64_t string_capacity, uint64_t* strings_len,
                        int32_t* error, uint32_t* stage1_status) {
    if (!c || (!buf && len) || !tape || !tape_len || !string_buffer || !strings_len || !error || !stage1_status || max_depth < 1)
        return SJMI_ERR_ARG;
    if (len > c->capacity || len >= (1ull << 32)) {
        c->err = "document larger than the context capacity";
        return SJMI_ERR_CAPACITY;
    }
    if (fail(c, "hipSetDevice", hipSetDevice(c->device))) return SJMI_ERR_HIP;
    const uint64_t bound = len + 1;  // structurals: at most one per byte
    const size_t need_sb = (size_t)len + 4 * ((size_t)len / 2 + 2) + 64;
    if (!grow(c, (void**)&c->d_sb, &c->sb_bytes, need_sb, "hipMalloc(sb)") ||
        !grow(c, (void**)&c->d_soff, &c->soff_bytes, ((size_t)len / 2 + 66) * sizeof(uint32_t), "hipMalloc(soff)") ||
        !grow(c, &c->d_ws_walk, &c->ws_walk_bytes, sjmi::walk_workspace_bytes(bound, 1), "hipMalloc(ws_walk)") ||
        !grow(c, (void**)&c->d_tape, &c->tape_bytes, (2 * (size_t)bound + 16) * sizeof(unsigned long long), "hipMalloc(tape)"))
        return SJMI_ERR_HIP;
    if (!c->d_single && fail(c, "hipMalloc(single)", hipMalloc(&c->d_single, 512))) return SJMI_ERR_HIP;
    if (!c->h_single && fail(c, "hipHostMalloc(single)", hipHostMalloc(&c->h_single, 512))) return SJMI_ERR_HIP;
    if (!c->copy_stream && (fail(c, "hipStreamCreate", hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking)) ||
                            fail(c, "hipEventCreate", hipEventCreateWithFlags(&c->strings_ready, hipEventDisableTiming))))
        return SJMI_ERR_HIP;
    // layout of d_single: doc_offsets[2] | index_offsets[2] | doc_str_offsets[2] | tape_offsets[2] | status | error | results
    unsigned long long* d64 = (unsigned long long*)c->d_single;
    unsigned long long *d_doc = d64, *d_io = d64 + 2, *d_dso = d64 + 4, *d_to = d64 + 6;
    uint32_t* d_st = (uint32_t*)(d64 + 8);
    int32_t* d_err = (int32_t*)(d64 + 9);
    sjmi::UnescapeResult* d_ures = nullptr;  // (inside the string pass's workspace, zeroed with it)
    sjmi::WalkResult* d_wres = (sjmi::WalkResult*)(d64 + 13);
    sjmi::Stage1Result* d_res1 = (sjmi::Stage1Result*)(d64 + 20);
    SingleDocResults* h = (SingleDocResults*)c->h_single;
    // ZERO-COPY OUTPUTS: a tape that lives in page-locked, device-visible host memory (sjmi_host_register / hipHostMalloc: what
    // SimdJsonParser does with its buffers) is written by the walkers directly, over PCIe, and the packed result records land in
    // the context's pinned page the same way -- no download of either, ONE host synchronisation instead of two (twitter.json:
    // 0.147 -> ~0.12 ms).  A pageable tape takes the staged path below.
    static const bool zero_copy_off = getenv("SJMI_ZERO_COPY") && atoi(getenv("SJMI_ZERO_COPY")) == 0;
    drop_stale_views(c);
    if (c->zc_host != tape) {
        c->zc_host = tape;
        c->zc_dev = nullptr;
        void* dp = nullptr;
        if (!zero_copy_off && tape_capacity >= 16 && hipHostGetDevicePointer(&dp, tape, 0) == hipSuccess && dp) c->zc_dev = (unsigned long long*)dp;
        else (void)hipGetLastError();  // (not device-visible: not an error)
    }
    if (!c->h_single_dev) {
        void* dp = nullptr;
        if (hipHostGetDevicePointer(&dp, c->h_single, 0) == hipSuccess && dp) c->h_single_dev = dp;
        else (void)hipGetLastError();
    }
    const bool zero_copy = c->zc_dev != nullptr && c->h_single_dev != nullptr;
    const int steps = c->forced_steps ? c->forced_steps : sjmi::stage1_pick_steps(len);
    if (!upload_document(c, buf, len)) return SJMI_ERR_HIP;
    c->soff_idx = nullptr;
    // The string records are complete long before the tape: their download runs on a second stream while the walker works
    // (a large document: a tenth of the call).  The size comes from the unescape result, fetched on that stream too.
    sjmi_unescape_result* h_u_early = (sjmi_unescape_result*)((uint8_t*)c->h_single + 256);
    bool strings_in_flight = false;
    const bool early_strings = len >= (256u << 10);  // (below that the second stream's synchronisation costs more than it hides)
    // Latency: besides the kernels only ONE memset is queued (the string pass's workspace, which also holds its record) --
    // stage 1 through the device entry point (double-buffered workspace, the scanner writes the record), the walk's record and
    // list header zeroed by the setup kernel.
    (void)steps;
    const AutoSafeOff own_retry(c);  // (the retry below is this call's own)
    // (the string pass's workspace is zeroed by the stage-1 workers on their way out, and the delimiters / zeroed records the walk
    //  needs are written by the stage-1 scanner together with the result record: a memset and a launch less in the chain)
    const size_t strm_bytes = (sjmi::strings_workspace_bytes(len) + 15) & ~(size_t)15;
    if (!grow(c, &c->d_ws_strm, &c->ws_strm_bytes, strm_bytes, "hipMalloc(ws_strm)")) return SJMI_ERR_HIP;
    for (int attempt = 0; attempt < 2; ++attempt) {
        strings_in_flight = false;
        c->s1_zero2 = c->d_ws_strm;
        c->s1_zero2_bytes = strm_bytes;
        c->s1_single.doc_offsets = d_doc;
        c->s1_single.index_offsets = d_io;
        c->s1_single.doc_status = d_st;
        c->s1_single.doc_str_offsets = d_dso;
        c->s1_single.walk_result = reinterpret_cast<uint32_t*>(d_wres);
        c->s1_single.slow_header = static_cast<uint32_t*>(sjmi::walk_slow_header(c->d_ws_walk, bound, 1));
        const int s1rc = stage1_device_impl(c, c->d_in, len, c->d_idx, c->capacity + 2, d_res1, c->stream, 0);
        c->s1_zero2 = nullptr;
        c->s1_zero2_bytes = 0;
        c->s1_single = sjmi::Stage1Single();
        const bool setup_done = c->s1_single_done;
        c->s1_single_done = false;
        if (s1rc != SJMI_OK) return s1rc;
        const int src = strings_device_impl(c, c->d_in, len, c->d_sb, c->sb_bytes, c->d_soff, (size_t)len / 2 + 2, nullptr, nullptr, c->stream, &d_ures, true);
        if (src != SJMI_OK) return src;
        sjmi::SingleDocTail tail;
        tail.s1 = d_res1;
        tail.u = d_ures;
        tail.pack = zero_copy ? (sjmi::SingleDocPack*)c->h_single_dev : (sjmi::SingleDocPack*)((uint8_t*)c->d_single + 256);
        tail.in_place_cap = zero_copy ? tape_capacity : 0;
        unsigned long long* const walk_tape = zero_copy ? c->zc_dev : c->d_tape;
        const uint64_t walk_cap = zero_copy ? tape_capacity : 2 * bound + 8;
        tail.optimistic = true;  // (a large document's chunk path ends in k_chunk_finish: a document it declines comes back flagged)
        if ((early_strings && fail(c, "event", hipEventRecord(c->strings_ready, c->stream))) ||
            (!setup_done && fail(c, "setup", sjmi::single_doc_setup_launch(d_res1, len, d_doc, d_io, d_st, d_dso, c->stream, d_wres,
                                                                           sjmi::walk_slow_header(c->d_ws_walk, bound, 1)))) ||
            fail(c, "walk launch",
                 sjmi::walk_launch(c->d_in, d_doc, 1, c->d_idx, bound, d_io, d_st, c->d_sb, d_dso, 0, max_depth, walk_tape,
                                   walk_cap, d_to, d_err, c->d_ws_walk, d_wres, c->stream, d_res1, d_ures, c->d_soff, true, true, tail)) ||
            (!zero_copy && fail(c, "D2H", hipMemcpyAsync(h, tail.pack, sizeof *h, hipMemcpyDeviceToHost, c->stream))))
            return SJMI_ERR_HIP;
        // (everything of the main stream is queued: now the early look at the string records)
        if (early_strings) {
            if (fail(c, "wait", hipStreamWaitEvent(c->copy_stream, c->strings_ready, 0)) ||
                fail(c, "D2H", hipMemcpyAsync(h_u_early, d_ures, sizeof *h_u_early, hipMemcpyDeviceToHost, c->copy_stream)) ||
                fail(c, "sync", hipStreamSynchronize(c->copy_stream)))
                return SJMI_ERR_HIP;
            if (!(h_u_early->flags & 1u) && h_u_early->total_bytes && h_u_early->total_bytes <= string_capacity) {
                if (fail(c, "D2H(sb)", hipMemcpyAsync(string_buffer, c->d_sb, h_u_early->total_bytes, hipMemcpyDeviceToHost, c->copy_stream)))
                    return SJMI_ERR_HIP;
                strings_in_flight = true;
            }
        }
        if (fail(c, "sync", hipStreamSynchronize(c->stream))) return SJMI_ERR_HIP;
        if (h->fallback && !(h->s1.status & SJMI_ST_INTERNAL)) {
            // (rare) the chunk path declined the document (a depth swing beyond its relative levels, more than 64 levels): the
            // single-wave sweep, queued only now
            tail.optimistic = false;
            tail.no_chunks = true;
            if (fail(c, "setup", sjmi::single_doc_setup_launch(d_res1, len, d_doc, d_io, d_st, d_dso, c->stream, d_wres,
                                                               sjmi::walk_slow_header(c->d_ws_walk, bound, 1))) ||
                fail(c, "walk launch",
                     sjmi::walk_launch(c->d_in, d_doc, 1, c->d_idx, bound, d_io, d_st, c->d_sb, d_dso, 0, max_depth, walk_tape,
                                       walk_cap, d_to, d_err, c->d_ws_walk, d_wres, c->stream, d_res1, d_ures, c->d_soff, true, true, tail)) ||
                (!zero_copy && fail(c, "D2H", hipMemcpyAsync(h, tail.pack, sizeof *h, hipMemcpyDeviceToHost, c->stream))) ||
                fail(c, "sync", hipStreamSynchronize(c->stream)))
                return SJMI_ERR_HIP;
        }
        if (!(h->s1.status & SJMI_ST_INTERNAL) || c->ticket_mode) break;
        if (strings_in_flight && fail(c, "sync", hipStreamSynchronize(c->copy_stream))) return SJMI_ERR_HIP;
        c->ticket_mode = true;  // fast-mode liveness assumption failed: latch the safe mode and run again
    }
    if (strings_in_flight && fail(c, "sync", hipStreamSynchronize(c->copy_stream))) return SJMI_ERR_HIP;
    c->last_valid = false;
    *stage1_status = h->s1.status & 0xFFu;
    *tape_len = 0;
    *strings_len = 0;
    *error = h->err;
    if (h->s1.status & SJMI_ST_INTERNAL) {
        c->err = "look-back timeout";
        return SJMI_ERR_INTERNAL;
    }
    if (h->s1.status & SJMI_ST_CAPACITY) return SJMI_ERR_CAPACITY;
    if (h->err != 0) return SJMI_OK;  // a JSON error or SJMI_WALK_NEEDS_HOST: no tape
    const uint64_t words = h->to[1] - h->to[0];
    if (words > tape_capacity || h->u.total_bytes > string_capacity || (h->u.flags & 1u)) {
        c->err = "tape or string capacity too small";
        return SJMI_ERR_CAPACITY;
    }
    const bool copy_strings = h->u.total_bytes && !strings_in_flight;
    if ((!zero_copy && fail(c, "D2H(tape)", hipMemcpyAsync(tape, c->d_tape + h->to[0], words * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream))) ||
        (copy_strings && fail(c, "D2H(sb)", hipMemcpyAsync(string_buffer, c->d_sb, h->u.total_bytes, hipMemcpyDeviceToHost, c->stream))) ||
        ((!zero_copy || copy_strings) && fail(c, "sync", hipStreamSynchronize(c->stream))))
        return SJMI_ERR_HIP;
    *tape_len = words;
    *strings_len = h->u.total_bytes;
    return SJMI_OK;
}

int sjmi_host_register(sjmi_ctx* c, void* ptr, uint64_t bytes) {
    if (!c || !ptr || !bytes) return SJMI_ERR_ARG;
    if (fail(c, "hipSetDevice", hipSetDevice(c->device))) return SJMI_ERR_HIP;
    if (fail(c, "hipHostRegister", hipHostRegister(ptr, bytes, hipHostRegisterDefault))) return SJMI_ERR_HIP;
    g_pin_epoch.fetch_add(1, std::memory_order_acq_rel);  // (behind the call: a view taken from now on sees the registration)
    return SJMI_OK;
}

int sjmi_host_unregister(sjmi_ctx* c, void* ptr) {
    if (!c || !ptr) return SJMI_ERR_ARG;
    // every context forgets its zero-copy views, whatever the outcome -- bumped on BOTH sides of the call: a view another thread
    // takes between the first bump and hipHostUnregister is dropped again by the second
    g_pin_epoch.fetch_add(1, std::memory_order_acq_rel);
    const hipError_t e = hipHostUnregister(ptr);
    g_pin_epoch.fetch_add(1, std::memory_order_acq_rel);
    if (fail(c, "hipHostUnregister", e)) return SJMI_ERR_HIP;
    return SJMI_OK;
}

int sjmi_set_auto_safe(sjmi_ctx* c, int on) {
    if (!c) return SJMI_ERR_ARG;
    c->auto_safe = on != 0;
    return SJMI_OK;
}

int sjmi_set_tile_mode(sjmi_ctx* c, int ticket) {
    if (!c) return SJMI_ERR_ARG;
    c->ticket_mode = ticket != 0;
    return SJMI_OK;
}

int sjmi_debug_set_flags(sjmi_ctx* c, uint32_t flags) {
    if (!c) return SJMI_ERR_ARG;
    c->dbg = flags;
    return SJMI_OK;
}

int sjmi_set_profiling(sjmi_ctx* c, int on) {
    if (!c) return SJMI_ERR_ARG;
    c->profiling = on != 0;
    c->events_used = 0;
    return SJMI_OK;
}

int sjmi_kernel_time(sjmi_ctx* c, double* sum_ms, uint32_t* launches) {
    if (!c || !sum_ms || !launches) return SJMI_ERR_ARG;
    double total = 0;
    for (size_t i = 0; i < c->events_used; ++i) {
        float ms = 0;
        if (fail(c, "hipEventSynchronize", hipEventSynchronize(c->events[i].second)) ||
            fail(c, "hipEventElapsedTime", hipEventElapsedTime(&ms, c->events[i].first, c->events[i].second)))
            return SJMI_ERR_HIP;
        total += ms;
    }
    *sum_ms = total;
    *launches = (uint32_t)c->events_used;
    return SJMI_OK;
}

// ---- one document as a stream of chunks / one shard of a document split over GPUs: the protocol state in C ----------------
// (the reference has one byte[] per parse and no counterpart)
struct sjmi_stream {
    sjmi_ctx* c = nullptr;
    uint8_t* d_buf = nullptr;       // [keep bytes of the stream in front of the chunk | the chunk | padding]
    uint32_t* d_idx = nullptr;
    sjmi_stage1_result* d_res = nullptr;
    uint64_t max_chunk = 0, keep = 0, halo = 0;
    uint64_t have = 0;              // bytes of the stream kept in front of the next chunk (<= keep, the END of [0, keep))
    uint64_t offset = 0;            // stream offset of the next chunk
    int parity = 0;                 // in-string parity after the chunks so far
    uint32_t status = 0;
    bool finished = false;
};

int sjmi_stream_open(sjmi_ctx* c, uint64_t max_chunk_bytes, uint64_t halo_bytes, sjmi_stream** out) {
    if (!c || !out || !max_chunk_bytes || max_chunk_bytes >= (1ull << 32) || (halo_bytes & 63)) return SJMI_ERR_ARG;
    *out = nullptr;
    if (fail(c, "hipSetDevice", hipSetDevice(c->device))) return SJMI_ERR_HIP;
    sjmi_stream* s = new (std::nothrow) sjmi_stream();
    if (!s) return SJMI_ERR_ARG;
    s->c = c;
    s->max_chunk = (max_chunk_bytes + 63) / 64 * 64;
    s->halo = halo_bytes ? halo_bytes : 64;
    s->keep = s->halo > 4096 ? s->halo : 4096;  // what a chunk can be repeated with when its halo proves too short
    if (fail(c, "hipMalloc(stream)", hipMalloc((void**)&s->d_buf, s->keep + s->max_chunk + 2 * SJMI_PADDING)) ||
        fail(c, "hipMalloc(stream idx)", hipMalloc((void**)&s->d_idx, (s->max_chunk + 66) * sizeof(uint32_t) + s->keep)) ||
        fail(c, "hipMalloc(stream res)", hipMalloc((void**)&s->d_res, sizeof(sjmi_stage1_result)))) {
        sjmi_stream_close(s);
        return SJMI_ERR_HIP;
    }
    *out = s;
    return SJMI_OK;
}

void sjmi_stream_close(sjmi_stream* s) {
    if (!s) return;
    (void)hipSetDevice(s->c->device);
    (void)hipStreamSynchronize(s->c->stream);
    if (s->d_buf) (void)hipFree(s->d_buf);
    if (s->d_idx) (void)hipFree(s->d_idx);
    if (s->d_res) (void)hipFree(s->d_res);
    delete s;
}

int sjmi_stream_push(sjmi_stream* s, const uint8_t* chunk, uint64_t len, int is_last, uint32_t* indexes, uint64_t index_capacity,
                     uint64_t* count, uint64_t* base, uint32_t* status) {
    if (!s || (!chunk && len) || !indexes || !count || !base || !status || s->finished) return SJMI_ERR_ARG;
    if (len > s->max_chunk || (!is_last && ((len & 63) || len == 0))) return SJMI_ERR_ARG;
    sjmi_ctx* c = s->c;
    if (fail(c, "hipSetDevice", hipSetDevice(c->device))) return SJMI_ERR_HIP;
    uint8_t* d_chunk = s->d_buf + s->keep;
    if (len && fail(c, "H2D(chunk)", hipMemcpyAsync(d_chunk, chunk, len, hipMemcpyHostToDevice, c->stream))) return SJMI_ERR_HIP;
    const uint64_t dev_cap = s->max_chunk + 66 < index_capacity ? s->max_chunk + 66 : index_capacity;
    sjmi_stage1_result r;
    for (uint64_t want = s->halo;;) {
        const uint64_t h = (want < s->have ? want : s->have) / 64 * 64;
        const bool from_start = h == s->offset;  // the halo reaches back to the stream's first byte
        const int rc = sjmi_stage1_shard_device2(c, d_chunk, len, h, from_start ? 1 : 0, is_last ? 1 : 0, s->parity, s->d_idx, dev_cap, s->d_res,
                                                 nullptr);
        if (rc != SJMI_OK) return rc;
        if (fail(c, "D2H(result)", hipMemcpyAsync(&r, s->d_res, sizeof r, hipMemcpyDeviceToHost, c->stream)) ||
            fail(c, "sync", hipStreamSynchronize(c->stream)))
            return SJMI_ERR_HIP;
        if ((r.status & SJMI_ST_INTERNAL) && !c->ticket_mode) {
            c->ticket_mode = true;  // fast-mode liveness assumption failed: latch the safe mode and run again
            continue;
        }
        if (!(r.status & SJMI_ST_HALO)) break;
        if (h >= s->have / 64 * 64) {  // everything that is kept of the stream is one backslash run
            c->err = "sjmi_stream_push: a backslash run longer than the bytes kept of the stream";
            return SJMI_ERR_CAPACITY;
        }
        want *= 4;
    }
    if (r.status & SJMI_ST_INTERNAL) return SJMI_ERR_INTERNAL;
    if (r.status & SJMI_ST_CAPACITY) return SJMI_ERR_CAPACITY;
    if (r.count + 1 > index_capacity) return SJMI_ERR_CAPACITY;
    if (fail(c, "D2H(indexes)", hipMemcpyAsync(indexes, s->d_idx, (r.count + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream)))
        return SJMI_ERR_HIP;
    // keep the end of the stream so far in front of the next chunk: [keep - have', keep)
    if (!is_last) {
        const uint64_t total = s->have + len, nh = total < s->keep ? total : s->keep;
        // source: the last nh bytes of [keep - have, keep + len); destination: [keep - nh, keep) -- they may overlap: move
        // through the free space behind the chunk when they do (nh <= keep <= max(halo, 4096): small)
        const uint8_t* src = s->d_buf + s->keep + len - nh;
        uint8_t* dst = s->d_buf + s->keep - nh;
        if (src >= dst + nh || src + nh <= dst) {
            if (fail(c, "D2D(keep)", hipMemcpyAsync(dst, src, nh, hipMemcpyDeviceToDevice, c->stream))) return SJMI_ERR_HIP;
        } else {
            uint8_t* bounce = reinterpret_cast<uint8_t*>(s->d_idx + s->max_chunk + 66);  // (`keep` spare bytes behind the index array)
            if (fail(c, "D2D(keep)", hipMemcpyAsync(bounce, src, nh, hipMemcpyDeviceToDevice, c->stream)) ||
                fail(c, "D2D(keep)", hipMemcpyAsync(dst, bounce, nh, hipMemcpyDeviceToDevice, c->stream)))
                return SJMI_ERR_HIP;
        }
        s->have = nh;
    }
    if (fail(c, "sync", hipStreamSynchronize(c->stream))) return SJMI_ERR_HIP;
    *count = r.count;
    *base = s->offset;
    s->parity = (r.status & SJMI_ST_UNCLOSED) ? 1 : 0;  // (a shard's UNCLOSED bit = the parity behind it)
    s->status |= r.status & (SJMI_ST_UTF8 | SJMI_ST_UNESCAPED);
    s->offset += len;
    if (is_last) {
        s->finished = true;
        if (s->parity) s->status |= SJMI_ST_UNCLOSED;  // StructuralIndexer.java:297-299
    }
    *status = s->status;
    c->par_valid = false;
    c->last_valid = false;
    return SJMI_OK;
}

// One rank's shard of a document split over GPUs (sjmi_stage1_shard_device2 with the protocol's bookkeeping): scan as if the
// shard began outside a string; exchange sjmi_split_flips() with the other ranks (one bit per rank: the first collective);
// sjmi_split_resolve(entry parity = XOR of the flips in front) scans again ONLY if that parity is 1; exchange the counts.
struct sjmi_split {
    sjmi_ctx* c = nullptr;
    const void* d_shard = nullptr;
    uint64_t len = 0, halo = 0, index_capacity = 0;
    int halo_from_start = 0, is_last = 0, entry = 0, scans = 0;
    void* d_indexes = nullptr;
    sjmi_stage1_result* d_res = nullptr;
    sjmi_stage1_result r{};
};

int sjmi_split_open(sjmi_ctx* c, const void* d_shard, uint64_t len, uint64_t halo_bytes, int halo_from_document_start, int is_last,
                    void* d_indexes, uint64_t index_capacity, sjmi_split** out) {
    if (!c || !d_shard || !d_indexes || !out) return SJMI_ERR_ARG;
    *out = nullptr;
    sjmi_split* s = new (std::nothrow) sjmi_split();
    if (!s) return SJMI_ERR_ARG;
    if (fail(c, "hipSetDevice", hipSetDevice(c->device)) || fail(c, "hipMalloc(split res)", hipMalloc((void**)&s->d_res, sizeof(sjmi_stage1_result)))) {
        delete s;
        return SJMI_ERR_HIP;
    }
    s->c = c;
    s->d_shard = d_shard;
    s->len = len;
    s->halo = halo_bytes;
    s->halo_from_start = halo_from_document_start;
    s->is_last = is_last;
    s->d_indexes = d_indexes;
    s->index_capacity = index_capacity;
    *out = s;
    return SJMI_OK;
}
void sjmi_split_close(sjmi_split* s) {
    if (!s) return;
    (void)hipSetDevice(s->c->device);
    if (s->d_res) (void)hipFree(s->d_res);
    delete s;
}
static int split_scan(sjmi_split* s, int entry_parity, void* stream) {
    sjmi_ctx* c = s->c;
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    const int rc = sjmi_stage1_shard_device2(c, s->d_shard, s->len, s->halo, s->halo_from_start, s->is_last, entry_parity, s->d_indexes,
                                             s->index_capacity, s->d_res, stream);
    if (rc != SJMI_OK) return rc;
    if (fail(c, "D2H(result)", hipMemcpyAsync(&s->r, s->d_res, sizeof s->r, hipMemcpyDeviceToHost, st)) || fail(c, "sync", hipStreamSynchronize(st)))
        return SJMI_ERR_HIP;
    s->entry = entry_parity;
    ++s->scans;
    if (s->r.status & SJMI_ST_INTERNAL) return SJMI_ERR_INTERNAL;
    if (s->r.status & SJMI_ST_CAPACITY) return SJMI_ERR_CAPACITY;
    return SJMI_OK;
}
int sjmi_split_scan(sjmi_split* s, void* stream, int* flips_parity, uint32_t* status) {
    if (!s || !flips_parity || !status) return SJMI_ERR_ARG;
    const int rc = split_scan(s, 0, stream);
    *flips_parity = (s->r.status & SJMI_ST_UNCLOSED) ? 1 : 0;  // entered outside a string: the parity behind the shard IS the flip
    *status = s->r.status & (SJMI_ST_UTF8 | SJMI_ST_UNESCAPED | SJMI_ST_HALO);
    return rc;
}
int sjmi_split_resolve(sjmi_split* s, int entry_parity, void* stream, uint64_t* count, uint32_t* status, int* parity_after) {
    if (!s || !count || !status || !parity_after || !s->scans) return SJMI_ERR_ARG;
    if ((entry_parity != 0) != (s->entry != 0)) {
        const int rc = split_scan(s, entry_parity ? 1 : 0, stream);
        if (rc != SJMI_OK) return rc;
    }
    *count = s->r.count;
    *status = s->r.status & (SJMI_ST_UTF8 | SJMI_ST_UNESCAPED | SJMI_ST_HALO);
    *parity_after = (s->r.status & SJMI_ST_UNCLOSED) ? 1 : 0;
    return SJMI_OK;
}

int sjmi_selftest(sjmi_ctx* c, uint32_t* mismatches) {
    if (!c || !mismatches) return SJMI_ERR_ARG;
    if (fail(c, "hipSetDevice", hipSetDevice(c->device))) return SJMI_ERR_HIP;
    const uint32_t nblocks = 4096;
    std::vector<uint32_t> words((size_t)nblocks * 16);
    uint64_t x = 0x9E3779B97F4A7C15ull;
    for (size_t i = 0; i < words.size(); ++i) {  // xorshift64*
        x ^= x >> 12;
        x ^= x << 25;
        x ^= x >> 27;
        words[i] = (uint32_t)((x * 0x2545F4914F6CDD1Dull) >> 32);
    }
    for (int i = 0; i < 16; ++i) {  // all-ones / all-zero / single-bit blocks
        words[i] = 0xFFFFFFFFu;
        words[16 + i] = 0;
        words[32 + i] = 0x80808080u;
        words[48 + i] = 0x01010101u;
    }
    uint32_t *d_w = nullptr, *d_m = nullptr;
    int rc = SJMI_OK;
    if (fail(c, "hipMalloc", hipMalloc((void**)&d_w, words.size() * 4)) ||
        fail(c, "hipMalloc", hipMalloc((void**)&d_m, 4)) ||
        fail(c, "H2D", hipMemcpyAsync(d_w, words.data(), words.size() * 4, hipMemcpyHostToDevice, c->stream)) ||
        fail(c, "memset", hipMemsetAsync(d_m, 0, 4, c->stream)) ||
        fail(c, "launch", sjmi::transpose_selftest_launch(d_w, nblocks, d_m, c->stream)) ||
        fail(c, "D2H", hipMemcpyAsync(mismatches, d_m, 4, hipMemcpyDeviceToHost, c->stream)) ||
        fail(c, "sync", hipStreamSynchronize(c->stream)))
        rc = SJMI_ERR_HIP;
    if (d_w) (void)hipFree(d_w);
    if (d_m) (void)hipFree(d_m);
    return rc;
}

}  // extern "C"
