// ukm_ctx.hip — context, workspace arena, pointer staging and error plumbing of
// libunikmer_hip.so.  Boundary conventions: include/unikmer_hip.h.
#include <stdarg.h>
#include <stdlib.h>

#include <algorithm>

#include "ukm_internal.h"

static thread_local char g_err[512] = "";

void ukm_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *ukm_last_error(void) { return g_err; }
extern "C" int ukm_version(void) { return 1000 * 0 + 1; }

extern "C" int ukm_device_count(int *n) {
    if (!n) UKM_FAIL(UKM_ERR_INVALID, "ukm_device_count: n is NULL");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        c = 0;
    }
    *n = c;
    return UKM_OK;
}

// ---- knobs and options ------------------------------------------------------------------------------------------------
extern char **environ;

// the option keys a host may set (ukm_ctx_set_option): key "punion" is knob UKM_PUNION, and so on
static const char *const UKM_OPTION_KEYS[] = {
    "punion", "punion_tax", "punion_ranked", "place", "srmerge", "kway", "no_kway", "no_fold", "no_pfold", "pfold_tax", "common_probe",
    "sort_local", "sort_counting", "sort_fan", "win_strip", "nthash_strip", "force_ticket", "setop_src", "setop_defer", "punion_clade", "srmerge_clade",
    // tuning / diagnostics (developer)
    "punion_k0", "punion_claim", "punion_debug", "kway_k", "kway_r", "kway_top2", "kway_debug", "srmerge_fill", "srmerge_spr", "srmerge_buckets",
    "srmerge_debug", "fold_debug", "sort_debug", "strip_l", "win_strip_l", "setop_fused_part", "setop_fix",
};

static std::string knob_name(const char *key) {
    std::string n = "UKM_";
    for (const char *p = key; *p; p++) n.push_back((char)toupper((unsigned char)*p));
    return n;
}

const char *ukm_env(const ukm_ctx *c, const char *name) {
    if (c) {
        auto o = c->opts.find(name);
        if (o != c->opts.end()) return o->second.c_str();
        if (!c->env_live) {
            auto k = c->knobs.find(name);
            return k == c->knobs.end() ? nullptr : k->second.c_str();
        }
    }
    return getenv(name);  // (no context, or a context created under UKM_ENV_LIVE=1: the test suite)
}

extern "C" int ukm_ctx_set_option(ukm_ctx *c, const char *key, long long value) {
    if (!c || !key) UKM_FAIL(UKM_ERR_INVALID, "ukm_ctx_set_option: NULL argument");
    for (const char *k : UKM_OPTION_KEYS)
        if (strcmp(k, key) == 0) {
            c->opts[knob_name(key)] = std::to_string(value);
            // (the watchdog's latch is a fact about the device, not an option: clearing the option never clears it)
            if (strcmp(key, "force_ticket") == 0) c->setop_force_ticket = c->ticket_latched || value != 0;
            return UKM_OK;
        }
    UKM_FAIL(UKM_ERR_INVALID, "ukm_ctx_set_option: unknown option '%s'", key);
}

extern "C" int ukm_ctx_unset_option(ukm_ctx *c, const char *key) {
    if (!c || !key) UKM_FAIL(UKM_ERR_INVALID, "ukm_ctx_unset_option: NULL argument");
    c->opts.erase(knob_name(key));
    if (strcmp(key, "force_ticket") == 0) c->setop_force_ticket = c->ticket_latched || ukm_env_is(c, "UKM_FORCE_TICKET", '1');
    return UKM_OK;
}

extern "C" int ukm_ctx_get_option(ukm_ctx *c, const char *key, long long *value, int *is_set) {
    if (!c || !key || !value || !is_set) UKM_FAIL(UKM_ERR_INVALID, "ukm_ctx_get_option: NULL argument");
    const std::string n = knob_name(key);
    const char *e = ukm_env(c, n.c_str());
    *is_set = (e && *e) ? 1 : 0;
    *value = *is_set ? atoll(e) : 0;
    return UKM_OK;
}

extern "C" int ukm_ctx_get_stat(ukm_ctx *c, const char *key, unsigned long long *value) {
    if (!c || !key || !value) UKM_FAIL(UKM_ERR_INVALID, "ukm_ctx_get_stat: NULL argument");
    if (strcmp(key, "punion_attempts") == 0) *value = c->stat_punion_attempts;
    else if (strcmp(key, "sort_fused_hist") == 0) *value = c->stat_sort_fused_hist;
    else if (strcmp(key, "workspace_bytes") == 0) {
        u64 t = 0;
        for (auto &b : c->blocks) t += b.cap;
        *value = t;
    } else UKM_FAIL(UKM_ERR_INVALID, "ukm_ctx_get_stat: unknown statistic '%s'", key);
    return UKM_OK;
}

extern "C" int ukm_ctx_create(int device, ukm_ctx **out) {
    if (!out) UKM_FAIL(UKM_ERR_INVALID, "ukm_ctx_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
        (void)hipGetLastError();
        UKM_FAIL(UKM_ERR_HIP, "ukm_ctx_create: no HIP device available (%s)",
                 e == hipSuccess ? "count=0" : hipGetErrorString(e));
    }
    if (device < 0 || device >= n)
        UKM_FAIL(UKM_ERR_INVALID, "ukm_ctx_create: device %d out of range [0,%d)", device, n);
    UKM_HIP(hipSetDevice(device));
    ukm_ctx *c = new ukm_ctx();
    c->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->num_cu = prop.multiProcessorCount;
    hipError_t e1 = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    hipError_t e2 = hipEventCreate(&c->ev_start);
    hipError_t e3 = hipEventCreate(&c->ev_stop);
    if (e3 == hipSuccess) e3 = hipEventCreate(&c->ev_k0);
    if (e3 == hipSuccess) e3 = hipEventCreate(&c->ev_k1);
    hipError_t e4 = hipHostMalloc((void **)&c->h_scratch, 64 * sizeof(u64), hipHostMallocDefault);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) {
        delete c;
        UKM_FAIL(UKM_ERR_HIP, "ukm_ctx_create: stream/event/pinned allocation failed");
    }
    c->own_stream = true;
    // the knobs of this context: every UKM_* variable as it is NOW (no later call looks at the environment)
    for (char **e = environ; e && *e; e++) {
        if (strncmp(*e, "UKM_", 4) != 0) continue;
        const char *eq = strchr(*e, '=');
        if (eq) c->knobs[std::string(*e, (size_t)(eq - *e))] = std::string(eq + 1);
    }
    c->env_live = ukm_env_is(c, "UKM_ENV_LIVE", '1');
    // developer/test knob: exercise the ticketed (dispatch-order independent) set-op kernel
    c->setop_force_ticket = ukm_env_is(c, "UKM_FORCE_TICKET", '1');
    *out = c;
    return UKM_OK;
}

static void ws_free_all(ukm_ctx *c) {
    for (auto &b : c->blocks) (void)hipFree(b.base);
    c->blocks.clear();
}

extern "C" int ukm_ctx_destroy(ukm_ctx *c) {
    if (!c) return UKM_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)ukm_comm_destroy(c);
    ws_free_all(c);
    if (c->tax_parent) (void)hipFree(c->tax_parent);
    if (c->tax_depth) (void)hipFree(c->tax_depth);
    if (c->tax_merged) (void)hipFree(c->tax_merged);
    if (c->tax_anc) (void)hipFree(c->tax_anc);
    if (c->tax_euler) (void)hipFree(c->tax_euler);
    if (c->tax_node_at) (void)hipFree(c->tax_node_at);
    if (c->tax_clade) (void)hipFree(c->tax_clade);
    if (c->tax_clade8) (void)hipFree(c->tax_clade8);
    if (c->tax_pair) (void)hipFree(c->tax_pair);
    if (c->tax_top) (void)hipFree(c->tax_top);
    if (c->h_scratch) (void)hipHostFree(c->h_scratch);
    if (c->sort_stat_dev) (void)hipFree(c->sort_stat_dev);
    if (c->ev_start) (void)hipEventDestroy(c->ev_start);
    if (c->ev_stop) (void)hipEventDestroy(c->ev_stop);
    if (c->ev_k0) (void)hipEventDestroy(c->ev_k0);
    if (c->ev_k1) (void)hipEventDestroy(c->ev_k1);
    if (c->xfer) {
        (void)hipStreamSynchronize(c->xfer);
        (void)hipStreamDestroy(c->xfer);
    }
    if (c->ev_xfer) (void)hipEventDestroy(c->ev_xfer);
    for (int i = 0; i < 2; i++) {
        if (c->side[i]) { (void)hipStreamSynchronize(c->side[i]); (void)hipStreamDestroy(c->side[i]); }
        if (c->ev_side[i]) (void)hipEventDestroy(c->ev_side[i]);
    }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_comp) (void)hipEventDestroy(c->ev_comp);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return UKM_OK;
}

extern "C" int ukm_ctx_set_stream(ukm_ctx *c, void *hip_stream) {
    if (!c) UKM_FAIL(UKM_ERR_INVALID, "ukm_ctx_set_stream: ctx is NULL");
    UKM_HIP(hipSetDevice(c->device));
    UKM_HIP(hipStreamSynchronize(c->stream));
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    // NULL is HIP's default (null) stream of the device — a valid stream to borrow
    c->stream = (hipStream_t)hip_stream;
    c->own_stream = false;
    return UKM_OK;
}

extern "C" int ukm_ctx_sync(ukm_ctx *c) {
    if (!c) UKM_FAIL(UKM_ERR_INVALID, "ukm_ctx_sync: ctx is NULL");
    UKM_HIP(hipSetDevice(c->device));
    UKM_HIP(hipStreamSynchronize(c->stream));
    return UKM_OK;
}

// ---- arena ---------------------------------------------------------------------------------
static const size_t WS_ALIGN = 256;
static const size_t WS_MIN_BLOCK = (size_t)64 << 20;

static int ws_new_block(ukm_ctx *c, size_t need) {
    size_t total = 0;
    for (auto &b : c->blocks) total += b.cap;
    size_t cap = std::max(need, std::max(WS_MIN_BLOCK, total));  // at least double
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, cap);
    if (e != hipSuccess && cap > need) {
        (void)hipGetLastError();
        cap = need;
        e = hipMalloc(&p, cap);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        UKM_FAIL(UKM_ERR_NOMEM, "device workspace allocation of %zu bytes failed: %s", cap,
                 hipGetErrorString(e));
    }
    c->blocks.push_back(WsBlock{(char *)p, cap, 0});
    return UKM_OK;
}

int ws_alloc(ukm_ctx *c, size_t bytes, void **out) {
    bytes = (bytes + WS_ALIGN - 1) / WS_ALIGN * WS_ALIGN;
    if (bytes == 0) bytes = WS_ALIGN;
    if (c->blocks.empty() || c->blocks.back().cap - c->blocks.back().used < bytes)
        UKM_TRY(ws_new_block(c, bytes));
    WsBlock &b = c->blocks.back();
    *out = b.base + b.used;
    b.used += bytes;
    size_t used = 0;
    for (auto &x : c->blocks) used += x.used;
    c->ws_high = std::max(c->ws_high, used);
    return UKM_OK;
}

WsMark ws_mark(ukm_ctx *c) {
    WsMark m;
    m.nblocks = c->blocks.size();
    m.used_last = c->blocks.empty() ? 0 : c->blocks.back().used;
    return m;
}

void ws_release(ukm_ctx *c, WsMark m) {
    // blocks created after the mark stay allocated (they are reused) but become empty
    for (size_t i = m.nblocks; i < c->blocks.size(); i++) c->blocks[i].used = 0;
    if (m.nblocks > 0) c->blocks[m.nblocks - 1].used = m.used_last;
    // keep allocation order valid: the "current" block is always blocks.back(); after a
    // release the tail blocks are empty, so move the largest empty tail block to the position
    // right after the mark to be used next.
    if (c->blocks.size() > m.nblocks + 1) {
        auto it = std::max_element(c->blocks.begin() + m.nblocks, c->blocks.end(),
                                   [](const WsBlock &a, const WsBlock &b) { return a.cap < b.cap; });
        std::iter_swap(it, c->blocks.end() - 1);
    }
}

static int ws_reset_top(ukm_ctx *c) {
    // consolidate into one block sized to the high-water mark so the next call does not grow
    if (c->blocks.size() > 1) {
        size_t want = c->ws_high + (c->ws_high >> 3) + WS_ALIGN * 64;
        ws_free_all(c);
        void *p = nullptr;
        if (hipMalloc(&p, want) == hipSuccess)
            c->blocks.push_back(WsBlock{(char *)p, want, 0});
        else
            (void)hipGetLastError();
    }
    for (auto &b : c->blocks) b.used = 0;
    c->ws_high = 0;
    return UKM_OK;
}

extern "C" int ukm_ctx_trim(ukm_ctx *c) {
    if (!c) UKM_FAIL(UKM_ERR_INVALID, "ukm_ctx_trim: ctx is NULL");
    UKM_HIP(hipSetDevice(c->device));
    UKM_HIP(hipStreamSynchronize(c->stream));
    ws_free_all(c);
    c->ws_high = 0;
    return UKM_OK;
}

extern "C" int ukm_ctx_reserve(ukm_ctx *c, uint64_t bytes) {
    if (!c) UKM_FAIL(UKM_ERR_INVALID, "ukm_ctx_reserve: ctx is NULL");
    UKM_HIP(hipSetDevice(c->device));
    UKM_HIP(hipStreamSynchronize(c->stream));
    size_t total = 0;
    for (auto &b : c->blocks) total = std::max(total, b.cap);
    if (total >= bytes) return UKM_OK;
    ws_free_all(c);
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        UKM_FAIL(UKM_ERR_NOMEM, "ukm_ctx_reserve: %llu bytes: %s", (unsigned long long)bytes,
                 hipGetErrorString(e));
    }
    c->blocks.push_back(WsBlock{(char *)p, (size_t)bytes, 0});
    return UKM_OK;
}

// ---- pointer classification + staging --------------------------------------------------------
bool ukm_is_device_ptr(const void *p) {
    if (!p) return false;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // unregistered host memory
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

int ukm_in(ukm_ctx *c, const void *p, size_t bytes, const void **dev) {
    if (ukm_is_device_ptr(p)) {
        *dev = p;
        return UKM_OK;
    }
    void *d = nullptr;
    UKM_TRY(ws_alloc(c, bytes, &d));
    if (bytes) UKM_HIP(hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, c->stream));
    *dev = d;
    return UKM_OK;
}

int ukm_out(ukm_ctx *c, void *p, size_t bytes, void **dev) {
    if (ukm_is_device_ptr(p)) {
        *dev = p;
        return UKM_OK;
    }
    void *d = nullptr;
    UKM_TRY(ws_alloc(c, bytes, &d));
    c->copybacks.push_back(ukm_ctx::CopyBack{p, d, bytes});
    *dev = d;
    return UKM_OK;
}

void ukm_out_resize(ukm_ctx *c, void *host, size_t bytes) {
    for (auto &cb : c->copybacks)
        if (cb.host == host) cb.bytes = std::min(cb.bytes, bytes);
}

int ukm_inout(ukm_ctx *c, void *p, size_t bytes, void **dev) {
    if (ukm_is_device_ptr(p)) {
        *dev = p;
        return UKM_OK;
    }
    void *d = nullptr;
    UKM_TRY(ws_alloc(c, bytes, &d));
    if (bytes) UKM_HIP(hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, c->stream));
    c->copybacks.push_back(ukm_ctx::CopyBack{p, d, bytes});
    *dev = d;
    return UKM_OK;
}

int ukm_begin(ukm_ctx *c, CallScope *s) {
    if (!c) UKM_FAIL(UKM_ERR_INVALID, "ctx is NULL");
    s->c = c;
    s->top = (c->depth == 0);
    c->depth++;
    if (s->top) {
        hipError_t e = hipSetDevice(c->device);
        if (e != hipSuccess) {
            c->depth--;
            UKM_FAIL(UKM_ERR_HIP, "hipSetDevice(%d): %s", c->device, hipGetErrorString(e));
        }
        c->copybacks.clear();
        (void)hipEventRecord(c->ev_start, c->stream);
        c->ev_valid = false;
        c->evk_valid = false;
    }
    s->mark = ws_mark(c);
    return UKM_OK;
}

int ukm_finish(CallScope *s, int rc) {
    ukm_ctx *c = s->c;
    c->depth--;
    if (!s->top) {
        ws_release(c, s->mark);
        return rc;
    }
    (void)hipEventRecord(c->ev_stop, c->stream);
    c->ev_valid = true;
    if (rc == UKM_OK) {
        for (auto &cb : c->copybacks) {
            if (cb.bytes == 0) continue;
            hipError_t e = hipMemcpyAsync(cb.host, cb.dev, cb.bytes, hipMemcpyDeviceToHost, c->stream);
            if (e != hipSuccess) {
                ukm_set_error("copy-back failed: %s", hipGetErrorString(e));
                rc = UKM_ERR_HIP;
                break;
            }
        }
    }
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess && rc == UKM_OK) {
        ukm_set_error("stream synchronize failed: %s", hipGetErrorString(e));
        rc = UKM_ERR_HIP;
    }
    c->copybacks.clear();
    ws_reset_top(c);
    return rc;
}

void ukm_switch_to_tickets(ukm_ctx *c, const char *where) {
    if (!c->setop_force_ticket)
        fprintf(stderr, "[unikmer_hip] look-back watchdog fired in %s: workgroups were not dispatched in order on device %d; "
                        "switching this context to ticketed tile ids\n", where, c->device);
    c->ticket_latched = true;
    c->setop_force_ticket = true;
}

int ukm_read_u64(ukm_ctx *c, const u64 *dev, u64 *host, int n) {
    if (n > 64) UKM_FAIL(UKM_ERR_INVALID, "ukm_read_u64: n too large");
    UKM_HIP(hipMemcpyAsync(c->h_scratch, dev, n * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    UKM_HIP(hipStreamSynchronize(c->stream));
    for (int i = 0; i < n; i++) host[i] = c->h_scratch[i];
    return UKM_OK;
}

extern "C" int ukm_last_call_ms(ukm_ctx *c, float *ms) {
    if (!c || !ms) UKM_FAIL(UKM_ERR_INVALID, "ukm_last_call_ms: NULL argument");
    if (!c->ev_valid) UKM_FAIL(UKM_ERR_INVALID, "ukm_last_call_ms: no completed call");
    UKM_HIP(hipEventSynchronize(c->ev_stop));
    UKM_HIP(hipEventElapsedTime(ms, c->ev_start, c->ev_stop));
    return UKM_OK;
}

extern "C" int ukm_last_route(ukm_ctx *c) { return c ? c->last_route : 0; }

extern "C" int ukm_last_kernel_ms(ukm_ctx *c, float *ms) {
    if (!c || !ms) UKM_FAIL(UKM_ERR_INVALID, "ukm_last_kernel_ms: NULL argument");
    if (!c->evk_valid) return ukm_last_call_ms(c, ms);
    UKM_HIP(hipEventSynchronize(c->ev_k1));
    UKM_HIP(hipEventElapsedTime(ms, c->ev_k0, c->ev_k1));
    return UKM_OK;
}

// ---- device memory helpers --------------------------------------------------------------------
extern "C" int ukm_dev_alloc(ukm_ctx *c, uint64_t bytes, void **dptr) {
    if (!c || !dptr) UKM_FAIL(UKM_ERR_INVALID, "ukm_dev_alloc: NULL argument");
    UKM_HIP(hipSetDevice(c->device));
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        UKM_FAIL(UKM_ERR_NOMEM, "ukm_dev_alloc(%llu): %s", (unsigned long long)bytes,
                 hipGetErrorString(e));
    }
    *dptr = p;
    return UKM_OK;
}

extern "C" int ukm_dev_free(ukm_ctx *c, void *dptr) {
    if (!c) UKM_FAIL(UKM_ERR_INVALID, "ukm_dev_free: ctx is NULL");
    if (!dptr) return UKM_OK;
    UKM_HIP(hipSetDevice(c->device));
    UKM_HIP(hipStreamSynchronize(c->stream));
    UKM_HIP(hipFree(dptr));
    return UKM_OK;
}

// ---- pinned host memory + asynchronous transfers on a second stream -------------------------------------------
extern "C" int ukm_host_alloc(ukm_ctx *c, uint64_t bytes, void **hptr) {
    if (!c || !hptr) UKM_FAIL(UKM_ERR_INVALID, "ukm_host_alloc: NULL argument");
    UKM_HIP(hipSetDevice(c->device));
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        UKM_FAIL(UKM_ERR_NOMEM, "ukm_host_alloc(%llu): %s", (unsigned long long)bytes, hipGetErrorString(e));
    }
    *hptr = p;
    return UKM_OK;
}

extern "C" int ukm_host_free(ukm_ctx *c, void *hptr) {
    if (!c) UKM_FAIL(UKM_ERR_INVALID, "ukm_host_free: ctx is NULL");
    if (!hptr) return UKM_OK;
    UKM_HIP(hipSetDevice(c->device));
    if (c->xfer) UKM_HIP(hipStreamSynchronize(c->xfer));
    UKM_HIP(hipHostFree(hptr));
    return UKM_OK;
}

static int xfer_init(ukm_ctx *c) {
    if (c->xfer) return UKM_OK;
    UKM_HIP(hipStreamCreateWithFlags(&c->xfer, hipStreamNonBlocking));
    UKM_HIP(hipEventCreateWithFlags(&c->ev_xfer, hipEventDisableTiming));
    UKM_HIP(hipEventCreateWithFlags(&c->ev_comp, hipEventDisableTiming));
    return UKM_OK;
}

extern "C" int ukm_copy_async(ukm_ctx *c, void *dst, const void *src, uint64_t bytes) {
    if (!c || (!dst && bytes) || (!src && bytes)) UKM_FAIL(UKM_ERR_INVALID, "ukm_copy_async: NULL argument");
    if (bytes == 0) return UKM_OK;
    UKM_HIP(hipSetDevice(c->device));
    UKM_TRY(xfer_init(c));
    // the transfer starts after everything the compute stream has been given so far (it may read a result,
    // or overwrite a buffer an earlier kernel still reads) ...
    UKM_HIP(hipEventRecord(c->ev_comp, c->stream));
    UKM_HIP(hipStreamWaitEvent(c->xfer, c->ev_comp, 0));
    UKM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, c->xfer));
    c->xfer_pending = true;
    return UKM_OK;
}

extern "C" int ukm_copy_fence(ukm_ctx *c) {
    if (!c) UKM_FAIL(UKM_ERR_INVALID, "ukm_copy_fence: ctx is NULL");
    if (!c->xfer || !c->xfer_pending) return UKM_OK;
    UKM_HIP(hipSetDevice(c->device));
    // ... and compute calls issued after this fence start after the transfers issued before it
    UKM_HIP(hipEventRecord(c->ev_xfer, c->xfer));
    UKM_HIP(hipStreamWaitEvent(c->stream, c->ev_xfer, 0));
    return UKM_OK;
}

extern "C" int ukm_copy_sync(ukm_ctx *c) {
    if (!c) UKM_FAIL(UKM_ERR_INVALID, "ukm_copy_sync: ctx is NULL");
    if (!c->xfer) return UKM_OK;
    UKM_HIP(hipSetDevice(c->device));
    UKM_HIP(hipStreamSynchronize(c->xfer));
    c->xfer_pending = false;
    return UKM_OK;
}

extern "C" int ukm_copy(ukm_ctx *c, void *dst, const void *src, uint64_t bytes) {
    if (!c || (!dst && bytes) || (!src && bytes)) UKM_FAIL(UKM_ERR_INVALID, "ukm_copy: NULL argument");
    if (bytes == 0) return UKM_OK;
    UKM_HIP(hipSetDevice(c->device));
    UKM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, c->stream));
    UKM_HIP(hipStreamSynchronize(c->stream));
    return UKM_OK;
}
