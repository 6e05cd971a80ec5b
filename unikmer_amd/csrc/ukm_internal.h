// ukm_internal.h — shared host-side plumbing of libunikmer_hip.so (context, workspace arena,
// host/device pointer staging, error reporting).  gfx950 only; no portability layer.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/unikmer_hip.h"

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint8_t u8;

// ---- error reporting (thread-local message, never exit) -------------------------------------
void ukm_set_error(const char *fmt, ...);

#define UKM_FAIL(code, ...)        \
    do {                           \
        ukm_set_error(__VA_ARGS__); \
        return (code);             \
    } while (0)

#define UKM_HIP(expr)                                                                    \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            ukm_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                          __LINE__);                                                     \
            return _e == hipErrorOutOfMemory ? UKM_ERR_NOMEM : UKM_ERR_HIP;              \
        }                                                                                \
    } while (0)

#define UKM_TRY(expr)            \
    do {                         \
        int _rc = (expr);        \
        if (_rc != UKM_OK) return _rc; \
    } while (0)

// ---- device taxonomy ------------------------------------------------------------------------
struct TaxDev {
    const u32 *parent;  // dense [size]; 0 = absent; root: parent[r] == r
    const u8 *depth;    // dense [size]; root depth 0
    const u32 *merged;  // dense [size] or nullptr; 0 = not merged
    u32 size;
    // root-path table (round 3): anc[c * size + t] = the ancestors of t at depths 4c .. 4c+3 (t itself at its own
    // depth, 0 beyond it; all 0 for an absent taxid).  Always present when a taxonomy is loaded: ukm_taxonomy_load builds
    // every table or none (it refuses dumps whose dense tables do not fit the device).
    const uint4 *anc;
    u32 nchunks;
    // pre-order numbers (round 3, ukm_pfold.hip): euler[t] = 1 + the position of t (merged ids: of their target) in a
    // depth-first walk of the forest, 0 for taxid 0 / absent / unknown ids; node_at[e] = the taxid with number e.
    // The LCA of a SET of nodes is the LCA of its members with the smallest and the largest number.
    const u32 *euler;
    const u32 *node_at;
    // clade codes (round 5): clade[t] = 1 + the index of t's ancestor at depth min(depth(t), D) among the nodes of depth <= D
    // (D <= 3, the deepest level whose nodes fit 16 bits; 0: absent / merged id, or no table), top[k] = the root path
    // (depths 0..3, as a row of anc[0]) of clade node k.  Two taxids with DIFFERENT codes have the LCA of their clade nodes:
    // two 2-byte reads of a table that mostly sits in L2 and two rows of a table of a few KB settle every pair that
    // diverges within D levels of the root, instead of two random 16-byte rows of the 16 B x ids root-path table.
    const unsigned short *clade;
    const uint4 *top;
    // the same codes in ONE byte per id when the nodes of depth <= 2 (or 1, or 0) are at most 255: a table a quarter the size
    // of a 4-byte column stays in L2 beside the streams (2.4 MB for 2.4 M ids); `clade` is null then
    const unsigned char *clade8;
    // beside clade8: pair[ca * kp + cb] = the LCA of clade nodes ca and cb (0 on the diagonal and in row / column 0: same
    // clade or no code -> the root paths answer); at most 256 x 256 words, one read instead of two rows of `top`
    const u32 *pair;
    u32 kp;
    // behind the pair table, in the same allocation (round 6): what a workgroup copies into LDS so that an unrelated pair
    // costs NO third gather.  cpath[c] = 16 bytes: byte d = the code of clade node c's ancestor at depth d (c itself at its
    // own depth, 0 below it; the first 16 levels of a deeper node), 256 rows; cnode[c] = the taxid of clade node c, 256
    // words.  Two different codes whose rows differ have the LCA cnode[the last byte their rows share] (no shared byte:
    // different trees, 0); rows that agree in all 16 bytes (both nodes 16 or more levels down one chain) ask `pair`.
    const uint4 *cpath;
    const u32 *cnode;
};
constexpr u32 TAX_CPATH_ROWS = 256;

// ---- workspace arena: chunked bump allocator on the ctx's device ------------------------------
struct WsBlock {
    char *base;
    size_t cap;
    size_t used;
};

struct ukm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;  // whole call
    bool ev_valid = false;
    hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr;  // dominant kernel of the call
    bool evk_valid = false;
    // transfer stream (ukm_copy_async): created on first use
    hipStream_t xfer = nullptr;
    hipEvent_t ev_xfer = nullptr, ev_comp = nullptr;
    bool xfer_pending = false;
    // two more streams for launches that do not depend on each other (the bucket sort's size classes): forked from and
    // joined to `stream` by events inside one call
    hipStream_t side[2] = {nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_side[2] = {nullptr, nullptr};
    // RCCL communicator of the multi-GPU exchange (ukm_comm.hip); nullptr until ukm_comm_init
    void *comm = nullptr;
    int comm_size = 0, comm_rank = 0;
    int depth = 0;  // nesting depth of API calls (n-way ops call 2-way ops)
    int last_route = 0;  // ukm_last_route(): which n-way route answered last

    std::vector<WsBlock> blocks;
    size_t ws_high = 0;  // high-water mark of one top-level call, for consolidation

    // pending host copy-backs of the current top-level call
    struct CopyBack {
        void *host;
        const void *dev;
        size_t bytes;
    };
    std::vector<CopyBack> copybacks;

    // taxonomy (device arrays owned by the ctx)
    u32 *tax_parent = nullptr;
    u8 *tax_depth = nullptr;
    u32 *tax_merged = nullptr;
    uint4 *tax_anc = nullptr;  // root-path table, see TaxDev
    u32 *tax_euler = nullptr, *tax_node_at = nullptr;  // pre-order numbers, see TaxDev
    unsigned short *tax_clade = nullptr;                // clade codes, see TaxDev (tax_clade8: the one-byte form)
    unsigned char *tax_clade8 = nullptr;
    u32 *tax_pair = nullptr;
    u32 tax_kp = 0;
    uint4 *tax_top = nullptr;
    u32 tax_top_n = 0;  // entries of tax_top
    u32 tax_nchunks = 0;
    u32 tax_size = 0;
    u32 tax_max = 0;

    // pinned host scratch for small D2H results
    u64 *h_scratch = nullptr;  // 64 x u64

    int num_cu = 256;

    // Route policy and developer knobs (round 5): the UKM_* variables of the environment are read ONCE, when the context is
    // created (knobs), and ukm_ctx_set_option overrides them per context (opts: "punion" <-> UKM_PUNION ...).  No compute
    // call reads the environment -- unless the context was created under UKM_ENV_LIVE=1 (the test suite, whose cases flip
    // knobs between calls on one context).
    std::map<std::string, std::string> knobs, opts;
    bool env_live = false;
    // statistics a host or a test may ask for (ukm_ctx_get_stat)
    u64 stat_punion_attempts = 0;  // base sets the last probe union / counting probes built (2: the retry with 4 x the files ran)

    // set once the blockIdx-ordered set-op kernel hit its watchdog on this device
    bool setop_force_ticket = false;  // = ticket_latched || option "force_ticket"
    unsigned long long stat_sort_fused_hist = 0;  // sorts of this context that took their first histogram from the producer of the keys (ukm_count)
    bool ticket_latched = false;      // the look-back watchdog fired on this device (ukm_switch_to_tickets): stays set
    // set once a sort found its keys crowded into few top-16-bit buckets (ukm_sort.hip): later sorts look at a sample first
    bool sort_skew_seen = false;
    // buckets of the last bucket-route sort that fell back from the counting step to the digit passes (device word, bumped by
    // ls_sort_kernel, fetched by the NEXT sort's classify kernel so that it rides on that call's read-back): keys with many
    // copies each (k-mers of reads at coverage) make every bucket fall back, and the next calls skip the attempt
    u64 *sort_stat_dev = nullptr;
    bool sort_last_counting = false;
    u64 sort_last_buckets = 0;
    int sort_counting_skip = 0;
    // set around the sort of the gathered oversized buckets: those keys are crowded by construction, the general passes take them
    bool sort_general_only = false;
};

// The value of knob UKM_<NAME> for this context (see ukm_ctx::knobs): nullptr when it is not set.  The pointer stays valid
// until the next ukm_ctx_set_option on the context.
const char *ukm_env(const ukm_ctx *c, const char *name);
static inline bool ukm_env_is(const ukm_ctx *c, const char *name, char v) {
    const char *e = ukm_env(c, name);
    return e && e[0] == v;
}
static inline int ukm_env_int(const ukm_ctx *c, const char *name, int unset) {
    const char *e = ukm_env(c, name);
    return (e && *e) ? atoi(e) : unset;
}

// Arena API.  Pointers stay valid until the enclosing top-level call returns.
int ws_alloc(ukm_ctx *c, size_t bytes, void **out);
template <typename T>
static inline int ws_alloc_t(ukm_ctx *c, size_t n, T **out) {
    void *p = nullptr;
    int rc = ws_alloc(c, n * sizeof(T), &p);
    *out = (T *)p;
    return rc;
}
struct WsMark {
    size_t nblocks;
    size_t used_last;
};
WsMark ws_mark(ukm_ctx *c);
void ws_release(ukm_ctx *c, WsMark m);  // frees allocations made after the mark (LIFO)

// true if p is a device pointer usable by kernels on this ctx's device
bool ukm_is_device_ptr(const void *p);

// Input staging: returns p itself for device pointers, else an arena copy (H2D on the stream).
int ukm_in(ukm_ctx *c, const void *p, size_t bytes, const void **dev);
template <typename T>
static inline int ukm_in_t(ukm_ctx *c, const T *p, size_t n, const T **dev) {
    const void *d = nullptr;
    if (p == nullptr || n == 0) {
        *dev = p;
        if (p == nullptr) return UKM_OK;
    }
    int rc = ukm_in(c, p, n * sizeof(T), &d);
    *dev = (const T *)d;
    return rc;
}
// Output staging: device pointers pass through; host pointers get an arena buffer whose
// first `bytes` are copied back by ukm_finish() (use ukm_out_resize to shrink the copy).
int ukm_out(ukm_ctx *c, void *p, size_t bytes, void **dev);
template <typename T>
static inline int ukm_out_t(ukm_ctx *c, T *p, size_t n, T **dev) {
    void *d = nullptr;
    if (p == nullptr) {
        *dev = nullptr;
        return UKM_OK;
    }
    int rc = ukm_out(c, p, n * sizeof(T), &d);
    *dev = (T *)d;
    return rc;
}
void ukm_out_resize(ukm_ctx *c, void *host, size_t bytes);
// In-place staging (sort): host array copied in, and copied back at finish.
int ukm_inout(ukm_ctx *c, void *p, size_t bytes, void **dev);

// Call bracket.  ukm_begin at the top of every compute entry; ukm_finish before returning:
// performs copy-backs, stream sync, arena reset (top level only).  Nested calls are cheap.
struct CallScope {
    ukm_ctx *c;
    bool top;
    WsMark mark;
};
int ukm_begin(ukm_ctx *c, CallScope *s);
int ukm_finish(CallScope *s, int rc);

// read one u64 (or several) from device memory to host, synchronising the stream
int ukm_read_u64(ukm_ctx *c, const u64 *dev, u64 *host, int n = 1);

static inline TaxDev ukm_taxdev(const ukm_ctx *c) {
    TaxDev t;
    t.parent = c->tax_parent;
    t.depth = c->tax_depth;
    t.merged = c->tax_merged;
    t.anc = c->tax_anc;
    t.euler = c->tax_euler;
    t.node_at = c->tax_node_at;
    t.clade = c->tax_clade;
    t.clade8 = c->tax_clade8;
    t.pair = c->tax_pair;
    t.kp = c->tax_kp;
    {   // [pair kp x kp | pad to 16 bytes | cpath 256 x 16 B | cnode 256 x 4 B]
        const size_t off = (((size_t)c->tax_kp * c->tax_kp) + 3) & ~(size_t)3;
        t.cpath = c->tax_pair ? reinterpret_cast<const uint4 *>(c->tax_pair + off) : nullptr;
        t.cnode = c->tax_pair ? c->tax_pair + off + 4 * TAX_CPATH_ROWS : nullptr;
    }
    t.top = c->tax_top;
    t.nchunks = c->tax_nchunks;
    t.size = c->tax_size;
    return t;
}

// ---- internal device-pointer entry points (all pointers are device pointers) ------------------
// internal fourth 2-way operation: merge keeping every record (k-way merge tree of ukm_merge_k)
#define UKM_OP_MERGE_INTERNAL 3
// a look-back watchdog fired in a blockIdx-ordered kernel: from now on this context uses the ticketed
// instantiations (one warning on stderr per context, so that an operator sees the slower mode)
void ukm_switch_to_tickets(ukm_ctx *c, const char *where);
int ukm_dev_setop2(ukm_ctx *c, int op, const u64 *a, const u32 *ta, u64 na, const u64 *b,
                   const u32 *tb, u64 nb, u32 flags, u64 *out, u32 *tout, u64 out_cap,
                   u64 *n_out);
// (cta, ctb): the taxid of every record of a stream whose per-record pointer is null -- the file's global taxid
// (round 5; 0 = none).  Both constant: the plain-key kernel with a taxid epilogue.
int ukm_dev_setop2_ct(ukm_ctx *c, int op, const u64 *a, const u32 *ta, u32 cta, u64 na, const u64 *b,
                      const u32 *tb, u32 ctb, u64 nb, u32 flags, u64 *out, u32 *tout, u64 out_cap,
                      u64 *n_out);
int ukm_dev_setop2_link(ukm_ctx *c, int op, const u64 *a, const u32 *ta, u64 na_max, const u64 *na_dev, const u64 *b,
                        const u32 *tb, u64 nb, u32 flags, u64 *out, u32 *tout, u64 out_cap, u64 *ctl, u32 cta = 0, u32 ctb = 0);
// fill n words with one value on the context's stream (file taxids; ukm_scan.hip)
int ukm_dev_fill_u32(ukm_ctx *c, u32 *dst, u64 n, u32 value);
// decisions over per-file taxids on the device (ukm_tax.hip): plan[0] = inter's left LCA fold, plan[2 + j] = diff -t keeps
int ukm_dev_ct_plan(ukm_ctx *c, const u32 *ct_host, int n, bool mix, u32 **plan);
int ukm_dev_fill_u32_from(ukm_ctx *c, u32 *dst, u64 n, u32 value, const u32 *value_dev);  // value_dev != null: the value is read there
// internal flag of ukm_dev_setop2 / _ct (op DIFF, first stream with duplicate codes): the survivors are NOT collapsed to one
// record per code -- the caller is in the middle of an n-file fold (diff.go:437: mc1 = mc2 keeps every record; the map of
// diff.go:449-453 only shapes the final result) and collapses once at the end
#define UKM_F_INTERNAL_KEEP_DUPS 0x10000u
// flag bits of the set-op result word [1]
enum { UKM_SETOP_FLAG_DUP = 1, UKM_SETOP_FLAG_UNSORTED = 2, UKM_SETOP_FLAG_TIMEOUT = 4 };
int ukm_dev_sort(ukm_ctx *c, u64 *keys, u32 *vals, u64 n, int key_bits);
// the same with the first scatter pass's digit histogram already counted by the producer of the keys (ukm_count)
int ukm_dev_sort_hist(ukm_ctx *c, u64 *keys, u32 *vals, u64 n, int key_bits, const u64 *first_hist, int first_shift);
int ukm_sort_first_shift(const ukm_ctx *c, u64 n, int key_bits);
int ukm_dev_unique(ukm_ctx *c, const u64 *keys, const u32 *taxids, u64 n, int mode, u64 *out,
                   u32 *tout, u64 out_cap, u64 *n_out);
// mode 5 = UNIQUE_LAST (last record of each run), 6 = COMMON (run length >= threshold)
int ukm_dev_unique_ex(ukm_ctx *c, const u64 *keys, const u32 *taxids, u64 n, int mode, u32 threshold,
                      u64 *out, u32 *tout, u64 out_cap, u64 *n_out);
int ukm_dev_check_sorted(ukm_ctx *c, const u64 *keys, u64 n, bool *sorted, bool *strict);
int ukm_dev_exclusive_scan_u64(ukm_ctx *c, const u64 *in, u64 *out, u64 n, u64 *total_dev);
