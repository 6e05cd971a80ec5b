// ukm_nway.hip — n-way set operations and the k-way merge, folded from the 2-way device path
// (ukm_setops.hip), the radix sort (ukm_sort.hip) and the sorted-stream scans (ukm_scan.hip).
// Replaces the per-file loops of
//   union  : union.go:126-305   -> pairwise merge TREE of unions (LCA is associative and
//                                  commutative on a tree, so the tree equals the reference's
//                                  arrival-order left fold on the sorted output stream)
//   inter  : inter.go:175-286   -> the same running intersection, one 2-way kernel per file,
//                                  with the reference's early exits
//   diff   : diff.go:280-523    -> sequential subtraction (= one worker, `-j 1`)
//   common : common.go:205-344  -> concatenate, stable radix sort, run-length threshold scan
//   merge  : util-sort.go:227-606 (mergeChunksFile) -> concatenate sorted chunks, stable radix
//                                  sort, unique/repeated scan (a heap has no place on a GPU)
// All of it is host-side orchestration; every byte of data stays on the device.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "ukm_device.h"
#include "ukm_fold.h"
#include "ukm_kway.h"
#include "ukm_punion.h"
#include "ukm_pfold.h"
#include "ukm_srmerge.h"

namespace {

// A stream's taxids: per record (t), or ONE value for the whole file (ct, used when t is null: the .unik header's global
// taxid, which unik.Reader hands out with every record -- `count -t`, count.go:466-468; union.go:187-201 folds it like any
// other).  ct = 0 with t null: the stream has no taxid information (taxid 0, mix-taxid).
struct Stream {
    const u64 *k;
    const u32 *t;
    u64 n;
    u32 ct;
};

// stage the caller's streams (host or device pointers) as device streams
// device_streams (UKM_F_DEVICE_STREAMS): the caller vouches that every pointer is a device pointer, so none is classified
int stage_streams(ukm_ctx *c, const uint64_t *const *keys, const uint32_t *const *taxids, const uint32_t *file_taxids,
                  const uint64_t *lens, int nstreams, bool want_tax, std::vector<Stream> &out, bool device_streams = false) {
    out.resize((size_t)nstreams);
    for (int i = 0; i < nstreams; i++) {
        Stream s;
        s.n = lens[i];
        s.k = nullptr;
        s.t = nullptr;
        s.ct = 0;
        if (s.n && !keys[i]) UKM_FAIL(UKM_ERR_INVALID, "stream %d: keys is NULL", i);
        if (device_streams) {
            s.k = keys[i];
            if (want_tax && taxids && taxids[i]) s.t = taxids[i];
        } else {
            UKM_TRY(ukm_in_t(c, keys[i], s.n, &s.k));
            if (want_tax && taxids && taxids[i]) UKM_TRY(ukm_in_t(c, taxids[i], s.n, &s.t));
        }
        if (want_tax && !s.t && file_taxids) s.ct = file_taxids[i];
        out[(size_t)i] = s;
    }
    return UKM_OK;
}

bool any_taxids(const uint32_t *const *taxids, const uint32_t *file_taxids, int nstreams) {
    for (int i = 0; i < nstreams; i++)
        if ((taxids && taxids[i]) || (file_taxids && file_taxids[i])) return true;
    return false;
}

// routes without a per-file form (the k-way / single-pass / placement merges, the range folds): the file's taxid as an array
int materialise_ct(ukm_ctx *c, Stream &s, bool tax) {
    if (!tax || s.t || s.n == 0 || s.ct == 0) return UKM_OK;
    u32 *t = nullptr;
    UKM_TRY(ws_alloc_t(c, s.n, &t));
    UKM_TRY(ukm_dev_fill_u32(c, t, s.n, s.ct));
    s.t = t;
    s.ct = 0;
    return UKM_OK;
}
int materialise_all(ukm_ctx *c, std::vector<Stream> &ss, bool tax) {
    for (auto &s : ss) UKM_TRY(materialise_ct(c, s, tax));
    return UKM_OK;
}
// every stream that takes part carries one taxid per file (or none)
bool all_per_file(const std::vector<Stream> &ss, bool tax) {
    if (!tax) return false;
    for (auto &s : ss)
        if (s.t) return false;
    return true;
}
void strip_taxids(std::vector<Stream> &ss) {
    for (auto &s : ss) {
        s.t = nullptr;
        s.ct = 0;
    }
}

int check_common_args(ukm_ctx *ctx, const uint64_t *const *keys, const uint64_t *lens, int nstreams,
                      uint64_t *out_keys, uint64_t out_cap, uint64_t *n_out, const char *name) {
    if (!ctx || !n_out || nstreams < 0 || (nstreams && (!keys || !lens)) || (!out_keys && out_cap))
        UKM_FAIL(UKM_ERR_INVALID, "%s: bad argument", name);
    return UKM_OK;
}

// make a sorted, duplicate-free (LCA-folded) copy of a stream if it is not already one
int normalise_set(ukm_ctx *c, Stream &s, bool tax) {
    tax = tax && s.t != nullptr;  // (one taxid per file: the fold over equal codes leaves it as it is, LCA(x, x) = x)
    bool sorted = true, strict = true;
    UKM_TRY(ukm_dev_check_sorted(c, s.k, s.n, &sorted, &strict));
    if (strict) return UKM_OK;
    u64 *k = nullptr;
    u32 *t = nullptr;
    UKM_TRY(ws_alloc_t(c, s.n, &k));
    if (tax) UKM_TRY(ws_alloc_t(c, s.n, &t));
    if (!sorted) {
        UKM_HIP(hipMemcpyAsync(k, s.k, s.n * sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
        if (tax) {
            if (s.t) UKM_HIP(hipMemcpyAsync(t, s.t, s.n * sizeof(u32), hipMemcpyDeviceToDevice, c->stream));
            else UKM_HIP(hipMemsetAsync(t, 0, s.n * sizeof(u32), c->stream));
        }
        UKM_TRY(ukm_dev_sort(c, k, tax ? t : nullptr, s.n, 64));
        u64 *k2 = nullptr;
        u32 *t2 = nullptr;
        UKM_TRY(ws_alloc_t(c, s.n, &k2));
        if (tax) UKM_TRY(ws_alloc_t(c, s.n, &t2));
        u64 nu = 0;
        UKM_TRY(ukm_dev_unique(c, k, tax ? t : nullptr, s.n, UKM_UNIQUE, k2, t2, s.n, &nu));
        s.k = k2; s.t = tax ? t2 : nullptr; s.n = nu;
    } else {
        u64 nu = 0;
        UKM_TRY(ukm_dev_unique(c, s.k, tax ? s.t : nullptr, s.n, UKM_UNIQUE, k, t, s.n, &nu));
        s.k = k; s.t = tax ? t : nullptr; s.n = nu;
    }
    return UKM_OK;
}

int copy_result(ukm_ctx *c, const Stream &s, bool tax, u64 *out, u32 *tout, u64 out_cap, u64 *n_out) {
    *n_out = s.n;
    if (s.n > out_cap)
        UKM_FAIL(UKM_ERR_CAPACITY, "output needs %llu records, capacity is %llu", (unsigned long long)s.n,
                 (unsigned long long)out_cap);
    if (s.n && out != s.k) UKM_HIP(hipMemcpyAsync(out, s.k, s.n * sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
    if (tax && tout && s.n) {
        if (s.t) {
            if (tout != s.t) UKM_HIP(hipMemcpyAsync(tout, s.t, s.n * sizeof(u32), hipMemcpyDeviceToDevice, c->stream));
        } else {
            UKM_TRY(ukm_dev_fill_u32(c, tout, s.n, s.ct));
        }
    }
    return UKM_OK;
}

// concatenate streams into one device buffer (missing taxids -> 0)
int concat_streams(ukm_ctx *c, const std::vector<Stream> &ss, bool tax, u64 **k, u32 **t, u64 *total) {
    u64 n = 0;
    for (auto &s : ss) n += s.n;
    *total = n;
    UKM_TRY(ws_alloc_t(c, n + 1, k));
    if (tax) UKM_TRY(ws_alloc_t(c, n + 1, t));
    u64 off = 0;
    for (auto &s : ss) {
        if (!s.n) continue;
        UKM_HIP(hipMemcpyAsync(*k + off, s.k, s.n * sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
        if (tax) {
            if (s.t) UKM_HIP(hipMemcpyAsync(*t + off, s.t, s.n * sizeof(u32), hipMemcpyDeviceToDevice, c->stream));
            else UKM_TRY(ukm_dev_fill_u32(c, *t + off, s.n, s.ct));
        }
        off += s.n;
    }
    return UKM_OK;
}

struct OutBufs {
    u64 *k = nullptr;
    u32 *t = nullptr;
};

template <typename F>
int run_entry(ukm_ctx *ctx, uint64_t *out_keys, uint32_t *out_taxids, uint64_t out_cap, uint64_t *n_out,
              bool tax, F body) {
    CallScope s;
    UKM_TRY(ukm_begin(ctx, &s));
    int rc = [&]() -> int {
        OutBufs o;
        UKM_TRY(ukm_out_t(ctx, out_keys, out_cap, &o.k));
        if (tax && !out_taxids) UKM_FAIL(UKM_ERR_INVALID, "records carry taxids but out_taxids is NULL");
        if (tax) UKM_TRY(ukm_out_t(ctx, out_taxids, out_cap, &o.t));
        *n_out = 0;
        ctx->last_route = 0;
        int r = body(o);
        u64 n = (r == UKM_OK) ? *n_out : 0;
        ukm_out_resize(ctx, out_keys, n * sizeof(u64));
        if (out_taxids) ukm_out_resize(ctx, out_taxids, n * sizeof(u32));
        return r;
    }();
    return ukm_finish(&s, rc);
}

// Pairwise reduction tree over >= 2 non-empty sorted streams with a 2-way operation (UNION: LCA is
// associative and commutative, so the tree equals the reference's arrival-order fold; MERGE: every
// record kept).  Levels ping-pong between two workspace buffers; the last level writes (fk, ft).
// lazy_normalise (UNION): the caller's streams are NOT pre-checked (that would read every input once more:
// 16 of 95 ms for 100 files x 1e8); the 2-way kernel checks order while it merges, and only if it reports an
// unsorted input are the two original streams sorted / deduplicated (the reference's hash-map union accepts
// unsorted files) and the pair retried.
int tree_reduce(ukm_ctx *ctx, std::vector<Stream> ss, int op, u32 flags, bool tax, u64 *fk, u32 *ft, u64 fcap, u64 *n_out,
                bool lazy_normalise = false) {
    u64 total = 0;
    for (auto &s : ss) total += s.n;
    ctx->last_route = ss.size() > 2 ? 1 : 0;
    std::vector<char> orig(ss.size(), lazy_normalise ? 1 : 0);
    u64 *bk[2] = {nullptr, nullptr};
    u32 *bt[2] = {nullptr, nullptr};
    if (ss.size() > 2) {
        for (int i = 0; i < 2; i++) {
            UKM_TRY(ws_alloc_t(ctx, total + 1, &bk[i]));
            if (tax) UKM_TRY(ws_alloc_t(ctx, total + 1, &bt[i]));
        }
    }
    int level = 0;
    while (ss.size() > 1) {
        std::vector<Stream> next;
        std::vector<char> next_orig;
        const bool last = ss.size() == 2;
        u64 off = 0;
        for (size_t i = 0; i + 1 < ss.size(); i += 2) {
            Stream &a = ss[i], &b = ss[i + 1];
            u64 *ok = last ? fk : bk[level & 1] + off;
            u32 *ot = tax ? (last ? ft : bt[level & 1] + off) : nullptr;
            const u64 cap = last ? fcap : a.n + b.n;  // a.n / b.n only shrink if they get normalised below
            u64 n = 0;
            // (a stream with one taxid per file goes into the 2-way kernel as it is: ukm_setops.hip, SetopArgs::cta)
            int r = ukm_dev_setop2_ct(ctx, op, a.k, a.t, tax ? a.ct : 0u, a.n, b.k, b.t, tax ? b.ct : 0u, b.n, flags, ok, ot, cap, &n);
            if (r == UKM_ERR_UNSORTED && (orig[i] || orig[i + 1])) {
                if (orig[i]) UKM_TRY(normalise_set(ctx, a, tax));
                if (orig[i + 1]) UKM_TRY(normalise_set(ctx, b, tax));
                orig[i] = orig[i + 1] = 0;
                r = ukm_dev_setop2_ct(ctx, op, a.k, a.t, tax ? a.ct : 0u, a.n, b.k, b.t, tax ? b.ct : 0u, b.n, flags, ok, ot, cap, &n);
            }
            if (last) *n_out = n;
            UKM_TRY(r);
            next.push_back(Stream{ok, ot, n, 0u});
            next_orig.push_back(0);
            off += cap;
        }
        if (ss.size() & 1) {
            // carry the odd stream INTO this level's buffer, so that the next level (which
            // writes the other buffer) never overwrites something it still has to read
            const Stream &z = ss.back();
            u64 *zk = bk[level & 1] + off;
            u32 *zt = tax ? bt[level & 1] + off : nullptr;
            UKM_HIP(hipMemcpyAsync(zk, z.k, z.n * sizeof(u64), hipMemcpyDeviceToDevice, ctx->stream));
            if (tax) {
                if (z.t) UKM_HIP(hipMemcpyAsync(zt, z.t, z.n * sizeof(u32), hipMemcpyDeviceToDevice, ctx->stream));
                else UKM_TRY(ukm_dev_fill_u32(ctx, zt, z.n, z.ct));
            }
            next.push_back(Stream{zk, zt, z.n, 0u});
            next_orig.push_back(orig.back());  // still unchecked: the level that merges it will see
        }
        ss.swap(next);
        orig.swap(next_orig);
        level++;
    }
    return UKM_OK;
}

// k-way streaming merge (ukm_kway.hip) over >= 3 non-empty sorted streams.  *done = false: the inputs need the
// general route (an unsorted stream, a very long run of one code); the workspace it used is given back.
int try_kway(ukm_ctx *ctx, int op, std::vector<Stream> ss, bool tax, u64 *fk, u32 *ft, u64 fcap, u64 *n_out,
             bool *done) {
    *done = false;
    if (ss.size() < 3 || !ukm_kway_enabled(ctx)) return UKM_OK;
    {
        // a handful of tiny streams: the pairwise tree (a few 60-us calls) beats the k-way set-up (sample, sort,
        // cuts, several small launches and read-backs).  UKM_KWAY=1 forces the k-way path (tests).
        u64 total = 0;
        for (auto &x : ss) total += x.n;
        const bool forced = ukm_env_is(ctx, "UKM_KWAY", '1');
        if (!forced && ss.size() <= 4 && total < (1u << 16)) return UKM_OK;
    }
    std::vector<const u64 *> kp(ss.size());
    std::vector<const u32 *> tp(ss.size());
    std::vector<u64> ln(ss.size());
    std::vector<u32> cv(ss.size());
    for (size_t i = 0; i < ss.size(); i++) {
        kp[i] = ss[i].k;
        tp[i] = ss[i].t;
        ln[i] = ss[i].n;
        cv[i] = ss[i].t ? 0u : ss[i].ct;
    }
    bool fallback = true;
    if (op == UKM_KWAY_MERGE) {
        // many files that share most of their codes: the records of every code are placed behind one another file by file
        // (ukm_punion.hip, pl_merge_kernel); it declines for few / small files, files that share little, a duplicate
        // inside a file, an unsorted file.  (A file with ONE taxid goes in as it is: the kernel writes the scalar.)
        WsMark pmark = ws_mark(ctx);
        const int prc = ukm_dev_place_merge(ctx, kp.data(), tax ? tp.data() : nullptr, ln.data(), (int)ss.size(), tax, fk, ft, fcap, n_out,
                                            &fallback, tax ? cv.data() : nullptr);
        if (prc != UKM_OK || !fallback) {
            ws_release(ctx, pmark);
            UKM_TRY(prc);
            ctx->last_route = 7;
            *done = true;
            return UKM_OK;
        }
        ws_release(ctx, pmark);
        fallback = true;
    }
    UKM_TRY(materialise_all(ctx, ss, tax));  // (the merges below read a taxid per record)
    for (size_t i = 0; i < ss.size(); i++) tp[i] = ss[i].t;
    WsMark mark = ws_mark(ctx);
    {
        // many short streams: one pass over HBM, every value range ordered inside LDS (ukm_srmerge.hip); it declines
        // (*fallback) for few streams, small inputs, unsorted streams and one code with thousands of copies
        const int src = ukm_dev_srmerge(ctx, op, kp.data(), tax ? tp.data() : nullptr, ln.data(), (int)ss.size(), tax, fk, ft, fcap,
                                        n_out, &fallback);
        if (src != UKM_OK || !fallback) {
            ws_release(ctx, mark);
            UKM_TRY(src);
            ctx->last_route = 4;
            *done = true;
            return UKM_OK;
        }
        ws_release(ctx, mark);
    }
    fallback = false;
    const int rc = ukm_dev_kway(ctx, op, kp.data(), tax ? tp.data() : nullptr, ln.data(), (int)ss.size(), tax, fk, ft, fcap,
                                n_out, &fallback);
    ws_release(ctx, mark);
    UKM_TRY(rc);
    *done = !fallback;
    if (*done) ctx->last_route = 2;
    return UKM_OK;
}

// `union` of many sets by LDS hash probes against the union of the first eight (with TaxIds: four) files
// (ukm_punion.hip): the shape of an n-file union over related genomes, where after a few files nearly every record is
// already in the result.  Taken for >= PUNION_MIN_STREAMS streams and >= 2^27 (with TaxIds 2^26) records behind the first eight; the path itself
// backs out (*done = false, nothing written) when a sample of the later files is not found in the base set, when a
// stream is unsorted or when its miss list overflows, and the k-way merge below answers.
constexpr int PUNION_MIN_STREAMS = 24;
int try_probe_union(ukm_ctx *ctx, const std::vector<Stream> &ss, bool tax, u64 *fk, u32 *ft, u64 fcap, u64 *n_out, bool *done) {
    *done = false;
    const int mode = ukm_punion_mode(ctx);
    if (mode == 0 || !ukm_kway_enabled(ctx)) return UKM_OK;
    if (tax && ukm_punion_tax_mode(ctx) == 0) return UKM_OK;
    if (mode < 1) {
        if ((int)ss.size() < PUNION_MIN_STREAMS) return UKM_OK;
        u64 later = 0;
        for (size_t i = 8; i < ss.size(); i++) later += ss[i].n;
        // (with taxids the other routes cost more per record: 1e8 records in 100 - 1000 files 2.5 - 3.8 ms here against
        //  3.3 - 4.8 ms; 3e7 records 1.4 - 3.5 against 1.3 - 2.8)
        if (later < (tax ? (1ull << 26) : (1ull << 27))) return UKM_OK;
    }
    std::vector<const u64 *> kp(ss.size());
    std::vector<const u32 *> tp(ss.size());
    std::vector<u64> ln(ss.size());
    std::vector<u32> cv(ss.size());
    for (size_t i = 0; i < ss.size(); i++) {
        kp[i] = ss[i].k;
        tp[i] = ss[i].t;
        ln[i] = ss[i].n;
        cv[i] = ss[i].t ? 0u : ss[i].ct;
    }
    WsMark mark = ws_mark(ctx);
    bool fallback = false;
    const int rc = ukm_dev_probe_union(ctx, kp.data(), tax ? tp.data() : nullptr, ln.data(), (int)ss.size(), tax, fk, ft, fcap, n_out,
                                       &fallback, tax ? cv.data() : nullptr);
    ws_release(ctx, mark);
    UKM_TRY(rc);
    *done = !fallback;
    if (*done) ctx->last_route = 3;
    return UKM_OK;
}

// All records of the (non-empty) streams as ONE sequence ordered by code, equal codes in stream
// order (= a stable sort of the concatenation).  Sorted streams (chunk files, .unik sets) go through
// the keep-everything merge tree; anything else is concatenated and radix sorted.
// (dk, dt): optional destination with room for every record (a PLAIN merge writes the caller's buffer directly instead
// of a workspace copy that is copied once more: 12 GB less traffic for 1e9 records with taxids)
int merged_sequence(ukm_ctx *ctx, const std::vector<Stream> &all, bool tax, u64 **k, u32 **t, u64 *total, u64 *dk = nullptr,
                    u32 *dt = nullptr) {
    std::vector<Stream> ss;
    u64 n = 0;
    for (auto &s : all)
        if (s.n) {
            ss.push_back(s);
            n += s.n;
        }
    *total = n;
    *k = nullptr;
    *t = nullptr;
    if (n == 0) return UKM_OK;
    bool need_sort = false;
    if (ss.size() > 1) {
        // optimistic: the merges check the order of what they read; an unsorted stream makes the tree fail
        // with UKM_ERR_UNSORTED and the whole input takes the concatenate + sort route instead
        if (dk && (!tax || dt)) {
            *k = dk;
            *t = tax ? dt : nullptr;
        } else {
            UKM_TRY(ws_alloc_t(ctx, n + 1, k));
            if (tax) UKM_TRY(ws_alloc_t(ctx, n + 1, t));
        }
        u64 nm = 0;
        bool done = false;
        UKM_TRY(try_kway(ctx, UKM_KWAY_MERGE, ss, tax, *k, *t, n, &nm, &done));
        if (done) return UKM_OK;
        const int r = tree_reduce(ctx, ss, UKM_OP_MERGE_INTERNAL, 0, tax, *k, *t, n, &nm);
        if (r == UKM_OK) return UKM_OK;
        if (r != UKM_ERR_UNSORTED) return r;
        need_sort = true;
    } else {
        bool sorted = true, strict = true;
        UKM_TRY(ukm_dev_check_sorted(ctx, ss[0].k, ss[0].n, &sorted, &strict));
        need_sort = !sorted;
    }
    UKM_TRY(concat_streams(ctx, ss, tax, k, t, total));
    if (need_sort) UKM_TRY(ukm_dev_sort(ctx, *k, *t, n, 64));
    return UKM_OK;
}

// ---- chained fold (inter / diff over many files) -----------------------------------------------------------
// The reference folds the files one after the other into a running result.  Done naively every link costs a
// host round trip for the running size (~40 us: with 1000 files of 1e6 codes that IS the run time).  Here the
// size stays on the device: link i reads |acc| from link i-1's result word, launches with the upper bound
// (the first file's size) and tiles beyond the real size exit at once.  The host looks at a count only every
// a few links (4, 8, 16, ... then every 64: early exit when the running result is empty: inter.go:283-286, diff.go:457-459) and reads all
// flags once at the end.  A duplicate code inside an input (multiset semantics need the rank path) or a
// look-back watchdog makes the caller fall back to the synchronous fold.
constexpr int CHAIN_MIN_STREAMS = 4;
constexpr int CHAIN_PEEK_FIRST = 4;   // the host looks at the running size after 4, 8, 16, ... links (then every 64)
constexpr int CHAIN_PEEK_MAX = 64;

struct ChainResult {
    Stream acc;           // device buffers of the final running result
    u64 n = 0;            // its size
    bool fallback = false;
    bool unsorted = false;
};

// streams[1..] are folded into acc0 with `op`; unsorted later files of diff must already be sorted copies.
// stop_at_empty_later: inter.go:211-217 (an empty later file ends the fold, the running result is kept).
int fold_chained(ukm_ctx *ctx, int op, const std::vector<Stream> &ss, u32 flags, bool tax, bool stop_at_empty_later,
                 u64 *bk[2], u32 *bt[2], ChainResult *res) {
    const int nlinks_max = (int)ss.size() - 1;
    u64 *ctl = nullptr;  // 8 words per link, alive until the end
    UKM_TRY(ws_alloc_t(ctx, (size_t)nlinks_max * 8 + 8, &ctl));
    UKM_HIP(hipMemsetAsync(ctl, 0, ((size_t)nlinks_max * 8 + 8) * sizeof(u64), ctx->stream));
    const u64 na_max = ss[0].n;
    Stream acc = ss[0];
    const u64 *na_dev = nullptr;  // first link: the size is the first file's
    int flip = 0, links = 0, next_peek = CHAIN_PEEK_FIRST;
    bool empty = false;
    for (size_t i = 1; i < ss.size(); i++) {
        const Stream &q = ss[i];
        if (q.n == 0) {
            if (stop_at_empty_later) break;
            continue;
        }
        u64 *c8 = ctl + (size_t)links * 8;
        WsMark mark = ws_mark(ctx);
        UKM_TRY(ukm_dev_setop2_link(ctx, op, acc.k, acc.t, na_max, na_dev, q.k, q.t, q.n, flags, bk[flip],
                                    tax ? bt[flip] : nullptr, na_max, c8, tax ? acc.ct : 0u, tax ? q.ct : 0u));
        ws_release(ctx, mark);
        acc = Stream{bk[flip], tax ? bt[flip] : nullptr, na_max, 0u};
        na_dev = c8;
        flip ^= 1;
        links++;
        if (links == next_peek) {
            next_peek += (next_peek < CHAIN_PEEK_MAX) ? next_peek : CHAIN_PEEK_MAX;
            u64 r2[2];
            UKM_TRY(ukm_read_u64(ctx, c8, r2, 2));
            if (r2[1] & (UKM_SETOP_FLAG_DUP | UKM_SETOP_FLAG_TIMEOUT | UKM_SETOP_FLAG_UNSORTED)) break;
            if (r2[0] == 0) { empty = true; break; }
        }
    }
    res->acc = acc;
    if (links == 0) {
        res->n = ss[0].n;
        return UKM_OK;
    }
    std::vector<u64> h((size_t)links * 8);
    UKM_HIP(hipMemcpyAsync(h.data(), ctl, h.size() * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
    UKM_HIP(hipStreamSynchronize(ctx->stream));
    u64 fl = 0;
    for (int l = 0; l < links; l++) fl |= h[(size_t)l * 8 + 1];
    res->unsorted = (fl & UKM_SETOP_FLAG_UNSORTED) != 0;
    res->fallback = (fl & (UKM_SETOP_FLAG_DUP | UKM_SETOP_FLAG_TIMEOUT)) != 0;
    if (fl & UKM_SETOP_FLAG_TIMEOUT) ukm_switch_to_tickets(ctx, "chained set-op fold");
    res->n = empty ? 0 : h[(size_t)(links - 1) * 8];
    return UKM_OK;
}

// `inter` / `diff` over many files as ONE range-partitioned launch (ukm_fold.hip) instead of one link per file.
// ss[0] is the running result's start; empty later files have already been handled by the caller's rule (inter:
// the list ends in front of the first one; diff: they are dropped).  *done = false: not eligible (too few files, a
// first file that is too large or far smaller than the others), or the kernel saw a duplicate code (the exact
// multiset route of the chained / synchronous fold answers then).
constexpr u64 FOLD_MAX_FIRST = 1ull << 24;  // larger first files: the 2-way tile kernel streams them faster per link
int try_range_fold(ukm_ctx *ctx, int op, std::vector<Stream> ss, u32 flags, bool tax, u64 *fk, u32 *ft, u64 fcap,
                   u64 *n_out, bool *done) {
    *done = false;
    if (!ukm_fold_enabled(ctx) || ss.size() < (size_t)CHAIN_MIN_STREAMS || ss[0].n == 0 || ss[0].n > FOLD_MAX_FIRST) return UKM_OK;
    UKM_TRY(materialise_all(ctx, ss, tax));  // (files with one taxid each beside files with one per record: rare; all per file: the callers' fills)
    std::vector<const u64 *> kp(ss.size());
    std::vector<const u32 *> tp(ss.size());
    std::vector<u64> ln(ss.size());
    for (size_t i = 0; i < ss.size(); i++) {
        kp[i] = ss[i].k;
        tp[i] = ss[i].t;
        ln[i] = ss[i].n;
    }
    if (ukm_pfold_enabled(ctx)) {
        // the order-independent rules (inter without --mix-taxid, diff without -t) by hash probes (ukm_pfold.hip)
        WsMark pm = ws_mark(ctx);
        bool fb = true;
        const int prc = ukm_dev_probe_fold(ctx, op, kp.data(), tax ? tp.data() : nullptr, ln.data(), (int)ss.size(), tax, flags, fk, ft,
                                           fcap, n_out, &fb);
        ws_release(ctx, pm);
        UKM_TRY(prc);
        if (!fb) {
            *done = true;
            return UKM_OK;
        }
    }
    WsMark mark = ws_mark(ctx);
    bool fallback = false;
    const int rc = ukm_dev_range_fold(ctx, op, kp.data(), tax ? tp.data() : nullptr, ln.data(), (int)ss.size(), tax, flags, fk, ft,
                                      fcap, n_out, &fallback);
    ws_release(ctx, mark);
    UKM_TRY(rc);
    *done = !fallback;
    return UKM_OK;
}

}  // namespace

extern "C" int ukm_union_ft(ukm_ctx *ctx, const uint64_t *const *keys, const uint32_t *const *taxids, const uint32_t *file_taxids,
                            const uint64_t *lens, int nstreams, uint32_t flags, uint64_t *out_keys,
                            uint32_t *out_taxids, uint64_t out_cap, uint64_t *n_out) {
    UKM_TRY(check_common_args(ctx, keys, lens, nstreams, out_keys, out_cap, n_out, "ukm_union"));
    const bool device_streams = (flags & UKM_F_DEVICE_STREAMS) != 0;  // every stream pointer is a device pointer
    flags &= ~(uint32_t)UKM_F_DEVICE_STREAMS;
    const bool tax = any_taxids(taxids, file_taxids, nstreams);
    return run_entry(ctx, out_keys, out_taxids, out_cap, n_out, tax, [&](OutBufs &o) -> int {
        std::vector<Stream> cur;
        UKM_TRY(stage_streams(ctx, keys, taxids, file_taxids, lens, nstreams, tax, cur, device_streams));
        // drop empty streams; a single stream is made a sorted set here, several are checked by the merges
        std::vector<Stream> ss;
        for (auto &s : cur)
            if (s.n) ss.push_back(s);
        if (ss.empty()) return UKM_OK;
        if (ss.size() == 1) {
            UKM_TRY(normalise_set(ctx, ss[0], tax));
            return copy_result(ctx, ss[0], tax, o.k, o.t, out_cap, n_out);
        }
        bool tx = tax;
        u32 same_ct = 0;
        if (all_per_file(ss, tax)) {
            // every file carries ONE taxid and it is the same one (k-mers of one species from many runs): the fold of
            // union.go:195-201 leaves it as it is -- the plain union, and a fill
            bool same = true;
            for (auto &q : ss) same = same && q.ct == ss[0].ct;
            if (same) {
                same_ct = ss[0].ct;
                strip_taxids(ss);
                tx = false;
            }
        }
        auto finish = [&]() -> int { return (tax && !tx && *n_out <= out_cap) ? ukm_dev_fill_u32(ctx, o.t, *n_out, same_ct) : UKM_OK; };
        bool done = false;
        UKM_TRY(try_probe_union(ctx, ss, tx, o.k, o.t, out_cap, n_out, &done));
        if (done) return finish();
        UKM_TRY(try_kway(ctx, UKM_KWAY_UNION, ss, tx, o.k, o.t, out_cap, n_out, &done));
        if (done) return finish();
        UKM_TRY(tree_reduce(ctx, ss, UKM_OP_UNION, flags, tx, o.k, o.t, out_cap, n_out, true));
        return finish();
    });
}

extern "C" int ukm_union(ukm_ctx *ctx, const uint64_t *const *keys, const uint32_t *const *taxids,
                         const uint64_t *lens, int nstreams, uint32_t flags, uint64_t *out_keys,
                         uint32_t *out_taxids, uint64_t out_cap, uint64_t *n_out) {
    return ukm_union_ft(ctx, keys, taxids, nullptr, lens, nstreams, flags, out_keys, out_taxids, out_cap, n_out);
}

namespace {

// inter.go:188-286 over staged streams (taxids per record, per file where the 2-way kernel takes them, or none)
int inter_body(ukm_ctx *ctx, std::vector<Stream> &ss, u32 flags, bool tax, OutBufs &o, u64 out_cap, u64 *n_out) {
    const int nstreams = (int)ss.size();
    Stream acc = ss[0];  // inter.go:189-200: the running result starts as file 1
    u64 *bk[2] = {nullptr, nullptr};
    u32 *bt[2] = {nullptr, nullptr};
    if (nstreams > 1 && acc.n) {
        for (int i = 0; i < 2; i++) {
            UKM_TRY(ws_alloc_t(ctx, acc.n + 1, &bk[i]));
            if (tax) UKM_TRY(ws_alloc_t(ctx, acc.n + 1, &bt[i]));
        }
    }
    if (nstreams >= CHAIN_MIN_STREAMS && acc.n) {
        // inter.go:211-217: an empty later file ends the fold and the running result is kept
        std::vector<Stream> live(ss.begin(), ss.begin() + 1);
        for (int i = 1; i < nstreams && ss[(size_t)i].n; i++) live.push_back(ss[(size_t)i]);
        bool done = false;
        UKM_TRY(try_range_fold(ctx, UKM_OP_INTER, live, flags, tax, o.k, o.t, out_cap, n_out, &done));
        if (done) return UKM_OK;
    }
    if (nstreams >= CHAIN_MIN_STREAMS && acc.n) {
        ChainResult cr;
        UKM_TRY(fold_chained(ctx, UKM_OP_INTER, ss, flags, tax, true, bk, bt, &cr));
        if (cr.unsorted) UKM_FAIL(UKM_ERR_UNSORTED, "ukm_setop2: an input stream is not sorted");
        if (!cr.fallback) {
            cr.acc.n = cr.n;
            return copy_result(ctx, cr.acc, tax, o.k, o.t, out_cap, n_out);
        }
    }
    int flip = 0;
    for (int i = 1; i < nstreams && acc.n > 0; i++) {
        const Stream &q = ss[(size_t)i];
        if (q.n == 0) break;  // inter.go:211-217 (flagBreak: the running result is kept)
        u64 n = 0;
        UKM_TRY(ukm_dev_setop2_ct(ctx, UKM_OP_INTER, acc.k, acc.t, tax ? acc.ct : 0u, acc.n, q.k, q.t, tax ? q.ct : 0u, q.n, flags,
                                  bk[flip], tax ? bt[flip] : nullptr, acc.n, &n));
        acc = Stream{bk[flip], tax ? bt[flip] : nullptr, n, 0u};
        flip ^= 1;
    }
    return copy_result(ctx, acc, tax, o.k, o.t, out_cap, n_out);
}

}  // namespace

extern "C" int ukm_inter_ft(ukm_ctx *ctx, const uint64_t *const *keys, const uint32_t *const *taxids, const uint32_t *file_taxids,
                            const uint64_t *lens, int nstreams, uint32_t flags, uint64_t *out_keys,
                            uint32_t *out_taxids, uint64_t out_cap, uint64_t *n_out) {
    UKM_TRY(check_common_args(ctx, keys, lens, nstreams, out_keys, out_cap, n_out, "ukm_inter"));
    const bool device_streams = (flags & UKM_F_DEVICE_STREAMS) != 0;  // every stream pointer is a device pointer
    flags &= ~(uint32_t)UKM_F_DEVICE_STREAMS;
    const bool tax = any_taxids(taxids, file_taxids, nstreams);
    return run_entry(ctx, out_keys, out_taxids, out_cap, n_out, tax, [&](OutBufs &o) -> int {
        if (nstreams == 0) return UKM_OK;
        std::vector<Stream> ss;
        UKM_TRY(stage_streams(ctx, keys, taxids, file_taxids, lens, nstreams, tax, ss, device_streams));
        // the files that take part: up to the first empty later file (inter.go:211-217)
        int m = 1;
        while (m < nstreams && ss[(size_t)m].n) m++;
        std::vector<Stream> part(ss.begin(), ss.begin() + m);
        if (all_per_file(part, tax) && ss[0].n) {
            // ONE taxid per file (`count -t`): every record that survives has met every file that takes part, so its taxid is
            // the same left fold for all of them -- LCA(LCA(t1, t2), t3) ... with the mix-taxid rule (inter.go:229-239) --
            // worked out once on the device: the PLAIN intersection, and a fill
            if (ctx->tax_parent == nullptr) {
                bool trivial = true;  // (no LCA is ever looked up when all taxids are equal)
                for (auto &q : part) trivial = trivial && q.ct == part[0].ct;
                if (!trivial) UKM_FAIL(UKM_ERR_NO_TAXONOMY, "ukm_setop2: records carry taxids but no taxonomy is loaded");
            }
            std::vector<u32> cts((size_t)m);
            for (int i = 0; i < m; i++) cts[(size_t)i] = part[(size_t)i].ct;
            u32 *plan = nullptr;
            UKM_TRY(ukm_dev_ct_plan(ctx, cts.data(), m, (flags & UKM_F_MIX_TAXID) != 0, &plan));
            strip_taxids(ss);
            UKM_TRY(inter_body(ctx, ss, flags, false, o, out_cap, n_out));
            return ukm_dev_fill_u32_from(ctx, o.t, *n_out, 0u, plan);
        }
        return inter_body(ctx, ss, flags, tax, o, out_cap, n_out);
    });
}

extern "C" int ukm_inter(ukm_ctx *ctx, const uint64_t *const *keys, const uint32_t *const *taxids,
                         const uint64_t *lens, int nstreams, uint32_t flags, uint64_t *out_keys,
                         uint32_t *out_taxids, uint64_t out_cap, uint64_t *n_out) {
    return ukm_inter_ft(ctx, keys, taxids, nullptr, lens, nstreams, flags, out_keys, out_taxids, out_cap, n_out);
}

namespace {

// diff.go:341-454 over staged streams; sorted_flags[i] == 0 marks an unsorted later file (NULL: all sorted)
int diff_body(ukm_ctx *ctx, std::vector<Stream> &ss, const std::vector<u8> &sorted_flags, u32 flags, bool tax, OutBufs &o, u64 out_cap,
              u64 *n_out) {
    const int nstreams = (int)ss.size();
    const bool have_flags = !sorted_flags.empty();
    Stream acc = ss[0];
    bool a_sorted = true, a_strict = true;
    UKM_TRY(ukm_dev_check_sorted(ctx, acc.k, acc.n, &a_sorted, &a_strict));
    if (!a_sorted) UKM_FAIL(UKM_ERR_UNSORTED, "ukm_diff: the first stream must be sorted (diff.go:115-117)");
    u64 *bk[3] = {nullptr, nullptr, nullptr};
    u32 *bt[3] = {nullptr, nullptr, nullptr};
    if (acc.n) {
        for (int i = 0; i < 3; i++) {
            UKM_TRY(ws_alloc_t(ctx, acc.n + 1, &bk[i]));
            if (tax) UKM_TRY(ws_alloc_t(ctx, acc.n + 1, &bt[i]));
        }
    }
    // a sorted copy of an unsorted file (diff.go:341-378)
    auto sorted_copy = [&](Stream &q) -> int {
        u64 *k = nullptr;
        u32 *t = nullptr;
        UKM_TRY(ws_alloc_t(ctx, q.n, &k));
        UKM_HIP(hipMemcpyAsync(k, q.k, q.n * sizeof(u64), hipMemcpyDeviceToDevice, ctx->stream));
        if (tax && (q.t || q.ct == 0)) {  // (one taxid per file: nothing to carry along)
            UKM_TRY(ws_alloc_t(ctx, q.n, &t));
            if (q.t) UKM_HIP(hipMemcpyAsync(t, q.t, q.n * sizeof(u32), hipMemcpyDeviceToDevice, ctx->stream));
            else UKM_HIP(hipMemsetAsync(t, 0, q.n * sizeof(u32), ctx->stream));
        }
        UKM_TRY(ukm_dev_sort(ctx, k, t, q.n, 64));
        q.k = k;
        q.t = t;
        return UKM_OK;
    };
    if (nstreams >= CHAIN_MIN_STREAMS && acc.n && a_strict) {
        // chained fold: sorted copies of the unsorted files first (diff.go:341-378), then one link per file
        std::vector<Stream> ss2 = ss;
        for (int i = 1; i < nstreams; i++) {
            Stream &q = ss2[(size_t)i];
            if (q.n == 0 || !have_flags || sorted_flags[(size_t)i]) continue;
            UKM_TRY(sorted_copy(q));
        }
        {
            std::vector<Stream> live;
            for (auto &q : ss2)
                if (q.n) live.push_back(q);  // diff.go: empty files subtract nothing
            bool done = false;
            UKM_TRY(try_range_fold(ctx, UKM_OP_DIFF, live, flags, tax, o.k, o.t, out_cap, n_out, &done));
            if (done) return UKM_OK;
        }
        ChainResult cr;
        UKM_TRY(fold_chained(ctx, UKM_OP_DIFF, ss2, flags, tax, false, bk, bt, &cr));
        if (cr.unsorted) UKM_FAIL(UKM_ERR_UNSORTED, "ukm_setop2: an input stream is not sorted");
        if (!cr.fallback) {
            cr.acc.n = cr.n;
            return copy_result(ctx, cr.acc, tax, o.k, o.t, out_cap, n_out);
        }
    }
    int flip = 0;
    for (int i = 1; i < nstreams && acc.n > 0; i++) {
        Stream q = ss[(size_t)i];
        if (q.n == 0) continue;
        WsMark mark = ws_mark(ctx);
        if (have_flags && !sorted_flags[(size_t)i]) UKM_TRY(sorted_copy(q));  // unsorted file (diff.go:341-378)
        u64 n = 0;
        // (a first file with duplicate codes: every record stays in the running list between the files, diff.go:437; the
        //  collapse to one record per code follows once, below)
        UKM_TRY(ukm_dev_setop2_ct(ctx, UKM_OP_DIFF, acc.k, acc.t, tax ? acc.ct : 0u, acc.n, q.k, q.t, tax ? q.ct : 0u, q.n,
                                  flags | UKM_F_INTERNAL_KEEP_DUPS, bk[flip], tax ? bt[flip] : nullptr, acc.n, &n));
        acc = Stream{bk[flip], tax ? bt[flip] : nullptr, n, 0u};
        flip ^= 1;
        ws_release(ctx, mark);
    }
    if (!a_strict && acc.n) {
        // the survivor map collapses duplicate codes, last record wins (diff.go:449-453)
        u64 n = 0;
        UKM_TRY(materialise_ct(ctx, acc, tax));
        UKM_TRY(ukm_dev_unique_ex(ctx, acc.k, acc.t, acc.n, 5, 0, bk[2], tax ? bt[2] : nullptr, acc.n, &n));
        acc = Stream{bk[2], tax ? bt[2] : nullptr, n, 0u};
    }
    return copy_result(ctx, acc, tax, o.k, o.t, out_cap, n_out);
}

}  // namespace

extern "C" int ukm_diff_ft(ukm_ctx *ctx, const uint64_t *const *keys, const uint32_t *const *taxids, const uint32_t *file_taxids,
                           const uint64_t *lens, int nstreams, const uint8_t *sorted_flags, uint32_t flags,
                           uint64_t *out_keys, uint32_t *out_taxids, uint64_t out_cap, uint64_t *n_out) {
    UKM_TRY(check_common_args(ctx, keys, lens, nstreams, out_keys, out_cap, n_out, "ukm_diff"));
    const bool device_streams = (flags & UKM_F_DEVICE_STREAMS) != 0;  // every stream pointer is a device pointer
    flags &= ~(uint32_t)UKM_F_DEVICE_STREAMS;
    const bool tax = any_taxids(taxids, file_taxids, nstreams);
    if ((flags & UKM_F_CMP_TAXID) && !tax) UKM_FAIL(UKM_ERR_INVALID, "ukm_diff: -t needs taxids");
    return run_entry(ctx, out_keys, out_taxids, out_cap, n_out, tax, [&](OutBufs &o) -> int {
        if (nstreams == 0) return UKM_OK;
        std::vector<Stream> ss;
        UKM_TRY(stage_streams(ctx, keys, taxids, file_taxids, lens, nstreams, tax, ss, device_streams));
        std::vector<u8> sf;
        if (sorted_flags) sf.assign(sorted_flags, sorted_flags + nstreams);
        const bool cmp = (flags & UKM_F_CMP_TAXID) != 0;
        if (tax && !ss[0].t && (!cmp || all_per_file(ss, tax))) {
            // The survivors keep the first file's taxids (diff.go:404-409) -- here ONE value, the file's own: the PLAIN
            // subtraction, and a fill.  With -t and one taxid per file, whether file j takes matched codes away is one
            // decision per FILE (its taxid equals the first file's or lies below it: it takes nothing): such files drop
            // out of the call.
            const u32 ct0 = ss[0].ct;
            if (cmp) {
                if (ctx->tax_parent == nullptr) UKM_FAIL(UKM_ERR_NO_TAXONOMY, "ukm_setop2: records carry taxids but no taxonomy is loaded");
                std::vector<u32> cts((size_t)nstreams);
                for (int i = 0; i < nstreams; i++) cts[(size_t)i] = ss[(size_t)i].ct;
                u32 *plan = nullptr;
                UKM_TRY(ukm_dev_ct_plan(ctx, cts.data(), nstreams, false, &plan));
                std::vector<u32> keep((size_t)nstreams);
                UKM_HIP(hipMemcpyAsync(keep.data(), plan + 2, (size_t)nstreams * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
                UKM_HIP(hipStreamSynchronize(ctx->stream));
                std::vector<Stream> ss2(1, ss[0]);
                std::vector<u8> sf2(1, 1);
                for (int i = 1; i < nstreams; i++)
                    if (!keep[(size_t)i]) {
                        ss2.push_back(ss[(size_t)i]);
                        sf2.push_back(sf.empty() ? (u8)1 : sf[(size_t)i]);
                    }
                ss.swap(ss2);
                if (!sf.empty()) sf.swap(sf2);
            }
            strip_taxids(ss);
            UKM_TRY(diff_body(ctx, ss, sf, flags & ~(u32)UKM_F_CMP_TAXID, false, o, out_cap, n_out));
            return ukm_dev_fill_u32(ctx, o.t, *n_out, ct0);
        }
        return diff_body(ctx, ss, sf, flags, tax, o, out_cap, n_out);
    });
}

extern "C" int ukm_diff(ukm_ctx *ctx, const uint64_t *const *keys, const uint32_t *const *taxids,
                        const uint64_t *lens, int nstreams, const uint8_t *sorted_flags, uint32_t flags,
                        uint64_t *out_keys, uint32_t *out_taxids, uint64_t out_cap, uint64_t *n_out) {
    return ukm_diff_ft(ctx, keys, taxids, nullptr, lens, nstreams, sorted_flags, flags, out_keys, out_taxids, out_cap, n_out);
}

// common.go:93-105
extern "C" uint32_t ukm_common_threshold(uint32_t nfiles, double proportion, uint32_t number) {
    if (number == 0) return (uint32_t)(uint16_t)((double)nfiles * proportion);
    return (uint32_t)(uint16_t)number;
}

static bool common_probe_enabled(const ukm_ctx *c) { return !ukm_env_is(c, "UKM_COMMON_PROBE", '0'); }  // 0 = `common` always by the counting merge

namespace {

struct StreamTables {
    std::vector<const u64 *> kp;
    std::vector<const u32 *> tp;
    std::vector<u64> ln;
    std::vector<u32> cv;  // the file taxid of a stream without per-record taxids
    explicit StreamTables(const std::vector<Stream> &ss, bool skip_empty) {
        for (auto &q : ss)
            if (q.n || !skip_empty) {
                kp.push_back(q.k);
                tp.push_back(q.t);
                ln.push_back(q.n);
                cv.push_back(q.t ? 0u : q.ct);
            }
    }
};

// common.go:220-344 over staged streams
int common_body(ukm_ctx *ctx, std::vector<Stream> &ss, u32 threshold, bool tax, OutBufs &o, u64 out_cap, u64 *n_out, bool *probe_fold_done) {
    const int nstreams = (int)ss.size();
    if (probe_fold_done) *probe_fold_done = false;
    // threshold = number of files (the default `-p 1`) over duplicate-free sorted files: a code reaches the count
    // only by being in every file, and its taxid is the same left fold of LCAs as `inter`'s (common.go:262-266 /
    // inter.go:252-262) -> the hash-probe fold of ukm_pfold.hip answers in one pass over the files.  It checks the
    // strict order of every file on the way; a duplicate, an unsorted or an empty file, an all-ones code in the
    // first file or an unsuitable shape leave the call to the counting merge below.
    if (threshold == (u32)nstreams && nstreams >= CHAIN_MIN_STREAMS && ukm_pfold_enabled(ctx) && common_probe_enabled(ctx)) {
        bool eligible = ss[0].n <= FOLD_MAX_FIRST;
        for (auto &q : ss) eligible = eligible && q.n > 0 && (!tax || q.t != nullptr);
        if (eligible) {
            StreamTables st(ss, false);
            WsMark pm = ws_mark(ctx);
            bool fb = true;
            const int prc = ukm_dev_probe_fold(ctx, UKM_OP_INTER, st.kp.data(), tax ? st.tp.data() : nullptr, st.ln.data(), nstreams, tax, 0,
                                               o.k, o.t, out_cap, n_out, &fb);
            ws_release(ctx, pm);
            UKM_TRY(prc);
            if (!fb) {
                if (probe_fold_done) *probe_fold_done = true;
                return UKM_OK;
            }
            *n_out = 0;
        }
    }
    if (probe_fold_done) return UKM_OK;  // (the caller only wanted the fold over plain codes)
    // first file: every code counts once (common.go:232,244) -> collapse duplicates, last wins
    if (ss[0].n) {
        bool sorted = true, strict = true;
        UKM_TRY(ukm_dev_check_sorted(ctx, ss[0].k, ss[0].n, &sorted, &strict));
        if (!strict) {
            const bool pt = tax && ss[0].t != nullptr;  // (one taxid per file: the last record's is the file's)
            u64 *k = nullptr, *k2 = nullptr;
            u32 *t = nullptr, *t2 = nullptr;
            UKM_TRY(ws_alloc_t(ctx, ss[0].n, &k));
            UKM_TRY(ws_alloc_t(ctx, ss[0].n, &k2));
            UKM_HIP(hipMemcpyAsync(k, ss[0].k, ss[0].n * sizeof(u64), hipMemcpyDeviceToDevice, ctx->stream));
            if (pt) {
                UKM_TRY(ws_alloc_t(ctx, ss[0].n, &t));
                UKM_TRY(ws_alloc_t(ctx, ss[0].n, &t2));
                UKM_HIP(hipMemcpyAsync(t, ss[0].t, ss[0].n * sizeof(u32), hipMemcpyDeviceToDevice, ctx->stream));
            }
            if (!sorted) UKM_TRY(ukm_dev_sort(ctx, k, t, ss[0].n, 64));
            u64 nu = 0;
            UKM_TRY(ukm_dev_unique_ex(ctx, k, t, ss[0].n, 5, 0, k2, t2, ss[0].n, &nu));
            ss[0] = Stream{k2, t2, nu, ss[0].ct};
        }
    }
    if (threshold > 1 && ukm_kway_enabled(ctx) && ss[0].n) {
        // files that share most of their codes with the first: one hash probe per record into tables that hold the
        // first file's codes (and claim what the later files add), a record count and the TaxId fold per entry
        // (ukm_punion.hip, pt_probe_kernel<true>).  It declines for few / small files, later files that share too
        // little with the first, an unsorted file.
        StreamTables st(ss, true);
        WsMark pm = ws_mark(ctx);
        bool fb = true;
        const int prc = ukm_dev_probe_common(ctx, st.kp.data(), tax ? st.tp.data() : nullptr, st.ln.data(), (int)st.kp.size(), tax, threshold, o.k,
                                             o.t, out_cap, n_out, &fb, true, tax ? st.cv.data() : nullptr);
        ws_release(ctx, pm);
        UKM_TRY(prc);
        if (!fb) {
            ctx->last_route = 6;
            return UKM_OK;
        }
        *n_out = 0;
    }
    UKM_TRY(materialise_all(ctx, ss, tax));  // (the merges below read a taxid per record)
    if (threshold > 1 && ukm_kway_enabled(ctx)) {
        // many files, a threshold below their number: the single-pass merge counts the records of every code inside
        // its tiles and writes only the codes that reach the threshold (ukm_srmerge.hip) -- otherwise the whole
        // merged sequence is written and read once more by the counting scan below.  It declines for few files,
        // small inputs, an unsorted file and a code with thousands of copies.
        StreamTables st(ss, true);
        if (st.kp.size() >= 3) {
            WsMark pm = ws_mark(ctx);
            bool fb = true;
            const int src = ukm_dev_srmerge(ctx, UKM_KWAY_UNION, st.kp.data(), tax ? st.tp.data() : nullptr, st.ln.data(), (int)st.kp.size(), tax,
                                            o.k, o.t, out_cap, n_out, &fb, threshold);
            ws_release(ctx, pm);
            UKM_TRY(src);
            if (!fb) {
                ctx->last_route = 5;
                return UKM_OK;
            }
            *n_out = 0;
        }
    }
    u64 *k = nullptr;
    u32 *t = nullptr;
    u64 total = 0;
    UKM_TRY(merged_sequence(ctx, ss, tax, &k, &t, &total));
    if (total == 0) return UKM_OK;
    return ukm_dev_unique_ex(ctx, k, t, total, 6, threshold ? threshold : 0, o.k, o.t, out_cap, n_out);
}

}  // namespace

extern "C" int ukm_common_ft(ukm_ctx *ctx, const uint64_t *const *keys, const uint32_t *const *taxids, const uint32_t *file_taxids,
                             const uint64_t *lens, int nstreams, uint32_t threshold, uint32_t flags,
                             uint64_t *out_keys, uint32_t *out_taxids, uint64_t out_cap, uint64_t *n_out) {
    UKM_TRY(check_common_args(ctx, keys, lens, nstreams, out_keys, out_cap, n_out, "ukm_common"));
    const bool device_streams = (flags & UKM_F_DEVICE_STREAMS) != 0;  // every stream pointer is a device pointer
    flags &= ~(uint32_t)UKM_F_DEVICE_STREAMS;
    if (nstreams > 65535) UKM_FAIL(UKM_ERR_INVALID, "ukm_common: at most 65535 streams (common.go:75-77)");
    const bool tax = any_taxids(taxids, file_taxids, nstreams);
    return run_entry(ctx, out_keys, out_taxids, out_cap, n_out, tax, [&](OutBufs &o) -> int {
        if (nstreams == 0) return UKM_OK;
        std::vector<Stream> ss;
        UKM_TRY(stage_streams(ctx, keys, taxids, file_taxids, lens, nstreams, tax, ss, device_streams));
        if (all_per_file(ss, tax) && threshold == (u32)nstreams && nstreams >= CHAIN_MIN_STREAMS) {
            // ONE taxid per file and every file needed: over duplicate-free files a code that reaches the count has met
            // every file, so its taxid is the fold over all the files' taxids (common.go:262-266) -- one value: the fold
            // over PLAIN codes, and a fill.  (The fold declines duplicates -- a code could then reach the count without
            // being in every file -- and the general routes below answer.)
            bool trivial = true;
            for (auto &q : ss) trivial = trivial && q.ct == ss[0].ct;
            if (!trivial && ctx->tax_parent == nullptr) UKM_FAIL(UKM_ERR_NO_TAXONOMY, "ukm_common: records carry taxids but no taxonomy is loaded");
            std::vector<Stream> plain = ss;
            strip_taxids(plain);
            bool done = false;
            UKM_TRY(common_body(ctx, plain, threshold, false, o, out_cap, n_out, &done));
            if (done) {
                std::vector<u32> cts((size_t)nstreams);
                for (int i = 0; i < nstreams; i++) cts[(size_t)i] = ss[(size_t)i].ct;
                u32 *plan = nullptr;
                UKM_TRY(ukm_dev_ct_plan(ctx, cts.data(), nstreams, false, &plan));
                return ukm_dev_fill_u32_from(ctx, o.t, *n_out, 0u, plan);
            }
            *n_out = 0;
        }
        return common_body(ctx, ss, threshold, tax, o, out_cap, n_out, nullptr);
    });
}

extern "C" int ukm_common(ukm_ctx *ctx, const uint64_t *const *keys, const uint32_t *const *taxids,
                          const uint64_t *lens, int nstreams, uint32_t threshold, uint32_t flags,
                          uint64_t *out_keys, uint32_t *out_taxids, uint64_t out_cap, uint64_t *n_out) {
    return ukm_common_ft(ctx, keys, taxids, nullptr, lens, nstreams, threshold, flags, out_keys, out_taxids, out_cap, n_out);
}

extern "C" int ukm_merge_k_ft(ukm_ctx *ctx, const uint64_t *const *keys, const uint32_t *const *taxids, const uint32_t *file_taxids,
                              const uint64_t *lens, int nstreams, int mode, int final_round,
                              uint64_t *out_keys, uint32_t *out_taxids, uint64_t out_cap, uint64_t *n_out) {
    UKM_TRY(check_common_args(ctx, keys, lens, nstreams, out_keys, out_cap, n_out, "ukm_merge_k"));
    if (mode != UKM_PLAIN && mode != UKM_UNIQUE && mode != UKM_REPEATED)
        UKM_FAIL(UKM_ERR_INVALID, "ukm_merge_k: mode must be UKM_PLAIN, UKM_UNIQUE or UKM_REPEATED");
    // -u over sorted streams is the union (distinct codes, LCA over every occurrence): the merge tree
    // moves 24 B per record and level instead of the 8 radix passes of concat + sort
    if (mode == UKM_UNIQUE && nstreams > 1)
        return ukm_union_ft(ctx, keys, taxids, file_taxids, lens, nstreams, 0, out_keys, out_taxids, out_cap, n_out);
    const bool tax = any_taxids(taxids, file_taxids, nstreams);
    return run_entry(ctx, out_keys, out_taxids, out_cap, n_out, tax, [&](OutBufs &o) -> int {
        std::vector<Stream> all;
        UKM_TRY(stage_streams(ctx, keys, taxids, file_taxids, lens, nstreams, tax, all));
        // util-sort.go:377-388,519-530: in a non-final round the one/two-copy protocol is kept
        int m = mode;
        if (mode == UKM_REPEATED && !final_round) m = UKM_REPEATED_CHUNK;
        if (m == UKM_REPEATED && ukm_kway_enabled(ctx)) {
            // -d in the final round = the codes that have at least two records, TaxId = LCA over all of them
            // (util-sort.go:519-530): for many files that share most of their codes the counting hash probes of
            // ukm_punion.hip with a threshold of two (every record of every file counts); it declines for few / small /
            // unsorted files and files that share little, and the merge + scan below answers.
            StreamTables st(all, true);
            if (st.kp.size() >= 3) {
                WsMark pm = ws_mark(ctx);
                bool fb = true;
                const int prc = ukm_dev_probe_common(ctx, st.kp.data(), tax ? st.tp.data() : nullptr, st.ln.data(), (int)st.kp.size(), tax, 2u, o.k, o.t,
                                                     out_cap, n_out, &fb, false, tax ? st.cv.data() : nullptr);
                ws_release(ctx, pm);
                UKM_TRY(prc);
                if (!fb) {
                    ctx->last_route = 6;
                    return UKM_OK;
                }
                *n_out = 0;
            }
        }
        {
            // every file carries the SAME one taxid (the chunk files of `sort -m` over a `count -t` file: util-sort.go writes
            // the input's global taxid into every chunk): whatever the mode folds, LCA(x, x) = x -- the PLAIN merge and a fill
            std::vector<Stream> live;
            for (auto &q : all)
                if (q.n) live.push_back(q);
            bool same = all_per_file(live, tax) && !live.empty();
            for (auto &q : live) same = same && q.ct == live[0].ct;
            if (same) {
                const u32 ct = live[0].ct;
                strip_taxids(all);
                u64 *k = nullptr;
                u32 *t = nullptr;
                u64 total = 0, need = 0;
                for (auto &s : all) need += s.n;
                const bool direct = m == UKM_PLAIN && need <= out_cap;
                UKM_TRY(merged_sequence(ctx, all, false, &k, &t, &total, direct ? o.k : nullptr, nullptr));
                if (total == 0) return UKM_OK;
                if (direct && k == o.k) *n_out = total;
                else UKM_TRY(ukm_dev_unique(ctx, k, nullptr, total, m, o.k, nullptr, out_cap, n_out));
                return ukm_dev_fill_u32(ctx, o.t, *n_out, ct);
            }
        }
        u64 *k = nullptr;
        u32 *t = nullptr;
        u64 total = 0, need = 0;
        for (auto &s : all) need += s.n;
        const bool direct = m == UKM_PLAIN && need <= out_cap;  // every record kept: merge straight into the caller's buffer
        UKM_TRY(merged_sequence(ctx, all, tax, &k, &t, &total, direct ? o.k : nullptr, direct ? o.t : nullptr));
        if (total == 0) return UKM_OK;
        if (direct && k == o.k) {
            *n_out = total;
            return UKM_OK;
        }
        return ukm_dev_unique(ctx, k, t, total, m, o.k, o.t, out_cap, n_out);
    });
}

extern "C" int ukm_merge_k(ukm_ctx *ctx, const uint64_t *const *keys, const uint32_t *const *taxids,
                           const uint64_t *lens, int nstreams, int mode, int final_round,
                           uint64_t *out_keys, uint32_t *out_taxids, uint64_t out_cap, uint64_t *n_out) {
    return ukm_merge_k_ft(ctx, keys, taxids, nullptr, lens, nstreams, mode, final_round, out_keys, out_taxids, out_cap, n_out);
}
