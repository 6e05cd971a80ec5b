// ukm_device.h — device-side building blocks for gfx950 (wave64): wave/block scans,
// single-pass decoupled look-back over tile aggregates, device LCA.
#pragma once

#include "ukm_internal.h"

#define UKM_WAVE 64

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// A u64 that an EARLIER kernel (or a host copy) wrote and this kernel only reads, fetched inside a loop that also
// stores: through a plain pointer the compiler cannot prove the word unchanged and uses a vector load, whose vmcnt(0)
// also waits for every prefetch the wave has in flight.  Read through the CONSTANT address space it becomes a scalar
// load (s_load_dwordx2/x4, lgkmcnt) that the compiler schedules early and waits for at the first use.  `p` must be
// wave-uniform and the word must not be written by the running kernel.
typedef const u64 __attribute__((address_space(4))) ukm_const_u64;
__device__ __forceinline__ u64 sload_u64(const u64 *p) { return *(ukm_const_u64 *)(uintptr_t)p; }

// A pointer that was fetched from a table in memory (streams of an n-way operation) is a GENERIC pointer to the
// compiler, and loads through it are FLAT loads: they count in vmcnt AND lgkmcnt, so every LDS wait behind them
// (s_waitcnt lgkmcnt(0)) also waits for the global data -- a prefetch that is supposed to fly during LDS work does not.
// All such pointers are device memory here: say so.
template <typename T>
using ukm_gptr = const T __attribute__((address_space(1))) *;
template <typename T>
__device__ __forceinline__ ukm_gptr<T> as_global(const T *p) {
    return (ukm_gptr<T>)(uintptr_t)p;
}

// ---- wave64 scans ---------------------------------------------------------------------------
// Inclusive scan across the 64 lanes with DPP row shifts / row broadcasts (gfx9 DPP controls:
// row_shr:n = 0x110+n, row_bcast:15 = 0x142, row_bcast:31 = 0x143): six VALU adds, no LDS
// round trip (__shfl_up lowers to ds_bpermute_b32 + s_waitcnt, ~100 cycles per step).
__device__ __forceinline__ u32 wave_incl_scan_u32(u32 v) {
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);  // row_shr:1
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);  // row_shr:2
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);  // row_shr:4
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);  // row_shr:8
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false); // row_bcast:15 -> rows 1,3
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false); // row_bcast:31 -> rows 2,3
    return v;
}

__device__ __forceinline__ u64 wave_reduce_sum_u64(u64 v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// Block-wide exclusive scan of one u32 per thread.  `smem` needs NT/64 + 1 words.
// Returns the exclusive prefix; *total gets the block total.  Contains two barriers.
template <int NT>
__device__ __forceinline__ u32 block_excl_scan_u32(u32 v, u32 *smem, u32 *total) {
    constexpr int NW = NT / 64;
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    u32 incl = wave_incl_scan_u32(v);
    if (lane == 63) smem[wave] = incl;
    __syncthreads();
    u32 wbase = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        u32 t = smem[w];
        if (w < wave) wbase += t;
        tot += t;
    }
    __syncthreads();
    *total = tot;
    return wbase + incl - v;
}

// ---- decoupled look-back ----------------------------------------------------------------------
// One 64-bit word per tile: [63:62] = state (0 empty, 1 aggregate, 2 inclusive), [61:0] value.
// Words are written/read with relaxed agent-scope atomics (one aligned 8-byte granule carries
// flag and data together, so no fence is needed; MI355X per-XCD L2s are not coherent, agent
// scope makes the accesses bypass them).  The status array must be zeroed before the launch.
//
// What the measurements on MI355X said (profiles/r01_notes.md):
//  * each tile's word sits in its OWN 64-byte line (LB_STRIDE = 8 words): several hundred
//    successor tiles poll a predecessor's word with uncached loads, and words sharing a line
//    serialised on it (look-back 6.9 us -> 4.0 us per tile when padded);
//  * the window is 64 tiles per hop (LB_W = 1): wider windows only multiplied the uncached
//    traffic and were slower;
//  * predecessors publish roughly in tile order, so the NEAREST one is the last to become
//    visible: one lane polls that single word (with a real s_sleep) and the 64-wide read is
//    issued only once it is there.
#ifndef LB_STRIDE
#define LB_STRIDE 8
#endif
#ifndef LB_W
#define LB_W 1
#endif
#ifndef LB_POLL_SLEEP
#define LB_POLL_SLEEP 8
#endif
#ifndef LB_SPIN_LIMIT
#define LB_SPIN_LIMIT (1u << 20)  /* polls before a waiter gives up (~1 s): watchdog only */
#endif
#define LB_AGG (1ull << 62)
#define LB_INCL (2ull << 62)
#define LB_VAL ((1ull << 62) - 1)

static inline size_t lb_status_words(u64 ntiles) { return (size_t)ntiles * LB_STRIDE; }

__device__ __forceinline__ void lb_store(u64 *p, u64 v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 lb_load(const u64 *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct LbHop {
    u64 w[LB_W];
};

// announce a tile's aggregate (tile 0 announces an inclusive value straight away)
__device__ __forceinline__ void lb_publish(u64 *status, u64 tile, u64 agg) {
    lb_store(&status[tile * LB_STRIDE], (tile == 0 ? LB_INCL : LB_AGG) | agg);
}

__device__ __forceinline__ void lb_hop_issue(const u64 *status, long long base, int lane, LbHop &h) {
#pragma unroll
    for (int q = 0; q < LB_W; q++) {
        const long long idx = base - q * 64 - lane;
        h.w[q] = (idx >= 0) ? lb_load(&status[idx * LB_STRIDE]) : LB_INCL;
    }
}

// returns 1 = found an inclusive value (done), 0 = window complete but no inclusive value yet
// (continue with the next window), -(q+1) = sub-window q has an unpublished predecessor.
__device__ __forceinline__ int lb_hop_eval(const LbHop &h, int lane, u64 &excl) {
    bool done = false;
    int retry_from = -1;
    u64 add = 0;
#pragma unroll
    for (int q = 0; q < LB_W; q++) {
        if (!done && retry_from < 0) {
            const u64 st = h.w[q] >> 62;
            const u64 incl_mask = __ballot(st == 2);
            const u64 empty_mask = __ballot(st == 0);
            const int first_incl = incl_mask ? __builtin_ctzll(incl_mask) : 64;
            const u64 need = (first_incl >= 63) ? ~0ull : ((2ull << first_incl) - 1);
            if (empty_mask & need) {
                retry_from = q;
            } else {
                add += (lane <= first_incl) ? (h.w[q] & LB_VAL) : 0;
                if (first_incl < 64) done = true;
            }
        }
    }
    excl += wave_reduce_sum_u64(add);
    if (done) return 1;
    if (retry_from >= 0) return -(retry_from + 1);
    return 0;
}

// One full wave calls this after lb_publish.  Returns the exclusive prefix of `tile` (same in
// every lane) and publishes the inclusive one.  *timed_out is set if a predecessor did not
// publish within LB_SPIN_LIMIT polls: only possible when tile ids come from blockIdx and the
// hardware did not dispatch workgroups in order; the caller reports it and the host re-runs
// the ticketed variant (tile ids from an atomic counter guarantee forward progress).
__device__ __forceinline__ u64 lb_resolve(u64 *status, u64 tile, u64 agg, int lane, bool *timed_out) {
    if (tile == 0) return 0;  // lb_publish already stored the inclusive value
    u64 excl = 0;
    long long base = (long long)tile - 1;
    u32 spins = 0;
    bool dead = false;
    LbHop hop;
#ifdef LB_OPTIMISTIC
    bool skip_poll = true;  // first hop: read the window straight away (one round trip when the predecessor is there)
#else
    const bool skip_poll = false;
#endif
    for (;;) {
        // wait for the nearest not-yet-counted predecessor with a single-word poll
        int ok = 1;
        if (lane == 0 && !skip_poll) {
            while ((lb_load(&status[base * LB_STRIDE]) >> 62) == 0) {
                // the watchdog is for tile ids taken from blockIdx; with ticketed ids (timed_out == nullptr)
                // every predecessor is running and the wait always ends
                if (timed_out && ++spins > LB_SPIN_LIMIT) { ok = 0; break; }
                __builtin_amdgcn_s_sleep(LB_POLL_SLEEP);
            }
        }
        if (!__shfl(ok, 0, 64)) { dead = true; break; }
        lb_hop_issue(status, base, lane, hop);
        const int r = lb_hop_eval(hop, lane, excl);
        if (r == 1) break;
        if (r == 0) base -= 64 * LB_W;
        else base -= 64 * (-r - 1);
#ifdef LB_OPTIMISTIC
        skip_poll = (r == 0);  // a complete window: try the next one directly; a hole: poll first
#endif
    }
    if (dead && timed_out) *timed_out = true;
    // publish even after a timeout so that successors drain quickly
    if (lane == 0) lb_store(&status[tile * LB_STRIDE], LB_INCL | (excl + agg));
    return excl;
}

// publish + resolve in one call (kernels that do nothing in between)
__device__ __forceinline__ u64 lb_lookback(u64 *status, u64 tile, u64 agg) {
    if (lane_id() == 0) lb_publish(status, tile, agg);
    return lb_resolve(status, tile, agg, lane_id(), nullptr);
}

// ---- device LCA (contract: include/unikmer_hip.h, ukm_taxonomy_load) ---------------------------
// LCA from the root-path table: the root paths of a and b agree on a prefix and differ behind it (or one of them
// ends: 0), so the LCA is the last equal entry.  Two INDEPENDENT 16-byte reads cover depths 0..3; the next pair is
// needed only when the first agrees completely (relatives inside one clade: cache-friendly by construction).
// Random taxid pairs diverge next to the root: one round of two loads instead of the chain of ~2 x depth dependent
// random reads of the parent / depth climb of rounds 1-2 (union of 2 x 1e8 records with random taxids 5.4 -> 3.1 ms).
// The LCA in two halves, so that a thread with several pairs can have the first-chunk reads of all of them in flight
// before it looks at any (ukm_fold.hip: up to four pairs per thread per file).
struct LcaReq {
    u32 a, b;
    uint4 pa, pb;
    u32 ca, cb;  // clade codes (when the taxonomy has them the root-path rows are fetched by lca_finish, and only for relatives)
};
__device__ __forceinline__ bool tax_has_clades(const TaxDev &T) {
#ifdef UKM_NO_CLADE
    return false;
#else
    return T.clade != nullptr || T.clade8 != nullptr;
#endif
}
__device__ __forceinline__ void lca_begin(const TaxDev &T, u32 a, u32 b, LcaReq &q) {
    const uint4 zero = make_uint4(0, 0, 0, 0);
    q.a = a;
    q.b = b;
    const bool trivial = a == 0 || b == 0 || a == b;
    q.ca = q.cb = 0;
    q.pa = q.pb = zero;
    if (tax_has_clades(T)) {  // (uniform over the launch)
        const bool go = !trivial && a < T.size && b < T.size;
        if (T.clade8) { q.ca = go ? T.clade8[a] : 0u; q.cb = go ? T.clade8[b] : 0u; }
        else { q.ca = go ? T.clade[a] : 0u; q.cb = go ? T.clade[b] : 0u; }
        return;
    }
    q.pa = (!trivial && a < T.size) ? T.anc[a] : zero;
    q.pb = (!trivial && b < T.size) ? T.anc[b] : zero;
}
// the LCA of two clade nodes with different codes: their root paths (at most four levels) part, or one of them ends
__device__ __forceinline__ u32 lca_clade_pair(const TaxDev &T, u32 ca, u32 cb) {
    if (T.pair) return T.pair[ca * T.kp + cb];  // (uniform; the one-byte form comes with the table of all pairs)
    const uint4 ta = T.top[ca], tb = T.top[cb];
    if (ta.x != tb.x) return 0;  // different trees
    if (ta.y != tb.y || ta.y == 0) return ta.x;
    if (ta.z != tb.z || ta.z == 0) return ta.y;
    return ta.z;
}
// a != b, both non-zero; pa / pb = their first root-path rows (all 0: absent or beyond the table)
__device__ __forceinline__ u32 lca_from_rows(const TaxDev &T, u32 a, u32 b, uint4 pa, uint4 pb) {
    if (pa.x == 0) {  // absent: merged into another taxid?
        const u32 m = (a < T.size && T.merged) ? T.merged[a] : 0u;
        a = (m && m < T.size) ? m : 0u;
        if (a) pa = T.anc[a];
        if (pa.x == 0) return 0;
    }
    if (pb.x == 0) {
        const u32 m = (b < T.size && T.merged) ? T.merged[b] : 0u;
        b = (m && m < T.size) ? m : 0u;
        if (b) pb = T.anc[b];
        if (pb.x == 0) return 0;
    }
    if (a == b) return a;
    if (pa.x != pb.x) return 0;  // different trees
    u32 last = pa.x;
    for (u32 c = 0;;) {
        if (pa.y != pb.y || pa.y == 0) return last;
        last = pa.y;
        if (pa.z != pb.z || pa.z == 0) return last;
        last = pa.z;
        if (pa.w != pb.w || pa.w == 0) return last;
        last = pa.w;
        if (++c == T.nchunks) return last;
        pa = T.anc[(size_t)c * T.size + a];
        pb = T.anc[(size_t)c * T.size + b];
        if (pa.x != pb.x || pa.x == 0) return last;
        last = pa.x;
    }
}
__device__ __forceinline__ u32 lca_finish(const TaxDev &T, const LcaReq &q) {
    const u32 a = q.a, b = q.b;
    if (a == 0 || b == 0) return 0;
    if (a == b) return a;
    uint4 pa = q.pa, pb = q.pb;
    if (tax_has_clades(T)) {
        if (q.ca != q.cb && q.ca != 0 && q.cb != 0) return lca_clade_pair(T, q.ca, q.cb);  // unrelated
        const uint4 zero = make_uint4(0, 0, 0, 0);
        pa = a < T.size ? T.anc[a] : zero;
        pb = b < T.size ? T.anc[b] : zero;
    }
    return lca_from_rows(T, a, b, pa, pb);
}
// ---- the clade-pair step out of LDS (round 6) ----------------------------------------------------------------------
// A divergent gather costs a 128-byte line of L1 fill whatever it returns; of the three an unrelated pair used to take
// (two clade bytes, one pair word) the pair word was 46 % of the time (2 x 3e8 records, union with random taxids: 5.75 ms,
// 4.28 with the pair word computed instead of read, 2.54 without any look-up).  The pair step now reads two 16-byte rows
// and one word of a 5 KB table the workgroup holds in LDS (TaxDev::cpath / cnode).
struct CladeLds {
    uint4 path[TAX_CPATH_ROWS];
    u32 node[TAX_CPATH_ROWS];
};
template <bool ON> struct CladeLdsOpt { CladeLds t; };  // (a kernel template's LDS member: nothing when it has no taxids)
template <> struct CladeLdsOpt<false> { u32 t; };
// (every thread of the workgroup; the caller's next barrier publishes the table)
__device__ __forceinline__ void clade_lds_load(const TaxDev &T, CladeLds &L, int tid, int nthreads) {
    if (T.cpath == nullptr) return;  // (uniform)
    for (u32 i = (u32)tid; i < TAX_CPATH_ROWS; i += (u32)nthreads) {
        L.path[i] = T.cpath[i];
        L.node[i] = T.cnode[i];
    }
}
// ca != cb, both non-zero
__device__ __forceinline__ u32 lca_clade_pair_lds(const TaxDev &T, const CladeLds &L, u32 ca, u32 cb) {
    const uint4 A = L.path[ca], B = L.path[cb];
    const u64 al = ((u64)A.y << 32) | A.x, ah = ((u64)A.w << 32) | A.z;
    const u64 xl = al ^ (((u64)B.y << 32) | B.x), xh = ah ^ (((u64)B.w << 32) | B.z);
    if ((xl | xh) == 0) return T.pair[ca * T.kp + cb];  // 16 shared levels: both far down one chain (rare)
    const bool in_lo = xl != 0;
    const int n = (__builtin_ctzll(in_lo ? xl : xh) >> 3) + (in_lo ? 0 : 8);  // levels the two root paths share
    if (n == 0) return 0;                                                     // different trees
    const int m = n - 1;
    const u32 code = (u32)((m < 8 ? al : ah) >> (8 * (m & 7))) & 255u;
    return L.node[code];
}
// lca_dev with the pair step out of LDS (the caller has checked T.cpath != nullptr: the one-byte codes are loaded)
__device__ __forceinline__ u32 lca_dev_lds(const TaxDev &T, const CladeLds &L, u32 a, u32 b) {
    if (a == 0 || b == 0) return 0;
    if (a == b) return a;
    const uint4 zero = make_uint4(0, 0, 0, 0);
    if (a < T.size && b < T.size) {
        const u32 ca = T.clade8[a], cb = T.clade8[b];
        if (ca != cb && ca != 0 && cb != 0) return lca_clade_pair_lds(T, L, ca, cb);
    }
    return lca_from_rows(T, a, b, a < T.size ? T.anc[a] : zero, b < T.size ? T.anc[b] : zero);
}
// With clade codes: unrelated pairs -- different codes -- are settled by two small reads and two rows of the `top` table,
// relatives (and absent / merged ids) take the root paths behind them.
__device__ __forceinline__ u32 lca_dev(const TaxDev &T, u32 a, u32 b) {
    if (a == 0 || b == 0) return 0;
    if (a == b) return a;
    const uint4 zero = make_uint4(0, 0, 0, 0);
    if (tax_has_clades(T) && a < T.size && b < T.size) {
        u32 ca, cb;
        if (T.clade8) { ca = T.clade8[a]; cb = T.clade8[b]; }  // (uniform over the launch)
        else { ca = T.clade[a]; cb = T.clade[b]; }
        if (ca != cb && ca != 0 && cb != 0) return lca_clade_pair(T, ca, cb);
    }
    return lca_from_rows(T, a, b, a < T.size ? T.anc[a] : zero, b < T.size ? T.anc[b] : zero);
}
