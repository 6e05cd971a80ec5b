// ukm_setops.hip — 2-way sorted set operations (union / inter / diff) on device-resident
// (code uint64 [, taxid uint32]) streams: the MI355X replacement for the reference's per-k-mer
// hash-map / 2-pointer loops (union.go:186-208, inter.go:205-278, diff.go:379-454).
//
// Design (HBM-bound integer path, no MFMA):
//   1. partition kernel  — one thread per tile boundary binary-searches the merge path
//                          (A before B on ties) -> mp[t]; ~30 dependent loads, massively parallel.
//   2. tile kernel       — one 256-thread workgroup per tile of TILE = NT*VT merged items:
//        * coalesced loads of the tile's A range and B range (+1-element halos) into LDS,
//        * per-thread merge-path search in LDS, then a VT-step serial merge that decides
//          emit/skip per item from the neighbouring element (sets: an equal pair is adjacent),
//          checking strict sortedness of both inputs on the fly,
//        * wave64 shuffle scan + LDS for the block prefix, single-pass decoupled look-back
//          over tile aggregates (ticketed tile ids, agent-scope 8-byte status words) for the
//          global output offset — inputs are read once and outputs written once,
//        * survivors compacted through LDS and stored as contiguous coalesced runs.
//   Algorithmic bytes per launch: 8(|A|+|B|) read + 8|out| written (12 B/record with taxids).
//   Multisets (duplicate codes inside an input; legal for `inter`/`diff`, inter.go:198) are
//   detected by the fast path and re-run on (code, rank-in-run) pairs, which reproduces the
//   reference's "equality advances both cursors" semantics exactly.
#include <algorithm>

#include "ukm_device.h"

namespace {

enum { FLAG_DUP = 1, FLAG_UNSORTED = 2, FLAG_TIMEOUT = 4 };

#ifndef SETOP_NT
#define SETOP_NT 512
#endif
#ifndef SETOP_VT
#define SETOP_VT 19
#endif


struct SetopArgs {
    const u64 *a, *b;
    const u32 *ta, *tb;
    const u32 *ra, *rb;
    u64 na, nb;
    u64 *mp;      // [ntiles + 1]
    u64 *status;  // [ntiles]
    u32 *ticket;
    u64 *result;  // [0] total, [1] flags
    u64 *out;
    u32 *tout;
    u64 out_cap;
    u64 ntiles;
    TaxDev tax;
    u32 flags;
    u64 *dbg;  // UKM_PROFILE_PHASES only
    // chained folds (ukm_inter / ukm_diff over many files): |A| is the previous call's result count and stays
    // on the device; na / ntiles above are then upper bounds used for the launch geometry only
    const u64 *na_dev;
    u32 zero_status;  // chained links with few tiles: the partition kernel clears this many status lines (one launch less)
    // Per-FILE taxids (round 5; the .unik header's global taxid, count.go:466-468: the reader hands the same value to the
    // set operation for every record of the file): a stream whose taxid pointer is null carries `cta` / `ctb` in every
    // record.  One such stream beside per-record taxids: the tile loader fills the constant in (no loads).  BOTH streams
    // constant: the plain-key kernel runs (CT instantiation: LDS-DMA staging, 19 items per thread, no taxid in LDS) and the
    // output taxid is one of three values -- A's, B's, or LCA(A's, B's) on a match -- resolved ONCE by setop_ct_kernel into
    // result[4] = lca | keep << 32 (keep: diff -t leaves a matched code in the result, diff.go:404-409).
    u32 cta, ctb;
    // DEFER instantiation (round 6): the pairs a tile does not settle itself -- relatives inside one clade, unknown and merged
    // ids: the root-path reads -- go to the tile's FIX_SLOTS places of this list as (output position, A's taxid, B's taxid) and
    // setop_taxid_fix_kernel settles them behind the launch; fix_cnt[tile] = how many (every tile writes it).  nullptr: none.
    uint4 *fix;
    u32 *fix_cnt;
};
constexpr u32 FIX_SLOTS = 16;

// the actual sizes of a chained call (workgroup-uniform: one scalar load)
__device__ __forceinline__ void setop_resolve_sizes(SetopArgs &p, u64 tile_items) {
    if (p.na_dev) {
        p.na = *p.na_dev;
        p.ntiles = (p.na + p.nb + tile_items - 1) / tile_items;
    }
}

#ifdef UKM_PROFILE_PHASES
#define PH(i) do { if (tid == 0) { u64 _t = clock64(); ph[i] += _t - tlast; tlast = _t; } } while (0)
#else
#define PH(i) do {} while (0)
#endif

template <bool RANK>
__device__ __forceinline__ bool key_le(u64 ka, u32 ra, u64 kb, u32 rb) {
    if (RANK) return ka < kb || (ka == kb && ra <= rb);
    return ka <= kb;
}
template <bool RANK>
__device__ __forceinline__ bool key_eq(u64 ka, u32 ra, u64 kb, u32 rb) {
    if (RANK) return ka == kb && ra == rb;
    return ka == kb;
}

// Merge-path split of every tile boundary.  With 2.4e5 tiles a 30-step search over the whole
// inputs per boundary is ~7e6 dependent random reads across 16 GB (0.3 ms, TLB-miss bound), so
// the search is done in two levels: LEVEL 1 places every PART_COARSE-th boundary (and the last
// one) with a full search, LEVEL 2 searches the others only between their two coarse neighbours
// (the path is monotone), i.e. inside a few MB that the group's threads share in cache.
// LEVEL 0 = single-level search of every boundary (small inputs).
#ifndef SETOP_PART_COARSE
#define SETOP_PART_COARSE 64
#endif
#ifndef SETOP_PART_INTERP1
#define SETOP_PART_INTERP1 (1 << 17)  // the same for the coarse level (whole-input diagonal): +-1 MB of keys
#endif
#ifndef SETOP_PART_INTERP
#define SETOP_PART_INTERP 512  // half-width of the bracket around the interpolated split (0: plain binary search)
#endif
constexpr int PART_COARSE = SETOP_PART_COARSE;
template <bool RANK, int LEVEL>
__global__ void setop_partition_kernel(SetopArgs p, int tile_items) {
    setop_resolve_sizes(p, (u64)tile_items);
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (LEVEL == 1) {
        t *= PART_COARSE;
        if (t > p.ntiles + PART_COARSE - 1) return;
        if (t > p.ntiles) t = p.ntiles;
    } else {
        if (LEVEL == 0 && t < p.zero_status) p.status[t * LB_STRIDE] = 0;
        if (t > p.ntiles) return;
        if (LEVEL == 2 && (t % PART_COARSE == 0 || t == p.ntiles)) return;  // placed by level 1
    }
    const u64 N = p.na + p.nb;
    u64 diag = t * (u64)tile_items;
    if (diag > N) diag = N;
    u64 lo = diag > p.nb ? diag - p.nb : 0;
    u64 hi = diag < p.na ? diag : p.na;
    if (LEVEL == 2) {
        const u64 c0 = t / PART_COARSE * PART_COARSE;
        const u64 c1 = (c0 + PART_COARSE < p.ntiles) ? c0 + PART_COARSE : p.ntiles;
        const u64 l0 = p.mp[c0], h0 = p.mp[c1];
        lo = lo > l0 ? lo : l0;
        hi = hi < h0 ? hi : h0;
#if SETOP_PART_INTERP
        // The path between two coarse neighbours is close to a straight line when the keys are spread evenly (k-mer
        // codes, hashes): two probes SETOP_PART_INTERP positions either side of the interpolated split usually bracket
        // it, and the search that follows stays inside a few KB (the probes of a plain binary search over the window
        // each touch another page).  Whatever the probes say narrows [lo, hi] correctly, so skewed inputs only lose
        // the two probes.
        if (lo < hi) {
            const u64 W = SETOP_PART_INTERP;
            u64 est = l0 + (h0 - l0) * (t - c0) / (c1 - c0);
            est = est < lo ? lo : (est > hi ? hi : est);
            const u64 L = est > lo + W ? est - W : lo;
            const u64 R = est + W < hi ? est + W : hi;
            bool pl = true, pr = false;
            if (L > lo) { const u64 j = diag - L; pl = key_le<RANK>(p.a[L - 1], RANK ? p.ra[L - 1] : 0, p.b[j], RANK ? p.rb[j] : 0); }
            if (R < hi) { const u64 j = diag - 1 - R; pr = key_le<RANK>(p.a[R], RANK ? p.ra[R] : 0, p.b[j], RANK ? p.rb[j] : 0); }
            if (L > lo) { if (pl) lo = L; else hi = L - 1; }
            if (R < hi) { if (!pr) hi = R; else lo = R + 1; }  // (R < hi fails when the left probe already cut below R)
        }
#endif
    }
    while (lo < hi) {
        u64 mid = (lo + hi) >> 1;
        u64 j = diag - 1 - mid;
        bool le = key_le<RANK>(p.a[mid], RANK ? p.ra[mid] : 0, p.b[j], RANK ? p.rb[j] : 0);
        if (le) lo = mid + 1; else hi = mid;
    }
    p.mp[t] = lo;
}

// Wave-cooperative variant of levels 0 and 1: ONE WAVE per boundary, 64-ary search.  Each round the 64 lanes
// probe 64 split candidates at once (the predicate is monotone along the diagonal, so the true lanes form a
// prefix and a ballot + popcount narrows [lo, hi) to one of 65 sub-ranges): log64 instead of log2 dependent
// round trips -- 4-5 instead of 20-30 for the few boundaries of a small input or of the coarse level, whose
// cost is pure latency (measured: level 1 at 2 x 1e9 40 us, the single-level kernel of a 2 x 1e6 call 12 us).
template <bool RANK, int LEVEL>
__global__ void setop_partition_coop_kernel(SetopArgs p, int tile_items) {
    static_assert(LEVEL == 0 || LEVEL == 1, "bulk level 2 stays one thread per boundary");
    setop_resolve_sizes(p, (u64)tile_items);
    const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (LEVEL == 0 && p.zero_status) {
        // status lines of the launch's upper bound of tiles: 64 per wave
        const u64 line = gid;
        if (line < p.zero_status) p.status[line * LB_STRIDE] = 0;
    }
    u64 t = gid >> 6;
    const int lane = (int)(threadIdx.x & 63);
    if (LEVEL == 1) {
        t *= PART_COARSE;
        if (t > p.ntiles + PART_COARSE - 1) return;
        if (t > p.ntiles) t = p.ntiles;
    } else if (t > p.ntiles) {
        return;
    }
    const u64 N = p.na + p.nb;
    u64 diag = t * (u64)tile_items;
    if (diag > N) diag = N;
    u64 lo = diag > p.nb ? diag - p.nb : 0;
    u64 hi = diag < p.na ? diag : p.na;
#if SETOP_PART_INTERP1
    if (lo < hi) {
        // evenly spread keys: the split of diagonal d lies near d * |A| / (|A| + |B|); two (wave-uniform) probes either
        // side of it cut two or three of the five 64-ary rounds.  Skewed inputs only lose the probes.
        const u64 W = SETOP_PART_INTERP1;
        u64 est = (u64)((double)diag * ((double)p.na / (double)N));
        est = est < lo ? lo : (est > hi ? hi : est);
        const u64 L = est > lo + W ? est - W : lo;
        const u64 R = est + W < hi ? est + W : hi;
        bool pl = true, pr = false;
        if (L > lo) { const u64 j = diag - L; pl = key_le<RANK>(p.a[L - 1], RANK ? p.ra[L - 1] : 0, p.b[j], RANK ? p.rb[j] : 0); }
        if (R < hi) { const u64 j = diag - 1 - R; pr = key_le<RANK>(p.a[R], RANK ? p.ra[R] : 0, p.b[j], RANK ? p.rb[j] : 0); }
        if (L > lo) { if (pl) lo = L; else hi = L - 1; }
        if (R < hi) { if (!pr) hi = R; else lo = R + 1; }
    }
#endif
    while (lo < hi) {  // wave-uniform
        const u64 span = hi - lo;
        // candidates: strictly increasing positions in [lo, hi); fewer than 64 when the span is short
        const u64 step_n = span <= 64 ? 1 : 0;
        const u64 cand = step_n ? lo + (u64)lane
                                : lo + (span / 65) * (u64)(lane + 1) + ((span % 65) * (u64)(lane + 1)) / 65;  // = lo + span*(lane+1)/65, no overflow
        const bool active = cand < hi;
        bool le = false;
        if (active) {
            const u64 j = diag - 1 - cand;
            le = key_le<RANK>(p.a[cand], RANK ? p.ra[cand] : 0, p.b[j], RANK ? p.rb[j] : 0);
        }
        const u64 m_le = __ballot(le), m_act = __ballot(active);
        const int n_true = __popcll(m_le);            // lanes 0 .. n_true-1 are true (monotone)
        const int n_act = __popcll(m_act);
        // first false candidate = lane n_true (if active) -> new hi; last true candidate -> new lo
        const u64 c_last_true = __shfl(cand, n_true > 0 ? n_true - 1 : 0, 64);
        const u64 c_first_false = __shfl(cand, n_true < 64 ? n_true : 63, 64);
        const u64 nlo = n_true > 0 ? c_last_true + 1 : lo;
        const u64 nhi = n_true < n_act ? c_first_false : hi;
        lo = nlo;
        hi = nhi;
    }
    if (lane == 0) p.mp[t] = lo;
}

// Both levels AND the clearing of the status lines in ONE launch (round 4): a workgroup owns one coarse segment of
// PART_COARSE boundaries.  Waves 0 and 1 place the segment's two coarse ends with the 64-ary search (every coarse end is
// found twice, by its two neighbouring segments: ~10 of them per CU, pure latency), the other boundaries are then searched
// between the two by one thread each, and the segment's status lines (and, by segment 0, the control words) are zeroed on
// the way.  Replaces hipMemsetAsync + the level-1 kernel + the level-2 kernel in front of every tile kernel of a plain
// call: three launches and two dependent kernel tails less (2 x 1e9 codes: 0.19 -> see profiles/r04_notes.md).
template <bool RANK>
__global__ __launch_bounds__(256) void setop_partition_fused_kernel(SetopArgs p, int tile_items, u32 nclear) {
    __shared__ u64 s_end[2];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64 c0 = (u64)blockIdx.x * PART_COARSE;
    const u64 c1 = (c0 + PART_COARSE < p.ntiles) ? c0 + PART_COARSE : p.ntiles;
    const u64 N = p.na + p.nb;
    // status lines of this segment's tiles; the control words in front of them
    if (tid < PART_COARSE && c0 + (u64)tid < p.ntiles) p.status[(c0 + (u64)tid) * LB_STRIDE] = 0;
    if (blockIdx.x == 0 && tid < (int)nclear) p.result[tid] = 0;
    if (wave < 2) {
        const u64 t = wave == 0 ? c0 : c1;
        u64 diag = t * (u64)tile_items;
        if (diag > N) diag = N;
        u64 lo = diag > p.nb ? diag - p.nb : 0;
        u64 hi = diag < p.na ? diag : p.na;
#if SETOP_PART_INTERP1
        if (lo < hi) {
            const u64 W = SETOP_PART_INTERP1;
            u64 est = (u64)((double)diag * ((double)p.na / (double)N));
            est = est < lo ? lo : (est > hi ? hi : est);
            const u64 L = est > lo + W ? est - W : lo;
            const u64 R = est + W < hi ? est + W : hi;
            bool pl = true, pr = false;
            if (L > lo) { const u64 j = diag - L; pl = key_le<RANK>(p.a[L - 1], RANK ? p.ra[L - 1] : 0, p.b[j], RANK ? p.rb[j] : 0); }
            if (R < hi) { const u64 j = diag - 1 - R; pr = key_le<RANK>(p.a[R], RANK ? p.ra[R] : 0, p.b[j], RANK ? p.rb[j] : 0); }
            if (L > lo) { if (pl) lo = L; else hi = L - 1; }
            if (R < hi) { if (!pr) hi = R; else lo = R + 1; }
        }
#endif
        while (lo < hi) {  // wave-uniform 64-ary search (see setop_partition_coop_kernel)
            const u64 span = hi - lo;
            const u64 cand = span <= 64 ? lo + (u64)lane
                                        : lo + (span / 65) * (u64)(lane + 1) + ((span % 65) * (u64)(lane + 1)) / 65;
            const bool active = cand < hi;
            bool le = false;
            if (active) {
                const u64 j = diag - 1 - cand;
                le = key_le<RANK>(p.a[cand], RANK ? p.ra[cand] : 0, p.b[j], RANK ? p.rb[j] : 0);
            }
            const u64 m_le = __ballot(le), m_act = __ballot(active);
            const int n_true = __popcll(m_le), n_act = __popcll(m_act);
            const u64 c_last_true = __shfl(cand, n_true > 0 ? n_true - 1 : 0, 64);
            const u64 c_first_false = __shfl(cand, n_true < 64 ? n_true : 63, 64);
            const u64 nlo = n_true > 0 ? c_last_true + 1 : lo;
            const u64 nhi = n_true < n_act ? c_first_false : hi;
            lo = nlo;
            hi = nhi;
        }
        if (lane == 0) {
            s_end[wave] = lo;
            if (wave == 0) p.mp[c0] = lo;
            else if (c1 == p.ntiles) p.mp[c1] = lo;  // the last boundary has no segment of its own
        }
    }
    __syncthreads();
    const u64 t = c0 + (u64)tid;
    if (tid == 0 || tid >= PART_COARSE || t >= c1) return;
    const u64 l0 = s_end[0], h0 = s_end[1];
    u64 diag = t * (u64)tile_items;
    if (diag > N) diag = N;
    u64 lo = diag > p.nb ? diag - p.nb : 0;
    u64 hi = diag < p.na ? diag : p.na;
    lo = lo > l0 ? lo : l0;
    hi = hi < h0 ? hi : h0;
#if SETOP_PART_INTERP
    if (lo < hi) {
        const u64 W = SETOP_PART_INTERP;
        u64 est = l0 + (h0 - l0) * (t - c0) / (c1 - c0);
        est = est < lo ? lo : (est > hi ? hi : est);
        const u64 L = est > lo + W ? est - W : lo;
        const u64 R = est + W < hi ? est + W : hi;
        bool pl = true, pr = false;
        if (L > lo) { const u64 j = diag - L; pl = key_le<RANK>(p.a[L - 1], RANK ? p.ra[L - 1] : 0, p.b[j], RANK ? p.rb[j] : 0); }
        if (R < hi) { const u64 j = diag - 1 - R; pr = key_le<RANK>(p.a[R], RANK ? p.ra[R] : 0, p.b[j], RANK ? p.rb[j] : 0); }
        if (L > lo) { if (pl) lo = L; else hi = L - 1; }
        if (R < hi) { if (!pr) hi = R; else lo = R + 1; }
    }
#endif
    while (lo < hi) {
        const u64 mid = (lo + hi) >> 1;
        const u64 j = diag - 1 - mid;
        const bool le = key_le<RANK>(p.a[mid], RANK ? p.ra[mid] : 0, p.b[j], RANK ? p.rb[j] : 0);
        if (le) lo = mid + 1; else hi = mid;
    }
    p.mp[t] = lo;
}

// UKM_OP_MERGE_INTERNAL (ukm_internal.h): plain 2-way MERGE of two non-decreasing streams, every record
// kept (A first on ties).  Output size is known (|A| + |B|), so the tile kernel needs no look-back for it.

// ---- tile building blocks (shared by the kernels below) -------------------------------------------
// Everything here is written branch-free on purpose: the first version of the merge step
// compiled to ~80 instructions (exec-mask juggling, phi copies, 64-bit selects) and the kernel
// was VALU/SALU-issue bound at 31 % of HBM peak.
struct TileGeom {
    u64 a0, b0;
    int na_t, nb_t;
    bool has_prev_a, has_prev_b, has_next_a, has_next_b;
    // LDS slots:  [sa0] prevA | A items [base_a, end_a) | [end_a] nextA | pad | [sb0] prevB |
    //             B items [base_b, end_b) | [end_b] nextB
    // sa0 / sb0 carry a one-slot parity offset so that an EVEN slot always holds an element
    // whose global address is 16-byte aligned: pairs of slots then move with one dwordx4 load,
    // one ds_write_b128 and (on the way out) one dwordx4 store.
    int sa0, base_a, end_a, sb0, base_b, end_b, split;  // split = first (even) slot of region B
};

template <int NTH, int VT>
__device__ __forceinline__ TileGeom tile_geom(const SetopArgs &p, u64 tile) {
    constexpr int TILE = NTH * VT;
    const u64 N = p.na + p.nb;
    const u64 d0 = tile * (u64)TILE;
    const u64 d1 = (d0 + TILE < N) ? d0 + TILE : N;
    const u64 a0 = p.mp[tile], a1 = p.mp[tile + 1];
    const u64 b0 = d0 - a0, b1 = d1 - a1;
    TileGeom g;
    g.a0 = a0; g.b0 = b0;
    g.na_t = (int)(a1 - a0); g.nb_t = (int)(b1 - b0);
    g.has_prev_a = a0 > 0; g.has_prev_b = b0 > 0;
    g.has_next_a = a1 < p.na; g.has_next_b = b1 < p.nb;
    // parity (in 8-byte units) of the address of a[a0 - 1] / b[b0 - 1]
    g.sa0 = (int)((((uintptr_t)p.a >> 3) + a0 + 1) & 1);
    g.base_a = g.sa0 + 1;
    g.end_a = g.base_a + g.na_t;
    g.split = (g.end_a + 2) & ~1;
    g.sb0 = g.split + (int)((((uintptr_t)p.b >> 3) + b0 + 1) & 1);
    g.base_b = g.sb0 + 1;
    g.end_b = g.base_b + g.nb_t;
    return g;
}

// pairs of LDS slots each thread moves: ceil((TILE + 8) / 2 / NTH)
template <int NTH, int VT> struct TilePairs { static constexpr int NP = ((NTH * VT + 8) / 2 + NTH - 1) / NTH; };

// Coalesced global loads of the tile (+halos) into registers, two slots per load; nothing is
// waited for here.
template <bool TAX, bool RANK, int NTH, int VT>
__device__ __forceinline__ void tile_load(const SetopArgs &p, const TileGeom &g, int tid,
                                          u64 (&rk)[2 * TilePairs<NTH, VT>::NP], u32 (&rt)[2 * TilePairs<NTH, VT>::NP],
                                          u32 (&rr)[2 * TilePairs<NTH, VT>::NP]) {
    constexpr int NP = TilePairs<NTH, VT>::NP;
    const int alo = g.sa0 + (g.has_prev_a ? 0 : 1), ahi = g.end_a + (g.has_next_a ? 1 : 0);
    const int blo = g.sb0 + (g.has_prev_b ? 0 : 1), bhi = g.end_b + (g.has_next_b ? 1 : 0);
    // slot s of region A is a[a0 - 1 + (s - sa0)]; slot s of region B is b[b0 - 1 + (s - sb0)]
    const u64 *pa = p.a + g.a0 - 1 - g.sa0;
    const u64 *pb = p.b + g.b0 - 1 - g.sb0;
    const u32 *pta = TAX && p.ta ? p.ta + g.a0 - 1 - g.sa0 : nullptr;
    const u32 *ptb = TAX && p.tb ? p.tb + g.b0 - 1 - g.sb0 : nullptr;
    const u32 *pra = RANK ? p.ra + g.a0 - 1 - g.sa0 : nullptr;
    const u32 *prb = RANK ? p.rb + g.b0 - 1 - g.sb0 : nullptr;
    // Every load below is UNCONDITIONAL (idle lanes read a harmless aligned word of the control
    // block): a branch around a load makes the compiler wait for the previous one at the join
    // (measured: vmcnt(0) in front of every dwordx4, kernel 30 % slower).
    const u64 *safe = p.result;  // 256-byte aligned, always mapped
    const u32 *safe32 = reinterpret_cast<const u32 *>(p.result);
#pragma unroll
    for (int j = 0; j < NP; j++) {
        const int s0 = 2 * (tid + j * NTH), s1 = s0 + 1;
        const bool in_a = s0 < g.split;
        const int lo = in_a ? alo : blo, hi = in_a ? ahi : bhi;
        const bool v0 = s0 >= lo && s0 < hi, v1 = s1 >= lo && s1 < hi;
        const bool both = v0 && v1;
        const u64 *src = in_a ? pa : pb;
        // (per-record taxids: the streams go past L2 with the non-temporal hint, in and out, so that the clade table stays in
        //  it -- union of 2 x 3e8 with random taxids 5.75 -> 5.03 ms, of which the stores are 0.5; the plain kernel, which
        //  gathers nothing, was slower with the hint: profiles/r01_notes.md)
        ulonglong2 q;
        if (TAX) {
            typedef u64 v2u64 __attribute__((ext_vector_type(2)));
            const v2u64 w = __builtin_nontemporal_load(reinterpret_cast<const v2u64 *>(both ? src + s0 : safe));
            q.x = w.x;
            q.y = w.y;
        } else {
            q = *reinterpret_cast<const ulonglong2 *>(both ? src + s0 : safe);  // aligned by construction
        }
        rk[2 * j] = q.x;      // slots without a real element keep whatever was read: they are
        rk[2 * j + 1] = q.y;  // never compared (order check and merge only touch real slots)
        u32 t0 = 0, t1 = 0, r0 = 0, r1 = 0;
        if (TAX) {
            const u32 *ts = in_a ? pta : ptb;
            const bool h0 = ts && v0, h1 = ts && v1;
            t0 = __builtin_nontemporal_load(h0 ? ts + s0 : safe32);
            t1 = __builtin_nontemporal_load(h1 ? ts + s1 : safe32);
            const u32 tc = in_a ? p.cta : p.ctb;  // a stream without per-record taxids carries its file's taxid (0: none, mix-taxid)
            t0 = ts ? t0 : tc;
            t1 = ts ? t1 : tc;
        }
        if (RANK) {
            const u32 *rs = in_a ? pra : prb;
            r0 = *(v0 ? rs + s0 : safe32);
            r1 = *(v1 ? rs + s1 : safe32);
        }
        rt[2 * j] = t0; rt[2 * j + 1] = t1;
        rr[2 * j] = r0; rr[2 * j + 1] = r1;
    }
    // the (at most four) pairs per tile with exactly one real element, at the region edges
#pragma unroll
    for (int j = 0; j < NP; j++) {
        const int s0 = 2 * (tid + j * NTH), s1 = s0 + 1;
        const bool in_a = s0 < g.split;
        const int lo = in_a ? alo : blo, hi = in_a ? ahi : bhi;
        const bool v0 = s0 >= lo && s0 < hi, v1 = s1 >= lo && s1 < hi;
        if (v0 != v1) {
            const u64 *src = in_a ? pa : pb;
            if (v0) rk[2 * j] = src[s0]; else rk[2 * j + 1] = src[s1];
        }
    }
}

template <bool TAX, bool RANK, int NTH, int VT>
__device__ __forceinline__ void tile_to_lds(int tid, const u64 (&rk)[2 * TilePairs<NTH, VT>::NP],
                                            const u32 (&rt)[2 * TilePairs<NTH, VT>::NP],
                                            const u32 (&rr)[2 * TilePairs<NTH, VT>::NP], u64 *s_keys, u32 *s_tax,
                                            u32 *s_rank) {
    constexpr int NP = TilePairs<NTH, VT>::NP;
    constexpr int SLOTS = NTH * VT + 8;
#pragma unroll
    for (int j = 0; j < NP; j++) {
        const int s0 = 2 * (tid + j * NTH);
        if (s0 < SLOTS) {
            *reinterpret_cast<ulonglong2 *>(s_keys + s0) = make_ulonglong2(rk[2 * j], rk[2 * j + 1]);
            if (TAX) *reinterpret_cast<uint2 *>(s_tax + s0) = make_uint2(rt[2 * j], rt[2 * j + 1]);
            if (RANK) *reinterpret_cast<uint2 *>(s_rank + s0) = make_uint2(rr[2 * j], rr[2 * j + 1]);
        }
    }
}

// Strict-order check of both inputs as one vector pass (call after the barrier that follows
// tile_to_lds).  A slot is compared with its predecessor only when both hold real elements of
// the same input; the in-pair comparison uses the registers, the cross-pair one a single LDS
// read.  Replaces two 64-bit compares per merge step.
template <bool RANK, int NTH, int VT>
__device__ __forceinline__ u32 tile_check_order(const TileGeom &g, int tid, const u64 (&rk)[2 * TilePairs<NTH, VT>::NP],
                                                const u32 (&rr)[2 * TilePairs<NTH, VT>::NP], const u64 *s_keys,
                                                const u32 *s_rank) {
    constexpr int NP = TilePairs<NTH, VT>::NP;
    const int alo = g.sa0 + (g.has_prev_a ? 0 : 1), ahi = g.end_a + (g.has_next_a ? 1 : 0);
    const int blo = g.sb0 + (g.has_prev_b ? 0 : 1), bhi = g.end_b + (g.has_next_b ? 1 : 0);
    u32 bad = 0;
#pragma unroll
    for (int j = 0; j < NP; j++) {
        const int s0 = 2 * (tid + j * NTH), s1 = s0 + 1;
        const bool in_a = s0 < g.split;
        const int lo = in_a ? alo : blo, hi = in_a ? ahi : bhi;
        // (s-1, s) is a real adjacent pair of one input iff lo < s < hi
        const bool c0 = s0 > lo && s0 < hi, c1 = s1 > lo && s1 < hi;
        const int ip = c0 ? s0 - 1 : 0;
        const u64 pk = s_keys[ip], k0 = rk[2 * j], k1 = rk[2 * j + 1];
        if (RANK) {
            const u32 pr = s_rank[ip], r0 = rr[2 * j], r1 = rr[2 * j + 1];
            bad |= (c0 & ((pk > k0) | ((pk == k0) & (pr >= r0)))) ? FLAG_UNSORTED : 0u;
            bad |= (c1 & ((k0 > k1) | ((k0 == k1) & (r0 >= r1)))) ? FLAG_UNSORTED : 0u;
        } else {
            bad |= ((c0 & (pk > k0)) | (c1 & (k0 > k1))) ? FLAG_UNSORTED : 0u;
            bad |= ((c0 & (pk == k0)) | (c1 & (k0 == k1))) ? FLAG_DUP : 0u;
        }
    }
    return bad;
}

// ---- fast load / order check for tiles well inside both inputs (plain keys only) ---------------------
// When a0 >= 2, a1 + 2 <= na, b0 >= 2 and b1 + 5 <= nb, every slot of region A [0, split) and of
// region B [split, SLOTS) can be filled from the input itself: the slots that are padding in the
// generic layout (slot 0 under odd parity, the gap in front of region B, the tail) then hold the
// real neighbours a[a0-2], a[a1+1], b[b0-2], b[b1+1 ...].  Each region is a run of adjacent input
// elements, so neither the loads nor the order check need per-slot validity logic (about 200 of
// the 1200 VALU instructions a wave spends on a tile).  A violation seen in an over-read element
// is a real violation of the input (the neighbouring tile reports it too).
template <int NTH, int VT>
__device__ __forceinline__ bool tile_is_fast(const SetopArgs &p, const TileGeom &g) {
    return g.na_t + g.nb_t == NTH * VT && g.a0 >= 2 && g.a0 + (u64)g.na_t + 2 <= p.na && g.b0 >= 2 &&
           g.b0 + (u64)g.nb_t + 5 <= p.nb;
}

// The same tile, global memory -> LDS without a register round trip (gfx950 LDS-DMA, `global_load_lds_dwordx4`): the
// LDS image is lane-linear (wave-uniform base + 16 bytes per lane), which is exactly the staging layout above; the
// source address is per lane.  Saves the 40 staging VGPRs and the ds_write_b128 pass of the tile.
template <int NTH, int VT>
__device__ __forceinline__ void tile_dma_fast(const SetopArgs &p, const TileGeom &g, int tid, u64 *s_keys) {
    constexpr int NP = TilePairs<NTH, VT>::NP;
    constexpr int SLOTS = NTH * VT + 8;
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(1))) const void glb_void;
    const u64 *pa = p.a + g.a0 - 1 - g.sa0;  // slot s of region A is pa[s]
    const u64 *pb = p.b + g.b0 - 1 - g.sb0;  // slot s of region B is pb[s]
    const int wbase = __builtin_amdgcn_readfirstlane(tid & ~63);
#pragma unroll
    for (int j = 0; j < NP; j++) {
        const int s0 = 2 * (tid + j * NTH);
        const u64 *src = (s0 < g.split ? pa : pb) + s0;
        u64 *dst = s_keys + 2 * (wbase + j * NTH);  // wave-uniform; lane l lands 16 l bytes behind it
        if (2 * (NTH - 1 + j * NTH) + 1 >= SLOTS) {  // last round only (compile time): lanes beyond the tile stay off
            if (s0 < SLOTS) __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)dst, 16, 0, 0);
        } else {
            __builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)dst, 16, 0, 0);
        }
    }
}

template <int NTH, int VT>
__device__ __forceinline__ u32 tile_check_order_lds(const TileGeom &g, int tid, const u64 *s_keys) {
    constexpr int NP = TilePairs<NTH, VT>::NP;
    constexpr int SLOTS = NTH * VT + 8;
    bool unsorted = false, dup = false;
#pragma unroll
    for (int j = 0; j < NP; j++) {
        const int s0 = 2 * (tid + j * NTH);
        bool c0 = (s0 != 0) & (s0 != g.split), c1 = true;  // slot s0 - 1 belongs to the same region
        if (2 * (NTH - 1 + j * NTH) + 1 >= SLOTS) { c1 = s0 < SLOTS; c0 &= c1; }
        const int sr = c1 ? s0 : 0;
        const ulonglong2 q = *reinterpret_cast<const ulonglong2 *>(s_keys + sr);
        const u64 pk = s_keys[sr > 0 ? sr - 1 : 0], k0 = q.x, k1 = q.y;
        unsorted |= (c0 & (pk > k0)) | (c1 & (k0 > k1));
        dup |= (c0 & (pk == k0)) | (c1 & (k0 == k1));
    }
    return (unsorted ? FLAG_UNSORTED : 0u) | (dup ? FLAG_DUP : 0u);
}

// Per-thread merge-path search inside the LDS tile, then a VT-step serial merge that decides
// emit/skip for each merged item.  Outputs stay in registers (ok/ot + bit mask).
// Sets are strictly increasing, so an equal (A[i], B[j]) pair is adjacent in merge order (A
// first): when A is taken and equals the pending B, that B is the next merged item.
//
// INTERIOR = the tile is full and both "next" halos hold real elements.  Then no cursor needs a
// bounds check: by the merge-path property every B item left in the tile is < A[a1] (the nextA
// halo) and every A item left is <= B[b1] (the nextB halo), so comparing against the halo picks
// the right side by itself, and the equality test against the nextB halo is a real match test.
// That removes ~10 of the ~29 instructions of a step; edge tiles take the checked loop.
// CT (both streams carry one taxid per FILE): the step also records which side an item came from (amask) and whether it
// was a matched pair (mmask) -- two bit masks instead of a taxid per item; ct_keep (diff -t, resolved once per call)
// leaves matched codes in the result.
// union / inter with per-record taxids, the LCAs BEHIND the merge loop and DENSE (round 6; needs the clade table in LDS).
// With the look-ups inside the loop every one of the VT serial steps waits for its own round trip to L2 with a quarter of
// the lanes active (53 % of a tile's cycles in the phase profile, profiles/r06_notes.md section 4).  Here the loop only
// notes which steps need an LCA (`need`) and keeps B's taxid of every step; after the compaction each such step queues
// (place of its output record, B's taxid) in the LDS words the matched records have left free -- a union's tile gives up
// one place per match, an intersection's more -- and the workgroup walks the queue with every lane busy: two clade bytes
// per lane and round, the pair step out of LDS (lca_clade_pair_lds), relatives / unknown / merged ids (about 1 % of
// uniformly random pairs) through the root paths in place.  ~3 dense rounds per tile instead of VT sparse ones.
// (Measured and dropped on the way: the same look-ups per THREAD behind the loop, 2 x 4 / 6 / 12 gathers in flight per
// lane with idle lanes reading entry 0 -- 6.35 / 6.65 / 7.5 ms against 5.56 inside the loop at 2 x 3e8.)
#ifndef SETOP_TAX_DEFER
#define SETOP_TAX_DEFER 1
#endif
template <int OP, bool RANK, bool INTERIOR, int VT>
__device__ __forceinline__ void tile_merge_loop_deferred(const SetopArgs &p, const TileGeom &g, int pa, int pb, const u64 *s_keys,
                                                         const u32 *s_tax, const u32 *s_rank, u64 (&ok)[VT], u32 (&ot)[VT],
                                                         u32 &mask, u32 (&tbm)[VT], u32 &need) {
    static_assert(OP == UKM_OP_UNION || OP == UKM_OP_INTER, "operations whose survivors do not depend on an LCA");
    const int base_a = g.base_a, end_a = g.end_a, end_b = g.end_b;
    const int end_bx = end_b + (g.has_next_b ? 1 : 0);
    u64 ak = s_keys[pa], bk = s_keys[pb];
    u32 ar = 0, br = 0;
    if (RANK) { ar = s_rank[pa]; br = s_rank[pb]; }
    bool eq_prev = false;
    if (OP == UKM_OP_UNION) {
        const bool pv = (pa > base_a) || g.has_prev_a;
        eq_prev = pv && (INTERIOR || pb < end_b) && key_eq<RANK>(s_keys[pa - 1], RANK ? s_rank[pa - 1] : 0, bk, br);
    }
    const bool mix = OP == UKM_OP_INTER && (p.flags & UKM_F_MIX_TAXID) != 0;
    need = 0;  // matched steps whose two taxids differ and are both non-zero: ot[s] holds A's, tbm[s] B's
    mask = 0;
#pragma unroll
    for (int s = 0; s < VT; s++) {
        bool take_a, take_b, match;
        if (INTERIOR) {
            take_a = key_le<RANK>(ak, ar, bk, br);
            take_b = !take_a;
            match = key_eq<RANK>(ak, ar, bk, br);
        } else {
            const bool a_ok = pa < end_a, b_ok = pb < end_b;
            take_a = a_ok && (!b_ok || key_le<RANK>(ak, ar, bk, br));
            take_b = !take_a && b_ok;
            match = take_a && (pb < end_bx) && key_eq<RANK>(ak, ar, bk, br);
        }
        bool emit;
        u64 ek = ak;
        if (OP == UKM_OP_UNION) {
            emit = take_a || (take_b && !eq_prev);
            ek = take_a ? ak : bk;
            eq_prev = match;
        } else {
            emit = match;
        }
        const u32 ta = s_tax[pa], tb = s_tax[pb];
        u32 et = (OP == UKM_OP_UNION && !take_a) ? tb : ta;  // (a match takes A: et = A's taxid)
        const bool zero = ta == 0 || tb == 0;
        if (match && zero) et = mix ? (ta | tb) : 0u;  // LCA(x, 0) = 0; inter --mix-taxid: the other one (inter.go:229-236)
        need |= (match && !zero && ta != tb) ? (1u << s) : 0u;
        tbm[s] = tb;
        ok[s] = ek;
        ot[s] = et;
        mask |= emit ? (1u << s) : 0u;
        pa += take_a ? 1 : 0;
        pb += take_b ? 1 : 0;
        const int idx = take_a ? pa : pb;
        const u64 nk = s_keys[idx];
        ak = take_a ? nk : ak;
        bk = take_a ? bk : nk;
        if (RANK) {
            const u32 nr = s_rank[idx];
            ar = take_a ? nr : ar;
            br = take_a ? br : nr;
        }
    }
}
// the queue: one 8-byte word per step in `need`, behind the tile's compacted records (call with tile_compact)
template <int VT>
__device__ __forceinline__ void tile_queue_lca(u32 excl, u32 mask, u32 need, u32 qpos, const u32 (&tbm)[VT], u64 *s_queue) {
#pragma unroll
    for (int s = 0; s < VT; s++) {
        if (need & (1u << s)) {
            const u32 below = (1u << s) - 1u;
            const u32 w = excl + (u32)__popc(mask & below);
            s_queue[qpos + (u32)__popc(need & below)] = ((u64)w << 32) | tbm[s];
        }
    }
}
// every lane busy: s_tax[w] = LCA(s_tax[w], b) for the n queued (w, b).  A pair the clade codes do not settle (about 1 % of
// uniformly random pairs) would cost its wave two or three more dependent table reads -- 7 % of the kernel, since most waves
// hold one: the tile's first FIX_SLOTS such pairs are handed to setop_taxid_fix_kernel instead (s_fix, s_nfix; their places
// keep A's taxid until then), only what does not fit is settled here.
struct FixLds {
    u32 n;
    u32 w[FIX_SLOTS], a[FIX_SLOTS], b[FIX_SLOTS];
};
template <bool ON> struct FixLdsOpt { FixLds t; };
template <> struct FixLdsOpt<false> { u32 t; };
template <int NTH>
__device__ __forceinline__ void tile_lca_dense(const TaxDev &T, const CladeLds &L, int tid, u32 n, const u64 *s_queue, u32 *s_tax,
                                               FixLds *fx) {
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    auto settle = [&](u32 w, u32 a, u32 b, u32 ca, u32 cb) {
        if (ca != cb && ca != 0 && cb != 0) {
            s_tax[w] = lca_clade_pair_lds(T, L, ca, cb);
            return;
        }
        if (fx) {
            const u32 k = atomicAdd(&fx->n, 1u);
            if (k < FIX_SLOTS) {
                fx->w[k] = w; fx->a[k] = a; fx->b[k] = b;
                return;
            }
        }
        s_tax[w] = lca_from_rows(T, a, b, a < T.size ? T.anc[a] : zero4, b < T.size ? T.anc[b] : zero4);
    };
    for (u32 i = (u32)tid; i < n; i += 2 * NTH) {
        const bool two = i + NTH < n;
        const u64 e0 = s_queue[i], e1 = s_queue[two ? i + NTH : i];
        const u32 w0 = (u32)(e0 >> 32), w1 = (u32)(e1 >> 32), b0 = (u32)e0, b1 = (u32)e1;
        const u32 a0 = s_tax[w0], a1 = s_tax[w1];
        const bool in0 = a0 < T.size && b0 < T.size, in1 = two && a1 < T.size && b1 < T.size;
        const u32 ca0 = T.clade8[in0 ? a0 : 0u], cb0 = T.clade8[in0 ? b0 : 0u];
        const u32 ca1 = T.clade8[in1 ? a1 : 0u], cb1 = T.clade8[in1 ? b1 : 0u];
        settle(w0, a0, b0, ca0, cb0);
        if (two) settle(w1, a1, b1, ca1, cb1);
    }
}

template <int OP, bool TAX, bool RANK, bool INTERIOR, bool CT, int VT, bool DEFER>
__device__ __forceinline__ void tile_merge_loop(const SetopArgs &p, const TileGeom &g, int pa, int pb,
                                                const u64 *s_keys, const u32 *s_tax, const u32 *s_rank,
                                                u64 (&ok)[VT], u32 (&ot)[VT], u32 &mask, u32 &amask, u32 &mmask, bool ct_keep,
                                                const CladeLds *cl, u32 (&tbm)[VT], u32 &need) {
    if constexpr (DEFER) {  // (chosen by the host: the taxonomy comes with one-byte clade codes, TaxDev::cpath)
        static_assert(TAX && !CT, "per-record taxids");
        amask = 0;
        mmask = 0;
        tile_merge_loop_deferred<OP, RANK, INTERIOR, VT>(p, g, pa, pb, s_keys, s_tax, s_rank, ok, ot, mask, tbm, need);
        return;
    }
    const int base_a = g.base_a, end_a = g.end_a, end_b = g.end_b;
    const int end_bx = end_b + (g.has_next_b ? 1 : 0);
    u64 ak = s_keys[pa], bk = s_keys[pb];
    u32 ar = 0, br = 0;
    if (RANK) { ar = s_rank[pa]; br = s_rank[pb]; }
    bool eq_prev = false;  // union: the pending B equals the A that precedes it in merge order
    if (OP == UKM_OP_UNION) {
        const bool pv = (pa > base_a) || g.has_prev_a;
        eq_prev = pv && (INTERIOR || pb < end_b) && key_eq<RANK>(s_keys[pa - 1], RANK ? s_rank[pa - 1] : 0, bk, br);
    }
    const bool mix = (p.flags & UKM_F_MIX_TAXID) != 0;
    const bool cmp = (p.flags & UKM_F_CMP_TAXID) != 0;
    // Files carry few distinct taxids over long stretches (one per genome, or one per clade after LCA
    // assignment), so consecutive matches of a thread mostly ask for the same pair: remember the last one.
    u32 memo_a = 0, memo_b = 0, memo_l = 0;  // LCA(0, 0) = 0
    const bool lds_pairs = TAX && p.tax.cpath != nullptr;  // (uniform) the clade-pair step out of LDS
    auto lca_memo = [&](u32 a, u32 b) -> u32 {
        if (a != memo_a || b != memo_b) {
            memo_l = lds_pairs ? lca_dev_lds(p.tax, *cl, a, b) : lca_dev(p.tax, a, b);
            memo_a = a;
            memo_b = b;
        }
        return memo_l;
    };
    mask = 0;
    amask = 0;
    mmask = 0;
#pragma unroll
    for (int s = 0; s < VT; s++) {
        bool take_a, take_b, match;
        if (INTERIOR) {
            take_a = key_le<RANK>(ak, ar, bk, br);
            take_b = !take_a;
            match = key_eq<RANK>(ak, ar, bk, br);  // implies take_a
        } else {
            const bool a_ok = pa < end_a, b_ok = pb < end_b;
            take_a = a_ok && (!b_ok || key_le<RANK>(ak, ar, bk, br));
            take_b = !take_a && b_ok;
            match = take_a && (pb < end_bx) && key_eq<RANK>(ak, ar, bk, br);
        }
        bool emit;
        u64 ek = ak;
        u32 et = 0;
        if (OP == UKM_OP_UNION) {
            emit = take_a || (take_b && !eq_prev);
            ek = take_a ? ak : bk;
            eq_prev = match;
        } else if (OP == UKM_OP_INTER) {
            emit = match;
        } else if (OP == UKM_OP_MERGE_INTERNAL) {
            emit = take_a || take_b;
            ek = take_a ? ak : bk;
        } else {
            emit = take_a && !match;
            if (CT) emit = take_a && (!match || ct_keep);
        }
        if (CT) {
            amask |= take_a ? (1u << s) : 0u;
            mmask |= match ? (1u << s) : 0u;
        }
        if (TAX) {
            const u32 ta = s_tax[pa], tb = s_tax[pb];
            if (OP == UKM_OP_UNION) {
                et = take_a ? ta : tb;
                if (match) et = lca_memo(ta, tb);
            } else if (OP == UKM_OP_INTER) {
                if (match) {
                    if (mix) et = (ta == 0) ? tb : ((tb == 0) ? ta : lca_memo(ta, tb));
                    else et = lca_memo(ta, tb);
                }
            } else if (OP == UKM_OP_MERGE_INTERNAL) {
                et = take_a ? ta : tb;
            } else {
                et = ta;
                if (match && cmp && (ta == tb || lca_memo(tb, ta) == ta)) emit = true;
            }
        }
        ok[s] = ek;
        ot[s] = et;
        mask |= emit ? (1u << s) : 0u;
        pa += take_a ? 1 : 0;
        pb += take_b ? 1 : 0;
        // one LDS read refills whichever cursor moved (when neither moved it re-reads bk)
        const int idx = take_a ? pa : pb;
        const u64 nk = s_keys[idx];
        ak = take_a ? nk : ak;
        bk = take_a ? bk : nk;
        if (RANK) {
            const u32 nr = s_rank[idx];
            ar = take_a ? nr : ar;
            br = take_a ? br : nr;
        }
    }
}

template <int OP, bool TAX, bool RANK, bool CT, int NTH, int VT, bool DEFER>
__device__ __forceinline__ void tile_merge(const SetopArgs &p, const TileGeom &g, int tid, const u64 *s_keys,
                                           const u32 *s_tax, const u32 *s_rank, u64 (&ok)[VT], u32 (&ot)[VT],
                                           u32 &mask, u32 &amask, u32 &mmask, bool ct_keep, int &ia0, int &ib0, const CladeLds *cl,
                                           u32 (&tbm)[VT], u32 &need) {
    const int na_t = g.na_t, nb_t = g.nb_t, total = na_t + nb_t;
    const int base_a = g.base_a, base_b = g.base_b;
    int diag = tid * VT;
    if (diag > total) diag = total;
    int lo = diag > nb_t ? diag - nb_t : 0;
    int hi = diag < na_t ? diag : na_t;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const int ia = base_a + mid, ib = base_b + diag - 1 - mid;
        const bool le = key_le<RANK>(s_keys[ia], RANK ? s_rank[ia] : 0, s_keys[ib], RANK ? s_rank[ib] : 0);
        lo = le ? mid + 1 : lo;
        hi = le ? hi : mid;
    }
    const int pa = base_a + lo, pb = base_b + diag - lo;
    ia0 = lo;         // the thread's first A / B record, counted from the tile's first (source words, below)
    ib0 = diag - lo;
    // wave-uniform choice (tile geometry): no divergence
    if (total == NTH * VT && g.has_next_a && g.has_next_b)
        tile_merge_loop<OP, TAX, RANK, true, CT, VT, DEFER>(p, g, pa, pb, s_keys, s_tax, s_rank, ok, ot, mask, amask, mmask, ct_keep, cl, tbm, need);
    else
        tile_merge_loop<OP, TAX, RANK, false, CT, VT, DEFER>(p, g, pa, pb, s_keys, s_tax, s_rank, ok, ot, mask, amask, mmask, ct_keep, cl, tbm, need);
}

// compact the emitted items of this thread into LDS at its exclusive offset
template <bool TAX, int VT>
__device__ __forceinline__ void tile_compact(u32 excl, u32 mask, const u64 (&ok)[VT], const u32 (&ot)[VT],
                                             u64 *s_keys, u32 *s_tax) {
#pragma unroll
    for (int s = 0; s < VT; s++) {
        if (mask & (1u << s)) {
            const u32 w = excl + (u32)__popc(mask & ((1u << s) - 1u));
            s_keys[w] = ok[s];
            if (TAX) s_tax[w] = ot[s];
        }
    }
}

// LDS -> HBM: contiguous coalesced run of `count` records at out[base ...), 16 bytes per store
// where the global address allows it (the first/last record may go out alone).
template <bool TAX, int NTH>
__device__ __forceinline__ void tile_flush(const SetopArgs &p, int tid, u64 base, u32 count, const u64 *s_keys,
                                           const u32 *s_tax) {
    if (base + count <= p.out_cap) {
        u64 *o = p.out + base;
        const int sh = (int)(((uintptr_t)o >> 3) & 1);  // 1: o[0] sits on an odd 8-byte slot
        const int npairs = ((int)count + sh + 1) >> 1;
        for (int m = tid; m < npairs; m += NTH) {
            const int i0 = 2 * m - sh, i1 = i0 + 1;
            const bool v0 = i0 >= 0, v1 = i1 < (int)count;
            const u64 k0 = s_keys[v0 ? i0 : 0], k1 = s_keys[v1 ? i1 : 0];
            if (TAX && v0 && v1) {  // (non-temporal beside the taxid look-ups: see tile_load)
                typedef u64 v2u64 __attribute__((ext_vector_type(2)));
                v2u64 w;
                w.x = k0;
                w.y = k1;
                __builtin_nontemporal_store(w, reinterpret_cast<v2u64 *>(o + i0));
            } else if (v0 && v1) *reinterpret_cast<ulonglong2 *>(o + i0) = make_ulonglong2(k0, k1);
            else if (v0) o[i0] = k0;
            else if (v1) o[i1] = k1;
        }
        if (TAX) {
            u32 *to = p.tout + base;
            for (u32 i = (u32)tid; i < count; i += NTH) __builtin_nontemporal_store(s_tax[i], to + i);
        }
    } else {  // capacity overflow: guarded stores; the host reports UKM_ERR_CAPACITY
        for (u32 i = (u32)tid; i < count; i += NTH) {
            const u64 pos = base + i;
            if (pos < p.out_cap) {
                p.out[pos] = s_keys[i];
                if (TAX) p.tout[pos] = s_tax[i];
            }
        }
    }
}

// Both streams carry one taxid per FILE (CT): the taxids of the tile's output.  inter: every record gets LCA(A's, B's)
// (mix-taxid rule included), diff: A's own -- a fill.  union / keep-everything merge: A's, B's or the LCA by where the
// record came from; the values go through the LDS words the compacted keys have just left (no LDS beyond the plain
// kernel's), so that the stores are as coalesced as the keys'.
template <int OP, int NTH, int VT>
__device__ __forceinline__ void tile_flush_ct(const SetopArgs &p, int tid, u64 base, u32 count, u32 excl, u32 mask, u32 amask,
                                              u32 mmask, u32 ct_lca, u32 *s_t32) {
    if (OP == UKM_OP_INTER || OP == UKM_OP_DIFF) {
        const u32 v = OP == UKM_OP_INTER ? ct_lca : p.cta;
        for (u32 i = (u32)tid; i < count; i += NTH)
            if (base + i < p.out_cap) p.tout[base + i] = v;
        return;
    }
    __syncthreads();  // every thread has stored its share of the compacted keys
#pragma unroll
    for (int s = 0; s < VT; s++) {
        if (mask & (1u << s)) {
            const u32 w = excl + (u32)__popc(mask & ((1u << s) - 1u));
            const bool m = OP == UKM_OP_UNION && ((mmask >> s) & 1u);
            s_t32[w] = m ? ct_lca : (((amask >> s) & 1u) ? p.cta : p.ctb);
        }
    }
    __syncthreads();
    for (u32 i = (u32)tid; i < count; i += NTH)
        if (base + i < p.out_cap) p.tout[base + i] = s_t32[i];
}

// ---- per-record taxids on plain sets, in two launches (round 5) ---------------------------------------------------------
// The taxid instantiation carries the taxids through the merge: 13 items per thread instead of 19, 80 KB of LDS, no LDS-DMA,
// 128 registers with spills, and the LCA of a matched pair -- two dependent table reads -- inside a thread's SERIAL merge
// loop: 132 vector and 85 scalar lane-instructions per record against 67 / 22 for plain codes (profiles/r05_setop_tax_pmc.txt).
// Which input an output record came from does not depend on any taxid (union, inter, diff without -t, the keep-everything
// merge), so: launch 1 = the PLAIN-key kernel, whose epilogue writes one SOURCE WORD per output record where its taxid will
// stand ([13:0] the record's place in the tile's A range, [27:14] in its B range, [28] a matched pair, [29] taken from
// B; put together from the two bit masks of the thread's steps: ~10 instructions per step); launch 2 = one workgroup per
// tile turns the words into taxids IN PLACE: a thread per output record, every taxid read and every LCA independent of
// every other -- the table reads of a whole tile are in flight together instead of one pair at a time per thread.
// 8 more bytes of HBM traffic per output record (the word written and read back).
constexpr u32 SRC_MATCH = 1u << 28, SRC_FROM_B = 1u << 29;
template <int OP, int NTH, int VT>
__device__ __forceinline__ void tile_flush_src(const SetopArgs &p, int tid, u64 base, u32 count, u32 excl, u32 mask, u32 amask,
                                               u32 mmask, int ia0, int ib0, u32 *s_t32) {
    static_assert(NTH * VT + 8 < (1 << 14), "a place in the tile takes 14 bits");
    __syncthreads();  // every thread has stored its share of the compacted codes
#pragma unroll
    for (int s = 0; s < VT; s++) {
        if (mask & (1u << s)) {
            const u32 below = (1u << s) - 1u;
            const int na_before = __popc(amask & below);
            const u32 ia = (u32)(ia0 + na_before), ib = (u32)(ib0 + s - na_before);
            const bool at = (amask >> s) & 1u, m = (mmask >> s) & 1u;
            s_t32[excl + (u32)__popc(mask & below)] = ia | (ib << 14) | (m ? SRC_MATCH : 0u) | (at ? 0u : SRC_FROM_B);
        }
    }
    __syncthreads();
    for (u32 i = (u32)tid; i < count; i += NTH)
        if (base + i < p.out_cap) p.tout[base + i] = s_t32[i];
}

constexpr int GATHER_NT = 256;
#ifndef GATHER_U
#define GATHER_U 8  /* records per thread and step (experiments: 2, 4, 8) */
#endif
template <int OP, int TILE>
__global__ __launch_bounds__(GATHER_NT) void setop_taxid_gather_kernel(SetopArgs p) {
    setop_resolve_sizes(p, (u64)TILE);
    const u64 tile = blockIdx.x;
    if (tile >= p.ntiles) return;
    if (sload_u64(&p.result[1]) & FLAG_TIMEOUT) return;  // (the host runs the pass again: the status words are incomplete)
    u64 base, end;
    if (OP == UKM_OP_MERGE_INTERNAL) {
        base = tile * (u64)TILE;
        end = base + TILE < p.na + p.nb ? base + TILE : p.na + p.nb;
    } else {
        base = tile ? (lb_load(&p.status[(tile - 1) * LB_STRIDE]) & LB_VAL) : 0ull;
        end = lb_load(&p.status[tile * LB_STRIDE]) & LB_VAL;
    }
    if (end > p.out_cap) end = p.out_cap;
    if (end <= base) return;
    const u64 a0 = p.mp[tile], b0 = tile * (u64)TILE - a0;
    const bool mix = (p.flags & UKM_F_MIX_TAXID) != 0;
    const u32 *safe32 = reinterpret_cast<const u32 *>(p.result);  // (always mapped: where a lane has nothing to read)
    constexpr bool LCA_OP = OP == UKM_OP_UNION || OP == UKM_OP_INTER;
    __shared__ CladeLdsOpt<LCA_OP> s_clade_tab;  // the clade-pair step out of LDS (ukm_device.h)
    const bool lds_pairs = LCA_OP && p.tax.cpath != nullptr;  // (uniform; every early return above is uniform too)
    if constexpr (LCA_OP) {
        clade_lds_load(p.tax, s_clade_tab.t, (int)threadIdx.x, GATHER_NT);
        __syncthreads();
    }
    // U records per thread and step: four rounds of loads -- the words, the taxids, the clade codes, the clade pairs -- each
    // round with all of its reads in flight, none inside a branch.  (Measured at 2 x 1e8, inter with one taxid per file as arrays:
    // 256 threads x 8 records 0.99 ms, x 4 1.06, x 2 1.13; a whole tile per 1024-thread workgroup in ONE step 1.22.)
    constexpr int U = GATHER_U;
    constexpr bool LCA = OP == UKM_OP_UNION || OP == UKM_OP_INTER;
    for (u64 i0 = base + threadIdx.x; i0 < end; i0 += (u64)GATHER_NT * U) {
        u32 w[U], va[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const u64 i = i0 + (u64)u * GATHER_NT;
            w[u] = p.tout[i < end ? i : base];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const u32 ia = w[u] & 0x3FFFu, ib = (w[u] >> 14) & 0x3FFFu;
            // Every lane reads BOTH inputs' taxid at its place, needed or not: the places of neighbouring records are neighbours, so
            // the unneeded reads fall into lines the wave fetches anyway.  A place beyond its input (the B record behind the last
            // one; an unsorted input, whose result the host discards) is clamped into it.
            const u64 ga = a0 + ia, gb = b0 + ib;
            va[u] = p.cta;
            vb[u] = p.ctb;
            if (p.ta) va[u] = *(p.na ? p.ta + (ga < p.na ? ga : p.na - 1) : safe32);  // (uniform branches)
            if (p.tb) vb[u] = *(p.nb ? p.tb + (gb < p.nb ? gb : p.nb - 1) : safe32);
        }
        u32 quick[U];
        u32 qmask = 0;
        if (LCA && p.tax.pair != nullptr && p.tax.clade8 != nullptr) {  // (uniform)
            u32 ca[U], cb[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const bool look = (w[u] & SRC_MATCH) != 0 && va[u] != vb[u] && va[u] != 0 && vb[u] != 0 && va[u] < p.tax.size && vb[u] < p.tax.size;
                ca[u] = p.tax.clade8[look ? va[u] : 0u];
                cb[u] = p.tax.clade8[look ? vb[u] : 0u];
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const bool q = ca[u] != cb[u] && ca[u] != 0 && cb[u] != 0;
                qmask |= q ? (1u << u) : 0u;
                if constexpr (LCA_OP) {
                    if (lds_pairs) quick[u] = q ? lca_clade_pair_lds(p.tax, s_clade_tab.t, ca[u], cb[u]) : 0u;
                    else quick[u] = p.tax.pair[q ? ca[u] * p.tax.kp + cb[u] : 0u];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const u64 i = i0 + (u64)u * GATHER_NT;
            if (i >= end) continue;
            const bool from_b = (w[u] & SRC_FROM_B) != 0, m = (w[u] & SRC_MATCH) != 0;
            u32 tv;
            if (LCA && m) {
                if (OP == UKM_OP_INTER && mix && (va[u] == 0 || vb[u] == 0)) tv = va[u] == 0 ? vb[u] : va[u];
                else if ((qmask >> u) & 1u) tv = quick[u];
                else tv = lca_dev(p.tax, va[u], vb[u]);  // relatives, zero / unknown / merged ids, no clade tables
            } else {
                tv = from_b ? vb[u] : va[u];
            }
            p.tout[i] = tv;
        }
    }
}

// the pairs the DEFER tiles left behind (SetopArgs::fix): one thread per place of the list
__global__ __launch_bounds__(256) void setop_taxid_fix_kernel(SetopArgs p) {
    const u64 g = (u64)blockIdx.x * 256 + threadIdx.x;
    const u64 tile = g / FIX_SLOTS;
    if (tile >= p.ntiles) return;
    if (sload_u64(&p.result[1]) & FLAG_TIMEOUT) return;  // (the host runs the pass again: not every tile has written its count)
    if ((u32)(g % FIX_SLOTS) >= p.fix_cnt[tile]) return;
    const uint4 e = p.fix[g];
    const u64 pos = ((u64)e.y << 32) | e.x;
    // (a != b, both non-zero, not settled by their clade codes: straight to the root paths, as the tile would have gone)
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    const TaxDev &T = p.tax;
    if (pos < p.out_cap) p.tout[pos] = lca_from_rows(T, e.z, e.w, e.z < T.size ? T.anc[e.z] : zero4, e.w < T.size ? T.anc[e.w] : zero4);
}

// result[4] of a CT call: [31:0] the taxid of a matched pair (inter --mix-taxid: a zero on either side yields the other,
// inter.go:229-236), [32] diff -t keeps matched codes (diff.go:404-409: the later file's taxid equals the first file's or
// lies below it).  One thread; runs between the partition launch (which clears the control words) and the tile kernel.
__global__ void setop_ct_kernel(SetopArgs p, int op) {
    const u32 a = p.cta, b = p.ctb;
    u32 l;
    if (op == UKM_OP_INTER && (p.flags & UKM_F_MIX_TAXID)) l = a == 0 ? b : (b == 0 ? a : lca_dev(p.tax, a, b));
    else l = lca_dev(p.tax, a, b);
    const bool keep = op == UKM_OP_DIFF && (p.flags & UKM_F_CMP_TAXID) && (a == b || lca_dev(p.tax, b, a) == a);
    p.result[4] = (u64)l | ((u64)(keep ? 1u : 0u) << 32);
}

// ---- the tile kernel: one workgroup per tile of NTH*VT merged items ------------------------------------
// TICKET = false (default): tile id = blockIdx.x, so the partition words are fetched with scalar
//   loads at kernel entry and no atomic sits in front of the tile loads (measured -0.7 ms of
//   7.4 ms).  Look-back then relies on the hardware dispatching workgroups in increasing
//   order (observed on MI355X; each XCD's dispatcher walks its share of the grid in order), so
//   every predecessor of a running tile is running or done.  Results never depend on that: if
//   a predecessor fails to publish within LB_SPIN_LIMIT polls the kernel raises FLAG_TIMEOUT
//   and the host re-runs with TICKET = true, where tile ids come from an atomic counter and
//   forward progress holds for any dispatch order.
// Up to 80 KB of LDS per workgroup (plain keys; taxids OR ranks riding along): two 512-thread workgroups per CU = 4
// waves per SIMD, so the register budget is 128 and the allocator is told so (by itself it budgets 256 for a
// 512-thread workgroup: the plain kernel came out at 129 with its two load paths, the union with taxids at 131 once
// the LCA grew -- one workgroup per CU, union with taxids 1.45 -> 1.92 ms).  Taxids AND ranks (106 KB of LDS, one
// workgroup per CU anyway) keep the default budget.
#ifndef SETOP_WAVES
#define SETOP_WAVES 4  /* experiments only: 6 = three workgroups per CU (needs SETOP_VT <= 12) */
#endif
#ifndef SETOP_WAVES_TAX
#define SETOP_WAVES_TAX 6  /* per-record taxids, no ranks: THREE workgroups per CU (SETOP_VT_TAX <= 7) */
#endif
template <int OP, bool TAX, bool RANK, bool TICKET, int NTH, int VT, bool CT = false, bool DEFER = false>
__global__ __launch_bounds__(NTH) __attribute__((amdgpu_waves_per_eu((TAX && RANK) ? 2 : ((TAX && VT <= 8) ? SETOP_WAVES_TAX : SETOP_WAVES), (TAX && RANK) ? 8 : ((TAX && VT <= 8) ? SETOP_WAVES_TAX : SETOP_WAVES))))
void setop_tile_kernel(SetopArgs p) {
    static_assert(!(CT && TAX), "CT: no per-record taxids");
    constexpr int TILE = NTH * VT;
    constexpr int SLOTS = TILE + 8;
    constexpr int NP = TilePairs<NTH, VT>::NP;
    __shared__ __attribute__((aligned(16))) u64 s_keys[SLOTS];
    __shared__ __attribute__((aligned(16))) u32 s_tax[TAX ? SLOTS : 2];
    __shared__ __attribute__((aligned(16))) u32 s_rank[RANK ? SLOTS : 2];
    __shared__ u32 s_scan[NTH / 64 + 1];
    __shared__ u64 s_misc[2];
    __shared__ CladeLdsOpt<TAX> s_clade_tab;
    __shared__ FixLdsOpt<DEFER> s_fix;
    const int tid = (int)threadIdx.x;
    if constexpr (DEFER) {
        if (tid == 0) s_fix.t.n = 0;  // (read behind the barriers of the tile's staging)
    }
#ifdef UKM_PROFILE_PHASES
    u64 ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    u64 tlast = clock64();
#endif
    u64 tile = blockIdx.x;
    if (TICKET) {
        if (tid == 0) s_misc[0] = (u64)atomicAdd(p.ticket, 1u);
        __syncthreads();
        tile = s_misc[0];
    }
    PH(0);
    setop_resolve_sizes(p, (u64)TILE);
    if (tile >= p.ntiles) return;  // chained call: the launch covers the upper bound of |A|
    const TileGeom g = tile_geom<NTH, VT>(p, tile);
    u32 bad = 0;
    bool fast = false;  // workgroup-uniform
#ifndef SETOP_NO_FAST
    if (!TAX && !RANK) fast = tile_is_fast<NTH, VT>(p, g);
#endif
    if (fast) {
        // interior tile of plain keys: LDS-DMA staging; its order check runs after the aggregate is published (below)
        tile_dma_fast<NTH, VT>(p, g, tid, s_keys);
        __syncthreads();  // hipcc puts the vmcnt(0) for the LDS-DMA in front of the barrier
        PH(1);
    } else {
        u64 rk[2 * NP];
        u32 rt[2 * NP], rr[2 * NP];
        tile_load<TAX, RANK, NTH, VT>(p, g, tid, rk, rt, rr);
        if constexpr (TAX) clade_lds_load(p.tax, s_clade_tab.t, tid, NTH);
        tile_to_lds<TAX, RANK, NTH, VT>(tid, rk, rt, rr, s_keys, s_tax, s_rank);
        __syncthreads();
        PH(1);
        bad = tile_check_order<RANK, NTH, VT>(g, tid, rk, rr, s_keys, s_rank);
    }
    u64 ok[VT];
    u32 ot[VT];
    u32 tbm[VT];  // DEFER only: B's taxid of every step, `need` = the steps that wait for an LCA
    u32 need = 0, nneed = 0, qpos = 0;
    u32 mask, amask = 0, mmask = 0;
    int ia0 = 0, ib0 = 0;
    u32 ct_lca = 0;
    bool ct_keep = false;
    if (CT) {  // (written by setop_ct_kernel in front of this launch: a scalar load)
        const u64 w = sload_u64(&p.result[4]);
        ct_lca = (u32)w;
        ct_keep = ((w >> 32) & 1ull) != 0;
    }
#ifdef SETOP_ABL_NOMERGE  // experiment only: timing without the search + serial merge
    mask = 0xAAAAu | (u32)(tid & 1);
#pragma unroll
    for (int s = 0; s < VT; s++) { ok[s] = s_keys[tid * VT + s]; ot[s] = 0; }
#else
    tile_merge<OP, TAX, RANK, CT, NTH, VT, DEFER>(p, g, tid, s_keys, s_tax, s_rank, ok, ot, mask, amask, mmask, ct_keep, ia0, ib0,
                                           TAX ? reinterpret_cast<const CladeLds *>(&s_clade_tab.t) : nullptr, tbm, need);
#endif
    PH(2);
    u32 tile_total;
    // The block scan, opened up: the tile's aggregate is known (and published) behind its FIRST barrier; the order
    // check of an interior tile's inputs -- which nothing downstream needs -- runs between the two barriers, i.e.
    // AFTER the publication instead of in front of the merge.  Successors see the aggregate ~1 k cycles earlier
    // relative to this tile's own look-back, which is what their look-back waits for (round 3: union 5.0 -> 4.8 ms,
    // inter 4.15 -> 4.0 ms at 2 x 1e9).  Unsorted input only makes the merge emit garbage that the flag voids.
    u32 excl;
    {
        constexpr int NW = NTH / 64;
        // (DEFER: the queue positions ride in the upper half of the same scan; both sums stay below 2^16)
        const u32 v = (u32)__popc(mask) | (DEFER ? (u32)__popc(need) << 16 : 0u);
        const int lane = lane_id(), wave = tid >> 6;
        const u32 incl = wave_incl_scan_u32(v);
        if (lane == 63) s_scan[wave] = incl;
        __syncthreads();
        u32 wbase = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const u32 t = s_scan[w];
            if (w < wave) wbase += t;
            tot += t;
        }
        tile_total = tot;
        excl = wbase + incl - v;
        if (DEFER) {
            static_assert(!DEFER || NTH * VT + 8 < (1 << 16), "two 16-bit sums in one word");
            nneed = tot >> 16;
            qpos = excl >> 16;
            tile_total = tot & 0xFFFFu;
            excl &= 0xFFFFu;
        }
        if (OP != UKM_OP_MERGE_INTERNAL && tid == 0) lb_publish(p.status, tile, (u64)tile_total);
        if (fast) bad |= tile_check_order_lds<NTH, VT>(g, tid, s_keys);
        __syncthreads();  // every thread has finished reading the tile from LDS; s_scan may be reused
    }
#ifndef SETOP_ABL_NOCOMPACT
    tile_compact<TAX, VT>(excl, mask, ok, ot, s_keys, s_tax);
#endif
    if constexpr (DEFER) {
        // the LCAs, dense: the queue lies behind the compacted records (a match leaves one place free in a union's tile
        // -- its B record is dropped, here or as the next tile's first step -- and more in an intersection's)
        tile_queue_lca<VT>(excl, mask, need, qpos, tbm, s_keys + tile_total);
        __syncthreads();
        tile_lca_dense<NTH>(p.tax, s_clade_tab.t, tid, nneed, s_keys + tile_total, s_tax, p.fix ? &s_fix.t : nullptr);
    }
    PH(3);
    if (OP == UKM_OP_MERGE_INTERNAL) {
        if (tid == 0) s_misc[1] = tile * (u64)TILE;  // every record is kept: the tile's output offset is known
    } else if (tid < 64) {
        bool timed_out = false;
#ifdef LB_PRE_SLEEP
        __builtin_amdgcn_s_sleep(LB_PRE_SLEEP);
#endif
#ifdef SETOP_ABL_NOLB  // experiment only: wrong output positions, no look-back
        const u64 base = tile * (u64)TILE;
#else
        const u64 base = lb_resolve(p.status, tile, (u64)tile_total, lane_id(), TICKET ? nullptr : &timed_out);
#endif
        if (tid == 0) s_misc[1] = base;
        if (timed_out) bad |= FLAG_TIMEOUT;
    }
    PH(4);
    if (OP == UKM_OP_MERGE_INTERNAL) bad &= ~FLAG_DUP;  // duplicates are legal in a plain merge
    {
        // one atomic per wave at most (a stream full of duplicates would otherwise send one per thread
        // to the same word)
        u32 wbad = 0;
#pragma unroll
        for (u32 f = 1; f <= FLAG_TIMEOUT; f <<= 1)
            if (__ballot((bad & f) != 0)) wbad |= f;
        if (wbad && lane_id() == 0) atomicOr((unsigned long long *)&p.result[1], (unsigned long long)wbad);
    }
    __syncthreads();
    const u64 base = s_misc[1];
    if constexpr (DEFER) {
        if (p.fix) {  // (uniform) the pairs left to setop_taxid_fix_kernel, at their output positions
            const u32 nf = s_fix.t.n < FIX_SLOTS ? s_fix.t.n : FIX_SLOTS;
            if ((u32)tid < nf) {
                const u64 pos = base + s_fix.t.w[tid];
                p.fix[tile * FIX_SLOTS + (u32)tid] = make_uint4((u32)pos, (u32)(pos >> 32), s_fix.t.a[tid], s_fix.t.b[tid]);
            }
            if (tid == 0) p.fix_cnt[tile] = nf;
        }
    }
#ifndef SETOP_ABL_NOFLUSH
    tile_flush<TAX, NTH>(p, tid, base, tile_total, s_keys, s_tax);
    if (CT) {
        if (p.ta != nullptr || p.tb != nullptr)  // (uniform) per-record taxids on a stream: source words now, setop_taxid_gather_kernel next
            tile_flush_src<OP, NTH, VT>(p, tid, base, tile_total, excl, mask, amask, mmask, ia0, ib0, reinterpret_cast<u32 *>(s_keys));
        else
            tile_flush_ct<OP, NTH, VT>(p, tid, base, tile_total, excl, mask, amask, mmask, ct_lca, reinterpret_cast<u32 *>(s_keys));
    }
#endif
    if (tid == 0 && tile == p.ntiles - 1) p.result[0] = base + tile_total;
    PH(5);
#ifdef UKM_PROFILE_PHASES
    if (tid == 0 && p.dbg) {
        for (int i = 0; i < 7; i++) p.dbg[blockIdx.x * 8 + i] = ph[i];
        p.dbg[blockIdx.x * 8 + 7] = 1;
    }
#endif
}

// rank of each element inside its run of equal codes (multiset path)
__global__ void rank_in_run_kernel(const u64 *k, u64 n, u32 *rank) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 key = k[i];
    u32 r = 0;
    while (r < 32 && i > r && k[i - r - 1] == key) r++;
    if (r == 32 && i > r && k[i - r - 1] == key) {  // long run: lower_bound
        u64 lo = 0, hi = i - 32;
        while (lo < hi) {
            u64 mid = (lo + hi) >> 1;
            if (k[mid] < key) lo = mid + 1; else hi = mid;
        }
        r = (u32)(i - lo);
    }
    rank[i] = r;
}

__global__ void lower_bound_kernel(const u64 *k, u64 n, const u64 *q, int nq, u64 *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    u64 key = q[i], lo = 0, hi = n;
    while (lo < hi) {
        u64 mid = (lo + hi) >> 1;
        if (k[mid] < key) lo = mid + 1; else hi = mid;
    }
    out[i] = lo;
}

template <int OP, bool TAX, bool RANK, int NTH, int VT, bool CT = false, bool DEFER = false>
void launch_tile(const SetopArgs &p, hipStream_t st, bool ticket) {
    if (CT) hipLaunchKernelGGL(setop_ct_kernel, dim3(1), dim3(1), 0, st, p, OP);
    if (ticket)
        hipLaunchKernelGGL((setop_tile_kernel<OP, TAX, RANK, true, NTH, VT, CT, DEFER>), dim3((unsigned)p.ntiles), dim3(NTH), 0, st, p);
    else
        hipLaunchKernelGGL((setop_tile_kernel<OP, TAX, RANK, false, NTH, VT, CT, DEFER>), dim3((unsigned)p.ntiles), dim3(NTH), 0, st, p);
    if constexpr (DEFER) {
        if (p.fix) hipLaunchKernelGGL(setop_taxid_fix_kernel, dim3((unsigned)((p.ntiles * FIX_SLOTS + 255) / 256)), dim3(256), 0, st, p);
    }
    if constexpr (CT && !RANK) {
        if (p.ta != nullptr || p.tb != nullptr)  // the source words of the launch above -> taxids
            hipLaunchKernelGGL((setop_taxid_gather_kernel<OP, NTH * VT>), dim3((unsigned)p.ntiles), dim3(GATHER_NT), 0, st, p);
    }
}

template <bool TAX, bool RANK, int NTH, int VT, bool CT = false, bool DEFER = false>
void launch_op(int op, const SetopArgs &p, hipStream_t st, bool ticket) {
    if (op == UKM_OP_UNION) launch_tile<UKM_OP_UNION, TAX, RANK, NTH, VT, CT, DEFER>(p, st, ticket);
    else if (op == UKM_OP_INTER) launch_tile<UKM_OP_INTER, TAX, RANK, NTH, VT, CT, DEFER>(p, st, ticket);
    else if (op == UKM_OP_MERGE_INTERNAL) {
        if constexpr (!RANK) launch_tile<UKM_OP_MERGE_INTERNAL, TAX, false, NTH, VT, CT>(p, st, ticket);
    } else launch_tile<UKM_OP_DIFF, TAX, RANK, NTH, VT, CT>(p, st, ticket);
}

constexpr int NTS = SETOP_NT;       // threads per workgroup (512: two workgroups per CU)
constexpr int VT_PLAIN = SETOP_VT;  // 19 items per thread: 76 KiB of keys in LDS per workgroup (2 x 78 KB fit the CU's 160 KB)
// With per-record taxids the tile is SMALL: 7 items per thread = 43 KB of tile + the 5 KB clade table, so that THREE
// workgroups share a CU (6 waves per SIMD at 75 registers).  A tile spends most of its time in round trips that depend on
// each other (loads, the clade bytes of its matches, look-back, flush); with two large tiles per CU neither HBM nor the
// address unit was busy.  2 x 3e8, random taxids: 12 items (two per CU) union 4.41 / inter 4.08 ms, 9: 4.55 / 4.22,
// 7 (three per CU): 3.94 / 3.66, 6: 4.08 / 3.81, 5 (four per CU, 8 waves per SIMD): 4.01 / 3.72.
#ifndef SETOP_VT_TAX
#define SETOP_VT_TAX 7
#endif
constexpr int VT_TAX = SETOP_VT_TAX;
constexpr int VT_RANK = 12;  // ranks ride along (the multiset re-run), with or without taxids

// One pass of the tiled set operation.  result_host[0] = total, [1] = flags.
// (cta, ctb): the file taxid of a stream whose ta / tb is null (SetopArgs); tax && !ta && !tb = the CT instantiation
int run_setop_pass(ukm_ctx *c, int op, const u64 *a, const u32 *ta, const u32 *ra, u64 na,
                   const u64 *b, const u32 *tb, const u32 *rb, u64 nb, bool tax, u32 flags,
                   u64 *out, u32 *tout, u64 out_cap, u64 result_host[2], u32 cta = 0, u32 ctb = 0) {
    const bool rank = ra != nullptr;
    bool ct = tax && !ta && !tb;
    // per-record taxids on plain sets: EITHER the taxid instantiation (on a taxonomy with one-byte clade codes its DEFER
    // form: small tiles, the LCAs walked densely behind the merge loop) OR the plain-key kernel writing a source word per
    // output record + a second launch that turns the words into taxids (tile_flush_src / setop_taxid_gather_kernel).
    // Round 6, 2 x 3e8 records, inter: uniformly random taxids 3.66 against 4.32 ms, taxids that repeat (one per file as
    // arrays, runs of 4096) 2.62 against 2.41 -- the taxid instantiation is the default where its DEFER form exists, the
    // source words where it does not (round 5: 1.60 against 1.78 ms at 2 x 1e8 with the look-ups inside the merge step);
    // union loses through the words by 20 % either way.  UKM_SETOP_SRC: 0 = never, 1 = inter, 2 = every operation but
    // diff -t, whose survivors depend on their taxids; never the multiset re-run with ranks.
    const bool defer_form = SETOP_TAX_DEFER != 0 && c->tax_pair != nullptr && !ukm_env_is(c, "UKM_SETOP_DEFER", '0');
    const int src_mode = ukm_env_int(c, "UKM_SETOP_SRC", defer_form ? 0 : 1);
    if (tax && !ct && !rank && src_mode != 0 && (op == UKM_OP_INTER || (src_mode == 2 && !(op == UKM_OP_DIFF && (flags & UKM_F_CMP_TAXID))))) ct = true;
    if (ct) tax = false;  // the plain-key kernel; the taxids are an epilogue of it
    // (diff WITHOUT -t carries the taxids along and looks nothing up: the large tile, two workgroups per CU -- 2.22 against
    //  2.35 ms at 2 x 3e8 through the small one)
    const bool tax_stream = tax && !rank && op == UKM_OP_DIFF && !(flags & UKM_F_CMP_TAXID);
    const int vt = rank ? VT_RANK : (tax ? (tax_stream ? VT_RANK : VT_TAX) : VT_PLAIN);
    const u64 tile_items = (u64)NTS * vt;
    const u64 N = na + nb;
    SetopArgs p;
    memset(&p, 0, sizeof(p));
    p.a = a; p.b = b; p.ta = ta; p.tb = tb; p.ra = ra; p.rb = rb;
    p.na = na; p.nb = nb;
    p.out = out; p.tout = tout; p.out_cap = out_cap;
    p.ntiles = (N + tile_items - 1) / tile_items;
    p.tax = ukm_taxdev(c);
    p.flags = flags;
    p.cta = cta;
    p.ctb = ctb;
    if (p.ntiles > 0xFFFFFFFFull) UKM_FAIL(UKM_ERR_INVALID, "setop: input too large");

    // control block: [result 2 x u64 | ticket | pad][status: one 64-byte line per tile][mp ntiles+1]
    u64 *ctl = nullptr;
    const size_t nzero = 8 + lb_status_words(p.ntiles);
    UKM_TRY(ws_alloc_t(c, nzero + p.ntiles + 1, &ctl));
    p.result = ctl;
    p.ticket = (u32 *)(ctl + 2);
    p.status = ctl + 8;
    p.mp = ctl + nzero;
    if (tax && !rank && defer_form && (op == UKM_OP_UNION || op == UKM_OP_INTER) && !ukm_env_is(c, "UKM_SETOP_FIX", '0')) {
        UKM_TRY(ws_alloc_t(c, (size_t)p.ntiles * FIX_SLOTS, &p.fix));
        UKM_TRY(ws_alloc_t(c, (size_t)p.ntiles, &p.fix_cnt));
    }
#ifdef UKM_PROFILE_PHASES
    UKM_TRY(ws_alloc_t(c, (size_t)p.ntiles * 8, &p.dbg));
    UKM_HIP(hipMemsetAsync(p.dbg, 0, (size_t)p.ntiles * 8 * sizeof(u64), c->stream));
#endif
    const unsigned pblocks = (unsigned)((p.ntiles + 1 + 255) / 256);
    // Attempt 0 takes tile ids from blockIdx (fast path); if its watchdog fires, attempt 1
    // re-runs with ticketed tile ids, which cannot stall whatever the dispatch order is.
    for (int attempt = c->setop_force_ticket ? 1 : 0; attempt < 2; attempt++) {
        const bool ticket = attempt == 1;
        const bool fused_on = !ukm_env_is(c, "UKM_SETOP_FUSED_PART", '0');  // developer knob
        const bool first = attempt == (c->setop_force_ticket ? 1 : 0);
        if (first && fused_on && p.ntiles >= 4 * PART_COARSE) {
            // status lines, control words and both partition levels in one launch
            const unsigned sblocks = (unsigned)((p.ntiles + PART_COARSE - 1) / PART_COARSE);
            if (rank) hipLaunchKernelGGL((setop_partition_fused_kernel<true>), dim3(sblocks), dim3(256), 0, c->stream, p, (int)tile_items, 8u);
            else hipLaunchKernelGGL((setop_partition_fused_kernel<false>), dim3(sblocks), dim3(256), 0, c->stream, p, (int)tile_items, 8u);
            UKM_HIP(hipGetLastError());
        } else {
        UKM_HIP(hipMemsetAsync(ctl, 0, nzero * sizeof(u64), c->stream));
        if (first) {
            if (p.ntiles >= 4 * PART_COARSE) {
                const unsigned cblocks = (unsigned)((p.ntiles / PART_COARSE + 2 + 3) / 4);  // one wave per coarse boundary
                if (rank) {
                    hipLaunchKernelGGL((setop_partition_coop_kernel<true, 1>), dim3(cblocks), dim3(256), 0, c->stream, p, (int)tile_items);
                    hipLaunchKernelGGL((setop_partition_kernel<true, 2>), dim3(pblocks), dim3(256), 0, c->stream, p, (int)tile_items);
                } else {
                    hipLaunchKernelGGL((setop_partition_coop_kernel<false, 1>), dim3(cblocks), dim3(256), 0, c->stream, p, (int)tile_items);
                    hipLaunchKernelGGL((setop_partition_kernel<false, 2>), dim3(pblocks), dim3(256), 0, c->stream, p, (int)tile_items);
                }
            } else {
                const unsigned wblocks = (unsigned)((p.ntiles + 1 + 3) / 4);  // one wave per boundary
                if (rank) hipLaunchKernelGGL((setop_partition_coop_kernel<true, 0>), dim3(wblocks), dim3(256), 0, c->stream, p, (int)tile_items);
                else hipLaunchKernelGGL((setop_partition_coop_kernel<false, 0>), dim3(wblocks), dim3(256), 0, c->stream, p, (int)tile_items);
            }
        }
        }
        (void)hipEventRecord(c->ev_k0, c->stream);
        if (rank) {
            if (tax) launch_op<true, true, NTS, VT_RANK>(op, p, c->stream, ticket);
            else if (ct) launch_op<false, true, NTS, VT_RANK, true>(op, p, c->stream, ticket);
            else launch_op<false, true, NTS, VT_RANK>(op, p, c->stream, ticket);
        } else {
            // union / inter with per-record taxids and one-byte clade codes: the LCAs behind the merge loop (tile_merge_loop_deferred)
            if (tax_stream) launch_tile<UKM_OP_DIFF, true, false, NTS, VT_RANK>(p, c->stream, ticket);
            else if (tax && defer_form) launch_op<true, false, NTS, VT_TAX, false, true>(op, p, c->stream, ticket);
            else if (tax) launch_op<true, false, NTS, VT_TAX>(op, p, c->stream, ticket);
            else if (ct) launch_op<false, false, NTS, VT_PLAIN, true>(op, p, c->stream, ticket);
            else launch_op<false, false, NTS, VT_PLAIN>(op, p, c->stream, ticket);
        }
        (void)hipEventRecord(c->ev_k1, c->stream);
        c->evk_valid = true;
        UKM_HIP(hipGetLastError());
        UKM_TRY(ukm_read_u64(c, p.result, result_host, 2));
        if (!(result_host[1] & FLAG_TIMEOUT)) break;
        if (ticket) UKM_FAIL(UKM_ERR_HIP, "setop: look-back watchdog fired in the ticketed kernel");
        ukm_switch_to_tickets(c, "set-op kernel");  // this device does not dispatch in order: stay on tickets
    }
#ifdef UKM_PROFILE_PHASES
    {
        std::vector<u64> h((size_t)p.ntiles * 8);
        UKM_HIP(hipMemcpy(h.data(), p.dbg, h.size() * sizeof(u64), hipMemcpyDeviceToHost));
        double sum[8] = {0};
        for (u64 b = 0; b < p.ntiles; b++) for (int i = 0; i < 8; i++) sum[i] += (double)h[(size_t)b * 8 + i];
        double tot = 0; for (int i = 0; i < 7; i++) tot += sum[i];
        fprintf(stderr, "[phases op=%d tiles=%llu] cycles per tile:", op, (unsigned long long)p.ntiles);
        for (int i = 0; i < 7; i++) fprintf(stderr, " p%d=%.0f", i, sum[i] / sum[7]);
        fprintf(stderr, " total=%.0f\n", tot / sum[7]);
    }
#endif
    return UKM_OK;
}

}  // namespace

// internal entry: all pointers are device pointers
// One link of a chained fold: A's size is *na_dev (<= na_max), nothing is read back.  `ctl` = 8 zeroed words
// owned by the caller ([0] result count, [1] flags, [2] ticket) that stay valid until the chain is read;
// status lines and partition points come from the arena (the caller marks / releases around the call: the
// stream orders the next link's memset after this link's kernels).  Plain sets only (no rank path).
int ukm_dev_setop2_link(ukm_ctx *c, int op, const u64 *a, const u32 *ta, u64 na_max, const u64 *na_dev, const u64 *b,
                        const u32 *tb, u64 nb, u32 flags, u64 *out, u32 *tout, u64 out_cap, u64 *ctl, u32 cta, u32 ctb) {
    bool tax = (ta != nullptr) || (tb != nullptr) || cta != 0 || ctb != 0;
    const bool ct = tax && !ta && !tb;  // both streams carry one taxid per file: plain kernel + taxid epilogue (ctl[4])
    if (ct) tax = false;
    const int vt = tax ? VT_TAX : VT_PLAIN;
    const u64 tile_items = (u64)NTS * vt;
    SetopArgs p;
    memset(&p, 0, sizeof(p));
    p.a = a; p.b = b; p.ta = ta; p.tb = tb;
    p.na = na_max; p.nb = nb; p.na_dev = na_dev;
    p.out = out; p.tout = tout; p.out_cap = out_cap;
    p.ntiles = (na_max + nb + tile_items - 1) / tile_items;
    p.tax = ukm_taxdev(c);
    p.flags = flags;
    p.cta = cta;
    p.ctb = ctb;
    if (p.ntiles == 0) return UKM_OK;
    u64 *st = nullptr;
    const unsigned pblocks = (unsigned)((p.ntiles + 1 + 255) / 256);
#ifndef SETOP_LINK_SMALL_TILES
#define SETOP_LINK_SMALL_TILES 2048
#endif
    // links of up to a few thousand tiles: ONE wave-cooperative partition kernel that also clears the status lines
    // (three launches less per link than memset + two-level partition; a 1000-file fold is launch bound)
    const bool small = p.ntiles < SETOP_LINK_SMALL_TILES;
    const size_t nstat = lb_status_words(p.ntiles + 1);
    UKM_TRY(ws_alloc_t(c, nstat + p.ntiles + 1, &st));
    if (small) p.zero_status = (u32)(p.ntiles + 1);  // cleared by the partition kernel: one launch less
    else UKM_HIP(hipMemsetAsync(st, 0, nstat * sizeof(u64), c->stream));
    p.result = ctl;
    p.ticket = (u32 *)(ctl + 2);
    p.status = st;
    p.mp = st + nstat;
    if (!small) {
        const unsigned cblocks = (unsigned)((p.ntiles / PART_COARSE + 2 + 3) / 4);  // one wave per coarse boundary
        hipLaunchKernelGGL((setop_partition_coop_kernel<false, 1>), dim3(cblocks), dim3(256), 0, c->stream, p, (int)tile_items);
        hipLaunchKernelGGL((setop_partition_kernel<false, 2>), dim3(pblocks), dim3(256), 0, c->stream, p, (int)tile_items);
    } else {
        const unsigned wblocks = (unsigned)((p.ntiles + 1 + 3) / 4);  // one wave per boundary
        hipLaunchKernelGGL((setop_partition_coop_kernel<false, 0>), dim3(wblocks), dim3(256), 0, c->stream, p, (int)tile_items);
    }
    if (tax) launch_op<true, false, NTS, VT_TAX>(op, p, c->stream, c->setop_force_ticket);
    else if (ct) launch_op<false, false, NTS, VT_PLAIN, true>(op, p, c->stream, c->setop_force_ticket);
    else {
        launch_op<false, false, NTS, VT_PLAIN>(op, p, c->stream, c->setop_force_ticket);
        // (neither stream carries taxids but the fold's later files do: these records have taxid 0)
        if (tout && out_cap) UKM_HIP(hipMemsetAsync(tout, 0, out_cap * sizeof(u32), c->stream));
    }
    UKM_HIP(hipGetLastError());
    return UKM_OK;
}

int ukm_dev_setop2(ukm_ctx *c, int op, const u64 *a, const u32 *ta, u64 na, const u64 *b,
                   const u32 *tb, u64 nb, u32 flags, u64 *out, u32 *tout, u64 out_cap, u64 *n_out) {
    return ukm_dev_setop2_ct(c, op, a, ta, 0u, na, b, tb, 0u, nb, flags, out, tout, out_cap, n_out);
}

// cta / ctb: the file taxid of a stream without per-record taxids (ta / tb null); 0 = the stream has no taxid at all
int ukm_dev_setop2_ct(ukm_ctx *c, int op, const u64 *a, const u32 *ta, u32 cta, u64 na, const u64 *b,
                      const u32 *tb, u32 ctb, u64 nb, u32 flags, u64 *out, u32 *tout, u64 out_cap, u64 *n_out) {
    if (op != UKM_OP_UNION && op != UKM_OP_INTER && op != UKM_OP_DIFF && op != UKM_OP_MERGE_INTERNAL)
        UKM_FAIL(UKM_ERR_INVALID, "ukm_setop2: unknown op %d", op);
    if (ta) cta = 0;
    if (tb) ctb = 0;
    const bool tax = (ta != nullptr) || (tb != nullptr) || cta != 0 || ctb != 0;
    if (tax && !tout) UKM_FAIL(UKM_ERR_INVALID, "ukm_setop2: taxids given but out_taxids is NULL");
    const bool need_lca = tax && op != UKM_OP_MERGE_INTERNAL && (op != UKM_OP_DIFF || (flags & UKM_F_CMP_TAXID));
    if (need_lca && c->tax_parent == nullptr)
        UKM_FAIL(UKM_ERR_NO_TAXONOMY, "ukm_setop2: records carry taxids but no taxonomy is loaded");
    *n_out = 0;
    if (na + nb == 0) return UKM_OK;
    if (na + nb >= (1ull << 61)) UKM_FAIL(UKM_ERR_INVALID, "ukm_setop2: input too large");

    u64 res[2] = {0, 0};
    UKM_TRY(run_setop_pass(c, op, a, ta, nullptr, na, b, tb, nullptr, nb, tax, flags, out, tout,
                           out_cap, res, cta, ctb));
    if (res[1] & FLAG_UNSORTED) UKM_FAIL(UKM_ERR_UNSORTED, "ukm_setop2: an input stream is not sorted");
    if ((res[1] & FLAG_DUP) && op != UKM_OP_MERGE_INTERNAL) {
        // multiset inputs: redo with the exact reference semantics
        if (op == UKM_OP_UNION) {
            // union is a set: fold duplicates (LCA) inside each input first, then merge
            u64 *ua = nullptr, *ub = nullptr;
            u32 *uta = nullptr, *utb = nullptr;
            u64 nua = 0, nub = 0;
            UKM_TRY(ws_alloc_t(c, na + 1, &ua));
            UKM_TRY(ws_alloc_t(c, nb + 1, &ub));
            if (ta) UKM_TRY(ws_alloc_t(c, na + 1, &uta));
            if (tb) UKM_TRY(ws_alloc_t(c, nb + 1, &utb));
            // (a stream with one taxid per file keeps it: LCA(x, x) = x)
            UKM_TRY(ukm_dev_unique(c, a, ta, na, UKM_UNIQUE, ua, uta, na, &nua));
            UKM_TRY(ukm_dev_unique(c, b, tb, nb, UKM_UNIQUE, ub, utb, nb, &nub));
            UKM_TRY(run_setop_pass(c, op, ua, uta, nullptr, nua, ub, utb, nullptr, nub, tax, flags,
                                   out, tout, out_cap, res, cta, ctb));
        } else {
            u32 *ra = nullptr, *rb = nullptr;
            UKM_TRY(ws_alloc_t(c, na + 1, &ra));
            UKM_TRY(ws_alloc_t(c, nb + 1, &rb));
            if (na) hipLaunchKernelGGL(rank_in_run_kernel, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, c->stream, a, na, ra);
            if (nb) hipLaunchKernelGGL(rank_in_run_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, c->stream, b, nb, rb);
            if (op == UKM_OP_DIFF && !(flags & UKM_F_INTERNAL_KEEP_DUPS)) {
                // the reference's survivor map collapses duplicate codes (diff.go:449-453);
                // the last record of a run wins.  (ONE later file: the n-file fold keeps the running list as it is between
                // its files -- mc1 = mc2, diff.go:437 -- and collapses once at the end: UKM_F_INTERNAL_KEEP_DUPS)
                u64 *tk = nullptr;
                u32 *tt = nullptr;
                UKM_TRY(ws_alloc_t(c, na + 1, &tk));
                if (tax) UKM_TRY(ws_alloc_t(c, na + 1, &tt));
                UKM_TRY(run_setop_pass(c, op, a, ta, ra, na, b, tb, rb, nb, tax, flags, tk, tt, na, res, cta, ctb));
                if (!(res[1] & FLAG_UNSORTED)) {
                    u64 nu = 0;
                    UKM_TRY(ukm_dev_unique(c, tk, tax ? tt : nullptr, res[0], 5 /*UNIQUE_LAST*/, out, tout, out_cap, &nu));
                    res[0] = nu;
                }
            } else {
                UKM_TRY(run_setop_pass(c, op, a, ta, ra, na, b, tb, rb, nb, tax, flags, out, tout,
                                       out_cap, res, cta, ctb));
            }
        }
        if (res[1] & FLAG_UNSORTED) UKM_FAIL(UKM_ERR_UNSORTED, "ukm_setop2: an input stream is not sorted");
    }
    *n_out = res[0];
    if (res[0] > out_cap)
        UKM_FAIL(UKM_ERR_CAPACITY, "ukm_setop2: output needs %llu records, capacity is %llu",
                 (unsigned long long)res[0], (unsigned long long)out_cap);
    // the caller wants taxids but neither stream carries any (two files without taxid information inside an n-file
    // operation whose other files have some; the only stream with taxids is empty): such records have taxid 0
    if (!tax && tout && res[0]) UKM_HIP(hipMemsetAsync(tout, 0, res[0] * sizeof(u32), c->stream));
    return UKM_OK;
}

extern "C" int ukm_setop2_ft(ukm_ctx *ctx, int op, const uint64_t *a_keys, const uint32_t *a_taxids, uint32_t a_file_taxid,
                             uint64_t na, const uint64_t *b_keys, const uint32_t *b_taxids, uint32_t b_file_taxid,
                             uint64_t nb, uint32_t flags, uint64_t *out_keys, uint32_t *out_taxids,
                             uint64_t out_cap, uint64_t *n_out) {
    if (!ctx || !n_out || (!a_keys && na) || (!b_keys && nb) || (!out_keys && out_cap))
        UKM_FAIL(UKM_ERR_INVALID, "ukm_setop2: NULL argument");
    if (op != UKM_OP_UNION && op != UKM_OP_INTER && op != UKM_OP_DIFF)
        UKM_FAIL(UKM_ERR_INVALID, "ukm_setop2: unknown op %d", op);
    CallScope s;
    UKM_TRY(ukm_begin(ctx, &s));
    int rc = [&]() -> int {
        const u64 *a = nullptr, *b = nullptr;
        const u32 *ta = nullptr, *tb = nullptr;
        u64 *out = nullptr;
        u32 *tout = nullptr;
        UKM_TRY(ukm_in_t(ctx, a_keys, na, &a));
        UKM_TRY(ukm_in_t(ctx, b_keys, nb, &b));
        UKM_TRY(ukm_in_t(ctx, a_taxids, na, &ta));
        UKM_TRY(ukm_in_t(ctx, b_taxids, nb, &tb));
        UKM_TRY(ukm_out_t(ctx, out_keys, out_cap, &out));
        UKM_TRY(ukm_out_t(ctx, out_taxids, out_cap, &tout));
        // (an empty stream's per-record pointer may be null: its file taxid plays no part then)
        const u32 cta = ta ? 0u : a_file_taxid, ctb = tb ? 0u : b_file_taxid;
        int r = ukm_dev_setop2_ct(ctx, op, a, ta, cta, na, b, tb, ctb, nb, flags & (UKM_F_MIX_TAXID | UKM_F_CMP_TAXID), out, tout, out_cap, n_out);
        u64 n = (r == UKM_OK) ? *n_out : 0;
        ukm_out_resize(ctx, out_keys, n * sizeof(u64));
        if (out_taxids) ukm_out_resize(ctx, out_taxids, n * sizeof(u32));
        return r;
    }();
    return ukm_finish(&s, rc);
}

extern "C" int ukm_setop2(ukm_ctx *ctx, int op, const uint64_t *a_keys, const uint32_t *a_taxids,
                          uint64_t na, const uint64_t *b_keys, const uint32_t *b_taxids,
                          uint64_t nb, uint32_t flags, uint64_t *out_keys, uint32_t *out_taxids,
                          uint64_t out_cap, uint64_t *n_out) {
    return ukm_setop2_ft(ctx, op, a_keys, a_taxids, 0u, na, b_keys, b_taxids, 0u, nb, flags, out_keys, out_taxids, out_cap, n_out);
}

extern "C" int ukm_partition_points(ukm_ctx *ctx, const uint64_t *keys, uint64_t n,
                                    const uint64_t *splitters, int n_split, uint64_t *cuts) {
    if (!ctx || (!keys && n) || !splitters || !cuts || n_split <= 0)
        UKM_FAIL(UKM_ERR_INVALID, "ukm_partition_points: bad argument");
    CallScope s;
    UKM_TRY(ukm_begin(ctx, &s));
    int rc = [&]() -> int {
        const u64 *k = nullptr, *q = nullptr;
        u64 *o = nullptr;
        UKM_TRY(ukm_in_t(ctx, keys, n, &k));
        UKM_TRY(ukm_in_t(ctx, splitters, (u64)n_split, &q));
        UKM_TRY(ukm_out_t(ctx, cuts, (u64)n_split, &o));
        hipLaunchKernelGGL(lower_bound_kernel, dim3((n_split + 63) / 64), dim3(64), 0, ctx->stream, k, n, q, n_split, o);
        UKM_HIP(hipGetLastError());
        return UKM_OK;
    }();
    return ukm_finish(&s, rc);
}
