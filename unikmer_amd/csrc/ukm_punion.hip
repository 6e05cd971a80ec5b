// ukm_punion.hip — `union` of MANY sorted sets that overlap heavily, the shape of an n-file `unikmer union` over
// related genomes (BASELINE config 3: 100 files drawn from one universe).  The reference answers every k-mer of every
// file with one probe of a hash map (union.go:186-208, 225-246); the k-way streaming merge of ukm_kway.hip pays three
// in-LDS merge rounds per input record instead (VALU bound: 119 lane-instructions per record, 34 ms of the 46.7 ms of
// config 3 for level 0 alone) although after the first few files nearly every record is already in the result.
//
// Here the reference's algorithm is laid out for the chip:
//   1. BASE  = k-way union of the first PU_K0 files (ukm_kway.hip): a sorted, duplicate-free set.
//   2. A sample of later records is looked up in BASE (global binary search): when fewer than PU_MIN_HIT of them are
//      found the inputs do not have this shape and the caller's k-way merge answers.
//   3. The VALUE SPACE is cut into ranges of PU_RANGE consecutive BASE entries.  One workgroup per range builds a
//      bucketised table of its entries in LDS (2048 buckets of four, 64 KB) and streams through its slice of EVERY later
//      file (lower-bound cuts of the range limits, one thread per (range, file)): per record one multiplicative hash
//      and one 32-byte bucket read; the order of every file is checked on the way (neighbouring records are compared once).
//      Records that are not in the table — not in BASE — are appended to a miss list (one atomic per 64 slots).
//   4. Result = 2-way union of BASE and sort + unique of the miss list (ukm_sort.hip, ukm_scan.hip, ukm_setops.hip).
// Whatever the data, BASE ∪ later records = BASE ∪ misses, because a hit is an exact 64-bit match; a bad hash or an
// unlucky range only costs probes.  An unsorted file or a full miss list raise a flag and the caller falls back.
// Records WITH TaxIds (round 4, pt_probe_kernel below): every table entry carries the TaxId it came with and the smallest /
// largest pre-order number of the records that differ from it; one table LCA per entry when its range is done.
// Algorithmic bytes: 8 B (12 B with TaxIds) per input record read once (+ the base and miss passes); nothing is written
// per hit.
#include <stdlib.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <vector>

#include "ukm_device.h"
#include "ukm_kway.h"
#include "ukm_punion.h"

namespace {

#ifndef PU_K0_N
#define PU_K0_N 8
#endif
constexpr int PU_K0 = PU_K0_N;    // files merged into the base set
#ifndef PU_WAVES
#define PU_WAVES 4
#endif
constexpr int PU_RANGE = 2048;    // base entries per range
constexpr int PU_BUCKET_BITS = 11;  // 2048 buckets x 4 slots x 8 B = 64 KB of LDS: two workgroups per CU
constexpr int PU_BUCKETS = 1 << PU_BUCKET_BITS;
constexpr int PU_SLOTS = 4 * PU_BUCKETS;
constexpr int PU_MAXS = 4096;     // later files per launch
constexpr int PU_LMISS = 512;     // new codes a range keeps in LDS before they go out in one piece
constexpr u32 PU_CHUNK = 32;      // slots of the miss list a wave reserves at a time
constexpr u64 PU_EMPTY = ~0ull;
constexpr double PU_MIN_HIT = 0.55;  // (a table takes as many new codes as it has base entries: see PT_MIN_HIT)
enum { PU_FLAG_UNSORTED = 1, PU_FLAG_OVERFLOW = 2, PU_FLAG_TAXID = 4, PU_FLAG_RAW = 8 };

struct PuArgs {
    const u64 *const *files;  // [S1] later files (device table of device pointers)
    const u64 *lens;          // [S1]
    u32 S1;
    const u64 *base;          // sorted, duplicate-free
    u64 n0;
    u32 R;                    // ranges = ceil(n0 / PU_RANGE)
    u64 *cuts;                // [R + 1][S1]
    u64 *miss;
    u64 miss_cap;
    u64 *ctl;                 // [0] misses, [1] flags, [2] sample hits, [3] samples
    u32 range;                // base entries per range (PU_RANGE; with TaxIds PT_RANGE)
    // with TaxIds (pt_probe_kernel)
    const u32 *const *tfiles; // [S1] TaxIds of the later files (an entry may be null: all 0)
    u32 *base_tax;            // [n0] in: the fold over the base files, out: over every file
    u32 *miss_tax;            // beside `miss`
    unsigned short *rec_idx;  // placement merge: [all records, file by file] the record's code as an index into its range
    const u64 *rec_off;       // [S1]: where file j's records begin in rec_idx
    u32 threshold;            // COUNT (`common`): a code leaves when at least this many records carried it
    u32 count0;               // COUNT: records a base entry starts with (1: the base set is the first file; 0: every file is probed)
    // files with ONE taxid each (round 5; the .unik header's global taxid): tfiles[j] is null and cte[j] = taxid | its
    // pre-order number << 32 (pu_cte_kernel) -- a slice of such a file loads no taxids and looks no number up
    const u64 *cte;           // [S1], or null: files without per-record taxids have taxid 0
    u32 base_ct;              // COUNT with the first file as the base set and no base_tax: its file taxid
    // Round 5, pt_probe_kernel: the taxids of the later records look UNRELATED to the entries' (the sample: ctl[6]): a record
    // brings the one-byte CLADE code of its taxid instead of the 4-byte pre-order number (see the kernel's fold)
    u32 clade_mode;
    TaxDev tax;
};

__device__ __forceinline__ u32 pu_hash(u64 x) {
    const u32 lo = (u32)x, hi = (u32)(x >> 32);
    return ((lo ^ __builtin_rotateleft32(hi, 15) ^ (hi >> 3)) * 0x9E3779B1u) >> (32 - PU_BUCKET_BITS);
}

__device__ __forceinline__ u64 pu_splitmix(u64 x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// cuts[r][j] = lower bound of the first base entry of range r in later file j (r = 0: 0, r = R: the file's length).
// (Bracketing every cut around its interpolated position made the kernel slower in round 3, 1.8 -> 2.9 ms; a two-level
//  search -- every 64th range, then an interpolated window between two coarse cuts -- measured the same 1.76 ms in round 6:
//  the kernel is bound by the ~8 cold lines of a search's last levels, which either form still touches.)
__global__ void pu_cuts_kernel(PuArgs a) {
    // a block = 16 ranges x 16 files; 16 neighbouring lanes hold one range's cuts in 16 consecutive files: one 128-byte store
    // (with the threads of a block on 256 ranges of ONE file every store was a line of its own: 1.77 -> 1.63 ms on config 3)
    const u32 tiles_j = (a.S1 + 15) / 16;
    const u32 tr = blockIdx.x / tiles_j, tj = blockIdx.x % tiles_j;
    const u32 j = tj * 16 + (threadIdx.x & 15), r = tr * 16 + (threadIdx.x >> 4);
    if (j >= a.S1 || r > a.R) return;
    const u64 len = a.lens[j];
    u64 res;
    if (r == 0) {
        res = 0;
    } else if (r == a.R) {
        res = len;
    } else {
        const u64 v = a.base[(u64)r * a.range];
        const auto f = as_global(a.files[j]);
        u64 lo = 0, hi = len;
        while (lo < hi) {
            const u64 mid = (lo + hi) >> 1;
            if (f[mid] < v) lo = mid + 1; else hi = mid;
        }
        res = lo;
    }
    a.cuts[(u64)r * a.S1 + j] = res;
}

static int pu_launch_cuts(ukm_ctx *c, const PuArgs &a) {
    const u64 blocks = (((u64)a.R + 1 + 15) / 16) * (((u64)a.S1 + 15) / 16);
    hipLaunchKernelGGL(pu_cuts_kernel, dim3((unsigned)blocks), dim3(256), 0, c->stream, a);
    UKM_HIP(hipGetLastError());
    return UKM_OK;
}

// heaviest range: records of all later files inside one range (ctl[4] = max over the ranges)
__global__ void pu_load_kernel(PuArgs a) {
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.R) return;
    u64 sum = 0;
    for (u32 j = 0; j < a.S1; j++) {
        const u64 b = a.cuts[(u64)r * a.S1 + j], e = a.cuts[(u64)(r + 1) * a.S1 + j];
        sum += e > b ? e - b : 0;
    }
    atomicMax((unsigned long long *)&a.ctl[4], (unsigned long long)sum);
}

// hit rate of a sample of later records in the base set
__global__ void pu_sample_kernel(PuArgs a, u32 nsamp, u32 nfiles_s) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    bool tested = false, hit = false, same = false, run = false;
    if (i < nsamp) {
        const u32 j = (u32)(((u64)(i % nfiles_s) * a.S1) / nfiles_s);
        const u64 len = a.lens[j];
        if (len) {
            // sample t of its file sits in the t-th of spf equal strides, at a hashed place inside it: no record is drawn
            // twice (pu_new_codes counts EQUAL sampled records; drawing with replacement showed it pairs that are one record)
            const u64 spf = (nsamp + nfiles_s - 1) / nfiles_s, t = i / nfiles_s;
            const u64 lo_p = (u64)(((unsigned __int128)t * len) / spf), hi_p = (u64)(((unsigned __int128)(t + 1) * len) / spf);
            const u64 key = as_global(a.files[j])[hi_p > lo_p ? lo_p + pu_splitmix(i) % (hi_p - lo_p) : (lo_p < len ? lo_p : len - 1)];
            u64 lo = 0, hi = a.n0;
            while (lo < hi) {
                const u64 mid = (lo + hi) >> 1;
                if (a.base[mid] < key) lo = mid + 1; else hi = mid;
            }
            tested = true;
            hit = lo < a.n0 && a.base[lo] == key;
            // (records with taxids: does the record's taxid differ from the entry's and lie in the entry's clade?  Those are
            //  the records that need their exact pre-order number in the fold: PuArgs::clade_mode)
            if (hit && a.tax.clade8 && a.tfiles && (a.base_tax || a.base_ct)) {
                const u32 *tf = a.tfiles[j];
                const u64 at = hi_p > lo_p ? lo_p + pu_splitmix(i) % (hi_p - lo_p) : (lo_p < len ? lo_p : len - 1);
                const u32 t = tf ? tf[at] : (a.cte ? (u32)a.cte[j] : 0u);
                const u32 bt = a.base_tax ? a.base_tax[lo] : a.base_ct;
                same = t != bt && t < a.tax.size && bt < a.tax.size && a.tax.clade8[t] == a.tax.clade8[bt];
                // (and does the file's NEXT record carry the same taxid?  Files whose neighbouring records share their taxid -- one
                //  taxid per genome, taxids assigned by clade -- read the 4-byte numbers from lines they have just used)
                run = !tf || (at + 1 < len && tf[at + 1] == t);
            }
            // (the sampled records the base set lacks are kept: how many DISTINCT new codes the files bring is read off
            //  the equal pairs among them, pu_new_codes)
            if (!hit && a.miss) {
                const u64 at = atomicAdd((unsigned long long *)&a.ctl[5], 1ull);
                if (at < a.miss_cap) a.miss[at] = key;
            }
        }
    }
    const u64 mh = __ballot(hit), mt = __ballot(tested), ms = __ballot(same), mr = __ballot(run);
    if (lane_id() == 0 && mt) {
        atomicAdd((unsigned long long *)&a.ctl[2], (unsigned long long)__popcll(mh));
        atomicAdd((unsigned long long *)&a.ctl[3], (unsigned long long)__popcll(mt));
        if (ms) atomicAdd((unsigned long long *)&a.ctl[6], (unsigned long long)__popcll(ms));
        if (mr) atomicAdd((unsigned long long *)&a.ctl[7], (unsigned long long)__popcll(mr));
    }
}

// cte[j]: the file taxid in the low word (host) gets its pre-order number in the high word
__global__ void pu_cte_kernel(u64 *cte, u32 n, TaxDev T, u32 clade_mode = 0) {
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const u32 t = (u32)cte[j];
    u32 e = T.euler ? T.euler[t < T.size ? t : 0u] : 0u;
    if (clade_mode && e) e |= (u32)T.clade8[t] << 24;  // (PuArgs::clade_mode: the numbers carry their clade code)
    cte[j] = (u64)t | ((u64)e << 32);
}

typedef u64 pu_u64x2 __attribute__((ext_vector_type(2)));
typedef pu_u64x2 __attribute__((aligned(8))) pu_pair;  // 16 bytes at 8-byte alignment

// PuArgs::clade_mode from the sample (hits: sampled later records found in the base set, same: those of them whose taxid
// differs from the entry's and lies in the entry's clade -- the records whose exact number the fold would have to fetch on
// the spot).  Unrelated taxa: next to none.  Related taxa (one species' strains): most -- the numbers are then read for
// every record in the pipeline's second stage, as in rounds 4-5.  UKM_PUNION_CLADE=0 / 1: never / always.
// runs: hits whose file's next record carries the same taxid: with most of them the numbers come from lines the wave has just
// used and the plain fold is the faster one (config 3's files with one taxid each as arrays: probe pass 27.5 ms against 34.0 in
// clade mode; uniformly random taxids: 67.8 against 41.7).
static u32 pu_clade_mode(const ukm_ctx *c, const TaxDev &T, bool tax, u64 hits, u64 same, u64 runs) {
    if (!tax || T.clade8 == nullptr || T.pair == nullptr || T.euler == nullptr) return 0u;
    const int k = ukm_env_int(c, "UKM_PUNION_CLADE", -1);
    if (k == 0) return 0u;
    if (k == 1) return 1u;
    return (hits > 0 && same * 16 < hits && runs * 2 < hits) ? 1u : 0u;
}

// ---- the plain pass (pu2_probe_kernel, round 6; rounds 3-5: pu_probe_kernel) -------------------------------------------------
// What bound pu_probe_kernel (profiles/r03_punion_pmc.txt, r05_notes.md 1c): it had three step shapes (U = 4 / 2 / 1) x two
// validity paths with the list code inlined in each -- 16,000 lines of ISA --, a THIRD load per lane and step for the order
// check (the record behind a lane's pair), 64-bit clamps on every address, 45 scalar instructions per record of slice
// bookkeeping, and four waves per SIMD at 93 registers: 16.5 ms on config 3 (4.5 TB/s).  Here (12.9 ms, 5.7 TB/s; what was
// measured on the way: profiles/r06_notes.md 1):
//   * ONE step shape: 128 records, two per lane, one 16-byte load per lane from a wave-uniform base + a 32-bit lane offset.
//     A slice = one general first step, batches of FULL steps (no validity masks, the batch's loads in flight together,
//     straight-line code: the compiler's s_waitcnt counts are exact), general steps for what is left;
//   * the order check needs no third load: the record in front of a lane's pair is its neighbour's second record (DPP
//     wave_shr:1), lane 0 takes the previous step's last record from a scalar; a slice that does not begin its file starts
//     one record early, so the boundary pair is checked inside lane 0 like every other pair;
//   * 1024 threads share the 64 KB table: two workgroups = eight waves per SIMD (55 registers);
//   * the table in two halves, P pair first (below); the list / claim code exists once, behind one wave-uniform branch.
// (A pipeline ACROSS slices with loads in inline assembly and hand-written waits was built first and measured the same for
//  2 / 3 / 4 steps in flight: latency was not the limit -- and inline-assembly loads hide hazards from the compiler, see the
//  notes.)
#ifndef PU2_NT_N
#define PU2_NT_N 1024
#endif
#ifndef PU2_U_N
#define PU2_U_N 2
#endif
constexpr int PU2_NT = PU2_NT_N;
constexpr int PU2_U = PU2_U_N;  // steps of a batch: their loads are in flight together
constexpr int PU2_WAVES = PU2_NT == 1024 ? 8 : 4;

__device__ __forceinline__ u64 pu2_shr1(u64 v, u64 carry) {  // lane l gets v of lane l - 1, lane 0 gets `carry`
    const u32 lo = (u32)__builtin_amdgcn_update_dpp((int)(u32)carry, (int)(u32)v, 0x138, 0xF, 0xF, false);          // wave_shr:1
    const u32 hi = (u32)__builtin_amdgcn_update_dpp((int)(u32)(carry >> 32), (int)(u32)(v >> 32), 0x138, 0xF, 0xF, false);
    return ((u64)hi << 32) | lo;
}

__global__ __launch_bounds__(PU2_NT) __attribute__((amdgpu_waves_per_eu(PU2_WAVES, PU2_WAVES))) void pu2_probe_kernel(PuArgs a) {
    // the table in two halves: slots 0 and 1 of every bucket in the first 32 KB (P), slots 2 and 3 behind them (Q).  Slots
    // fill in order, so a record is looked up in its bucket's P pair first (one 16-byte read at a 16-byte stride: all bank
    // groups in use) and only the lanes that did not find it there AND see slot 1 taken read the Q pair: a tenth of them.
    __shared__ __attribute__((aligned(32))) u64 s_tab[PU_SLOTS];
    __shared__ u64 s_miss[PU_LMISS];
    auto slot = [&](u32 h, int q) -> u64 * { return &s_tab[(q >> 1) * (PU_SLOTS / 2) + 2 * h + (q & 1)]; };
    __shared__ u32 s_next, s_nmiss, s_nins;
    __shared__ u64 s_flush_at;
    const int tid = (int)threadIdx.x, lane = lane_id();
    const u32 r = blockIdx.x, S1 = a.S1;
    for (int i = tid; i < PU_SLOTS; i += PU2_NT) s_tab[i] = PU_EMPTY;
    if (tid == 0) { s_next = 0; s_nmiss = 0; s_nins = 0; }
    __syncthreads();
    {   // the table of this range's base entries (distinct; an all-ones code can not be told from an empty slot and is left
        // out: records with that code are "misses" and meet their base entry again in the final union)
        const u64 b0 = (u64)r * PU_RANGE;
        const u32 nb = (u32)((a.n0 - b0 < (u64)PU_RANGE) ? (a.n0 - b0) : (u64)PU_RANGE);
        constexpr int PER = (PU_RANGE + PU2_NT - 1) / PU2_NT;
        u64 ent[PER];
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const u32 idx = (u32)tid + (u32)i * PU2_NT;
            ent[i] = a.base[b0 + (idx < nb ? idx : 0)];
            if (idx >= nb) ent[i] = PU_EMPTY;
        }
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const u64 e = ent[i];
            if (e == PU_EMPTY) continue;
            u32 h = pu_hash(e);
            for (bool placed = false; !placed; h = (h + 1) & (PU_BUCKETS - 1)) {
#pragma unroll
                for (int k = 0; k < 4 && !placed; k++) {
                    const u64 old = atomicCAS((unsigned long long *)slot(h, k), (unsigned long long)PU_EMPTY, (unsigned long long)e);
                    placed = old == PU_EMPTY || old == e;
                }
            }
        }
    }
    __syncthreads();
    auto member_from = [&](u64 x, u32 h) -> bool {
        for (;;) {
            const ulonglong2 p = *reinterpret_cast<const ulonglong2 *>(slot(h, 0)), q = *reinterpret_cast<const ulonglong2 *>(slot(h, 2));
            if (p.x == x || p.y == x || q.x == x || q.y == x) return x != PU_EMPTY;
            if (q.y == PU_EMPTY) return false;
            h = (h + 1) & (PU_BUCKETS - 1);
        }
    };
    auto claim = [&](u64 x) -> bool {
        if (x == PU_EMPTY || s_nins >= (u32)PU_RANGE) return true;
        u32 h = pu_hash(x);
        for (;; h = (h + 1) & (PU_BUCKETS - 1)) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const u64 old = atomicCAS((unsigned long long *)slot(h, k), (unsigned long long)PU_EMPTY, (unsigned long long)x);
                if (old == PU_EMPTY) { atomicAdd(&s_nins, 1u); return true; }
                if (old == x) return false;
            }
        }
    };
    const u64 lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    u64 chunk_at = 0, fill = 0;  // wave-uniform
    u32 chunk_cap = 0, chunk_used = 0;
    auto close_chunk = [&]() {
        if ((u32)lane < chunk_cap - chunk_used) a.miss[chunk_at + chunk_used + (u32)lane] = fill;
        chunk_cap = chunk_used = 0;
    };
    auto append_global = [&](bool m, u64 x) {
        const u64 mask = __ballot(m);
        if (mask == 0ull) return;
        const u32 n = (u32)__popcll(mask);
        const int lead = __ffsll((long long)mask) - 1;
        if (n > chunk_cap - chunk_used) {
            close_chunk();
            const u32 want = n > PU_CHUNK ? 64u : PU_CHUNK;
            u64 at = 0;
            if (lane == lead) at = atomicAdd((unsigned long long *)&a.ctl[0], (unsigned long long)want);
            at = __shfl(at, lead, 64);
            if (at + want > a.miss_cap) {
                if (lane == lead) atomicOr((unsigned long long *)&a.ctl[1], (unsigned long long)PU_FLAG_OVERFLOW);
                return;
            }
            chunk_at = at;
            chunk_cap = want;
        }
        fill = __shfl(x, lead, 64);
        if (m) a.miss[chunk_at + chunk_used + (u32)__popcll(mask & lt)] = x;
        chunk_used += n;
    };
    // (the one copy of the list code: both records of a step go through it in a loop that is NOT unrolled)
    auto append2 = [&](bool m0, u64 x0, bool m1, u64 x1) {
#pragma nounroll
        for (int h = 0; h < 2; h++) {
            const bool missing = h ? m1 : m0;
            const u64 x = h ? x1 : x0;
            if (__ballot(missing) == 0ull) continue;
            const bool m = missing && claim(x);
            const u64 mask = __ballot(m);
            if (mask == 0ull) continue;
            const int lead = __ffsll((long long)mask) - 1;
            u32 at = 0;
            if (lane == lead) at = atomicAdd(&s_nmiss, (u32)__popcll(mask));
            at = (u32)__shfl((int)at, lead, 64) + (u32)__popcll(mask & lt);
            const bool in_lds = m && at < (u32)PU_LMISS;
            if (in_lds) s_miss[at] = x;
            append_global(m && !in_lds, x);
        }
    };
    // ---- slices -> batches of PU2_U steps ----
    // A step = up to 128 records [lo, hi) of the 128 at `ptr` (two per lane: 2 l and 2 l + 1; lanes whose pair lies beyond
    // hi - 2 re-read the last pair that fits: duplicates of real records, harmless to the table and masked out of the order
    // check).  A slice that does not begin its file starts ONE RECORD EARLY with lo = 1: the pair (f[beg - 1], f[beg]) is
    // then checked inside lane 0 like every other pair -- no separate load for the record in front of the slice -- and a
    // last step of ONE record is moved back by one record the same way, so every load is a 16-byte pair inside the file
    // (files of fewer than two records never come here: the host lists their record itself).  The loads of a batch are
    // issued together and UNCONDITIONALLY (a step behind the slice's end re-reads the batch's first pair): straight-line
    // code, so the compiler's own s_waitcnt counts are exact; the scalar work per step is a handful of instructions.
    auto take = [&]() -> u32 {
        u32 j = 0;
        if (lane == 0) j = atomicAdd(&s_next, 1u);
        return (u32)__builtin_amdgcn_readfirstlane((int)j);
    };
    struct Meta { u64 beg, end, f; };
    auto fetch = [&](u32 j) -> Meta {
        Meta m = {0, 0, 0};
        if (j < S1) {
            m.beg = sload_u64(&a.cuts[(u64)r * S1 + j]);
            m.end = sload_u64(&a.cuts[(u64)(r + 1) * S1 + j]);
            m.f = sload_u64((const u64 *)&a.files[j]);
        }
        return m;
    };
    bool bad = false, raw = false;
    const u32 l2 = 2u * (u32)lane;
    u64 run_carry = 0;
    // FULL: all 128 records of the step are the slice's (lo = 0, hi = 128): no validity masks
    auto consume = [&](auto FULL, const pu_pair &pr, u32 lo, u32 hi) {
        constexpr bool full = decltype(FULL)::value;
        const u64 x0 = pr.x, x1 = pr.y;
        const u32 pmax = hi > 2u ? hi - 2u : 0u;
        const u64 prev = pu2_shr1(x1, run_carry);
        if (full) bad |= prev > x0 || x0 > x1;
        else bad |= (prev > x0 && l2 <= pmax) || x0 > x1;
        run_carry = ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(x1 >> 32), 63) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)x1, 63);
        const u32 h0 = pu_hash(x0), h1 = pu_hash(x1);
        const ulonglong2 p0 = *reinterpret_cast<const ulonglong2 *>(slot(h0, 0)), p1 = *reinterpret_cast<const ulonglong2 *>(slot(h1, 0));
        bool ha = p0.x == x0 || p0.y == x0, hb = p1.x == x1 || p1.y == x1;
        // ONE masked region for the lanes of which either record has to look at slots 2 and 3 (a tenth of the records)
        if ((!ha && p0.y != PU_EMPTY) || (!hb && p1.y != PU_EMPTY)) {
            const ulonglong2 q0 = *reinterpret_cast<const ulonglong2 *>(slot(h0, 2)), q1 = *reinterpret_cast<const ulonglong2 *>(slot(h1, 2));
            ha = ha || q0.x == x0 || q0.y == x0;
            hb = hb || q1.x == x1 || q1.y == x1;
            // a full bucket (0.4 %): the slow way
            if (!ha && q0.y != PU_EMPTY) ha = member_from(x0, (h0 + 1) & (PU_BUCKETS - 1));
            if (!hb && q1.y != PU_EMPTY) hb = member_from(x1, (h1 + 1) & (PU_BUCKETS - 1));
        }
        ha = ha && x0 != PU_EMPTY;  // (an all-ones record would "match" an empty slot)
        hb = hb && x1 != PU_EMPTY;
        bool m0 = !ha, m1 = !hb;
        if (!full) {
            const u32 i0 = l2 < pmax ? l2 : pmax;  // the records this lane holds: i0, i0 + 1
            m0 = m0 && i0 - lo < hi - lo;
            m1 = m1 && i0 + 1u - lo < hi - lo;
        }
        if (__ballot(m0 || m1)) append2(m0, x0, m1, x1);
    };
    u32 g_j = take();          // the next slice; its cut points and pointer are already on their way
    Meta g_m = fetch(g_j);
    u64 ptr = 0;
    u32 rem = 0;
    // one step that is not (known to be) full: the first step of a slice, and what is left behind its full steps
    auto general_step = [&](u32 lo, bool first) {
        const u32 cnt = rem < 128u ? rem : 128u;
        // one record: at the start of its file the pair (0, 1) -- the file has two records --, else the pair (-1, 0)
        const u32 back = (cnt == 1u && !first) ? 1u : 0u;
        const u32 slo = back ? 1u : lo, shi = cnt + back;
        const u32 pmax = shi > 2u ? shi - 2u : 0u;
        // (offsets from ptr - 8, so that the moved-back pair has a non-negative one)
        const u32 voff = 8u - 8u * back + 8u * (l2 < pmax ? l2 : pmax);
        const pu_pair pr = *(const pu_pair __attribute__((address_space(1))) *)((const char __attribute__((address_space(1))) *)(uintptr_t)(ptr - 8) + voff);
        consume(std::false_type{}, pr, slo, shi);
        rem -= cnt;
        ptr += 1024;
    };
    const u32 voff_full = 16u * (u32)lane;
    while (g_j < S1) {
        const Meta m = g_m;
        g_j = take();
        g_m = fetch(g_j);
        const u64 n = m.end > m.beg ? m.end - m.beg : 0ull;
        if (n == 0) continue;
        if (n >= 0xFFFFFF00ull) { raw = true; continue; }  // (a slice of 2^32 records: the caller's other routes)
        const u32 lo = m.beg ? 1u : 0u;                    // 1: the first loaded record lies in front of the slice
        ptr = m.f + 8ull * (m.beg - lo);
        rem = (u32)n + lo;
        run_carry = 0;
        general_step(lo, true);
        while (rem >= 128u * PU2_U) {                      // batches of full steps: their loads are in flight together
            pu_pair pr[PU2_U];
#pragma unroll
            for (int u = 0; u < PU2_U; u++)
                pr[u] = *(const pu_pair __attribute__((address_space(1))) *)((const char __attribute__((address_space(1))) *)(uintptr_t)ptr + (voff_full + 1024u * (u32)u));
#pragma unroll
            for (int u = 0; u < PU2_U; u++) consume(std::true_type{}, pr[u], 0u, 128u);
            rem -= 128u * PU2_U;
            ptr += 1024ull * PU2_U;
        }
        while (rem) general_step(0u, false);
    }
    if (raw && lane == 0) atomicOr((unsigned long long *)&a.ctl[1], (unsigned long long)PU_FLAG_OVERFLOW);
    close_chunk();
    if (bad) atomicOr((unsigned long long *)&a.ctl[1], (unsigned long long)PU_FLAG_UNSORTED);
    __syncthreads();
    const u32 nl = s_nmiss < (u32)PU_LMISS ? s_nmiss : (u32)PU_LMISS;
    if (nl == 0) return;
    if (tid == 0) {
        const u64 at = atomicAdd((unsigned long long *)&a.ctl[0], (unsigned long long)nl);
        if (at + nl > a.miss_cap) atomicOr((unsigned long long *)&a.ctl[1], (unsigned long long)PU_FLAG_OVERFLOW);
        s_flush_at = at;
    }
    __syncthreads();
    const u64 at = s_flush_at;
    if (at + nl <= a.miss_cap)
        for (u32 i = (u32)tid; i < nl; i += PU2_NT) a.miss[at + i] = s_miss[i];
}

// ---- the same pass over records WITH TaxIds (union.go:195-201: the TaxId of a code is the LCA over all its records) ----
// Beside every table slot one 16-byte word of LDS: the TaxId the entry came with (t0: the base files' fold, or the first
// record of a new code), the smallest pre-order number (TaxDev::euler) among its records, the COMPLEMENT of the largest
// (so that widening the interval at either end is the same instruction, an atomic minimum, on one of two words), and a
// flag for the one case the interval cannot show: a record with another TaxId but the same number as everything so far
// (an alias of a merged id; two different unknown ids).  A hit is one more LDS read and — only while it still widens the
// interval — one LDS atomic; the LCA of a set of nodes is the LCA of its members with the smallest and the largest
// number, so ONE table LCA per entry at the end equals the reference's left fold (the contract of lca_dev: 0 / unknown
// ids absorb unless every TaxId is the same).
// New codes are claimed in the table as in the plain pass and leave WITH their fold when the range is done; what
// cannot be claimed (all-ones codes, a table that has doubled) is listed record by record and folded by the final
// sort + unique + 2-way union.  The three words of a slot sit in ONE 16-byte LDS word (a hit reads them with one
// ds_read_b128 beside the two of its bucket): 24 bytes per slot, 1536 buckets of four (144 KB), one workgroup of 1024
// threads per CU (768 buckets and two workgroups of 512 measured the same probe time; the larger range halves the cut
// points and the per-slice steps: 31.0 -> 29.9 ms on config 3's shape at half size).
#ifndef PT_BUCKETS_N
#define PT_BUCKETS_N 1536
#endif
#ifndef PT_NT_N
#define PT_NT_N 1024
#endif
constexpr int PT_NT = PT_NT_N;
constexpr int PT_WAVES = PT_NT == 1024 ? 4 : PU_WAVES;
constexpr int PT_BUCKETS = PT_BUCKETS_N;
constexpr int PT_SLOTS = 4 * PT_BUCKETS;
constexpr int PT_RANGE = PT_BUCKETS;
constexpr int PT_K0 = 4;               // files merged into the base set
// New codes are claimed in the tables with their fold, and a table takes as many of them as it has base entries: the pass
// works as long as the later files bring fewer new codes than the base set holds, i.e. from a hit rate of one half on.
// (1000 files x 1e6 with taxids, a fifth of a universe each: 59 % hits, 18.8 ms against 34 ms through the single-pass
// merge; a tenth each: 34 % hits, the tables fill up and every further record is listed: 179 ms.)
constexpr double PT_MIN_HIT = 0.55;
constexpr u32 PT_UNSET = 0xFFFFFFFFu;  // t0 of a slot: nobody has set it yet

__device__ __forceinline__ u32 pt_hash(u64 x) {
    const u32 lo = (u32)x, hi = (u32)(x >> 32);
    return (u32)(((u64)((lo ^ __builtin_rotateleft32(hi, 15) ^ (hi >> 3)) * 0x9E3779B1u) * (u64)PT_BUCKETS) >> 32);
}

typedef u32 pt_u32x2 __attribute__((ext_vector_type(2)));
typedef pt_u32x2 __attribute__((aligned(4))) pt_tpair;  // 8 bytes at 4-byte alignment

// COUNT = `common` (common.go:220-344) through the same tables: BASE is the first file (every code once: common.go:232,244),
// every slot also counts its records ([31:2] of the flag word), and when the range is done the codes that reached the
// threshold leave — base entries and new codes alike — with their fold; nothing is listed record by record (a record that
// cannot be counted in a table raises PU_FLAG_RAW and the caller's counting merge answers).
template <bool COUNT, bool CM = false>
__global__ __launch_bounds__(PT_NT) __attribute__((amdgpu_waves_per_eu(PT_WAVES, PT_WAVES))) void pt_probe_kernel(PuArgs a) {
    __shared__ __attribute__((aligned(32))) u64 s_tab[PT_SLOTS];
    // x = t0, y = smallest number, z = ~largest, w = [0] another TaxId with the same number was seen, [1] settled, [31:2] records (COUNT)
    __shared__ __attribute__((aligned(16))) uint4 s_st[PT_SLOTS];
    __shared__ u32 s_next, s_nins;
    __shared__ u32 s_scan[PT_NT / 64 + 1];
    __shared__ u64 s_flush_at;
    const int tid = (int)threadIdx.x, lane = lane_id();
    const u32 r = blockIdx.x, S1 = a.S1;
    const TaxDev &T = a.tax;
    for (int i = tid; i < PT_SLOTS; i += PT_NT) {
        s_tab[i] = PU_EMPTY;
        s_st[i] = make_uint4(PT_UNSET, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u);  // (an empty interval)
    }
    if (tid == 0) { s_next = 0; s_nins = 0; }
    __syncthreads();
    auto next_bucket = [](u32 h) -> u32 { return h + 1 == (u32)PT_BUCKETS ? 0u : h + 1; };
    // first free slot of the first bucket of the probe sequence that is not full, or the slot that already holds x
    auto insert = [&](u64 x, bool &fresh) -> int {
        u32 h = pt_hash(x);
        for (;; h = next_bucket(h)) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const u64 old = atomicCAS((unsigned long long *)&s_tab[4 * h + k], (unsigned long long)PU_EMPTY, (unsigned long long)x);
                if (old == PU_EMPTY || old == x) {
                    fresh = old == PU_EMPTY;
                    return (int)(4 * h + k);
                }
            }
        }
    };
    const u64 b0 = (u64)r * a.range;  // (a.range <= PT_RANGE: pt_range_for)
    const u32 nb = (u32)((a.n0 - b0 < (u64)a.range) ? (a.n0 - b0) : (u64)a.range);
    constexpr int PER = (PT_RANGE + PT_NT - 1) / PT_NT;
    u64 ent[PER];
    u32 et[PER];
    bool bad = false, bad_t = false, bad_raw = false;  // an unsorted file; a TaxId of 2^32 - 1 (the table's own "not set"); COUNT: a record no table could count
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const u32 idx = (u32)tid + (u32)i * PT_NT;
        ent[i] = a.base[b0 + (idx < nb ? idx : 0)];
        et[i] = a.base_tax ? a.base_tax[b0 + (idx < nb ? idx : 0)] : a.base_ct;  // (no array: COUNT only -- plain codes, or the first file's one taxid)
        if (idx >= nb) ent[i] = PU_EMPTY;
        else bad_t |= et[i] == PT_UNSET;
    }
    // Clade mode (PuArgs::clade_mode, round 5): a number is `clade code << 24 | pre-order number` -- codes are handed out in
    // pre-order, so the composite orders taxids as the numbers do -- and a record of a file with per-record taxids brings only
    // the code (one byte of a table that stays in L2, instead of 4 bytes of one that does not): see fold.
    constexpr bool cm = CM;  // (an instantiation of its own: the plain fold keeps its instruction count)
    u32 ee[PER];
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const u32 tq = et[i] < T.size ? et[i] : 0u;
        ee[i] = T.euler ? T.euler[tq] : 0u;
        if (cm && ee[i]) ee[i] |= (u32)T.clade8[tq] << 24;
    }
#pragma unroll
    for (int i = 0; i < PER; i++) {
        if (ent[i] == PU_EMPTY) continue;  // (an all-ones code: its records are listed, the final union folds them)
        bool fresh;
        const int slot = insert(ent[i], fresh);
        s_st[slot] = make_uint4(et[i], ee[i], ~ee[i], COUNT ? 4u * a.count0 : 0u);
    }
    __syncthreads();
    auto find_from = [&](u64 x, u32 h) -> int {
        for (;;) {
            const ulonglong2 *b = reinterpret_cast<const ulonglong2 *>(&s_tab[4 * h]);
            const ulonglong2 p = b[0], q = b[1];
            const int k = p.x == x ? 0 : (p.y == x ? 1 : (q.x == x ? 2 : (q.y == x ? 3 : -1)));
            if (k >= 0) return x != PU_EMPTY ? (int)(4 * h) + k : -1;
            if (q.y == PU_EMPTY) return -1;
            h = next_bucket(h);
        }
    };
    auto find2 = [&](u64 xa, u64 xb, int &sa, int &sb) {
        const u32 h0 = pt_hash(xa), h1 = pt_hash(xb);
        const ulonglong2 *b0p = reinterpret_cast<const ulonglong2 *>(&s_tab[4 * h0]);
        const ulonglong2 *b1p = reinterpret_cast<const ulonglong2 *>(&s_tab[4 * h1]);
        const ulonglong2 p0 = b0p[0], q0 = b0p[1], p1 = b1p[0], q1 = b1p[1];
        const int k0 = p0.x == xa ? 0 : (p0.y == xa ? 1 : (q0.x == xa ? 2 : (q0.y == xa ? 3 : -1)));
        const int k1 = p1.x == xb ? 0 : (p1.y == xb ? 1 : (q1.x == xb ? 2 : (q1.y == xb ? 3 : -1)));
        sa = (k0 >= 0 && xa != PU_EMPTY) ? (int)(4 * h0) + k0 : -1;
        sb = (k1 >= 0 && xb != PU_EMPTY) ? (int)(4 * h1) + k1 : -1;
        if (k0 < 0 && q0.y != PU_EMPTY) sa = find_from(xa, next_bucket(h0));
        if (k1 < 0 && q1.y != PU_EMPTY) sb = find_from(xb, next_bucket(h1));
    };
    // one record's TaxId into its entry; e = its pre-order number (0: taxid 0 / unknown)
    // Clade mode: e = code << 24 | number when `exact`, else code << 24 (the number was not read).  An entry whose interval
    // spans two clades has the LCA of those two clade nodes whatever the exact numbers are, so a record needs its number
    // only while the entry's interval lies inside ONE clade and the record is of that clade (or the interval is not written
    // yet): then -- for unrelated taxa next to never -- it is fetched here; a record outside the interval widens it with a
    // sentinel number (all ones below / zero above the code), one inside an interval of several clades does nothing.  The
    // interval only ever widens, so an entry that is still inside one clade at the end had every record folded exactly.
    auto fold = [&](int slot, u32 t, u32 e, bool exact) {
        uint4 st = s_st[slot];
        if (st.x == PT_UNSET) {  // a new code: whoever comes first gives it its TaxId (any order gives the same fold)
            const u32 old = atomicCAS(&s_st[slot].x, PT_UNSET, t);
            if (old == PT_UNSET) {
                if (cm && !exact && e != 0) e |= T.euler[t];  // (the claimer's own number: e != 0 says t is inside the table)
                atomicMin(&s_st[slot].y, e);
                atomicMin(&s_st[slot].z, ~e);
                return;
            }
            st = s_st[slot];
        }
        if (t == st.x) return;
        u32 e_lo = e, e_hi = e;  // what the record puts to the interval's two ends
        if (cm && !exact && e != 0) {
            const u32 c = e >> 24;
            if (st.y > ~st.z || ((st.y >> 24) == c && ((~st.z) >> 24) == c)) {
                e |= T.euler[t];
                e_lo = e_hi = e;
                exact = true;
            } else {
                e_lo = e | 0xFFFFFFu;
            }
        }
        const bool lo = e_lo < st.y, hi = ~e_hi < st.z;
        if (lo | hi) {
            atomicMin(lo ? &s_st[slot].y : &s_st[slot].z, lo ? e_lo : ~e_hi);
            if (lo & hi) atomicMin(&s_st[slot].z, ~e_hi);  // (an interval that is still empty: a new code a moment after its claim)
            // The snapshot was taken between the claimer's CAS on x and its two minima (empty or half-written interval:
            // smallest > largest): this record's number may be the CLAIMER'S -- the alias of a merged id, another unknown
            // id -- and the interval would then never show that two different taxids met.  Say so; a flag too many only
            // sends settle() through LCA(node_at[min], node_at[max]), which is always right.  (Base entries are written
            // in front of the barrier and are never seen half-way.)
            if (st.y > ~st.z && (st.w & 1u) == 0u) atomicOr(&s_st[slot].w, 1u);
        } else if ((!cm || exact || e == 0) && st.y == e && st.z == ~e && (st.w & 1u) == 0u) {
            atomicOr(&s_st[slot].w, 1u);
        }
    };
    const u64 lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    u64 chunk_at = 0, fill = 0;  // wave-uniform
    u32 fill_t = 0;
    u32 chunk_cap = 0, chunk_used = 0;
    auto close_chunk = [&]() {
        if ((u32)lane < chunk_cap - chunk_used) {
            a.miss[chunk_at + chunk_used + (u32)lane] = fill;
            if (a.miss_tax) a.miss_tax[chunk_at + chunk_used + (u32)lane] = fill_t;
        }
        chunk_cap = chunk_used = 0;
    };
    auto append_global = [&](bool m, u64 x, u32 t) {
        const u64 mask = __ballot(m);
        if (mask == 0ull) return;
        const u32 n = (u32)__popcll(mask);
        const int lead = __ffsll((long long)mask) - 1;
        if (n > chunk_cap - chunk_used) {
            close_chunk();
            const u32 want = n > PU_CHUNK ? 64u : PU_CHUNK;
            u64 at = 0;
            if (lane == lead) at = atomicAdd((unsigned long long *)&a.ctl[0], (unsigned long long)want);
            at = __shfl(at, lead, 64);
            if (at + want > a.miss_cap) {
                if (lane == lead) atomicOr((unsigned long long *)&a.ctl[1], (unsigned long long)PU_FLAG_OVERFLOW);
                return;  // (the host discards everything)
            }
            chunk_at = at;
            chunk_cap = want;
        }
        fill = __shfl(x, lead, 64);
        fill_t = (u32)__shfl((int)t, lead, 64);
        if (m) {
            const u64 at = chunk_at + chunk_used + (u32)__popcll(mask & lt);
            a.miss[at] = x;
            if (a.miss_tax) a.miss_tax[at] = t;
        }
        chunk_used += n;
    };
    // a record: found -> fold; not found -> claim a slot for its code (then it is a hit like any other), or list it
    // (plain codes -- no file has TaxIds -- through these tables: a hit has nothing to do beyond the count)
    const bool folds = a.base_tax != nullptr || a.miss_tax != nullptr;
    auto record = [&](bool valid, int slot, u64 x, u32 t, u32 e, bool exact) {
        bool raw = false;
        if (valid) {
            if (slot < 0) {
                if (x == PU_EMPTY || s_nins >= (u32)PT_RANGE) raw = true;
                else {
                    bool fresh;
                    slot = insert(x, fresh);
                    if (fresh) atomicAdd(&s_nins, 1u);
                }
            }
            if (!raw) {
                if (COUNT) atomicAdd(&s_st[slot].w, 4u);
                if (folds) fold(slot, t, e, exact);
            }
        }
        if (COUNT) bad_raw |= raw;
        else append_global(raw, x, t);
    };
    // The lanes stream a slice 128 records per step.  A step is three things that each wait for the one before: the loads
    // of codes and TaxIds (A), the pre-order numbers of those TaxIds (B: a second round trip), the probes (C).  Slices are
    // short here (a range of 1536 entries: several hundred records per file), so a wave that did A, B, C one after the other
    // spent its time waiting twice per step (24 ms on config 3's shape at half size).  The steps of ALL slices of the wave
    // form one sequence instead and run as a pipeline: A of step i + 2 and B of step i + 1 are issued before C of step i.
#ifndef PT_U
#define PT_U 1     /* 16-byte loads per lane and step (2: 31.0 ms on config 3's shape at half size with one taxid per file, 1: 29.5; without the pipeline 2: 33.6, 4: 31.9; one stage deeper 1: 29.9, 2: 34.2) */
#endif
    constexpr int U = PT_U;
    struct Desc { u64 f, tf, p0, end, len; u32 ct, ce; bool valid; };  // wave-uniform (ct, ce: the file's own taxid and its number when tf == 0)
    struct RegA { pu_pair pr[U]; pt_tpair tp[U]; u64 nx[U]; };
    struct RegB { u32 eu[U][2]; };
    auto issue_a = [&](const Desc &d, RegA &ra) {
        if (!d.valid) return;
        const auto f = as_global((const u64 *)(uintptr_t)d.f);
        const bool has_t = d.tf != 0;
        const auto tf = as_global((const u32 *)(uintptr_t)(has_t ? d.tf : d.f));
#pragma unroll
        for (int u = 0; u < U; u++) {
            const u64 pos = d.p0 + (u64)u * 128 + 2u * (u32)lane;
            const u64 q = pos < d.len - 2 ? pos : d.len - 2;
            const u64 q2 = pos + 2 < d.len ? pos + 2 : d.len - 1;
            ra.pr[u] = *(const pu_pair __attribute__((address_space(1))) *)(f + q);
            ra.nx[u] = f[q2];
            ra.tp[u] = pt_tpair{d.ct, d.ct};
            if (has_t) ra.tp[u] = *(const pt_tpair __attribute__((address_space(1))) *)(tf + q);  // (wave-uniform branch)
        }
    };
    auto issue_b = [&](const Desc &d, const RegA &ra, RegB &rb) {
        if (!d.valid) return;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const u32 ta = ra.tp[u].x, tb = ra.tp[u].y;
            bad_t |= ta == PT_UNSET || tb == PT_UNSET;
            rb.eu[u][0] = rb.eu[u][1] = d.ce;
            if (d.tf != 0) {  // (wave-uniform; a file without per-record TaxIds: its own one's number -- 0 without any, and there may be no taxonomy at all)
                if (cm) {     // the clade codes alone (fold fetches a number where it matters)
                    rb.eu[u][0] = (u32)T.clade8[ta < T.size ? ta : 0u] << 24;
                    rb.eu[u][1] = (u32)T.clade8[tb < T.size ? tb : 0u] << 24;
                } else {
                    rb.eu[u][0] = T.euler[ta < T.size ? ta : 0u];
                    rb.eu[u][1] = T.euler[tb < T.size ? tb : 0u];
                }
            }
        }
    };
    auto process = [&](const Desc &d, const RegA &ra, const RegB &rb) {
        const u64 p0 = d.p0, end = d.end, len = d.len;
        if (p0 + (u64)U * 128 + 2 <= len) {
            // every lane read two records of the file and the record behind them (wave-uniform test): the order check
            // needs no validity logic (what lies behind the slice's end is still the file), the probes only `pos < end`
#pragma unroll
            for (int u = 0; u < U; u++) {
                const u64 pos = p0 + (u64)u * 128 + 2u * (u32)lane;
                const u64 x0 = ra.pr[u].x, x1 = ra.pr[u].y;
                bad |= x0 > x1 || x1 > ra.nx[u];
                int s0, s1;
                find2(x0, x1, s0, s1);
                record(pos < end, s0, x0, ra.tp[u].x, rb.eu[u][0], d.tf == 0);
                record(pos + 1 < end, s1, x1, ra.tp[u].y, rb.eu[u][1], d.tf == 0);  // (a code claimed a moment ago is found again by the insert)
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const u64 pos = p0 + (u64)u * 128 + 2u * (u32)lane;
            const u32 nv = pos + 1 < end ? 2u : (pos < end ? 1u : 0u);
            const bool shifted = pos > len - 2;  // pos = len - 1 (or beyond: nv = 0): the record is the pair's second
            const u64 x0 = shifted ? ra.pr[u].y : ra.pr[u].x;
            const u32 y0 = shifted ? ra.tp[u].y : ra.tp[u].x, e0 = shifted ? rb.eu[u][1] : rb.eu[u][0];
            const u64 x1 = nv == 2 ? ra.pr[u].y : x0;
            // the record behind the last valid one (order check, also across slices); all ones behind the file
            const u64 x2 = nv == 2 ? (pos + 2 < len ? ra.nx[u] : PU_EMPTY) : ((!shifted && pos + 1 < len) ? ra.pr[u].y : PU_EMPTY);
            const bool v0 = nv >= 1, v1 = nv == 2;
            if (v0) bad |= x0 > x1 || x1 > x2;
            int s0, s1;
            find2(x0, x1, s0, s1);
            record(v0, s0, x0, y0, e0, d.tf == 0);
            record(v1, s1, x1, ra.tp[u].y, rb.eu[u][1], d.tf == 0);
        }
    };
    auto take = [&]() -> u32 {
        u32 j = 0;
        if (lane == 0) j = atomicAdd(&s_next, 1u);
        return (u32)__builtin_amdgcn_readfirstlane((int)j);
    };
    struct Meta { u64 beg, end, len, f, tf, cte; };
    auto fetch = [&](u32 j) -> Meta {
        Meta m = {0, 0, 0, 0, 0, 0};
        if (j < S1) {
            m.beg = sload_u64(&a.cuts[(u64)r * S1 + j]);
            m.end = sload_u64(&a.cuts[(u64)(r + 1) * S1 + j]);
            m.len = sload_u64(&a.lens[j]);
            m.f = sload_u64((const u64 *)&a.files[j]);
            m.tf = sload_u64((const u64 *)&a.tfiles[j]);
            if (a.cte) m.cte = sload_u64(&a.cte[j]);
        }
        return m;
    };
    // the wave's sequence of steps: slices are taken from the workgroup's counter, the cut points of the slice after
    // the current one are already on their way
    u32 j = take();
    Meta cur = fetch(j);
    u32 jn = take();
    Meta nxt = fetch(jn);
    u64 pos = cur.beg;
    auto next_desc = [&]() -> Desc {
        for (;;) {
            if (j >= S1) return Desc{0, 0, 0, 0, 0, 0u, 0u, false};
            const u64 end = cur.end < cur.beg ? cur.beg : cur.end;
            const u32 fct = cur.tf ? 0u : (u32)cur.cte, fce = cur.tf ? 0u : (u32)(cur.cte >> 32);
            if (cur.len >= 2 && pos < end) {
                const Desc d = {cur.f, cur.tf, pos, end, cur.len, fct, fce, true};
                pos += (u64)U * 128;
                return d;
            }
            if (cur.len < 2 && end > cur.beg) {  // a one-record file (no 16-byte load fits): done on the spot
                const auto f = as_global((const u64 *)(uintptr_t)cur.f);
                const u64 x = f[0];
                const u32 t = cur.tf ? as_global((const u32 *)(uintptr_t)cur.tf)[0] : fct;
                bad_t |= t == PT_UNSET;
                u32 e = cur.tf ? T.euler[t < T.size ? t : 0u] : fce;
                if (cm && cur.tf && e) e |= (u32)T.clade8[t] << 24;
                record(lane == 0, lane == 0 ? find_from(x, pt_hash(x)) : -1, x, t, e, true);
            }
            j = jn;
            cur = nxt;
            jn = take();
            nxt = fetch(jn);
            pos = cur.beg;
        }
    };
    {
        Desc d0 = next_desc(), d1 = next_desc();
        RegA a0, a1, a2;
        RegB b0, b1;
        issue_a(d0, a0);
        issue_a(d1, a1);
        issue_b(d0, a0, b0);
        while (d0.valid) {
            const Desc d2 = next_desc();
            issue_a(d2, a2);
            issue_b(d1, a1, b1);
            process(d0, a0, b0);
            d0 = d1; a0 = a1; b0 = b1;
            d1 = d2; a1 = a2;
        }
    }
    close_chunk();
    if (bad) atomicOr((unsigned long long *)&a.ctl[1], (unsigned long long)PU_FLAG_UNSORTED);
    if (bad_t) atomicOr((unsigned long long *)&a.ctl[1], (unsigned long long)PU_FLAG_TAXID);
    if (bad_raw) atomicOr((unsigned long long *)&a.ctl[1], (unsigned long long)PU_FLAG_RAW);
    __syncthreads();
    // ---- the folds: one table LCA per entry that met a different TaxId -------------------------------------------------
    auto settle = [&](int slot) -> u32 {
        const uint4 st = s_st[slot];
        const u32 mn = st.y, mx = ~st.z;
        if (mn == mx && (st.w & 1u) == 0u) return st.x;  // every record carried t0
        if (mn == 0u) return 0u;                  // TaxId 0 / an unknown id among records that differ
        if (cm) {
            if ((mn >> 24) != (mx >> 24)) return lca_clade_pair(T, mn >> 24, mx >> 24);
            return lca_dev(T, T.node_at[mn & 0xFFFFFFu], T.node_at[mx & 0xFFFFFFu]);
        }
        return lca_dev(T, T.node_at[mn], T.node_at[mx]);
    };
    if (!COUNT) {
#pragma unroll
        for (int i = 0; i < PER; i++) {
            if (ent[i] == PU_EMPTY) continue;
            const int slot = find_from(ent[i], pt_hash(ent[i]));
            const u32 res = settle(slot);
            if (a.base_tax && res != et[i]) a.base_tax[b0 + (u32)tid + (u32)i * PT_NT] = res;
            s_st[slot].w = 2u;  // (this entry is done)
        }
        __syncthreads();
    }
    // what is left in the table are the new codes of this range (COUNT: every code that reached the threshold)
    auto leaves = [&](int sl) -> bool {
        if (sl >= PT_SLOTS || s_tab[sl] == PU_EMPTY) return false;
        const u32 w = s_st[sl].w;
        return COUNT ? (w >> 2) >= a.threshold : w != 2u;
    };
    constexpr int SPT = (PT_SLOTS + PT_NT - 1) / PT_NT;
    u32 mine = 0;
#pragma unroll
    for (int i = 0; i < SPT; i++) {
        const int sl = tid * SPT + i;
        if (leaves(sl)) mine++;
    }
    u32 tot;
    u32 at_l = block_excl_scan_u32<PT_NT>(mine, s_scan, &tot);
    if (tot == 0) return;
    if (tid == 0) {
        const u64 at = atomicAdd((unsigned long long *)&a.ctl[0], (unsigned long long)tot);
        if (at + tot > a.miss_cap) atomicOr((unsigned long long *)&a.ctl[1], (unsigned long long)PU_FLAG_OVERFLOW);
        s_flush_at = at;
    }
    __syncthreads();
    const u64 at = s_flush_at;
    if (at + tot > a.miss_cap) return;
#pragma unroll
    for (int i = 0; i < SPT; i++) {
        const int sl = tid * SPT + i;
        if (leaves(sl)) {
            a.miss[at + at_l] = s_tab[sl];
            if (a.miss_tax) a.miss_tax[at + at_l] = settle(sl);
            at_l++;
        }
    }
}

// ---- every file carries ONE taxid (round 5: the .unik header's global taxid, `count -t`) ---------------------------------
// union.go:195-201 then folds, for every code, the taxids of the FILES that hold it.  The distinct taxid values of a call
// are few (at most one per file), so the host RANKS them: sorted by (pre-order number, taxid value), rank 1 .. D.  The LCA
// of a set of nodes is the LCA of its members with the smallest and the largest pre-order number, and ranks order the
// values by that number -- so all an entry has to keep is the smallest and the largest RANK among the files that hold its
// code: one 32-bit word per slot ([15:0] smallest rank, [31:16] the complement of the largest; 0xFFFFFFFF = no file yet).
// A record costs its hash probe (the plain kernel's: one 32-byte bucket read) and ONE more 4-byte LDS read; the rank of
// its file is a wave-uniform scalar, and the word only changes while the file's rank lies outside the entry's interval --
// the host hands the files over in the order lowest rank, highest, second lowest, second highest ..., so after an entry's
// first two or three files hardly any does (a CAS loop then; no atomic otherwise).  No taxid is loaded, no pre-order
// number looked up, no LCA evaluated while the files stream.
// BASE is the PLAIN union of the largest files (no LCA in its k-way union) and EVERY file is probed, the base files
// included: their ranks are folded like anybody's.  The entries' words live in base_st[] between launches (more files than
// one launch takes: the next batch starts from them); pr_settle_kernel turns them into taxids at the end -- one rank: that
// taxid itself (LCA(x, x) = x, also for an unknown id); else 0 when the smallest number is 0 (taxid 0 / unknown ids among
// files that differ); else LCA(node_at[smallest number], node_at[largest]) -- the left fold of lca_dev, as in ukm_pfold.hip.
// New codes are claimed in the table as in the other kernels and leave with their settled taxid when the range is done;
// what cannot be claimed (all-ones codes, a table that has doubled) is listed record by record with the file's taxid and
// the final sort + unique + 2-way union folds it (associativity is all that is used).
constexpr u32 PR_NONE = 0xFFFFFFFFu; // no file yet
constexpr u32 PR_MAX_RANK = 0xFFFEu;

struct PrTables {
    const u32 *tax_of_rank;  // [D + 1]
    const u32 *eul_of_rank;  // [D + 1]: the pre-order number of the rank's taxid (non-decreasing in the rank)
    u32 *base_st;            // [n0] in / out: the rank interval of every base entry
    // the settled taxid of every (smallest rank, largest rank) pair, [D + 1][D + 1], when the call has at most PR_PAIR_MAX
    // distinct taxids (pr_pairs_kernel): 2e8 entries then settle with one cached read instead of an LCA each
    u32 *pair;
    u32 D;
};
constexpr u32 PR_PAIR_MAX = 1024;

__device__ __forceinline__ u32 pr_settle_pair(const PrTables &t, const TaxDev &T, u32 mn, u32 mx) {
    if (mn == mx) return t.tax_of_rank[mn];
    const u32 en = t.eul_of_rank[mn], ex = t.eul_of_rank[mx];
    if (en == 0u) return 0u;
    return lca_dev(T, T.node_at[en], T.node_at[ex]);
}

__device__ __forceinline__ u32 pr_settle(const PrTables &t, const TaxDev &T, u32 w) {
    const u32 mn = w & 0xFFFFu, mx = 0xFFFFu - (w >> 16);
    if (w == PR_NONE || mn > mx || mx > t.D) return 0u;  // (no file held the code: cannot happen for an entry that is in the table)
    if (t.pair) return t.pair[(size_t)mn * (t.D + 1) + mx];
    return pr_settle_pair(t, T, mn, mx);
}

__global__ void pr_pairs_kernel(PrTables t, TaxDev T) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 W = t.D + 1;
    if (i >= W * W) return;
    const u32 mn = i / W, mx = i % W;
    t.pair[i] = (mn >= 1 && mn <= mx) ? pr_settle_pair(t, T, mn, mx) : 0u;
}

__global__ void pr_settle_kernel(PrTables t, TaxDev T, u64 n0, u32 *base_tax) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n0) base_tax[i] = pr_settle(t, T, t.base_st[i]);
}

// Table layout: round 5 kept 4-byte TAGS per slot beside the codes (a record read 16 bytes of tags, then the code and rank
// word of the slot whose tag matched); round 6 reads the codes themselves, P pair first (see the kernel): 12 bytes per slot.
#ifndef PR_TNT_N
#define PR_TNT_N 1024
#endif
#ifndef PR_TBUCKETS_N
#define PR_TBUCKETS_N 1536
#endif
#ifndef PR_TWAVES
#define PR_TWAVES 8
#endif
// (measured on config 3's shape at half size, probe pass: 2304 buckets x 1024 threads, one workgroup per CU = 4 waves per SIMD
//  14.3 ms; 1152 x 512 x 2 workgroups 14.9; 768 x 512 x 3 = 6 waves 12.6; 1152 x 1024 x 2 = 8 waves per SIMD 12.55)
constexpr int PR_TNT = PR_TNT_N;               // threads of a workgroup
constexpr int PR_TBUCKETS = PR_TBUCKETS_N;     // x 4 slots x (8 + 4) bytes = 72 KB of LDS: two workgroups of 16 waves per CU
constexpr int PR_TSLOTS = 4 * PR_TBUCKETS;
// ONE multiplicative hash per code (v_mul_lo_u32 runs at a quarter of the VALU rate: the two products + the mul_hi of
// the first version were a fifth of the kernel's vector work): its top bits pick the bucket, the word itself (odd: never
// 0) is the tag -- inside a bucket the tags still differ in 21 bits, and a false positive only costs the exact look-up.
__device__ __forceinline__ u32 prt_hash(u64 x) {
    const u32 lo = (u32)x, hi = (u32)(x >> 32);
    return (lo ^ __builtin_rotateleft32(hi, 15) ^ (hi >> 3)) * 0x9E3779B1u;
}
__device__ __forceinline__ u32 prt_bucket_of(u32 h) {
    if ((PR_TBUCKETS & (PR_TBUCKETS - 1)) == 0) return h >> (32 - __builtin_ctz((unsigned)PR_TBUCKETS));
    return (u32)(((u64)h * (u64)PR_TBUCKETS) >> 32);
}
__device__ __forceinline__ u32 prt_bucket(u64 x) { return prt_bucket_of(prt_hash(x)); }

__global__ __launch_bounds__(PR_TNT) __attribute__((amdgpu_waves_per_eu(PR_TWAVES, PR_TWAVES))) void pr_probe_kernel(PuArgs a, PrTables t) {
    // (round 6: no tag words any more -- the codes in two halves as in pu2_probe_kernel: slots 0 and 1 of every bucket in the
    //  first half (P), slots 2 and 3 behind them (Q); a record reads its bucket's P pair, and only a lane that does not find
    //  it there and sees slot 1 taken reads the Q pair; then the rank word of the slot that matched.  Four tag compares, a
    //  select chain and the dependent code read per record are gone, and the table shrinks from 16 to 12 bytes per slot.)
    __shared__ __attribute__((aligned(16))) u64 s_key[PR_TSLOTS];
    __shared__ u32 s_st[PR_TSLOTS];
    auto sidx = [](u32 h, int q) -> int { return (q >> 1) * (PR_TSLOTS / 2) + (int)(2 * h) + (q & 1); };
    __shared__ u32 s_next, s_nins;
    __shared__ u32 s_scan[PR_TNT / 64 + 1];
    __shared__ u64 s_flush_at;
    const int tid = (int)threadIdx.x, lane = lane_id();
    const u32 r = blockIdx.x, S1 = a.S1;
    for (int i = tid; i < PR_TSLOTS; i += PR_TNT) {
        s_key[i] = PU_EMPTY;
        s_st[i] = PR_NONE;
    }
    if (tid == 0) { s_next = 0; s_nins = 0; }
    __syncthreads();
    auto next_bucket = [](u32 h) -> u32 { return h + 1 == (u32)PR_TBUCKETS ? 0u : h + 1; };
    // first free slot of the first bucket of the probe sequence that is not full, or the slot that already holds x
    auto insert = [&](u64 x, bool &fresh) -> int {
        u32 h = prt_bucket(x);
        for (;; h = next_bucket(h)) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const u64 old = atomicCAS((unsigned long long *)&s_key[sidx(h, k)], (unsigned long long)PU_EMPTY, (unsigned long long)x);
                if (old == PU_EMPTY || old == x) {
                    fresh = old == PU_EMPTY;
                    return sidx(h, k);
                }
            }
        }
    };
    // (the slow, exact way: walk the codes of x's probe sequence)
    auto find_codes = [&](u64 x) -> int {
        if (x == PU_EMPTY) return -1;
        u32 h = prt_bucket(x);
        for (;; h = next_bucket(h)) {
            const ulonglong2 p = *reinterpret_cast<const ulonglong2 *>(&s_key[sidx(h, 0)]), q = *reinterpret_cast<const ulonglong2 *>(&s_key[sidx(h, 2)]);
            const int k = p.x == x ? 0 : (p.y == x ? 1 : (q.x == x ? 2 : (q.y == x ? 3 : -1)));
            if (k >= 0) return sidx(h, k);
            if (q.y == PU_EMPTY) return -1;
        }
    };
    const u64 b0 = (u64)r * a.range;
    const u32 nb = (u32)((a.n0 - b0 < (u64)a.range) ? (a.n0 - b0) : (u64)a.range);
    constexpr int PER = (PR_TBUCKETS + PR_TNT - 1) / PR_TNT;
    u64 ent[PER];
    u32 est[PER];
    int eslot[PER];
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const u32 idx = (u32)tid + (u32)i * PR_TNT;
        ent[i] = a.base[b0 + (idx < nb ? idx : 0)];
        est[i] = t.base_st[b0 + (idx < nb ? idx : 0)];
        if (idx >= nb) ent[i] = PU_EMPTY;
    }
#pragma unroll
    for (int i = 0; i < PER; i++) {
        eslot[i] = -1;
        if (ent[i] == PU_EMPTY) continue;  // (an all-ones code: its records are listed, the final union folds them)
        bool fresh;
        eslot[i] = insert(ent[i], fresh);
        s_st[eslot[i]] = est[i];
    }
    __syncthreads();
    const u64 lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    u64 chunk_at = 0, fill = 0;  // wave-uniform
    u32 fill_t = 0;
    u32 chunk_cap = 0, chunk_used = 0;
    auto close_chunk = [&]() {
        if ((u32)lane < chunk_cap - chunk_used) {
            a.miss[chunk_at + chunk_used + (u32)lane] = fill;
            a.miss_tax[chunk_at + chunk_used + (u32)lane] = fill_t;
        }
        chunk_cap = chunk_used = 0;
    };
    auto append_global = [&](bool m, u64 x, u32 tx) {
        const u64 mask = __ballot(m);
        if (mask == 0ull) return;
        const u32 n = (u32)__popcll(mask);
        const int lead = __ffsll((long long)mask) - 1;
        if (n > chunk_cap - chunk_used) {
            close_chunk();
            const u32 want = n > PU_CHUNK ? 64u : PU_CHUNK;
            u64 at = 0;
            if (lane == lead) at = atomicAdd((unsigned long long *)&a.ctl[0], (unsigned long long)want);
            at = __shfl(at, lead, 64);
            if (at + want > a.miss_cap) {
                if (lane == lead) atomicOr((unsigned long long *)&a.ctl[1], (unsigned long long)PU_FLAG_OVERFLOW);
                return;  // (the host discards everything)
            }
            chunk_at = at;
            chunk_cap = want;
        }
        fill = __shfl(x, lead, 64);
        fill_t = tx;  // (wave-uniform: the file's taxid)
        if (m) {
            const u64 at = chunk_at + chunk_used + (u32)__popcll(mask & lt);
            a.miss[at] = x;
            a.miss_tax[at] = tx;
        }
        chunk_used += n;
    };
    // widen the interval of `slot` by a file of rank rk (crk = its complement)
    auto widen = [&](int slot, u32 rk, u32 crk) {
        u32 *w = &s_st[slot];
        u32 old = *w;
        for (;;) {
            const u32 mn = old & 0xFFFFu, cmx = old >> 16;
            const u32 nw = (mn < rk ? mn : rk) | ((cmx < crk ? cmx : crk) << 16);
            if (nw == old) break;
            const u32 prev = atomicCAS(w, old, nw);
            if (prev == old) break;
            old = prev;
        }
    };
    // the rare part of a record: the tags did not settle it (slot < 0: look the codes up, claim a slot or list the record), or
    // its file's rank lies outside the entry's interval
    auto rare = [&](bool valid, int slot, u64 x, u32 rk, u32 crk, u32 ftax) {
        bool raw = false;
        if (valid) {
            if (slot < 0) slot = find_codes(x);
            if (slot < 0) {
                if (x == PU_EMPTY || s_nins >= (u32)PR_TBUCKETS) raw = true;
                else {
                    bool fresh;
                    slot = insert(x, fresh);
                    if (fresh) atomicAdd(&s_nins, 1u);
                }
            }
            if (!raw) widen(slot, rk, crk);
        }
        append_global(raw, x, ftax);
    };
    bool bad = false, raw_slice = false;
    // N records of one file of rank rk, from registers: the P pairs of ALL of them are read; one masked region reads the Q
    // pairs of the lanes that need them; then the rank words of the slots that matched; then every record is judged.
    auto judge = [&](auto NN, const u64 *x, const bool *v, u32 rk, u32 crk, u32 ftax) {
        constexpr int N = decltype(NN)::value;
        ulonglong2 pp[N];
        u32 hh[N];
#pragma unroll
        for (int i = 0; i < N; i++) {
            hh[i] = prt_bucket(x[i]);
            pp[i] = *reinterpret_cast<const ulonglong2 *>(&s_key[sidx(hh[i], 0)]);
        }
        int sl[N];
        bool needq = false;
#pragma unroll
        for (int i = 0; i < N; i++) {
            sl[i] = pp[i].x == x[i] ? sidx(hh[i], 0) : (pp[i].y == x[i] ? sidx(hh[i], 1) : -1);
            needq = needq || (sl[i] < 0 && pp[i].y != PU_EMPTY);
        }
        if (needq) {
#pragma unroll
            for (int i = 0; i < N; i++) {
                const ulonglong2 q = *reinterpret_cast<const ulonglong2 *>(&s_key[sidx(hh[i], 2)]);
                if (sl[i] < 0) sl[i] = q.x == x[i] ? sidx(hh[i], 2) : (q.y == x[i] ? sidx(hh[i], 3) : -1);
            }
        }
        u32 ww[N];
#pragma unroll
        for (int i = 0; i < N; i++) ww[i] = s_st[sl[i] < 0 ? 0 : sl[i]];
        bool more[N];
        bool any = false;
#pragma unroll
        for (int i = 0; i < N; i++) {
            const bool hit = sl[i] >= 0 && x[i] != PU_EMPTY;
            const bool outside = rk < (ww[i] & 0xFFFFu) || crk < (ww[i] >> 16);
            if (!hit) sl[i] = -1;  // (not in the bucket's four slots, or a full bucket: the codes of the probe sequence decide)
            more[i] = v[i] && (!hit || outside);
            any = any || more[i];
        }
        if (__ballot(any) == 0ull) return;
#pragma unroll
        for (int i = 0; i < N; i++) {
            if (__ballot(more[i]) == 0ull) continue;
            rare(more[i], sl[i], x[i], rk, crk, ftax);
        }
    };
    auto take = [&]() -> u32 {
        u32 j = 0;
        if (lane == 0) j = atomicAdd(&s_next, 1u);
        return (u32)__builtin_amdgcn_readfirstlane((int)j);
    };
    struct Meta { u64 beg, end, len, f, cte; };
    auto fetch = [&](u32 j) -> Meta {
        Meta m = {0, 0, 0, 0, 0};
        if (j < S1) {
            m.beg = sload_u64(&a.cuts[(u64)r * S1 + j]);
            m.end = sload_u64(&a.cuts[(u64)(r + 1) * S1 + j]);
            m.len = sload_u64(&a.lens[j]);
            m.f = sload_u64((const u64 *)&a.files[j]);
            m.cte = sload_u64(&a.cte[j]);  // [31:0] the file's taxid, [63:32] its rank
        }
        return m;
    };
    // The streaming skeleton of pu2_probe_kernel (round 6): a slice = one general first step (it starts one record early when
    // the slice does not begin its file: the boundary pair is checked inside lane 0), batches of full steps without
    // validity masks, general steps for what is left; the order check takes a pair's predecessor from the neighbouring
    // lane (DPP wave_shr:1) instead of a third load per lane.
    const u32 l2 = 2u * (u32)lane;
    u64 run_carry = 0, ptr = 0;
    u32 rem = 0, rk = 0, crk = 0, ftax = 0;
    auto order = [&](u64 x0, u64 x1, bool cross) {
        const u64 prev = pu2_shr1(x1, run_carry);
        bad |= (cross && prev > x0) || x0 > x1;
        run_carry = ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(x1 >> 32), 63) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)x1, 63);
    };
    auto general_step = [&](u32 lo, bool first) {
        const u32 cnt = rem < 128u ? rem : 128u;
        // one record: at the start of its file the pair (0, 1) -- the file has two records --, else the pair (-1, 0)
        const u32 back = (cnt == 1u && !first) ? 1u : 0u;
        const u32 slo = back ? 1u : lo, shi = cnt + back;
        const u32 pmax = shi > 2u ? shi - 2u : 0u;
        const u32 i0 = l2 < pmax ? l2 : pmax;  // the records this lane holds: i0, i0 + 1 (lanes beyond re-read the last pair)
        const pu_pair pr = *(const pu_pair __attribute__((address_space(1))) *)((const char __attribute__((address_space(1))) *)(uintptr_t)(ptr - 8) +
                                                                               (8u - 8u * back + 8u * i0));
        order(pr.x, pr.y, l2 <= pmax);
        const u64 x[2] = {pr.x, pr.y};
        const bool v[2] = {i0 - slo < shi - slo, i0 + 1u - slo < shi - slo};
        judge(std::integral_constant<int, 2>{}, x, v, rk, crk, ftax);
        rem -= cnt;
        ptr += 1024;
    };
    const u32 voff_full = 16u * (u32)lane;
#ifndef PR_U_N
#define PR_U_N 1
#endif
    constexpr int PRU = PR_U_N;
    u32 j = take();
    Meta cur = fetch(j);
    while (j < S1) {
        const u32 jn = take();
        const Meta nxt = fetch(jn);
        const u64 end = cur.end < cur.beg ? cur.beg : cur.end, n = end - cur.beg;
        ftax = (u32)cur.cte;
        rk = (u32)(cur.cte >> 32);
        crk = 0xFFFFu - rk;
        if (n >= 0xFFFFFF00ull) raw_slice = true;  // (a slice of 2^32 records: the caller's other routes)
        else if (cur.len < 2) {  // (a one-record file: no 16-byte load fits)
            if (n) {
                const u64 x = as_global((const u64 *)(uintptr_t)cur.f)[0];
                rare(lane == 0, -1, x, rk, crk, ftax);
            }
        } else if (n) {
            const u32 lo = cur.beg ? 1u : 0u;
            ptr = cur.f + 8ull * (cur.beg - lo);
            rem = (u32)n + lo;
            run_carry = 0;
            general_step(lo, true);
            while (rem >= 128u * PRU) {
                pu_pair pr[PRU];
#pragma unroll
                for (int u = 0; u < PRU; u++)
                    pr[u] = *(const pu_pair __attribute__((address_space(1))) *)((const char __attribute__((address_space(1))) *)(uintptr_t)ptr + (voff_full + 1024u * (u32)u));
                u64 x[2 * PRU];
                bool v[2 * PRU];
#pragma unroll
                for (int u = 0; u < PRU; u++) {
                    order(pr[u].x, pr[u].y, true);
                    x[2 * u] = pr[u].x;
                    x[2 * u + 1] = pr[u].y;
                    v[2 * u] = v[2 * u + 1] = true;
                }
                judge(std::integral_constant<int, 2 * PRU>{}, x, v, rk, crk, ftax);
                rem -= 128u * PRU;
                ptr += 1024ull * PRU;
            }
            while (rem) general_step(0u, false);
        }
        j = jn;
        cur = nxt;
    }
    if (raw_slice && lane == 0) atomicOr((unsigned long long *)&a.ctl[1], (unsigned long long)PU_FLAG_OVERFLOW);
    close_chunk();
    if (bad) atomicOr((unsigned long long *)&a.ctl[1], (unsigned long long)PU_FLAG_UNSORTED);
    __syncthreads();
    // the base entries hand their intervals back (the next batch of files, or pr_settle_kernel, goes on from them) ...
#pragma unroll
    for (int i = 0; i < PER; i++) {
        if (eslot[i] < 0) continue;
        const u32 w = s_st[eslot[i]];
        if (w != est[i]) t.base_st[b0 + (u32)tid + (u32)i * PR_TNT] = w;
        s_key[eslot[i]] = PU_EMPTY;  // (... and leave the table: what is left are the new codes of this range)
    }
    __syncthreads();
    constexpr int SPT = (PR_TSLOTS + PR_TNT - 1) / PR_TNT;
    u32 mine = 0;
#pragma unroll
    for (int i = 0; i < SPT; i++) {
        const int q = tid * SPT + i;
        if (q < PR_TSLOTS && s_key[q] != PU_EMPTY) mine++;
    }
    u32 tot;
    u32 at_l = block_excl_scan_u32<PR_TNT>(mine, s_scan, &tot);
    if (tot == 0) return;
    if (tid == 0) {
        const u64 at = atomicAdd((unsigned long long *)&a.ctl[0], (unsigned long long)tot);
        if (at + tot > a.miss_cap) atomicOr((unsigned long long *)&a.ctl[1], (unsigned long long)PU_FLAG_OVERFLOW);
        s_flush_at = at;
    }
    __syncthreads();
    const u64 at = s_flush_at;
    if (at + tot > a.miss_cap) return;
#pragma unroll
    for (int i = 0; i < SPT; i++) {
        const int q = tid * SPT + i;
        if (q < PR_TSLOTS && s_key[q] != PU_EMPTY) {
            a.miss[at + at_l] = s_key[q];
            a.miss_tax[at + at_l] = pr_settle(t, a.tax, s_st[q]);
            at_l++;
        }
    }
}

// Do the files share codes at all?  Records drawn from random files are looked up in ONE other random file each: the
// share that is found estimates how much of a collection a file holds.  (The chunk files of an out-of-core sort share
// nothing: without this look the placement merge below would build a base set and sample it before it declines.)
__global__ void pu_overlap_kernel(PuArgs a, u32 nsamp) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    bool tested = false, hit = false;
    if (i < nsamp && a.S1 >= 2) {
        const u64 h = pu_splitmix(i);
        const u32 fa = (u32)(h % a.S1), fb = (fa + 1 + (u32)((h >> 20) % (a.S1 - 1))) % a.S1;
        const u64 la = a.lens[fa], lb = a.lens[fb];
        if (la && lb) {
            const u64 key = as_global(a.files[fa])[pu_splitmix(h) % la];
            const auto f = as_global(a.files[fb]);
            u64 lo = 0, hi = lb;
            while (lo < hi) {
                const u64 mid = (lo + hi) >> 1;
                if (f[mid] < key) lo = mid + 1; else hi = mid;
            }
            tested = true;
            hit = lo < lb && f[lo] == key;
        }
    }
    const u64 mh = __ballot(hit), mt = __ballot(tested);
    if (lane_id() == 0 && mt) {
        atomicAdd((unsigned long long *)&a.ctl[2], (unsigned long long)__popcll(mh));
        atomicAdd((unsigned long long *)&a.ctl[3], (unsigned long long)__popcll(mt));
    }
}

// ---- keep-everything merge of MANY files that share most of their codes, by placement (`merge` / mergeChunksFile's heap,
// util-sort.go:196-225,289-351, when a code is in hundreds of the files) --------------------------------------------------
// The merged sequence is, code by code, the records of that code in file order.  With the sorted distinct codes (BASE, the
// plain probe union above) cut into ranges of PL_RANGE, one workgroup per range
//   A. streams its slice of every file once: the file's order is checked, every record's code is looked up (one hash
//      probe), counted, and the code's INDEX in the range (2 bytes) is kept in a scratch array; a scan of the counts says
//      where every code's run begins -- the range's own beginning is the sum of its cut points;
//   B. writes every code's run of CODES in one piece (a wave per code): plain codes are done here;
//   C. goes through the files in ORDER, PL_BATCH at a time: index and TaxId of the batch's records set a bit and a TaxId
//      cell [code][file of the batch] in LDS; then every code's TaxIds of the batch -- neighbours in the result -- are
//      written in one piece behind what the earlier batches wrote.
// Codes are read once, TaxIds once, 2 bytes per record go to scratch and back; no sorting, no merge rounds.  Files must be strictly increasing (a code
// twice in one file would share a cell): a duplicate, an unsorted file or a code the tables do not know raise a flag and
// the caller's merge answers.
#ifndef PL_PER_N
#define PL_PER_N 1
#endif
#ifndef PL_BATCH_N
#define PL_BATCH_N 8   /* 16: 9.4 / 10.4 / 16.9 ms for the kernel on 1000 files x 1e6 (90 / 50 / 20 % of a universe each), 8: 8.6 / 10.1 / 16.5, 4: 9.7 / 11.7 / 21.4, 32 (one workgroup per CU): 13.0 / 15.6 */
#endif
constexpr int PL_NT = 512;
constexpr int PL_PER = PL_PER_N;             // codes per thread
constexpr int PL_RANGE = PL_NT * PL_PER;     // codes per range
constexpr int PL_BUCKET_BITS = PL_PER == 1 ? 9 : (PL_PER == 2 ? 10 : 11);
constexpr int PL_BUCKETS = 1 << PL_BUCKET_BITS;
constexpr int PL_SLOTS = 4 * PL_BUCKETS;
constexpr int PL_BATCH = PL_BATCH_N;         // files per batch (one bit each in a code's word; one TaxId cell each)
enum { PL_FLAG_ORDER = 1, PL_FLAG_UNKNOWN = 2 };

__device__ __forceinline__ u32 pl_hash(u64 x) {
    const u32 lo = (u32)x, hi = (u32)(x >> 32);
    return ((lo ^ __builtin_rotateleft32(hi, 15) ^ (hi >> 3)) * 0x9E3779B1u) >> (32 - PL_BUCKET_BITS);
}

template <bool TAX>
__global__ __launch_bounds__(PL_NT) void pl_merge_kernel(PuArgs a) {
    __shared__ __attribute__((aligned(32))) u64 s_tab[PL_SLOTS];
    __shared__ unsigned short s_idx[PL_SLOTS];
    __shared__ u32 s_cnt[PL_RANGE];   // A: records of the code; B: where its next record goes (relative to the range)
    __shared__ u32 s_mask[PL_RANGE];  // B: the files of the batch that hold the code
    __shared__ u32 s_btax[TAX ? PL_RANGE * PL_BATCH : 1];  // B: their TaxIds
    __shared__ u64 s_code[PL_RANGE];
    __shared__ u32 s_seg[PL_RANGE], s_n[PL_RANGE];         // B: where the batch's records of the code go, how many
    __shared__ u32 s_scan[PL_NT / 64 + 1];
    __shared__ u32 s_next;
    __shared__ unsigned long long s_gbase;
    const int tid = (int)threadIdx.x, lane = lane_id(), wave = tid >> 6;
    const u32 r = blockIdx.x, S1 = a.S1;
    const u64 b0 = (u64)r * a.range;
    const u32 ne = (u32)((a.n0 - b0 < (u64)a.range) ? (a.n0 - b0) : (u64)a.range);
    for (int i = tid; i < PL_SLOTS; i += PL_NT) s_tab[i] = PU_EMPTY;
    for (int i = tid; i < PL_RANGE; i += PL_NT) { s_cnt[i] = 0; s_mask[i] = 0; }
    if (tid == 0) { s_next = 0; s_gbase = 0ull; }
    u32 flags = 0;
    __syncthreads();
    u64 code[PL_PER];  // thread t owns the codes t * PL_PER .. (consecutive: one block scan gives their places)
#pragma unroll
    for (int k = 0; k < PL_PER; k++) {
        const u32 i = (u32)tid * PL_PER + (u32)k;
        code[k] = i < ne ? a.base[b0 + i] : PU_EMPTY;
        s_code[i] = code[k];
        if (i >= ne) continue;
        if (code[k] == PU_EMPTY) { flags |= PL_FLAG_UNKNOWN; continue; }  // (an all-ones code is the table's empty marker)
        u32 h = pl_hash(code[k]);
        for (bool placed = false; !placed; h = (h + 1) & (PL_BUCKETS - 1)) {
#pragma unroll
            for (int q = 0; q < 4 && !placed; q++) {
                const u64 old = atomicCAS((unsigned long long *)&s_tab[4 * h + q], (unsigned long long)PU_EMPTY, (unsigned long long)code[k]);
                if (old == PU_EMPTY) { s_idx[4 * h + q] = (unsigned short)i; placed = true; }
            }
        }
    }
    {   // where the range begins in the result: everything the files hold below its first code
        unsigned long long mine = 0;
        for (u32 j = (u32)tid; j < S1; j += PL_NT) mine += a.cuts[(u64)r * S1 + j];
        if (mine) atomicAdd(&s_gbase, mine);
    }
    __syncthreads();
    auto find = [&](u64 x) -> int {
        u32 h = pl_hash(x);
        for (;;) {
            const ulonglong2 *b = reinterpret_cast<const ulonglong2 *>(&s_tab[4 * h]);
            const ulonglong2 p = b[0], q = b[1];
            const bool m0 = p.x == x, m1 = p.y == x, m2 = q.x == x, m3 = q.y == x;
            if (m0 | m1 | m2 | m3) return x == PU_EMPTY ? -1 : (int)s_idx[4 * h + (m0 ? 0 : (m1 ? 1 : (m2 ? 2 : 3)))];
            if (q.y == PU_EMPTY) return -1;
            h = (h + 1) & (PL_BUCKETS - 1);
        }
    };
    // A. one slice of one file, 256 records per step: the file's order is checked, every record's code is looked up, counted
    // and its index kept for the second pass (2 bytes per record instead of the code, and no second look-up)
    auto count_slice = [&](u32 j) {
        // (everything about a slice is wave-uniform: scalar loads)
        const u64 beg = sload_u64(&a.cuts[(u64)r * S1 + j]), end0 = sload_u64(&a.cuts[(u64)(r + 1) * S1 + j]), len = sload_u64(&a.lens[j]);
        const u64 end = end0 < beg ? beg : end0;
        const auto f = as_global((const u64 *)(uintptr_t)sload_u64((const u64 *)&a.files[j]));
        unsigned short *ri = a.rec_idx + sload_u64(&a.rec_off[j]);
        for (u64 p0 = beg; p0 < end; p0 += 256) {
            u64 x[2][2], nx[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const u64 pos = p0 + (u64)u * 128 + 2u * (u32)lane;
                const u64 q0 = pos < len ? pos : len - 1, q1 = pos + 1 < len ? pos + 1 : len - 1, q2 = pos + 2 < len ? pos + 2 : len - 1;
                x[u][0] = f[q0];
                x[u][1] = f[q1];
                nx[u] = f[q2];
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const u64 pos = p0 + (u64)u * 128 + 2u * (u32)lane;
                const bool v0 = pos < end, v1 = pos + 1 < end;
                // strictly increasing, every neighbouring pair of the file once (also across slices)
                if (v0 && pos + 1 < len && x[u][0] >= x[u][1]) flags |= PL_FLAG_ORDER;
                if (v1 && pos + 2 < len && x[u][1] >= nx[u]) flags |= PL_FLAG_ORDER;
#pragma unroll
                for (int w = 0; w < 2; w++) {
                    if (!(w ? v1 : v0)) continue;
                    int i = find(x[u][w]);
                    if (i < 0) {  // (an unsorted file's cut points, an all-ones code: the result is dropped, but every index
                        flags |= PL_FLAG_UNKNOWN;  //  the second pass reads has to be one of the range's)
                        i = 0;
                    }
                    atomicAdd(&s_cnt[i], 1u);
                    ri[pos + (u64)w] = (unsigned short)i;
                }
            }
        }
    };
    for (;;) {
        u32 j = 0;
        if (lane == 0) j = atomicAdd(&s_next, 1u);
        j = (u32)__builtin_amdgcn_readfirstlane((int)j);
        if (j >= S1) break;
        count_slice(j);
    }
    __syncthreads();
    {   // where every code's run begins (relative to the range): exclusive scan of the counts in code order
        u32 v[PL_PER], sum = 0;
#pragma unroll
        for (int k = 0; k < PL_PER; k++) {
            const u32 i = (u32)tid * PL_PER + (u32)k;
            v[k] = i < ne ? s_cnt[i] : 0u;
            sum += v[k];
        }
        u32 tot;
        u32 ex = block_excl_scan_u32<PL_NT>(sum, s_scan, &tot);
#pragma unroll
        for (int k = 0; k < PL_PER; k++) {
            s_cnt[(u32)tid * PL_PER + (u32)k] = ex;
            s_n[(u32)tid * PL_PER + (u32)k] = v[k];
            ex += v[k];
        }
    }
    __syncthreads();
    const u64 gbase = (u64)s_gbase;
    // The CODES of the result need no placement: a code's run is that code, count times -- written here in one piece per
    // code, a wave at a time (1 KB per store), instead of 16 records at a time with the batches below.  Plain codes are
    // done after this.
    for (u32 i = (u32)wave; i < ne; i += PL_NT / 64) {
        const u64 at = gbase + s_cnt[i], cd = s_code[i];
        const u32 n = s_n[i];
        for (u32 q = 2u * (u32)lane; q < n; q += 128) {
            if (q + 1 < n) {
                typedef u64 pl_k2 __attribute__((ext_vector_type(2)));
                typedef pl_k2 __attribute__((aligned(8))) pl_kpair;
                *reinterpret_cast<pl_kpair *>(a.miss + at + q) = pl_kpair{cd, cd};
            } else {
                a.miss[at + q] = cd;
            }
        }
    }
    if (!TAX) {
        if (flags) atomicOr((unsigned long long *)&a.ctl[1], (unsigned long long)flags);
        return;
    }
    __syncthreads();  // (s_n is reused by the batches)
    // B. the files in order, PL_BATCH at a time: index and TaxId of every record of the batch into its code's cells; then the
    // owner of a code writes the batch's records of that code -- neighbours in the result -- in one piece.  (Every record
    // written by the thread that read it, at the place the complete words give it, was measured at 20.6 - 24.7 ms
    // against 14.3: 64 lanes storing 8 bytes into 64 different lines.)
    auto place_slice = [&](u32 j, u32 fj) {
        const u64 beg = sload_u64(&a.cuts[(u64)r * S1 + j]), end0 = sload_u64(&a.cuts[(u64)(r + 1) * S1 + j]);
        const u64 end = end0 < beg ? beg : end0;
        const unsigned short *ri = a.rec_idx + sload_u64(&a.rec_off[j]);
        const u32 *tp = TAX ? (const u32 *)(uintptr_t)sload_u64((const u64 *)&a.tfiles[j]) : nullptr;
        const u32 ftax = (TAX && a.cte) ? (u32)sload_u64(&a.cte[j]) : 0u;  // the file's ONE taxid when it has no array (round 5)
        for (u64 p0 = beg; p0 < end; p0 += 512) {
            u32 i[8], t[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const u64 pos = p0 + (u64)u * 64 + (u32)lane;
                const u64 q = pos < end ? pos : beg;
                i[u] = ri[q];
                i[u] = i[u] < (u32)PL_RANGE ? i[u] : 0u;
                t[u] = (TAX && tp) ? tp[q] : ftax;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const u64 pos = p0 + (u64)u * 64 + (u32)lane;
                if (pos >= end) continue;
                atomicOr(&s_mask[i[u]], 1u << fj);
                if (TAX) s_btax[i[u] * PL_BATCH + (int)fj] = t[u];
            }
        }
    };
    for (u32 bj = 0; bj < S1; bj += PL_BATCH) {
        for (u32 fj = (u32)wave; fj < (u32)PL_BATCH && bj + fj < S1; fj += PL_NT / 64) place_slice(bj + fj, fj);
        __syncthreads();
        // the owner of a code: the batch's TaxIds of the code side by side, its place, its count
#pragma unroll
        for (int k = 0; k < PL_PER; k++) {
            const u32 i = (u32)tid * PL_PER + (u32)k;
            u32 m = s_mask[i];
            s_n[i] = (u32)__popc(m);
            if (m) {
                s_seg[i] = s_cnt[i];
                s_cnt[i] += (u32)__popc(m);
                s_mask[i] = 0;
                if (TAX) {
                    int q = 0;
                    while (m) {
                        const int fj = __ffs((int)m) - 1;
                        m &= m - 1;
                        s_btax[i * PL_BATCH + q] = s_btax[i * PL_BATCH + fj];  // (q <= fj: moving forward in place)
                        q++;
                    }
                }
            }
        }
        __syncthreads();
        // The batch's TaxIds of a code are written by PL_BATCH / 2 lanes, two each: one or two cache lines, and the lanes that share
        // a line share the request.  (One thread writing its code's records -- codes and TaxIds -- one after the other
        // was 32 requests per code and batch: 10.4 of the kernel's 14.2 ms were those stores, at 1.2 TB/s.)
        constexpr u32 LPC = PL_BATCH / 2, CPW = 64 / LPC;  // lanes per code, codes per wave and step
        static_assert(PL_BATCH == 4 || PL_BATCH == 8 || PL_BATCH == 16 || PL_BATCH == 32, "two records per lane");
        for (u32 c0 = (u32)wave * CPW; c0 < (u32)PL_RANGE; c0 += (PL_NT / 64) * CPW) {
            const u32 i = c0 + (u32)lane / LPC, part = (u32)lane % LPC;
            const u32 n = s_n[i], q0 = 2u * part;
            if (q0 < n) {
                const u64 pos = gbase + s_seg[i] + q0;
                if (q0 + 1 < n) {
                    typedef u32 pl_t2 __attribute__((ext_vector_type(2)));
                    typedef pl_t2 __attribute__((aligned(4))) pl_tpair;
                    *reinterpret_cast<pl_tpair *>(a.miss_tax + pos) = pl_tpair{s_btax[i * PL_BATCH + q0], s_btax[i * PL_BATCH + q0 + 1]};
                } else {
                    a.miss_tax[pos] = s_btax[i * PL_BATCH + q0];
                }
            }
        }
        __syncthreads();
    }
    if (flags) atomicOr((unsigned long long *)&a.ctl[1], (unsigned long long)flags);
}

// Base entries per range of the TaxId / counting pass: PT_RANGE, or less when that leaves only a few rounds of workgroups
// (one per CU) with the last one partly empty -- 1e6 base entries: 651 ranges are 2.54 rounds of 256, 768 ranges of 1302
// entries are three full ones.
u32 pt_range_for(const ukm_ctx *c, u64 n0) {
    const u64 cus = (u64)std::max(1, c->num_cu);
    const u64 r_full = (n0 + PT_RANGE - 1) / PT_RANGE;
    if (r_full >= 16 * cus) return (u32)PT_RANGE;
    const u64 rounds = (r_full + cus - 1) / cus;
    const u64 range = (n0 + rounds * cus - 1) / (rounds * cus);
    return (u32)std::min<u64>(PT_RANGE, std::max<u64>(range, 64));
}

double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace

int ukm_punion_mode(const ukm_ctx *c) { return ukm_env_int(c, "UKM_PUNION", -1); }

int ukm_punion_tax_mode(const ukm_ctx *c) { return ukm_env_int(c, "UKM_PUNION_TAX", -1); }

// equal neighbours in a sorted array
__global__ void pu_eqpairs_kernel(const u64 *k, u64 n, u64 *out) {
    const u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const bool eq = g > 0 && g < n && k[g] == k[g - 1];
    const u64 m = __ballot(eq);
    if (m && lane_id() == 0) atomicAdd((unsigned long long *)out, (unsigned long long)__popcll(m));
}

// How many DISTINCT codes do the later files add to the base set?  `smiss` holds the m sampled records the base set lacks
// (pu_sample_kernel).  Two records drawn from the files' new records carry the same code with probability 1 / (distinct new
// codes), so m (m - 1) / 2 pairs show about that many equal pairs: distinct ~ m (m - 1) / (2 pairs) (solved exactly below,
// for samples that see a code several times), and at least the number that would show ONE pair when none is seen (the sample is sized so that a count that just fills the tables'
// room would show about four).  The tables of a range take as many new codes as the range has base
// entries; what comes beyond is listed record by record through global atomics (100 strains that each bring 3 % PRIVATE
// k-mers: 13.5 M new codes on a base set of 5.6 M -- the probe pass took 4.9 ms where the k-way merge finishes the whole
// union in 3.4).  Only looked at when the files' new RECORDS outnumber the room at all.  *too_many: the estimate exceeds it.
static int pu_new_codes(ukm_ctx *c, PuArgs a, u32 nf, double miss_rate, u64 later, bool *too_many) {
    *too_many = false;
    const double expected_misses = (double)later * miss_rate, room = 1.25 * (double)a.n0;
    if (expected_misses <= room || miss_rate <= 0.0) return UKM_OK;
    // enough sampled new records to see ~4 equal pairs if the distinct new codes just filled the tables' room
    const double want = std::sqrt(8.0 * room);
    const u64 nsamp = (u64)std::min(4194304.0, std::max(65536.0, want / miss_rate));
    WsMark mk = ws_mark(c);
    u64 *smiss = nullptr;
    UKM_TRY(ws_alloc_t(c, (size_t)nsamp, &smiss));
    UKM_HIP(hipMemsetAsync(a.ctl, 0, 8 * sizeof(u64), c->stream));
    a.miss = smiss;
    a.miss_cap = nsamp;
    hipLaunchKernelGGL(pu_sample_kernel, dim3((unsigned)((nsamp + 255) / 256)), dim3(256), 0, c->stream, a, (u32)nsamp, nf);
    UKM_HIP(hipGetLastError());
    u64 m = 0;
    UKM_TRY(ukm_read_u64(c, a.ctl + 5, &m));
    m = std::min<u64>(m, nsamp);
    u64 pairs = 0;
    if (m >= 2) {
        UKM_TRY(ukm_dev_sort(c, smiss, nullptr, m, 64));
        hipLaunchKernelGGL(pu_eqpairs_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream, smiss, m, a.ctl + 6);
        UKM_HIP(hipGetLastError());
        UKM_TRY(ukm_read_u64(c, a.ctl + 6, &pairs));
    }
    UKM_HIP(hipMemsetAsync(a.ctl, 0, 8 * sizeof(u64), c->stream));
    ws_release(c, mk);
    // `pairs` = equal NEIGHBOURS of the sorted sample, so m - pairs distinct codes were seen; m draws from D equally likely
    // codes show D (1 - exp(-m / D)) distinct ones: solved for D (few pairs: D ~ m^2 / (2 pairs); a sample that has seen most
    // codes several times: D ~ the codes seen).  No pair at all: at least what would have shown one.
    double distinct = expected_misses;
    if (m >= 2) {
        const double seen = (double)(m - std::min<u64>(std::max<u64>(pairs, 1), m - 1)), ratio = seen / (double)m;
        double lo = 1e-12, hi = 64.0;  // x = m / D; (1 - exp(-x)) / x falls from 1 to 0
        for (int it = 0; it < 80; it++) {
            const double x = 0.5 * (lo + hi);
            if (-std::expm1(-x) / x > ratio) lo = x; else hi = x;
        }
        distinct = std::min(expected_misses, (double)m / (0.5 * (lo + hi)));
    }
    *too_many = distinct > room;
    if (ukm_env(c, "UKM_PUNION_DEBUG"))
        fprintf(stderr, "[punion] %llu sampled new records, %llu equal pairs: ~%.3g distinct new codes among %.3g new records, base set %llu%s\n",
                (unsigned long long)m, (unsigned long long)pairs, distinct, expected_misses, (unsigned long long)a.n0, *too_many ? " -> not this route" : "");
    return UKM_OK;
}

// the share of sampled records that are found in another file (pu_overlap_kernel); the workspace it takes is given back
static int pu_overlap_share(ukm_ctx *c, const u64 *const *keys, const u64 *lens, int S, double *share) {
    *share = 0.0;
    WsMark m = ws_mark(c);
    std::vector<u64> tab((size_t)2 * S);
    for (int j = 0; j < S; j++) {
        tab[(size_t)j] = (u64)(uintptr_t)keys[j];
        tab[(size_t)S + j] = lens[j];
    }
    u64 *d_tab = nullptr, *ctl = nullptr;
    UKM_TRY(ws_alloc_t(c, tab.size(), &d_tab));
    UKM_TRY(ws_alloc_t(c, 8, &ctl));
    UKM_HIP(hipMemcpyAsync(d_tab, tab.data(), tab.size() * sizeof(u64), hipMemcpyHostToDevice, c->stream));
    UKM_HIP(hipMemsetAsync(ctl, 0, 8 * sizeof(u64), c->stream));
    UKM_HIP(hipStreamSynchronize(c->stream));  // `tab` is a pageable host buffer of this frame
    PuArgs a;
    memset(&a, 0, sizeof(a));
    a.files = (const u64 *const *)d_tab;
    a.lens = d_tab + S;
    a.S1 = (u32)S;
    a.ctl = ctl;
    const u32 nsamp = 1u << 14;
    hipLaunchKernelGGL(pu_overlap_kernel, dim3(nsamp / 256), dim3(256), 0, c->stream, a, nsamp);
    UKM_HIP(hipGetLastError());
    u64 h[4] = {0, 0, 0, 0};
    UKM_TRY(ukm_read_u64(c, ctl, h, 4));
    ws_release(c, m);
    if (h[3]) *share = (double)h[2] / (double)h[3];
    return UKM_OK;
}

// one attempt with a base set of k0 files; *low_hit: the later files share too little with it (the caller may try more files)
// ctax (may be null): the file taxid of a stream whose taxids[j] is null
static int probe_union_k0(ukm_ctx *c, const u64 *const *keys, const u32 *const *taxids, const u64 *lens, int S, bool tax, u64 *out,
                          u32 *tout, u64 out_cap, u64 *n_out, bool *fallback, int k0, bool *low_hit, double *hit_rate, const u32 *ctax) {
    *fallback = true;
    *n_out = 0;
    *low_hit = false;
    if (S < k0 + 1) return UKM_OK;
    if (tax) {
        if (!tout) UKM_FAIL(UKM_ERR_INVALID, "union: taxids given but out_taxids is NULL");
        if (c->tax_parent == nullptr) UKM_FAIL(UKM_ERR_NO_TAXONOMY, "union: records carry taxids but no taxonomy is loaded");
        if (c->tax_euler == nullptr || c->tax_node_at == nullptr) return UKM_OK;
    }
    // The base set is built from the k0 LARGEST files (a union does not depend on the order of its files, and neither does
    // the TaxId fold): a small first file -- a plasmid in front of the genomes -- would leave the tables nearly empty
    // and every later record a new code.  Files of one size keep their order.
    std::vector<const u64 *> keys_v(keys, keys + S);
    std::vector<const u32 *> tax_v((size_t)S, nullptr);
    std::vector<u64> lens_v(lens, lens + S);
    std::vector<u32> ct_v((size_t)S, 0u);
    if (tax && taxids) tax_v.assign(taxids, taxids + S);
    {
        std::vector<int> ord((size_t)S);
        for (int j = 0; j < S; j++) ord[(size_t)j] = j;
        std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return lens[x] > lens[y]; });
        std::vector<char> in_base((size_t)S, 0);
        for (int j = 0; j < k0; j++) in_base[(size_t)ord[(size_t)j]] = 1;
        size_t b = 0, l = (size_t)k0;
        for (int j = 0; j < S; j++) {
            const size_t at = in_base[(size_t)j] ? b++ : l++;
            keys_v[at] = keys[j];
            lens_v[at] = lens[j];
            if (tax && taxids) tax_v[at] = taxids[j];
            if (tax && ctax && !(taxids && taxids[j])) ct_v[at] = ctax[j];
        }
    }
    keys = keys_v.data();
    lens = lens_v.data();
    if (tax && (taxids || ctax)) {
        // the base files go through the k-way union, which reads a taxid per record: a base file with ONE taxid gets its array
        for (int j = 0; j < k0; j++)
            if (!tax_v[(size_t)j] && ct_v[(size_t)j] != 0 && lens[j]) {
                u32 *t = nullptr;
                UKM_TRY(ws_alloc_t(c, lens[j], &t));
                UKM_TRY(ukm_dev_fill_u32(c, t, lens[j], ct_v[(size_t)j]));
                tax_v[(size_t)j] = t;
                ct_v[(size_t)j] = 0;
            }
        taxids = tax_v.data();
    }
    u32 range = (u32)PU_RANGE;  // (with TaxIds: chosen below, when the base set's size is known)
    const int mode = ukm_punion_mode(c);
    const bool dbg = ukm_env(c, "UKM_PUNION_DEBUG") != nullptr;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!dbg) return;
        (void)hipStreamSynchronize(c->stream);
        fprintf(stderr, "[punion] %-10s %8.3f ms\n", what, ms_since(t0));
        t0 = std::chrono::steady_clock::now();
    };
    // 1. the base set
    u64 cap0 = 0, later = 0;
    for (int j = 0; j < k0; j++) cap0 += lens[j];
    for (int j = k0; j < S; j++) later += lens[j];
    u64 *base = nullptr;
    u32 *base_tax = nullptr;
    UKM_TRY(ws_alloc_t(c, cap0 + 1, &base));
    if (tax) UKM_TRY(ws_alloc_t(c, cap0 + 1, &base_tax));
    u64 n0 = 0;
    bool fb = false;
    UKM_TRY(ukm_dev_kway(c, UKM_KWAY_UNION, keys, tax ? taxids : nullptr, lens, k0, tax, base, base_tax, cap0, &n0, &fb));
    if (fb || n0 == 0) return UKM_OK;
    lap("base");

    // (the pipelined plain kernel loads 16-byte pairs: later files of fewer than two records do not go through it -- their
    //  one record is put on the list of new codes by the host, which is what the list is: records the final union adds)
    const bool claiming = tax || ukm_env(c, "UKM_PUNION_CLAIM") != nullptr;
    const bool v2 = !claiming;
    std::vector<const u64 *> tiny;
    if (v2) {
        int w = k0;
        for (int j = k0; j < S; j++) {
            if (lens_v[(size_t)j] == 1) { tiny.push_back(keys_v[(size_t)j]); continue; }
            if (lens_v[(size_t)j] == 0) continue;
            keys_v[(size_t)w] = keys_v[(size_t)j];
            lens_v[(size_t)w] = lens_v[(size_t)j];
            w++;
        }
        S = w;
        if (S < k0 + 1) {  // (nothing left to probe: the base set and the listed records are everything)
            keys_v.push_back(nullptr);
            lens_v.push_back(0);
            keys = keys_v.data();
            lens = lens_v.data();
            S = k0 + 1;
            keys_v[(size_t)k0] = nullptr;
            lens_v[(size_t)k0] = 0;
        }
    }
    // device tables of the later files: [pointers S1][lens S1][TaxId pointers S1][file taxid | its number << 32, S1]
    const int S1all = S - k0;
    std::vector<u64> tab((size_t)4 * S1all);
    bool any_ct = false;
    for (int j = 0; j < S1all; j++) {
        tab[(size_t)j] = (u64)(uintptr_t)keys[k0 + j];
        tab[(size_t)S1all + j] = lens[k0 + j];
        tab[(size_t)2 * S1all + j] = (u64)(uintptr_t)((tax && taxids) ? taxids[k0 + j] : nullptr);
        tab[(size_t)3 * S1all + j] = tax ? (u64)ct_v[(size_t)(k0 + j)] : 0ull;
        any_ct = any_ct || tab[(size_t)3 * S1all + j] != 0;
    }
    u64 *d_tab = nullptr, *ctl = nullptr;
    UKM_TRY(ws_alloc_t(c, tab.size(), &d_tab));
    UKM_TRY(ws_alloc_t(c, 8, &ctl));
    UKM_HIP(hipMemcpyAsync(d_tab, tab.data(), tab.size() * sizeof(u64), hipMemcpyHostToDevice, c->stream));
    UKM_HIP(hipMemsetAsync(ctl, 0, 8 * sizeof(u64), c->stream));
    UKM_HIP(hipStreamSynchronize(c->stream));  // `tab` is a pageable host buffer of this frame
    if (any_ct) {
        hipLaunchKernelGGL(pu_cte_kernel, dim3((unsigned)((S1all + 255) / 256)), dim3(256), 0, c->stream, d_tab + 3 * (size_t)S1all, (u32)S1all,
                           ukm_taxdev(c));
        UKM_HIP(hipGetLastError());
    }

    PuArgs a;
    memset(&a, 0, sizeof(a));
    a.base = base;
    a.n0 = n0;
    a.ctl = ctl;
    a.base_tax = base_tax;
    if (tax) a.tax = ukm_taxdev(c);

    // 2. do the later files look like the base set?
    double miss_rate = 0.0;
    {
        a.files = (const u64 *const *)d_tab;
        a.lens = d_tab + S1all;
        a.tfiles = (const u32 *const *)(d_tab + 2 * (size_t)S1all);  // (the sample also looks at the taxids: clade_mode)
        a.cte = any_ct ? d_tab + 3 * (size_t)S1all : nullptr;
        a.S1 = (u32)S1all;
        const u32 nsamp = 1u << 16, nf = (u32)std::min(S1all, 16);
        hipLaunchKernelGGL(pu_sample_kernel, dim3(nsamp / 256), dim3(256), 0, c->stream, a, nsamp, nf);
        UKM_HIP(hipGetLastError());
        u64 h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        UKM_TRY(ukm_read_u64(c, ctl, h, 8));
        if (h[3] == 0) return UKM_OK;
        miss_rate = 1.0 - (double)h[2] / (double)h[3];
        a.clade_mode = pu_clade_mode(c, a.tax, tax, h[2], h[6], h[7]);
        if (a.clade_mode && any_ct) {  // (the file taxids' numbers with their clade codes)
            hipLaunchKernelGGL(pu_cte_kernel, dim3((unsigned)((S1all + 255) / 256)), dim3(256), 0, c->stream, d_tab + 3 * (size_t)S1all, (u32)S1all,
                               a.tax, 1u);
            UKM_HIP(hipGetLastError());
        }
        if (dbg) fprintf(stderr, "[punion] sample: %llu of %llu later records in the base set (n0 = %llu); %llu of them in the entry's clade with another taxid: clade mode %u\n",
                         (unsigned long long)h[2], (unsigned long long)h[3], (unsigned long long)n0, (unsigned long long)h[6], a.clade_mode);
        *hit_rate = 1.0 - miss_rate;
        if (mode != 2 && 1.0 - miss_rate < (tax ? PT_MIN_HIT : PU_MIN_HIT)) {
            *low_hit = true;
            return UKM_OK;
        }
        bool too_many = false;
        if (mode != 2) UKM_TRY(pu_new_codes(c, a, nf, miss_rate, later, &too_many));
        if (too_many) return UKM_OK;  // (more files in the base set would not help: the new codes are the files' own)
    }
    lap("sample");
    // (Plain files stay with the plain kernel down to the same hit rate: its tables claim new codes too -- 2048 per range,
    //  512 of them listed from LDS, the rest in chunks -- and a record costs half of what it costs in the tables of the
    //  TaxId pass even without TaxIds: 1000 files x 1e6, a fifth / an eighth of a universe each: 4.1 / 5.9 ms against
    //  5.9 / 7.7.  UKM_PUNION_CLAIM=1: plain files through the TaxId pass's tables all the same, an experiment.)
    if (claiming) range = pt_range_for(c, n0);
    const u64 R64 = (n0 + range - 1) / range;
    if (R64 > 0x7FFFFFFEull) return UKM_OK;
    a.R = (u32)R64;
    a.range = range;

    // 3. probe pass
    // (+ one partly used chunk of 64 per wave of the grid)
    u64 miss_cap = (u64)((double)later * std::min(1.0, 2.0 * miss_rate + 0.01)) + (1u << 20);
    miss_cap = std::min(miss_cap, later) + 64ull * (std::max(PU2_NT, PT_NT) / 64) * R64 * (u64)((S1all + PU_MAXS - 1) / PU_MAXS) + later / 32;
    miss_cap += tiny.size();
    UKM_TRY(ws_alloc_t(c, miss_cap + 1, &a.miss));
    if (tax) UKM_TRY(ws_alloc_t(c, miss_cap + 1, &a.miss_tax));
    a.miss_cap = miss_cap;
    if (!tiny.empty()) {  // (ctl[0] = records on the list)
        for (size_t i = 0; i < tiny.size(); i++)
            UKM_HIP(hipMemcpyAsync(a.miss + i, tiny[i], sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
        const u64 nt = tiny.size();
        UKM_HIP(hipMemcpyAsync(ctl, &nt, sizeof(u64), hipMemcpyHostToDevice, c->stream));
        UKM_HIP(hipStreamSynchronize(c->stream));  // (`nt` is an object of this frame)
    }
    for (int s0 = 0; s0 < S1all; s0 += PU_MAXS) {
        const int s1 = std::min(PU_MAXS, S1all - s0);
        // (the pointer and length rows of a batch are not adjacent in d_tab: lens sits S1all entries behind)
        a.files = (const u64 *const *)(d_tab + s0);
        a.lens = d_tab + S1all + s0;
        a.tfiles = (const u32 *const *)(d_tab + 2 * (size_t)S1all + s0);
        a.cte = any_ct ? d_tab + 3 * (size_t)S1all + s0 : nullptr;
        a.S1 = (u32)s1;
        WsMark mark = ws_mark(c);
        UKM_TRY(ws_alloc_t(c, ((size_t)a.R + 1) * s1, &a.cuts));
        UKM_TRY(pu_launch_cuts(c, a));
        lap("cuts");
        {
            // One workgroup streams everything that falls into its range: later files whose records crowd into a few
            // ranges (codes beyond the base set's last entry, a dense cluster the base set does not have) would leave
            // the pass to a handful of CUs.  More than 64 x the average load in one range: not this path.
            hipLaunchKernelGGL(pu_load_kernel, dim3((a.R + 255) / 256), dim3(256), 0, c->stream, a);
            UKM_HIP(hipGetLastError());
            u64 heaviest = 0;
            UKM_TRY(ukm_read_u64(c, ctl + 4, &heaviest));
            u64 batch_records = 0;
            for (int j = 0; j < s1; j++) batch_records += lens[k0 + s0 + j];
            const u64 avg = batch_records / a.R + 1;
            if (dbg) fprintf(stderr, "[punion] heaviest range %llu records, average %llu\n", (unsigned long long)heaviest, (unsigned long long)avg);
            if (mode != 2 && heaviest > 64 * avg + 65536) {
                ws_release(c, mark);
                return UKM_OK;  // (*fallback is still true; a batch that already ran only produced list entries)
            }
            UKM_HIP(hipMemsetAsync(ctl + 4, 0, sizeof(u64), c->stream));
        }
        (void)hipEventRecord(c->ev_k0, c->stream);
        if (claiming && a.clade_mode) hipLaunchKernelGGL((pt_probe_kernel<false, true>), dim3(a.R), dim3(PT_NT), 0, c->stream, a);
        else if (claiming) hipLaunchKernelGGL((pt_probe_kernel<false, false>), dim3(a.R), dim3(PT_NT), 0, c->stream, a);
        else hipLaunchKernelGGL(pu2_probe_kernel, dim3(a.R), dim3(PU2_NT), 0, c->stream, a);
        (void)hipEventRecord(c->ev_k1, c->stream);
        c->evk_valid = true;
        UKM_HIP(hipGetLastError());
        lap("probe");
        ws_release(c, mark);  // (the stream orders the next batch's cuts behind this probe)
    }
    u64 h[2] = {0, 0};
    UKM_TRY(ukm_read_u64(c, ctl, h, 2));
    if (dbg) fprintf(stderr, "[punion] S=%d n0=%llu R=%u later=%llu misses=%llu (cap %llu) flags=%llu\n", S, (unsigned long long)n0,
                     a.R, (unsigned long long)later, (unsigned long long)h[0], (unsigned long long)miss_cap, (unsigned long long)h[1]);
    if (h[1] != 0) return UKM_OK;  // unsorted input / overflow: the general route reports or handles it

    // 4. base ∪ misses
    const u64 nm = h[0];
    if (nm == 0) {
        *n_out = n0;
        if (n0 > out_cap)
            UKM_FAIL(UKM_ERR_CAPACITY, "output needs %llu records, capacity is %llu", (unsigned long long)n0, (unsigned long long)out_cap);
        UKM_HIP(hipMemcpyAsync(out, base, n0 * sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
        if (tax) UKM_HIP(hipMemcpyAsync(tout, base_tax, n0 * sizeof(u32), hipMemcpyDeviceToDevice, c->stream));
        *fallback = false;
        return UKM_OK;
    }
    // (with TaxIds: new codes arrive once per range and batch with their fold, unclaimed records one by one: the LCA
    //  over equal codes of the sorted list and over the codes the list shares with the base set finishes the fold)
    UKM_TRY(ukm_dev_sort(c, a.miss, tax ? a.miss_tax : nullptr, nm, 64));
    u64 *mu = nullptr;
    u32 *mut = nullptr;
    UKM_TRY(ws_alloc_t(c, nm + 1, &mu));
    if (tax) UKM_TRY(ws_alloc_t(c, nm + 1, &mut));
    u64 nmu = 0;
    UKM_TRY(ukm_dev_unique(c, a.miss, tax ? a.miss_tax : nullptr, nm, UKM_UNIQUE, mu, mut, nm, &nmu));
    lap("miss sort");
    // (capacity: the 2-way kernel reports the size it needs)
    UKM_TRY(ukm_dev_setop2(c, UKM_OP_UNION, base, base_tax, n0, mu, mut, nmu, 0, out, tout, out_cap, n_out));
    lap("final");
    *fallback = false;
    return UKM_OK;
}

// Base entries per range of the ranked pass (pt_range_for's rule with its own table size)
static u32 pr_range_for(const ukm_ctx *c, u64 n0) {
    const u64 slots = (u64)std::max(1, c->num_cu) * (u64)std::max(1, (160 * 1024) / (PR_TSLOTS * 12 + 1024));  // (workgroups resident at once)
    const u64 r_full = (n0 + PR_TBUCKETS - 1) / PR_TBUCKETS;
    if (r_full >= 16 * slots) return (u32)PR_TBUCKETS;
    const u64 rounds = (r_full + slots - 1) / slots;
    const u64 range = (n0 + rounds * slots - 1) / (rounds * slots);
    return (u32)std::min<u64>(PR_TBUCKETS, std::max<u64>(range, 64));
}

// `union` of files that carry ONE taxid each (ctax[j]; pr_probe_kernel).  Same contract as probe_union_k0.
static int probe_union_ranked(ukm_ctx *c, const u64 *const *keys_in, const u64 *lens_in, int S, const u32 *ctax, u64 *out, u32 *tout,
                              u64 out_cap, u64 *n_out, bool *fallback, int k0, bool *low_hit, double *hit_rate) {
    *fallback = true;
    *n_out = 0;
    *low_hit = false;
    if (S < k0 + 1) return UKM_OK;
    if (!tout) UKM_FAIL(UKM_ERR_INVALID, "union: taxids given but out_taxids is NULL");
    if (c->tax_parent == nullptr) UKM_FAIL(UKM_ERR_NO_TAXONOMY, "union: records carry taxids but no taxonomy is loaded");
    if (c->tax_euler == nullptr || c->tax_node_at == nullptr) return UKM_OK;
    const int mode = ukm_punion_mode(c);
    const bool dbg = ukm_env(c, "UKM_PUNION_DEBUG") != nullptr;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!dbg) return;
        (void)hipStreamSynchronize(c->stream);
        fprintf(stderr, "[punion/ranked] %-10s %8.3f ms\n", what, ms_since(t0));
        t0 = std::chrono::steady_clock::now();
    };
    // 0. the distinct taxid values, ranked by (pre-order number, value)
    std::vector<u32> vals(ctax, ctax + S);
    std::sort(vals.begin(), vals.end());
    vals.erase(std::unique(vals.begin(), vals.end()), vals.end());
    const size_t D = vals.size();
    if (D > (size_t)PR_MAX_RANK) return UKM_OK;
    std::vector<u64> ve(D);
    for (size_t i = 0; i < D; i++) ve[i] = (u64)vals[i];
    u64 *d_ve = nullptr;
    UKM_TRY(ws_alloc_t(c, D, &d_ve));
    UKM_HIP(hipMemcpyAsync(d_ve, ve.data(), D * sizeof(u64), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(pu_cte_kernel, dim3((unsigned)((D + 255) / 256)), dim3(256), 0, c->stream, d_ve, (u32)D, ukm_taxdev(c));
    UKM_HIP(hipGetLastError());
    UKM_HIP(hipMemcpyAsync(ve.data(), d_ve, D * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    UKM_HIP(hipStreamSynchronize(c->stream));
    std::vector<size_t> byrank(D);
    for (size_t i = 0; i < D; i++) byrank[i] = i;
    std::sort(byrank.begin(), byrank.end(), [&](size_t x, size_t y) {
        const u32 ex = (u32)(ve[x] >> 32), ey = (u32)(ve[y] >> 32);
        return ex != ey ? ex < ey : vals[x] < vals[y];
    });
    std::vector<u32> rank_of_val(D), tor(D + 1, 0u), eor(D + 1, 0u);
    for (size_t rnk = 0; rnk < D; rnk++) {
        rank_of_val[byrank[rnk]] = (u32)rnk + 1;
        tor[rnk + 1] = vals[byrank[rnk]];
        eor[rnk + 1] = (u32)(ve[byrank[rnk]] >> 32);
    }
    std::vector<u32> rank_of_file((size_t)S);
    for (int j = 0; j < S; j++)
        rank_of_file[(size_t)j] = rank_of_val[(size_t)(std::lower_bound(vals.begin(), vals.end(), ctax[j]) - vals.begin())];
    // 1. the base set: the PLAIN union of the k0 largest files (files of one size in their order)
    std::vector<int> ord((size_t)S);
    for (int j = 0; j < S; j++) ord[(size_t)j] = j;
    std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return lens_in[x] > lens_in[y]; });
    std::vector<char> in_base((size_t)S, 0);
    std::vector<const u64 *> bkeys((size_t)k0);
    std::vector<u64> blens((size_t)k0);
    u64 cap0 = 0, later = 0, total = 0;
    for (int j = 0; j < k0; j++) {
        in_base[(size_t)ord[(size_t)j]] = 1;
        bkeys[(size_t)j] = keys_in[ord[(size_t)j]];
        blens[(size_t)j] = lens_in[ord[(size_t)j]];
        cap0 += blens[(size_t)j];
    }
    for (int j = 0; j < S; j++) {
        total += lens_in[j];
        if (!in_base[(size_t)j]) later += lens_in[j];
    }
    u64 *base = nullptr;
    UKM_TRY(ws_alloc_t(c, cap0 + 1, &base));
    u64 n0 = 0;
    bool fb = false;
    UKM_TRY(ukm_dev_kway(c, UKM_KWAY_UNION, bkeys.data(), nullptr, blens.data(), k0, false, base, nullptr, cap0, &n0, &fb));
    if (fb || n0 == 0) return UKM_OK;
    lap("base");
    // 2. device tables.  [0, S): every file in the order lowest rank, highest, second lowest, second highest ... (an entry's
    // interval is then final after its first two or three files); [S, 2S): lengths; [2S, 3S): taxid | rank << 32;
    // [3S, 3S + 2 S1): the files outside the base set and their lengths, for the hit-rate sample
    std::vector<int> byr((size_t)S);
    for (int j = 0; j < S; j++) byr[(size_t)j] = j;
    std::stable_sort(byr.begin(), byr.end(), [&](int x, int y) { return rank_of_file[(size_t)x] < rank_of_file[(size_t)y]; });
    std::vector<int> visit;
    visit.reserve((size_t)S);
    for (int lo = 0, hi = S - 1; lo <= hi; lo++, hi--) {
        visit.push_back(byr[(size_t)lo]);
        if (hi != lo) visit.push_back(byr[(size_t)hi]);
    }
    const int S1 = S - k0;
    std::vector<u64> tab((size_t)3 * S + 2 * (size_t)S1);
    for (int q = 0; q < S; q++) {
        const int j = visit[(size_t)q];
        tab[(size_t)q] = (u64)(uintptr_t)keys_in[j];
        tab[(size_t)S + q] = lens_in[j];
        tab[(size_t)2 * S + q] = (u64)ctax[j] | ((u64)rank_of_file[(size_t)j] << 32);
    }
    for (int j = 0, q = 0; j < S; j++)
        if (!in_base[(size_t)j]) {
            tab[(size_t)3 * S + q] = (u64)(uintptr_t)keys_in[j];
            tab[(size_t)3 * S + S1 + q] = lens_in[j];
            q++;
        }
    u64 *d_tab = nullptr, *ctl = nullptr;
    u32 *d_rank = nullptr;
    UKM_TRY(ws_alloc_t(c, tab.size(), &d_tab));
    UKM_TRY(ws_alloc_t(c, 8, &ctl));
    UKM_TRY(ws_alloc_t(c, 2 * (D + 1), &d_rank));
    UKM_HIP(hipMemcpyAsync(d_tab, tab.data(), tab.size() * sizeof(u64), hipMemcpyHostToDevice, c->stream));
    UKM_HIP(hipMemcpyAsync(d_rank, tor.data(), (D + 1) * sizeof(u32), hipMemcpyHostToDevice, c->stream));
    UKM_HIP(hipMemcpyAsync(d_rank + D + 1, eor.data(), (D + 1) * sizeof(u32), hipMemcpyHostToDevice, c->stream));
    UKM_HIP(hipMemsetAsync(ctl, 0, 8 * sizeof(u64), c->stream));
    UKM_HIP(hipStreamSynchronize(c->stream));  // (pageable host buffers of this frame)
    PuArgs a;
    memset(&a, 0, sizeof(a));
    a.base = base;
    a.n0 = n0;
    a.ctl = ctl;
    a.tax = ukm_taxdev(c);
    // 3. do the other files look like the base set?
    double miss_rate = 0.0;
    {
        a.files = (const u64 *const *)(d_tab + 3 * (size_t)S);
        a.lens = d_tab + 3 * (size_t)S + S1;
        a.S1 = (u32)S1;
        const u32 nsamp = 1u << 16, nf = (u32)std::min(S1, 16);
        hipLaunchKernelGGL(pu_sample_kernel, dim3(nsamp / 256), dim3(256), 0, c->stream, a, nsamp, nf);
        UKM_HIP(hipGetLastError());
        u64 h[4] = {0, 0, 0, 0};
        UKM_TRY(ukm_read_u64(c, ctl, h, 4));
        if (h[3] == 0) return UKM_OK;
        miss_rate = 1.0 - (double)h[2] / (double)h[3];
        {
            bool too_many = false;
            if (mode != 2 && 1.0 - miss_rate >= PT_MIN_HIT) UKM_TRY(pu_new_codes(c, a, nf, miss_rate, later, &too_many));
            if (too_many) return UKM_OK;
        }
        if (dbg) fprintf(stderr, "[punion/ranked] sample: %llu of %llu later records in the base set (n0 = %llu, %zu distinct taxids)\n",
                         (unsigned long long)h[2], (unsigned long long)h[3], (unsigned long long)n0, D);
        *hit_rate = 1.0 - miss_rate;
        if (mode != 2 && 1.0 - miss_rate < PT_MIN_HIT) {
            *low_hit = true;
            return UKM_OK;
        }
        UKM_HIP(hipMemsetAsync(ctl, 0, 8 * sizeof(u64), c->stream));
    }
    lap("sample");
    const u32 range = pr_range_for(c, n0);
    const u64 R64 = (n0 + range - 1) / range;
    if (R64 > 0x7FFFFFFEull) return UKM_OK;
    a.R = (u32)R64;
    a.range = range;
    PrTables t;
    t.tax_of_rank = d_rank;
    t.eul_of_rank = d_rank + D + 1;
    t.D = (u32)D;
    t.pair = nullptr;
    if (D <= (size_t)PR_PAIR_MAX) {
        u32 *pair = nullptr;
        UKM_TRY(ws_alloc_t(c, (D + 1) * (D + 1), &pair));
        t.pair = pair;
        hipLaunchKernelGGL(pr_pairs_kernel, dim3((unsigned)(((D + 1) * (D + 1) + 255) / 256)), dim3(256), 0, c->stream, t, a.tax);
        UKM_HIP(hipGetLastError());
    }
    UKM_TRY(ws_alloc_t(c, n0 + 1, &t.base_st));
    UKM_HIP(hipMemsetAsync(t.base_st, 0xFF, (n0 + 1) * sizeof(u32), c->stream));
    // 4. the probe pass over EVERY file
    u64 miss_cap = (u64)((double)later * std::min(1.0, 2.0 * miss_rate + 0.01)) + (1u << 20);
    miss_cap = std::min(miss_cap, total) + 64ull * (PR_TNT / 64) * R64 * (u64)((S + PU_MAXS - 1) / PU_MAXS) + total / 32;
    UKM_TRY(ws_alloc_t(c, miss_cap + 1, &a.miss));
    UKM_TRY(ws_alloc_t(c, miss_cap + 1, &a.miss_tax));
    a.miss_cap = miss_cap;
    for (int s0 = 0; s0 < S; s0 += PU_MAXS) {
        const int s1 = std::min(PU_MAXS, S - s0);
        a.files = (const u64 *const *)(d_tab + s0);
        a.lens = d_tab + S + s0;
        a.cte = d_tab + 2 * (size_t)S + s0;
        a.S1 = (u32)s1;
        WsMark mark = ws_mark(c);
        UKM_TRY(ws_alloc_t(c, ((size_t)a.R + 1) * s1, &a.cuts));
        UKM_TRY(pu_launch_cuts(c, a));
        lap("cuts");
        {
            hipLaunchKernelGGL(pu_load_kernel, dim3((a.R + 255) / 256), dim3(256), 0, c->stream, a);
            UKM_HIP(hipGetLastError());
            u64 heaviest = 0;
            UKM_TRY(ukm_read_u64(c, ctl + 4, &heaviest));
            u64 batch_records = 0;
            for (int q = 0; q < s1; q++) batch_records += tab[(size_t)S + s0 + q];
            const u64 avg = batch_records / a.R + 1;
            if (mode != 2 && heaviest > 64 * avg + 65536) {
                ws_release(c, mark);
                return UKM_OK;
            }
            UKM_HIP(hipMemsetAsync(ctl + 4, 0, sizeof(u64), c->stream));
        }
        (void)hipEventRecord(c->ev_k0, c->stream);
        hipLaunchKernelGGL(pr_probe_kernel, dim3(a.R), dim3(PR_TNT), 0, c->stream, a, t);
        (void)hipEventRecord(c->ev_k1, c->stream);
        c->evk_valid = true;
        UKM_HIP(hipGetLastError());
        lap("probe");
        ws_release(c, mark);
    }
    // (an all-ones code is the tables' empty marker: none of its records was folded into its base entry -- every one of
    //  them is in the list instead --, so the entry, the base set's last, stays out of the final union)
    u64 h[2] = {0, 0}, last = 0;
    UKM_TRY(ukm_read_u64(c, ctl, h, 2));
    UKM_TRY(ukm_read_u64(c, base + n0 - 1, &last));
    if (dbg) fprintf(stderr, "[punion/ranked] S=%d n0=%llu R=%u range=%u records=%llu listed=%llu (cap %llu) flags=%llu\n", S, (unsigned long long)n0,
                     a.R, range, (unsigned long long)total, (unsigned long long)h[0], (unsigned long long)miss_cap, (unsigned long long)h[1]);
    if (h[1] != 0) return UKM_OK;  // unsorted input / overflow: the general route reports or handles it
    const u64 n0e = last == PU_EMPTY ? n0 - 1 : n0;
    u32 *base_tax = nullptr;
    UKM_TRY(ws_alloc_t(c, n0 + 1, &base_tax));
    if (n0e) hipLaunchKernelGGL(pr_settle_kernel, dim3((unsigned)((n0e + 255) / 256)), dim3(256), 0, c->stream, t, a.tax, n0e, base_tax);
    UKM_HIP(hipGetLastError());
    lap("settle");
    const u64 nm = h[0];
    if (nm == 0) {
        *n_out = n0e;
        if (n0e > out_cap)
            UKM_FAIL(UKM_ERR_CAPACITY, "output needs %llu records, capacity is %llu", (unsigned long long)n0e, (unsigned long long)out_cap);
        UKM_HIP(hipMemcpyAsync(out, base, n0e * sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
        UKM_HIP(hipMemcpyAsync(tout, base_tax, n0e * sizeof(u32), hipMemcpyDeviceToDevice, c->stream));
        *fallback = false;
        return UKM_OK;
    }
    UKM_TRY(ukm_dev_sort(c, a.miss, a.miss_tax, nm, 64));
    u64 *mu = nullptr;
    u32 *mut = nullptr;
    UKM_TRY(ws_alloc_t(c, nm + 1, &mu));
    UKM_TRY(ws_alloc_t(c, nm + 1, &mut));
    u64 nmu = 0;
    UKM_TRY(ukm_dev_unique(c, a.miss, a.miss_tax, nm, UKM_UNIQUE, mu, mut, nm, &nmu));
    lap("list sort");
    UKM_TRY(ukm_dev_setop2(c, UKM_OP_UNION, base, base_tax, n0e, mu, mut, nmu, 0, out, tout, out_cap, n_out));
    lap("final");
    *fallback = false;
    return UKM_OK;
}

int ukm_dev_probe_union(ukm_ctx *c, const u64 *const *keys, const u32 *const *taxids, const u64 *lens, int S, bool tax, u64 *out,
                        u32 *tout, u64 out_cap, u64 *n_out, bool *fallback, const u32 *ctax, bool overlap_known) {
    // files of the base set: eight, with TaxIds four (the base union pays an LCA per shared code: 8 files of config 3's
    // shape took as long as a third of the probe pass; the codes the later files add are claimed in the tables anyway).
    // When the later files share too little with it, ONE more attempt with four times as many files -- if the first
    // sample promises enough: files that each hold a share p of a collection hit a base set of k of them with probability
    // 1 - (1 - p)^k, so the next set is expected at 1 - (1 - hit)^4.  (1000 files x 1e6, a tenth / a twentieth of a
    // collection each: with taxids 20.7 / 25.7 ms against 35.2 / 36.1 through the single-pass merge, plain 7.1 / 7.7 against
    // 10.8 / 13.2; a fiftieth each would need 64 files with taxids -- their union with its LCAs alone is 15 ms -- and is
    // left to the merges: 46.4 against 38.3.)
    int k0 = tax ? PT_K0 : PU_K0;
    if (ukm_env(c, "UKM_PUNION_K0")) k0 = std::max(3, std::min(64, atoi(ukm_env(c, "UKM_PUNION_K0"))));  // developer knob
    *fallback = true;
    *n_out = 0;
    if (ukm_punion_mode(c) < 1 && S >= 2 && !overlap_known) {  // (overlap_known: the caller has just taken this sample itself)
        // files that share next to nothing (a record of one is in another with less than 3 % probability: even 32 of them
        // would cover too little): one small kernel says so before a base set is built
        double share = 0.0;
        UKM_TRY(pu_overlap_share(c, keys, lens, S, &share));
        if (share < 0.03) return UKM_OK;
    }
    // every file carries ONE taxid: the ranked pass (its base set is a plain union: eight files)
    bool ranked = tax && ctax != nullptr && !ukm_env_is(c, "UKM_PUNION_RANKED", '0');
    for (int j = 0; j < S && ranked; j++) ranked = !(taxids && taxids[j]);
    if (ranked && !ukm_env(c, "UKM_PUNION_K0")) k0 = PU_K0;
    c->stat_punion_attempts = 0;
    for (int attempt = 0; attempt < 2; attempt++) {
        bool low_hit = false;
        double hit = 0.0;
        WsMark m = ws_mark(c);
        c->stat_punion_attempts++;
        const int rc = ranked ? probe_union_ranked(c, keys, lens, S, ctax, out, tout, out_cap, n_out, fallback, k0, &low_hit, &hit)
                              : probe_union_k0(c, keys, taxids, lens, S, tax, out, tout, out_cap, n_out, fallback, k0, &low_hit, &hit, ctax);
        if (rc != UKM_OK || !*fallback || !low_hit) return rc;
        ws_release(c, m);
        const double miss4 = (1.0 - hit) * (1.0 - hit) * (1.0 - hit) * (1.0 - hit);
        k0 *= 4;
        if (1.0 - miss4 < (tax ? PT_MIN_HIT : PU_MIN_HIT) || k0 > S / 4) break;
    }
    *fallback = true;
    *n_out = 0;
    return UKM_OK;
}

// `common` with a threshold below the number of files by the counting tables of pt_probe_kernel<true>.  first_once: keys[0]
// is the first file as a sorted, duplicate-free set (ukm_common makes it one: every code of the first file counts once,
// common.go:232,244); every record of every other file counts (common.go:262-266).  !first_once: every record of every
// file counts (`merge -d` in its final round = the codes with at least two records, util-sort.go:519-530).  *fallback = true: not this path (few
// or small files, more than PU_MAXS of them, later files that share too little with the first, an unsorted file, a record
// no table could count): nothing that matters was written and the caller's counting merge answers.
int ukm_dev_probe_common(ukm_ctx *c, const u64 *const *keys, const u32 *const *taxids, const u64 *lens, int S, bool tax,
                         u32 threshold, u64 *out, u32 *tout, u64 out_cap, u64 *n_out, bool *fallback, bool first_once, const u32 *ctax) {
    *fallback = true;
    *n_out = 0;
    const int mode = ukm_punion_mode(c);
    if (mode == 0 || S < 3 || S > PU_MAXS || lens[0] == 0) return UKM_OK;
    u64 later = 0;
    for (int j = 1; j < S; j++) later += lens[j];
    if (mode < 1 && (S < 24 || later < (1ull << 26))) return UKM_OK;
    if (tax) {
        if (!tout) UKM_FAIL(UKM_ERR_INVALID, "common: taxids given but out_taxids is NULL");
        if (c->tax_parent == nullptr) UKM_FAIL(UKM_ERR_NO_TAXONOMY, "common: records carry taxids but no taxonomy is loaded");
        if (c->tax_euler == nullptr || c->tax_node_at == nullptr) return UKM_OK;
    }
    const bool dbg = ukm_env(c, "UKM_PUNION_DEBUG") != nullptr;
    if (mode < 1) {
        double share = 0.0;
        UKM_TRY(pu_overlap_share(c, keys, lens, S, &share));
        if (share < 0.03) return UKM_OK;  // (files that share next to nothing: see ukm_dev_probe_union)
    }
    // device tables of ALL files: [pointers S][lens S][TaxId pointers S][file taxid | its number << 32, S]
    std::vector<u64> tab((size_t)4 * S);
    std::vector<u32> ct_v((size_t)S, 0u);
    bool any_ct = false;
    for (int j = 0; j < S; j++) {
        tab[(size_t)j] = (u64)(uintptr_t)keys[j];
        tab[(size_t)S + j] = lens[j];
        tab[(size_t)2 * S + j] = (u64)(uintptr_t)((tax && taxids) ? taxids[j] : nullptr);
        if (tax && ctax && !(taxids && taxids[j])) ct_v[(size_t)j] = ctax[j];
        tab[(size_t)3 * S + j] = (u64)ct_v[(size_t)j];
        any_ct = any_ct || ct_v[(size_t)j] != 0;
    }
    u64 *d_tab = nullptr, *ctl = nullptr;
    UKM_TRY(ws_alloc_t(c, tab.size(), &d_tab));
    UKM_TRY(ws_alloc_t(c, 8, &ctl));
    UKM_HIP(hipMemcpyAsync(d_tab, tab.data(), tab.size() * sizeof(u64), hipMemcpyHostToDevice, c->stream));
    UKM_HIP(hipMemsetAsync(ctl, 0, 8 * sizeof(u64), c->stream));
    UKM_HIP(hipStreamSynchronize(c->stream));  // `tab` is a pageable host buffer of this frame
    if (any_ct) {
        hipLaunchKernelGGL(pu_cte_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, c->stream, d_tab + 3 * (size_t)S, (u32)S, ukm_taxdev(c));
        UKM_HIP(hipGetLastError());
    }
    PuArgs a;
    memset(&a, 0, sizeof(a));
    a.ctl = ctl;
    a.threshold = threshold;
    if (tax) a.tax = ukm_taxdev(c);
    // BASE = the first file, its codes counted once, the other files probed -- when they share enough with it.  Else
    // BASE = the union of the first four files (the TaxIds folded; folding them once more is harmless: LCA(x, x) = x)
    // with no record counted yet, and EVERY file is probed.
    u64 n0 = 0;
    int first = 1;
    auto hit_rate = [&](double *rate) -> int {
        const u32 nsamp = 1u << 16, nf = (u32)std::min((int)a.S1, 16);
        UKM_HIP(hipMemsetAsync(ctl, 0, 8 * sizeof(u64), c->stream));
        hipLaunchKernelGGL(pu_sample_kernel, dim3(nsamp / 256), dim3(256), 0, c->stream, a, nsamp, nf);
        UKM_HIP(hipGetLastError());
        u64 h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        UKM_TRY(ukm_read_u64(c, ctl, h, 8));
        *rate = h[3] ? (double)h[2] / (double)h[3] : 0.0;
        a.clade_mode = pu_clade_mode(c, a.tax, tax, h[2], h[6], h[7]);  // (the last sample in front of the launch decides)
        if (dbg) fprintf(stderr, "[pcommon] sample: %llu of %llu records of the probed files in the base set (n0 = %llu); clade mode %u\n", (unsigned long long)h[2],
                         (unsigned long long)h[3], (unsigned long long)a.n0, a.clade_mode);
        UKM_HIP(hipMemsetAsync(ctl, 0, 8 * sizeof(u64), c->stream));
        return UKM_OK;
    };
    {
        a.base = keys[0];
        a.base_tax = (tax && taxids) ? const_cast<u32 *>(taxids[0]) : nullptr;  // (read only in this mode)
        a.base_ct = ct_v[0];
        a.n0 = n0 = lens[0];
        a.count0 = 1;
        a.files = (const u64 *const *)(d_tab + 1);
        a.lens = d_tab + S + 1;
        a.tfiles = (const u32 *const *)(d_tab + 2 * (size_t)S + 1);
        a.cte = any_ct ? d_tab + 3 * (size_t)S + 1 : nullptr;
        a.S1 = (u32)(S - 1);
        double rate = 0.0;
        if (first_once) UKM_TRY(hit_rate(&rate));
        if (!first_once || (mode != 2 && rate < PT_MIN_HIT)) {
            // (four files, or -- when their sample promises enough, see ukm_dev_probe_union -- sixteen)
            int k0 = std::min(S, PT_K0);
            c->stat_punion_attempts = 0;
            for (int attempt = 0;; attempt++) {
                c->stat_punion_attempts++;
                u64 cap0 = 0;
                for (int j = 0; j < k0; j++) cap0 += lens[j];
                u64 *base = nullptr;
                u32 *base_tax = nullptr;
                WsMark bm = ws_mark(c);
                UKM_TRY(ws_alloc_t(c, cap0 + 1, &base));
                if (tax) UKM_TRY(ws_alloc_t(c, cap0 + 1, &base_tax));
                bool fb = false;
                // (the k-way union reads a taxid per record: a base file with ONE taxid gets its array)
                std::vector<const u32 *> bt((size_t)k0, nullptr);
                for (int j = 0; j < k0 && tax; j++) {
                    bt[(size_t)j] = taxids ? taxids[j] : nullptr;
                    if (!bt[(size_t)j] && ct_v[(size_t)j] != 0 && lens[j]) {
                        u32 *t = nullptr;
                        UKM_TRY(ws_alloc_t(c, lens[j], &t));
                        UKM_TRY(ukm_dev_fill_u32(c, t, lens[j], ct_v[(size_t)j]));
                        bt[(size_t)j] = t;
                    }
                }
                UKM_TRY(ukm_dev_kway(c, UKM_KWAY_UNION, keys, tax ? bt.data() : nullptr, lens, k0, tax, base, base_tax, cap0, &n0, &fb));
                if (fb || n0 == 0) return UKM_OK;
                first = 0;
                a.base = base;
                a.base_tax = base_tax;
                a.base_ct = 0;
                a.n0 = n0;
                a.count0 = 0;
                a.files = (const u64 *const *)d_tab;
                a.lens = d_tab + S;
                a.tfiles = (const u32 *const *)(d_tab + 2 * (size_t)S);
                a.cte = any_ct ? d_tab + 3 * (size_t)S : nullptr;
                a.S1 = (u32)S;
                UKM_TRY(hit_rate(&rate));
                if (mode == 2 || rate >= PT_MIN_HIT) break;
                const double miss4 = (1.0 - rate) * (1.0 - rate) * (1.0 - rate) * (1.0 - rate);
                if (attempt > 0 || 1.0 - miss4 < PT_MIN_HIT || 4 * k0 > S / 4) return UKM_OK;
                ws_release(c, bm);
                k0 *= 4;
            }
        }
    }
    const int S1 = (int)a.S1;
    if (first == 0) later += lens[0];
    const u32 range = pt_range_for(c, n0);
    const u64 R64 = (n0 + range - 1) / range;
    if (R64 > 0x7FFFFFFEull) return UKM_OK;
    a.R = (u32)R64;
    a.range = range;
    // every code leaves at most once: the first file's codes and what the tables claim (as many again at most)
    const u64 list_cap = 2 * n0 + 64 * R64 + 64;
    UKM_TRY(ws_alloc_t(c, list_cap + 1, &a.miss));
    if (tax) UKM_TRY(ws_alloc_t(c, list_cap + 1, &a.miss_tax));
    a.miss_cap = list_cap;
    UKM_TRY(ws_alloc_t(c, ((size_t)a.R + 1) * S1, &a.cuts));
    UKM_TRY(pu_launch_cuts(c, a));
    {
        hipLaunchKernelGGL(pu_load_kernel, dim3((a.R + 255) / 256), dim3(256), 0, c->stream, a);
        UKM_HIP(hipGetLastError());
        u64 heaviest = 0;
        UKM_TRY(ukm_read_u64(c, ctl + 4, &heaviest));
        const u64 avg = later / a.R + 1;
        if (mode != 2 && heaviest > 64 * avg + 65536) return UKM_OK;  // (one workgroup would stream most of the input)
    }
    if (a.clade_mode && a.cte) {  // (the file taxids' numbers with their clade codes)
        hipLaunchKernelGGL(pu_cte_kernel, dim3((unsigned)((S1 + 255) / 256)), dim3(256), 0, c->stream, const_cast<u64 *>(a.cte), (u32)S1, a.tax, 1u);
        UKM_HIP(hipGetLastError());
    }
    (void)hipEventRecord(c->ev_k0, c->stream);
    if (a.clade_mode) hipLaunchKernelGGL((pt_probe_kernel<true, true>), dim3(a.R), dim3(PT_NT), 0, c->stream, a);
    else hipLaunchKernelGGL((pt_probe_kernel<true, false>), dim3(a.R), dim3(PT_NT), 0, c->stream, a);
    (void)hipEventRecord(c->ev_k1, c->stream);
    c->evk_valid = true;
    UKM_HIP(hipGetLastError());
    u64 h[2] = {0, 0};
    UKM_TRY(ukm_read_u64(c, ctl, h, 2));
    if (dbg) fprintf(stderr, "[pcommon] S=%d n0=%llu R=%u later=%llu threshold=%u out=%llu flags=%llu\n", S, (unsigned long long)n0, a.R,
                     (unsigned long long)later, threshold, (unsigned long long)h[0], (unsigned long long)h[1]);
    if (h[1] != 0) return UKM_OK;
    const u64 nm = h[0];
    *fallback = false;
    *n_out = nm;
    if (nm > out_cap)
        UKM_FAIL(UKM_ERR_CAPACITY, "common: output needs %llu records, capacity is %llu", (unsigned long long)nm, (unsigned long long)out_cap);
    if (nm == 0) return UKM_OK;
    // the ranges wrote their codes in the order they finished: one sort puts them in code order (every code is in the
    // list once)
    UKM_TRY(ukm_dev_sort(c, a.miss, tax ? a.miss_tax : nullptr, nm, 64));
    UKM_HIP(hipMemcpyAsync(out, a.miss, nm * sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
    if (tax) UKM_HIP(hipMemcpyAsync(tout, a.miss_tax, nm * sizeof(u32), hipMemcpyDeviceToDevice, c->stream));
    return UKM_OK;
}

int ukm_place_mode(const ukm_ctx *c) { return ukm_env_int(c, "UKM_PLACE", -1); }

// Keep-everything merge by placement (pl_merge_kernel).  *fallback = true: not this path (few or small files, files that
// share too little, a duplicate inside a file, an unsorted file): nothing that matters was written.
int ukm_dev_place_merge(ukm_ctx *c, const u64 *const *keys, const u32 *const *taxids, const u64 *lens, int S, bool tax, u64 *out,
                        u32 *tout, u64 out_cap, u64 *n_out, bool *fallback, const u32 *ctax) {
    *fallback = true;
    *n_out = 0;
    const int mode = ukm_place_mode(c);
    if (mode == 0 || S < 3 || S > PU_MAXS) return UKM_OK;
    u64 N = 0;
    for (int j = 0; j < S; j++) {
        if (lens[j] == 0) return UKM_OK;  // (callers drop empty streams)
        N += lens[j];
    }
    if (mode < 1 && (S < 96 || N < (1ull << 26))) return UKM_OK;  // (64 files x 4e6: 4.7 ms against the k-way merge's 3.7)
    if (tax && !tout) UKM_FAIL(UKM_ERR_INVALID, "merge: taxids given but out_taxids is NULL");
    if (N > out_cap) {
        *n_out = N;
        UKM_FAIL(UKM_ERR_CAPACITY, "merge: output needs %llu records, capacity is %llu", (unsigned long long)N, (unsigned long long)out_cap);
    }
    const bool dbg = ukm_env(c, "UKM_PUNION_DEBUG") != nullptr;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!dbg) return;
        (void)hipStreamSynchronize(c->stream);
        fprintf(stderr, "[place] %-10s %8.3f ms\n", what, ms_since(t0));
        t0 = std::chrono::steady_clock::now();
    };
    // 0. device tables of the files: [pointers S][lens S][TaxId pointers S][offsets of the files' records S][file taxids S]
    std::vector<u64> tab((size_t)5 * S);
    u64 off = 0;
    bool any_ct = false;
    for (int j = 0; j < S; j++) {
        tab[(size_t)j] = (u64)(uintptr_t)keys[j];
        tab[(size_t)S + j] = lens[j];
        tab[(size_t)2 * S + j] = (u64)(uintptr_t)((tax && taxids) ? taxids[j] : nullptr);
        tab[(size_t)3 * S + j] = off;
        tab[(size_t)4 * S + j] = (tax && ctax && !(taxids && taxids[j])) ? (u64)ctax[j] : 0ull;
        any_ct = any_ct || tab[(size_t)4 * S + j] != 0;
        off += lens[j];
    }
    u64 *d_tab = nullptr, *ctl = nullptr;
    UKM_TRY(ws_alloc_t(c, tab.size(), &d_tab));
    UKM_TRY(ws_alloc_t(c, 8, &ctl));
    UKM_HIP(hipMemcpyAsync(d_tab, tab.data(), tab.size() * sizeof(u64), hipMemcpyHostToDevice, c->stream));
    UKM_HIP(hipMemsetAsync(ctl, 0, 8 * sizeof(u64), c->stream));
    UKM_HIP(hipStreamSynchronize(c->stream));  // `tab` is a pageable host buffer of this frame
    PuArgs a;
    memset(&a, 0, sizeof(a));
    a.files = (const u64 *const *)d_tab;
    a.lens = d_tab + S;
    a.tfiles = (const u32 *const *)(d_tab + 2 * (size_t)S);
    a.rec_off = d_tab + 3 * (size_t)S;
    a.cte = any_ct ? d_tab + 4 * (size_t)S : nullptr;
    a.S1 = (u32)S;
    a.ctl = ctl;
    if (mode < 1) {
        // files that share (next to) nothing -- the chunks of an out-of-core sort -- are not for this path: one small kernel
        // says so before a base set is built
        const u32 nsamp = 1u << 14;
        hipLaunchKernelGGL(pu_overlap_kernel, dim3(nsamp / 256), dim3(256), 0, c->stream, a, nsamp);
        UKM_HIP(hipGetLastError());
        u64 h[4] = {0, 0, 0, 0};
        UKM_TRY(ukm_read_u64(c, ctl, h, 4));
        if (dbg) fprintf(stderr, "[place] overlap sample: %llu of %llu records found in another file\n", (unsigned long long)h[2],
                         (unsigned long long)h[3]);
        if (h[3] == 0 || (double)h[2] < 0.05 * (double)h[3]) return UKM_OK;
        if (S <= 1024) {
            // the same sample bounds the slice length the pass would meet (the test behind the base set below): a record's
            // code is in ~1 + share * (S - 1) files, and that record-weighted mean is never below the records-per-code the
            // exact test uses -- a decline here is a decline there, 5 - 7 ms (the base set) earlier
            const double copies = 1.0 + (double)h[2] / (double)h[3] * (double)(S - 1);
            if ((double)PL_RANGE * copies / (double)S < 0.8 * (tax ? 96.0 : 40.0)) return UKM_OK;
        }
        UKM_HIP(hipMemsetAsync(ctl, 0, 8 * sizeof(u64), c->stream));
    }
    lap("overlap");
    // 1. the distinct codes: at most an eighth of the records, or the runs are too short for this path
    const u64 cap0 = mode >= 1 ? N : N / 8 + 1024;
    u64 *base = nullptr;
    UKM_TRY(ws_alloc_t(c, cap0 + 1, &base));
    u64 n0 = 0;
    {
        bool fb = true;
        WsMark m = ws_mark(c);
        const int rc = ukm_dev_probe_union(c, keys, nullptr, lens, S, false, base, nullptr, cap0, &n0, &fb, nullptr, mode < 1);
        ws_release(c, m);
        if (rc == UKM_ERR_CAPACITY) return UKM_OK;
        UKM_TRY(rc);
        if (fb || n0 == 0) return UKM_OK;
    }
    lap("union");
    if (mode < 1 && S <= 1024) {
        // What a workgroup reads of one file for its 512 codes: 512 x (records per code) / files.  Short slices leave the
        // pass to its per-slice work (three loads and, with TaxIds, a share of three barriers per batch): 1000 files x
        // 1e6 with taxids, a fifth of a universe each (102 records per slice) 23.6 ms against the single pass's 26.3, a
        // tenth each (51) 40.5 against 26.9; plain codes 17.6 against 19.1 there.  (More than 1024 files: the other
        // routes end in the pairwise tree -- 3000 x 3e5: 19 ms here, 200 ms there.)
        const double per_slice = (double)PL_RANGE * ((double)N / (double)n0) / (double)S;
        if (per_slice < (tax ? 96.0 : 40.0)) return UKM_OK;
    }
    // 2. cut points of every file at the ranges' first codes
    const u32 range = (u32)PL_RANGE;
    const u64 R64 = (n0 + range - 1) / range;
    if (R64 > 0x7FFFFFFEull) return UKM_OK;
    UKM_TRY(ws_alloc_t(c, N + 8, &a.rec_idx));
    a.S1 = (u32)S;
    a.base = base;
    a.n0 = n0;
    a.R = (u32)R64;
    a.range = range;
    a.ctl = ctl;
    a.miss = out;
    a.miss_tax = tax ? tout : nullptr;
    UKM_TRY(ws_alloc_t(c, ((size_t)a.R + 1) * S, &a.cuts));
    UKM_TRY(pu_launch_cuts(c, a));
    lap("cuts");
    (void)hipEventRecord(c->ev_k0, c->stream);
    if (tax) hipLaunchKernelGGL(pl_merge_kernel<true>, dim3(a.R), dim3(PL_NT), 0, c->stream, a);
    else hipLaunchKernelGGL(pl_merge_kernel<false>, dim3(a.R), dim3(PL_NT), 0, c->stream, a);
    (void)hipEventRecord(c->ev_k1, c->stream);
    c->evk_valid = true;
    UKM_HIP(hipGetLastError());
    lap("place");
    u64 h[2] = {0, 0};
    UKM_TRY(ukm_read_u64(c, ctl, h, 2));
    if (dbg) fprintf(stderr, "[place] S=%d N=%llu n0=%llu R=%u flags=%llu\n", S, (unsigned long long)N, (unsigned long long)n0, a.R,
                     (unsigned long long)h[1]);
    if (h[1] != 0) return UKM_OK;
    *fallback = false;
    *n_out = N;
    return UKM_OK;
}
