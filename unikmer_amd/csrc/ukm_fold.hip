// ukm_fold.hip — `inter` / `diff` over MANY sorted files as ONE launch (round 3; config 4 of BASELINE.json).
//
// The reference folds the files one after the other into a running result (inter.go:205-286, diff.go:379-454).
// Rounds 1-2 did the same with one partition kernel + one 2-way tile kernel per file ("chained fold": 46 us per link,
// i.e. 46 ms for 1000 files of 1e6 codes -- launch / latency bound, the data of a link is 12 MB).  The running result
// only ever SHRINKS and is a subset of the first file, so the fold parallelises over the VALUE SPACE instead of over
// the files:
//   * the first file is cut into ranges of L consecutive records (L = 1024 .. 2560, chosen so that all ranges are
//     resident at once when the file allows it); range r of file j is [lower_bound(f_j, f_0[r L]),
//     lower_bound(f_j, f_0[(r + 1) L])): one kernel does all S x R searches;
//   * one 256-thread workgroup per range keeps its L survivors in REGISTERS (ten per thread, with their taxids)
//     and walks the files: the file's slice is staged in LDS (next slice already in flight into registers), every live
//     survivor binary-searches it, and the reference's per-file rule is applied -- inter: found, else dead, taxid :=
//     LCA (or the mix-taxid rule); diff: dead if found, unless -t keeps it (diff.go:404-409).  No look-back, no
//     inter-workgroup traffic, no per-file launch: 1000 files are 1000 iterations of 5-8 us inside one kernel
//     (measured, config 4: 4 us stream + stage + order check, ~3 us search, ~1.3 us LCAs);
//   * survivors are compacted per range behind the loop; an exclusive scan of the R counts and a gather kernel make
//     the contiguous output.
// Every record of every file is read exactly once and checked for strict order on the way (halo element in front of
// every staged slice); a duplicate code or an unsorted stream raises a flag and the caller takes the exact route of
// rounds 1-2 (multiset semantics need the rank path).  Algorithmic bytes: 8 (+4) per input record read + the
// survivors written twice (temporary + gather).
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <vector>

#include "ukm_device.h"
#include "ukm_fold.h"

namespace {

constexpr int FD_NT = 256;                // threads per workgroup
#ifndef FD_SPT_
#define FD_SPT_ 10
#endif
#ifndef FD_LPT_
#define FD_LPT_ 10
#endif
#ifndef FD_LCAB
#define FD_LCAB 5  /* LCAs a thread has in flight together */
#endif
constexpr int FD_SPT = FD_SPT_;           // survivors per thread (registers)
constexpr int FD_RANGE_MAX = FD_NT * FD_SPT;  // records of the first file per range, at most
constexpr int FD_RANGE_MIN = 1024;
constexpr int FD_LPT = FD_LPT_;           // staged slots per thread
constexpr int FD_SLOTS = FD_NT * FD_LPT;  // LDS slots: one halo + up to FD_SLOTS - 1 records of a slice
constexpr int FD_CH = FD_SLOTS - 1;
// Waves per SIMD the register allocator is held to.  The fold is a chain of ~S dependent steps per workgroup, each a
// latency (a step does not get faster with more resident workgroups), so what counts is that ALL ranges are resident at
// once: with taxids 2 waves per SIMD (up to 256 VGPRs: ten survivors per thread, five LCAs in flight, no spills) x 2560
// records per range = 1.3e6 first-file records per round; the tighter 128-register builds spilled and ran 2x slower.
#ifndef FD_WAVES_TAX
#define FD_WAVES_TAX 2
#endif
#ifndef FD_WAVES_PLAIN
#define FD_WAVES_PLAIN 2
#endif
enum { FD_FLAG_DUP = 1, FD_FLAG_UNSORTED = 2 };
static_assert(FD_SPT % FD_LCAB == 0, "the LCA batches must tile the survivors");

struct FoldArgs {
    const u64 *meta;   // [S][2]: (keys pointer, taxids pointer or 0) of every stream
    const u64 *lens;   // [S]
    u64 *cuts;         // [R][S][2]: (first, end) of range r in file j
    u32 S, R;
    u32 range_len;     // records of the first file per range (multiple of FD_SPT, <= FD_RANGE_MAX)
    u64 *tmp_k;        // survivors of range r, compacted, at r range_len
    u32 *tmp_t;
    u64 *cnt;          // [R]
    u64 *ctl;          // [0] total (written by the scan), [1] flags
    TaxDev T;
    u32 flags;         // UKM_F_MIX_TAXID / UKM_F_CMP_TAXID
};

// cut(r, j) = lower_bound(f_j, f_0[r range_len]) for r = 0 .. R: the first record of range r and the end of range r - 1
__global__ void fd_cuts_kernel(FoldArgs a) {
    const u64 idx = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 total = ((u64)a.R + 1) * a.S;
    if (idx >= total) return;
    const u32 r = (u32)(idx / a.S), j = (u32)(idx % a.S);
    const u64 n = a.lens[j];
    const u64 *k0 = (const u64 *)(uintptr_t)a.meta[0];
    u64 res;
    if (r == 0) res = 0;
    else if (r == a.R) res = n;
    else if (j == 0) res = (u64)r * a.range_len;
    else {
        const u64 split = k0[(u64)r * a.range_len];
        const u64 *k = (const u64 *)(uintptr_t)a.meta[2 * (size_t)j];
        u64 lo = 0, hi = n;
        while (lo < hi) {
            const u64 mid = (lo + hi) >> 1;
            if (k[mid] < split) lo = mid + 1; else hi = mid;
        }
        res = lo;
    }
    if (r < a.R) a.cuts[((size_t)r * a.S + j) * 2] = res;
    if (r > 0) a.cuts[((size_t)(r - 1) * a.S + j) * 2 + 1] = res;
}

struct FdFile {  // what a workgroup needs to know about its slice of one file (all wave-uniform)
    u64 lo, hi;
    const u64 *k;
    const u32 *t;
};

// cnt[i] = number of records of the staged chunk (slots 1 .. m, sorted) below key[i], for the first N slots of a thread,
// in LOCK STEP: a fixed descent over power-of-two strides without a data-dependent branch, so the N chains of dependent
// LDS reads overlap and the whole search costs one chain's latency.  (N is a template argument on purpose: with
// `if (i < spt_now)` around each slot the bodies became separate basic blocks and ran one after the other.)
#ifndef FD_SEARCH_ARITY
#define FD_SEARCH_ARITY 2  /* 4: measured slower (15.1 against 10.6 ms on config 4-core: three probes per round cost more VALU than the halved rounds save, and the taxid build spills at 256 VGPRs) */
#endif
template <int N, int SPT>
__device__ __forceinline__ void fd_search(const u64 *s_k, u32 m, const u64 (&key)[SPT], u32 (&cnt)[SPT]) {
#pragma unroll
    for (int i = 0; i < N; i++) cnt[i] = 0;
#if FD_SEARCH_ARITY == 4
    // 4-ary descent: three independent probes per round, six rounds for a chunk of up to 4095 records instead of twelve
    // binary ones -- the kernel runs at 2 waves per SIMD, where a round is a bare LDS latency + its dependent VALU chain
    u32 stride = 1;
    while (4ull * stride <= m) stride <<= 2;  // largest power of four <= m (m >= 1: the slice is not empty)
    for (; stride; stride >>= 2) {
#pragma unroll
        for (int i = 0; i < N; i++) {
            const u32 t1 = cnt[i] + stride, t2 = t1 + stride, t3 = t2 + stride;
            const u64 k1 = s_k[t1 <= m ? t1 : m], k2 = s_k[t2 <= m ? t2 : m], k3 = s_k[t3 <= m ? t3 : m];
            const u32 c = (u32)(t1 <= m && k1 < key[i]) + (u32)(t2 <= m && k2 < key[i]) + (u32)(t3 <= m && k3 < key[i]);
            cnt[i] += c * stride;  // (sorted chunk: the three answers are monotone)
        }
    }
#else
    for (u32 stride = m ? (1u << (31 - __builtin_clz(m))) : 0u; stride; stride >>= 1) {
#pragma unroll
        for (int i = 0; i < N; i++) {
            const u32 t = cnt[i] + stride;
            const u32 tc = t <= m ? t : m;  // (clamped read; m >= 1 here)
            const bool below = s_k[tc] < key[i];
            cnt[i] = (t <= m && below) ? t : cnt[i];
        }
    }
#endif
}

#ifdef FD_PROFILE
#define FPH(i) do { if (tid == 0) { const u64 _t = clock64(); fph[i] += _t - flast; flast = _t; } } while (0)
#else
#define FPH(i) do {} while (0)
#endif

template <int OP, bool TAX>
__global__ __launch_bounds__(FD_NT) __attribute__((amdgpu_waves_per_eu(TAX ? FD_WAVES_TAX : FD_WAVES_PLAIN, TAX ? FD_WAVES_TAX : FD_WAVES_PLAIN)))
void fd_fold_kernel(FoldArgs a) {
    __shared__ __attribute__((aligned(16))) u64 s_k[FD_SLOTS];
    __shared__ u32 s_t[TAX ? FD_SLOTS : 1];
    __shared__ u32 s_scan[FD_NT / 64 + 1];
    __shared__ u32 s_alive[2];  // live survivors of the workgroup behind the last finished file (ping-pong)
    const int tid = (int)threadIdx.x;
    const u32 r = blockIdx.x, S = a.S;
    const bool mix = (a.flags & UKM_F_MIX_TAXID) != 0;
    const bool cmp = (a.flags & UKM_F_CMP_TAXID) != 0;
    u32 bad = 0;
    // Tables written before this launch: SCALAR loads through the constant address space (sload_u64), issued one file
    // ahead.  (Vector loads from the uniform address + readfirstlane were tried to get the words out of lgkmcnt's way:
    // no gain in the phase counters, and wrong metadata once slices needed several chunks -- not understood, dropped.)
    struct FdRaw { u64 lo, hi, k, t; };
    auto meta_issue = [&](u32 j) -> FdRaw {
        const u32 jj = j < S ? j : S - 1;  // (one past the end: a harmless reload of the last entry)
        FdRaw w;
        const u64 *c = a.cuts + ((size_t)r * S + jj) * 2, *m = a.meta + 2 * (size_t)jj;
        w.lo = sload_u64(c);
        w.hi = sload_u64(c + 1);
        w.k = sload_u64(m);
        w.t = sload_u64(m + 1);
        return w;
    };
    auto meta_take = [&](const FdRaw &w) -> FdFile {
        FdFile f;
        f.lo = w.lo;
        f.hi = w.hi;
        f.k = (const u64 *)(uintptr_t)w.k;
        f.t = (const u32 *)(uintptr_t)w.t;
        return f;
    };
    auto file_meta = [&](u32 j) -> FdFile { return meta_take(meta_issue(j)); };

    // ---- survivors: this range's records of the first file, FD_SPT consecutive ones per thread ----------------------
    u64 sk[FD_SPT];
    u32 st[FD_SPT];
    u32 alive = 0;
    {
        const FdFile f0 = file_meta(0);
        const u64 n0 = f0.hi;  // end of this range in the first file
        const u64 first = f0.lo + (u64)tid * FD_SPT;
        u64 prev = 0;
        bool has_prev = false;
        if (first > 0 && first < n0) { prev = as_global(f0.k)[first - 1]; has_prev = true; }
#pragma unroll
        for (int i = 0; i < FD_SPT; i++) {
            const u64 g = first + i;
            sk[i] = 0;
            st[i] = 0;
            if (g < n0) {
                sk[i] = as_global(f0.k)[g];
                if (TAX && f0.t) st[i] = as_global(f0.t)[g];
                alive |= 1u << i;
                if (has_prev) {
                    if (prev > sk[i]) bad |= FD_FLAG_UNSORTED;
                    if (prev == sk[i]) bad |= FD_FLAG_DUP;
                }
                prev = sk[i];
                has_prev = true;
            }
        }
    }

    // ---- the files, slice by slice -----------------------------------------------------------------------------------
    // a "chunk" = up to FD_CH records of one file's slice, staged behind one halo record (the record in front of it in
    // the file, or nothing at the file's start).  The registers pk / pt hold the chunk that is committed next; the
    // metadata of the file after that is already on its way (scalar loads issued one file ahead).
    u64 pk[FD_LPT];
    u32 pt[FD_LPT];
    u32 nj = 1;              // file of the pending chunk
    FdFile cur = file_meta(1);
    FdRaw nxt = meta_issue(2);  // consumed one file later
    u64 npos = cur.lo;       // first record of the pending chunk
    bool kill_all = false;   // inter: a file without a single record in this range empties it
    auto skip_empty = [&]() {
        while (nj < S && cur.lo >= cur.hi) {
            if (OP == UKM_OP_INTER) kill_all = true;
            nj++;
            cur = meta_take(nxt);
            nxt = meta_issue(nj + 1);
        }
        npos = cur.lo;
    };
    auto issue = [&]() {  // global -> registers for the pending chunk; slot 0 is the halo
        const u64 m = (cur.hi - npos < (u64)FD_CH) ? cur.hi - npos : (u64)FD_CH;
#pragma unroll
        for (int q = 0; q < FD_LPT; q++) {
            const u32 slot = (u32)tid + (u32)q * FD_NT;
            const bool ok = slot <= m && (slot > 0 || npos > 0);
            const u64 g = ok ? npos + slot - 1 : 0;
            pk[q] = as_global(cur.k)[g];  // (record 0 is always mapped: the slice is not empty); GLOBAL loads, see as_global
            if (TAX) pt[q] = cur.t ? as_global(cur.t)[g] : 0u;
        }
    };
    skip_empty();
    if (nj < S) issue();
    u32 found = 0;     // bit i: survivor i matched a record of the current file
    u32 ft[FD_SPT];    // ... and that record's taxid
#pragma unroll
    for (int i = 0; i < FD_SPT; i++) ft[i] = 0;
    // Survivors only die, and the searches below cost VALU / LDS work per survivor SLOT, dead or alive.  So the
    // workgroup re-packs its live survivors densely whenever that frees a slot per thread: spt_now slots per thread are
    // in use (workgroup-uniform), and a fold whose result thins out to 10 % searches one slot per thread, not ten.
    u32 spt_now = FD_SPT;
    u32 files_done = 0;  // files finished since the counter was last looked at (parity selects the counter word)
    bool file_ended = false;
    if (tid < 2) s_alive[tid] = 0;
#ifdef FD_PROFILE
    u64 fph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    u64 flast = clock64();
#endif

    while (nj < S) {
        // (A) commit the pending chunk
        const u64 cpos = npos;
        const u32 m = (u32)((cur.hi - npos < (u64)FD_CH) ? cur.hi - npos : (u64)FD_CH);
        const bool has_halo = cpos > 0;
        const bool last_of_file = cpos + m >= cur.hi;
        __syncthreads();  // the searches of the previous chunk are done
        FPH(0);
        if (file_ended) {
            // re-pack?  (the count was accumulated behind the previous file's rule; one LDS word, read by all)
            const u32 word = (files_done - 1) & 1u;
            const u32 live = (u32)__builtin_amdgcn_readfirstlane((int)s_alive[word]);  // wave-uniform for the compiler too
            const u32 want = live ? (live + FD_NT - 1) / FD_NT : 1u;
            if (want < spt_now) {  // workgroup-uniform
                u32 tot;
                const u32 excl = block_excl_scan_u32<FD_NT>((u32)__popc(alive), s_scan, &tot);
                u32 w = excl;
#pragma unroll
                for (int i = 0; i < FD_SPT; i++) {
                    if (alive & (1u << i)) {
                        s_k[w] = sk[i];
                        if (TAX) s_t[w] = st[i];
                        w++;
                    }
                }
                __syncthreads();
                alive = 0;
#pragma unroll
                for (int i = 0; i < FD_SPT; i++) {
                    const u32 g = (u32)tid * want + (u32)i;
                    if ((u32)i < want && g < tot) {
                        sk[i] = s_k[g];
                        if (TAX) st[i] = s_t[g];
                        alive |= 1u << i;
                    }
                }
                spt_now = want;
                __syncthreads();  // the staging buffer is free again
            }
            if (tid == 0) s_alive[word ^ 1u] = 0;  // the OTHER word: nobody reads it now, the file in progress adds to it
            file_ended = false;
        }
        FPH(1);
#pragma unroll
        for (int q = 0; q < FD_LPT; q++) {
            const u32 slot = (u32)tid + (u32)q * FD_NT;
            s_k[slot] = pk[q];
            if (TAX) s_t[slot] = pt[q];
        }
        FPH(2);
        // (B) the next chunk starts travelling before this one is searched
        if (last_of_file) {
            nj++;
            cur = meta_take(nxt);
            nxt = meta_issue(nj + 1);
            skip_empty();
        } else {
            npos += m;
        }
        if (nj < S) issue();
        FPH(3);
        __syncthreads();
        FPH(4);
        // (C) strict order of what was staged (the halo ties the chunk to the record in front of it)
#ifndef FD_ABL_NOCHECK  /* ablation builds only (tools/build_variant_any.sh fold ...) */
#pragma unroll
        for (int q = 0; q < FD_LPT; q++) {
            const u32 slot = (u32)tid + (u32)q * FD_NT;
            if (slot >= 1 && slot <= m && (slot > 1 || has_halo)) {
                const u64 x = s_k[slot - 1], y = s_k[slot];
                if (x > y) bad |= FD_FLAG_UNSORTED;
                if (x == y) bad |= FD_FLAG_DUP;
            }
        }
#endif
        FPH(5);
#ifndef FD_ABL_NOSEARCH
        // (D) every survivor looks itself up in records [1, m] -- all of a thread's searches in LOCK STEP (a fixed
        //     descent over power-of-two strides, no data-dependent branch): the FD_SPT chains of dependent LDS reads
        //     overlap, so a step costs one search latency whatever FD_SPT is (separate while-loops ran one after the
        //     other: 18 us per file with ten survivors per thread).  cnt = records of the chunk below the key.
        {
            u32 cnt[FD_SPT];
#pragma unroll
            for (int i = 0; i < FD_SPT; i++) cnt[i] = 0;
            // (a few dead slots searched along cost nothing: the variants exist for 1, 2, 3, 4, 6, 8, ... slots)
            if (spt_now <= 1) fd_search<1, FD_SPT>(s_k, m, sk, cnt);
            else if (spt_now == 2) fd_search<2, FD_SPT>(s_k, m, sk, cnt);
            else if (spt_now == 3) fd_search<(FD_SPT < 3 ? FD_SPT : 3), FD_SPT>(s_k, m, sk, cnt);
            else if (spt_now == 4) fd_search<(FD_SPT < 4 ? FD_SPT : 4), FD_SPT>(s_k, m, sk, cnt);
            else if (spt_now <= 6) fd_search<(FD_SPT < 6 ? FD_SPT : 6), FD_SPT>(s_k, m, sk, cnt);
            else if (spt_now <= 8) fd_search<(FD_SPT < 8 ? FD_SPT : 8), FD_SPT>(s_k, m, sk, cnt);
            else fd_search<FD_SPT, FD_SPT>(s_k, m, sk, cnt);
            // (all slots, branch-free: unused slots have cnt = 0 and no alive bit; their reads are harmless)
#pragma unroll
            for (int i = 0; i < FD_SPT; i++) {
                const u32 cand = cnt[i] + 1;
                const u32 cc = cand <= m ? cand : m;
                const bool hit = (alive & (1u << i)) && cand <= m && s_k[cc] == sk[i];
                found |= hit ? (1u << i) : 0u;
                if (TAX) ft[i] = hit ? s_t[cc] : ft[i];
            }
        }
#endif
        FPH(6);
        // (E) behind a file's last chunk: the reference's rule for this file.  The LCAs of four survivors are in
        //     flight together (lca_begin / lca_finish): a thread's pairs are independent random reads.
        if (last_of_file) {
#pragma unroll
            for (int g0 = 0; g0 < FD_SPT; g0 += FD_LCAB) {
                if ((u32)g0 >= spt_now) break;  // workgroup-uniform
                bool need[FD_LCAB];
                bool any_need = false;
#pragma unroll
                for (int u = 0; u < FD_LCAB; u++) {
                    const int i = g0 + u;
                    const u32 bit = 1u << i;
                    const bool hit = (alive & bit) && (found & bit);
                    const u32 ta = st[i], tb = ft[i];
                    if (OP == UKM_OP_INTER) need[u] = TAX && hit && !(mix && (ta == 0 || tb == 0));
                    else need[u] = TAX && cmp && hit && ta != tb;
                    any_need |= need[u];
                }
                LcaReq rq[FD_LCAB];
                const bool lca_round = TAX && __ballot(any_need) != 0;  // wave-uniform: no lane needs one -> no LCA code at all
                if (lca_round) {
#pragma unroll
                    for (int u = 0; u < FD_LCAB; u++) {
                        const int i = g0 + u;
                        if (OP == UKM_OP_INTER) lca_begin(a.T, need[u] ? st[i] : 0u, ft[i], rq[u]);
                        else lca_begin(a.T, need[u] ? ft[i] : 0u, st[i], rq[u]);
                    }
                }
#pragma unroll
                for (int u = 0; u < FD_LCAB; u++) {
                    const int i = g0 + u;
                    const u32 bit = 1u << i;
                    if (!(alive & bit)) continue;
                    const bool hit = (found & bit) != 0;
                    const u32 ta = st[i], tb = ft[i];
                    if (OP == UKM_OP_INTER) {
                        if (!hit) {
                            alive &= ~bit;
                        } else if (TAX) {
                            if (need[u]) st[i] = lca_round ? lca_finish(a.T, rq[u]) : 0u;  // (need implies lca_round)
                            else st[i] = (ta == 0) ? tb : ta;  // mix-taxid with a missing side (tb == 0 when ta != 0)
                        }
                    } else if (hit) {
                        bool keep = false;
                        if (TAX && cmp) keep = (ta == tb) || (need[u] && lca_round && lca_finish(a.T, rq[u]) == ta);  // diff.go:404-409
                        if (!keep) alive &= ~bit;
                    }
                }
            }
            found = 0;
            if (OP == UKM_OP_INTER && kill_all) alive = 0;  // (files skipped above had nothing in this range)
            {
                u32 wl = (u32)__popc(alive);
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) wl += __shfl_xor(wl, d, 64);
                if (lane_id() == 0 && wl) atomicAdd(&s_alive[files_done & 1u], wl);
            }
            files_done++;
            file_ended = true;
        }
        FPH(7);
    }
#ifdef FD_PROFILE
    if (tid == 0 && (r == 0 || r == a.R / 2)) {
        printf("[fold r=%u] cycles per step: topbar=%llu repack=%llu commit=%llu meta+issue=%llu bar=%llu check=%llu search=%llu rule=%llu (steps %u)\n", r,
               (unsigned long long)(fph[0] / (S - 1)), (unsigned long long)(fph[1] / (S - 1)), (unsigned long long)(fph[2] / (S - 1)),
               (unsigned long long)(fph[3] / (S - 1)), (unsigned long long)(fph[4] / (S - 1)), (unsigned long long)(fph[5] / (S - 1)),
               (unsigned long long)(fph[6] / (S - 1)), (unsigned long long)(fph[7] / (S - 1)), S - 1);
    }
#endif
    if (OP == UKM_OP_INTER && kill_all) alive = 0;

    // ---- survivors of the range, compacted in order ----------------------------------------------------------------
    u32 total;
    const u32 excl = block_excl_scan_u32<FD_NT>((u32)__popc(alive), s_scan, &total);
    {
        u64 *ok = a.tmp_k + (size_t)r * a.range_len;
        u32 *ot = TAX ? a.tmp_t + (size_t)r * a.range_len : nullptr;
        u32 w = excl;
#pragma unroll
        for (int i = 0; i < FD_SPT; i++) {
            if (alive & (1u << i)) {
                ok[w] = sk[i];
                if (TAX) ot[w] = st[i];
                w++;
            }
        }
    }
    if (tid == 0) a.cnt[r] = total;
    u32 wbad = 0;
#pragma unroll
    for (u32 f = 1; f <= FD_FLAG_UNSORTED; f <<= 1)
        if (__ballot((bad & f) != 0)) wbad |= f;
    if (wbad && lane_id() == 0) atomicOr((unsigned long long *)&a.ctl[1], (unsigned long long)wbad);
}

// ranges -> contiguous output: workgroup r copies its cnt[r] survivors to out[excl[r] ...)
__global__ void fd_gather_kernel(const u64 *tmp_k, const u32 *tmp_t, const u64 *cnt, const u64 *excl, u64 *out, u32 *tout,
                                 u64 out_cap, u32 range_len) {
    const u32 r = blockIdx.x;
    const u64 n = cnt[r], base = excl[r];
    for (u64 i = threadIdx.x; i < n; i += blockDim.x) {
        const u64 pos = base + i;
        if (pos < out_cap) {
            out[pos] = tmp_k[(size_t)r * range_len + i];
            if (tout) tout[pos] = tmp_t ? tmp_t[(size_t)r * range_len + i] : 0u;
        }
    }
}

}  // namespace

bool ukm_fold_enabled(const ukm_ctx *c) { return !ukm_env_is(c, "UKM_NO_FOLD", '1'); }

// All pointers are device pointers; every stream is non-empty, sorted (the kernel verifies it) and the fold is the
// reference's left fold of streams[1..] into streams[0].  *fallback: a duplicate code was seen (the caller takes the
// exact multiset route); UKM_ERR_UNSORTED: an input is not sorted.
int ukm_dev_range_fold(ukm_ctx *c, int op, const u64 *const *keys, const u32 *const *taxids, const u64 *lens, int S, bool tax,
                       u32 flags, u64 *out, u32 *tout, u64 out_cap, u64 *n_out, bool *fallback) {
    *fallback = false;
    *n_out = 0;
    if (op != UKM_OP_INTER && op != UKM_OP_DIFF) UKM_FAIL(UKM_ERR_INVALID, "range fold: unknown op %d", op);
    if (S < 2 || lens[0] == 0) UKM_FAIL(UKM_ERR_INVALID, "range fold: needs two non-empty streams");
    const bool need_lca = tax && (op == UKM_OP_INTER || (flags & UKM_F_CMP_TAXID));
    if (need_lca && c->tax_parent == nullptr)
        UKM_FAIL(UKM_ERR_NO_TAXONOMY, "ukm_setop2: records carry taxids but no taxonomy is loaded");
    if (tax && !tout) UKM_FAIL(UKM_ERR_INVALID, "range fold: taxids given but out_taxids is NULL");
    // Range length: the whole fold is ONE round of resident workgroups when the first file allows it (a step costs a
    // latency, not a bandwidth: ~1000 dependent steps per workgroup, so a second round of workgroups doubles the time).
    static std::atomic<int> slots_cache[4];  // (zero-initialised; racing first calls compute the same value)
    const int vi = (op == UKM_OP_INTER ? 0 : 2) + (tax ? 1 : 0);
    if (!slots_cache[vi].load(std::memory_order_relaxed)) {
        int per_cu = 0;
        hipError_t e;
        if (vi == 0) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fd_fold_kernel<UKM_OP_INTER, false>, FD_NT, 0);
        else if (vi == 1) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fd_fold_kernel<UKM_OP_INTER, true>, FD_NT, 0);
        else if (vi == 2) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fd_fold_kernel<UKM_OP_DIFF, false>, FD_NT, 0);
        else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fd_fold_kernel<UKM_OP_DIFF, true>, FD_NT, 0);
        if (e != hipSuccess || per_cu <= 0) per_cu = 2;
        slots_cache[vi].store(per_cu * c->num_cu, std::memory_order_relaxed);
    }
    const u64 slots = (u64)slots_cache[vi].load(std::memory_order_relaxed);
    u64 range_len = (lens[0] + slots - 1) / slots;
    range_len = std::max<u64>(range_len, FD_RANGE_MIN);
    range_len = std::min<u64>((range_len + FD_SPT - 1) / FD_SPT * FD_SPT, FD_RANGE_MAX);
    const u64 R64 = (lens[0] + range_len - 1) / range_len;
    if (R64 > 0x7FFFFFFFull) UKM_FAIL(UKM_ERR_INVALID, "range fold: first stream too large");
    const u32 R = (u32)R64;
    {
        // A workgroup walks ITS slice of every file chunk by chunk, ~7 us per chunk.  That is the right trade when the
        // slices are about as long as the range (files comparable to the first one).  A tiny first file against huge
        // later ones would make a handful of workgroups stream whole files through one CU each (10 records against
        // 1000 files of 1e6: one workgroup, 4e5 chunks, seconds) -- the per-file 2-way kernels use the whole chip for
        // that shape.  Not eligible: the caller's chained fold answers.
        u64 rest = 0;
        for (int j = 1; j < S; j++) rest += lens[j];
        const u64 avg_slice = rest / (u64)(S - 1) / R64;
        if (avg_slice > 4ull * FD_CH) {
            *fallback = true;
            return UKM_OK;
        }
    }

    // device tables: [meta S x 2][lens S]
    const size_t ntab = (size_t)3 * S;
    std::vector<u64> tab(ntab);
    for (int j = 0; j < S; j++) {
        tab[(size_t)2 * j] = (u64)(uintptr_t)keys[j];
        tab[(size_t)2 * j + 1] = (u64)(uintptr_t)((tax && taxids) ? taxids[j] : nullptr);
        tab[(size_t)2 * S + j] = lens[j];
    }
    u64 *d_tab = nullptr;
    UKM_TRY(ws_alloc_t(c, ntab, &d_tab));
    UKM_HIP(hipMemcpyAsync(d_tab, tab.data(), ntab * sizeof(u64), hipMemcpyHostToDevice, c->stream));
    UKM_HIP(hipStreamSynchronize(c->stream));  // `tab` is a pageable host buffer of this frame

    FoldArgs a;
    memset(&a, 0, sizeof(a));
    a.meta = d_tab;
    a.lens = d_tab + 2 * (size_t)S;
    a.S = (u32)S;
    a.R = R;
    a.range_len = (u32)range_len;
    a.T = ukm_taxdev(c);
    a.flags = flags;
    u64 *excl = nullptr;
    UKM_TRY(ws_alloc_t(c, (size_t)R * S * 2, &a.cuts));
    UKM_TRY(ws_alloc_t(c, (size_t)R * range_len, &a.tmp_k));
    if (tax) UKM_TRY(ws_alloc_t(c, (size_t)R * range_len, &a.tmp_t));
    UKM_TRY(ws_alloc_t(c, (size_t)R, &a.cnt));
    UKM_TRY(ws_alloc_t(c, (size_t)R + 1, &excl));
    UKM_TRY(ws_alloc_t(c, 8, &a.ctl));
    UKM_HIP(hipMemsetAsync(a.ctl, 0, 8 * sizeof(u64), c->stream));

    const u64 ncuts = ((u64)R + 1) * S;
    hipLaunchKernelGGL(fd_cuts_kernel, dim3((unsigned)((ncuts + 255) / 256)), dim3(256), 0, c->stream, a);
    (void)hipEventRecord(c->ev_k0, c->stream);
    if (op == UKM_OP_INTER) {
        if (tax) hipLaunchKernelGGL((fd_fold_kernel<UKM_OP_INTER, true>), dim3(R), dim3(FD_NT), 0, c->stream, a);
        else hipLaunchKernelGGL((fd_fold_kernel<UKM_OP_INTER, false>), dim3(R), dim3(FD_NT), 0, c->stream, a);
    } else {
        if (tax) hipLaunchKernelGGL((fd_fold_kernel<UKM_OP_DIFF, true>), dim3(R), dim3(FD_NT), 0, c->stream, a);
        else hipLaunchKernelGGL((fd_fold_kernel<UKM_OP_DIFF, false>), dim3(R), dim3(FD_NT), 0, c->stream, a);
    }
    (void)hipEventRecord(c->ev_k1, c->stream);
    c->evk_valid = true;
    UKM_HIP(hipGetLastError());
    UKM_TRY(ukm_dev_exclusive_scan_u64(c, a.cnt, excl, R, a.ctl));  // ctl[0] = total
    hipLaunchKernelGGL(fd_gather_kernel, dim3(R), dim3(256), 0, c->stream, a.tmp_k, tax ? a.tmp_t : nullptr, a.cnt, excl, out,
                       tax ? tout : nullptr, out_cap, a.range_len);
    UKM_HIP(hipGetLastError());
    u64 h[2] = {0, 0};
    UKM_TRY(ukm_read_u64(c, a.ctl, h, 2));
    if (ukm_env(c, "UKM_FOLD_DEBUG"))
        fprintf(stderr, "[fold] op=%d S=%d R=%u range_len=%llu slots=%llu tax=%d flags=%llu out=%llu\n", op, S, R,
                (unsigned long long)range_len, (unsigned long long)slots, (int)tax, (unsigned long long)h[1], (unsigned long long)h[0]);
    if (h[1] & FD_FLAG_UNSORTED) UKM_FAIL(UKM_ERR_UNSORTED, "ukm_setop2: an input stream is not sorted");
    if (h[1] & FD_FLAG_DUP) {
        *fallback = true;
        return UKM_OK;
    }
    *n_out = h[0];
    if (h[0] > out_cap)
        UKM_FAIL(UKM_ERR_CAPACITY, "output needs %llu records, capacity is %llu", (unsigned long long)h[0],
                 (unsigned long long)out_cap);
    return UKM_OK;
}
