// ukm_kway.hip — k-way union / merge of MANY sorted streams in a few passes over HBM: the MI355X
// replacement for the reference's container/heap k-way merge (mergeChunksFile, util-sort.go:196-225,
// 227-606) and for the per-k-mer hash-map probes of an n-file `union` (union.go:186-208).
//
// Why: a binary tree of 2-way merges moves every record log2(n) times through HBM (100 files:
// 7 levels, ~4.3x the algorithmic bytes).  Here a level merges K = 8 streams at once, so 100 files
// take 3 levels and the data crosses HBM ~1.6x.
//
// Design (integer, HBM-bound; no MFMA):
//   * The CODE SPACE is cut once into R value ranges by splitters taken from a regular sample of all
//     inputs (every D-th element of every stream, radix-sorted; every (ns/R)-th sample is a splitter),
//     so ranges carry about the same number of records.  cut[j][r] = lower_bound(stream j, splitter r).
//     Equal codes of different streams always fall into the same range.
//   * One workgroup per (node, range): a node merges up to K child streams.  The workgroup STREAMS
//     through its range like a CPU merge would: per iteration it loads the next chunk (C records) of every
//     child into LDS with coalesced loads, takes v = the smallest "last loaded code" of the children that
//     still have more to come, and consumes every loaded record < v (all copies of such a code are
//     already in LDS, whatever stream they come from) — log2(K) rounds of pairwise merge-path merges in
//     LDS (ping-pong between two buffers), then for `union` a head-of-run scan with the TaxId LCA fold,
//     a block scan and a compacted, coalesced store.  Unconsumed records are simply re-read by the next
//     iteration (they are in L2).  No look-back chain: the output of (node, range) goes to a slot that
//     is known up front (the rank of the range's first input record among the node's leaves), and the
//     record count of every slot is written to a table the next level reads.
//   * Levels share the SAME ranges, so from level 1 on no searching is needed at all: range r of a
//     child is (offset table, count table).  The last level of a `merge` (every record kept) lands
//     contiguously in the caller's buffer; the last level of a `union` is compacted by one small copy
//     kernel (exclusive scan of R counts).
//   Algorithmic bytes of one level: 8 B (+4 B TaxId) per input record read, the same per output record
//   written.
//   Inputs are checked for order while they are loaded; an unsorted stream (legal for the reference's
//   hash-map union) or a run of >= C equal codes inside one stream makes the caller take the
//   sort-based / pairwise route instead.
#include <stdlib.h>

#include <algorithm>
#include <utility>
#include <vector>

#include "ukm_device.h"
#include "ukm_kway.h"

namespace {

constexpr u64 KW_MAX = ~0ull;

enum { KW_FLAG_UNSORTED = 2, KW_FLAG_DEGENERATE = 8 };

struct KwArgs {
    // level 0: the caller's streams
    const u64 *const *leaf_keys;  // [S]
    const u32 *const *leaf_tax;   // [S] (entries may be null: stream without taxids) or nullptr
    // level >= 1: the previous level's output
    const u64 *in_keys;
    const u32 *in_tax;
    const u64 *in_cnt;  // [nprev][R]
    u64 *out_keys;
    u32 *out_tax;
    u64 *out_cnt;     // [nodes][R]
    const u64 *cuts;  // [S][R + 1]
    const u64 *P;     // [S + 1][R + 1]: P[j][r] = sum of cuts[j'][r] over j' < j
    u64 *result;      // [1] = flags
    u32 S, R, nprev;
    u64 span;  // leaves under one CHILD of this level (K^level)
    TaxDev tax;
};

__device__ __forceinline__ u64 wave_min_u64(u64 v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const u64 o = __shfl_xor(v, d, 64);
        v = o < v ? o : v;
    }
    return v;
}

// One round of pairwise merges inside LDS.  Thread `lt` of the pair merges VT consecutive positions
// of merge(A, B) (A first on ties: stable in child order) into out[obase + lt * VT ...].  Both runs are
// followed by a KW_MAX sentinel; positions >= la + lb receive garbage that nobody reads.
// CHECKED: a consumed code equals KW_MAX (a real 2^64-1 hash), so exhaustion is tested by index.
template <bool TAX, bool CHECKED, int VT>
__device__ __forceinline__ void kw_merge_round(const u64 *in, const u32 *tin, u64 *out, u32 *tout, int abase, int la,
                                               int bbase, int lb, int obase, int lt, int tpp) {
    const int L = la + lb;
    int diag = lt * VT;
    diag = diag < L ? diag : L;
    int lo = diag > lb ? diag - lb : 0;
    int hi = diag < la ? diag : la;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const bool le = in[abase + mid] <= in[bbase + diag - 1 - mid];
        lo = le ? mid + 1 : lo;
        hi = le ? hi : mid;
    }
    int pa = abase + lo, pb = bbase + diag - lo;
    const int ea = abase + la, eb = bbase + lb;
    u64 ak = in[pa], bk = in[pb];
    const int o0 = obase + lt * VT;
#pragma unroll
    for (int s = 0; s < VT; s++) {
        bool take_a;
        if (CHECKED) take_a = (pa < ea) && (pb >= eb || ak <= bk);
        else take_a = ak <= bk;
        out[o0 + s] = take_a ? ak : bk;
        if (TAX) tout[o0 + s] = tin[take_a ? pa : pb];
        pa += take_a ? 1 : 0;
        pb += take_a ? 0 : 1;
        const u64 nk = in[take_a ? pa : pb];
        ak = take_a ? nk : ak;
        bk = take_a ? bk : nk;
    }
    // sentinel behind the merged run (written after this thread's own garbage, if any)
    const bool mine = (L >= lt * VT && L < lt * VT + VT) || (lt == tpp - 1 && L == tpp * VT);
    if (mine) out[obase + L] = KW_MAX;
}

// The same merge with the VT outputs kept in REGISTERS (single LDS buffer: the caller stores them after a barrier).
template <bool TAX, bool CHECKED, int VT>
__device__ __forceinline__ void kw_merge_round_regs(const u64 *in, const u32 *tin, int abase, int la, int bbase, int lb, int lt,
                                                    u64 (&ro)[VT], u32 (&rt)[VT]) {
    const int L = la + lb;
    int diag = lt * VT;
    diag = diag < L ? diag : L;
    // merge-path search with BYTE offsets (two adds per probe instead of shifts and index arithmetic)
    const char *inb = reinterpret_cast<const char *>(in);
    const char *a8 = inb + abase * 8, *b8 = inb + (bbase + diag - 1) * 8;
    int lo8 = (diag > lb ? diag - lb : 0) * 8;
    int hi8 = (diag < la ? diag : la) * 8;
    while (lo8 < hi8) {
        const int mid8 = ((lo8 + hi8) >> 1) & ~7;
        const bool le = *reinterpret_cast<const u64 *>(a8 + mid8) <= *reinterpret_cast<const u64 *>(b8 - mid8);
        lo8 = le ? mid8 + 8 : lo8;
        hi8 = le ? hi8 : mid8;
    }
    if (!CHECKED && !TAX) {
        // the kernel is bound by VALU issue: ONE compare per step feeds the minimum, the cursor and the head update;
        // only the B cursor is tracked, in bytes (pa + pb grows by one record per step): 12 instructions instead of 16
        const int sum8 = (abase + bbase + diag) * 8;
        int pb8 = (bbase + diag) * 8 - lo8;
        u64 ak = *reinterpret_cast<const u64 *>(a8 + lo8), bk = *reinterpret_cast<const u64 *>(inb + pb8);
#pragma unroll
        for (int s = 0; s < VT; s++) {
            const bool take_b = bk < ak;
            ro[s] = take_b ? bk : ak;
            asm volatile("" : "+v"(ro[s]));
            pb8 += take_b ? 8 : 0;
            const int idx8 = take_b ? pb8 : sum8 + 8 * (s + 1) - pb8;
            const u64 nk = *reinterpret_cast<const u64 *>(inb + idx8);
            ak = take_b ? ak : nk;
            bk = take_b ? nk : bk;
        }
        return;
    }
    const int lo = lo8 >> 3;
    int pa = abase + lo, pb = bbase + diag - lo;
    const int ea = abase + la, eb = bbase + lb;
    u64 ak = in[pa], bk = in[pb];
#pragma unroll
    for (int s = 0; s < VT; s++) {
        bool take_a;
        if (CHECKED) take_a = (pa < ea) && (pb >= eb || ak <= bk);
        else take_a = ak <= bk;
        ro[s] = take_a ? ak : bk;
        if (TAX) rt[s] = tin[take_a ? pa : pb];
        pa += take_a ? 1 : 0;
        pb += take_a ? 0 : 1;
        const u64 nk = in[take_a ? pa : pb];
        ak = take_a ? nk : ak;
        bk = take_a ? bk : nk;
    }
}

// Which record of the K x C chunk layout does thread t hold as its i-th element?
// VT == M * K + 1: element i < M * K is record (i / K) * NT + t of child i % K, the last element is one of the NT / K
// records behind them (child t / (NT / K)) — the child is a compile-time constant or a shift, no division.
// Other shapes: record (t + i * NT) of the row-major layout.
template <int K, int NT, int VT>
struct KwMap {
    static constexpr int C = NT * VT / K;
    static constexpr int M = (VT - 1) / K;               // whole rows of NT records per child
    static constexpr bool NICE = (VT == M * K + 1) && M >= 1;
    static constexpr int UNIFORM = NICE ? M * K : 0;     // elements i < UNIFORM: the whole workgroup holds records of child i % K
    static __device__ __forceinline__ void at(unsigned t, int i, unsigned &slot, unsigned &e) {
        if (NICE) {
            if (i < M * K) { slot = (unsigned)(i % K); e = (unsigned)(i / K) * NT + t; }
            else { slot = t / (NT / K); e = (unsigned)M * NT + t % (NT / K); }
        } else {
            const unsigned idx = t + (unsigned)i * NT;
            slot = idx / C;
            e = idx - slot * C;
        }
    }
};

// K child streams -> one output stream, for the value range blockIdx.x % R of node blockIdx.x / R.
// UNION: one record per distinct code, TaxId = LCA over every occurrence.  !UNION: every record kept.
// LEAF: the children are the caller's streams (cut table, order check); otherwise the previous level's output.
//
// Per iteration: chunk -> registers -> LDS | barrier | (leaf level: order check | barrier) | wave 0: v, per-child
// lower bound of v in its sorted chunk, cursors, sentinels | barrier | log2(K) merge rounds | emit.
// ONEBUF: one LDS buffer; a merge round keeps its VT outputs in registers and stores them after a barrier (twice the
// records per thread at the same LDS footprint: every round's merge-path search is amortised over 17 instead of 9).
template <int K, int LOGK, bool TAX, bool UNION, bool LEAF, int NT, int VT, bool ONEBUF>
// occupancy target (waves per SIMD): the single-buffer VT = 9 shape fits 64 VGPRs and 19 KB of LDS, i.e. 8 workgroups
// of 256 threads per CU; the kernel is bound by LDS latency and barriers, so more waves in flight is what pays
// (measured per 2e9 input records of level 0: VT 17 / 4 waves 9.0 ms, VT 9 / 5, 6, 8 waves 9.1, 8.5, 8.1 ms)
#ifndef KW_WAVES
#define KW_WAVES ((ONEBUF && VT <= 9) ? (TAX ? 5 : 8) : 4)
#endif
__global__ __launch_bounds__(NT, KW_WAVES) void kway_kernel(KwArgs p) {
    constexpr int TIN = NT * VT;
    constexpr int C = TIN / K;
    constexpr int CS = C + 1;
    constexpr int BUF = K * CS + VT + 7;
    static_assert(TIN % K == 0, "chunk size");
    static_assert((1 << LOGK) == K, "K");
    static_assert(C >= 64, "a wave's 64 consecutive slots touch at most two children");
    static_assert((NT / (K / 2)) % 64 == 0, "a merge pair is handled by whole waves");
    __shared__ __attribute__((aligned(16))) u64 s_a[BUF];
    __shared__ __attribute__((aligned(16))) u64 s_b[ONEBUF ? 2 : BUF];
    __shared__ __attribute__((aligned(16))) u32 s_ta[TAX ? BUF : 2];
    __shared__ __attribute__((aligned(16))) u32 s_tb[(TAX && !ONEBUF) ? BUF : 2];
    __shared__ const u64 *s_ptr[K];
    __shared__ const u32 *s_tptr[K];
    __shared__ u64 s_rem[K];
    __shared__ u64 s_prev[K];
    __shared__ int s_pre[K + 1];
    __shared__ int s_ctl[4];  // [0] M, [1] danger, [2] more
    __shared__ u32 s_scan[NT / 64 + 1];

    const int tid = (int)threadIdx.x;
    const u32 R = p.R;
    const u64 node = blockIdx.x / R;
    const u32 r = blockIdx.x % R;
    const u64 RP = (u64)R + 1;

    // ---- segment of every child in this range ------------------------------------------------------
    if (tid < K) {
        const u64 c = node * K + (u64)tid;
        const u64 *ptr = nullptr;
        const u32 *tptr = nullptr;
        u64 len = 0, prev = 0;
        int bad = 0;
        if (c < p.nprev) {
            if (LEAF) {
                const u64 c0 = p.cuts[c * RP + r], c1 = p.cuts[c * RP + r + 1];
                if (c1 < c0) bad = KW_FLAG_UNSORTED;  // a binary search over an unsorted stream
                else len = c1 - c0;
                ptr = p.leaf_keys[c] + c0;
                if (TAX && p.leaf_tax && p.leaf_tax[c]) tptr = p.leaf_tax[c] + c0;
                if (c0 > 0 && len > 0) prev = ptr[-1];
            } else {
                u64 f = c * p.span, l = f + p.span;
                l = l < p.S ? l : p.S;
                const u64 off = p.P[f * RP + R] + (p.P[l * RP + r] - p.P[f * RP + r]);
                len = p.in_cnt[c * R + r];
                ptr = p.in_keys + off;
                if (TAX) tptr = p.in_tax + off;
            }
        }
        s_ptr[tid] = ptr;
        s_tptr[tid] = tptr;
        s_rem[tid] = len;
        s_prev[tid] = prev;
        if (bad) atomicOr((unsigned long long *)&p.result[1], (unsigned long long)bad);
    }
    // output slot of (node, range): rank of the range's first record among the node's leaves
    u64 out_pos;
    {
        u64 fo = node * K * p.span, lo = fo + K * p.span;
        lo = lo < p.S ? lo : p.S;
        out_pos = p.P[fo * RP + R] + (p.P[lo * RP + r] - p.P[fo * RP + r]);
    }
    const u64 out_pos0 = out_pos;
    __syncthreads();
    {
        u64 tot = 0;
#pragma unroll
        for (int j = 0; j < K; j++) tot += s_rem[j];
        if (tot == 0) {
            if (tid == 0) p.out_cnt[node * R + r] = 0;
            return;
        }
    }
    // always mapped and all ones: lanes without a record load KW_MAX from here (no validity mask, no select
    // afterwards: the kernel is bound by VALU issue)
    const u64 *safe = p.result + 4;

    // the chunk of the coming iteration, in registers: record (tid + i * NT) of the K x C layout
    u64 kk[VT];
    u32 tt[VT];
    auto load_chunk = [&]() {
        unsigned t = (unsigned)tid;
        asm volatile("" : "+v"(t));  // index math of this phase is recomputed, not kept live across the loop
#pragma unroll
        for (int i = 0; i < VT; i++) {
            unsigned slot, e;
            KwMap<K, NT, VT>::at(t, i, slot, e);
            const bool valid = (u64)e < s_rem[slot];
            const u64 *src = s_ptr[slot];
            kk[i] = *as_global(valid ? src + e : safe);  // GLOBAL, not FLAT loads (ukm_device.h: as_global)
            if (TAX) {
                const u32 *ts = s_tptr[slot];
                const bool tv = valid && ts != nullptr;
                tt[i] = *as_global(tv ? ts + e : reinterpret_cast<const u32 *>(safe));
                tt[i] = tv ? tt[i] : 0u;
            }
        }
    };
    if (!ONEBUF) load_chunk();

    for (;;) {
        // (single-buffer shape: 17 staged outputs + 17 prefetched records do not fit 128 VGPRs, so the chunk is loaded
        //  here and the other three workgroups of the CU cover its latency)
        if (ONEBUF) load_chunk();
        // ---- A. chunk -> LDS -------------------------------------------------------------------------
        unsigned ta = (unsigned)tid;
        asm volatile("" : "+v"(ta));
#pragma unroll
        for (int i = 0; i < VT; i++) {
            unsigned slot, e;
            KwMap<K, NT, VT>::at(ta, i, slot, e);
            s_a[slot * CS + e] = kk[i];
            if (TAX) s_ta[slot * CS + e] = tt[i];
        }
        __syncthreads();
        // ---- B. (leaf level) every stream must be sorted: each record against its predecessor -----------------
        if (LEAF) {
            u32 unsorted = 0;
            unsigned tb = (unsigned)tid;
            asm volatile("" : "+v"(tb));
#pragma unroll
            for (int i = 0; i < VT; i++) {
                unsigned slot, e;
                KwMap<K, NT, VT>::at(tb, i, slot, e);
                const u64 pk = e > 0 ? s_a[slot * CS + e - 1] : s_prev[slot];
                unsorted |= (pk > kk[i]) ? 1u : 0u;  // (a lane without a record holds KW_MAX)
            }
            if (__ballot(unsorted != 0) && lane_id() == 0)
                atomicOr((unsigned long long *)&p.result[1], (unsigned long long)KW_FLAG_UNSORTED);
            __syncthreads();  // phase C overwrites a record of every chunk with a sentinel
        }
        // ---- C. v = min over children with more to come of their last loaded code; records < v are consumed
        //         (every copy of such a code is in LDS): per child a lower bound in its sorted chunk; cursors,
        //         sentinels, run lengths ----------------------------------------------------------------------
        if (tid < 64) {
            const int j = tid;
            const bool act = j < K;
            const int jj = j & (K - 1);
            const u64 rem = act ? s_rem[jj] : 0;
            const int avail = rem < (u64)C ? (int)rem : C;
            const bool full = act && rem > (u64)C;
            u64 v = full ? s_a[jj * CS + C - 1] : KW_MAX;
#pragma unroll
            for (int d = K / 2; d >= 1; d >>= 1) {
                const u64 o = __shfl_xor(v, d, 64);
                v = o < v ? o : v;
            }
            const bool any_full = __ballot(full) != 0;
            int cnt = avail;
            if (any_full) {  // first record >= v
                int lo = 0, hi = avail;
                const u64 *ch = s_a + jj * CS;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    const bool lt = ch[mid] < v;
                    lo = lt ? mid + 1 : lo;
                    hi = lt ? hi : mid;
                }
                cnt = lo;
            }
            cnt = act ? cnt : 0;
            const u32 incl = wave_incl_scan_u32((u32)cnt);
            const bool dng = act && !any_full && avail > 0 && s_a[jj * CS + avail - 1] == KW_MAX;
            const bool any_dng = __ballot(dng) != 0;
            const u64 left = wave_reduce_sum_u64(act ? rem - (u64)cnt : 0);
            if (act) {
                s_pre[j + 1] = (int)incl;
                if (cnt > 0) s_prev[j] = s_a[j * CS + cnt - 1];
                s_a[j * CS + cnt] = KW_MAX;  // sentinel (the record it overwrites is re-read next time)
                s_ptr[j] += cnt;
                if (TAX && s_tptr[j]) s_tptr[j] += cnt;
                s_rem[j] = rem - (u64)cnt;
            }
            if (j == 0) {
                s_pre[0] = 0;
                s_ctl[1] = any_dng ? 1 : 0;
                s_ctl[2] = left != 0 ? 1 : 0;
            }
            if (j == K - 1) s_ctl[0] = (int)incl;
        }
        __syncthreads();
        const int M = s_ctl[0];
        const bool danger = s_ctl[1] != 0;
        const bool more = s_ctl[2] != 0;
        if (M == 0) {
            // nothing consumable although records remain: one child's whole chunk is a single code
            if (more && tid == 0) atomicOr((unsigned long long *)&p.result[1], (unsigned long long)KW_FLAG_DEGENERATE);
            break;
        }
        // ---- D. the next chunk's loads are issued now and land while the merges run -----------------------
        if (!ONEBUF && more) load_chunk();
        // ---- E. log2(K) rounds of pairwise merges, ping-pong between the two LDS buffers -----------------
#ifdef KW_ABL_ROUNDS  // experiment only: run fewer merge rounds (wrong results)
        constexpr int NRD = KW_ABL_ROUNDS;
#else
        constexpr int NRD = LOGK;
#endif
#pragma unroll
        for (int rd = 1; rd <= NRD; rd++) {
            const int half = 1 << (rd - 1);       // child slots per input run
            const int npairs = K >> rd;
            const int tpp = NT / npairs;          // threads per pair; tpp * VT = capacity of the output slot
            unsigned te = (unsigned)tid;
            asm volatile("" : "+v"(te));
            const int pr = (int)(te / (unsigned)tpp), lt = (int)(te % (unsigned)tpp);
            const int sa = 2 * pr * half, sb = sa + half;
            const int la = s_pre[sb] - s_pre[sa], lb = s_pre[sb + half] - s_pre[sb];
            if (ONEBUF) {
                u64 ro[VT];
                u32 rt[VT];
                if (danger) kw_merge_round_regs<TAX, true, VT>(s_a, s_ta, sa * CS, la, sb * CS, lb, lt, ro, rt);
                else kw_merge_round_regs<TAX, false, VT>(s_a, s_ta, sa * CS, la, sb * CS, lb, lt, ro, rt);
                __syncthreads();  // every thread has read its inputs: the run is rewritten in place
                const int o0 = sa * CS + lt * VT, L = la + lb;
#pragma unroll
                for (int q = 0; q < VT; q++) {
                    s_a[o0 + q] = ro[q];
                    if (TAX) s_ta[o0 + q] = rt[q];
                }
                if ((L >= lt * VT && L < lt * VT + VT) || (lt == tpp - 1 && L == tpp * VT)) s_a[sa * CS + L] = KW_MAX;
            } else {
                const u64 *in = (rd & 1) ? s_a : s_b;
                u64 *out = (rd & 1) ? s_b : s_a;
                const u32 *tin = (rd & 1) ? s_ta : s_tb;
                u32 *tout = (rd & 1) ? s_tb : s_ta;
                if (danger) kw_merge_round<TAX, true, VT>(in, tin, out, tout, sa * CS, la, sb * CS, lb, sa * CS, lt, tpp);
                else kw_merge_round<TAX, false, VT>(in, tin, out, tout, sa * CS, la, sb * CS, lb, sa * CS, lt, tpp);
            }
            __syncthreads();
        }
        u64 *fin = (ONEBUF || !(LOGK & 1)) ? s_a : s_b;  // merged run [0, M)
        u32 *tfin = (ONEBUF || !(LOGK & 1)) ? s_ta : s_tb;
        u64 *oth = ONEBUF ? s_a : ((LOGK & 1) ? s_a : s_b);
        u32 *toth = ONEBUF ? s_ta : ((LOGK & 1) ? s_ta : s_tb);
        // ---- F. emit ------------------------------------------------------------------------------------------
        const u64 *flush_k = fin;
        const u32 *flush_t = tfin;
        int count = M;
#ifdef KW_ABL_NOEMIT
        if (false) {
#else
        if (UNION) {
#endif
            int tf = tid;
            asm volatile("" : "+v"(tf));
            const int i0 = tf * VT;
            u32 mask = 0;
            u64 hk[VT];  // the thread's VT merged records: read once, the heads are written from these registers
            u32 ht[VT];
            {
                u64 prevk = (i0 > 0 && i0 <= M) ? fin[i0 - 1] : 0;
#pragma unroll
                for (int s = 0; s < VT; s++) {
                    const int i = i0 + s;
                    const u64 k = fin[i];  // (slots behind M hold garbage inside the buffer)
                    hk[s] = k;
                    const bool head = i < M && (i == 0 || k != prevk);
                    prevk = k;
                    mask |= head ? (1u << s) : 0u;
                }
            }
            if (TAX) {
#pragma unroll
                for (int s = 0; s < VT; s++) {
                    ht[s] = 0;
                    if (mask & (1u << s)) {
                        const int i = i0 + s;
                        u32 tx = tfin[i];
                        for (int q = i + 1; q < M && fin[q] == hk[s]; q++) tx = lca_dev(p.tax, tfin[q], tx);
                        ht[s] = tx;
                    }
                }
            }
            u32 tot;
            const u32 excl = block_excl_scan_u32<NT>((u32)__popc(mask), s_scan, &tot);
            // (ONEBUF: compaction in place; the scan's barriers lie between every thread's last read of fin / tfin and
            //  these writes)
            u32 w = excl;
#pragma unroll
            for (int s = 0; s < VT; s++) {
                if (mask & (1u << s)) {
                    oth[w] = hk[s];
                    if (TAX) toth[w] = ht[s];
                    w++;
                }
            }
            __syncthreads();
            flush_k = oth;
            flush_t = toth;
            count = (int)tot;
        }
#ifdef KW_ABL_NOFLUSH
        if (count < 0)
#endif
        {
            u64 *o = p.out_keys + out_pos;
            for (int i = tid; i < count; i += NT) o[i] = flush_k[i];
            if (TAX) {
                u32 *to = p.out_tax + out_pos;
                for (int i = tid; i < count; i += NT) to[i] = flush_t[i];
            }
            out_pos += (u64)count;
        }
        if (!more) break;
        __syncthreads();  // the buffers are refilled by the next iteration
    }
    if (tid == 0) p.out_cnt[node * R + r] = out_pos - out_pos0;
}

// ---- range partition ------------------------------------------------------------------------------------
// sample g = the ((i + 1) * D - 1)-th record of stream j, where g = sample_base[j] + i
__global__ void kw_sample_kernel(const u64 *const *leaf_keys, const u64 *sample_base, u32 S, u64 D, u64 ns,
                                 u64 *samples) {
    const u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ns) return;
    u32 lo = 0, hi = S;  // last j with sample_base[j] <= g
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (sample_base[mid] <= g) lo = mid; else hi = mid;
    }
    const u64 i = g - sample_base[lo];
    samples[g] = leaf_keys[lo][(i + 1) * D - 1];
}

__global__ void kw_cuts_kernel(const u64 *const *leaf_keys, const u64 *leaf_len, u32 S, u32 R, const u64 *samples,
                               u64 ns, u64 *cuts) {
    const u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 RP = (u64)R + 1;
    if (g >= (u64)S * RP) return;
    const u64 j = g / RP, r = g - j * RP;
    const u64 n = leaf_len[j];
    u64 res;
    if (r == 0) res = 0;
    else if (r == R) res = n;
    else {
        // splitter r = sample at rank r * ns / R  (ns >= R is guaranteed by the host)
        const u64 v = samples[r * ns / R];  // r <= 4096, ns ~ 256 R: no overflow
        const u64 *k = leaf_keys[j];
        u64 lo = 0, hi = n;
        while (lo < hi) {
            const u64 mid = (lo + hi) >> 1;
            if (k[mid] < v) lo = mid + 1; else hi = mid;
        }
        res = lo;
    }
    cuts[g] = res;
}

__global__ void kw_prefix_kernel(const u64 *cuts, u32 S, u32 R, u64 *P) {
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 RP = (u64)R + 1;
    if (r > R) return;
    u64 acc = 0;
    for (u32 j = 0; j < S; j++) {
        P[(u64)j * RP + r] = acc;
        acc += cuts[(u64)j * RP + r];
    }
    P[(u64)S * RP + r] = acc;
}

// gather the ranges of the last union level into the caller's buffer: range r goes to dst + excl[r]
__global__ void kw_compact_kernel(const u64 *src, const u32 *tsrc, const u64 *P, u32 S, u32 R, const u64 *cnt,
                                  const u64 *excl, u64 *dst, u32 *tdst, u64 cap, int parts) {
    const u32 r = blockIdx.x / parts, part = blockIdx.x % parts;
    const u64 RP = (u64)R + 1;
    const u64 off = P[(u64)S * RP + r];
    const u64 n = cnt[r], d0 = excl[r];
    if (d0 + n > cap) return;  // the host reports UKM_ERR_CAPACITY
    const u64 lo = n * part / parts, hi = n * (part + 1) / parts;
    for (u64 i = lo + threadIdx.x; i < hi; i += blockDim.x) dst[d0 + i] = src[off + i];
    if (tsrc)
        for (u64 i = lo + threadIdx.x; i < hi; i += blockDim.x) tdst[d0 + i] = tsrc[off + i];
}

// Shapes.  ONE LDS buffer, 256 threads, VT = 9 records per thread for K = 4 and 8 (19 KB plain / 28 KB with TaxIds);
// K = 16 (UKM_KWAY_K=16, plain keys only): 512 threads, VT = 17, 70 KB.  VT is odd so that the threads' consecutive
// 8-byte LDS accesses fall into different banks, and VT = M * K + 1 gives the division-free chunk layout.
// KW_TWOBUF (developer knob): the round-2 first version, two buffers and VT = 9.
template <int K, int LOGK, bool TAX, bool UNION>
void kw_launch(const KwArgs &a, bool leaf, unsigned grid, hipStream_t st) {
#ifdef KW_TWOBUF
    constexpr int VT = TAX ? 5 : 9;
    constexpr bool ONE = false;
#else
#ifndef KW_VT_PLAIN
#define KW_VT_PLAIN 9
#endif
#ifndef KW_NT_SMALLK
#define KW_NT_SMALLK 256
#endif
    constexpr int VT = TAX ? 9 : ((KW_VT_PLAIN - 1) % K == 0 ? KW_VT_PLAIN : 17);
    constexpr bool ONE = true;
#endif
    constexpr int NT = (K == 16) ? 512 : KW_NT_SMALLK;  // a merge pair is handled by whole waves
    if (leaf) hipLaunchKernelGGL((kway_kernel<K, LOGK, TAX, UNION, true, NT, VT, ONE>), dim3(grid), dim3(NT), 0, st, a);
    else hipLaunchKernelGGL((kway_kernel<K, LOGK, TAX, UNION, false, NT, VT, ONE>), dim3(grid), dim3(NT), 0, st, a);
}

template <int K, int LOGK>
void kw_launch_k(const KwArgs &a, bool tax, bool uni, bool leaf, unsigned grid, hipStream_t st) {
    if (tax) {
        if constexpr (K <= 8) {
            if (uni) kw_launch<K, LOGK, true, true>(a, leaf, grid, st);
            else kw_launch<K, LOGK, true, false>(a, leaf, grid, st);
        }
    } else {
        if (uni) kw_launch<K, LOGK, false, true>(a, leaf, grid, st);
        else kw_launch<K, LOGK, false, false>(a, leaf, grid, st);
    }
}

}  // namespace

int ukm_kway_fanin(const ukm_ctx *c) {  // 0 = automatic
    const int k = ukm_env_int(c, "UKM_KWAY_K", 0);
    return (k == 4 || k == 8 || k == 16) ? k : 0;
}

static bool kw_top2_enabled(const ukm_ctx *c) { return !ukm_env_is(c, "UKM_KWAY_TOP2", '0'); }  // developer knob

bool ukm_kway_enabled(const ukm_ctx *c) { return !ukm_env_is(c, "UKM_NO_KWAY", '1'); }

// All pointers are device pointers.  op: UKM_KWAY_UNION / UKM_KWAY_MERGE.  *fallback is set when the inputs
// need the caller's general route (unsorted stream, degenerate run); the output is then undefined.
int ukm_dev_kway(ukm_ctx *c, int op, const u64 *const *keys, const u32 *const *taxids, const u64 *lens, int S,
                 bool tax, u64 *out, u32 *tout, u64 out_cap, u64 *n_out, bool *fallback) {
    *fallback = false;
    *n_out = 0;
    const bool uni = op == UKM_KWAY_UNION;
    if (S <= 0) return UKM_OK;
    if (tax && !tout) UKM_FAIL(UKM_ERR_INVALID, "k-way merge: taxids given but out_taxids is NULL");
    if (tax && uni && c->tax_parent == nullptr)
        UKM_FAIL(UKM_ERR_NO_TAXONOMY, "k-way union: records carry taxids but no taxonomy is loaded");
    u64 N = 0, nmax = 0;
    for (int j = 0; j < S; j++) {
        N += lens[j];
        nmax = std::max<u64>(nmax, lens[j]);
    }
    if (N == 0) return UKM_OK;
    if (!uni && N > out_cap) {
        *n_out = N;
        UKM_FAIL(UKM_ERR_CAPACITY, "k-way merge: output needs %llu records, capacity is %llu", (unsigned long long)N,
                 (unsigned long long)out_cap);
    }
    // fan-in per level: 4 for <= 4 children, else 8.  (16 — one level less for 100 files — was measured slower: its
    // level 0 costs 12.6 ms per 2e9 records against 9.1 ms, more than the saved level; UKM_KWAY_K=16 selects it for plain
    // keys, the TaxId shape does not fit LDS at that fan-in.)
    const int kpref = ukm_kway_fanin(c);
    auto pick_k = [&](u64 nchildren) -> int {
        if (nchildren <= 4 || kpref == 4) return 4;
        if (nchildren <= 8 || tax) return 8;
        return kpref == 16 ? 16 : 8;
    };
    int K = pick_k((u64)S);
    // UKM_KWAY_DEBUG=1: per-phase device times on stderr (developer knob; adds events + one sync)
    const bool dbg = ukm_env(c, "UKM_KWAY_DEBUG") != nullptr;
    std::vector<std::pair<const char *, hipEvent_t>> marks;
    auto mark = [&](const char *name) {
        if (!dbg) return;
        hipEvent_t e;
        if (hipEventCreate(&e) == hipSuccess) {
            (void)hipEventRecord(e, c->stream);
            marks.emplace_back(name, e);
        }
    };
    mark("start");
    int levels = 0;
    for (u64 nn = (u64)S; ; ) { const int kk = pick_k(nn); levels++; nn = (nn + kk - 1) / kk; if (nn <= 1) break; }

    // ---- ranges -------------------------------------------------------------------------------------------
    // about one range per 2048 records of an average stream (a child's share of a range should be several
    // chunks long), at most 4096, and a bounded table
    u64 R64 = (N / (u64)S) / 2048;
    R64 = std::min<u64>(R64, 4096);
    R64 = std::min<u64>(R64, ((u64)1 << 22) / ((u64)S + 1));
    // The top level has one workgroup per range, and a level moves whatever is left of the N records: with many short
    // files the rule above leaves the upper levels too few workgroups (1000 files x 1e6: 488 ranges; a keep-everything
    // merge with taxids ran levels of 7, 7, 24 and 54 ms, with 4096 ranges 8, 6, 7 and 12 ms; a union of files that
    // hardly overlap 61 ms against 24, with taxids 149 against 52).  Only a union WITH taxids of files that overlap
    // heavily prefers the longer shares of the old rule (its upper levels are small, its level 0 slows from 24 to 34 ms
    // at 4096 ranges): whether files overlap is not known here, so it takes 2048 (33 / 63 ms in the two cases).
    {
        u64 want = std::min<u64>((uni && tax) ? 2048 : 4096, N / 16384);
        if (uni) want = std::min<u64>(want, (N / (u64)S) / 64);  // (a union's level 0 wants a child's share >= 64 records)
        R64 = std::max<u64>(R64, want);
    }
    R64 = std::min<u64>(R64, ((u64)1 << 22) / ((u64)S + 1));
    if (R64 < 1) R64 = 1;
    {
        const char *e = ukm_env(c, "UKM_KWAY_R");  // developer knob
        if (e && atoll(e) > 0) R64 = std::min<u64>((u64)atoll(e), ((u64)1 << 22) / ((u64)S + 1));
    }
    u64 D = 1, ns = 0;
    std::vector<u64> sample_base((size_t)S + 1, 0);
    if (R64 > 1) {
        // every D-th record of every stream is a sample: about 256 samples per range
        D = std::max<u64>(1, N / (R64 * 256));
        for (int j = 0; j < S; j++) sample_base[(size_t)j + 1] = sample_base[(size_t)j] + lens[j] / D;
        ns = sample_base[(size_t)S];
        if (ns < R64) R64 = 1;
    }
    const u32 R = (u32)R64;
    const u64 RP = (u64)R + 1;

    // ---- device tables ---------------------------------------------------------------------------------------
    // [leaf_keys S][leaf_tax S][leaf_len S][sample_base S+1]
    const size_t ntab = (size_t)4 * S + 1;
    std::vector<u64> tab(ntab);
    for (int j = 0; j < S; j++) {
        tab[(size_t)j] = (u64)(uintptr_t)keys[j];
        tab[(size_t)S + j] = (u64)(uintptr_t)((tax && taxids) ? taxids[j] : nullptr);
        tab[(size_t)2 * S + j] = lens[j];
    }
    for (int j = 0; j <= S; j++) tab[(size_t)3 * S + j] = sample_base[(size_t)j];
    u64 *d_tab = nullptr;
    UKM_TRY(ws_alloc_t(c, ntab, &d_tab));
    UKM_HIP(hipMemcpyAsync(d_tab, tab.data(), ntab * sizeof(u64), hipMemcpyHostToDevice, c->stream));
    UKM_HIP(hipStreamSynchronize(c->stream));  // `tab` is a pageable host buffer of this frame
    const u64 *const *d_keys = reinterpret_cast<const u64 *const *>(d_tab);
    const u32 *const *d_tax = reinterpret_cast<const u32 *const *>(d_tab + S);
    const u64 *d_len = d_tab + 2 * (size_t)S;
    const u64 *d_sbase = d_tab + 3 * (size_t)S;

    u64 *cuts = nullptr, *P = nullptr, *ctl = nullptr;
    UKM_TRY(ws_alloc_t(c, (size_t)S * RP, &cuts));
    UKM_TRY(ws_alloc_t(c, ((size_t)S + 1) * RP, &P));
    UKM_TRY(ws_alloc_t(c, 8, &ctl));
    UKM_HIP(hipMemsetAsync(ctl, 0, 8 * sizeof(u64), c->stream));
    UKM_HIP(hipMemsetAsync(ctl + 4, 0xFF, 2 * sizeof(u64), c->stream));  // ctl[4]: KW_MAX for the lanes without a record
    u64 *samples = nullptr;
    if (R > 1) {
        UKM_TRY(ws_alloc_t(c, ns, &samples));
        hipLaunchKernelGGL(kw_sample_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, c->stream, d_keys, d_sbase,
                           (u32)S, D, ns, samples);
        UKM_TRY(ukm_dev_sort(c, samples, nullptr, ns, 64));
    }
    mark("sample+sort");
    hipLaunchKernelGGL(kw_cuts_kernel, dim3((unsigned)(((u64)S * RP + 255) / 256)), dim3(256), 0, c->stream, d_keys, d_len,
                       (u32)S, R, samples, ns, cuts);
    hipLaunchKernelGGL(kw_prefix_kernel, dim3((unsigned)((RP + 255) / 256)), dim3(256), 0, c->stream, cuts, (u32)S, R, P);

    mark("cuts+prefix");
    // ---- buffers of the levels ---------------------------------------------------------------------------------
    // a `merge`'s last level writes the caller's buffer; everything else goes to (at most two) N-record buffers
    const int ntemp = uni ? std::min(levels, 2) : std::min(levels - 1, 2);
    u64 *tk[2] = {nullptr, nullptr};
    u32 *tt[2] = {nullptr, nullptr};
    for (int i = 0; i < ntemp; i++) {
        UKM_TRY(ws_alloc_t(c, N + 1, &tk[i]));
        if (tax) UKM_TRY(ws_alloc_t(c, N + 1, &tt[i]));
    }
    u64 nprev = (u64)S, span = 1;
    const u64 *in_k = nullptr, *in_cnt = nullptr;
    const u32 *in_t = nullptr;
    u64 *last_cnt = nullptr;
    const u64 *last_k = nullptr;
    const u32 *last_t = nullptr;
    for (int lv = 0; lv < levels; lv++) {
        K = pick_k(nprev);
        const u64 nodes = (nprev + (u64)K - 1) / (u64)K;
        const bool final_lv = lv == levels - 1;
        u64 *ok;
        u32 *ot;
        if (final_lv && !uni) {
            ok = out;
            ot = tout;
        } else {
            ok = tk[lv & 1];
            ot = tt[lv & 1];
        }
        if (final_lv && !uni && nprev == 2 && lv > 0 && N >= (1u << 20) && kw_top2_enabled(c)) {
            // A keep-everything merge leaves every node's output as ONE sorted array (slot = rank among the node's
            // leaves), so a top level of two children is a plain 2-way merge: the merge-path tile kernel of
            // ukm_setops.hip (every record kept, the first child's copies of a code first = stream order) runs it at
            // ≈ 4.5 TB/s where one streaming workgroup per range manages 2 (1000 files x 1e6 with taxids: 12 -> 6 ms).
            u64 fl = 0;
            UKM_TRY(ukm_read_u64(c, ctl + 1, &fl));
            if (fl & (KW_FLAG_UNSORTED | KW_FLAG_DEGENERATE)) {
                if (dbg) for (auto &m : marks) (void)hipEventDestroy(m.second);
                *fallback = true;
                return UKM_OK;
            }
            u64 nA = 0;
            for (u64 j = 0; j < span && j < (u64)S; j++) nA += lens[(size_t)j];
            u64 nm = 0;
            UKM_TRY(ukm_dev_setop2(c, UKM_OP_MERGE_INTERNAL, in_k, tax ? in_t : nullptr, nA, in_k + nA, tax ? in_t + nA : nullptr,
                                   N - nA, 0, out, tax ? tout : nullptr, out_cap, &nm));
            mark("top-2way");
            break;
        }
        u64 *cnt = nullptr;
        UKM_TRY(ws_alloc_t(c, (size_t)nodes * R, &cnt));
        KwArgs a;
        memset(&a, 0, sizeof(a));
        a.leaf_keys = d_keys;
        a.leaf_tax = tax ? d_tax : nullptr;
        a.in_keys = in_k;
        a.in_tax = in_t;
        a.in_cnt = in_cnt;
        a.out_keys = ok;
        a.out_tax = tax ? ot : nullptr;
        a.out_cnt = cnt;
        a.cuts = cuts;
        a.P = P;
        a.result = ctl;
        a.S = (u32)S;
        a.R = R;
        a.nprev = (u32)nprev;
        a.span = span;
        a.tax = ukm_taxdev(c);
        const u64 grid = nodes * R;
        if (grid > 0x7FFFFFFFull) UKM_FAIL(UKM_ERR_INVALID, "k-way merge: too many work items");
        if (lv == 0) (void)hipEventRecord(c->ev_k0, c->stream);
        if (K == 4) kw_launch_k<4, 2>(a, tax, uni, lv == 0, (unsigned)grid, c->stream);
        else if (K == 16) kw_launch_k<16, 4>(a, tax, uni, lv == 0, (unsigned)grid, c->stream);
        else kw_launch_k<8, 3>(a, tax, uni, lv == 0, (unsigned)grid, c->stream);
        if (lv == 0) {
            (void)hipEventRecord(c->ev_k1, c->stream);
            c->evk_valid = true;
        }
        UKM_HIP(hipGetLastError());
        mark(lv == 0 ? "level0" : (lv == 1 ? "level1" : "level2+"));
        if (lv == 0 && levels > 1 && N >= (1u << 22)) {
            // order and degenerate runs are checked while level 0 loads the leaves: a flagged input goes to the
            // caller's general route NOW instead of after the remaining levels and the compaction have moved all N
            // records again (one 20-us read-back against a level of >= 4e6 records)
            u64 fl = 0;
            UKM_TRY(ukm_read_u64(c, ctl + 1, &fl));
            if (fl & (KW_FLAG_UNSORTED | KW_FLAG_DEGENERATE)) {
                if (dbg) for (auto &m : marks) (void)hipEventDestroy(m.second);
                *fallback = true;
                return UKM_OK;
            }
        }
        in_k = ok;
        in_t = ot;
        in_cnt = cnt;
        last_cnt = cnt;
        last_k = ok;
        last_t = ot;
        nprev = nodes;
        span *= (u64)K;
    }
    u64 h[2] = {0, 0};
    if (uni) {
        // ranges of the root -> contiguous output
        u64 *excl = nullptr;
        UKM_TRY(ws_alloc_t(c, (size_t)R + 1, &excl));
        UKM_TRY(ukm_dev_exclusive_scan_u64(c, last_cnt, excl, R, ctl));  // ctl[0] = total
        const int parts = R >= 2048 ? 1 : (int)std::min<u64>(64, std::max<u64>(1, 2048 / R));
        hipLaunchKernelGGL(kw_compact_kernel, dim3(R * (unsigned)parts), dim3(256), 0, c->stream, last_k, tax ? last_t : nullptr,
                           P, (u32)S, R, last_cnt, excl, out, tout, out_cap, parts);
        UKM_HIP(hipGetLastError());
    }
    mark("compact");
    UKM_TRY(ukm_read_u64(c, ctl, h, 2));
    if (dbg) {
        fprintf(stderr, "[kway] S=%d N=%llu K=%d levels=%d R=%u ns=%llu flags=%llu out=%llu :", S, (unsigned long long)N, K,
                levels, R, (unsigned long long)ns, (unsigned long long)h[1], (unsigned long long)h[0]);
        for (size_t i = 1; i < marks.size(); i++) {
            float ms = 0;
            (void)hipEventElapsedTime(&ms, marks[i - 1].second, marks[i].second);
            fprintf(stderr, " %s=%.3fms", marks[i].first, ms);
        }
        fprintf(stderr, "\n");
        for (auto &m : marks) (void)hipEventDestroy(m.second);
    }
    if (h[1] & (KW_FLAG_UNSORTED | KW_FLAG_DEGENERATE)) {
        *fallback = true;
        return UKM_OK;
    }
    *n_out = uni ? h[0] : N;
    if (*n_out > out_cap)
        UKM_FAIL(UKM_ERR_CAPACITY, "k-way union: output needs %llu records, capacity is %llu", (unsigned long long)*n_out,
                 (unsigned long long)out_cap);
    return UKM_OK;
}
